"""Summarise rocprofv3 --pmc passes (one counter set per pass, as the MI355X guide prescribes) into a
small JSON/markdown that is committed under profiles/ and read back by bench.py for `roofline.traffic`.

  python tools/summarize_pmc.py gpurun_out/pmc_r1 profiles/r1_pmc_summary

Units: FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE
counts 128-byte requests as 64 bytes, i.e. reads HALF the bytes of fully coalesced >=128-B-per-row streams
(calibration below: global_max_concat_kernel reads a known 134.2 MB per launch at B=64 and reports 65.5 MiB),
but is exact for 64-byte requests - which is what the conv kernel issues (16 channels x 4 B per pixel-tap).
Both the raw and the x2 figure are reported; WRITE_SIZE is exact on every calibration kernel."""
import csv, json, sys, collections

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    for r in csv.DictReader(open(f"{src}/{n}_counter_collection.csv")):
        d = agg[r["Kernel_Name"]][r["Counter_Name"]]
        d[0] += 1; d[1] += float(r["Counter_Value"]); d[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for k, v in agg.items():
    kk = k.replace("void ", "").replace("(anonymous namespace)::", "")
    if not kk.startswith(("conv_igemm", "conv_x3", "conv_h2", "split_h2", "_ZN12_GLOBAL__N_1", "split_bf16x3", "stem_", "maxpool", "global_max", "upsample", "dense_glue", "groupnorm", "ransac", "wgrad", "bn_", "chan_", "conv_lp")):
        continue
    m = {c: s / n for c, (n, s, t) in v.items()}
    us = v["FETCH_SIZE"][2] / v["FETCH_SIZE"][0] / 1e3
    e = {"launches_profiled": v["FETCH_SIZE"][0], "avg_us": round(us, 1),
         "fetch_MB_raw": round(m["FETCH_SIZE"] * 1024 / 1e6, 1), "fetch_MB_x2": round(2 * m["FETCH_SIZE"] * 1024 / 1e6, 1),
         "write_MB": round(m["WRITE_SIZE"] * 1024 / 1e6, 1)}
    mops = m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) + m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) + m.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0)
    if mops > 0:
        xcd_cycles = m["GRBM_GUI_ACTIVE"] / 8.0  # the counter is summed over the 8 XCDs
        e["mfma_gflop_issued"] = round(mops * 512 / 1e9, 2)
        e["mfma_util_pct"] = round(100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (xcd_cycles * 1024), 1)
        e["clock_GHz"] = round(xcd_cycles / (v["GRBM_GUI_ACTIVE"][2] / v["GRBM_GUI_ACTIVE"][0]), 2)
    out[k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]] = e
json.dump(out, open(dst + ".json", "w"), indent=1)
with open(dst + ".md", "w") as f:
    f.write("| kernel | launches | avg us | FETCH raw MB | FETCH x2 MB | WRITE MB | MFMA GFLOP issued | MfmaUtil % | clock GHz |\n|---|---|---|---|---|---|---|---|---|\n")
    for k, e in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_profiled"]):
        f.write(f"| `{k}` | {e['launches_profiled']} | {e['avg_us']} | {e['fetch_MB_raw']} | {e['fetch_MB_x2']} | {e['write_MB']} | "
                f"{e.get('mfma_gflop_issued','')} | {e.get('mfma_util_pct','')} | {e.get('clock_GHz','')} |\n")
print(open(dst + ".md").read())
