"""Compile the gfx950 kernels + C ABI into rdpn6d_amd/librdpn6d_hip.so (in-tree, so that the
built library travels with the repo snapshot to the GPU box).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdpn6d_hip.so")
SOURCES = ["api.cpp", "conv_igemm.hip", "conv_igemm_bf16.hip", "conv_igemm_bf16_8ph.hip", "conv_igemm_bf16_pp.hip", "conv_igemm_bf16x3.hip", "conv_igemm_bf16x3_tile.hip", "conv_igemm_h2.hip", "conv_igemm_h2_pp.hip", "pointwise.hip", "pointwise_bf16.hip", "pointwise_h2.hip", "fps.hip", "ransac.hip", "train_norm.hip", "train_wgrad.hip", "train_misc.hip", "ranger.hip", "targets_eval.hip", "crop_builder.hip", "pnp.hip"]
NO_CONTRACT = {"fps.hip", "ransac.hip", "targets_eval.hip", "crop_builder.hip", "pnp.hip"}  # bit-exact integer outputs depend on un-fused fp32 arithmetic
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# The 16-bit kernels are written once and built twice (csrc/common.h): bf16 with the sources above, IEEE fp16 - the reference's
# AMP dtype - by compiling this group again with -DRDPN6D_LP_FP16.  The fp16 objects are merged (ld -r), their rdpn6d_*_bf16
# entry points renamed to rdpn6d_*_fp16 and every other symbol made local (llvm-objcopy), so the two builds never meet.
LP_SOURCES = ["conv_igemm_bf16.hip", "conv_igemm_bf16_8ph.hip", "conv_igemm_bf16_pp.hip", "pointwise_bf16.hip", "pointwise.hip", "train_norm.hip", "train_misc.hip",
              "train_wgrad.hip"]
LP_EXTRA_RENAMES = {"rdpn6d_repack_f32": "rdpn6d_repack_fp16"}  # writes 16-bit weight mirrors next to the fp32 packed weights
if os.environ.get("RDPN6D_PROBE"):  # timing-only ablation variants of the kernels (tools/, never the shipped build)
    FLAGS.append("-DRDPN6D_PROBE")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "rdpn6d.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src + ".o")
        objs.append(obj)
        extra = ["-ffp-contract=off"] if src in NO_CONTRACT else []
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    lp_objs = []
    for src in LP_SOURCES:
        obj = os.path.join(HERE, "build", "fp16_" + src + ".o")
        lp_objs.append(obj)
        extra = ["-ffp-contract=off"] if src in NO_CONTRACT else []
        cmd = [hipcc] + FLAGS + extra + ["-DRDPN6D_LP_FP16", "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src + " [fp16]", subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    objs.append(_fp16_object(lp_objs))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    return LIB


def _fp16_object(lp_objs):
    """merge the -DRDPN6D_LP_FP16 objects, rename their 16-bit entry points *_bf16 -> *_fp16, localise everything else"""
    llvm = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    merged = os.path.join(HERE, "build", "lp_fp16_merged.o")
    final = os.path.join(HERE, "build", "lp_fp16.o")
    # --force-group-allocation: template instantiations (kernel host stubs, their handles) sit in COMDAT groups that the final
    # link would otherwise de-duplicate BY NAME against the bf16 objects' copies - and launch the bf16 kernel for the fp16 entry
    subprocess.run(["ld", "-r", "--force-group-allocation", "-o", merged] + lp_objs, check=True)
    nm = subprocess.run(["nm", "--defined-only", "--extern-only", merged], check=True, capture_output=True, text=True).stdout  # (binutils)
    renames = dict(LP_EXTRA_RENAMES)
    for line in nm.splitlines():
        sym = line.split()[-1]
        if sym.startswith("rdpn6d_") and "bf16" in sym and "bf16x3" not in sym:
            renames[sym] = sym.replace("bf16", "fp16")
    rfile, kfile = merged + ".renames", merged + ".keep"
    with open(rfile, "w") as f:
        f.write("".join(f"{a} {b}\n" for a, b in sorted(renames.items())))
    with open(kfile, "w") as f:
        f.write("".join(b + "\n" for b in sorted(renames.values())))
    objcopy = os.path.join(llvm, "llvm-objcopy")
    subprocess.run([objcopy, f"--redefine-syms={rfile}", merged, final + ".tmp"], check=True)
    subprocess.run([objcopy, f"--keep-global-symbols={kfile}", final + ".tmp", final], check=True)
    return final


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
