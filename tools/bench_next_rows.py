"""Throughput of the SURVEY 8f "next" rows and of the side kernels of the path, with their HBM roofline and the CPU oracle
beside them (bounded samples).  Prints one JSON object per row; run on the GPU box:
    python tools/bench_next_rows.py > gpurun_out/next_rows.jsonl
"""
import ctypes, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from rdpn6d_amd import _lib, ops, synth          # noqa: E402
from rdpn6d_amd.crop import build_crops           # noqa: E402
from rdpn6d_amd.gdrn import _ptr                  # noqa: E402

HBM = 8000.0  # GB/s, MI355X_MICROARCH.md
dev = torch.device("cuda:0")
lib = _lib.load()


def gpu_time(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def cpu_time(fn, budget=3.0):
    fn()
    t0, it = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget:
        fn()
        it += 1
    return (time.perf_counter() - t0) / max(it, 1)


def row(name, units, unit_name, sec, bytes_algo, cpu_sec=None, cpu_units=None, note=""):
    r = {"row": name, "value": round(units / sec, 1), "unit": unit_name + "/s", "ms": round(sec * 1e3, 4),
         "roofline": {"bound": "hbm", "achieved": round(bytes_algo / sec / 1e9, 1), "peak": HBM, "unit": "GB/s",
                      "frac": round(bytes_algo / sec / 1e9 / HBM, 4)}}
    if cpu_sec is not None:
        r["cpu_baseline"] = {"value": round((cpu_units or units) / cpu_sec, 2), "unit": unit_name + "/s", "kind": "port"}
    if note:
        r["note"] = note
    print(json.dumps(r), flush=True)


# ---- 1. crop builder: 64 crops out of 8 VGA frames (data_loader.py:523-627)
from oracle import crop_oracle  # noqa: E402  (test infrastructure: CPU baseline only)
rng = np.random.default_rng(0)
N, H, W, B = 8, 480, 640, 64
img = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
depth = (0.5 + rng.random((N, H, W), dtype=np.float32)).astype(np.float32)
x0, y0 = rng.uniform(0, 400, B), rng.uniform(0, 280, B)
wh = rng.uniform(40, 200, (B, 2))
boxes = np.stack([x0, y0, x0 + wh[:, 0], y0 + wh[:, 1]], 1)
idx = rng.integers(0, N, B)
cams = np.stack([synth.LM_K.astype(np.float32)] * B)
dimg, ddep = torch.from_numpy(img).to(dev), torch.from_numpy(depth).to(dev)
sec = gpu_time(lambda: build_crops(dimg, ddep, idx, boxes, cams), n=10)
out_bytes = B * (6 * 256 * 256 + 5 * 64 * 64) * 4
c = np.array([0.5 * (boxes[0, 0] + boxes[0, 2]), 0.5 * (boxes[0, 1] + boxes[0, 3])])
sc = min(max(wh[0, 0], wh[0, 1], 1) * 1.5, 640) * 1.0
cpu = cpu_time(lambda: crop_oracle.build_roi(img[idx[0]], depth[idx[0]], cams[0], c, sc), 3.0)
Bl = 1024
idx_l, boxes_l, cams_l = np.tile(idx, Bl // B), np.tile(boxes, (Bl // B, 1)), np.tile(cams, (Bl // B, 1, 1))
secl = gpu_time(lambda: build_crops(dimg, ddep, idx_l, boxes_l, cams_l), n=5)
row("crop builder (frames + boxes -> roi_img, roi_coord_2d; incl. the host-side affine set-up)", B, "crops", sec,
    out_bytes + B * 260 * 260 * 7, cpu, 1, "algorithmic bytes: outputs + the source window of every crop (uint8 RGB + fp32 depth); 64 crops = "
    f"136 MB in {sec * 1e6:.0f} us includes the host-side affine set-up and its small H2D copy per call; {Bl} crops in one call: "
    f"{secl * 1e3:.3f} ms = {(Bl * (6 * 256 * 256 + 5 * 64 * 64) * 4 + Bl * 260 * 260 * 7) / secl / 1e9:.0f} GB/s = "
    f"{(Bl * (6 * 256 * 256 + 5 * 64 * 64) * 4 + Bl * 260 * 260 * 7) / secl / 1e9 / HBM:.2f} of the HBM peak")

# ---- 3. training targets: nearest anchor + residual (data_utils.py:229-244, data_loader.py:881-903)
from oracle import targets_eval_oracle as teo  # noqa: E402
Bt, K = 64, 32
xyz = torch.randn(Bt, 64, 64, 3, device=dev) * 0.05
fps64 = torch.randn(Bt, K, 3, dtype=torch.float64, device=dev) * 0.05
rot = torch.eye(3, device=dev).repeat(Bt, 1, 1)
ext = torch.rand(Bt, 3, device=dev) * 0.2 + 0.05
sec = gpu_time(lambda: ops.region_targets(xyz, fps64, rot, ext))
Bl = 2048  # the same kernel on a batch that outlasts the launch floor
xyzl, fpsl = torch.randn(Bl, 64, 64, 3, device=dev) * 0.05, torch.randn(Bl, K, 3, dtype=torch.float64, device=dev) * 0.05
rotl, extl = torch.eye(3, device=dev).repeat(Bl, 1, 1), torch.rand(Bl, 3, device=dev) * 0.2 + 0.05
secl = gpu_time(lambda: ops.region_targets(xyzl, fpsl, rotl, extl))
row("training targets (region labels + residual xyz)", Bt, "crops", sec, Bt * 4096 * (12 + 12 + 8),
    note=f"64 crops = 8.4 MB: a {sec * 1e6:.0f}-us launch sits on the launch floor, not on HBM; {Bl} crops in one launch: {secl * 1e3:.3f} ms = "
    f"{Bl * 4096 * 32 / secl / 1e9:.0f} GB/s = {Bl * 4096 * 32 / secl / 1e9 / HBM:.2f} of the HBM peak")

# ---- 4. ADD / ADI / re / te (lib/pysixd/pose_error.py:297-337,400-436)
Be, n = 256, 3000
Re, Rg = torch.eye(3, device=dev).repeat(Be, 1, 1), torch.eye(3, device=dev).repeat(Be, 1, 1)
te_, tg = torch.rand(Be, 3, device=dev), torch.rand(Be, 3, device=dev)
pts = torch.randn(n, 3, device=dev) * 0.05
sec = gpu_time(lambda: ops.pose_errors(Re, te_, Rg, tg, pts), n=5)
r = {"row": "ADD / ADI / re / te in fp64 (ADI = exact brute-force nearest neighbour over 3000 model points)", "value": round(Be / sec, 1),
     "unit": "poses/s", "ms": round(sec * 1e3, 3),
     "roofline": {"bound": "fp64 VALU", "achieved": round(Be * n * n * 9 / sec / 1e12, 2), "peak": 78.6, "unit": "TFLOP/s",
                  "frac": round(Be * n * n * 9 / sec / 1e12 / 78.6, 4)}}
print(json.dumps(r), flush=True)

# ---- FPS (A11): 32 anchors of a 50k-point cloud, host-pointer ABI (includes H2D / D2H) and the device-resident batched form
pts_h = rng.normal(size=(50000, 3)).astype(np.float32)
t0 = time.perf_counter()
for _ in range(5):
    ops.farthest_point_sampling(pts_h, 32, init_center=True)
sec = (time.perf_counter() - t0) / 5
so = os.path.join(ROOT, "oracle", "liboracle.so")
cpu = None
if os.path.exists(so):
    ol = ctypes.CDLL(so)
    idxs = np.zeros(32, np.int32)
    P = ctypes.c_void_p
    cpu = cpu_time(lambda: ol.oracle_fps_init_center(pts_h.ctypes.data_as(P), idxs.ctypes.data_as(P), 50000, 32), 2.0)
row("fps, host ABI (50 000 points, 32 samples, init_center; includes the copies)", 1, "clouds", sec, 50000 * 12, cpu,
    note="the reference's ABI hands over host pointers: 4 hipMalloc + 2 copies + 1 launch + 1 copy per call dominate; the kernel alone is the next row")
# device-resident: one launch, the cloud on ceil(N / 16384) workgroups (points in registers) vs one workgroup streaming it from L2
d_pts = torch.from_numpy(pts_h).to(dev)
d_off = torch.tensor([0, 50000], dtype=torch.int32, device=dev)
d_idx, d_md = torch.zeros(32, dtype=torch.int32, device=dev), torch.empty(50000, device=dev)
d_ws = torch.zeros(int(lib.rdpn6d_fps_workspace_bytes(1)), dtype=torch.uint8, device=dev)
stf = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
sec1 = gpu_time(lambda: lib.rdpn6d_fps_device(_ptr(d_pts), _ptr(d_off), 1, 50000, 32, -1, _ptr(d_idx), _ptr(d_md), stf()))
secm = gpu_time(lambda: lib.rdpn6d_fps_device_ws(_ptr(d_pts), _ptr(d_off), 1, 50000, 32, -1, _ptr(d_idx), _ptr(d_md), _ptr(d_ws), d_ws.numel(), stf()))
row("fps, device-resident (50 000 points, 32 samples): 4 workgroups, points in registers, one barrier per sample", 1, "clouds", secm,
    50000 * 12, note=f"one workgroup streaming the cloud from L2 every sample: {sec1 * 1e3:.3f} ms; sequential by construction (32 dependent "
    "arg-max rounds): latency bound, bytes = the cloud once")

# ---- RANSAC / Kabsch: 64 crops, 100 hypotheses each, on synthetic maps
from tests.ransac_cases import make_case  # noqa: E402
c = make_case(B=64, outliers=0.3, seed=1)
t = {k: torch.from_numpy(np.ascontiguousarray(c[k])).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
sec = gpu_time(lambda: ops.ransac_kabsch(t["out_nchw"].reshape(64, 37, 64, 64), t["coord2d"], t["fps"], t["extents"], t["ratios"], t["argmax"]))
cpu = None
if os.path.exists(so):
    from tests.test_ransac_oracle import run_oracle  # noqa: E402
    c4 = make_case(B=4, outliers=0.3, seed=1)
    cpu = cpu_time(lambda: run_oracle(ol, c4), 3.0)
row("RANSAC + Kabsch (100 hypotheses per crop, scoring from LDS)", 64, "crops", sec, 64 * 4096 * 20, cpu, 4,
    "bytes: 20 B per correspondence once (SURVEY 8d); the kernel is latency / ALU bound, not HBM bound")

# ---- A8: correspondence selection (gdrn_evaluator.py:89-126) on 64 crops of model-shaped maps
from oracle import select_oracle  # noqa: E402
maps_h = rng.standard_normal((64, 37, 64, 64)).astype(np.float32)
maps_h[:, 1:4] = rng.random((64, 3, 64, 64), dtype=np.float32)
c5_h = rng.random((64, 5, 64, 64), dtype=np.float32)
ext_h = (rng.random((64, 3), dtype=np.float32) * 0.2 + 0.05).astype(np.float32)
maps_d, c5_d, ext_d = torch.from_numpy(maps_h).to(dev), torch.from_numpy(c5_h).to(dev), torch.from_numpy(ext_h).to(dev)
sec = gpu_time(lambda: ops.select_correspondences(maps_d, c5_d, ext_d, 480, 640))
nm0 = select_oracle.out_mask_l1(maps_h[:1, :1])
cpu = cpu_time(lambda: select_oracle.select_correspondences(nm0[0, 0], maps_h[0, 1:4].transpose(1, 2, 0), c5_h[0, 3:5].transpose(1, 2, 0), 480, 640, ext_h[0]), 2.0)
Bl = 1024
maps_l = torch.from_numpy(np.tile(maps_h, (Bl // 64, 1, 1, 1))).to(dev)
c5_l, ext_l = torch.from_numpy(np.tile(c5_h, (Bl // 64, 1, 1, 1))).to(dev), torch.from_numpy(np.tile(ext_h, (Bl // 64, 1))).to(dev)
secl = gpu_time(lambda: ops.select_correspondences(maps_l, c5_l, ext_l, 480, 640))
row("correspondence selection A8 (mask / coordinate filter + ordered compaction)", 64, "crops", sec, 64 * 4096 * (6 * 4 + 20), cpu, 1,
    note=f"64 crops = 11.5 MB: a {sec * 1e6:.0f}-us launch (one workgroup per crop, ordered compaction = two passes + a scan) is launch / "
    f"latency bound; {Bl} crops in one launch: {secl * 1e3:.3f} ms = {Bl * 4096 * 44 / secl / 1e9:.0f} GB/s = "
    f"{Bl * 4096 * 44 / secl / 1e9 / HBM:.2f} of the HBM peak")

# ---- A9: 2D-3D RANSAC-PnP, 64 crops x 100 hypotheses (P3P per wavefront, reprojection scoring from LDS, Gauss-Newton refit)
from tests.pnp_cases import make_pnp_case  # noqa: E402
cp = make_pnp_case(B=64, n=1600, outliers=0.3, seed=2)
tp = {k: torch.from_numpy(np.ascontiguousarray(cp[k])).to(dev) for k in ("image_points", "model_points", "counts", "cams")}
sec = gpu_time(lambda: ops.ransac_pnp(tp["image_points"], tp["model_points"], tp["counts"], tp["cams"].reshape(64, 3, 3)))
cpu = None
if os.path.exists(so):
    from tests.test_pnp_oracle import run_pnp_oracle  # noqa: E402
    cp4 = make_pnp_case(B=4, n=1600, outliers=0.3, seed=2)
    cpu = cpu_time(lambda: run_pnp_oracle(ol, cp4), 3.0)
row("2D-3D RANSAC-PnP (100 P3P hypotheses per crop in fp64, 3 px reprojection scoring from LDS, Gauss-Newton refit)", 64, "crops", sec,
    64 * 1600 * 20, cpu, 4, "latency / fp64-ALU bound; bytes: 20 B per correspondence once")

# ---- A9 with the reference call's own minimal solver: EPnP on sets of five + EPnP refit (cfg.TEST.PNP_MINIMAL = "epnp", round 6)
sec = gpu_time(lambda: ops.ransac_pnp(tp["image_points"], tp["model_points"], tp["counts"], tp["cams"].reshape(64, 3, 3), minimal="epnp"))
cpu = None
if os.path.exists(so):
    cpu = cpu_time(lambda: run_pnp_oracle(ol, cp4, minimal="epnp"), 3.0)
row("2D-3D RANSAC-PnP, EPnP minimal solver (100 five-point EPnP hypotheses per crop in fp64: 12x12 Jacobi on an LDS scratch per wavefront; EPnP refit "
    "on the inliers)", 64, "crops", sec, 64 * 1600 * 20, cpu, 4, "latency / fp64-ALU bound; bytes: 20 B per correspondence once")
