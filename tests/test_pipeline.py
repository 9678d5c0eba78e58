"""The hot path with its two neighbours, end to end ON DEVICE: full frames + boxes -> GPU crop builder ->
GDRN forward + per-crop RANSAC (the path) -> ADD / ADI / re / te.  Weights are the seeded random ones, so
the poses mean nothing; what is checked is the plumbing - every stage consumes the previous stage's device tensors as they
are, and each stage's output equals its oracle fed the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_frames_to_pose_errors_on_device(golden_dir):
    import os

    from oracle import crop_oracle, model_oracle, targets_eval_oracle as teo
    from rdpn6d_amd import ops, synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.crop import build_crops
    from rdpn6d_amd.gdrn import build_model_optimizer
    from tests.test_crop import _frames

    dev = torch.device("cuda:0")
    B = 4
    img, depth = _frames(7, 2)
    boxes = np.array([[200.0, 150.0, 330.0, 260.0], [60.0, 40.0, 190.0, 200.0], [400.0, 250.0, 560.0, 420.0], [300.5, 100.25, 380.75, 231.0]])
    idx = np.array([0, 1, 1, 0])
    cams = np.stack([synth.LM_K.astype(np.float32)] * B)
    crops = build_crops(torch.from_numpy(img).to(dev), torch.from_numpy(depth).to(dev), idx, boxes, cams)

    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
    cfg.TEST.USE_PNP = True
    model, _ = build_model_optimizer(cfg)
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1.npz"))
    sd.update({k: bn[k] for k in bn.files})
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    model.eval()
    rng = np.random.Generator(np.random.PCG64(3))
    extents = (rng.random((B, 3)) * 0.2 + 0.05).astype(np.float32)
    fps = np.stack([synth.ellipsoid_points(tuple(extents[i]), 32, 10 + i) for i in range(B)]).astype(np.float32)
    t_cams = torch.from_numpy(cams).to(dev)
    with torch.no_grad():
        out = model(crops["roi_img"], roi_classes=torch.zeros(B, dtype=torch.long, device=dev), roi_coord_2d=crops["roi_coord_2d"],
                    roi_cams=t_cams, roi_centers=crops["bbox_center"], roi_whs=crops["roi_wh"], roi_extents=torch.from_numpy(extents).to(dev),
                    resize_ratios=crops["resize_ratio"], do_loss=False, fps=torch.from_numpy(fps).to(dev),
                    im_H=img.shape[1], im_W=img.shape[2])  # the frames' own size (the batch's im_H / im_W, engine_utils.py:71)
    torch.cuda.synchronize()
    assert all(out[k].is_cuda for k in ("rot", "trans", "mask", "region", "pnp_pose"))

    # stage 1 == its oracle; stage 2 on the oracle-built crops == oracle model (the usual fp32 yardstick is in test_gpu_kernels)
    oc = [crop_oracle.build_roi(img[idx[i]], depth[idx[i]], cams[i], np.array([0.5 * (boxes[i, 0] + boxes[i, 2]), 0.5 * (boxes[i, 1] + boxes[i, 3])]),
                                min(max(boxes[i, 2] - boxes[i, 0], boxes[i, 3] - boxes[i, 1], 1) * 1.5, 640) * 1.0) for i in range(B)]
    roi_img = np.stack([o[0] for o in oc]).astype(np.float32)
    assert np.abs(crops["roi_img"].cpu().numpy() - roi_img).max() <= 1e-5 * max(1.0, np.abs(roi_img).max())
    orc = model_oracle.GDRNOracle(32, "mul")
    orc.load_state_dict(sd, strict=True)
    orc.eval()
    with torch.no_grad():
        oo = orc(crops["roi_img"].cpu(), crops["roi_coord_2d"].cpu(), torch.from_numpy(fps), torch.from_numpy(cams), crops["bbox_center"].cpu(),
                 crops["roi_wh"].cpu(), crops["resize_ratio"].cpu())
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        assert (out[k].cpu() - oo[k]).abs().max().item() < 5e-3, k

    # stage 3: ADD / ADI / re / te of the network pose against a synthetic ground truth, on device vs the numpy restatement
    pts = np.stack([synth.ellipsoid_points(tuple(extents[i]), 500, 50 + i) for i in range(B)]).astype(np.float32)
    R_gt = np.stack([np.linalg.qr(rng.standard_normal((3, 3)))[0] for _ in range(B)]).astype(np.float32)
    R_gt *= np.sign(np.linalg.det(R_gt))[:, None, None]
    t_gt = np.stack([rng.random(B) * 0.2 - 0.1, rng.random(B) * 0.2 - 0.1, rng.random(B) + 0.5], 1).astype(np.float32)
    err = ops.pose_errors(out["rot"], out["trans"], torch.from_numpy(R_gt).to(dev), torch.from_numpy(t_gt).to(dev), torch.from_numpy(pts).to(dev))
    torch.cuda.synchronize()
    Re, te_ = out["rot"].cpu().numpy().astype(np.float64), out["trans"].cpu().numpy().astype(np.float64)
    for i in range(B):
        want = [teo.add(Re[i], te_[i], R_gt[i].astype(np.float64), t_gt[i].astype(np.float64), pts[i].astype(np.float64)),
                teo.adi(Re[i], te_[i], R_gt[i].astype(np.float64), t_gt[i].astype(np.float64), pts[i].astype(np.float64)),
                teo.re(Re[i], R_gt[i].astype(np.float64)), teo.te(te_[i], t_gt[i].astype(np.float64))]
        got = err[i].cpu().numpy()
        assert np.allclose(got, want, rtol=1e-6, atol=1e-9), (i, got, want)


def test_bench_contract_two_ranks_on_one_gpu():
    """the driver's multi-GPU launch line (torch.distributed.run, one rank per GPU, barrier + max-over-ranks timing, ONE JSON
    line from rank 0) exercised with two ranks sharing this box's GPU; RCCL cannot put two ranks on one device, so the
    rendezvous uses gloo here (RDPN6D_BENCH_BACKEND) - everything else is the code path of a real N-GPU run."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RDPN6D_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 16
    assert "batch=8 per GPU" in out["config"]["workload"] and len(out["per_rank_crops_per_s"]) == 2 and min(out["per_rank_crops_per_s"]) > 0
    assert out["value"] > 0 and abs(out["value"] - 16 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-2 * out["value"]
    assert out["roofline"]["frac"] > 0 and "cpu_baseline" not in out


@pytest.mark.parametrize("extra", [[], ["--train", "--dtype", "bf16"]])
def test_bench_gpus_flag_launches_its_own_ranks(extra):
    """`python bench.py --gpus 2` started BARE (no launcher, no WORLD_SIZE): the process must start two ranks itself (a fresh
    torch.distributed.run child) and print ONE line with n_gpus == 2.  With --train --dtype bf16 the two ranks
    also run the event-timed roofline steps and the pre-heat blocks, whose gradient all-reduces hang unless every rank runs the
    same number of them (bench.py:367)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RDPN6D_BENCH_BACKEND="gloo", RDPN6D_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--preheat", "0.5",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world"] == 2 and out["backend"] == "gloo" and out["device_count"] >= 1
    assert out["config"]["global_batch"] == 16 and out["value"] > 0
    if extra:  # (at 8 crops per rank no layer takes the 256x256 kernel the training roofline times: the key is there, possibly None)
        assert "roofline" in out and out["steps"] == 3


def test_bench_gpus_flag_must_match_the_launcher():
    """--gpus 2 under a launcher that started ONE rank is an error, not a silent one-rank run labelled n_gpus 1"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


@pytest.mark.parametrize("train", [False, True])
def test_bench_contract_over_rccl_one_rank(train):
    """the same launch line with the backend of a real run - RCCL ("nccl") - on the one GPU this box has: process-group creation
    bound to the device, the barriers around the timed region, the all_gather / all_reduce(MAX) of the per-rank times on device
    tensors and (--train) the bucketed gradient all-reduces all go through RCCL."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RDPN6D_BENCH_FORCE_DIST="1")
    env.pop("RDPN6D_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline"] + (["--train"] if train else [])
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["global_batch"] == 8


@pytest.mark.parametrize("extra,check", [
    (["--cam", "ycbv", "--mask-attention", "mul", "--batch", "16", "--test-cfg", "FOLD_GLOBAL_MAX=0,COMPOSE_CONV3_CONVT=0"], "YCB-V"),
    (["--train", "--dtype", "fp16", "--backbone", "50", "--res", "320", "--batch", "4"], "ResNet-50"),
])
def test_bench_secondary_lines(extra, check):
    """the BASELINE-shaped secondary lines of bench.py (C4: YCB-V intrinsics; C5: ResNet-50 / 320x320 / fp16 training) and the
    --test-cfg switch run and label themselves"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] > 0 and check in out["config"]["workload"]
    if "--train" not in extra:
        assert "flops_note" in out and "does not execute" not in out["flops_note"]  # the rewrites were switched off


def test_bench_default_line_carries_the_training_leg():
    """the driver's ONE command (`python bench.py`, batch 64, fp32-accurate inference) also records the training
    step: after the headline's timed region a short bf16 AMP B = 32 leg runs and lands under "train" - never in `value`.  (Short
    windows here; the contract keys are what is checked: the headline's metric / roofline / the train leg's step time, FLOP
    accounting, dominant kernel and the per-bucket issue times in the backward's completion order.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--preheat", "0.3", "--train-leg", "0.3",
                        "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE JSON line
    out = json.loads(lines[0])
    assert out["metric"].startswith("RGB-D crops/sec (fwd+PnP)") and out["n_gpus"] == 1 and out["config"]["batch_per_gpu"] == 64
    assert out["roofline"]["kernel"].startswith("conv_h2_8ph_kernel_t") and 1.0 < out["roofline"]["clock_ghz"] < 2.6
    t = out["train"]
    assert t["batch_per_gpu"] == 32 and t["steps"] == 60 and t["ms_per_step"] > 0 and t["dtype"].startswith("bf16")
    assert abs(t["crops_per_s"] - 32 * 1e3 / t["ms_per_step"]) < 0.01 * t["crops_per_s"]
    assert abs(t["tflops"] - 132.3e-3 * t["crops_per_s"]) < 0.5 and abs(t["frac_of_2500"] - t["tflops"] / 2500.0) < 1e-3
    assert t["dominant_kernel"]["kernel"].startswith("conv_igemm_bf16_8ph_kernel")
    names = [b["name"] for b in t["gradient_sync"]["buckets"]]
    assert names == ["pnp_net", "rot_head_net", "backbone.layer4", "backbone.layer3", "backbone.rest"]
    ahead = [b["issued_ms_before_backward_end"] for b in t["gradient_sync"]["buckets"]]
    assert all(x >= y for x, y in zip(ahead, ahead[1:])) and ahead[2] > 0.5 and ahead[-1] < 0.1  # layer4 well before the end, the tail at it
    # --train-leg 0 switches it off
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--preheat", "0", "--train-leg", "0",
                        "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "train" not in json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
