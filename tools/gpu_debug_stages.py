"""GPU-box diagnostic: run the HIP plan launch by launch and compare every stage with the torch-CPU
oracle evaluated in fp64 ("truth") and in fp32 (what the reference's own arithmetic gives)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import synth, _lib
from rdpn6d_amd.config import gdrn_base_cfg
from rdpn6d_amd.gdrn import build_model_optimizer, _ptr
from oracle import model_oracle
import ctypes

dev = torch.device("cuda:0")
gold_dir = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
bn = np.load(os.path.join(gold_dir, "bn_stats_c1.npz"))
inp = synth.make_inputs(4, seed=0)
model, _ = build_model_optimizer(gdrn_base_cfg(device="cuda"))
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
sd.update({k: bn[k] for k in bn.files})
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
model.load_state_dict(sd); model.eval()
o32 = model_oracle.GDRNOracle(32, "none"); o32.load_state_dict(sd); o32.eval()
o64 = model_oracle.GDRNOracle(32, "none"); o64.load_state_dict(sd); o64.double().eval()

def stages(orc, x):
    out = {}
    bb = orc.backbone
    y = bb.relu(bb.bn1(bb.conv1(x[:, :3]))); out["stem"] = y
    y = bb.maxpool(y); out["maxpool"] = y
    for li in range(4):
        for bi, blk in enumerate(getattr(bb, f"layer{li+1}")):
            y = blk(y); out[f"layer{li+1}.{bi}.conv2"] = y
    y = torch.nn.functional.interpolate(y, scale_factor=4, mode="bilinear", align_corners=True); out["upsample"] = y
    xyz = x[:, 3:, ::8, ::8]
    f = bb.spatial_net(y, xyz); out["global_max_concat"] = f
    h = f
    for i, l in enumerate(orc.rot_head_net.features):
        h = l(h)
        if i == 2: out["rot_head.convT.phase11"] = h
        elif i % 3 == 2: out[f"rot_head.features.{i-2}"] = h
    out["rot_head.out"] = h
    return out

t = {k: torch.from_numpy(v) for k, v in inp.items()}
with torch.no_grad():
    s32 = stages(o32, t["roi_img"]); s64 = stages(o64, t["roi_img"].double())
plan = model.plan(4, dev)
x = t["roi_img"].to(dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = plan.lib
lib.rdpn6d_stem_conv7x7_f32(_ptr(x), 4, 6, plan.R, *plan.stem_args[3:], st)
lib.rdpn6d_xyz_subsample_f32(_ptr(x), 4, 6, plan.R, *plan.xyz_args[3:], st)
def cmp(name, buf):
    if name not in s64: return
    ref64 = s64[name]; ref32 = s32[name]
    c = ref64.shape[1]
    mine = buf.reshape(4, ref64.shape[2], ref64.shape[3], -1)[..., :c].permute(0, 3, 1, 2).cpu().double()
    e_mine = (mine - ref64).abs().max().item(); e_cpu = (ref32.double() - ref64).abs().max().item()
    print(f"{name:32s} |ref|max {ref64.abs().max().item():8.3f}  err(HIP vs fp64) {e_mine:.3e}  err(CPU fp32 vs fp64) {e_cpu:.3e}  HIP vs CPU32 {(mine-ref32.double()).abs().max().item():.3e}")
torch.cuda.synchronize(); cmp("stem", plan.bufs["stem"])
import re
for L in plan.launches:
    L.fn(*L.args, st); torch.cuda.synchronize()
    d = L.keep[0] if L.keep else None
    if L.name == "maxpool": cmp("maxpool", plan.bufs["pool"])
    elif L.name == "upsample": cmp("upsample", plan.bufs["up"])
    elif L.name == "global_max_concat": cmp("global_max_concat", plan.bufs["feat"])
    elif d is not None and L.name in s64:
        # find the output buffer by pointer
        for bname, b in plan.bufs.items():
            if b.data_ptr() == d.y:
                cmp(L.name, b); break
