"""fp32 inference with the ResNet-50 (Bottleneck) trunk at B=64, with and without the bf16x3 kernels (cfg.TEST.BF16X3)."""
import sys, time, torch, numpy as np
import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from rdpn6d_amd import synth
from rdpn6d_amd.config import gdrn_base_cfg
from rdpn6d_amd.gdrn import build_model_optimizer
dev = torch.device('cuda:0')
for x3 in (True, False):
    cfg = gdrn_base_cfg(mask_attention='none', device='cuda')
    cfg.MODEL.CDPN.BACKBONE.NUM_LAYERS = 50
    cfg.TEST.BF16X3 = x3
    model, _ = build_model_optimizer(cfg)
    model.eval()
    B = 64
    t = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(B, seed=0).items()}
    def step():
        with torch.no_grad():
            return model(t['roi_img'], roi_classes=t['roi_cls'], roi_coord_2d=t['roi_coord_2d'], roi_cams=t['roi_cam'], roi_centers=t['roi_center'],
                         roi_whs=t['roi_wh'], roi_extents=t['roi_extent'], resize_ratios=t['resize_ratio'], do_loss=False, fps=t['fps'])
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"ResNet-50 trunk, B=64, fp32, BF16X3={x3}: {dt*1e3:.2f} ms/step = {B/dt:.0f} crops/s (x3 launches {model.plan(B, dev).x3_launches})")
    del model; torch.cuda.empty_cache()
