"""BUILD-CONTAINER ONLY: training-step golden values of the REAL reference (/root/reference) on EIGHT UNSEARCHED input seeds.

  python tools/oracle/gen_train_golden_seeds.py        # writes tests/golden/train_c1w_seeds.npz

model_c1w.npz holds the reference's nine losses and 164 gradients on ONE training batch whose seed was searched for the absence of
region arg-max ties (synth.C1W_TRAIN_INPUT_SEED).  This fixture runs the same capture (the reference built by its own factory from
its own config, strict load of the c1w weights, model.train(), forward(do_loss=True) + backward) on synth.make_inputs(4, seed=0..7)
without looking at the batches first, MASK_ATTENTION none and mul, and records next to every value what the tie rule of
tests/c1w_cases.py needs:

  * ``s{seed}_{att}_{loss}``            - the nine losses (8 threads) and ``..._noise_{loss}`` = |1 thread - 8 threads|;
  * ``s{seed}_{att}_gradnorm/<param>``  - ||grad|| of every parameter (float64) and ``..._gradnoise/<param>`` (1 vs 8 threads, relative);
  * ``s{seed}_argmax``, ``s{seed}_top2_gap`` - train-mode region arg-max (what feeds the pose branch) and the top-2 region-logit gap
    per pixel; ``s{seed}_flip_1v8`` - pixels the reference's own arg-max changes between 1 and 8 threads (packed bits).
    Dense losses (mask / xyz / region) are continuous in the maps; the pose-branch losses (PM_R, centroid, z) see the maps through the
    arg-max: the GPU test holds them to the bare tolerance on the seeds whose arg-max the HIP step reproduces exactly and otherwise
    only outside the recorded tie set.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_stubs  # noqa: E402

ref_stubs.install()

from rdpn6d_amd import synth  # noqa: E402
from gen_model_golden import GOLD, build_reference  # noqa: E402
from gen_model_golden_w import ref_train  # noqa: E402
from tests.c1w_cases import SEEDS, TIE_GAP, c1w_state_dict  # noqa: E402


def main():
    torch.manual_seed(0)
    B = 4
    bn = np.load(os.path.join(GOLD, "bn_stats_c1w.npz"))
    ref, _ = build_reference("none")
    sd = c1w_state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, bn)
    full_sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    gold_w = np.load(os.path.join(GOLD, "model_c1w.npz"))
    sha_w = synth.sha256_of([full_sd[k].numpy() for k in sorted(full_sd) if not k.endswith("num_batches_tracked")])
    assert sha_w == str(gold_w["sha256_weights"]), "weights differ from model_c1w.npz's"
    out = {"seeds": np.asarray(SEEDS, dtype=np.int64), "tie_gap": np.float64(TIE_GAP), "sha256_weights": sha_w}
    for s in SEEDS:
        inp = synth.make_inputs(B, seed=s, res=256, num_regions=32, cam="lm")
        gt = synth.make_train_gt(B, inp)
        tin = {k: torch.from_numpy(v) for k, v in inp.items()}
        tgt = {k: torch.from_numpy(v) for k, v in gt.items()}
        p = f"s{s}_"
        out[p + "sha256_inputs"] = synth.sha256_of([inp[k] for k in sorted(inp)])
        out[p + "sha256_gt"] = synth.sha256_of([gt[k] for k in sorted(gt)])
        # train-mode region logits (batch statistics), 8 and 1 threads
        am = {}
        for nt in (8, 1):
            torch.set_num_threads(nt)
            r, _ = build_reference("none")
            r.load_state_dict(full_sd, strict=True)
            r.train()
            with torch.no_grad():
                reg = r.rot_head_net(r.backbone(tin["roi_img"]))[-1]
            am[nt] = torch.softmax(reg[:, 1:], 1).reshape(B, 32, -1).argmax(1).reshape(B, 64, 64)
            if nt == 8:
                top2 = reg[:, 1:].topk(2, dim=1).values
                gap = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
        out[p + "argmax"], out[p + "top2_gap"] = am[8].numpy().astype(np.int8), gap
        f18 = (am[1] != am[8]).numpy()
        out[p + "flip_1v8"] = np.packbits(f18)
        print(f"[seed {s}] train-mode smallest top-2 gap {gap.min():.2e}; pixels with gap < {TIE_GAP:g}: {int((gap < TIE_GAP).sum())}; "
              f"reference flips 1-vs-8 threads {int(f18.sum())}")
        for att in ("none", "mul"):
            L8, g8 = ref_train(att, full_sd, tin, tgt, 8)
            L1, g1 = ref_train(att, full_sd, tin, tgt, 1)
            torch.set_num_threads(8)
            for k, v in L8.items():
                out[p + f"{att}_{k}"] = np.float64(v)
                out[p + f"{att}_noise_{k}"] = np.float64(abs(L1[k] - v))
            noises = []
            for n, g in g8.items():
                nrm = g.double().norm().item()
                out[p + f"{att}_gradnorm/{n}"] = np.float64(nrm)
                nz = (g1[n].double() - g.double()).norm().item() / max(nrm, 1e-30)
                out[p + f"{att}_gradnoise/{n}"] = np.float64(nz)
                if nrm > 1e-4:
                    noises.append(nz)
            print(f"[seed {s} {att}] losses " + " ".join(f"{k.replace('loss_', '')} {v:.6f}" for k, v in L8.items())
                  + f" | 1-vs-8 threads: losses {max(abs(L1[k] - L8[k]) for k in L8):.1e}, gradients median {np.median(noises):.1e} max {max(noises):.1e}")
    path = os.path.join(GOLD, "train_c1w_seeds.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
