import os, sys, numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rdpn6d_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, H, Cin, N, k = 64, 32, 256, 512, 1
x = torch.randn(B, H, H, Cin, generator=g).to(dev)
w = (torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
sc, sh = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
y, ((h2, shape), flag) = ops.conv2d_nhwc_h2(x, w, sc, sh, pad=k // 2, want_h2=True)
rec, keys, fl = ops.conv2d_h2_colmax(x, w, sc, sh, H * H)
torch.cuda.synchronize()
h = h2.view(B, H * H, N // 32, 2, 32).float()
r = h[:, :, :, 0] + h[:, :, :, 1]                      # [B, HW, N/32, 32]
idx = r.argmax(dim=1, keepdim=True)                    # [B,1,N/32,32]
want_hi = torch.gather(h[:, :, :, 0], 1, idx)[:, 0]
want_lo = torch.gather(h[:, :, :, 1], 1, idx)[:, 0]
got_hi, got_lo = rec[:, :, 0].float(), rec[:, :, 1].float()
bad = ((got_hi + got_lo) != (want_hi + want_lo))
print("keys zero:", int(keys.abs().sum()), "flag", int(fl), "mismatching maxima:", int(bad.sum()), "of", bad.numel())
if bad.any():
    b, c32, c = [int(t[0]) for t in torch.nonzero(bad, as_tuple=True)]
    print("first bad: crop", b, "channel", c32 * 32 + c, "got", float(got_hi[b, c32, c] + got_lo[b, c32, c]), "want", float(want_hi[b, c32, c] + want_lo[b, c32, c]),
          "argmax row", int(idx[b, 0, c32, c]))
    bb = torch.nonzero(bad, as_tuple=True)
    print("bad channels mod 64 histogram:", torch.bincount((bb[1] * 32 + bb[2]) % 64, minlength=64).tolist())
    print("bad argmax rows mod 256 /32 histogram:", torch.bincount((idx[:, 0][bad] % 256) // 32, minlength=8).tolist())
