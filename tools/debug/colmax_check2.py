import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rdpn6d_amd import ops, _lib
from rdpn6d_amd.ops import split_h2, pack_conv_weight, _pad_vec, _ptr, _stream
from rdpn6d_amd.gdrn import pack_h2_weight
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, H, Cin, N, k = 64, 32, 256, 512, 1
x = torch.randn(B, H, H, Cin, generator=g).to(dev)
w = (torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
sc, sh = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
y, ((h2, shape), flag) = ops.conv2d_nhwc_h2(x, w, sc, sh, pad=0, want_h2=True)
lib = _lib.load()
xh, _ = split_h2(x)
wp32 = pack_conv_weight(w.float(), cin_pad=Cin)
wh, inv = pack_h2_weight(wp32)
scp = _pad_vec(sc.float(), wp32.shape[0], 1.0) * inv
shp = _pad_vec(sh.float(), wp32.shape[0], 0.0)
d = _lib.ConvDesc()
d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(xh), _ptr(wh), _ptr(scp), _ptr(shp), None, None
d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, H, Cin, Cin, 0
d.Ho, d.Wo, d.stride, d.ntaps = H, H, 1, 1
d.N, d.Npad, d.OH, d.OW = N, wp32.shape[0], H, H
d.osy = d.osx = 1
d.out_cs = N
keys = torch.zeros(B, d.Npad, dtype=torch.int64, device=dev)
fl = torch.zeros(1, dtype=torch.int32, device=dev)
_lib.check(lib.rdpn6d_conv2d_h2_colmax(ctypes.byref(d), _ptr(keys), H * H, _ptr(fl), _stream()))
torch.cuda.synchronize()
kk = keys.cpu().numpy().astype(np.uint64)
ordv = (kk >> np.uint64(32)).astype(np.uint32)
u = np.where(ordv & np.uint32(0x80000000), ordv & np.uint32(0x7fffffff), ~ordv)
got = u.view(np.float32)                                # 16 x value
h = h2.view(B, H * H, N // 32, 2, 32).float()
r = (h[:, :, :, 0] + h[:, :, :, 1]).reshape(B, H * H, N).cpu().numpy()
want = r.max(1)
print("match full max:", float((got == want).mean()))
for rows in (256, 128, 64, 32, 16):
    part = r.reshape(B, H * H // rows, rows, N).max(2)   # [B, parts, N]
    m = (got[:, None, :] == part).any(1).mean()
    print(f"got equals the max of SOME aligned block of {rows} rows: {m:.3f}")
print("zero keys:", float((kk == 0).mean()), " sample got/want", got[0, :4], want[0, :4])
out = torch.full((B, N // 32, 2, 32), 7.0, dtype=torch.float16, device=dev)
kcopy = keys.clone()
_lib.check(lib.rdpn6d_h2_colmax_decode(_ptr(keys), B, N, d.Npad, _ptr(out), _stream()))
torch.cuda.synchronize()
pay = (kk & np.uint64(0xffffffff)).astype(np.uint32)
hi = (pay >> np.uint32(16)).astype(np.uint16).view(np.float16).astype(np.float32)
lo = (pay & np.uint32(0xffff)).astype(np.uint16).view(np.float16).astype(np.float32)
print("payload hi+lo == ord value:", float(((hi + lo) == got).mean()))
o = out.cpu().numpy().astype(np.float32)
oh, ol = o[:, :, 0].reshape(B, N), o[:, :, 1].reshape(B, N)
print("decode hi matches:", float((oh == hi).mean()), "lo matches:", float((ol == lo).mean()), "keys zeroed:", int(keys.abs().sum()))
print(oh[0, :4], hi[0, :4], ol[0, :4], lo[0, :4])
hh = h[:, :, :, 0].reshape(B, H * H, N).cpu().numpy()
for n in range(4):
    rows = np.nonzero(hh[0, :, n] == hi[0, n])[0]
    print("channel", n, "payload hi", hi[0, n], "found at rows", rows[:8], "argmax row", int(r[0, :, n].argmax()), "value there", r[0, rows[0], n] if len(rows) else None)
