"""Training step of the RDPN6D hot path on the HIP kernels: forward with batch statistics, the nine
losses of the shipped configs, and the full backward into ``param.grad``.

Mirrors what ``GDRN.forward(do_loss=True)`` + ``losses.backward()`` do in the reference
(core/gdrn_modeling/models/GDRN.py:107-371 forward, :373-633 losses; engine.py:292-308), but every
tensor op is a C-ABI launch of librdpn6d_hip.so:

  forward   raw conv (rdpn6d_conv2d_f32, no folded BN) -> rdpn6d_bn_train_stats -> rdpn6d_bn_apply
            (+residual, ReLU); GroupNorm train form; same glue / FC kernels as inference
  losses    rdpn6d_dense_losses (coor x/y/z, mask, region CE, region_my + d/dhead in one pass),
            rdpn6d_pose_train (pose decode train variant + PM_R / centroid / z, gradient by dual numbers)
  backward  dgrad = rdpn6d_conv2d_f32 with flipped / transposed / phase-split weights,
            wgrad = rdpn6d_wgrad_f32 (split-K implicit GEMM over pixels), BN/GN backward,
            maxpool / upsample / global-max / glue backward kernels

PyTorch is used for memory, for the (tiny, once per step) weight re-packing and for moving the packed
gradient tiles into ``param.grad`` layout.  Gradients of all 164 parameter tensors are produced.
"""
import ctypes
import os

import torch

from . import _lib
from .gdrn import _pad_to, _ptr

BN_EPS, BN_MOM = 1e-5, 0.1


def _taps(k, pad):
    return [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]


class TrainEngine:
    def __init__(self, model, B, device, amp=False):
        """amp=True (cfg.SOLVER.AMP.ENABLED, the reference's autocast + GradScaler switch: engine.py:279-309): the forward and
        input-gradient convolutions of the trunk, the fusion branch and the dense head run on the bf16 matrix pipe (fp32
        accumulation) and their activations / gradients are STORED in bf16; BatchNorm arithmetic and statistics, losses, the
        head output, ConvPnPNet, parameter gradients, master weights and the optimizer stay fp32 - the split torch.autocast
        makes.  bf16 keeps the fp32 exponent range, so no loss scaling is needed."""
        self.lib = _lib.load()
        self.model, self.B, self.dev = model, B, device
        # amp: False | True / "bf16" | "fp16" (cfg.SOLVER.AMP.DTYPE; the reference's autocast + GradScaler run in fp16: loss
        # scaling then comes from torch's GradScaler through the autograd node of GDRN.forward, or from `loss_scale` below)
        self.amp = bool(amp)
        self.lp = "fp16" if amp == "fp16" else "bf16"   # 16-bit storage format of the mixed-precision step
        self.lp_dtype = torch.float16 if self.lp == "fp16" else torch.bfloat16
        self.loss_scale = 1.0  # forward_backward() multiplies the backward seeds by it and divides the parameter gradients again
        self.accumulate_grad = False  # backward() WRITES param.grad; True = add to what is there (see backward)
        self._seed_factor = 1.0  # factor currently applied to the backward seeds d_head / d_rt (seed_backward)
        self._consumed = True  # backward() has used up the last forward's buffers (nothing to differentiate before the first forward)
        self.adt = self.lp_dtype if self.amp else torch.float32  # storage type of trunk / head activations and gradients
        self.sfx = self.lp if self.amp else "f32"
        self._casts = {}     # (address, stride, offset, channels) of an fp32 activation slice -> its bf16 copy
        self.mirrors = []    # (bf16 tensor, fp32 packed weight) pairs refreshed with the weights
        self.mirrors3 = []   # (three bf16 planes, fp32 packed weight) pairs of the bf16x3 layers, re-split after every re-pack
        self._planes = {}    # fp32 activation -> its three bf16 planes (split once per step, shared by the consumers)
        # fp32 training: forward and input-gradient convolutions of the wide head layers as fp32-accurate bf16x3 convolutions
        # (csrc/conv_igemm_bf16x3.hip) when the batch fills the chip; cfg.SOLVER.BF16X3 = False keeps them on the fp32 MFMA
        self.x3 = (not self.amp) and bool(model.cfg.get("SOLVER", {}).get("BF16X3", True))
        # BatchNorm + ReLU backward without the stored activation (the mask is re-derived from the BN input: csrc/train_norm.hip,
        # rdpn6d_bn_relu_backward_*): one tensor read less in both backward passes of every BN that has no residual before its ReLU
        # conv + BatchNorm pairs of the mixed-precision step: the convolution's epilogue writes the BatchNorm statistics' partial sums
        self.bn_fuse_stats = bool(model.cfg.get("SOLVER", {}).get("BN_FUSE_STATS", os.environ.get("RDPN6D_BN_FUSE_STATS", "1") != "0"))
        # (BatchNorm + ReLU) -> conv pairs of the mixed-precision step: the later layer's input-gradient convolution writes the
        # BatchNorm's backward sums from its epilogue (rdpn6d_conv2d_bf16_bnbwd)
        self.bn_fuse_bwd = bool(model.cfg.get("SOLVER", {}).get("BN_FUSE_BWD", os.environ.get("RDPN6D_BN_FUSE_BWD", "1") != "0"))
        self._bn_by_dy, self._scratch_bnb_need = {}, 0
        # transposing weight packs through LDS tiles (repack_kernel); RDPN6D_REPACK_TILES=0: the pair form for every entry (A/B, tests)
        self.repack_tiles = os.environ.get("RDPN6D_REPACK_TILES", "1") != "0"
        # ... and the LAST BatchNorm of a residual block from the next block's first input-gradient convolution (rdpn6d_conv2d_bf16_bnbwd_y)
        self.bn_fuse_bwd_res = bool(model.cfg.get("SOLVER", {}).get("BN_FUSE_BWD_RES", os.environ.get("RDPN6D_BN_FUSE_BWD_RES", "1") != "0"))
        self.bn_remask = bool(model.cfg.get("SOLVER", {}).get("BN_REMASK", os.environ.get("RDPN6D_BN_REMASK", "1") != "0"))
        self.x3_launches = 0
        cfg = model.cfg
        self.R = int(cfg.MODEL.CDPN.BACKBONE.INPUT_RES)
        self.K = int(cfg.MODEL.CDPN.ROT_HEAD.NUM_REGIONS)
        pc, rc = cfg.MODEL.CDPN.PNP_NET, cfg.MODEL.CDPN.ROT_HEAD
        self.mask_attention = 1 if pc.MASK_ATTENTION == "mul" else 0
        if pc.MASK_ATTENTION not in ("none", "mul"):
            raise ValueError("MASK_ATTENTION must be none | mul")
        if rc.XYZ_LOSS_TYPE != "L1" or rc.REGION_LOSS_TYPE != "CE":
            raise NotImplementedError("only the xyz / region loss types of the shipped RGB-D configs (L1 / CE) are implemented")
        from .gdrn import MASK_TYPES

        if rc.MASK_LOSS_TYPE not in MASK_TYPES:
            raise NotImplementedError(f"unknown mask loss type: {rc.MASK_LOSS_TYPE}")
        # ROT_HEAD.MASK_LOSS_TYPE: L1 (the shipped configs) | BCE (BCEWithLogits, sigmoid attention) | CE (two mask channels): GDRN.py:450-463
        self.mask_type = MASK_TYPES[str(rc.MASK_LOSS_TYPE)]
        if self.mask_type == 2 and self.mask_attention:
            raise NotImplementedError("MASK_ATTENTION with ROT_HEAD.MASK_LOSS_TYPE='CE': get_mask_prob's CE branch raises in the reference "
                                      "itself (torch.softmax has no keepdim argument, models/model_utils.py:39)")
        if rc.XYZ_LOSS_MASK_GT != "visib" or rc.MASK_LOSS_GT != "trunc" or rc.REGION_LOSS_MASK_GT != "visib":
            raise NotImplementedError("loss mask selection other than visib/trunc/visib is not implemented")
        if not (pc.PM_R_ONLY and pc.PM_LOSS_TYPE == "L1" and pc.PM_LW > 0):
            raise NotImplementedError("only the PM_R_ONLY L1 point-matching loss is implemented")
        self.pm_sym = bool(pc.PM_LOSS_SYM)  # closest symmetric target, pm_loss.py:97-99
        if pc.CENTROID_LOSS_TYPE != "L1" or pc.Z_LOSS_TYPE != "L1" or pc.ROT_LW > 0 or pc.TRANS_LW > 0 or pc.get("BIND_LW", 0) > 0:
            raise NotImplementedError("only L1 centroid / z losses (ROT_LW = TRANS_LW = BIND_LW = 0) are implemented")
        self.lw = dict(xyz=float(rc.XYZ_LW), mask=float(rc.MASK_LW), region=float(rc.REGION_LW), pm=float(pc.PM_LW),
                       centroid=float(pc.CENTROID_LW), z=float(pc.Z_LW), pm_norm=1 if pc.PM_NORM_BY_EXTENT else 0)
        self.is_allo = 1 if "allo" in pc.ROT_TYPE else 0
        self.bufs = {}
        self._sym_cache = None
        self._bn_counters = []  # num_batches_tracked buffers, bumped together once per forward
        self.repack = []     # table entries of the one-launch weight re-pack (see _pack_map)
        self.fwd, self.bwd = [], []   # launch closures
        # what each unit reads and writes (tensors by reference, no copies): the local, teacher-forced parity test of the mixed-precision
        # step recomputes every layer's weight / BatchNorm gradients from exactly these stored operands (tests/test_gpu_c1w.py)
        self.records = []
        self._scratch_d = torch.empty(512 * 1024 * 2 + 4096, dtype=torch.float64, device=device)
        self._wg_floats = 0
        self.group_wgrad = bool(model.cfg.get("SOLVER", {}).get("GROUP_WGRAD", True)) and os.environ.get("RDPN6D_GROUP_WGRAD", "1") != "0"  # (env: profiling)
        self._wgrad_groups, self._wgrad_group_list = {}, []
        # cfg.SOLVER.WGRAD_SIDE_STREAM (default OFF): weight gradients whose operands are read in place from per-layer buffers on a second
        # HIP stream (_side_launch) - nothing downstream of them in the backward depends on a weight gradient.  Bit-identical
        # (tests/test_gpu_train.py), and measured NOT faster: bf16 B = 32 10.07 ms against 9.86 on one stream, fp16 10.38 against 10.50
        # (profiles/r5_experiments.md) - the step's kernels already run back to back (sum of kernel durations = step time), two streams
        # only make the big kernels share the chip.  Kept as a switch; RDPN6D_WGRAD_SIDE=1 forces it on for profiling runs.
        self.wgrad_side = (bool(model.cfg.get("SOLVER", {}).get("WGRAD_SIDE_STREAM", False)) or os.environ.get("RDPN6D_WGRAD_SIDE", "0") == "1")
        # cfg.SOLVER.AMP.PNP_NET: ConvPnPNet's three convolutions on the 16-bit pipe too, as torch autocast runs them in the reference's AMP
        # step (engine.py:279-309) - see conv_unit.  "all" | "inner" (the second and third only) | "none"; True / False = all / none.
        # Default: "all" with DTYPE "fp16" (the reference's AMP dtype: the pose branch then sees what it sees there), "inner" with "bf16":
        # the FIRST convolution reads the caller's depth-xyz and 2D coordinates directly, and an 8-bit significand puts metres on a
        # 4 - 8 mm grid (fp16: 1 mm) - it stays fp32; the other two read GroupNorm outputs like any 16-bit layer of the trunk.
        # Measured (B = 32, one MI355X): all three -0.18 ms per step bf16 (9.38 -> 9.20), -0.23 ms fp16.  RDPN6D_PNP_LOWP=0|1|inner: A/B runs.
        pn = model.cfg.get("SOLVER", {}).get("AMP", {}).get("PNP_NET", None)
        pn = ("all" if self.lp == "fp16" else "inner") if pn is None else ({True: "all", False: "none"}.get(pn, str(pn).lower()))
        pn = {"0": "none", "1": "all", "inner": "inner"}.get(os.environ.get("RDPN6D_PNP_LOWP", ""), pn)
        if pn not in ("all", "inner", "none"):
            raise ValueError(f"SOLVER.AMP.PNP_NET={pn!r}: all | inner | none")
        self.pnp_lowp = pn if self.amp else "none"
        self._side, self._side_join, self._side_dirty, self._wg_partial_side = None, None, False, None
        self._bwd_writes, self._side_reads = {}, []  # (build-time bookkeeping of _check_side_operands)
        self._build()
        self._check_side_operands()
        for grp in self._wgrad_group_list:
            _, _, _, ca, _, _, _, cb, _, yhw, _, _, k, _, _ = grp["geom"]
            if len(grp["members"]) > 1:
                self._wg_floats = max(self._wg_floats, int(self.lib.rdpn6d_wgrad_group_scratch_floats(len(grp["members"]), B, yhw[0], yhw[1],
                                                                                                         ca, cb, k * k)))
        if getattr(self, "_scratch_need", 0) > self._scratch_d.numel():  # (every launch reads the pointer when it runs)
            self._scratch_d = torch.empty(self._scratch_need, dtype=torch.float64, device=device)
        self._scratch_bnb = torch.empty(max(self._scratch_bnb_need, 1), dtype=torch.float64, device=device)
        self._wg_partial = torch.empty(max(self._wg_floats, 1), dtype=torch.float32, device=device)
        self.refresh_weights()

    # ------------------------------------------------------------------ helpers
    _FP32_TAGS = ("pnp", "fc1", "fc2", ":rt", "head_out")  # the pose branch and the head output stay fp32 under AMP

    def buf(self, name, *shape, dtype=None, zero=False):
        """named persistent buffer.  Under AMP the stored activations / gradients of trunk, fusion branch and dense head
        (names raw:* act:* d:* dres:*) are bf16, everything else fp32."""
        if dtype is None:
            act = name.startswith(("raw:", "act:", "d:", "dres:")) and not any(t in name for t in self._FP32_TAGS)
            dtype = self.adt if act else torch.float32
        if name not in self.bufs:
            self.bufs[name] = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
        return self.bufs[name]

    def st(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def lpf(self, bf16_name):
        """the entry point of the step's 16-bit format: rdpn6d_*_bf16 -> rdpn6d_*_fp16 under cfg.SOLVER.AMP.DTYPE = "fp16"""
        return getattr(self.lib, bf16_name.replace("bf16", self.lp))

    def _grad(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def _conv_desc(self, x, xhw, in_cs, in_co, cin, w, y, yhw, out_cs, out_co, N, taps, stride=1, phase=None, shift=None,
                   res=None, res_cs=0, act=0, slope=0.0):
        d = _lib.ConvDesc()
        d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(x), _ptr(w), None, _ptr(shift), _ptr(res), _ptr(y)
        d.B, d.H, d.W = self.B, xhw[0], xhw[1]
        d.Cin, d.in_cs, d.in_co = cin, in_cs, in_co
        d.ntaps = len(taps)
        for t, (dy, dx) in enumerate(taps):
            d.dy[t], d.dx[t] = dy, dx
        d.stride = stride
        d.N, d.Npad = N, w.shape[0]
        d.OH, d.OW = yhw
        if phase is None:
            d.Ho, d.Wo, d.osy, d.osx, d.ooy, d.oox = yhw[0], yhw[1], 1, 1, 0, 0
        else:
            d.Ho, d.Wo, d.osy, d.osx, d.ooy, d.oox = phase
        d.out_cs, d.out_co, d.res_cs, d.res_co = out_cs, out_co, res_cs, 0
        d.act, d.slope = act, slope
        assert w.shape[1] == d.ntaps and w.shape[2] == cin, (tuple(w.shape), d.ntaps, cin)
        return d

    def _bf16_of(self, launches, src, src_cs, src_co, C, npix, cache):
        """compact bf16 copy [npix, pad32(C)] of an fp32 channel slice; the cast launch is appended to `launches`
        (forward activations are cast once and shared by their consumers)"""
        Cp = _pad_to(C, 32)
        key = (src.data_ptr(), src_cs, src_co, C)
        if cache and key in self._casts:
            return self._casts[key], Cp
        t = torch.zeros(npix, Cp, dtype=self.lp_dtype, device=self.dev)
        lib = self.lib

        def run():
            _lib.check(self.lpf("rdpn6d_cast_f32_bf16")(_ptr(src), src_cs, src_co, C, _ptr(t), Cp, npix, self.st()), "cast bf16")

        run.keep = (src, t)
        launches.append(run)
        if cache:
            self._casts[key] = t
        return t, Cp

    def _mirror(self, t):
        tb = torch.zeros(*t.shape, dtype=self.lp_dtype, device=self.dev)
        self.mirrors.append((tb, t))
        return tb

    def _mirror3(self, t):
        n = t.numel()
        p3 = torch.zeros(3, _pad_to(n, 8), dtype=torch.bfloat16, device=self.dev)
        self.mirrors3.append((p3, t))
        return p3

    def _planes_of(self, launches, src, cache):
        """three bf16 planes of a contiguous fp32 tensor (input format of the bf16x3 convolution); the split launch is
        appended to `launches`"""
        key = src.data_ptr()
        if cache and key in self._planes:
            return self._planes[key]
        n = src.numel()
        p3 = torch.empty(3, _pad_to(n, 8), dtype=torch.bfloat16, device=self.dev)
        lib = self.lib

        def run():
            _lib.check(lib.rdpn6d_split_bf16x3(_ptr(src), n, _ptr(p3), p3.shape[1], self.st()), "split bf16x3")

        run.keep = (src, p3)
        launches.append(run)
        if cache:
            self._planes[key] = p3
        return p3

    def _x3_wanted(self, d, allow_tile=False):
        """the 256x256 bf16x3 kernel would take this convolution and fills the chip with it; allow_tile: also the 128x128..64x64
        tile kernel (trunk layers with >= 128 channels at batches >= 16: forward, input gradient AND the bf16x3 weight gradient
        share the two split passes - 1.3-1.6x per GEMM against ~12 us per split)"""
        if 6 * d.B * max(d.H * d.W * d.in_cs, d.OH * d.OW * d.out_cs) >= (1 << 32) - 64:  # three planes behind one descriptor
            return False
        if not (self.x3 and d.Cin == d.in_cs and d.in_co == 0):
            return False
        which = self.lib.rdpn6d_conv_bf16x3_kernel_for(ctypes.byref(d))
        return which == 2 or (allow_tile and which == 1 and d.Cin >= 128 and d.N >= 128 and self.B * self.R * self.R >= 16 * 65536)

    def _launch_conv_x3(self, name, d, xp3, wp3, keep):
        lib = self.lib
        self.x3_launches += 1

        def run():
            _lib.check(lib.rdpn6d_conv2d_bf16x3(ctypes.byref(d), xp3.shape[1], wp3.shape[1], None, 0, self.st()), name)

        run.keep = (d, keep, xp3, wp3)
        return run

    def _launch_conv(self, name, d, keep, ksplit=False, lowp=False, out_f32=True):
        lib = self.lib
        from .gdrn import pick_ksplit

        if lowp:
            of = 1 if out_f32 else 0
            nkb = d.ntaps * d.Cin // (64 if d.Cin % 64 == 0 else 32)
            ksb = pick_ksplit(d.B * d.Ho * d.Wo, d.Npad, nkb) if (ksplit and d.osy == 1 and d.ooy == 0 and d.OH == d.Ho) else 1
            if ksb > 1:
                wsb = self.buf("splitk_ws:" + name, int(lib.rdpn6d_conv_splitk_ws_floats(ctypes.byref(d), ksb)))

                def run():
                    _lib.check(self.lpf("rdpn6d_conv2d_splitk_bf16")(ctypes.byref(d), of, ksb, _ptr(wsb), self.st()), name)
            else:
                rows = ctypes.c_int(0)

                def run():
                    if run.bn_stats:  # a BatchNorm follows: its statistics' partial sums come out of this launch's epilogue (bn_unit)
                        _lib.check(self.lpf("rdpn6d_conv2d_bf16_bnstats")(ctypes.byref(d), _ptr(self._scratch_d), 0, ctypes.byref(rows),
                                                                          self.st()), name)
                        run.stats_rows = rows.value
                    elif run.bn_bwd is not None:  # an input-gradient convolution whose output is the gradient of a BatchNorm + ReLU:
                        r = run.bn_bwd            # that BatchNorm's backward sums come out of this launch's epilogue (conv_unit)
                        if r["y"] is not None:    # (the last BatchNorm of a residual block: mask from the stored block output)
                            _lib.check(self.lpf("rdpn6d_conv2d_bf16_bnbwd_y")(ctypes.byref(d), _ptr(r["x_raw"]), r["cs"], r["co"], _ptr(r["y"]),
                                                                              r["ycs"], r["yco"], _ptr(r["mean"]), _ptr(r["invstd"]),
                                                                              _ptr(self._scratch_bnb), ctypes.byref(rows), self.st()), name)
                        else:
                            _lib.check(self.lpf("rdpn6d_conv2d_bf16_bnbwd")(ctypes.byref(d), _ptr(r["x_raw"]), r["cs"], r["co"], _ptr(r["mean"]),
                                                                            _ptr(r["invstd"]), _ptr(r["ga"]), _ptr(r["be"]),
                                                                            _ptr(self._scratch_bnb), ctypes.byref(rows), self.st()), name)
                        run.stats_rows = rows.value
                        self._bnb_owner = r  # the ONE shared sums buffer now holds this BatchNorm's partial sums (checked by its bwd)
                    else:
                        _lib.check(self.lpf("rdpn6d_conv2d_bf16")(ctypes.byref(d), of, self.st()), name)

                run.bn_capable, run.bn_stats, run.bn_bwd, run.stats_rows, run.desc = not of, False, None, 0, d

            run.keep = (d, keep)
            return run

        ks = pick_ksplit(d.B * d.Ho * d.Wo, d.Npad, d.ntaps * d.Cin // 16) if ksplit else 1
        if ks > 1:
            ws = self.buf("splitk_ws:" + name, int(lib.rdpn6d_conv_splitk_ws_floats(ctypes.byref(d), ks)))

            def run():
                _lib.check(lib.rdpn6d_conv2d_splitk_f32(ctypes.byref(d), ks, _ptr(ws), self.st()), name)
        else:
            def run():
                _lib.check(lib.rdpn6d_conv2d_f32(ctypes.byref(d), self.st()), name)

        run.keep = (d, keep)
        return run

    # persistent packed-weight buffers, all refreshed from the live parameters by ONE rdpn6d_repack_f32 launch:
    #   dst[(o*dT + t)*dIpad + i] = src[operm(o)*so + iperm(i)*si + toff[t]]   (see include/rdpn6d.h)
    def _pack_map(self, shape, src, O, T, I, so, si, toff, operm=None, iperm=None, dst=None, dst_off=0, dT=None, dIpad=None):
        if dst is None:
            dst = torch.zeros(*shape, dtype=torch.float32, device=self.dev)
        assert src.dtype == torch.float32 and T <= 9 and len(toff) == T
        dev_i32 = lambda v: None if v is None else torch.as_tensor(list(v), dtype=torch.int32).to(self.dev)  # noqa: E731
        self.repack.append(dict(src=src, dst=dst, off=int(dst_off), O=int(O), T=int(T), I=int(I), so=int(so), si=int(si),
                                toff=[int(v) for v in toff], operm=dev_i32(operm), iperm=dev_i32(iperm),
                                dT=int(dT if dT is not None else (shape[1] if len(shape) == 3 else 1)),
                                dIpad=int(dIpad if dIpad is not None else shape[-1])))
        return dst

    def _repack_table(self):
        import numpy as np

        ptrs = tuple(e["src"].data_ptr() for e in self.repack)
        if getattr(self, "_repack_ptrs", None) == ptrs:
            return
        dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("dst_bf16", "<u8"), ("operm", "<u8"), ("iperm", "<u8"), ("so", "<i8"),
                       ("si", "<i8"), ("start", "<i8"), ("O", "<i4"), ("T", "<i4"), ("I", "<i4"), ("dT", "<i4"), ("dIpad", "<i4"),
                       ("toff", "<i4", (9,))])
        assert dt.itemsize == 120
        tab = np.zeros(len(self.repack), dt)
        start = 0
        mirror_of = {id(t): tb for tb, t in self.mirrors}
        for r, e in zip(tab, self.repack):
            tb = mirror_of.get(id(e["dst"]))
            # a packed tensor with a 16-bit mirror is only ever read through the mirror (mixed precision: every convolution that has
            # one takes it): its fp32 form is not written - half of the re-pack's 580 MB per step
            r["src"], r["dst"] = e["src"].data_ptr(), (e["dst"].data_ptr() + 4 * e["off"] if tb is None else 0)
            r["dst_bf16"] = tb.data_ptr() + 2 * e["off"] if tb is not None else 0
            r["operm"] = e["operm"].data_ptr() if e["operm"] is not None else 0
            r["iperm"] = e["iperm"].data_ptr() if e["iperm"] is not None else 0
            r["so"], r["si"], r["start"] = e["so"], e["si"], start
            r["O"], r["T"], r["I"], r["dT"], r["dIpad"] = e["O"], e["T"], e["I"], e["dT"], e["dIpad"]
            r["toff"][: e["T"]] = e["toff"]
            start += e["O"] * e["T"] * e["I"]
        # workgroup map: 1024 (o, i) pairs (x T taps; 2048 when T = 1) per workgroup, never straddling two entries
        # (transposing entries - consecutive i far apart in the source - go through LDS in 64-i x 16|64-o tiles: bit 30 of the map entry,
        #  offset = the tile's first pair; see repack_kernel)
        bd, bo = [], []
        for di, e in enumerate(self.repack):
            n = e["O"] * e["I"]
            if self.repack_tiles and e["operm"] is None and e["iperm"] is None and e["si"] > e["so"] and e["I"] >= 32:
                ot = 64 if e["T"] <= 2 else 16
                o0, i0 = np.meshgrid(np.arange(0, e["O"], ot, dtype=np.int64), np.arange(0, e["I"], 64, dtype=np.int64), indexing="ij")
                offs = (o0 * e["I"] + i0).reshape(-1)
                bo.append(offs)
                bd.append(np.full(len(offs), di | 0x40000000, dtype=np.int32))
                continue
            offs = np.arange(0, n, 2048 if e["T"] == 1 else 1024, dtype=np.int64)
            bo.append(offs)
            bd.append(np.full(len(offs), di, dtype=np.int32))
        self._repack_dev = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(self.dev)
        self._repack_bd = torch.from_numpy(np.concatenate(bd)).to(self.dev)
        self._repack_bo = torch.from_numpy(np.concatenate(bo)).to(self.dev)
        self._repack_ptrs = ptrs

    def refresh_weights(self):
        """re-pack forward / dgrad weights (and their bf16 mirrors) from the current parameters: one launch.  Call after
        every optimizer step.  The table is rebuilt only when a parameter's storage moved (e.g. an optimizer that re-homes
        the parameters into a flat buffer)."""
        self._repack_table()
        repack = self.lib.rdpn6d_repack_fp16 if (self.amp and self.lp == "fp16") else self.lib.rdpn6d_repack_f32  # 16-bit mirrors
        _lib.check(repack(_ptr(self._repack_dev), _ptr(self._repack_bd), _ptr(self._repack_bo),
                                              int(self._repack_bd.numel()), self.st()), "repack")
        for p3, t in self.mirrors3:
            _lib.check(self.lib.rdpn6d_split_bf16x3(_ptr(t), t.numel(), _ptr(p3), p3.shape[1], self.st()), "split weights")
        if getattr(self, "_stem_h2", None) is not None:
            w, wh, inv = self._stem_h2
            _lib.check(self.lib.rdpn6d_stem_pack_h2(_ptr(w), _ptr(wh), _ptr(inv), self.st()), "stem weights")

    # ------------------------------------------------------------------ layer builders
    def conv_unit(self, name, P, x, xhw, in_cs, in_co, cin_real, y, yhw, out_cs, out_co, *, stride=1, perm=None, bias=None,
                  dx=None, dx_cs=None, dx_res=None, act_out=None, slope=0.0, first=False, dx_final=False):
        """One convolution: forward launch now, backward launches pushed on self.bwd (executed in reverse).
        x: input activation tensor; y: raw output tensor; dx: gradient buffer for the input (None = not needed).
        The gradient w.r.t. y is expected in self.bufs['d:'+name] when the backward runs."""
        w = P.weight
        cout, _, k, _ = w.shape
        pad = k // 2
        self._note_bwd_write(dx)
        # ConvPnPNet stays fp32 under AMP (pose regression) unless cfg.SOLVER.AMP.PNP_NET says otherwise (round 6): its three stride-2
        # convolutions then run like the reference's autocast runs them - 16-bit operands (one compact copy of the fp32 input / output
        # gradient each), fp32 accumulation and fp32 outputs; GroupNorm, the FC stack and the pose decode stay fp32
        lowp = self.amp and (not name.startswith("pnp_net") or self.pnp_lowp == "all" or (self.pnp_lowp == "inner" and not name.endswith(".0")))
        x3_fwd, xp3, g3 = False, None, None  # bf16x3 forward taken; planes of x / of the output gradient
        cin_pad = _pad_to(cin_real, 32 if lowp else 16)
        npad = _pad_to(cout, 64)
        lib, B = self.lib, self.B
        # ---- forward weights
        kk2, cin_w = k * k, w.shape[1]
        wf = self._pack_map((npad, kk2, cin_pad), w, cout, kk2, cin_real, cin_w * kk2, kk2, range(kk2), iperm=perm)
        bvec = None
        if bias is not None:
            bvec = self._pack_map((npad,), bias, 1, 1, cout, 0, 1, [0])
        taps = _taps(k, pad)
        if lowp:
            if x.dtype == self.lp_dtype:  # stored in 16 bits: read in place (channels beyond cin_real inside the slice are zero)
                assert in_cs % 8 == 0 and in_co % 8 == 0 and in_co + cin_pad <= in_cs, (name, in_cs, in_co, cin_pad)
                xb, xb_cs, xb_co = x, in_cs, in_co
            else:
                xb, xb_cs, xb_co = self._bf16_of(self.fwd, x, in_cs, in_co, cin_real, B * xhw[0] * xhw[1], cache=True)[0], cin_pad, 0
            wfb = self._mirror(wf)
            d = self._conv_desc(xb, xhw, xb_cs, xb_co, cin_pad, wfb, y, yhw, out_cs, out_co, cout, taps, stride=stride, shift=bvec,
                                act=act_out or 0, slope=slope)
            self.fwd.append(self._launch_conv(name, d, (wf, wfb, bvec, xb), lowp=True, out_f32=y.dtype == torch.float32, ksplit=True))
        else:
            d = self._conv_desc(x, xhw, in_cs, in_co, cin_pad, wf, y, yhw, out_cs, out_co, cout, taps, stride=stride, shift=bvec,
                                act=act_out or 0, slope=slope)
            tile_ok = name.startswith("layer")  # residual trunk
            if stride == 1 and perm is None and self._x3_wanted(d, tile_ok):
                x3_fwd = True
                xp3 = self._planes_of(self.fwd, x, cache=True)
                wf3 = self._mirror3(wf)
                d.x, d.w = _ptr(xp3), _ptr(wf3)
                self.fwd.append(self._launch_conv_x3(name, d, xp3, wf3, (wf, bvec, x)))
            else:
                self.fwd.append(self._launch_conv(name, d, (wf, bvec), ksplit=True))
        # ---- backward
        dy = self.buf("d:" + name, *y.shape, zero=True, dtype=y.dtype)  # gradient w.r.t. the raw conv output (same layout as y)
        self.records.append(dict(kind="conv", name=name, P=P, bias=bias, x=x, xhw=xhw, in_cs=in_cs, in_co=in_co, cin=cin_real, y=y, yhw=yhw,
                                 out_cs=out_cs, out_co=out_co, cout=cout, k=k, stride=stride, perm=perm, dy=dy, dx=dx, dx_cs=dx_cs or in_cs,
                                 dx_res=dx_res, lowp=lowp))
        M = B * yhw[0] * yhw[1]
        ca = _pad_to(cout, 4)
        cb = _pad_to(cin_real, 4)
        tdy = (ctypes.c_int * 9)(*[t[0] for t in taps] + [0] * (9 - len(taps)))
        tdx = (ctypes.c_int * 9)(*[t[1] for t in taps] + [0] * (9 - len(taps)))
        wg_out = self.buf("wg:" + name, ca, k * k, cb) if perm is not None else None
        self._wg_floats = max(self._wg_floats, int(lib.rdpn6d_wgrad_scratch_floats(B, yhw[0], yhw[1], ca, cb, k * k)))
        inv_perm = None
        if perm is not None:
            inv_perm = torch.empty(len(perm), dtype=torch.long)
            inv_perm[torch.tensor(perm)] = torch.arange(len(perm))
            inv_perm = inv_perm.to(self.dev)

        launches = []
        if lowp:
            n_red_b = _pad_to(cout, 32)
            if dy.dtype == self.lp_dtype and out_cs % 8 == 0 and out_co % 8 == 0 and out_co + n_red_b <= out_cs:
                dyb, dyb_cs, dyb_co = dy, out_cs, out_co      # stored in bf16: read in place
            else:  # fp32 gradient (the head output's): one compact bf16 copy for the wgrad and the dgrad
                dyb, dyb_cs, dyb_co = self._bf16_of(launches, dy, out_cs, out_co, cout, M, cache=False)[0], n_red_b, 0

        # bf16x3 weight gradient (fp32-accurate, 1.6x the fp32-MFMA kernel on the head layers): needs the planes of x (from the
        # forward) and of dy (split here, shared with the input-gradient convolution below)
        x3_w = x3_fwd and inv_perm is None and cout > 64 and cin_real > 64 and cout % 8 == 0 and cin_real % 8 == 0 and out_co == 0
        if x3_w:
            g3 = self._planes_of(launches, dy, cache=False)

        def wgrad():
            # the split-K reduce scatters straight into the parameter's own OIHW gradient (element (n, tap, c) at
            # n*Cin*k*k + c*k*k + tap) unless the input channels are permuted in the activation buffer
            direct = inv_perm is None
            gw = self._grad(w)
            tgt = (_ptr(gw), cin_real * k * k, 1, k * k, cout, cin_real) if direct else None
            if lowp:
                args = (_ptr(dyb), dyb_cs, dyb_co, ca, min(_pad_to(cout, 8), dyb_cs - dyb_co), _ptr(xb), xb_cs, xb_co, cb,
                        min(_pad_to(cin_real, 8), xb_cs - xb_co), B, yhw[0], yhw[1], xhw[0], xhw[1], stride, k * k, tdy, tdx)
                if direct:
                    _lib.check(self.lpf("rdpn6d_wgrad_bf16_strided")(*args, *tgt, _ptr(self._wg_partial), self.st()), "wgrad " + name)
                else:
                    _lib.check(self.lpf("rdpn6d_wgrad_bf16")(*args, _ptr(wg_out), _ptr(self._wg_partial), self.st()), "wgrad " + name)
            elif x3_w:
                _lib.check(lib.rdpn6d_wgrad_bf16x3_strided(_ptr(g3), g3.shape[1], out_cs, out_co, ca, cout, _ptr(xp3), xp3.shape[1], in_cs,
                                                           in_co, cb, cin_real, B, yhw[0], yhw[1], xhw[0], xhw[1], stride, k * k, tdy,
                                                           tdx, *tgt, _ptr(self._wg_partial), self.st()), "wgrad " + name)
            else:
                args = (_ptr(dy), out_cs, out_co, ca, _ptr(x), in_cs, in_co, cb, B, yhw[0], yhw[1], xhw[0], xhw[1], stride, k * k,
                        tdy, tdx)
                if direct:
                    _lib.check(lib.rdpn6d_wgrad_f32_strided(*args, *tgt, _ptr(self._wg_partial), self.st()), "wgrad " + name)
                else:
                    _lib.check(lib.rdpn6d_wgrad_f32(*args, _ptr(wg_out), _ptr(self._wg_partial), self.st()), "wgrad " + name)
            if not direct:
                g = wg_out[:cout, :, :cin_real].view(cout, k, k, cin_real).permute(0, 3, 1, 2)
                gw.copy_(g[:, inv_perm])
            if bias is not None:
                csum = self.lpf("rdpn6d_channel_sum_bf16") if dy.dtype == self.lp_dtype else lib.rdpn6d_channel_sum_f32
                if cout % 4 == 0:
                    _lib.check(csum(_ptr(dy), M, ca, out_cs, out_co, _ptr(self._grad(bias)), 0, _ptr(self._scratch_d), self.st()),
                               "bias grad " + name)
                else:
                    bg = self.buf("bg:" + name, npad)
                    _lib.check(csum(_ptr(dy), M, ca, out_cs, out_co, _ptr(bg), 0, _ptr(self._scratch_d), self.st()), "bias grad " + name)
                    self._grad(bias).copy_(bg[:cout])

        # The same-shaped k x k convolutions of a ResNet stage (5 .. 11 of them) take their weight gradients in ONE grouped launch
        # (rdpn6d_wgrad_bf16_group) issued where the stage's first such convolution would have taken its own - the last of them in
        # the backward order: every member's gradient and input activation live in their own buffers until the step ends.  One
        # launch + one reduce instead of two per convolution, and the chip is filled by the members' tiles instead of 14 .. 23
        # K-splits of each (cfg.SOLVER.GROUP_WGRAD, default on).
        grouped = False
        if (lowp and self.group_wgrad and name.startswith("layer") and k > 1 and bias is None and inv_perm is None and dyb is dy
                and xb is x and cin_real % 4 == 0 and cout % 4 == 0):
            geom = (name.split(".")[0], dyb_cs, dyb_co, ca, min(_pad_to(cout, 8), dyb_cs - dyb_co), xb_cs, xb_co, cb,
                    min(_pad_to(cin_real, 8), xb_cs - xb_co), yhw, xhw, stride, k, cout, cin_real)
            grp = self._wgrad_groups.get(geom)
            if grp is None or len(grp["members"]) >= 16:
                grp = self._wgrad_groups[geom] = dict(members=[], geom=geom, tdy=tdy, tdx=tdx)
                self._wgrad_group_list.append(grp)  # (a stage with more than 16 same-shaped convolutions - ResNet-101 / 152 - has several)
                launches.append(self._side_launch(lambda grp=grp: self._run_wgrad_group(grp), lambda grp=grp: [t[3] for t in grp["members"]],
                                                  reads=lambda grp=grp: [t for m_ in grp["members"] for t in m_[1:3]]))
            grp["members"].append((name, dyb, xb, w))
            grouped = True
        if not grouped:
            # (operands read in place from this layer's own buffers, gradient scattered straight into the parameter's: safe on the side stream)
            side_ok = lowp and inv_perm is None and bias is None and dyb is dy and xb is x
            launches.append(self._side_launch(wgrad, [w], reads=[dy, x]) if side_ok else wgrad)
        if dx is not None:
            n_red = _pad_to(cout, 32 if lowp else 16)  # reduction channels of the dgrad = output channels of the forward
            cdx = _pad_to(cin_real, 64)
            dx_cs = dx_cs or in_cs
            if lowp:
                g_src, g_cs, g_co = dyb, dyb_cs, dyb_co   # bf16 gradient (in place or the compact copy); weights from the bf16 mirror
            else:
                assert out_cs - out_co >= n_red, (name, out_cs, n_red)
                g_src, g_cs, g_co = dy, out_cs, out_co

            def dlaunch(nm, dd, wd):
                if lowp:
                    wdb = self._mirror(wd)
                    dd.w = _ptr(wdb)
                    return self._launch_conv(nm, dd, (wd, wdb, g_src), lowp=True, out_f32=dx.dtype == torch.float32, ksplit=True)
                return self._launch_conv(nm, dd, wd, ksplit=dd.osy == 1 and dd.ooy == 0 and dd.OH == dd.Ho)  # split-K needs a linear output

            if stride == 1:
                # dgrad weights: dst[c][t][n] = w[n][perm(c)][flipped t]
                wd = self._pack_map((cdx, kk2, n_red), w, cin_real, kk2, cout, kk2, cin_w * kk2, [kk2 - 1 - t for t in range(kk2)],
                                    operm=perm)
                dd = self._conv_desc(g_src, yhw, g_cs, g_co, n_red, wd, dx, xhw, dx_cs, in_co, cin_real, taps, stride=1,
                                     res=dx_res, res_cs=dx_cs)
                if not lowp and perm is None and self._x3_wanted(dd, name.startswith("layer")):
                    if g3 is None:
                        g3 = self._planes_of(launches, dy, cache=False)  # split launch first, then the convolution
                    wd3 = self._mirror3(wd)
                    dd.x, dd.w = _ptr(g3), _ptr(wd3)
                    launches.append(self._launch_conv_x3("dgrad " + name, dd, g3, wd3, (wd, dy)))
                else:
                    drun = dlaunch("dgrad " + name, dd, wd)
                    rec = self._bn_by_dy.get(dx.data_ptr())
                    # (a residual block's last BatchNorm - rec["y"] - takes the gradient w.r.t. the block output: only the launch that
                    #  completes it, residual added, may carry its sums - and only a launch with work to hide the two extra tensor reads
                    #  behind: a Bottleneck's 1x1 conv1 is HBM-bound, its fused epilogue cost 117 us against 57 + the separate pass's 47,
                    #  C5 shape, round 5)
                    if (rec is not None and rec["producer"] is None and getattr(drun, "bn_capable", False)
                            and (dx_res is None if rec["y"] is None else (dx_final and n_red * kk2 >= 576))
                            and rec["dy_cs"] == dx_cs and rec["dy_co"] == in_co and rec["C"] == cin_real
                            and rec["M"] == B * xhw[0] * xhw[1]):
                        drun.bn_bwd, rec["producer"] = rec, drun
                        self._scratch_bnb_need = max(self._scratch_bnb_need, ((rec["M"] + 63) // 64) * 2 * rec["C"] * 2)
                    launches.append(drun)
            else:
                assert stride == 2 and perm is None
                if k == 3:
                    for py in (0, 1):
                        for px in (0, 1):
                            ys = [(1, 0)] if py == 0 else [(0, 1), (2, 0)]
                            xs = [(1, 0)] if px == 0 else [(0, 1), (2, 0)]
                            ptaps = [(dyo, dxo) for _, dyo in ys for _, dxo in xs]
                            kk = [(ky, kx) for ky, _ in ys for kx, _ in xs]

                            wd = self._pack_map((cdx, len(ptaps), n_red), w, cin_real, len(ptaps), cout, kk2, cin_w * kk2,
                                                [ky * k + kx for ky, kx in kk])
                            dd = self._conv_desc(g_src, yhw, g_cs, g_co, n_red, wd, dx, xhw, dx_cs, in_co, cin_real, ptaps,
                                                 phase=(yhw[0], yhw[1], 2, 2, py, px))
                            launches.append(dlaunch(f"dgrad {name} phase{py}{px}", dd, wd))
                else:  # 1x1 stride 2: only the even pixels receive a gradient; accumulate onto what is there
                    wd = self._pack_map((cdx, 1, n_red), w, cin_real, 1, cout, kk2, cin_w * kk2, [0])
                    dd = self._conv_desc(g_src, yhw, g_cs, g_co, n_red, wd, dx, xhw, dx_cs, in_co, cin_real, [(0, 0)],
                                         phase=(yhw[0], yhw[1], 2, 2, 0, 0), res=dx, res_cs=dx_cs)
                    launches.append(dlaunch("dgrad " + name, dd, wd))
        self.bwd.append(launches)
        return dy

    def _note_bwd_write(self, *tensors):
        """(build time) the backward launches of the unit being registered - list index len(self.bwd) - write these buffers"""
        for t in tensors:
            if t is not None:
                self._bwd_writes.setdefault(t.data_ptr(), []).append(len(self.bwd))

    def _check_side_operands(self):
        """cfg.SOLVER.WGRAD_SIDE_STREAM: a side-stream weight gradient reads its layer's output gradient and input activation while the
        main stream goes on with the backward - safe only if no LATER backward launch (= a unit registered EARLIER: lower list index)
        writes those buffers.  Checked here for every side launch against every buffer a unit declared as its dx / dres, instead of
        trusting the buffer naming; a violation is a build error, not a timing-dependent wrong gradient."""
        for idx, reads in self._side_reads:
            for t in (reads() if callable(reads) else reads):
                later = [j for j in self._bwd_writes.get(t.data_ptr(), []) if j < idx]
                if later:
                    raise RuntimeError(f"WGRAD_SIDE_STREAM: an operand of the side-stream weight gradient at backward list {idx} is written "
                                       f"again by list(s) {later}, which run after it on the main stream")

    def _side_launch(self, fn, params, reads=()):
        """`fn` (a weight-gradient launch closure writing the gradients of `params`) on the engine's side stream: the side stream waits for an event recorded on the
        current stream at the closure's place in the launch list (its operands - this layer's output gradient and input activation -
        are complete there and are not written again before the step ends), uses its own split-K scratch, and is joined by
        _join_side() before a parameter group's gradients are handed on."""
        if not self.wgrad_side:
            return fn
        self._side_reads.append((len(self.bwd), reads))
        ev = [None]

        def run():
            main = torch.cuda.current_stream(self.dev)
            for q in params() if callable(params) else params:
                self._grad(q)  # (a gradient tensor that has to be created is created - and zero-filled - on the main stream, in front of the event)
            if self._side is None:
                self._side, self._side_join = torch.cuda.Stream(self.dev), torch.cuda.Event()
                self._wg_partial_side = torch.empty_like(self._wg_partial)
            if ev[0] is None:
                ev[0] = torch.cuda.Event()
            ev[0].record(main)
            self._side.wait_event(ev[0])
            keep, self._wg_partial = self._wg_partial, self._wg_partial_side
            try:
                with torch.cuda.stream(self._side):
                    fn()
            finally:
                self._wg_partial = keep
            self._side_dirty = True

        return run

    def _join_side(self):
        if self._side_dirty:
            self._side_join.record(self._side)
            torch.cuda.current_stream(self.dev).wait_event(self._side_join)
            self._side_dirty = False

    def _run_wgrad_group(self, grp):
        m = grp["members"]
        _, dcs, dco, ca, ca_ld, xcs, xco, cb, cb_ld, yhw, xhw, stride, k, cout, cin = grp["geom"]
        if len(m) == 1:
            name, dyb, xb, w = m[0]
            _lib.check(self.lpf("rdpn6d_wgrad_bf16_strided")(
                _ptr(dyb), dcs, dco, ca, ca_ld, _ptr(xb), xcs, xco, cb, cb_ld, self.B, yhw[0], yhw[1], xhw[0], xhw[1], stride, k * k,
                grp["tdy"], grp["tdx"], _ptr(self._grad(w)), cin * k * k, 1, k * k, cout, cin, _ptr(self._wg_partial), self.st()),
                "wgrad " + name)
            return
        G = len(m)
        P = ctypes.c_void_p * G
        _lib.check(self.lpf("rdpn6d_wgrad_bf16_group")(
            G, P(*[t[1].data_ptr() for t in m]), dcs, dco, ca, ca_ld, P(*[t[2].data_ptr() for t in m]), xcs, xco, cb, cb_ld, self.B,
            yhw[0], yhw[1], xhw[0], xhw[1], stride, k * k, grp["tdy"], grp["tdx"], P(*[self._grad(t[3]).data_ptr() for t in m]),
            cin * k * k, 1, k * k, cout, cin, _ptr(self._wg_partial), self._wg_partial.numel(), self.st()),
            f"wgrad group {m[0][0]} .. {m[-1][0]}")

    def bn_unit(self, name, bn, x_raw, cs, co, C, M, y, ycs, yco, relu, res=None, res_cs=0, dx=None, dres=None, dy=None,
                dy_cs=None):
        """BatchNorm (train) forward now; the backward consumes dy (gradient w.r.t. y; allocated here with y's layout
        unless given) and writes dx (same layout as x_raw) and optionally dres (the ReLU-masked gradient, which is
        the gradient of the residual branch)."""
        lib = self.lib
        self._note_bwd_write(dx, dres)
        mean, invstd = self.buf("mean:" + name, _pad_to(C, 4)), self.buf("istd:" + name, _pad_to(C, 4))
        ga, be = bn.weight, bn.bias  # read in place (C % 4 == 0); pointers are taken at launch time

        t = self.lp if x_raw.dtype == self.lp_dtype else "f32"   # storage type of x_raw / y / res / dy / dx / dres alike
        assert y.dtype == x_raw.dtype and (res is None or res.dtype == x_raw.dtype), name
        f_stats, f_apply, f_bwd = (getattr(lib, f"rdpn6d_bn_{n}_{t}") for n in ("train_stats", "apply", "backward"))

        # The convolution that produced x_raw is the previous launch and wrote all of it in 16 bits: its epilogue also writes the
        # statistics' partial sums (rdpn6d_conv2d_bf16_bnstats) and only the finalize is left here - no second pass over x_raw
        prev = self.fwd[-1] if self.fwd else None
        pd = getattr(prev, "desc", None)
        fused = (self.bn_fuse_stats and getattr(prev, "bn_capable", False) and pd.y == x_raw.data_ptr() and pd.out_cs == cs
                 and pd.out_co == co and pd.N == C and pd.B * pd.Ho * pd.Wo == M and (pd.OH, pd.OW) == (pd.Ho, pd.Wo))
        if fused:
            prev.bn_stats = True
            # rows the launch can write at most: two wave rows per 64-row tile (the kernels report the real count at run time)
            self._scratch_need = max(getattr(self, "_scratch_need", 0), ((M + 63) // 64) * 2 * C * 2)

        def fwd():
            if fused and prev.stats_rows > 0:
                assert prev.stats_rows * C * 2 <= self._scratch_d.numel(), name
                _lib.check(lib.rdpn6d_bn_stats_finalize(_ptr(self._scratch_d), prev.stats_rows, C, M, BN_EPS, BN_MOM, _ptr(mean),
                                                        _ptr(invstd), _ptr(bn.running_mean), _ptr(bn.running_var), self.st()),
                           "bn stats (finalize) " + name)
            else:
                _lib.check(f_stats(_ptr(x_raw), M, C, cs, co, BN_EPS, BN_MOM, _ptr(mean), _ptr(invstd), _ptr(bn.running_mean),
                                   _ptr(bn.running_var), _ptr(self._scratch_d), self.st()), "bn stats " + name)
            _lib.check(f_apply(_ptr(x_raw), cs, co, _ptr(mean), _ptr(invstd), _ptr(ga), _ptr(be), _ptr(res), res_cs, 0, _ptr(y), ycs, yco,
                               M, C, 1 if relu else 0, self.st()), "bn apply " + name)

        self.fwd.append(fwd)
        self._bn_counters.append(bn.num_batches_tracked)
        if dy is None:
            dy, dy_cs, dy_co = self.buf("d:" + name, *y.shape, zero=True, dtype=y.dtype), ycs, yco
        else:
            dy_co = 0
        assert C % 4 == 0

        remask = relu and res is None and dres is None and self.bn_remask
        f_bwd_remask = getattr(lib, f"rdpn6d_bn_relu_backward_{t}")
        # the convolution that CONSUMES y registers after this unit: if its input-gradient launch can also produce this BatchNorm's
        # backward sums (conv_unit), the reduction pass over dy and x_raw is skipped here
        res_form = relu and res is not None and dres is not None  # y = relu(bn(x) + identity): the mask is the stored y
        rec = dict(x_raw=x_raw, cs=cs, co=co, C=C, M=M, mean=mean, invstd=invstd, ga=ga, be=be, dy_cs=dy_cs, dy_co=dy_co, producer=None,
                   y=y if res_form else None, ycs=ycs, yco=yco)
        self.records.append(dict(kind="bn", name=name, bn=bn, x_raw=x_raw, cs=cs, co=co, C=C, M=M, y=y, ycs=ycs, yco=yco, relu=relu, res=res,
                                 res_cs=res_cs, dx=dx, dres=dres, dy=dy, dy_cs=dy_cs, dy_co=dy_co))
        if (remask or (res_form and self.bn_fuse_bwd_res)) and t != "f32" and self.bn_fuse_bwd:
            self._bn_by_dy[dy.data_ptr()] = rec

        def bwd():
            # dgamma / dbeta land directly in the parameters' gradients (C entries each, also read back by the dx pass)
            prod = rec["producer"]
            if prod is not None and prod.stats_rows > 0:
                if getattr(self, "_bnb_owner", None) is not rec:
                    # every fused producer writes the same scratch rows: the consumer must be the NEXT fused user after its producer
                    # (true for the chained ResNet / head topologies; a re-ordered _build would otherwise read another layer's sums)
                    raise RuntimeError(f"fused BatchNorm backward of {name}: the shared partial-sum buffer was overwritten by another "
                                       "layer's input-gradient convolution before this BatchNorm consumed it (set SOLVER.BN_FUSE_BWD=False)")
                if rec["y"] is not None:
                    _lib.check(self.lpf("rdpn6d_bn_backward_apply_bf16")(
                        _ptr(x_raw), cs, co, _ptr(dy), dy_cs, dy_co, _ptr(y), ycs, yco, _ptr(mean), _ptr(invstd), _ptr(ga),
                        _ptr(self._grad(bn.weight)), _ptr(self._grad(bn.bias)), _ptr(dx), cs, co, _ptr(dres), dres.shape[-1], 0, M, C,
                        _ptr(self._scratch_bnb), prod.stats_rows, self.st()), "bn bwd (apply) " + name)
                    return
                _lib.check(self.lpf("rdpn6d_bn_relu_backward_apply_bf16")(
                    _ptr(x_raw), cs, co, _ptr(dy), dy_cs, dy_co, _ptr(mean), _ptr(invstd), _ptr(ga), _ptr(be), _ptr(self._grad(bn.weight)),
                    _ptr(self._grad(bn.bias)), _ptr(dx), cs, co, M, C, _ptr(self._scratch_bnb), prod.stats_rows, self.st()),
                    "bn+relu bwd (apply) " + name)
                return
            if remask:
                _lib.check(f_bwd_remask(_ptr(x_raw), cs, co, _ptr(dy), dy_cs, dy_co, _ptr(mean), _ptr(invstd), _ptr(ga), _ptr(be),
                                        _ptr(self._grad(bn.weight)), _ptr(self._grad(bn.bias)), _ptr(dx), cs, co, M, C,
                                        _ptr(self._scratch_d), self.st()), "bn+relu bwd " + name)
                return
            _lib.check(f_bwd(_ptr(x_raw), cs, co, _ptr(dy), dy_cs, dy_co, _ptr(y), ycs, yco, _ptr(mean), _ptr(invstd), _ptr(ga),
                             _ptr(self._grad(bn.weight)), _ptr(self._grad(bn.bias)), _ptr(dx), cs, co, _ptr(dres),
                             (dres.shape[-1] if dres is not None else 0), 0, M, C, 1 if relu else 0, _ptr(self._scratch_d), self.st()),
                       "bn bwd " + name)

        self.bwd.append([bwd])
        return dy

    # ------------------------------------------------------------------ the network
    def _build(self):
        m, B, R, K, lib = self.model, self.B, self.R, self.K, self.lib
        bb, head, pnp = m.backbone, m.rot_head_net, m.pnp_net
        R2, R4, R8 = R // 2, R // 4, R // 8
        self.x = self.buf("x", B, 6, R, R)
        # ---- stem
        # [64][7][7][3] <- OIHW [64][3][7][7]: row o' = n*49 + tap reads src[n*147 + tap + c*49]
        wst = self._pack_map((64, 7, 7, 3), bb.conv1.weight, 64 * 49, 1, 3, 1, 49, [0],
                             operm=[n * 147 + t for n in range(64) for t in range(49)], dT=1, dIpad=3)
        raw0 = self.buf("raw:stem", B, R2, R2, 64)
        a0 = self.buf("act:stem", B, R2, R2, 64)
        sfx = self.sfx  # "bf16" under AMP: the kernels between the convolutions work on bf16-stored activations
        f_stem = getattr(lib, f"rdpn6d_stem_conv7x7_raw_{sfx}")
        self._stem_h2 = None
        if self.amp and R % 4 == 0 and bool(self.model.cfg.get("SOLVER", {}).get("MFMA_STEM", os.environ.get("RDPN6D_MFMA_STEM", "1") != "0")):
            # mixed precision: the raw stem convolution on the matrix pipe - the inference front's kernel (fp32-accurate h2 arithmetic on
            # the patch in LDS) without ReLU / pooling, 16-bit output; its weight record is re-packed by refresh_weights (one launch)
            wh = torch.empty(64, 6, 2, 32, dtype=torch.float16, device=self.dev)
            inv, zero = torch.empty(64, dtype=torch.float32, device=self.dev), torch.zeros(64, dtype=torch.float32, device=self.dev)
            self._stem_h2 = (bb.conv1.weight, wh, inv)
            fmt = 3 if self.lp == "bf16" else 4
            self.fwd.append(lambda: _lib.check(lib.rdpn6d_stem_pool_h2_ex(_ptr(self.x), B, 6, R, _ptr(wh), _ptr(inv), _ptr(zero), _ptr(raw0), fmt,
                                                                          None, self.st()), "stem (MFMA)"))
        else:
            self.fwd.append(lambda: _lib.check(f_stem(_ptr(self.x), B, 6, R, _ptr(wst), _ptr(raw0), self.st()), "stem"))
        d_raw0 = self.buf("d:stem", B, R2, R2, 64, zero=True)
        # dW(conv1): the stem's row-patch matrix (horizontal taps unrolled, the two input-row parities side by side: 64 columns) and ONE
        # stride-1 four-tap weight gradient over it - rdpn6d_stem_rowpatch_*; 67 MB + one read of the output gradient instead of the
        # 160-column patch matrix (168 MB, three column tiles): 81 + 124 us -> see profiles/r5_experiments.md
        xrow = self.buf("x_rowpatch", B * R2 * R2, 64, dtype=self.adt)   # (built in the backward)
        wg_stem = self.buf("wg:stem", 64, 4, 64)
        self._wg_floats = max(self._wg_floats, int(lib.rdpn6d_wgrad_scratch_floats(B, R2, R2, 64, 64, 4)))
        t_dy, t_dx = (ctypes.c_int * 9)(-2, -1, 0, 1, 0, 0, 0, 0, 0), (ctypes.c_int * 9)(*([0] * 9))

        def stem_wgrad():
            _lib.check(getattr(lib, f"rdpn6d_stem_rowpatch_{sfx}")(_ptr(self.x), B, 6, R, _ptr(xrow), self.st()), "stem row patches")
            if self.amp:
                _lib.check(self.lpf("rdpn6d_wgrad_bf16")(_ptr(d_raw0), 64, 0, 64, 64, _ptr(xrow), 64, 0, 64, 64, B, R2, R2, R2, R2, 1, 4, t_dy, t_dx,
                                                 _ptr(wg_stem), _ptr(self._wg_partial), self.st()), "wgrad stem")
            else:
                _lib.check(lib.rdpn6d_wgrad_f32(_ptr(d_raw0), 64, 0, 64, _ptr(xrow), 64, 0, 64, B, R2, R2, R2, R2, 1, 4, t_dy, t_dx,
                                                _ptr(wg_stem), _ptr(self._wg_partial), self.st()), "wgrad stem")
            # [n][t][r][kx*3 + c] -> ky + 1 = 2t + r (entry 0 = the unused ky = -1) -> [n][c][ky][kx]
            g = wg_stem.view(64, 4, 2, 32)[..., :21].reshape(64, 8, 7, 3)[:, 1:]
            self._grad(bb.conv1.weight).copy_(g.permute(0, 3, 1, 2))

        self.bwd.append([stem_wgrad])
        self.records.append(dict(kind="stem", name="backbone.conv1", P=bb.conv1, x=self.x, y=raw0, dy=d_raw0, lowp=self.amp))
        d_a0 = self.bn_unit("bn1", bb.bn1, raw0, 64, 0, 64, B * R2 * R2, a0, 64, 0, True, dx=d_raw0)
        p0 = self.buf("act:pool", B, R4, R4, 64)
        f_pool, f_pool_b = getattr(lib, f"rdpn6d_maxpool3x3s2_{sfx}"), getattr(lib, f"rdpn6d_maxpool3x3s2_backward_{sfx}")
        self.fwd.append(lambda: _lib.check(f_pool(_ptr(a0), B, R2, R2, 64, _ptr(p0), self.st()), "maxpool"))
        d_p0 = self.buf("d:pool", B, R4, R4, 64, zero=True)
        self.bwd.append([lambda: _lib.check(f_pool_b(_ptr(a0), _ptr(d_p0), B, R2, R2, 64, _ptr(d_a0), self.st()), "maxpool bwd")])
        # ---- residual trunk
        cur, d_cur, hw, c = p0, d_p0, R4, 64
        stage_first = {}  # first launch list of a ResNet stage = the LAST one its backward runs (the stage's grouped weight gradient included)
        for li in range(4):
            stage_first[li + 1] = len(self.bwd)
            for bi, blk in enumerate(getattr(bb, f"layer{li + 1}")):
                nm = f"layer{li + 1}.{bi}"
                if hasattr(blk, "conv3"):  # Bottleneck: 1x1 - 3x3(s) - 1x1(x4)
                    width, s, cout = blk.conv1.weight.shape[0], blk.conv2.stride, blk.conv3.weight.shape[0]
                    ohw = hw // s
                    M1, M = B * hw * hw, B * ohw * ohw
                    r1, a1 = self.buf(f"raw:{nm}.c1", B, hw, hw, width), self.buf(f"act:{nm}.c1", B, hw, hw, width)
                    r2, a2 = self.buf(f"raw:{nm}.c2", B, ohw, ohw, width), self.buf(f"act:{nm}.c2", B, ohw, ohw, width)
                    r3, out = self.buf(f"raw:{nm}.c3", B, ohw, ohw, cout), self.buf(f"act:{nm}", B, ohw, ohw, cout)
                    d_out = self.buf(f"d:{nm}.out", B, ohw, ohw, cout, zero=True)
                    dres = self.buf(f"dres:{nm}", B, ohw, ohw, cout)
                    has_ds = blk.downsample is not None
                    res_t = cur
                    if has_ds:
                        rd, ad = self.buf(f"raw:{nm}.ds", B, ohw, ohw, cout), self.buf(f"act:{nm}.ds", B, ohw, ohw, cout)
                        # registered first = runs last in the backward: accumulates onto conv1's input gradient
                        d_rd = self.conv_unit(f"{nm}.downsample.0", blk.downsample[0], cur, (hw, hw), c, 0, c, rd, (ohw, ohw), cout, 0,
                                              stride=s, dx=d_cur, dx_res=d_cur if s == 1 else None)
                        res_t = ad
                    d_r1 = self.conv_unit(f"{nm}.conv1", blk.conv1, cur, (hw, hw), c, 0, c, r1, (hw, hw), width, 0, dx=d_cur,
                                          dx_res=None if has_ds else dres, dx_final=not has_ds)
                    if has_ds:
                        self.bn_unit(f"{nm}.downsample.1", blk.downsample[1], rd, cout, 0, cout, M, ad, cout, 0, False, dx=d_rd,
                                     dy=dres, dy_cs=cout)
                    d_a1 = self.bn_unit(f"{nm}.bn1", blk.bn1, r1, width, 0, width, M1, a1, width, 0, True, dx=d_r1)
                    d_r2 = self.conv_unit(f"{nm}.conv2", blk.conv2, a1, (hw, hw), width, 0, width, r2, (ohw, ohw), width, 0, stride=s,
                                          dx=d_a1)
                    d_a2 = self.bn_unit(f"{nm}.bn2", blk.bn2, r2, width, 0, width, M, a2, width, 0, True, dx=d_r2)
                    d_r3 = self.conv_unit(f"{nm}.conv3", blk.conv3, a2, (ohw, ohw), width, 0, width, r3, (ohw, ohw), cout, 0, dx=d_a2)
                    self.bn_unit(f"{nm}.bn3", blk.bn3, r3, cout, 0, cout, M, out, cout, 0, True, res=res_t, res_cs=cout, dx=d_r3,
                                 dres=dres, dy=d_out, dy_cs=cout)
                    cur, d_cur, hw, c = out, d_out, ohw, cout
                    continue
                cout, s = blk.conv1.weight.shape[0], blk.conv1.stride
                ohw = hw // s
                M = B * ohw * ohw
                r1, a1 = self.buf(f"raw:{nm}.c1", B, ohw, ohw, cout), self.buf(f"act:{nm}.c1", B, ohw, ohw, cout)
                r2, out = self.buf(f"raw:{nm}.c2", B, ohw, ohw, cout), self.buf(f"act:{nm}", B, ohw, ohw, cout)
                d_out = self.buf(f"d:{nm}.out", B, ohw, ohw, cout, zero=True)  # gradient w.r.t. the block output
                dres = self.buf(f"dres:{nm}", B, ohw, ohw, cout)
                has_ds = blk.downsample is not None
                # registration order = forward order; the backward runs it in reverse:
                #   bn2 -> conv2 -> bn1 -> [ds.bn] -> conv1 (dgrad writes d_cur, + identity gradient) -> [ds.conv (accumulates)]
                res_t = cur
                if has_ds:
                    rd, ad = self.buf(f"raw:{nm}.ds", B, ohw, ohw, cout), self.buf(f"act:{nm}.ds", B, ohw, ohw, cout)
                    d_rd = self.conv_unit(f"{nm}.downsample.0", blk.downsample[0], cur, (hw, hw), c, 0, c, rd, (ohw, ohw), cout, 0,
                                          stride=s, dx=d_cur)
                    res_t = ad
                d_r1 = self.conv_unit(f"{nm}.conv1", blk.conv1, cur, (hw, hw), c, 0, c, r1, (ohw, ohw), cout, 0, stride=s,
                                      dx=d_cur, dx_res=None if has_ds else dres, dx_final=not has_ds)
                if has_ds:
                    self.bn_unit(f"{nm}.downsample.1", blk.downsample[1], rd, cout, 0, cout, M, ad, cout, 0, False, dx=d_rd,
                                 dy=dres, dy_cs=cout)
                d_a1 = self.bn_unit(f"{nm}.bn1", blk.bn1, r1, cout, 0, cout, M, a1, cout, 0, True, dx=d_r1)
                d_r2 = self.conv_unit(f"{nm}.conv2", blk.conv2, a1, (ohw, ohw), cout, 0, cout, r2, (ohw, ohw), cout, 0, dx=d_a1)
                self.bn_unit(f"{nm}.bn2", blk.bn2, r2, cout, 0, cout, M, out, cout, 0, True, res=res_t, res_cs=cout, dx=d_r2,
                             dres=dres, dy=d_out, dy_cs=cout)
                cur, d_cur, hw, c = out, d_out, ohw, cout
        # ---- upsample + point-wise fusion
        C4 = c  # layer4 channels: 512 (BasicBlock) | 2048 (Bottleneck)
        up = self.buf("act:up", B, R8, R8, C4)
        f = R8 // hw
        l4out, d_l4out, l4hw = cur, d_cur, hw
        f_up, f_up_b = getattr(lib, f"rdpn6d_upsample_bilinear_{sfx}"), getattr(lib, f"rdpn6d_upsample_bilinear_backward_{sfx}")
        self.fwd.append(lambda: _lib.check(f_up(_ptr(l4out), B, l4hw, l4hw, C4, f, _ptr(up), self.st()), "upsample"))
        d_up = self.buf("d:up", B, R8, R8, C4, zero=True)
        self.bwd.append([lambda: _lib.check(f_up_b(_ptr(d_up), B, l4hw, l4hw, C4, f, _ptr(d_l4out), self.st()), "upsample bwd")])
        sn = bb.spatial_net
        Mp = B * R8 * R8
        pcs = 96 if self.amp else 80  # [emb(64) | xyz(3) | 0-pad] to the K-chunk granularity of the conv kernel in use
        pin = self.buf("act:pn_in", B, R8, R8, pcs, zero=True)
        d_pin = self.buf("d:pn_in", B, R8, R8, pcs, zero=True)
        f_xyz = getattr(lib, f"rdpn6d_xyz_subsample_{sfx}")
        self.fwd.append(lambda: _lib.check(f_xyz(_ptr(self.x), B, 6, R, 8, _ptr(pin), pcs, 64, self.st()), "xyz"))
        r_e = self.buf("raw:pn.emb", B, R8, R8, 64)
        d_re = self.conv_unit("spatial_net.xyz_emb", sn.xyz_emb, up, (R8, R8), C4, 0, C4, r_e, (R8, R8), 64, 0, bias=sn.xyz_emb.bias, dx=d_up)
        self.bn_unit("spatial_net.xb", sn.xb, r_e, 64, 0, 64, Mp, pin, pcs, 0, True, dx=d_re, dy=d_pin, dy_cs=pcs)
        r1p, a1p = self.buf("raw:pn.c1", B, R8, R8, 128), self.buf("act:pn.c1", B, R8, R8, 128)
        perm = list(range(3, 67)) + [0, 1, 2]
        d_r1p = self.conv_unit("spatial_net.conv1", sn.conv1, pin, (R8, R8), pcs, 0, 67, r1p, (R8, R8), 128, 0, perm=perm,
                               bias=sn.conv1.bias, dx=d_pin)
        d_a1p = self.bn_unit("spatial_net.b1", sn.b1, r1p, 128, 0, 128, Mp, a1p, 128, 0, True, dx=d_r1p)
        r2p, a2p = self.buf("raw:pn.c2", B, R8, R8, 256), self.buf("act:pn.c2", B, R8, R8, 256)
        d_r2p = self.conv_unit("spatial_net.conv2", sn.conv2, a1p, (R8, R8), 128, 0, 128, r2p, (R8, R8), 256, 0, bias=sn.conv2.bias, dx=d_a1p)
        d_a2p = self.bn_unit("spatial_net.b2", sn.b2, r2p, 256, 0, 256, Mp, a2p, 256, 0, True, dx=d_r2p)
        r3p = self.buf("raw:pn.c3", B, R8, R8, 512)
        feat = self.buf("act:feat", B, R8, R8, 1024)
        d_feat = self.buf("d:feat", B, R8, R8, 1024, zero=True)
        d_l3 = self.buf("d:l3", B, R8, R8, 512, zero=True)
        d_r3p = self.conv_unit("spatial_net.conv3", sn.conv3, a2p, (R8, R8), 256, 0, 256, r3p, (R8, R8), 512, 0, bias=sn.conv3.bias, dx=d_a2p)
        # b3 writes channels [0,512) of feat; its output gradient arrives as a dense [.,512] tensor from the gmax backward
        self.bn_unit("spatial_net.b3", sn.b3, r3p, 512, 0, 512, Mp, feat, 1024, 0, False, dx=d_r3p, dy=d_l3, dy_cs=512)
        f_gm, f_gm_b = getattr(lib, f"rdpn6d_global_max_concat_{sfx}"), getattr(lib, f"rdpn6d_global_max_concat_backward_{sfx}")
        self.fwd.append(lambda: _lib.check(f_gm(_ptr(feat), B, R8 * R8, 512, 1024, self.st()), "gmax"))
        self.bwd.append([lambda: _lib.check(f_gm_b(_ptr(feat), _ptr(d_feat), B, R8 * R8, 512, 1024, _ptr(d_l3), self.st()), "gmax bwd")])
        # ---- dense head: ConvTranspose as 4 phase convs (forward), stride-2 conv (dgrad), gathered wgrad
        # bwd index after which a stage's parameter gradients are complete (parallel.STAGES, the order the backward finishes them):
        # layer4 / layer3 as soon as their own launches are through - their all-reduces then overlap the rest of the trunk's backward
        self._group_marks = {0: "backbone.rest", stage_first[3]: "backbone.layer3", stage_first[4]: "backbone.layer4",
                             len(self.bwd): "rot_head_net"}
        F = head.features[0].weight.shape[1]
        Mh = B * R4 * R4
        rt0, at0 = self.buf("raw:head0", B, R4, R4, F), self.buf("act:head0", B, R4, R4, F)
        wt = head.features[0].weight
        # fp32 mode, batches that fill the chip: the ConvTranspose (four phase convolutions, its stride-2 input-gradient convolution
        # and its weight gradient) on the bf16x3 kernels as well - two split passes (feat, d_rt0) pay for three GEMMs
        x3_ct = self.x3 and F % 256 == 0 and B * R8 * R8 >= 16384 and 6 * B * R8 * R8 * 1024 < (1 << 32) - 64
        for py in (0, 1):
            for px in (0, 1):
                ys = [(1, 0)] if py == 0 else [(0, 1), (2, 0)]
                xs = [(1, 0)] if px == 0 else [(0, 1), (2, 0)]
                ptaps = [(dyo, dxo) for _, dyo in ys for _, dxo in xs]
                kk = [(ky, kx) for ky, _ in ys for kx, _ in xs]

                # dst[f][ti][cin] = wt[cin][f][ky][kx]
                wp = self._pack_map((_pad_to(F, 64), len(ptaps), 1024), wt, F, len(ptaps), 1024, 9, F * 9, [ky * 3 + kx for ky, kx in kk])
                if self.amp:
                    featb = feat  # bf16-stored
                    wpb = self._mirror(wp)
                    d = self._conv_desc(featb, (R8, R8), 1024, 0, 1024, wpb, rt0, (R4, R4), F, 0, F, ptaps, phase=(R8, R8, 2, 2, py, px))
                    self.fwd.append(self._launch_conv(f"convT phase{py}{px}", d, (wp, wpb, featb), lowp=True, out_f32=False))
                else:
                    d = self._conv_desc(feat, (R8, R8), 1024, 0, 1024, wp, rt0, (R4, R4), F, 0, F, ptaps, phase=(R8, R8, 2, 2, py, px))
                    if x3_ct:  # bf16x3 (tile kernel at this row count): planes of feat, split once for the four phases
                        featp = self._planes_of(self.fwd, feat, cache=True)
                        wp3 = self._mirror3(wp)
                        d.x, d.w = _ptr(featp), _ptr(wp3)
                        self.fwd.append(self._launch_conv_x3(f"convT phase{py}{px}", d, featp, wp3, (wp, feat)))
                    else:
                        self.fwd.append(self._launch_conv(f"convT phase{py}{px}", d, wp))
        d_rt0 = self.buf("d:head0", B, R4, R4, F, zero=True)
        wdT = self._pack_map((1024, 9, F), wt, 1024, 9, F, F * 9, 9, range(9))  # ConvT weight IS the OIHW of its dgrad conv
        convT_bwd = []
        if self.amp:
            assert F % 32 == 0
            d_rt0b = d_rt0  # bf16-stored
            wdTb = self._mirror(wdT)
            ddT = self._conv_desc(d_rt0b, (R4, R4), F, 0, F, wdTb, d_feat, (R8, R8), 1024, 0, 1024, _taps(3, 1), stride=2)
        else:
            ddT = self._conv_desc(d_rt0, (R4, R4), F, 0, F, wdT, d_feat, (R8, R8), 1024, 0, 1024, _taps(3, 1), stride=2)
        d_rt0p = None
        if x3_ct:
            x3_ct = bool(lib.rdpn6d_conv_bf16x3_eligible(ctypes.byref(ddT)))
        if x3_ct:
            d_rt0p = self._planes_of(convT_bwd, d_rt0, cache=False)  # first launch of the group: split the output gradient
        self._wg_floats = max(self._wg_floats, int(lib.rdpn6d_wgrad_scratch_floats(B, R8, R8, 1024, F, 9)))
        t9y = (ctypes.c_int * 9)(*[t[0] for t in _taps(3, 1)])
        t9x = (ctypes.c_int * 9)(*[t[1] for t in _taps(3, 1)])

        def convT_wgrad():
            # ConvTranspose weight is (Cin, Cout, 3, 3): element (cin, tap, cout) -> cin*F*9 + cout*9 + tap
            tgt = (_ptr(self._grad(wt)), F * 9, 1, 9, 1024, F)
            if x3_ct:
                _lib.check(lib.rdpn6d_wgrad_bf16x3_strided(_ptr(featp), featp.shape[1], 1024, 0, 1024, 1024, _ptr(d_rt0p), d_rt0p.shape[1], F,
                                                           0, F, F, B, R8, R8, R4, R4, 2, 9, t9y, t9x, *tgt, _ptr(self._wg_partial),
                                                           self.st()), "wgrad convT")
            elif self.amp:
                _lib.check(self.lpf("rdpn6d_wgrad_bf16_strided")(_ptr(featb), 1024, 0, 1024, 1024, _ptr(d_rt0b), F, 0, F, F, B, R8, R8, R4, R4, 2,
                                                         9, t9y, t9x, *tgt, _ptr(self._wg_partial), self.st()), "wgrad convT")
            else:
                _lib.check(lib.rdpn6d_wgrad_f32_strided(_ptr(feat), 1024, 0, 1024, _ptr(d_rt0), F, 0, F, B, R8, R8, R4, R4, 2, 9, t9y,
                                                        t9x, *tgt, _ptr(self._wg_partial), self.st()), "wgrad convT")

        if x3_ct:
            wdT3 = self._mirror3(wdT)
            ddT.x, ddT.w = _ptr(d_rt0p), _ptr(wdT3)
            dgradT = self._launch_conv_x3("dgrad convT", ddT, d_rt0p, wdT3, (wdT, d_rt0))
        else:
            dgradT = self._launch_conv("dgrad convT", ddT, wdT, lowp=self.amp, out_f32=not self.amp)
        self.bwd.append(convT_bwd + [convT_wgrad, dgradT])
        self.records.append(dict(kind="convT", name="rot_head_net.features.0", P=head.features[0], x=feat, y=rt0, dy=d_rt0, dx=d_feat, lowp=self.amp))
        d_prev = self.bn_unit("head.bn0", head.features[1], rt0, F, 0, F, Mh, at0, F, 0, True, dx=d_rt0)
        a_prev = at0
        nfeat = len(head.features)
        for i in range(3, nfeat - 1, 3):
            r_i, a_i = self.buf(f"raw:head{i}", B, R4, R4, F), self.buf(f"act:head{i}", B, R4, R4, F)
            d_ri = self.conv_unit(f"rot_head.features.{i}", head.features[i], a_prev, (R4, R4), F, 0, F, r_i, (R4, R4), F, 0, dx=d_prev)
            d_prev = self.bn_unit(f"head.bn{i}", head.features[i + 1], r_i, F, 0, F, Mh, a_i, F, 0, True, dx=d_ri)
            a_prev = a_i
        last = head.features[nfeat - 1]
        nout = last.weight.shape[0]
        MC = 2 if self.mask_type == 2 else 1
        if nout != MC + 4 + K:  # (head_cs is padded: the C-side width check would pass a head one mask channel off, and the glue / loss /
            #                      backward kernels would then read xyz and region one channel plane off, silently - same rule as InferencePlan)
            raise ValueError(f"the head's output convolution has {nout} channels; ROT_HEAD.MASK_LOSS_TYPE={self.model.cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE!r} "
                             f"with NUM_REGIONS={K} needs {MC} + 3 + {K + 1} (GDRN.py:637-659): build the model with the config it is run with")
        self.head_cs = _pad_to(nout, 16)
        ho = self.buf("act:head_out", B, R4 * R4, self.head_cs, zero=True)
        self.head_out = ho
        d_ho = self.conv_unit("rot_head.out", last, a_prev, (R4, R4), F, 0, F, ho.view(B, R4, R4, self.head_cs), (R4, R4), self.head_cs, 0,
                              bias=last.bias, dx=d_prev)
        self.d_head = d_ho
        # ---- glue
        HW = R4 * R4
        self.pnp_cs = _pad_to(11 + K, 16)
        self.out_nchw = self.buf("out_nchw", B, nout, R4, R4)
        pnp_in = self.buf("act:pnp_in", B, R4, R4, self.pnp_cs)
        d_pnp_in = self.buf("d:pnp_in", B, R4, R4, self.pnp_cs, zero=True)
        self.argmax = self.buf("argmax", B, HW, dtype=torch.int32)
        self.minmax = self.buf("minmax", B, 2)
        datt = self.buf("datt", B, HW)
        self.coord2d = self.buf("coord2d", B, 5, R4, R4)
        self.fps_t = self.buf("fps", B, K, 3)
        self.fwd.append(lambda: _lib.check(lib.rdpn6d_dense_glue_mt_f32(_ptr(ho), self.head_cs, _ptr(self.coord2d), _ptr(self.fps_t), B, HW, K,
                                                                     self.mask_attention, self.mask_type, _ptr(self.minmax), _ptr(self.out_nchw), _ptr(pnp_in),
                                                                  self.pnp_cs, _ptr(self.argmax), self.st()), "glue"))
        self.bwd.append([lambda: _lib.check(lib.rdpn6d_dense_glue_backward_mt_f32(_ptr(ho), self.head_cs, _ptr(self.coord2d), _ptr(self.fps_t),
                                                                               _ptr(self.argmax), _ptr(d_pnp_in), self.pnp_cs, B, HW, K,
                                                                               self.mask_attention, self.mask_type, _ptr(self.minmax), _ptr(d_ho),
                                                                               _ptr(datt), self.st()), "glue bwd")])
        # ---- ConvPnPNet
        self._group_marks[len(self.bwd)] = "pnp_net"
        x, d_x, hw, cin_real, cs = pnp_in, d_pnp_in, R4, 11 + K, self.pnp_cs
        for i in range(0, 9, 3):
            conv, gn = pnp.features[i], pnp.features[i + 1]
            fd = conv.weight.shape[0]
            oh = hw // 2
            r_i, a_i = self.buf(f"raw:pnp{i}", B, oh, oh, fd), self.buf(f"act:pnp{i}", B, oh, oh, fd)
            d_ri = self.conv_unit(f"pnp_net.features.{i}", conv, x, (hw, hw), cs, 0, cin_real, r_i, (oh, oh), fd, 0, stride=2, dx=d_x)
            stats = self.buf(f"gnstats:{i}", B, gn.groups, 2)
            ga, be = gn.weight, gn.bias  # read in place
            self.fwd.append(lambda r_i=r_i, a_i=a_i, oh=oh, fd=fd, gn=gn, ga=ga, be=be, stats=stats: _lib.check(
                lib.rdpn6d_groupnorm_relu_train_f32(_ptr(r_i), _ptr(a_i), B, oh * oh, fd, gn.groups, _ptr(ga), _ptr(be), _ptr(stats), self.st()), "gn"))
            d_ai = self.buf(f"d:pnp_act{i}", B, oh, oh, fd, zero=True)
            dgb = self.buf(f"dgb:pnp{i}", B, 2, fd)

            def gn_bwd(r_i=r_i, a_i=a_i, d_ai=d_ai, d_ri=d_ri, oh=oh, fd=fd, gn=gn, ga=ga, stats=stats, dgb=dgb):
                _lib.check(lib.rdpn6d_groupnorm_relu_backward_f32(_ptr(r_i), _ptr(a_i), _ptr(d_ai), _ptr(ga), _ptr(stats), _ptr(d_ri),
                                                                  _ptr(self._grad(gn.weight)), _ptr(self._grad(gn.bias)), _ptr(dgb),
                                                                  _ptr(self._scratch_d), B, oh * oh, fd, gn.groups, self.st()), "gn bwd")

            self.bwd.append([gn_bwd])
            x, d_x, hw, cin_real, cs = a_i, d_ai, oh, fd, fd
        # FC stack (1x1 convs over B "pixels"); LeakyReLU(0.1) fused in the forward epilogue, masked in backward
        kin = cs * hw * hw
        f1, f2 = self.buf("act:fc1", B, 1024), self.buf("act:fc2", B, 256)
        self.rt = self.buf("act:rt", B, 16, zero=True)
        d_f1, d_f2 = self.buf("d:fc1", B, 1024, zero=True), self.buf("d:fc2", B, 256, zero=True)
        self.d_rt = self.buf("d:rt", B, 16, zero=True)
        # fc1 reads the map in the reference's NCHW-flatten order (conv_pnp_net.py:151): a 1-MB transposition of the activation (and of
        # its gradient on the way back) instead of a permutation of fc1's 8.4 M weights in every re-pack and of their gradient - the
        # permuted pack read one float per 128-byte line (round 5: ~100 us of the 218-us re-pack, 29 us of gradient copy)
        x_t, d_xt = self.buf("act:pnp_flat", B, cs, hw * hw), self.buf("d:pnp_flat", B, cs, hw * hw, zero=True)
        self.fwd.append(lambda x=x, x_t=x_t: _lib.check(lib.rdpn6d_transpose_rc_f32(_ptr(x), B, hw * hw, cs, _ptr(x_t), self.st()), "flatten"))
        self.bwd.append([lambda d_x=d_x, d_xt=d_xt: _lib.check(lib.rdpn6d_transpose_rc_f32(_ptr(d_xt), B, cs, hw * hw, _ptr(d_x), self.st()),
                                                              "flatten bwd")])
        self._fc("fc1", pnp.fc1.weight, pnp.fc1.bias, x_t.view(B, kin), d_xt.view(B, kin), kin, f1, d_f1, 1024, act=2)
        self._fc("fc2", pnp.fc2.weight, pnp.fc2.bias, f1, d_f1, 1024, f2, d_f2, 256, act=2)
        self._fc("fc_rt", (pnp.fc_r.weight, pnp.fc_t.weight), (pnp.fc_r.bias, pnp.fc_t.bias), f2, d_f2, 256, self.rt, self.d_rt, 9, act=0, out_cs=16)
        self.rot, self.trans = self.buf("rot", B, 3, 3), self.buf("trans", B, 3)
        self.gt_rot_used = self.buf("gt_rot_used", B, 3, 3)  # PM target after the symmetry choice
        self.losses9 = self.buf("losses", 16, zero=True)

    def _fc(self, name, w, b, x, d_x, kin, y, d_y, nout, act, out_cs=None, w_view=None, g_view=None):
        lib, B = self.lib, self.B
        ws = w if isinstance(w, tuple) else (w,)
        bs = b if isinstance(b, tuple) else (b,)
        out_cs = out_cs or nout
        npad = _pad_to(nout, 64)

        # k_perm[k'] = position in the parameter's own (NCHW-flatten) K axis of the buffer's (NHWC-flatten) column k'
        k_perm = None
        if w_view is not None:
            k_perm = w_view(torch.arange(kin, dtype=torch.float32).view(1, kin))[0].long().tolist()
        wf = torch.zeros(npad, 1, kin, dtype=torch.float32, device=self.dev)
        bvec = torch.zeros(npad, dtype=torch.float32, device=self.dev)
        r0 = 0
        for wp_, bp_ in zip(ws, bs):
            n = wp_.shape[0]
            self._pack_map(None, wp_, n, 1, kin, kin, 1, [0], iperm=k_perm, dst=wf, dst_off=r0 * kin, dT=1, dIpad=kin)
            self._pack_map(None, bp_, 1, 1, n, 0, 1, [0], dst=bvec, dst_off=r0, dT=1, dIpad=npad)
            r0 += n
        d = self._conv_desc(x, (1, 1), kin, 0, kin, wf, y, (1, 1), out_cs, 0, nout, [(0, 0)], shift=bvec, act=act, slope=0.1)
        self.fwd.append(self._launch_conv(name, d, (wf, bvec), ksplit=True))
        ca = _pad_to(nout, 4)
        wg_out = self.buf("wg:" + name, ca, 1, kin)
        self._wg_floats = max(self._wg_floats, int(lib.rdpn6d_wgrad_scratch_floats(B, 1, 1, ca, kin, 1)))
        z9 = (ctypes.c_int * 9)(*([0] * 9))
        bg = self.buf("bg:" + name, npad)
        n_red = _pad_to(nout, 16)
        assert out_cs >= n_red
        wd = torch.zeros(_pad_to(kin, 64), 1, n_red, dtype=torch.float32, device=self.dev)  # dst[k'][0][r0+j] = W_p[j][k_perm(k')]
        r0 = 0
        for wp_ in ws:
            n = wp_.shape[0]
            self._pack_map(None, wp_, kin, 1, n, 1, kin, [0], operm=k_perm, dst=wd, dst_off=r0, dT=1, dIpad=n_red)
            r0 += n
        dd = self._conv_desc(d_y, (1, 1), out_cs, 0, n_red, wd, d_x, (1, 1), kin, 0, kin, [(0, 0)])

        def bwd():
            if act == 2:
                _lib.check(lib.rdpn6d_act_backward_f32(_ptr(d_y), _ptr(y), B * out_cs, 0.1, self.st()), "leaky bwd " + name)
            # one weight in the buffer's own K order: the [nout][kin] result IS the parameter's gradient - written in place (fc1: 34 MB
            # not copied)
            direct = len(ws) == 1 and g_view is None and ca == nout and tuple(ws[0].shape) == (nout, kin)
            gw = self._grad(ws[0]) if direct else None
            direct = direct and gw.is_contiguous()
            _lib.check(lib.rdpn6d_wgrad_f32(_ptr(d_y), out_cs, 0, ca, _ptr(x), kin, 0, kin, B, 1, 1, 1, 1, 1, 1, z9, z9,
                                            _ptr(gw if direct else wg_out), _ptr(self._wg_partial), self.st()), "wgrad " + name)
            g = wg_out[:nout, 0]
            if g_view:
                g = g_view(g)
            gb = self._grad(bs[0]) if len(bs) == 1 and ca == nout else None  # one bias of ca entries: summed in place as well
            _lib.check(lib.rdpn6d_channel_sum_f32(_ptr(d_y), B, ca, out_cs, 0, _ptr(bg if gb is None else gb), 0, _ptr(self._scratch_d),
                                                  self.st()), "bias " + name)
            o = 0
            for wp, bp in zip(ws, bs):
                n = wp.shape[0]
                if not direct:
                    self._grad(wp).copy_(g[o:o + n])
                if gb is None:
                    self._grad(bp).copy_(bg[o:o + n])
                o += n

        self.bwd.append([bwd, self._launch_conv("dgrad " + name, dd, wd, ksplit=True)])

    # ------------------------------------------------------------------ one step
    LOSS_NAMES = ("loss_coor_x", "loss_coor_y", "loss_coor_z", "loss_mask", "loss_region", "loss_region_my", "loss_PM_R",
                  "loss_centroid", "loss_z")

    def _pack_sym_infos(self, sym_infos):
        """list of [K,3,3] / [3,3] / None per crop (engine_utils.py:60-61) -> ([B,Kmax,9] fp32, [B] int32, Kmax) on the
        device.  One small upload; the selection itself runs inside the pose kernel (the reference copies every predicted
        rotation to the host instead, pose_utils.py:475-481)."""
        if sym_infos is None:
            raise ValueError("PNP_NET.PM_LOSS_SYM is set: sym_infos (list of Kx3x3 or None per crop) is required")
        if len(sym_infos) != self.B:
            raise ValueError(f"sym_infos has {len(sym_infos)} entries for a batch of {self.B}")
        key = tuple(id(s) for s in sym_infos)
        if self._sym_cache is not None and self._sym_cache[0] == key:
            return self._sym_cache[2]
        mats = [None if s is None else torch.as_tensor(s, dtype=torch.float32).reshape(-1, 9).cpu() for s in sym_infos]
        kmax = max([0] + [m.shape[0] for m in mats if m is not None])
        if kmax == 0:
            packed = (None, None, 0)
        else:
            tab = torch.zeros(self.B, kmax, 9, dtype=torch.float32)
            cnt = torch.zeros(self.B, dtype=torch.int32)
            for i, m in enumerate(mats):
                if m is not None:
                    tab[i, :m.shape[0]] = m
                    cnt[i] = m.shape[0]
            packed = (tab.to(self.dev), cnt.to(self.dev), kmax)
        self._sym_cache = (key, list(sym_infos), packed)  # holds the objects so the ids stay valid
        return packed

    def forward_losses(self, batch):
        """Forward in training mode + the nine losses (also seeds the gradient buffers d_head / d_rt).
        batch: dict in the reference's batch_data contract (engine_utils.py:6-63), device tensors.
        Returns {loss name: 0-dim device tensor}."""
        lib, B, K = self.lib, self.B, self.K
        f32 = lambda t: t.detach().to(device=self.dev, dtype=torch.float32).contiguous()  # noqa: E731
        self.x.copy_(f32(batch["roi_img"]))
        self.coord2d.copy_(f32(batch["roi_coord_2d"]))
        self.fps_t.copy_(f32(batch["fps"]))
        cams, centers, whs = f32(batch["roi_cam"]), f32(batch["roi_center"]), f32(batch["roi_wh"])
        ratios, extents = f32(batch["resize_ratio"]), f32(batch["roi_extent"])
        gt_xyz, mv, mt = f32(batch["roi_xyz"]), f32(batch["roi_mask_visib"]), f32(batch["roi_mask_trunc"])
        gt_region = batch["roi_region"].to(device=self.dev, dtype=torch.int64).contiguous()
        gt_rot, gt_ratio, pts = f32(batch["ego_rot"]), f32(batch["roi_trans_ratio"]), f32(batch["roi_points"])
        for fn in self.fwd:
            fn()
        torch._foreach_add_(self._bn_counters, 1)  # BatchNorm2d.num_batches_tracked (one fused launch)
        from .gdrn import bump_weights_epoch

        bump_weights_epoch(self.model)  # running statistics moved under this model's InferencePlans' folded copies
        sym = self._pack_sym_infos(batch.get("sym_info")) if self.pm_sym else (None, None, 0)
        self._loss_ctx = (cams, centers, whs, ratios, extents, gt_xyz, mv, mt, gt_region, gt_rot, gt_ratio, pts, sym)
        self._run_loss_kernels(self.lw, self.losses9)
        self._seed_factor = 1.0  # fresh seeds
        self._consumed = False
        return {n: self.losses9[i] for i, n in enumerate(self.LOSS_NAMES)}

    def _run_loss_kernels(self, lw, losses9):
        """the two loss kernels: nine losses into `losses9` and the backward seeds d(sum_i w_i * loss_i)/d(head output) and
        /d(pose head output) into d_head / d_rt, the weights w being the six entries of `lw`"""
        lib, B, K = self.lib, self.B, self.K
        cams, centers, whs, ratios, extents, gt_xyz, mv, mt, gt_region, gt_rot, gt_ratio, pts, (sym_rots, sym_counts, ksym) = self._loss_ctx
        HW = (self.R // 4) ** 2
        sc = self.buf("pose_scratch", 3 * B)
        _lib.check(lib.rdpn6d_pose_train_sym_f32(_ptr(self.rt), 16, _ptr(cams), _ptr(centers), _ptr(whs), _ptr(ratios),
                                                 _ptr(extents), _ptr(gt_rot), _ptr(gt_ratio), _ptr(pts), pts.shape[1], B,
                                                 self.is_allo, lw["pm"], lw["pm_norm"], lw["centroid"], lw["z"],
                                                 _ptr(sym_rots) if ksym else None, _ptr(sym_counts) if ksym else None, ksym,
                                                 _ptr(self.gt_rot_used), _ptr(self.rot), _ptr(self.trans), _ptr(self.d_rt),
                                                 _ptr(losses9[6:]), _ptr(sc), self.st()), "pose_train")
        _lib.check(lib.rdpn6d_dense_losses_mt_f32(_ptr(self.head_out), self.head_cs, _ptr(gt_xyz), _ptr(mv), _ptr(mt), _ptr(gt_region), B, HW,
                                                  K, lw["xyz"], lw["mask"], lw["region"], self.mask_type, _ptr(self.d_head), _ptr(losses9),
                                                  _ptr(self._scratch_d), self.st()), "dense_losses")

    def seed_backward(self, weights):
        """d(total)/d(loss_i) for the next backward() - what autograd hands to the node the nine losses hang off
        (gdrn._HipBackward): all 1 for the reference's plain sum, the loss scale under a GradScaler (engine.py:302-309),
        1/n for gradient accumulation, 0 for a loss left out.  The backward is linear in its two seeds, so a common factor
        scales them; different factors re-run the two loss kernels with the loss weights multiplied (losses that share a
        weight in the kernels - coor_x/y/z, region/region_my - must then share their factor).
        ABSOLUTE, not cumulative: the factor applied to d_head / d_rt since the last forward_losses() is remembered, so calling
        this again (a second backward through the same node with retain_graph, a caller that seeds and then runs
        forward_backward) re-scales by new / old instead of compounding."""
        if self._consumed:
            raise RuntimeError("seed_backward: no forward to differentiate (backward already ran for it; retain_graph is not supported)")
        w = [float(weights.get(n, 0.0)) for n in self.LOSS_NAMES]
        if any(x != x or x in (float("inf"), float("-inf")) for x in w):
            raise FloatingPointError(f"non-finite gradient flowing into the losses: {dict(zip(self.LOSS_NAMES, w))}")
        cur = self._seed_factor  # a float (common factor in place) or None (per-loss factors / zeroed: seeds must be recomputed)
        if all(x == w[0] for x in w) and cur is not None and cur != 0.0:
            if w[0] != cur:
                self.d_head.mul_(w[0] / cur)
                self.d_rt.mul_(w[0] / cur)
                self._seed_factor = w[0]
            return
        if not (w[0] == w[1] == w[2] and w[4] == w[5]):
            raise NotImplementedError("different upstream gradients for loss_coor_x/y/z or for loss_region/loss_region_my: the loss "
                                      f"kernels weight them as one group ({dict(zip(self.LOSS_NAMES, w))})")
        lw = dict(self.lw)
        for key, i in (("xyz", 0), ("mask", 3), ("region", 4), ("pm", 6), ("centroid", 7), ("z", 8)):
            lw[key] = self.lw[key] * w[i]
        self._run_loss_kernels(lw, self.buf("losses9_scratch", 9))
        self._seed_factor = w[0] if all(x == w[0] for x in w) else None

    def _flat_or_list(self):
        """(the one flat buffer all parameter gradients live in - GradBuckets / Ranger layout - or None, the gradient list).
        Recognised by STORAGE, not by ``_base``: a gradient that went through autograd's AccumulateGrad (gdrn._HipBackward hands the
        views back to it so that DDP's hooks fire) is a detached alias of the same memory, no longer a view of the flat tensor."""
        from .parallel import flat_grad_storage

        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        return flat_grad_storage(grads), grads

    def backward_stages(self, unscale=1.0):
        """The backward of sum(seed_i * loss_i) into param.grad as a GENERATOR: yields the names of parallel.STAGES ('pnp_net',
        'rot_head_net', 'backbone.layer4', 'backbone.layer3', 'backbone.rest') - each right after the launches that complete that
        stage's parameter gradients (the order the backward finishes them).  backward()
        drives it for the bucket hooks of parallel.GradBuckets; gdrn._HipBackward drives it from five chained autograd nodes, so
        that under torch DDP the AccumulateGrad hooks of a group - and with them DDP's bucket all-reduces - fire while the rest
        of the backward is still being issued.  Must be run to exhaustion."""
        if self._consumed:
            # the backward kernels work in place on the gradient buffers the loss kernels seeded (and on saved activations): a second
            # pass over the same forward would start from overwritten data - autograd's "backward through the graph a second time"
            raise RuntimeError("TrainEngine.backward was already run for this forward; the HIP backward consumes the forward's buffers "
                               "(retain_graph is not supported) - call forward_losses / the model's forward again")
        self._consumed = True
        marks = getattr(self, "_group_marks", None) or {}
        prev = None
        if self.accumulate_grad:
            flat, _ = self._flat_or_list()
            if flat is not None:  # (+ where each parameter's gradient sits in it, should the gradients have left the buffer afterwards)
                prev = (flat.clone(), [(p, p.grad.storage_offset(), p.grad.numel()) for p in self.model.parameters() if p.grad is not None])
            else:
                prev = {id(p): p.grad.clone() for p in self.model.parameters() if p.grad is not None}
        # un-scale / add-back run exactly ONCE, right after the LAST launch (list index 0) and before the group completed by it is
        # yielded - whatever the marks are (a build that marks no group at index 0 must neither skip nor repeat it)
        for idx in range(len(self.bwd) - 1, -1, -1):
            for fn in self.bwd[idx]:
                fn()
            if idx == 0 or idx in marks:
                self._join_side()  # the weight gradients issued on the side stream belong to the groups completed here
            if idx == 0:
                self._finish_backward(unscale, prev)
            if idx in marks:
                yield marks[idx]

    def _finish_backward(self, unscale, prev):
        if unscale != 1.0 or prev is not None:
            flat, grads = self._flat_or_list()
            if unscale != 1.0:
                flat.mul_(1.0 / unscale) if flat is not None else torch._foreach_mul_(grads, 1.0 / unscale)
            if prev is not None:
                if isinstance(prev, tuple):  # (the flat buffer of GradBuckets / Ranger)
                    if flat is not None and flat.numel() == prev[0].numel():  # same layout before and after
                        flat.add_(prev[0])
                    else:  # a gradient was cloned out of the buffer in between (AccumulateGrad does when it cannot adopt): slice-wise
                        ps = [(p, o, n) for p, o, n in prev[1] if p.grad is not None]
                        if ps:
                            torch._foreach_add_([p.grad for p, _, _ in ps], [prev[0][o:o + n].view_as(p.grad) for p, o, n in ps])
                else:  # (a gradient this backward created had nothing accumulated before)
                    ps = [p for p in self.model.parameters() if p.grad is not None and id(p) in prev]
                    if ps:
                        torch._foreach_add_([p.grad for p in ps], [prev[id(p)] for p in ps])

    def backward(self, on_group_done=None, unscale=1.0):
        """Backward of sum(seed_i * loss_i) into param.grad.  on_group_done(name) is called after the gradients of each stage of
        parallel.STAGES are complete (gradient-bucket all-reduce hook: parallel.GradBuckets.reduce).
        The kernels WRITE param.grad (split-K reduces, BatchNorm / bias sums store, they do not add): that equals autograd's
        accumulate after the zero_grad the reference loop issues before every backward (engine.py:304-308).  For gradient
        accumulation over micro-batches - not part of the reference loop - set `accumulate_grad = True`: the gradients already
        in param.grad are saved and added back after this backward (one extra pass over the 146 MB; not with bucket hooks, whose
        all-reduce would average the running sum again).
        unscale: divide THIS backward's gradients by it (the loss scale the seeds carried) - before anything accumulated
        earlier is added back, so earlier micro-batches are never divided twice."""
        if on_group_done is not None:  # (checked before anything runs: a refused call leaves the forward differentiable)
            if self.accumulate_grad:
                raise NotImplementedError("accumulate_grad with gradient-bucket hooks: reduce once, after the last micro-batch")
            if unscale != 1.0:
                raise NotImplementedError("unscale with gradient-bucket hooks: un-scale the flat buffer after buckets.finish()")
        for group in self.backward_stages(unscale=unscale):
            if on_group_done is not None:
                on_group_done(group)

    def forward_backward(self, batch):
        """forward + losses + backward.  With `loss_scale` != 1 (fp16 storage of the activation gradients: the un-scaled seeds
        1 / (B * HW) ~ 4e-6 are fp16 subnormals) the seeds are multiplied by it and the parameter gradients of THIS backward divided
        again - what torch's GradScaler does around the reference's step (engine.py:302-309); non-finite gradients are left for
        the caller's overflow check, as GradScaler.step would see them."""
        losses = self.forward_losses(batch)
        S = float(self.loss_scale)
        self.seed_backward({n: S for n in self.LOSS_NAMES})
        self.backward(unscale=S)
        return losses
