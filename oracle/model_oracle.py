"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``rdpn6d_amd``) never routes through it.

Plain PyTorch-CPU fp32 restatement of the reference's RGB-D forward / loss path, with the same
``state_dict`` key names so that one seeded weight set drives the reference, this oracle and the
HIP path.  Each piece cites the reference code it follows (paths relative to /root/reference):

  backbone   core/gdrn_modeling/models/resnet_backbone.py:264-340 (+ md_pointnet :23-54,
             torchvision 0.17.1 BasicBlock: conv3x3(s)-BN-ReLU-conv3x3-BN (+downsample) add ReLU)
  head       core/gdrn_modeling/models/cdpn_rot_head_region.py:83-146 (layers), :185-198 (split)
  glue       core/gdrn_modeling/models/GDRN.py:196-233, models/model_utils.py:24-42
  ConvPnPNet core/gdrn_modeling/models/conv_pnp_net.py:41-163
  rot6d      core/utils/rot_reps.py:9-49
  pose       core/gdrn_modeling/models/pose_from_pred_centroid_z.py:52-141 (test), :144-227 (train)
             core/utils/utils.py:39-94 (numpy allo->ego), :208-236 (torch allo->ego)
             transforms3d 0.4.2 axangles.axangle2mat (restated: third-party, not in /root/reference)
  losses     core/gdrn_modeling/models/GDRN.py:373-633, losses/pm_loss.py:82-173

Parity is PINNED: tools/oracle/gen_model_golden.py imports the real reference (with stub third-party
packages) in the build container, runs it on the seeded inputs/weights of rdpn6d_amd/synth.py and
commits its outputs to tests/golden/model_c1.npz; tests/test_model_oracle.py checks this file
against those vectors.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------- trunk
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, cout, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = downsample

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        idt = x if self.downsample is None else self.downsample(x)
        return self.relu(y + idt)


class Bottleneck(nn.Module):
    """torchvision 0.17.1 Bottleneck (ResNet v1.5: the stride sits on the 3x3): 1x1-BN-ReLU, 3x3(s)-BN-ReLU, 1x1(x4)-BN,
    (+downsample) add ReLU.  resnet_backbone.py:15-21 selects it for 50 / 101 / 152 layers; the reference itself cannot
    RUN those (md_pointnet(512, ...) is hard-coded at :270 while layer4 then has 2048 channels), so for these trunks the
    oracle is the build's own generalisation (pointnet in = channels of layer4) and parity with the reference is unpinned."""
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        idt = x if self.downsample is None else self.downsample(x)
        return self.relu(y + idt)


class PointFusion(nn.Module):
    """md_pointnet (resnet_backbone.py:23-54): point-wise MLP over (trunk feature, depth xyz)."""

    def __init__(self, cin=512, ch=(64, 128, 256, 512)):
        super().__init__()
        self.xyz_emb = nn.Conv2d(cin, ch[0], 1)
        self.xb = nn.BatchNorm2d(ch[0])
        self.conv1 = nn.Conv2d(ch[0] + 3, ch[1], 1)
        self.conv2 = nn.Conv2d(ch[1], ch[2], 1)
        self.conv3 = nn.Conv2d(ch[2], ch[3], 1)
        self.b1 = nn.BatchNorm2d(ch[1])
        self.b2 = nn.BatchNorm2d(ch[2])
        self.b3 = nn.BatchNorm2d(ch[3])
        self.relu = nn.ReLU()  # (a module, not F.relu, so that forced_relu_masks() below can reach the three call sites)

    def forward(self, feat, xyz):
        emb = self.relu(self.xb(self.xyz_emb(feat)))
        l1 = self.relu(self.b1(self.conv1(torch.cat([xyz, emb], 1))))
        l2 = self.relu(self.b2(self.conv2(l1)))
        l3 = self.b3(self.conv3(l2))  # BN, no ReLU (:49)
        g = l3.amax(dim=(2, 3), keepdim=True).expand_as(l3)  # global max, broadcast (:51-52)
        return torch.cat([l3, g], 1)


class Backbone(nn.Module):
    LAYERS = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
              101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}  # resnet_backbone.py:15-21

    def __init__(self, num_layers=34):
        super().__init__()
        block, counts = self.LAYERS[num_layers]
        self.spatial_net = PointFusion(512 * block.expansion, (64, 128, 256, 512))
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), counts)):
            stride = 1 if li == 0 else 2
            blocks = []
            for bi in range(nblk):
                ds = None
                cout = planes * block.expansion
                if bi == 0 and (stride != 1 or cin != cout):
                    ds = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
                blocks.append(block(cin, planes, stride if bi == 0 else 1, ds))
                cin = cout
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))

    def forward(self, x):
        xyz = x[:, 3:]
        rgb = x[:, :3]
        r8 = x.shape[-1] // 8
        xyz = F.interpolate(xyz, (r8, r8), mode="nearest")  # == xyz[:, :, ::8, ::8]
        y = self.maxpool(self.relu(self.bn1(self.conv1(rgb))))
        y = self.layer4(self.layer3(self.layer2(self.layer1(y))))
        y = F.interpolate(y, scale_factor=4, mode="bilinear", align_corners=True)  # UpsamplingBilinear2d
        return self.spatial_net(y, xyz)


# --------------------------------------------------------------------------------------- heads
class RotHead(nn.Module):
    def __init__(self, num_regions=32, num_filters=256, num_layers=3, in_channels=1024, mask_out_dim=1):
        """mask_out_dim: 1 (MASK_LOSS_TYPE L1 | BCE) or 2 (CE) - get_xyz_mask_region_out_dim, GDRN.py:637-659"""
        super().__init__()
        self.mask_out_dim = mask_out_dim
        f = [nn.ConvTranspose2d(in_channels, num_filters, 3, 2, 1, output_padding=1, bias=False),
             nn.BatchNorm2d(num_filters), nn.ReLU(inplace=True)]
        for _ in range(2 * num_layers):
            f += [nn.Conv2d(num_filters, num_filters, 3, 1, 1, bias=False), nn.BatchNorm2d(num_filters),
                  nn.ReLU(inplace=True)]
        f.append(nn.Conv2d(num_filters, mask_out_dim + 3 + num_regions + 1, 1, bias=True))
        self.features = nn.ModuleList(f)

    def forward(self, x):
        for l in self.features:
            x = l(x)
        m = self.mask_out_dim  # cdpn_rot_head_region.py:190-197
        return x[:, :m], x[:, m:m + 1], x[:, m + 1:m + 2], x[:, m + 2:m + 3], x[:, m + 3:]  # mask, coor_x, coor_y, coor_z, region


class ConvPnP(nn.Module):
    def __init__(self, n_in=43, featdim=128, rot_dim=6, out_res=64):
        super().__init__()
        f = []
        for i in range(3):
            f += [nn.Conv2d(n_in if i == 0 else featdim, featdim, 3, 2, 1, bias=False), nn.GroupNorm(32, featdim),
                  nn.ReLU(inplace=True)]
        self.features = nn.ModuleList(f)
        self.fc1 = nn.Linear(featdim * (out_res // 8) ** 2, 1024)
        self.fc2 = nn.Linear(1024, 256)
        self.fc_r = nn.Linear(256, rot_dim)
        self.fc_t = nn.Linear(256, 3)
        self.act = nn.LeakyReLU(0.1, inplace=True)

    def forward(self, coor_feat, region, mask_attention=None):
        x = torch.cat([coor_feat, region], 1)
        if mask_attention is not None:
            x = x * mask_attention
        for l in self.features:
            x = l(x)
        x = x.flatten(1)
        x = self.act(self.fc2(self.act(self.fc1(x))))
        return self.fc_r(x), self.fc_t(x)


# --------------------------------------------------------------------------------------- pose
def rot6d_to_mat(p):
    x = F.normalize(p[:, 0:3], p=2, dim=1)
    z = F.normalize(torch.cross(x, p[:, 3:6], dim=1), p=2, dim=1)
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


def axangle2mat(axis, angle):
    """transforms3d.axangles.axangle2mat restated (Rodrigues, normalises the axis)."""
    x, y, z = axis
    n = math.sqrt(x * x + y * y + z * z)
    x, y, z = x / n, y / n, z / n
    c, s = math.cos(angle), math.sin(angle)
    C = 1 - c
    xs, ys, zs = x * s, y * s, z * s
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    return np.array([[x * xC + c, xyC - zs, zxC + ys], [xyC + zs, y * yC + c, yzC - xs], [zxC - ys, yzC + xs, z * zC + c]])


def allo_to_ego_numpy(rot_allo, trans):
    """core/utils/utils.py:39-94 for src=dst='mat': fp32 inputs, fp64 axis-angle, fp32 result."""
    rot_allo = np.asarray(rot_allo, dtype=np.float32)
    trans = np.asarray(trans, dtype=np.float32)
    cam_ray = np.asarray((0, 0, 1.0))
    obj_ray = trans.copy() / np.linalg.norm(trans)
    angle = math.acos(cam_ray.dot(obj_ray))
    if angle > 0:
        rot = axangle2mat(np.cross(cam_ray, obj_ray), angle)
        return np.dot(rot, rot_allo).astype(np.float32)
    return rot_allo.copy()


def site_translation(pred_t, roi_cams, roi_centers, roi_whs, resize_ratios):
    cx = pred_t[:, 0:1] * roi_whs[:, 0:1] + roi_centers[:, 0:1]
    cy = pred_t[:, 1:2] * roi_whs[:, 1:2] + roi_centers[:, 1:2]
    z = pred_t[:, 2:3] * resize_ratios.view(-1, 1)
    return torch.cat(
        [z * (cx - roi_cams[:, 0:1, 2]) / roi_cams[:, 0:1, 0], z * (cy - roi_cams[:, 1:2, 2]) / roi_cams[:, 1:2, 1], z], 1
    )


def quat2mat(q):
    """core/utils/pose_utils.py:323-355 with its default eps=0 (as called from utils.py:232)."""
    qn = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    B = q.shape[0]
    X, Y, Z = x * 2.0, y * 2.0, z * 2.0
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return torch.stack(
        [1.0 - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.0 - (xX + zZ), yZ - wX, xZ - wY, yZ + wX, 1.0 - (xX + yY)], dim=1
    ).reshape(B, 3, 3)


def allo_to_ego_torch(trans, rot_allo, eps=1e-4):
    cam_ray = torch.tensor([0, 0, 1.0], dtype=trans.dtype)
    obj_ray = trans / (trans.norm(dim=1, keepdim=True) + eps)
    angle = obj_ray[:, 2:3].acos()
    axis = torch.cross(cam_ray.expand_as(obj_ray), obj_ray, dim=1)
    axis = axis / (axis.norm(dim=1, keepdim=True) + eps)
    q = torch.cat([torch.cos(angle / 2), axis * torch.sin(angle / 2)], 1)
    return torch.matmul(quat2mat(q), rot_allo)


# --------------------------------------------------------------------------------------- model
class GDRNOracle(nn.Module):
    def __init__(self, num_regions=32, mask_attention="none", out_res=64, num_layers=34, mask_loss_type="L1"):
        super().__init__()
        assert mask_loss_type in ("L1", "BCE", "CE"), mask_loss_type
        self.backbone = Backbone(num_layers)
        self.rot_head_net = RotHead(num_regions, mask_out_dim=2 if mask_loss_type == "CE" else 1)
        self.pnp_net = ConvPnP(11 + num_regions, out_res=out_res)
        self.mask_attention = mask_attention
        self.mask_loss_type = mask_loss_type  # cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE
        self.out_res = out_res

    def dense(self, x):
        return self.rot_head_net(self.backbone(x))

    def glue(self, mask, cx, cy, cz, region, roi_coord_2d, fps, force_argmax=None):
        """force_argmax (B, r, r) int: take the region decision from the implementation under test instead of the arg-max of the
        oracle's own softmax (tests only: compares the pose branch GIVEN the same decisions at near-tie pixels)"""
        B = mask.shape[0]
        r = self.out_res
        coor_feat = torch.cat([cx, cy, cz, roi_coord_2d], 1)
        prob = F.softmax(region[:, 1:], dim=1)
        amax = prob.reshape(B, prob.shape[1], -1).argmax(dim=1)  # (B, HW), arg-max ON the softmax output
        if force_argmax is not None:
            amax = torch.as_tensor(force_argmax).reshape(B, -1).to(torch.int64)
        anchors = torch.gather(fps, 1, amax.unsqueeze(2).expand(-1, -1, 3))  # (B,HW,3)
        anchors = anchors.reshape(B, r, r, 3).permute(0, 3, 1, 2)
        coor_feat = torch.cat([coor_feat, anchors], 1)
        att = None
        if self.mask_attention != "none":  # get_mask_prob, models/model_utils.py:24-42
            if self.mask_loss_type == "L1":
                mx = mask.reshape(B, -1).max(dim=1)[0].view(B, 1, 1, 1)
                mn = mask.reshape(B, -1).min(dim=1)[0].view(B, 1, 1, 1)
                att = (mask - mn) / (mx - mn)
            elif self.mask_loss_type == "BCE":
                att = torch.sigmoid(mask)
            else:  # :39 - torch.softmax(pred_mask, dim=1, keepdim=True): softmax has no keepdim, the reference raises here
                raise TypeError("get_mask_prob's CE branch raises in the reference (softmax() got an unexpected keyword 'keepdim')")
        return coor_feat, prob, att, amax.reshape(B, r, r)

    def forward(self, x, roi_coord_2d, fps, roi_cams, roi_centers, roi_whs, resize_ratios, train_pose=False, dense_maps=None,
                force_argmax=None):
        """dense_maps = (mask, coor_x, coor_y, coor_z, region): start from given head outputs (teacher forcing: the reference's own
        golden maps) instead of running trunk + head on x; force_argmax: see glue()"""
        mask, cx, cy, cz, region = self.dense(x) if dense_maps is None else dense_maps
        coor_feat, prob, att, amax = self.glue(mask, cx, cy, cz, region, roi_coord_2d, fps, force_argmax=force_argmax)
        rot6d, pred_t = self.pnp_net(coor_feat, prob, att)
        rot_allo = rot6d_to_mat(rot6d)
        trans = site_translation(pred_t, roi_cams, roi_centers, roi_whs, resize_ratios)
        if train_pose:
            rot = allo_to_ego_torch(trans, rot_allo, eps=1e-4)
        else:
            up = (lambda v: v.float() if v.dtype == torch.bfloat16 else v)  # numpy has no bf16 (autocast runs)
            ra, tn = up(rot_allo.detach()).numpy(), up(trans.detach()).numpy()
            rot = torch.from_numpy(np.stack([allo_to_ego_numpy(ra[i], tn[i]) for i in range(ra.shape[0])]))
        return {"rot": rot, "trans": trans, "mask": mask, "coor_x": cx, "coor_y": cy, "coor_z": cz, "region": region,
                "pred_rot6d": rot6d, "pred_t_": pred_t, "region_argmax": amax}


# --------------------------------------------------------------------------------------- forced ReLU decisions
class forced_relu_masks:
    """Context manager for gradient parity tests: every ReLU / LeakyReLU call site of `model` takes its on/off decision from
    `masks[(module name, call index)]` (a 0/1 tensor of the activation's shape) instead of the sign of its own input:
    out = x * mask (LeakyReLU: x * (mask + slope * (1 - mask))).

    Why: the fp32 backward of a ReLU network is reproducible only up to the units whose pre-activation sits within round-off
    of zero - one flipped unit of a 1024-wide layer moves every upstream gradient by ~2 % (measured on the real reference,
    tests/golden/README.md).  With the decisions of the implementation under test forced into the oracle, what remains is
    the arithmetic of the backward itself, which can then be compared at 1e-4.  The forward values change by at most the
    magnitude of the near-zero pre-activations concerned (1e-6)."""

    def __init__(self, model, masks, round_dtype=None):
        """round_dtype (torch.bfloat16 | torch.float16): additionally round the OUTPUT of every trunk / fusion / head ReLU - and the
        gradient arriving at it - to that format: the mixed-precision step stores these activations (and their gradients) in 16
        bits (lowp_storage below emulates the rest of that step)."""
        self.model, self.masks, self.saved, self.used, self.round_dtype = model, masks, [], set(), round_dtype
        self.outputs = {}  # (module name, call index) -> the activation this call produced (stage-by-stage debugging)

    def __enter__(self):
        for name, m in self.model.named_modules():
            if isinstance(m, (nn.ReLU, nn.LeakyReLU)):
                slope = m.negative_slope if isinstance(m, nn.LeakyReLU) else 0.0
                calls = [0]

                def fwd(x, name=name, slope=slope, calls=calls):
                    key = (name, calls[0])
                    calls[0] += 1
                    mask = self.masks[key].to(x.dtype)
                    assert mask.shape == x.shape, (key, tuple(mask.shape), tuple(x.shape))
                    self.used.add(key)
                    y = x * (mask + slope * (1 - mask)) if slope else x * mask
                    if self.round_dtype is not None and name.startswith(("backbone", "rot_head_net")):
                        y = _RoundSTE.apply(y, self.round_dtype)
                    self.outputs[key] = y.detach()
                    return y

                self.saved.append((m, m.__dict__.get("forward")))
                m.forward = fwd
        return self

    def __exit__(self, *exc):
        for m, old in self.saved:
            if old is None:
                del m.__dict__["forward"]
            else:
                m.forward = old
        return False


# --------------------------------------------------------------------------------------- 16-bit storage emulation
class _RoundSTE(torch.autograd.Function):
    """value AND gradient rounded to a 16-bit format (round-to-nearest-even), carried in the wider dtype"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _RoundValue(torch.autograd.Function):
    """value rounded to a 16-bit format, gradient passed through unchanged (weights: their 16-bit mirror is what the matrix pipe
    reads, their gradient is accumulated and kept in fp32)"""

    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundGrad(torch.autograd.Function):
    """identity forward; the gradient is rounded to a 16-bit format"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class lowp_storage:
    """Context manager for the parity test of the MIXED-PRECISION training step (cfg.SOLVER.AMP: the reference's autocast +
    GradScaler switch, engine.py:279-309): the fp32 oracle evaluated on 16-BIT-ROUNDED OPERANDS with fp32 accumulation, rounding
    wherever the HIP step stores 16 bits (rdpn6d_amd/train.py: buffers raw:* act:* d:* dres:*):

      * every convolution of trunk / fusion branch / dense head reads a 16-bit input and 16-bit weights, accumulates in fp32 and
        stores a 16-bit output (conv1 of the stem: fp32-accurate arithmetic on the fp32 crop, 16-bit output; the head's 1x1
        output convolution: 16-bit operands, fp32 output whose gradient is rounded for its dgrad / wgrad);
      * BatchNorm arithmetic is fp32 on the stored values; what it hands on is stored in 16 bits: after the ReLU
        (forced_relu_masks(round_dtype=...)), after the two ReLU-less norms (downsample.1, spatial_net.b3);
      * the same rounding is applied to the gradients flowing back through those points (activation gradients are stored in
        16 bits); ConvPnPNet, the losses, parameter gradients and the pose branch stay fp32.
    Combine with forced_relu_masks(model, masks, round_dtype=dtype) and GDRNOracle.forward(force_argmax=...)."""

    def __init__(self, model, dtype):
        self.model, self.dtype, self.saved = model, dtype, []

    def __enter__(self):
        dt = self.dtype
        last = f"rot_head_net.features.{len(self.model.rot_head_net.features) - 1}"
        for name, m in self.model.named_modules():
            if not name.startswith(("backbone", "rot_head_net")):
                continue
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                def fwd(x, m=m, name=name):
                    first = name == "backbone.conv1"
                    xin = x if first else _RoundSTE.apply(x, dt)
                    w = m.weight if first else _RoundValue.apply(m.weight, dt)
                    if isinstance(m, nn.ConvTranspose2d):
                        y = F.conv_transpose2d(xin, w, m.bias, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
                    else:
                        y = F.conv2d(xin, w, m.bias, m.stride, m.padding, m.dilation, m.groups)
                    return _RoundGrad.apply(y, dt) if name == last else _RoundSTE.apply(y, dt)
            elif isinstance(m, nn.BatchNorm2d) and name.endswith(("downsample.1", "spatial_net.b3")):
                def fwd(x, m=m, orig=type(m).forward):
                    return _RoundSTE.apply(orig(m, x), dt)
            else:
                continue
            self.saved.append((m, m.__dict__.get("forward")))
            m.forward = fwd
        return self

    def __exit__(self, *exc):
        for m, old in self.saved:
            if old is None:
                del m.__dict__["forward"]
            else:
                m.forward = old
        return False


# --------------------------------------------------------------------------------------- losses
def rot_error_deg(r_est, r_gt):
    """Rotation error in degrees (lib/pysixd/pose_error.py:400-415): acos of the clamped (trace(R_est R_gt^T) - 1) / 2."""
    tr = float(np.trace(np.asarray(r_est) @ np.asarray(r_gt).T))
    tr = min(tr, 3.0)
    return float(np.rad2deg(np.arccos(min(1.0, max(-1.0, 0.5 * (tr - 1.0))))))


VIS_NAMES = ("vis/error_R", "vis/error_t", "vis/error_tx", "vis/error_ty", "vis/error_tz", "vis/tx_pred", "vis/ty_pred", "vis/tz_pred",
             "vis/tx_net", "vis/ty_net", "vis/tz_net", "vis/tx_gt", "vis/ty_gt", "vis/tz_gt", "vis/tx_rel_gt", "vis/ty_rel_gt", "vis/tz_rel_gt")


def train_vis_scalars(pred_trans, pred_rot, pred_t_, gt_trans, gt_rot, gt_trans_ratio):
    """The 17 scalars the reference's train forward pushes to EventStorage (core/gdrn_modeling/models/GDRN.py:306-328), numpy:
    compute_mean_re_te (models/model_utils.py:45-57): float32 arrays of re (lib/pysixd/pose_error.py:400-415, degrees) and te
    (:428-440) per crop, their float32 means; the rest are reads of crop 0.  Pinned by tests/golden/vis_scalars_golden.npz."""
    f = lambda a: np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float32)  # noqa: E731
    pred_trans, pred_rot, pred_t_, gt_trans, gt_rot, gt_trans_ratio = (f(a) for a in (pred_trans, pred_rot, pred_t_, gt_trans, gt_rot, gt_trans_ratio))
    bs = pred_rot.shape[0]
    r_errs, t_errs = np.zeros((bs,), dtype=np.float32), np.zeros((bs,), dtype=np.float32)
    for i in range(bs):
        trace = np.trace(np.dot(pred_rot[i], gt_rot[i].T))
        trace = trace if trace <= 3 else 3
        r_errs[i] = np.rad2deg(np.arccos(min(1.0, max(-1.0, 0.5 * (trace - 1.0)))))
        t_errs[i] = np.linalg.norm(gt_trans[i].flatten() - pred_trans[i].flatten())
    v = {"vis/error_R": r_errs.mean(), "vis/error_t": t_errs.mean() * 100}
    for a, ax in enumerate("xyz"):
        v[f"vis/error_t{ax}"] = np.abs(float(pred_trans[0, a]) - float(gt_trans[0, a])) * 100
        v[f"vis/t{ax}_pred"], v[f"vis/t{ax}_net"] = float(pred_trans[0, a]), float(pred_t_[0, a])
        v[f"vis/t{ax}_gt"], v[f"vis/t{ax}_rel_gt"] = float(gt_trans[0, a]), float(gt_trans_ratio[0, a])
    return {k: float(v[k]) for k in VIS_NAMES}


def closest_sym_rots(pred_rots, gt_rots, sym_infos):
    """PM_LOSS_SYM target choice (core/utils/pose_utils.py:430-482): per sample, among R_gt and R_gt @ S_k (S_k the
    model-to-model symmetry rotations, None / empty = not symmetric) the one with the smallest rotation error to the
    detached prediction; a later candidate replaces the incumbent only when strictly better."""
    pr = pred_rots.detach().cpu().numpy()
    out = gt_rots.detach().cpu().numpy().copy()
    for i, sym in enumerate(sym_infos):
        if sym is None:
            continue
        sym = np.asarray(sym.cpu().numpy() if isinstance(sym, torch.Tensor) else sym).reshape(-1, 3, 3)
        g0 = out[i].copy()
        best = rot_error_deg(pr[i], g0)
        for k in range(sym.shape[0]):
            cand = g0 @ sym[k]
            e = rot_error_deg(pr[i], cand)
            if e < best:
                best, out[i] = e, cand
    return torch.as_tensor(out, dtype=gt_rots.dtype, device=gt_rots.device)


def gdrn_losses(out, gt, roi_extents, sym_infos=None, mask_loss_type="L1"):
    """Active losses of the shipped configs (GDRN.py:411-424,452-454,470-483,529-531,552-554;
    pm_loss.py:97-114 with PM_R_ONLY, PM_NORM_BY_EXTENT, L1; sym_infos != None = PM_LOSS_SYM).
    mask_loss_type = ROT_HEAD.MASK_LOSS_TYPE: the three branches of GDRN.py:450-463."""
    mv = gt["roi_mask_visib"]
    denom = mv.sum().float().clamp(min=1.0)
    L = {}
    for i, k in enumerate(("coor_x", "coor_y", "coor_z")):
        L[f"loss_{k}"] = F.l1_loss(out[k] * mv[:, None], gt["roi_xyz"][:, i:i + 1] * mv[:, None], reduction="sum") / denom
    if mask_loss_type == "L1":
        L["loss_mask"] = F.l1_loss(out["mask"][:, 0], gt["roi_mask_trunc"], reduction="mean")
    elif mask_loss_type == "BCE":  # nn.BCEWithLogitsLoss(reduction="mean")
        L["loss_mask"] = F.binary_cross_entropy_with_logits(out["mask"][:, 0], gt["roi_mask_trunc"], reduction="mean")
    elif mask_loss_type == "CE":   # nn.CrossEntropyLoss(reduction="mean")(out_mask, gt_mask.long())
        L["loss_mask"] = F.cross_entropy(out["mask"], gt["roi_mask_trunc"].long(), reduction="mean")
    else:
        raise NotImplementedError(f"unknown mask loss type: {mask_loss_type}")
    L["loss_region"] = F.cross_entropy(out["region"] * mv[:, None], gt["roi_region"].long() * mv.long(),
                                       reduction="sum") / denom
    L["loss_region_my"] = F.l1_loss(mv, out["region"][:, 0], reduction="mean")
    pts = gt["roi_points"]
    w = 1.0 / roi_extents.max(1, keepdim=True)[0]
    pe = torch.bmm(pts, out["rot"].transpose(1, 2))
    gt_rot = gt["ego_rot"] if sym_infos is None else closest_sym_rots(out["rot"], gt["ego_rot"], sym_infos)
    pg = torch.bmm(pts, gt_rot.transpose(1, 2))
    L["loss_PM_R"] = 3 * F.l1_loss(w[:, :, None] * pe, w[:, :, None] * pg, reduction="mean")
    L["loss_centroid"] = F.l1_loss(out["pred_t_"][:, :2], gt["roi_trans_ratio"][:, :2], reduction="mean")
    L["loss_z"] = F.l1_loss(out["pred_t_"][:, 2], gt["roi_trans_ratio"][:, 2], reduction="mean")
    return L


def calibrate_bn(model, x):
    """One train-mode pass of backbone+head with momentum=None so running stats == batch stats."""
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
    model.train()
    with torch.no_grad():
        model.dense(x)
    model.eval()
    for m in bns:
        m.momentum = 0.1
