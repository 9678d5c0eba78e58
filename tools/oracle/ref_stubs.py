"""BUILD-CONTAINER ONLY.  Makes the real reference (/root/reference) importable without its
third-party stack (mmcv, detectron2, torchvision, cv2, transforms3d, ... are not installed here).

Recipe from SURVEY.md §8c.  Real mini-modules are registered for the handful of third-party
symbols whose arithmetic the hot path actually uses (restated from their pinned versions:
torchvision 0.17.1 BasicBlock, mmcv 1.7.2 normal_init/constant_init, transforms3d 0.4.2
axangle2mat); every other missing import resolves to a MagicMock package.

Nothing here is shipped to, or used on, the GPU box.
"""
import importlib.abc
import importlib.machinery
import math
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = "/root/reference"


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    # consulted LAST (appended to sys.meta_path): anything the real finders cannot resolve and that is
    # not one of the reference's / this repo's own packages becomes a MagicMock package.
    OWN = ("core", "lib", "ref", "configs", "tools", "rdpn6d_amd", "oracle", "tests")

    STUBBED = {"mmcv", "detectron2", "torchvision", "transforms3d"}  # partially real stub packages
    mocked_tops = set()

    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in self.OWN or fullname in sys.modules:
            return None
        if "." not in fullname:
            if not self._asked_by_reference():
                return None  # optional import inside a real third-party package (e.g. scipy -> uarray)
            self.mocked_tops.add(top)  # the real finders already failed for this top-level name
        elif top not in self.mocked_tops and top not in self.STUBBED:
            return None  # missing optional submodule of a REAL package (e.g. scipy._lib._uarray): stay missing
        return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    @staticmethod
    def _asked_by_reference():
        f = sys._getframe(2)
        while f is not None:
            fn = f.f_code.co_filename
            if "importlib" not in fn and not fn.startswith("<frozen"):
                return fn.startswith(REF_ROOT)
            f = f.f_back
        return False

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


def install():
    import numpy as np
    import numpy.ma  # noqa: F401  (must be imported before the alias shims below)
    import scipy.linalg  # noqa: F401
    import scipy.spatial  # noqa: F401
    import torch
    import torch.nn as nn

    # the reference targets numpy 1.23
    np.float = float
    np.bool = bool
    np.int = int
    np.maximum_sctype = lambda t: np.float64

    # ---- mmcv.cnn init helpers (mmcv 1.7.2 semantics) ----
    mmcv = _pkg("mmcv")
    cnn = _pkg("mmcv.cnn")

    def normal_init(module, mean=0, std=1, bias=0):
        if hasattr(module, "weight") and module.weight is not None:
            nn.init.normal_(module.weight, mean, std)
        if hasattr(module, "bias") and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        if hasattr(module, "weight") and module.weight is not None:
            nn.init.constant_(module.weight, val)
        if hasattr(module, "bias") and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        if hasattr(module, "bias") and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    cnn.normal_init, cnn.constant_init, cnn.kaiming_init = normal_init, constant_init, kaiming_init
    mmcv.cnn = cnn
    runner = _pkg("mmcv.runner")
    runner.load_checkpoint = lambda *a, **k: None
    mmcv.runner = runner

    # ---- torchvision BasicBlock / Bottleneck (0.17.1) ----
    tv = _pkg("torchvision")
    tvm = _pkg("torchvision.models")
    tvr = _pkg("torchvision.models.resnet")

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                     norm_layer=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.downsample = downsample
            self.stride = stride

        def forward(self, x):
            identity = x
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            if self.downsample is not None:
                identity = self.downsample(x)
            out += identity
            return self.relu(out)

    class Bottleneck(nn.Module):
        expansion = 4

        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError("Bottleneck is outside the reference's runnable envelope here")

    tvr.BasicBlock, tvr.Bottleneck = BasicBlock, Bottleneck
    tv.models, tvm.resnet = tvm, tvr

    # ---- detectron2 bits ----
    d2 = _pkg("detectron2")
    d2l = _pkg("detectron2.layers")
    d2bn = _pkg("detectron2.layers.batch_norm")
    d2bn.BatchNorm2d = nn.BatchNorm2d
    d2bn.FrozenBatchNorm2d = nn.BatchNorm2d
    d2bn.NaiveSyncBatchNorm = nn.BatchNorm2d
    d2l.cat = torch.cat
    d2l.batch_norm = d2bn
    d2u = _pkg("detectron2.utils")
    d2env = _pkg("detectron2.utils.env")
    d2env.TORCH_VERSION = (2, 10)
    d2ev = _pkg("detectron2.utils.events")

    class _Storage:
        def __init__(self):
            self.scalars = {}

        def put_scalars(self, **kw):
            self.scalars.update(kw)

        def put_scalar(self, k, v, **kw):
            self.scalars[k] = v

    _st = _Storage()
    d2ev.get_event_storage = lambda: _st
    d2ev.EventStorage = _Storage
    d2u.env, d2u.events = d2env, d2ev
    d2u.comm = MagicMock()
    sys.modules["detectron2.utils.comm"] = d2u.comm
    d2.layers, d2.utils = d2l, d2u

    # ---- transforms3d.axangles.axangle2mat (0.4.2) ----
    t3 = _pkg("transforms3d")
    t3a = _pkg("transforms3d.axangles")

    def axangle2mat(axis, angle, is_normalized=False):
        x, y, z = axis
        if not is_normalized:
            n = math.sqrt(x * x + y * y + z * z)
            x, y, z = x / n, y / n, z / n
        c, s = math.cos(angle), math.sin(angle)
        C = 1 - c
        xs, ys, zs = x * s, y * s, z * s
        xC, yC, zC = x * C, y * C, z * C
        xyC, yzC, zxC = x * yC, y * zC, z * xC
        return np.array([[x * xC + c, xyC - zs, zxC + ys], [xyC + zs, y * yC + c, yzC - xs],
                         [zxC - ys, yzC + xs, z * zC + c]])

    t3a.axangle2mat = axangle2mat
    t3a.mat2axangle = MagicMock()
    t3.axangles = t3a

    sys.meta_path.append(_MockFinder())
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


class AttrDict(dict):
    """dict with attribute access (the reference's factory uses both .get()/.pop() and attributes)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    return d
