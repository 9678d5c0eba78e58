"""mmcv-free loader for the reference's python config files.

The reference reads configs with ``mmcv.Config.fromfile`` (/root/reference/core/gdrn_modeling/
main_gdrn.py:39-41): python files whose module-level names are the config, ``_base_`` (str or
list, relative path) names parents that are merged first, a dict carrying ``_delete_=True``
replaces instead of merging, and ``--opts A.B=V`` pairs are merged last
(core/utils/default_args_setup.py:65-67).  This module reproduces exactly that surface so the
reference's own ``configs/gdrn/**.py`` load unchanged, plus ``gdrn_base_cfg()`` - the hot-path
defaults of ``configs/_base_/gdrn_base.py:5-143`` + ``configs/gdrn/lm/a6_cPnP_lm13.py:44-67`` -
for when no config file is at hand (bench, smoke, tests).
"""
import ast
import copy
import os


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(f"config has no key '{k}'") from e

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(child, base):
    """mmcv semantics: child overrides base key-by-key; ``_delete_=True`` in a child dict replaces."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get("_delete_", False):
            out[k] = _merge(v, out[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = copy.deepcopy(v)
    return out


def _load_file(path):
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    scope = {"__file__": path}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)  # config files are python by design
    cfg = {k: v for k, v in scope.items() if not k.startswith("__") and not callable(v) and not _is_module(v)}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(_load_file(os.path.join(os.path.dirname(path), b)), merged)
    return _merge(cfg, merged)


def _is_module(v):
    import types

    return isinstance(v, types.ModuleType)


def _parse_value(s):
    if not isinstance(s, str):
        return s
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return s


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        return Config(_load_file(path))

    def merge_from_dict(self, opts):
        """``{"MODEL.CDPN.PNP_NET.MASK_ATTENTION": "mul"}`` or a list ``["A.B=1", ...]``."""
        if isinstance(opts, (list, tuple)):
            opts = dict(o.split("=", 1) for o in opts)
        for key, val in opts.items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = ConfigDict()
                node = node[p]
            node[parts[-1]] = _parse_value(val)
        return self


def gdrn_base_cfg(num_regions=32, mask_attention="none", device="cuda", num_classes=13):
    """Hot-path keys with the values of gdrn_base.py merged with lm/a6_cPnP_lm13.py."""
    return Config(
        MODEL=dict(
            DEVICE=device,
            WEIGHTS="",
            PIXEL_MEAN=[0.0, 0.0, 0.0],
            PIXEL_STD=[255.0, 255.0, 255.0],
            CDPN=dict(
                NAME="GDRN",
                TASK="rot",
                USE_MTL=False,
                BACKBONE=dict(PRETRAINED="", ARCH="resnet", NUM_LAYERS=34, INPUT_CHANNEL=3, INPUT_RES=256,
                              OUTPUT_RES=64, FREEZE=False),
                ROT_HEAD=dict(
                    FREEZE=False, ROT_CONCAT=False, XYZ_BIN=64, NUM_LAYERS=3, NUM_FILTERS=256, CONV_KERNEL_SIZE=3,
                    NORM="BN", NUM_GN_GROUPS=32, OUT_CONV_KERNEL_SIZE=1, NUM_CLASSES=num_classes,
                    ROT_CLASS_AWARE=False, XYZ_LOSS_TYPE="L1", XYZ_LOSS_MASK_GT="visib", XYZ_LW=1.0,
                    MASK_CLASS_AWARE=False, MASK_LOSS_TYPE="L1", MASK_LOSS_GT="trunc", MASK_LW=1.0,
                    MASK_THR_TEST=0.5, NUM_REGIONS=num_regions, REGION_CLASS_AWARE=False, REGION_LOSS_TYPE="CE",
                    REGION_LOSS_MASK_GT="visib", REGION_LW=1.0,
                ),
                PNP_NET=dict(
                    FREEZE=False, R_ONLY=False, LR_MULT=1.0,
                    PNP_HEAD_CFG=dict(type="ConvPnPNet", norm="GN", num_gn_groups=32, drop_prob=0.0),
                    WITH_2D_COORD=True, REGION_ATTENTION=True, MASK_ATTENTION=mask_attention,
                    TRANS_WITH_BOX_INFO="none", ROT_TYPE="allo_rot6d", TRANS_TYPE="centroid_z", Z_TYPE="REL",
                    NUM_PM_POINTS=3000, PM_LOSS_TYPE="L1", PM_SMOOTH_L1_BETA=1.0, PM_LOSS_SYM=False,
                    PM_NORM_BY_EXTENT=True, PM_R_ONLY=True, PM_DISENTANGLE_T=False, PM_DISENTANGLE_Z=False,
                    PM_T_USE_POINTS=False, PM_LW=1.0, ROT_LOSS_TYPE="angular", ROT_LW=0.0,
                    CENTROID_LOSS_TYPE="L1", CENTROID_LW=1.0, Z_LOSS_TYPE="L1", Z_LW=1.0, TRANS_LOSS_TYPE="L1",
                    TRANS_LOSS_DISENTANGLE=True, TRANS_LW=0.0, BIND_LOSS_TYPE="L1", BIND_LW=0.0,
                ),
                TRANS_HEAD=dict(ENABLED=False, FREEZE=True, LR_MULT=1.0, NUM_LAYERS=3, NUM_FILTERS=256, NORM="BN",
                                NUM_GN_GROUPS=32, CONV_KERNEL_SIZE=3, OUT_CHANNEL=3, TRANS_TYPE="centroid_z",
                                Z_TYPE="REL"),
            ),
        ),
        SOLVER=dict(IMS_PER_BATCH=24, BASE_LR=1e-4, BF16X3=True, GROUP_WGRAD=True, OPTIMIZER_CFG=dict(type="Ranger", lr=1e-4, weight_decay=0),
                    WEIGHT_DECAY=0.0, AMP=dict(ENABLED=False, DTYPE="bf16")),
        INPUT=dict(FORMAT="BGR", DZI_PAD_SCALE=1.5),
        # VIS_SCALARS / VIS_PERIOD (not reference keys): the vis/* scalars of GDRN.py:306-368 computed on the device, delivered every N steps
        TRAIN=dict(VIS_SCALARS=False, VIS_PERIOD=20),
        TEST=dict(USE_PNP=False, PNP_TYPE="ransac_pnp", AMP_TEST=False, AMP_DTYPE="bf16", BF16X3=True, FP16X2=True, FOLD_GLOBAL_MAX=True, CONV_BEFORE_UPSAMPLE=True, COMPOSE_CONV3_CONVT=True, PNP_H2=True, FUSE_HEAD_OUT=True, PNP_SIDE_STREAM=True, TEST_BBOX_TYPE="est"),
    )
