"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/rdpn6d.h declares; the ctypes table covers them all; without a GPU the product path fails
loudly instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    from rdpn6d_amd import build, _lib

    build.build(verbose=False)
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rdpn6d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", src)))


def test_every_declared_symbol_is_exported(lib):
    from rdpn6d_amd import _lib

    names = _declared_symbols()
    assert "farthest_point_sampling" in names and "farthest_point_sampling_init_center" in names
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rdpn6d.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def _header_prototypes():
    """(name, return type, [parameter types]) of every prototype in include/rdpn6d.h, comments stripped"""
    src = open(os.path.join(ROOT, "include", "rdpn6d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b([a-z_0-9]+)\s*\(([^;{}]*)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith(("typedef", "#")) or not ret:
            continue
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        out.append((name, ret, plist))
    return out


def _ctype_of(decl, is_return=False):
    """the ctypes type a C parameter declaration binds to in rdpn6d_amd/_lib.py (pointers are untyped there, by design)"""
    from rdpn6d_amd import _lib

    d = re.sub(r"\bconst\b|\b__restrict__\b", " ", decl).strip()
    if "rdpn6d_conv_desc" in d and "*" in d:
        return ctypes.POINTER(_lib.ConvDesc)
    if "*" in d or "[" in d:
        if is_return and re.match(r"char\s*\*", d):
            return ctypes.c_char_p
        return ctypes.c_void_p
    base = re.sub(r"\b[A-Za-z_][A-Za-z_0-9]*$", "", d).strip() if not is_return else d  # drop the parameter name
    base = " ".join(base.split())
    table = {"int": ctypes.c_int, "float": ctypes.c_float, "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint,
             "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong, "double": ctypes.c_double, "void": None,
             "size_t": ctypes.c_size_t}
    if base not in table:
        raise AssertionError(f"unmapped C type {decl!r} -> {base!r}")
    return table[base]


def test_ctypes_signatures_match_the_header_prototypes():
    """rdpn6d_amd/_lib.py mirrors include/rdpn6d.h by hand (~170 prototypes).  Symbol presence says nothing about
    an argument that was added to one side only - a shifted int / float argument is a silently wrong call on this ABI (ints and floats
    travel in different registers).  Parse every prototype and hold the ctypes table to it: return type, parameter COUNT, and the
    class of every parameter (pointer | int | unsigned | float | long long | double | the descriptor pointer)."""
    from rdpn6d_amd import _lib

    protos = _header_prototypes()
    assert len(protos) == len(_declared_symbols()) == len(_lib.SIGNATURES)

    def same(a, b):  # (c_int / c_uint are different classes, c_longlong / c_ulonglong too; a descriptor pointer bound as void* is a pointer)
        ptr = (ctypes.c_void_p, ctypes.POINTER(_lib.ConvDesc))
        return a is b or (a in ptr and b in ptr)

    bad = []
    for name, ret, params in protos:
        rt, at = _lib.SIGNATURES[name]
        want_r = _ctype_of(ret, is_return=True)
        if not same(rt, want_r):
            bad.append(f"{name}: returns {ret!r} -> {want_r}, ctypes has {rt}")
        want = [_ctype_of(p) for p in params]
        if len(want) != len(at):
            bad.append(f"{name}: {len(want)} parameters in the header, {len(at)} in _lib.SIGNATURES")
            continue
        for i, (w, g, p) in enumerate(zip(want, at, params)):
            if not same(w, g):
                bad.append(f"{name}: parameter {i} {p!r} -> {w}, ctypes has {g}")
    assert not bad, "\n".join(bad)


def test_conv_desc_layout_matches_header():
    from rdpn6d_amd._lib import ConvDesc

    # 6 pointers + 10 ints + 2*9 ints + 12 ints + float  (see rdpn6d_conv_desc)
    assert ctypes.sizeof(ConvDesc) == 6 * 8 + (10 + 18 + 13) * 4 + 4
    assert ConvDesc.dy.offset == 6 * 8 + 10 * 4 and ConvDesc.N.offset == 6 * 8 + 28 * 4


def test_version_and_argument_validation(lib):
    assert lib.rdpn6d_version() >= 100
    assert lib.rdpn6d_device_count() >= 0
    from rdpn6d_amd._lib import ConvDesc

    d = ConvDesc()  # all zero -> rejected before any launch
    assert lib.rdpn6d_conv2d_f32(ctypes.byref(d), None) == -1
    assert b"null pointer" in lib.rdpn6d_last_error()
    assert lib.rdpn6d_fps_host(None, None, 10, 4, -1) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="this checks the behaviour of a box WITHOUT a GPU")
def test_no_silent_cpu_fallback(lib):
    """fps with host pointers must fail loudly (idxs = -1) when there is no device."""
    pts = np.random.default_rng(0).standard_normal((100, 3)).astype(np.float32)
    idx = np.zeros(8, dtype=np.int32)
    P = ctypes.c_void_p
    assert lib.rdpn6d_fps_host(pts.ctypes.data_as(P), idx.ctypes.data_as(P), 100, 8, -1) == -2
    lib.farthest_point_sampling_init_center(pts.ctypes.data_as(P), idx.ctypes.data_as(P), 100, 8)
    assert (idx == -1).all()
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    model, _ = build_model_optimizer(gdrn_base_cfg(device="cpu"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 6, 256, 256), roi_coord_2d=torch.zeros(1, 5, 64, 64), fps=torch.zeros(1, 32, 3),
              roi_cams=torch.eye(3)[None], roi_centers=torch.zeros(1, 2), roi_whs=torch.ones(1, 2),
              resize_ratios=torch.ones(1))
