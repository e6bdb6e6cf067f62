"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/vltk_hip.h declares;
host-only entry points (weight packing, geometry helpers, argument validation) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vltk_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "vltk_hip.h")).read()
    declared = set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert lib.vk_version() == 1


def test_struct_layout_matches_header(lib):
    # vk_config: 7 ints, int+8f, int+8f, f, int, f, (pad) double, 2 ints, 4f, 6 ints, 4f, int  -> checked via sizeof
    assert C.sizeof(L.vk_roi_params) == 8 + 8 * 8 + 8
    assert C.sizeof(L.vk_outputs) == 7 * 8
    assert L.vk_config.rpn_nms_thresh.offset % 8 == 0


def test_stem_geometry(lib):
    ho, wo = C.c_int(), C.c_int()
    lib.vk_stem_out_hw(800, 1333, 1, C.byref(ho), C.byref(wo))
    assert (ho.value, wo.value) == (200, 333)
    lib.vk_stem_out_hw(800, 1333, 0, C.byref(ho), C.byref(wo))
    assert (ho.value, wo.value) == (200, 334)
    lib.vk_stem_out_hw(160, 224, 1, C.byref(ho), C.byref(wo))
    assert (ho.value, wo.value) == (40, 56)


def _half_bits(a):
    return np.asarray(a, dtype=np.float16).view(np.uint16)


@pytest.mark.parametrize("dt,npdt", [(L.VK_F32, np.float32), (L.VK_F16, np.float16)])
def test_pack_conv_weight_folds_bn(lib, dt, npdt):
    g = np.random.Generator(np.random.PCG64(0))
    cout, cin, k = 5, 64, 3
    w = g.standard_normal((cout, cin, k, k)).astype(np.float32)
    bn = np.concatenate([g.uniform(0.5, 1.5, cout), g.standard_normal(cout), g.standard_normal(cout),
                         g.uniform(0.5, 1.5, cout)]).astype(np.float32)
    nbytes = lib.vk_packed_weight_bytes(cout, cin, k, k, 1, dt)
    cp = lib.vk_packed_cout(cout)
    assert cp == 128 and nbytes == cp * k * k * cin * np.dtype(npdt).itemsize
    wp = np.zeros(nbytes, np.uint8)
    bp = np.zeros(cp, np.float32)
    L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p), bn.ctypes.data_as(C.c_void_p), None, cout, cin, k, k, 1, dt,
           wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
    s = bn[:cout].astype(np.float64) / np.sqrt(bn[3 * cout:].astype(np.float64) + 1e-5)
    ref_w = (w.astype(np.float64) * s[:, None, None, None]).astype(np.float32).transpose(0, 2, 3, 1).reshape(cout, -1)
    ref_b = (bn[cout:2 * cout].astype(np.float64) - bn[2 * cout:3 * cout].astype(np.float64) * s).astype(np.float32)
    got = wp.view(npdt).reshape(cp, -1)
    np.testing.assert_array_equal(got[:cout], ref_w.astype(npdt))
    assert (got[cout:] == 0).all()
    np.testing.assert_array_equal(bp[:cout], ref_b)
    assert (bp[cout:] == 0).all()


@pytest.mark.parametrize("c,groups", [(128, 32), (256, 8), (256, 2), (32, 8)])
def test_pack_grouped_is_slice_diagonal(lib, c, groups):
    """Grouped 3x3 weights (BottleneckBlock conv2, frcnn.py:942-952) are packed as dense rows over the
    input-channel slice of each 64-channel output tile, zero outside the channel's own group."""
    g = np.random.Generator(np.random.PCG64(5))
    cpg, k = c // groups, 3
    w = g.standard_normal((c, cpg, k, k)).astype(np.float32)
    sw = lib.vk_conv_slice_channels(c, groups)
    assert sw == min(max(cpg, 64), c)
    nbytes = lib.vk_packed_weight_bytes(c, c, k, k, groups, L.VK_F32)
    cp = lib.vk_packed_cout(c)
    assert nbytes == cp * k * k * sw * 4
    wp = np.zeros(nbytes, np.uint8)
    bp = np.zeros(cp, np.float32)
    L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p), None, None, c, c, k, k, groups, L.VK_F32,
           wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
    got = wp.view(np.float32).reshape(cp, k * k, sw)
    for co in range(c):
        slice0 = (co // 64 * 64) // sw * sw
        g0 = co // cpg * cpg
        ref = np.zeros((k * k, sw), np.float32)
        ref[:, g0 - slice0:g0 - slice0 + cpg] = w[co].transpose(1, 2, 0).reshape(k * k, cpg)
        np.testing.assert_array_equal(got[co], ref)
    assert (got[c:] == 0).all()


def test_pack_grouped_rejects_odd_width(lib):
    assert lib.vk_conv_slice_channels(96, 8) == -1          # 12 channels per group: not a power of two
    w = np.zeros((96, 12, 3, 3), np.float32)
    out = np.zeros(1 << 20, np.uint8)
    b = np.zeros(128, np.float32)
    with pytest.raises(ValueError, match="power of two"):
        L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p), None, None, 96, 96, 3, 3, 8, L.VK_F32,
               out.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))


def test_pack_rejects_bad_cin(lib):
    w = np.zeros((4, 3, 1, 1), np.float32)
    out = np.zeros(1 << 16, np.uint8)
    b = np.zeros(128, np.float32)
    with pytest.raises(ValueError, match="K-tiles"):
        L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p), None, None, 4, 3, 1, 1, 1, L.VK_F16,
               out.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))


def test_workspace_sizes(lib):
    assert lib.vk_rpn_workspace_bytes(2, 63000, 6000) > 2 * 6000 * 94 * 8
    assert lib.vk_nms_workspace_bytes(300) > 300 * 16
    assert lib.vk_stem_workspace_bytes(1, 800, 1333, 64, L.VK_F16) > 400 * 667 * 64 * 2
