"""-m gpu: N4, the FPN-side ops on the HIP path.  Pinned by the reference's own fragments where it has them (top blocks, level
assignment: tests/golden/fpn_ops.npz); RoIAlign and the neck have no reference counterpart (parity unpinned) and are held
to the oracle's restatement of torchvision roi_align / detectron2 FPN."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fpn_oracle as fo                    # noqa: E402
from vltk_amd import _lib as L                         # noqa: E402
from vltk_amd.fpn import FPNNeck, LastLevelP6P7, MultiLevelRoIAlign   # noqa: E402

import gpu_util as G                                   # noqa: E402


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "fpn_ops.npz"))


def nhwc(x, td=torch.float32):
    return x.permute(0, 2, 3, 1).contiguous().to(td).to(G.DEV)


def nchw(y):
    return y.float().permute(0, 3, 1, 2).contiguous().cpu()


def test_last_level_maxpool_vs_reference(g):
    x = torch.from_numpy(g["maxpool/x"])
    xd = nhwc(x)
    N, H, W, Cc = xd.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=torch.float32, device=G.DEV)
    L.call("vk_subsample2", G.P(xd), G.P(y), N, H, W, Cc, L.VK_F32, G.stream())
    np.testing.assert_array_equal(nchw(y).numpy(), g["maxpool/y"])


def test_last_level_p6p7_vs_reference(g):
    sd = {k.split("/sd/")[1]: g[k] for k in g.files if k.startswith("p6p7/sd/")}
    # the fixture's 16 output channels are half an fp32 K-tile for the second conv: widen with zero channels (same math)
    w6 = np.concatenate([sd["p6.weight"], np.zeros_like(sd["p6.weight"])], 0)
    b6 = np.concatenate([sd["p6.bias"], np.zeros_like(sd["p6.bias"])])
    w7 = np.concatenate([sd["p7.weight"], np.zeros_like(sd["p7.weight"])], 1)
    blk = LastLevelP6P7(w6, b6, w7, sd["p7.bias"], precision="fp32")
    p6, p7 = blk(nhwc(torch.from_numpy(g["p6p7/c5"])))
    assert G.rel_err(nchw(p6)[:, :16], g["p6p7/p6"]) <= 2e-5 and G.rel_err(nchw(p7), g["p6p7/p7"]) <= 2e-5


def test_assign_levels_vs_reference(g):
    boxes = torch.from_numpy(g["levels/boxes"]).to(G.DEV)
    out = torch.empty(boxes.shape[0], dtype=torch.int32, device=G.DEV)
    L.call("vk_assign_levels", G.P(boxes), 4, boxes.shape[0], 2, 5, 224.0, 4, G.P(out), G.stream())
    np.testing.assert_array_equal(out.cpu().numpy(), g["levels/assigned"])


def _rois(gen, K, N, w, h):
    xy = gen.uniform(-20, [w, h], (K, 2))
    wh = np.exp(gen.uniform(0, np.log(max(w, h) * 1.2), (K, 2)))
    r = np.concatenate([gen.integers(0, N, (K, 1)), xy, xy + wh], 1).astype(np.float32)
    r[0, 1:] = [w + 50, h + 50, w + 90, h + 120]        # completely outside
    r[1, 1:] = [10.2, 11.7, 10.2, 11.7]                 # empty box
    r[2, 1:] = [-30, -30, w + 30, h + 30]               # larger than the image
    return r


@pytest.mark.parametrize("dt,td,tol", [(L.VK_F32, torch.float32, 1e-5), (L.VK_F16, torch.float16, 2e-3)], ids=["fp32", "fp16"])
@pytest.mark.parametrize("sr,aligned", [(0, True), (2, True), (2, False), (0, False)])
def test_roi_align_single_level(dt, td, tol, sr, aligned):
    gen = np.random.Generator(np.random.PCG64(sr * 2 + aligned))
    N, Cc, H, W = 2, 40, 23, 31
    x = torch.from_numpy(gen.standard_normal((N, Cc, H, W)).astype(np.float32)).to(td).float()
    rois = _rois(gen, 40, N, W * 8, H * 8)
    pool = MultiLevelRoIAlign(7, [1 / 8], sr, aligned, precision="fp32" if dt == L.VK_F32 else "fp16")
    out, _ = pool([nhwc(x, td)], torch.from_numpy(rois))
    ref = fo.roi_align(x, rois, 7, 1 / 8, sr, aligned)
    if dt == L.VK_F16:
        ref = ref.half().float()
    assert G.rel_err(nchw(out), ref) <= tol


@pytest.mark.parametrize("P,sr", [(2, 0), (3, 0), (2, 3), (14, 0)])
def test_roi_align_footprints_beyond_the_weight_tables(P, sr):
    """The separable kernel keeps per-axis weight tables of 12 rows: bins of 15+ pixels under adaptive sampling, a fixed
    sampling_ratio over bins much larger than it (gaps between the samples) and inverted boxes take its per-sample loop;
    P = 14 runs two passes of the bin loop.  All against the oracle, fp32."""
    gen = np.random.Generator(np.random.PCG64(100 + P * 7 + sr))
    N, Cc, H, W = 2, 16, 40, 52
    x = torch.from_numpy(gen.standard_normal((N, Cc, H, W)).astype(np.float32))
    rois = _rois(gen, 48, N, W * 4, H * 4)
    rois[3, 1:] = [5, 6, W * 4 - 3, H * 4 - 2]            # the whole map in P x P bins
    rois[4, 1:] = [90, 80, 40, 20]                        # inverted
    pool = MultiLevelRoIAlign(P, [1 / 4], sr, True, precision="fp32")
    out, _ = pool([nhwc(x)], torch.from_numpy(rois))
    ref = fo.roi_align(x, rois, P, 1 / 4, sr, True)
    assert G.rel_err(nchw(out), ref) <= 1e-5


def test_roi_align_pyramid_routes_by_level():
    gen = np.random.Generator(np.random.PCG64(11))
    N, Cc = 2, 256
    feats = [torch.from_numpy(gen.standard_normal((N, Cc, 200 >> i, 336 >> i)).astype(np.float32)) for i in range(4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    rois = _rois(gen, 300, N, 1333, 800)
    pool = MultiLevelRoIAlign(7, scales, 2, True, precision="fp32")
    out, lv = pool([nhwc(f) for f in feats], torch.from_numpy(rois))
    ref, rlv = fo.multilevel_pool(feats, scales, rois, 7, "align", 2, True)
    np.testing.assert_array_equal(lv.cpu().numpy(), rlv.numpy())
    assert len(set(rlv.tolist())) == 4                  # every level is exercised
    assert G.rel_err(nchw(out), ref) <= 1e-5


@pytest.mark.parametrize("precision,td,tol", [("fp32", torch.float32, 2e-5), ("fp16", torch.float16, 2e-3)])
def test_fpn_neck_vs_oracle(precision, td, tol):
    """detectron2-style neck (no reference class: parity unpinned): lateral 1x1, nearest-2x top-down, 3x3 outputs, P6."""
    gen = np.random.Generator(np.random.PCG64(5))
    chans, Cc = [64, 128, 256, 512], 256
    sizes = [(50, 67), (25, 34), (13, 17), (7, 9)]      # odd sizes: the top-down map is cropped like detectron2 on padded inputs
    feats = [torch.from_numpy(gen.standard_normal((2, c, h, w)).astype(np.float32)).to(td).float() for c, (h, w) in zip(chans, sizes)]
    lat = [((gen.standard_normal((Cc, c, 1, 1)) * (1.0 / c) ** 0.5).astype(np.float32), gen.standard_normal(Cc).astype(np.float32) * 0.1)
           for c in chans]
    outc = [((gen.standard_normal((Cc, Cc, 3, 3)) * (1.0 / (9 * Cc)) ** 0.5).astype(np.float32), gen.standard_normal(Cc).astype(np.float32) * 0.1)
            for _ in chans]
    neck = FPNNeck(lat, outc, precision=precision)
    got = neck([nhwc(f, td) for f in feats])
    q = (lambda t: t.to(td).float())
    ref = fo.fpn_neck(feats, [(q(torch.from_numpy(w)), torch.from_numpy(b)) for w, b in lat],
                      [(q(torch.from_numpy(w)), torch.from_numpy(b)) for w, b in outc])
    assert len(got) == 5
    for a, b in zip(got, ref):
        assert tuple(nchw(a).shape) == tuple(b.shape)
        assert G.rel_err(nchw(a), b) <= tol


def _ml_call(objs, dlts, cells, strides, shapes, pre, post, thr, min_size=0.0):
    import ctypes as C
    nl, N, A = len(objs), objs[0].shape[0], objs[0].shape[1]
    lg = [o.permute(0, 2, 3, 1).contiguous().to(G.DEV) for o in objs]                      # [N,H,W,A]
    dl = [d.view(N, A, 4, d.shape[2], d.shape[3]).permute(0, 3, 4, 1, 2).reshape(N, d.shape[2], d.shape[3], 4 * A).contiguous().to(G.DEV)
          for d in dlts]                                                                   # [N,H,W,4A] in (a, coord) order
    ce = [torch.from_numpy(np.ascontiguousarray(c, np.float32)).to(G.DEV) for c in cells]
    P_ = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    I_ = lambda vs: (C.c_int32 * nl)(*[int(v) for v in vs])
    hw = torch.tensor(shapes, dtype=torch.int32, device=G.DEV)
    wts = (C.c_float * 4)(1.0, 1.0, 1.0, 1.0)
    ob = torch.zeros((N, post, 4), device=G.DEV)
    ol = torch.zeros((N, post), device=G.DEV)
    oc = torch.zeros(N, dtype=torch.int32, device=G.DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=G.DEV)
    nb = L.load().vk_rpn_multilevel_workspace_bytes(N, nl, pre, post)
    ws = torch.empty(nb, dtype=torch.uint8, device=G.DEV)
    L.call("vk_rpn_proposals_multilevel", P_(lg), I_([A] * nl), P_(dl), I_([4 * A] * nl), nl, N, I_([o.shape[2] for o in objs]),
           I_([o.shape[3] for o in objs]), A, P_(ce), I_(strides), 0.0, G.P(hw), wts, min_size, thr, pre, post, G.P(ob), G.P(ol), G.P(oc),
           G.P(flag), G.P(ws), nb, G.stream())
    torch.cuda.synchronize()
    assert int(flag) == 0
    return ob.cpu(), ol.cpu(), oc.cpu()


def test_multilevel_proposals_vs_reference(g):
    from test_fpn_oracle import _ml_inputs
    objs, dlts, cells, strides, shapes, pre, post, thr = _ml_inputs(g)
    ob, ol, oc = _ml_call(objs, dlts, cells, strides, shapes, pre, post, thr)
    for i in range(len(shapes)):
        c = int(oc[i])
        assert c == len(g[f"mlrpn/logits_{i}"])
        np.testing.assert_array_equal(ol[i, :c].numpy(), g[f"mlrpn/logits_{i}"])
        assert G.rel_err(ob[i, :c], g[f"mlrpn/boxes_{i}"]) <= 2e-6


def test_multilevel_proposals_fpn_size_vs_oracle():
    """Five levels of an 800x1333 image (strides 4..64, 3 anchors per cell, 1000 per level -> 1000 kept)."""
    gen = np.random.Generator(np.random.PCG64(4))
    strides = [4, 8, 16, 32, 64]
    hw = [(200, 334), (100, 167), (50, 84), (25, 42), (13, 21)]
    A, N = 3, 2
    objs = [torch.from_numpy(gen.standard_normal((N, A, h, w)).astype(np.float32)) for h, w in hw]
    dlts = [torch.from_numpy((gen.standard_normal((N, 4 * A, h, w)) * 0.5).astype(np.float32)) for h, w in hw]
    cells = []
    for sz in (32, 64, 128, 256, 512):
        c = []
        for r in (0.5, 1.0, 2.0):
            w_ = (sz * sz / r) ** 0.5
            h_ = r * w_
            c.append([-w_ / 2, -h_ / 2, w_ / 2, h_ / 2])
        cells.append(np.asarray(c, np.float32))
    shapes = [[800, 1333], [760, 1200]]
    ob, ol, oc = _ml_call(objs, dlts, cells, strides, shapes, 1000, 1000, 0.7)
    ref = fo.multilevel_proposals(objs, dlts, cells, strides, shapes, 1000, 1000, 0.7)
    for i, (b, s_) in enumerate(ref):
        c = int(oc[i])
        assert c == len(s_)
        np.testing.assert_array_equal(ol[i, :c].numpy(), s_.numpy())
        assert G.rel_err(ob[i, :c], b) <= 2e-6
