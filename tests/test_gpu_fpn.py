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
