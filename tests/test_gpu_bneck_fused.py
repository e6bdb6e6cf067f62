"""-m gpu: a whole res2 BottleneckBlock as one kernel (csrc/bneck_fused.hip, vk_bottleneck64; BottleneckBlock.forward
frcnn.py:963-979) against (a) a torch restatement of the block with the fast mode's rounding points (f16 weights with BN
folded, f16 t1 / t2 / output, fp32 accumulate), (b) the same block chained from the layer-by-layer kernels (vk_conv2d /
vk_conv1x1_dual) -- expected bit-identical: same MFMA, same K order, same epilogue arithmetic -- and (c) through the whole
model with the kernel switched off (VK_BNECK_FUSED=0)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from vltk_amd import _lib as L                        # noqa: E402
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config   # noqa: E402

import gpu_util as G                                   # noqa: E402


def _bn(g, c):
    return (g.uniform(0.5, 1.5, c).astype(np.float32), (g.standard_normal(c) * 0.1).astype(np.float32),
            (g.standard_normal(c) * 0.1).astype(np.float32), g.uniform(0.5, 1.5, c).astype(np.float32))


def _h(t):
    return t.half().float()


def make_block(seed, cin, proj):
    g = np.random.Generator(np.random.PCG64(seed))
    p = {"w1": (g.standard_normal((64, cin, 1, 1)) * np.sqrt(2.0 / cin)).astype(np.float32), "bn1": _bn(g, 64),
         "w2": (g.standard_normal((64, 64, 3, 3)) * np.sqrt(2.0 / 576)).astype(np.float32), "bn2": _bn(g, 64),
         "w3": (g.standard_normal((256, 64, 1, 1)) * np.sqrt(2.0 / 64)).astype(np.float32), "bn3": _bn(g, 256)}
    if proj:
        p["wsc"] = (g.standard_normal((256, cin, 1, 1)) * np.sqrt(1.0 / cin)).astype(np.float32)
        p["bnsc"] = _bn(g, 256)
    return p


def reference(x, p, proj):
    """The block with the fast mode's rounding points (what oracle/frcnn_oracle.py emulate="fp16" does per block)."""
    x = _h(x)
    w1, b1 = G.fold_ref(p["w1"], p["bn1"], L.VK_F16)
    w2, b2 = G.fold_ref(p["w2"], p["bn2"], L.VK_F16)
    w3, b3 = G.fold_ref(p["w3"], p["bn3"], L.VK_F16)
    t1 = _h(F.relu(F.conv2d(x, w1) + b1.view(1, -1, 1, 1)))
    t2 = _h(F.relu(F.conv2d(t1, w2, padding=1) + b2.view(1, -1, 1, 1)))
    y = F.conv2d(t2, w3) + b3.view(1, -1, 1, 1)
    if proj:
        wsc, bsc = G.fold_ref(p["wsc"], p["bnsc"], L.VK_F16)
        y = y + F.conv2d(x, wsc) + bsc.view(1, -1, 1, 1)       # not rounded on its own: part of conv3's GEMM (model.hip can_fuse_shortcut)
    else:
        y = y + x
    return _h(F.relu(y))


def packed(p, proj):
    w1d, b1d = G.pack_conv(p["w1"], p["bn1"], None, L.VK_F16)
    w2d, b2d = G.pack_conv(p["w2"], p["bn2"], None, L.VK_F16)
    w3d, b3d = G.pack_conv(p["w3"], p["bn3"], None, L.VK_F16)
    if proj:      # rows [conv3 | shortcut], biases summed: vk_conv1x1_dual's layout (model.hip finalize_block)
        wsd, bsd = G.pack_conv(p["wsc"], p["bnsc"], None, L.VK_F16)
        cin = p["wsc"].shape[1]
        a = w3d.cpu().numpy().view(np.uint8).reshape(-1, 64 * 2)
        b = wsd.cpu().numpy().view(np.uint8).reshape(-1, cin * 2)
        w3d = torch.from_numpy(np.ascontiguousarray(np.concatenate([a, b], axis=1)).reshape(-1)).to(G.DEV)
        b3d = b3d + bsd
    return w1d, b1d, w2d, b2d, w3d, b3d


def run_fused(x, p, proj):
    w1d, b1d, w2d, b2d, w3d, b3d = packed(p, proj)
    xd = G.to_nhwc(x, L.VK_F16)
    N, H, W, cin = xd.shape
    y = torch.full((N, H, W, 256), float("nan"), dtype=torch.float16, device=G.DEV)
    L.call("vk_bottleneck64", G.P(xd), N, H, W, cin, int(proj), G.P(w1d), G.P(b1d), G.P(w2d), G.P(b2d), G.P(w3d), G.P(b3d), G.P(y), G.stream())
    torch.cuda.synchronize()
    return y


def run_layers(x, p, proj):
    """The same block from the layer-by-layer kernels (what the model ran before the fused kernel existed)."""
    w1d, b1d, w2d, b2d, w3d, b3d = packed(p, proj)
    xd = G.to_nhwc(x, L.VK_F16)
    N, H, W, cin = xd.shape
    dt = L.VK_F16
    t1 = torch.empty((N, H, W, 64), dtype=torch.float16, device=G.DEV)
    t2 = torch.empty_like(t1)
    y = torch.empty((N, H, W, 256), dtype=torch.float16, device=G.DEV)
    L.call("vk_conv2d", G.P(xd), N, H, W, cin, G.P(w1d), G.P(b1d), None, G.P(t1), 64, 64, 1, 1, 1, 0, 1, 1, 1, dt, dt, G.stream())
    L.call("vk_conv2d", G.P(t1), N, H, W, 64, G.P(w2d), G.P(b2d), None, G.P(t2), 64, 64, 3, 3, 1, 1, 1, 1, 1, dt, dt, G.stream())
    if proj:
        L.call("vk_conv1x1_dual", G.P(t2), 64, G.P(xd), cin, N * H * W, G.P(w3d), G.P(b3d), None, G.P(y), 256, 1, G.stream())
    else:
        L.call("vk_conv2d", G.P(t2), N, H, W, 64, G.P(w3d), G.P(b3d), G.P(xd), G.P(y), 256, 256, 1, 1, 1, 0, 1, 1, 1, dt, dt, G.stream())
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("form", ["rows", "tiles"])
@pytest.mark.parametrize("proj", [False, True], ids=["identity", "projection"])
@pytest.mark.parametrize("shape", [(2, 16, 64), (1, 8, 32), (3, 13, 45), (2, 40, 70), (1, 5, 7), (1, 1, 1), (2, 3, 31), (1, 67, 30), (1, 200, 333)],
                         ids=lambda s: "x".join(map(str, s)))
def test_bottleneck64_vs_reference_and_layers(shape, proj, form, monkeypatch):
    """Both forms of the kernel (row-streaming column strips, the default; 8 x 32 tiles): whole strips / tiles, ragged edges in both
    directions, images smaller than a strip or a tile (down to one pixel), a strip of exactly 30 columns, several row units, and
    the real res2 map of an 800 x 1333 image (200 x 333: 12 strips of 28 columns / 25 x 11 tiles)."""
    monkeypatch.setenv("VK_BNECK_ROWS", "1" if form == "rows" else "0")
    N, H, W = shape
    cin = 64 if proj else 256
    g = np.random.Generator(np.random.PCG64(H * 1000 + W))
    x = torch.from_numpy(g.standard_normal((N, cin, H, W)).astype(np.float32))
    x = F.relu(x) if not proj else x                 # a block input is a ReLU output (identity blocks) / the pooled stem output
    p = make_block(11 + int(proj), cin, proj)
    y = run_fused(x, p, proj)
    assert torch.isfinite(y.float()).all(), "unwritten output pixels"
    ref = reference(x, p, proj)
    e = G.rel_err(G.to_nchw(y, L.VK_F16), ref)
    print(f"\n[bottleneck64 {shape} proj={proj}] rel err vs the f16-rounding torch restatement {e:.3e}")
    assert e <= 1e-3
    y2 = run_layers(x, p, proj)
    same = torch.equal(y, y2)
    if not same:
        d = (y.float() - y2.float()).abs()
        print(f"[bottleneck64 {shape} proj={proj}] differs from the layer-by-layer kernels in {int((d > 0).sum())} of {d.numel()} "
              f"elements, max {float(d.max()):.3e}")
    assert same, "fused block != layer-by-layer kernels"


@pytest.mark.parametrize("form", ["rows", "tiles"])
def test_bottleneck64_many_tiles_reproducible(form, monkeypatch):
    """Several units per workgroup (the tile form's ring runs across tiles and changes slot phase from tile to tile; the row form
    restarts its pipeline per unit) at batch size: 8 images of the real res2 map; twice, bit-identical, and equal to the
    layer-by-layer kernels."""
    monkeypatch.setenv("VK_BNECK_ROWS", "1" if form == "rows" else "0")
    g = np.random.Generator(np.random.PCG64(5))
    x = F.relu(torch.from_numpy(g.standard_normal((8, 256, 200, 333)).astype(np.float32)))
    p = make_block(3, 256, False)
    y1 = run_fused(x, p, False)
    y2 = run_fused(x, p, False)
    assert torch.equal(y1, y2)
    assert torch.equal(y1, run_layers(x, p, False))


def test_bottleneck64_beyond_two_gigabytes():
    """Batch 64 of the real res2 map: 2.18 GB of x and of y -- byte offsets beyond 2^31 (the row form keeps them unsigned / 64-bit);
    the last image against the layer-by-layer kernels on a batch of its own."""
    N, H, W = 64, 200, 333
    g = torch.Generator(device=G.DEV).manual_seed(7)
    xd = torch.randn((N, H, W, 256), generator=g, device=G.DEV, dtype=torch.float16).relu_()
    p = make_block(5, 256, False)
    w1d, b1d, w2d, b2d, w3d, b3d = packed(p, False)
    y = torch.full((N, H, W, 256), float("nan"), dtype=torch.float16, device=G.DEV)
    L.call("vk_bottleneck64", G.P(xd), N, H, W, 256, 0, G.P(w1d), G.P(b1d), G.P(w2d), G.P(b2d), G.P(w3d), G.P(b3d), G.P(y), G.stream())
    torch.cuda.synchronize()
    assert torch.isfinite(y[-1].float()).all() and torch.isfinite(y[0].float()).all()
    for n in (0, 40, 63):                                  # images below, across and above the 2^31-byte mark
        y1 = torch.empty((1, H, W, 256), dtype=torch.float16, device=G.DEV)
        x1 = xd[n:n + 1].contiguous()
        L.call("vk_bottleneck64", G.P(x1), 1, H, W, 256, 0, G.P(w1d), G.P(b1d), G.P(w2d), G.P(b2d), G.P(w3d), G.P(b3d), G.P(y1), G.stream())
        torch.cuda.synchronize()
        assert torch.equal(y[n:n + 1], y1), n


@pytest.mark.parametrize("case", ["three_ragged", "one_small", "two_full_size"])
def test_model_with_and_without_the_fused_block(case, monkeypatch):
    """The whole forward with res2 on the fused kernel (default) and on the layer-by-layer kernels (VK_BNECK_FUSED=0):
    identical outputs and res4 -- a ragged batch, a single small image, and two images at the bench's size (800 x 1333, ResNet-101)."""
    if case == "two_full_size":
        cfg = vg_c4_config(post_nms_topk=300, detections=100)
        sd = make_state_dict(cfg, seed=1234)
        x = torch.from_numpy(synthetic_images(2, 800, 1333, seed=0xF2C))
        shapes = torch.tensor([[800, 1333], [800, 1333]])
    elif case == "one_small":
        cfg = vg_c4_config(depth=50, post_nms_topk=40, detections=10)
        sd = make_state_dict(cfg, seed=31)
        x = torch.from_numpy(synthetic_images(1, 64, 96, seed=11))
        shapes = torch.tensor([[64, 96]])
    else:
        cfg = vg_c4_config(depth=50, post_nms_topk=40, detections=10)
        sd = make_state_dict(cfg, seed=31)
        x = torch.from_numpy(synthetic_images(3, 131, 203, seed=11))
        shapes = torch.tensor([[131, 203], [97, 180], [120, 161]])
    m = FRCNN(cfg, precision="fp16").load_state_dict(sd).eval()
    m(x, shapes)
    a = {k: v.clone() for k, v in m.forward_padded().items()}
    r4 = m.get_stage("res4")
    monkeypatch.setenv("VK_BNECK_FUSED", "0")
    m(x, shapes)
    b = m.forward_padded()
    assert torch.equal(r4, m.get_stage("res4"))
    for k in a:
        assert torch.equal(a[k], b[k]), k
