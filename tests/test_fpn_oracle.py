"""CPU: the FPN-side oracle (oracle/fpn_oracle.py) against vectors from the reference's own fragments
(LastLevelMaxPool, LastLevelP6P7, assign_boxes_to_levels), plus self-consistency of the unpinned RoIAlign restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import fpn_oracle as fo


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "fpn_ops.npz"))


def test_top_blocks_and_levels_vs_reference(g):
    np.testing.assert_array_equal(fo.last_level_maxpool(torch.from_numpy(g["maxpool/x"])).numpy(), g["maxpool/y"])
    sd = {k.split("/sd/")[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p6p7/sd/")}
    p6, p7 = fo.last_level_p6p7(torch.from_numpy(g["p6p7/c5"]), sd["p6.weight"], sd["p6.bias"], sd["p7.weight"], sd["p7.bias"])
    np.testing.assert_allclose(p6.numpy(), g["p6p7/p6"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(p7.numpy(), g["p6p7/p7"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(fo.assign_boxes_to_levels(g["levels/boxes"], 2, 5).numpy(), g["levels/assigned"])


def test_roi_align_properties():
    """No reference vector exists (parity unpinned): the restatement must at least reproduce a constant map, interpolate a
    linear ramp exactly at bin centres, and zero samples outside the map."""
    H, W = 12, 16
    const = torch.full((1, 3, H, W), 2.5)
    rois = torch.tensor([[0, 1.0, 2.0, 9.0, 7.0], [0, 3.3, 0.4, 15.2, 10.9]])
    for aligned in (True, False):
        for sr in (0, 2):
            y = fo.roi_align(const, rois, 7, 1.0, sr, aligned)
            np.testing.assert_allclose(y.numpy(), 2.5, atol=1e-6)
    ramp = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(1, 1, H, W).contiguous()
    y = fo.roi_align(ramp, torch.tensor([[0, 2.0, 2.0, 9.0, 9.0]]), 7, 1.0, 2, True)[0, 0]
    centres = 2.0 - 0.5 + (np.arange(7) + 0.5)           # aligned: pixel centres at +0.5, so the ramp value is x - 0.5
    np.testing.assert_allclose(y[3].numpy(), centres, atol=1e-5)
    far = fo.roi_align(const, torch.tensor([[0, 100.0, 100.0, 120.0, 130.0]]), 7, 1.0, 2, True)
    assert float(far.abs().max()) == 0.0


def test_multilevel_pool_routes_by_level():
    g_ = torch.Generator().manual_seed(0)
    feats = [torch.randn(2, 4, 64 >> i, 80 >> i, generator=g_) for i in range(4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    rois = torch.tensor([[0, 10, 10, 40, 50], [1, 0, 0, 250, 200], [0, 5, 5, 300, 240], [1, 100, 40, 160, 90]], dtype=torch.float32)
    out, lv = fo.multilevel_pool(feats, scales, rois, 7, "align", 2, True)
    assert lv.tolist() == fo.assign_boxes_to_levels(rois[:, 1:], 2, 5).tolist()
    for k in range(4):
        li = int(lv[k])
        np.testing.assert_array_equal(out[k].numpy(), fo.roi_align(feats[li], rois[k:k + 1], 7, scales[li], 2, True)[0].numpy())


def _ml_inputs(g):
    L_ = 3
    objs = [torch.from_numpy(g[f"mlrpn/obj_{i}"]) for i in range(L_)]
    dlts = [torch.from_numpy(g[f"mlrpn/dlt_{i}"]) for i in range(L_)]
    cells = [g[f"mlrpn/cell_{i}"] for i in range(L_)]
    pre, post, thr = g["mlrpn/pre_post_thr"].tolist()
    return objs, dlts, cells, g["mlrpn/strides"].tolist(), g["mlrpn/shapes"].tolist(), int(pre), int(post), float(thr)


def test_multilevel_proposals_vs_reference(g):
    """find_top_rpn_proposals over three levels, run by the reference itself on per-level inputs decoded with its own
    AnchorGenerator / Box2BoxTransform."""
    objs, dlts, cells, strides, shapes, pre, post, thr = _ml_inputs(g)
    res = fo.multilevel_proposals(objs, dlts, cells, strides, shapes, pre, post, thr)
    for i, (b, s_) in enumerate(res):
        np.testing.assert_array_equal(s_.numpy(), g[f"mlrpn/logits_{i}"])
        np.testing.assert_allclose(b.numpy(), g[f"mlrpn/boxes_{i}"], rtol=0, atol=1e-4)
