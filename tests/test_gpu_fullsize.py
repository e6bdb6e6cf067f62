"""-m gpu: BASELINE-size parity.

(a) configs[1] as bench.py runs it -- 32 synthetic 800x1333 images, ResNet-101-C4 fp16, R = 300, D = 100 -- checked
    through size-independent properties, against an N = 2 run of its first two images (bitwise) and one-stream against
    two-stream backbone (bitwise) at real size;
(b) ONE full-size image against the oracle: strict fp32 stage chain + free-running (1e-3), and the fp16 fast mode stage
    by stage against the fp16-emulating oracle (1e-3), with its deviation from the fp32 oracle reported.
The oracle needs ~5-10 s per full-size image and pass on the box's 16 host cores.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.frcnn_oracle import FRCNNOracle            # noqa: E402
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config   # noqa: E402

import gpu_util as G                                   # noqa: E402
from test_gpu_e2e import nchw, stage_chain_check       # noqa: E402


@pytest.fixture(scope="module")
def bench_model():
    cfg = vg_c4_config(post_nms_topk=300, detections=100)
    sd = make_state_dict(cfg, seed=1234)
    return cfg, sd, FRCNN(cfg, precision="fp16").load_state_dict(sd).eval()


def test_configs1_batch32_fp16(bench_model):
    cfg, sd, m = bench_model
    B = 32
    x = torch.from_numpy(synthetic_images(B, 800, 1333, seed=0xF2C)).cuda()
    shapes = torch.tensor([[800, 1333]] * B)
    out = m(x, shapes)
    full = {k: v.clone() for k, v in m.forward_padded().items()}
    res4 = m.get_stage("res4")
    assert res4.shape == (B, 50, 84, 1024) and res4.dtype == torch.float16 and torch.isfinite(res4).all()
    assert m.get_stage("proposal_counts").cpu().tolist() == [300] * B
    ppi = out["preds_per_image"].tolist()
    assert all(1 <= p <= 100 for p in ppi), ppi
    for i in range(B):
        f, b, p = out["roi_features"][i], out["boxes"][i], out["obj_probs"][i]
        assert f.shape == (ppi[i], 2048) and torch.isfinite(f).all() and (f >= 0).all() and f.max() > 0
        assert (b[:, 0] >= 0).all() and (b[:, 2] <= 1333).all() and (b[:, 1] >= 0).all() and (b[:, 3] <= 800).all()
        assert (b[:, 2] >= b[:, 0]).all() and (b[:, 3] >= b[:, 1]).all()
        assert (p[:-1] >= p[1:]).all() and (p > 0).all() and (p <= 1).all()        # NMS keeps score order
        assert (out["obj_ids"][i] >= 0).all() and (out["obj_ids"][i] < 1600).all()
        assert (out["attr_ids"][i] >= 0).all() and (out["attr_ids"][i] < 400).all()
        assert (full["roi_features"][i, ppi[i]:] == 0).all()                        # rows past the count stay zero
    assert len({tuple(out["obj_ids"][i].tolist()) for i in range(B)}) > 1           # images differ, so do detections
    # an image's outputs do not depend on what else is in the batch: images 0-1 of the 32 == a batch of just those two
    m(x[:2], shapes[:2])
    two = m.forward_padded()
    for k in full:
        assert torch.equal(full[k][:2], two[k]), k
    assert torch.equal(res4[:2], m.get_stage("res4"))
    # res3/res4 as two half-batches on two streams (the default at this size) == one stream, at real size
    m.set_option("backbone_streams", 1)
    m(x, shapes)
    one = m.forward_padded()
    for k in full:
        assert torch.equal(full[k], one[k]), k
    assert torch.equal(res4, m.get_stage("res4"))
    m.set_option("backbone_streams", 2)
    # run-to-run reproducibility at real size
    m(x, shapes)
    for k in full:
        assert torch.equal(full[k], m.forward_padded()[k]), k


@pytest.fixture(scope="module")
def one_image():
    x = synthetic_images(1, 800, 1333, seed=0xF2C)
    return torch.from_numpy(x), [[800, 1333]]


def test_full_size_image_fp32_strict_vs_oracle(bench_model, one_image):
    """One 800x1333 image, R = 300, D = 100: every stage of the strict mode against the oracle fed with the GPU's own
    upstream tensors (indices bit-exact: 63 000 anchors -> 6000 -> 300; 300 RoIs -> detections), and the free-running
    comparison of res4 / RoI features / outputs at 1e-3 (north_star)."""
    cfg, sd, _ = bench_model
    x, shapes = one_image
    m = FRCNN(cfg, precision="fp32").load_state_dict(sd).eval()
    out = m(x, torch.tensor(shapes))
    oracle = FRCNNOracle(cfg, sd)
    torch.set_num_threads(16)
    res4, feat = stage_chain_check(m, out, oracle, shapes, tol=1e-4)
    ref, st = oracle.forward(x, shapes, return_stages=True)
    e_res4 = G.rel_err(res4, st["res4"])
    print(f"\n[full size, fp32 strict vs oracle, free-running] res4 rel err {e_res4:.3e}")
    assert e_res4 <= 1e-3
    same = out["obj_ids"][0].cpu().tolist() == ref["obj_ids"][0].tolist()
    print(f"[full size, fp32 strict vs oracle, free-running] detections identical: {same} ({int(out['preds_per_image'][0])})")
    assert same
    for k in ("roi_features", "boxes", "obj_probs", "attr_probs"):
        e = G.rel_err(out[k][0].cpu(), ref[k][0])
        print(f"[full size, fp32 strict vs oracle, free-running] {k} rel err {e:.3e}")
        assert e <= 1e-3, k
    np.testing.assert_array_equal(out["attr_ids"][0].cpu().numpy(), ref["attr_ids"][0].numpy())


def test_full_size_image_fp16_fast_vs_emulating_oracle(bench_model, one_image):
    """The benched mode on one full-size image: stage by stage against the fp16-emulating oracle at 1e-3 (the conv stages
    at their real grids: 525-tile res4, 300-RoI head chunk), and its deviation from the fp32 oracle on the SAME RoIs
    (RoI features and class logits, bound 2e-3 at this depth and size; the 1e-3 of north_star is asserted on the reference
    golden fixture in test_gpu_e2e.py)."""
    cfg, sd, m = bench_model
    x, shapes = one_image
    out = m(x, torch.tensor(shapes))
    torch.set_num_threads(16)
    oracle16 = FRCNNOracle(cfg, sd, emulate="fp16")
    res4 = nchw(m.get_stage("res4"))
    e = G.rel_err(res4, oracle16.backbone(x))
    print(f"\n[full size, fp16 vs fp16-emulating oracle, free-running backbone] res4 rel err {e:.3e}")
    assert e <= 5e-3
    stage_chain_check(m, out, oracle16, shapes, tol=1e-3)
    ref, st = FRCNNOracle(cfg, sd).forward(x, shapes, return_stages=True)
    print(f"[full size, fp16 vs fp32 oracle] res4 rel err {G.rel_err(res4, st['res4']):.3e}")
    # RoI features of the SAME RoIs: the fp32 oracle's own backbone map pooled at the GPU's proposal boxes.  (Free-running,
    # the two pipelines' proposals drift by fractions of a pixel, RoIPool's integer bin edges move and the pooled features
    # of a "matched" detection differ by 2e-2 -- a property of RoIPool under any input perturbation, not an arithmetic error.)
    R = m.config.RPN.POST_NMS_TOPK_TEST
    pb, pc = m.get_stage("proposal_boxes").cpu(), m.get_stage("proposal_counts").cpu()
    boxes = [pb[0, :int(pc[0])]]
    o32 = FRCNNOracle(cfg, sd)
    feat32 = o32.res5(o32.pool(st["res4"], boxes)).mean(dim=[2, 3])
    feat16 = m.get_stage("feature_pooled").cpu()[:int(pc[0])]
    e = G.rel_err(feat16, feat32)
    print(f"[full size, fp16 vs fp32 oracle, same {int(pc[0])} RoIs] feature_pooled rel err {e:.3e}")
    assert e <= 1e-3          # north_star's bound on what the path hands over (5.8e-4 measured)
    s32, a32, _ = o32.predictor(feat32)
    s16 = m.get_stage("obj_logits").cpu()[:int(pc[0]), :s32.shape[1]]
    e_l = G.rel_err(s16, s32)
    print(f"[full size, fp16 vs fp32 oracle, same RoIs] obj_logits rel err {e_l:.3e}")
    # 1.3e-3 measured; DESIGN.md 5a: no stage carries it (f16 weights alone: 1.07e-3, f16 storage alone: 1.22e-3 at this size)
    assert e_l <= 2e-3
    # free-running detections, matched by box and class; every detection of the fp16 run that the fp32 oracle does not have is
    # traced to its RoI and printed with its margin: the fp32 oracle's score of the SAME proposal against the oracle's cut (the
    # lowest score it kept among its max_detections) -- a detection at the cut, or one whose NMS rival sits at the IoU threshold,
    # flips under any perturbation of the logits, fp16 rounding included
    gb, rb = out["boxes"][0].cpu(), ref["boxes"][0]
    keep16 = m.get_stage("keep_ids").cpu()[0]
    p32_all, c32_all = torch.softmax(st["obj_logits"], -1)[:, :-1].max(-1)
    pb32 = st["proposal_boxes"][0]
    cut = float(ref["obj_probs"][0].min())
    kept32 = set(int(v) for v in st["keep_ids"][0].tolist())
    matched, unmatched = 0, []
    for i in range(len(gb)):
        d = (rb - gb[i]).abs().max(dim=1).values
        j = int(d.argmin())
        if d[j] <= 1.0 and int(out["obj_ids"][0][i]) == int(ref["obj_ids"][0][j]):
            matched += 1
            continue
        r16 = int(keep16[i])
        dd = (pb32 - pb[0, r16]).abs().max(dim=1).values
        r32 = int(dd.argmin())
        unmatched.append((i, r16, r32, float(dd[r32]), float(out["obj_probs"][0][i]), float(p32_all[r32]), int(out["obj_ids"][0][i]),
                          int(c32_all[r32]), r32 in kept32))
    print(f"[full size, fp16 vs fp32 oracle, free-running] {matched} of {len(gb)} detections matched by box (1 px) and class; "
          f"the oracle's cut (lowest kept score) {cut:.4f}")
    for i, r16, r32, dist, p16, p32, c16, c32, in32 in unmatched:
        print(f"    detection {i}: RoI {r16} (fp32 RoI {r32}, proposal boxes {dist:.3f} px apart) score fp16 {p16:.4f} / fp32 {p32:.4f}, "
              f"class {c16} / {c32}, margin to the cut {p32 - cut:+.4f}, {'kept by the oracle too (box or class differs)' if in32 else 'not kept by the oracle'}")
    assert matched >= len(gb) - 5, unmatched
