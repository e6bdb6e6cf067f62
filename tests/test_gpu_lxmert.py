"""-m gpu: N3, the LXMERT-style encoder on the HIP path (vltk_amd.LxmertEncoder -> C ABI) against
(a) vectors produced by transformers.LxmertModel itself (fp32 strict mode, 1e-3 as north_star asks, measured ~1e-5),
(b) the oracle restating the bf16 / fp16 storage roundings (free-running over 6 layers + embeddings + pooler).
Op-level checks first: linear (+ GELU / tanh / residual), LayerNorm (scale / accumulate), embedding + LayerNorm, attention."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.lxmert_oracle import LxmertOracle          # noqa: E402
from vltk_amd import _lib as L                         # noqa: E402
from vltk_amd.lxmert import LxmertEncoder              # noqa: E402

import gpu_util as G                                   # noqa: E402
from lxmert_util import case_kwargs, golden_inputs     # noqa: E402

TDT = {L.VK_F32: torch.float32, L.VK_F16: torch.float16, L.VK_BF16: torch.bfloat16}
EPS = {L.VK_F32: 2e-5, L.VK_F16: 1e-3, L.VK_BF16: 8e-3}          # one rounding of the storage type (+ margin)
IDS = ["fp32", "fp16", "bf16"]


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "lxmert_small.npz"))


@pytest.mark.parametrize("dt", [L.VK_F32, L.VK_F16, L.VK_BF16], ids=IDS)
@pytest.mark.parametrize("M,K,N,act,res", [(77, 128, 128, 0, False), (396, 256, 768, 2, False), (108, 768, 256, 0, True),
                                           (33, 2048, 128, 3, False), (640, 128, 384, 1, True)])
def test_linear(dt, M, K, N, act, res):
    """nn.Linear + residual + {ReLU, GELU(erf), tanh} on the MFMA GEMM (bf16: v_mfma_f32_16x16x32_bf16)."""
    gen = np.random.Generator(np.random.PCG64(M + K))
    td = TDT[dt]
    x = torch.from_numpy(gen.standard_normal((M, K)).astype(np.float32)).to(td)
    w = (gen.standard_normal((N, K)) * (1.0 / K) ** 0.5).astype(np.float32)
    b = gen.standard_normal(N).astype(np.float32) * 0.1
    r = torch.from_numpy(gen.standard_normal((M, N)).astype(np.float32)).to(td) if res else None
    wp, bp = G.pack_conv(w.reshape(N, K, 1, 1), None, b, dt)
    xd, rd = x.to(G.DEV), (r.to(G.DEV) if res else None)
    y = torch.empty((M, N), dtype=td, device=G.DEV)
    L.call("vk_linear", G.P(xd), M, K, G.P(wp), G.P(bp), G.P(rd), G.P(y), N, N, act, dt, dt, G.stream())
    torch.cuda.synchronize()
    ref = x.float() @ torch.from_numpy(w).to(td).float().t() + torch.from_numpy(b)
    if res:
        ref = ref + r.float()
    ref = {0: lambda t: t, 1: F.relu, 2: F.gelu, 3: torch.tanh}[act](ref).to(td).float()
    assert rel(y, ref) <= EPS[dt]


@pytest.mark.parametrize("dt", [L.VK_F32, L.VK_F16, L.VK_BF16], ids=IDS)
@pytest.mark.parametrize("M,Cc", [(50, 128), (7, 768), (130, 2048), (3, 100)])
def test_layernorm(dt, M, Cc):
    gen = np.random.Generator(np.random.PCG64(Cc))
    td = TDT[dt]
    x = torch.from_numpy((gen.standard_normal((M, Cc)) * 3 + 1).astype(np.float32)).to(td)
    gm = torch.from_numpy(gen.uniform(0.5, 1.5, Cc).astype(np.float32))
    bt = torch.from_numpy(gen.standard_normal(Cc).astype(np.float32))
    xd, gd, bd = x.to(G.DEV), gm.to(G.DEV), bt.to(G.DEV)
    y = torch.empty((M, Cc), dtype=td, device=G.DEV)
    L.call("vk_layernorm", G.P(xd), Cc, G.P(gd), G.P(bd), G.P(y), Cc, M, Cc, 1e-12, 0.5, 0, dt, G.stream())
    ref1 = (F.layer_norm(x.float(), (Cc,), gm, bt, 1e-12) * 0.5).to(td)
    L.call("vk_layernorm", G.P(xd), Cc, G.P(gd), G.P(bd), G.P(y), Cc, M, Cc, 1e-12, 0.5, 1, dt, G.stream())     # y += 0.5 * LN(x)
    torch.cuda.synchronize()
    ref2 = (F.layer_norm(x.float(), (Cc,), gm, bt, 1e-12) * 0.5 + ref1.float()).to(td).float()
    assert rel(y, ref2) <= EPS[dt]


@pytest.mark.parametrize("dt", [L.VK_F32, L.VK_BF16], ids=["fp32", "bf16"])
def test_embed_layernorm(dt):
    gen = np.random.Generator(np.random.PCG64(3))
    td = TDT[dt]
    B, Lq, Cc, Vv = 3, 9, 128, 50
    word, pos, typ = (torch.from_numpy(gen.standard_normal(s).astype(np.float32)).to(td) for s in ((Vv, Cc), (16, Cc), (2, Cc)))
    ids = torch.from_numpy(gen.integers(0, Vv, (B, Lq)))
    tts = torch.from_numpy(gen.integers(0, 2, (B, Lq)))
    gm = torch.from_numpy(gen.uniform(0.5, 1.5, Cc).astype(np.float32))
    bt = torch.from_numpy(gen.standard_normal(Cc).astype(np.float32))
    dev = [t.to(G.DEV) for t in (ids, tts, word, pos, typ, gm, bt)]
    y = torch.empty((B * Lq, Cc), dtype=td, device=G.DEV)
    L.call("vk_embed_layernorm", G.P(dev[0]), G.P(dev[1]), B, Lq, G.P(dev[2]), G.P(dev[3]), G.P(dev[4]), G.P(dev[5]), G.P(dev[6]), G.P(y),
           Cc, 1e-12, dt, G.stream())
    torch.cuda.synchronize()
    e = word.float()[ids] + pos.float()[torch.arange(Lq)][None] + typ.float()[tts]
    ref = F.layer_norm(e, (Cc,), gm, bt, 1e-12).to(td).float().view(B * Lq, Cc)
    assert rel(y, ref) <= EPS[dt]


@pytest.mark.parametrize("dt", [L.VK_F32, L.VK_BF16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("B,heads,Lq,Lk,d,masked", [(3, 4, 11, 36, 32, True), (2, 12, 36, 20, 64, True), (1, 2, 5, 5, 16, False),
                                                    (5, 12, 20, 36, 64, True), (3, 12, 36, 36, 64, False), (2, 3, 48, 64, 64, True),
                                                    (7, 1, 1, 1, 32, False), (2, 2, 49, 20, 64, False)])
def test_attention(dt, B, heads, Lq, Lk, d, masked):
    """LxmertAttention after the projections: soft-max(QK^T/sqrt(d) + mask) V, heads side by side in a row; q / k / v
    may be column slices of a fused projection (row strides differ from the width)."""
    gen = np.random.Generator(np.random.PCG64(Lq * Lk))
    td = TDT[dt]
    H = heads * d
    q = torch.from_numpy(gen.standard_normal((B * Lq, H)).astype(np.float32)).to(td)
    kv = torch.from_numpy(gen.standard_normal((B * Lk, 2 * H)).astype(np.float32)).to(td)
    m = np.ones((B, Lk), np.float32)
    if masked:
        m[0, Lk // 2:] = 0
    add = torch.from_numpy((1.0 - m) * np.finfo(np.float32).min)
    qd, kvd, md = q.to(G.DEV), kv.to(G.DEV), add.to(G.DEV)
    out = torch.empty((B * Lq, H), dtype=td, device=G.DEV)
    L.call("vk_attention", G.P(qd), H, G.P(kvd), 2 * H, C.c_void_p(kvd.data_ptr() + H * kvd.element_size()), 2 * H,
           G.P(md) if masked else None, G.P(out), H, B, heads, Lq, Lk, d, dt, G.stream())
    torch.cuda.synchronize()
    qh = q.float().view(B, Lq, heads, d).transpose(1, 2)
    kh = kv.float()[:, :H].reshape(B, Lk, heads, d).transpose(1, 2)
    vh = kv.float()[:, H:].reshape(B, Lk, heads, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / d ** 0.5
    if masked:
        s = s + add[:, None, None, :]
    # the 16-bit kernel (matrix cores: d 32 / 64, Lq <= 48, Lk <= 64) rounds the probabilities to the storage type before
    # the second product; the fp32 kernel (and the fall-back shapes) do not
    mfma = dt != L.VK_F32 and d in (32, 64) and Lq <= 48 and Lk <= 64
    p = F.softmax(s, -1)
    ref = ((p.to(td).float() if mfma else p) @ vh).permute(0, 2, 1, 3).reshape(B * Lq, H).to(td).float()
    assert rel(out, ref) <= EPS[dt]


@pytest.mark.parametrize("tag", ["masked", "plain"])
def test_encoder_fp32_vs_transformers_golden(g, tag):
    cfg, sd, feats = golden_inputs(g)
    m = LxmertEncoder(cfg, precision="fp32").load_state_dict(sd)
    kw = {k: torch.from_numpy(v) for k, v in case_kwargs(g, tag).items()}
    lang, visn, pooled = m(torch.from_numpy(g["input_ids"]), torch.from_numpy(feats), torch.from_numpy(g["visual_pos"]), **kw)
    errs = (rel(lang, g[f"{tag}/language_output"]), rel(visn, g[f"{tag}/vision_output"]), rel(pooled, g[f"{tag}/pooled_output"]))
    print(f"\n[LXMERT fp32 strict vs transformers, {tag}] rel err lang {errs[0]:.2e} visn {errs[1]:.2e} pooled {errs[2]:.2e}")
    assert max(errs) <= 1e-3


@pytest.mark.parametrize("precision,tol", [("bf16", 4e-2), ("fp16", 6e-3)])
def test_encoder_reduced_precision_vs_emulating_oracle(g, precision, tol):
    cfg, sd, feats = golden_inputs(g)
    m = LxmertEncoder(cfg, precision=precision).load_state_dict(sd)
    kw = case_kwargs(g, "masked")
    lang, visn, pooled = m(torch.from_numpy(g["input_ids"]), torch.from_numpy(feats), torch.from_numpy(g["visual_pos"]),
                           **{k: torch.from_numpy(v) for k, v in kw.items()})
    o_lang, o_visn, o_pooled = LxmertOracle(cfg, sd, emulate=precision).forward(g["input_ids"], feats, g["visual_pos"], **kw)
    errs = (rel(lang, o_lang), rel(visn, o_visn), rel(pooled, o_pooled))
    dev = (rel(lang, g["masked/language_output"]), rel(visn, g["masked/vision_output"]))
    print(f"\n[LXMERT {precision} vs emulating oracle] lang {errs[0]:.2e} visn {errs[1]:.2e} pooled {errs[2]:.2e};"
          f" vs transformers fp32: lang {dev[0]:.2e} visn {dev[1]:.2e}")
    # free-running over 6 layers: an accumulation-order difference flips single storage roundings (bf16: 3.9e-3 each)
    assert max(errs) <= tol


def test_encoder_full_size_runs_and_is_reproducible():
    """transformers' default LXMERT geometry (9 / 5 / 5 layers, hidden 768) at B = 8: finite, bit-reproducible."""
    from vltk_amd.lxmert import lxmert_config, make_lxmert_state_dict
    cfg = lxmert_config()
    m = LxmertEncoder(cfg, precision="bf16").load_state_dict(make_lxmert_state_dict(cfg, 1))
    gen = np.random.Generator(np.random.PCG64(0))
    ids = torch.from_numpy(gen.integers(1, cfg["vocab_size"], (8, 20)))
    feats = torch.from_numpy(np.maximum(gen.standard_normal((8, 36, 2048)), 0).astype(np.float32))
    pos = torch.from_numpy(gen.uniform(0, 1, (8, 36, 4)).astype(np.float32))
    a = m(ids, feats, pos)
    b = m(ids, feats, pos)
    for u, v in zip(a, b):
        assert torch.isfinite(u.float()).all() and torch.equal(u, v)
    assert a[0].shape == (8, 20, 768) and a[1].shape == (8, 36, 768) and a[2].shape == (8, 768)


def test_frcnn_features_into_lxmert_chain(golden_dir):
    """BASELINE config 5 at test size: FRCNN (HIP, fp16) -> padded [N, D, 2048] features + normalised boxes -> LXMERT (HIP),
    checked against the LXMERT oracle fed with the same features."""
    from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config
    from vltk_amd.lxmert import lxmert_config, make_lxmert_state_dict
    D = 12
    fcfg = vg_c4_config(depth=50, post_nms_topk=30, detections=D)
    det = FRCNN(fcfg).load_state_dict(make_state_dict(fcfg, seed=1234)).eval()
    x = torch.from_numpy(synthetic_images(2, 160, 224, seed=9))
    out = det(x, torch.tensor([[160, 224], [160, 224]]), padding="max_detections", max_detections=D, return_tensors="pt")
    feats, boxes = out["roi_features"].float().cpu(), out["normalized_boxes"].float().cpu()
    assert feats.shape == (2, D, 2048) and boxes.shape == (2, D, 4)
    vmask = (torch.arange(D)[None] < out["preds_per_image"].cpu()[:, None]).float()
    cfg = lxmert_config(vocab_size=300, hidden_size=128, num_attention_heads=4, intermediate_size=256, l_layers=1, x_layers=2, r_layers=1,
                        max_position_embeddings=16)
    sd = make_lxmert_state_dict(cfg, 5)
    ids = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).integers(1, 300, (2, 7)))
    got = LxmertEncoder(cfg, precision="fp32").load_state_dict(sd)(ids, feats, boxes, visual_attention_mask=vmask)
    ref = LxmertOracle(cfg, sd).forward(ids.numpy(), feats.numpy(), boxes.numpy(), visual_attention_mask=vmask.numpy())
    for a, b in zip(got, ref):
        assert rel(a, b) <= 1e-3


def test_graph_replay_matches_eager(g):
    """One forward captured into a HIP graph (torch.cuda.CUDAGraph over the library's launches) replays bit-identically
    and follows new inputs of the same shape."""
    cfg, sd, feats = golden_inputs(g)
    m = LxmertEncoder(cfg, precision="bf16").load_state_dict(sd)
    ids, f, p = torch.from_numpy(g["input_ids"]), torch.from_numpy(feats), torch.from_numpy(g["visual_pos"])
    am, vm = torch.from_numpy(g["attention_mask"]), torch.from_numpy(g["visual_attention_mask"])
    eager = [t.clone() for t in m(ids, f, p, attention_mask=am, visual_attention_mask=vm)]
    replay = m.capture(ids, f, p, attention_mask=am, visual_attention_mask=vm)
    got = replay(ids, f, p, attention_mask=am, visual_attention_mask=vm)
    torch.cuda.synchronize()
    for a, b in zip(got, eager):
        assert torch.equal(a, b)
    ids2 = torch.roll(ids, 1, dims=1)
    eager2 = [t.clone() for t in m(ids2, f, p, attention_mask=am, visual_attention_mask=vm)]
    got2 = replay(ids2, f, p, attention_mask=am, visual_attention_mask=vm)
    torch.cuda.synchronize()
    for a, b in zip(got2, eager2):
        assert torch.equal(a, b)
    assert not torch.equal(eager2[0], eager[0])


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_question_answering_head(g, precision, tol):
    """BASELINE config 5's model: LxmertForQuestionAnswering on the HIP path vs transformers' vectors (fp32) / the
    bf16-emulating oracle; the arg-max answers agree with transformers'."""
    from vltk_amd.lxmert import LxmertForQuestionAnswering, make_lxmert_qa_state_dict
    cfg, _, feats = golden_inputs(g)
    nqa = int(g["qa/num_labels"])
    sd = make_lxmert_qa_state_dict(cfg, nqa, int(g["seed"]))
    m = LxmertForQuestionAnswering(cfg, nqa, precision=precision).load_state_dict(sd)
    kw = case_kwargs(g, "masked")
    score = m(torch.from_numpy(g["input_ids"]), torch.from_numpy(feats), torch.from_numpy(g["visual_pos"]),
              **{k: torch.from_numpy(v) for k, v in kw.items()})
    ref = g["qa/question_answering_score"] if precision == "fp32" else \
        LxmertOracle(cfg, sd, emulate=precision).qa_forward(g["input_ids"], feats, g["visual_pos"], **kw)
    assert score.shape == (3, nqa)
    assert rel(score, ref) <= tol
    assert score.float().argmax(-1).cpu().tolist() == g["qa/question_answering_score"].argmax(-1).tolist()
