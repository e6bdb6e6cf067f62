"""N3: the LXMERT-style vision-language encoder that consumes the extractor's output.

The reference hands `roi_features [B,36,2048]` and `boxes [B,36,4]` to transformers' LXMERT
(`vltk/legacy/legacy_train.py:30-39`; `transformers` is a dependency of the reference, `requirements.txt`).  This module is
the host-side mirror of `transformers.LxmertModel` (modeling_lxmert.py v5.15: `LxmertModel.forward` :691-824,
`LxmertEncoder.forward` :498-557) with the state-dict key layout of that class, so a real checkpoint's tensors load by
name.  All arithmetic runs in `libvltk_hip.so` (MFMA GEMM with bias / residual / GELU / tanh epilogues, LayerNorm,
attention, embedding kernels); torch only owns the device buffers.  No CPU fallback.
"""
import ctypes as C
import zlib

import numpy as np
import torch

from . import _lib as L

DEFAULT_CONFIG = dict(vocab_size=30522, hidden_size=768, num_attention_heads=12, intermediate_size=3072, l_layers=9, x_layers=5,
                      r_layers=5, max_position_embeddings=512, type_vocab_size=2, visual_feat_dim=2048, visual_pos_dim=4)
LN_EPS = 1e-12


def lxmert_config(**kw):
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(kw)
    assert cfg["hidden_size"] % cfg["num_attention_heads"] == 0
    return cfg


def _att(p, H, ctx=None):
    ctx = H if ctx is None else ctx
    return [(p + ".query.weight", (H, H)), (p + ".query.bias", (H,)), (p + ".key.weight", (H, ctx)), (p + ".key.bias", (H,)),
            (p + ".value.weight", (H, ctx)), (p + ".value.bias", (H,))]


def _dense_ln(p, n_in, n_out):
    return [(p + ".dense.weight", (n_out, n_in)), (p + ".dense.bias", (n_out,)), (p + ".LayerNorm.weight", (n_out,)),
            (p + ".LayerNorm.bias", (n_out,))]


def _bert_layer(p, H, I):
    return (_att(p + ".attention.self", H) + _dense_ln(p + ".attention.output", H, H) +
            [(p + ".intermediate.dense.weight", (I, H)), (p + ".intermediate.dense.bias", (I,))] + _dense_ln(p + ".output", I, H))


def lxmert_param_spec(cfg):
    """(name, shape) of every tensor of `transformers.LxmertModel(config).state_dict()`, in its order."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    spec = [("embeddings.word_embeddings.weight", (cfg["vocab_size"], H)),
            ("embeddings.position_embeddings.weight", (cfg["max_position_embeddings"], H)),
            ("embeddings.token_type_embeddings.weight", (cfg["type_vocab_size"], H)),
            ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,)),
            ("encoder.visn_fc.visn_fc.weight", (H, cfg["visual_feat_dim"])), ("encoder.visn_fc.visn_fc.bias", (H,)),
            ("encoder.visn_fc.visn_layer_norm.weight", (H,)), ("encoder.visn_fc.visn_layer_norm.bias", (H,)),
            ("encoder.visn_fc.box_fc.weight", (H, cfg["visual_pos_dim"])), ("encoder.visn_fc.box_fc.bias", (H,)),
            ("encoder.visn_fc.box_layer_norm.weight", (H,)), ("encoder.visn_fc.box_layer_norm.bias", (H,))]
    for i in range(cfg["l_layers"]):
        spec += _bert_layer(f"encoder.layer.{i}", H, I)
    for i in range(cfg["x_layers"]):
        p = f"encoder.x_layers.{i}"
        spec += _att(p + ".visual_attention.att", H) + _dense_ln(p + ".visual_attention.output", H, H)
        spec += _att(p + ".lang_self_att.self", H) + _dense_ln(p + ".lang_self_att.output", H, H)
        spec += _att(p + ".visn_self_att.self", H) + _dense_ln(p + ".visn_self_att.output", H, H)
        spec += [(p + ".lang_inter.dense.weight", (I, H)), (p + ".lang_inter.dense.bias", (I,))] + _dense_ln(p + ".lang_output", I, H)
        spec += [(p + ".visn_inter.dense.weight", (I, H)), (p + ".visn_inter.dense.bias", (I,))] + _dense_ln(p + ".visn_output", I, H)
    for i in range(cfg["r_layers"]):
        spec += _bert_layer(f"encoder.r_layers.{i}", H, I)
    spec += [("pooler.dense.weight", (H, H)), ("pooler.dense.bias", (H,))]
    return spec


def lxmert_qa_param_spec(cfg, num_qa_labels):
    """`transformers.LxmertForQuestionAnswering(config).state_dict()`: the encoder under `lxmert.` + the answer head."""
    H = cfg["hidden_size"]
    return [("lxmert." + k, s) for k, s in lxmert_param_spec(cfg)] + [
        ("answer_head.logit_fc.0.weight", (2 * H, H)), ("answer_head.logit_fc.0.bias", (2 * H,)),
        ("answer_head.logit_fc.2.weight", (2 * H,)), ("answer_head.logit_fc.2.bias", (2 * H,)),
        ("answer_head.logit_fc.3.weight", (num_qa_labels, 2 * H)), ("answer_head.logit_fc.3.bias", (num_qa_labels,))]


def make_lxmert_qa_state_dict(cfg, num_qa_labels, seed=0):
    sd = {"lxmert." + k: v for k, v in make_lxmert_state_dict(cfg, seed).items()}
    for name, shape in lxmert_qa_param_spec(cfg, num_qa_labels)[-6:]:
        g = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        sd[name] = (g.uniform(0.5, 1.5, shape) if name.endswith("fc.2.weight") else g.standard_normal(shape) * 0.05).astype(np.float32)
    return sd


def make_lxmert_state_dict(cfg, seed=0):
    """Seeded synthetic weights (no checkpoint can be fetched offline): BERT-style N(0, 0.05) matrices (wider than the
    0.02 init so every layer matters in a parity check), LayerNorm gamma ~ U(0.5, 1.5), small biases."""
    sd = {}
    for name, shape in lxmert_param_spec(cfg):
        g = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if "LayerNorm.weight" in name or "layer_norm.weight" in name:
            v = g.uniform(0.5, 1.5, shape)
        elif name.endswith(".bias"):
            v = g.standard_normal(shape) * 0.05
        else:
            v = g.standard_normal(shape) * 0.05
        sd[name] = v.astype(np.float32)
    return sd


_DT = {"fp32": (L.VK_F32, torch.float32), "fp16": (L.VK_F16, torch.float16), "bf16": (L.VK_BF16, torch.bfloat16)}


class LxmertEncoder:
    """`transformers.LxmertModel` on the HIP path: `model(input_ids, visual_feats, visual_pos, attention_mask=None,
    visual_attention_mask=None, token_type_ids=None)` -> (language_output, vision_output, pooled_output)."""

    def __init__(self, config, precision="bf16", device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("vltk_amd.LxmertEncoder needs a GPU: there is no CPU fallback")
        L.load()
        self.cfg = dict(config)
        self.dt, self.tdt = _DT[precision]
        self.device = torch.device(device)
        self.es = 4 if precision == "fp32" else 2
        self.ktile = 128 // self.es                    # elements per 128-byte K-tile of the GEMM
        self._lin, self._ln, self._tab = {}, {}, {}
        self._loaded = False

    # ---- weights -------------------------------------------------------------------------------------------
    def _pack(self, w, b):
        """nn.Linear weight [N, K] (+ bias) -> device (packed rows padded to the K-tile, f32 bias)."""
        N, K = w.shape
        Kp = (K + self.ktile - 1) // self.ktile * self.ktile
        wp_ = np.zeros((N, Kp, 1, 1), np.float32)
        wp_[:, :K, 0, 0] = w
        lib = L.load()
        nb = lib.vk_packed_weight_bytes(N, Kp, 1, 1, 1, self.dt)
        wp = np.zeros(nb, np.uint8)
        bp = np.zeros(lib.vk_packed_cout(N), np.float32)
        bb = np.ascontiguousarray(b, dtype=np.float32)
        L.call("vk_pack_conv_weight", wp_.ctypes.data_as(C.c_void_p), None, bb.ctypes.data_as(C.c_void_p), N, Kp, 1, 1, 1, self.dt,
               wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
        return torch.from_numpy(wp).to(self.device), torch.from_numpy(bp).to(self.device), N, Kp

    def load_state_dict(self, sd, strict=True):
        sd = {k: (v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)) for k, v in sd.items()}
        want = dict(lxmert_param_spec(self.cfg))
        if strict:
            missing, extra = sorted(set(want) - set(sd)), sorted(set(sd) - set(want))
            if missing or extra:
                raise RuntimeError(f"LxmertEncoder.load_state_dict: missing {missing[:4]} unexpected {extra[:4]}")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"weight '{k}' has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")

        def lin(name, *parts):      # one GEMM for several nn.Linear applied to the same input (rows concatenated)
            w = np.concatenate([sd[p + ".weight"] for p in parts], 0)
            b = np.concatenate([sd[p + ".bias"] for p in parts], 0)
            self._lin[name] = self._pack(w, b)

        def ln(name, p):
            self._ln[name] = (torch.from_numpy(sd[p + ".weight"]).to(self.device), torch.from_numpy(sd[p + ".bias"]).to(self.device))

        def att_self(p):            # q, k, v of a self-attention share the input: one GEMM
            lin(p + ".qkv", p + ".query", p + ".key", p + ".value")

        def bert_layer(p):
            att_self(p + ".attention.self")
            lin(p + ".attention.output.dense", p + ".attention.output.dense")
            ln(p + ".attention.output.LayerNorm", p + ".attention.output.LayerNorm")
            lin(p + ".intermediate.dense", p + ".intermediate.dense")
            lin(p + ".output.dense", p + ".output.dense")
            ln(p + ".output.LayerNorm", p + ".output.LayerNorm")

        for t in ("word_embeddings", "position_embeddings", "token_type_embeddings"):
            self._tab[t] = torch.from_numpy(sd[f"embeddings.{t}.weight"]).to(self.device).to(self.tdt).contiguous()
        ln("embeddings.LayerNorm", "embeddings.LayerNorm")
        lin("encoder.visn_fc.visn_fc", "encoder.visn_fc.visn_fc")
        lin("encoder.visn_fc.box_fc", "encoder.visn_fc.box_fc")
        ln("encoder.visn_fc.visn_layer_norm", "encoder.visn_fc.visn_layer_norm")
        ln("encoder.visn_fc.box_layer_norm", "encoder.visn_fc.box_layer_norm")
        for i in range(self.cfg["l_layers"]):
            bert_layer(f"encoder.layer.{i}")
        for i in range(self.cfg["r_layers"]):
            bert_layer(f"encoder.r_layers.{i}")
        for i in range(self.cfg["x_layers"]):
            p = f"encoder.x_layers.{i}"
            lin(p + ".visual_attention.att.q", p + ".visual_attention.att.query")
            lin(p + ".visual_attention.att.kv", p + ".visual_attention.att.key", p + ".visual_attention.att.value")
            lin(p + ".visual_attention.output.dense", p + ".visual_attention.output.dense")
            ln(p + ".visual_attention.output.LayerNorm", p + ".visual_attention.output.LayerNorm")
            for m in ("lang", "visn"):
                att_self(f"{p}.{m}_self_att.self")
                lin(f"{p}.{m}_self_att.output.dense", f"{p}.{m}_self_att.output.dense")
                ln(f"{p}.{m}_self_att.output.LayerNorm", f"{p}.{m}_self_att.output.LayerNorm")
                lin(f"{p}.{m}_inter.dense", f"{p}.{m}_inter.dense")
                lin(f"{p}.{m}_output.dense", f"{p}.{m}_output.dense")
                ln(f"{p}.{m}_output.LayerNorm", f"{p}.{m}_output.LayerNorm")
        lin("pooler.dense", "pooler.dense")
        self._loaded = True
        return self

    def eval(self):
        return self

    # ---- ops (each one call into the C ABI) ----------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _linear(self, x, name, act=L.VK_ACT_NONE, residual=None):
        w, b, N, Kp = self._lin[name]
        M, K = x.shape
        if K != Kp:                                     # box_fc: K = 4 -> one zero-padded K-tile (layout only)
            xp = torch.zeros((M, Kp), dtype=x.dtype, device=self.device)
            xp[:, :K] = x
            x = xp
        ldy = (N + 7) // 8 * 8
        y = torch.empty((M, ldy), dtype=self.tdt, device=self.device)
        L.call("vk_linear", x.data_ptr(), M, Kp, w.data_ptr(), b.data_ptr(), residual.data_ptr() if residual is not None else None,
               y.data_ptr(), N, ldy, act, self.dt, self.dt, self._stream())
        return y if ldy == N else y[:, :N]

    def _layernorm(self, x, name, out=None, scale=1.0, accumulate=False):
        g, b = self._ln[name]
        M, Cc = x.shape
        y = torch.empty((M, Cc), dtype=self.tdt, device=self.device) if out is None else out
        L.call("vk_layernorm", x.data_ptr(), x.stride(0), g.data_ptr(), b.data_ptr(), y.data_ptr(), y.stride(0), M, Cc, LN_EPS, scale,
               int(accumulate), self.dt, self._stream())
        return y

    def _attention(self, q, k, v, mask, B, Lq, Lk):
        H, heads = self.cfg["hidden_size"], self.cfg["num_attention_heads"]
        out = torch.empty((B * Lq, H), dtype=self.tdt, device=self.device)
        L.call("vk_attention", q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
               mask.data_ptr() if mask is not None else None, out.data_ptr(), H, B, heads, Lq, Lk, H // heads, self.dt, self._stream())
        return out

    # ---- layers (modeling_lxmert.py) -----------------------------------------------------------------------------
    def _self_att_block(self, p, x, mask, B, Lx):       # LxmertSelfAttentionLayer :298-316
        H = self.cfg["hidden_size"]
        qkv = self._linear(x, p + ".self.qkv")
        ctx = self._attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, B, Lx, Lx)
        t = self._linear(ctx, p + ".output.dense", residual=x)
        return self._layernorm(t, p + ".output.LayerNorm")

    def _cross_att_block(self, p, x, ctx_in, ctx_mask, B, Lx, Lc):   # LxmertCrossAttentionLayer :283-295
        H = self.cfg["hidden_size"]
        q = self._linear(x, p + ".att.q")
        kv = self._linear(ctx_in, p + ".att.kv")
        ctx = self._attention(q, kv[:, :H], kv[:, H:], ctx_mask, B, Lx, Lc)
        t = self._linear(ctx, p + ".output.dense", residual=x)
        return self._layernorm(t, p + ".output.LayerNorm")

    def _ffn(self, inter, outp, x):                     # LxmertIntermediate + LxmertOutput :319-342
        h = self._linear(x, inter + ".dense", act=L.VK_ACT_GELU)
        t = self._linear(h, outp + ".dense", residual=x)
        return self._layernorm(t, outp + ".LayerNorm")

    def _bert_layer(self, p, x, mask, B, Lx):           # LxmertLayer :345-358
        a = self._self_att_block(p + ".attention", x, mask, B, Lx)
        return self._ffn(p + ".intermediate", p + ".output", a)

    def capture(self, input_ids, visual_feats, visual_pos, attention_mask=None, visual_attention_mask=None, token_type_ids=None):
        """Capture one forward for these input SHAPES into a HIP graph (~430 launches become one graph launch: the
        small-batch forward is launch-bound).  Returns `replay(input_ids, visual_feats, visual_pos, ...)`, which copies the
        new inputs into the captured buffers, launches the graph and returns the (static) output tensors."""
        dev = self.device
        static = dict(ids=input_ids.to(dev).clone(), vf=visual_feats.to(dev).clone(), vp=visual_pos.to(dev).clone(),
                      am=None if attention_mask is None else attention_mask.to(dev).clone(),
                      vm=None if visual_attention_mask is None else visual_attention_mask.to(dev).clone(),
                      tt=None if token_type_ids is None else token_type_ids.to(dev).clone())
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                    # warm-up: one-time attribute / lazy-init calls must not be captured
            self._forward(static["ids"], static["vf"], static["vp"], static["am"], static["vm"], static["tt"])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = self._forward(static["ids"], static["vf"], static["vp"], static["am"], static["vm"], static["tt"])

        def replay(input_ids, visual_feats, visual_pos, attention_mask=None, visual_attention_mask=None, token_type_ids=None):
            for key, val in (("ids", input_ids), ("vf", visual_feats), ("vp", visual_pos), ("am", attention_mask),
                             ("vm", visual_attention_mask), ("tt", token_type_ids)):
                if static[key] is not None:
                    static[key].copy_(val)
            graph.replay()
            return outs
        return replay

    def __call__(self, input_ids, visual_feats, visual_pos, attention_mask=None, visual_attention_mask=None, token_type_ids=None):
        out = self._forward(input_ids, visual_feats, visual_pos, attention_mask, visual_attention_mask, token_type_ids)
        torch.cuda.synchronize(self.device)
        return out

    @torch.no_grad()
    def _forward(self, input_ids, visual_feats, visual_pos, attention_mask=None, visual_attention_mask=None, token_type_ids=None):
        if not self._loaded:
            raise RuntimeError("LxmertEncoder: load_state_dict() first")
        cfg, dev = self.cfg, self.device
        B, Lq = input_ids.shape
        V = visual_feats.shape[1]
        H = cfg["hidden_size"]
        ids = input_ids.to(dev, torch.int64).contiguous()
        tts = (torch.zeros_like(ids) if token_type_ids is None else token_type_ids.to(dev, torch.int64).contiguous())
        fmin = torch.finfo(torch.float32).min

        def ext(m):                                     # LxmertModel.forward :766-784
            return None if m is None else ((1.0 - m.to(dev, torch.float32)) * fmin).contiguous()
        lmask = ext(torch.ones((B, Lq), device=dev) if attention_mask is None else attention_mask)
        vmask = ext(visual_attention_mask)
        # embeddings :191-214
        lang = torch.empty((B * Lq, H), dtype=self.tdt, device=dev)
        g, b = self._ln["embeddings.LayerNorm"]
        L.call("vk_embed_layernorm", ids.data_ptr(), tts.data_ptr(), B, Lq, self._tab["word_embeddings"].data_ptr(),
               self._tab["position_embeddings"].data_ptr(), self._tab["token_type_embeddings"].data_ptr(), g.data_ptr(), b.data_ptr(),
               lang.data_ptr(), H, LN_EPS, self.dt, self._stream())
        # visual feature encoder :468-476: (LN(fc(feats)) + LN(fc(pos))) / 2
        vf = visual_feats.to(dev).reshape(B * V, -1).to(self.tdt).contiguous()
        vp = visual_pos.to(dev).reshape(B * V, -1).to(self.tdt).contiguous()
        visn = self._layernorm(self._linear(vf, "encoder.visn_fc.visn_fc"), "encoder.visn_fc.visn_layer_norm", scale=0.5)
        self._layernorm(self._linear(vp, "encoder.visn_fc.box_fc"), "encoder.visn_fc.box_layer_norm", out=visn, scale=0.5, accumulate=True)
        # encoder :521-546
        for i in range(cfg["l_layers"]):
            lang = self._bert_layer(f"encoder.layer.{i}", lang, lmask, B, Lq)
        for i in range(cfg["r_layers"]):
            visn = self._bert_layer(f"encoder.r_layers.{i}", visn, vmask, B, V)
        for i in range(cfg["x_layers"]):
            p = f"encoder.x_layers.{i}"
            # cross attention, both directions through the SAME module (:377-398)
            l_att = self._cross_att_block(p + ".visual_attention", lang, visn, vmask, B, Lq, V)
            v_att = self._cross_att_block(p + ".visual_attention", visn, lang, lmask, B, V, Lq)
            l_att = self._self_att_block(p + ".lang_self_att", l_att, lmask, B, Lq)
            v_att = self._self_att_block(p + ".visn_self_att", v_att, vmask, B, V)
            lang = self._ffn(p + ".lang_inter", p + ".lang_output", l_att)
            visn = self._ffn(p + ".visn_inter", p + ".visn_output", v_att)
        # pooler :566-572: tanh(dense(first token)); the first rows are picked by a stride-Lq 1x1 "convolution"
        w, bb, N, Kp = self._lin["pooler.dense"]
        pooled = torch.empty((B, H), dtype=self.tdt, device=dev)
        L.call("vk_conv2d", lang.data_ptr(), B, Lq, 1, H, w.data_ptr(), bb.data_ptr(), None, pooled.data_ptr(), H, H, 1, 1, Lq, 0, 1, 1,
               L.VK_ACT_TANH, self.dt, self.dt, self._stream())
        return lang.view(B, Lq, H), visn.view(B, V, H), pooled


class LxmertForQuestionAnswering:
    """`transformers.LxmertForQuestionAnswering` (modeling_lxmert.py :1123-1290, the VQA/GQA model of BASELINE config 5) on the
    HIP path: encoder + `LxmertVisualAnswerHead` :602-614 (Linear -> GELU -> LayerNorm -> Linear on the pooled output).
    `model(...)` returns `question_answering_score [B, num_qa_labels]` (storage dtype)."""

    def __init__(self, config, num_qa_labels, precision="bf16", device="cuda:0"):
        self.lxmert = LxmertEncoder(config, precision, device)
        self.num_qa_labels = int(num_qa_labels)

    def load_state_dict(self, sd, strict=True):
        sd = {k: (v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)) for k, v in sd.items()}
        want = dict(lxmert_qa_param_spec(self.lxmert.cfg, self.num_qa_labels))
        if strict and set(want) != set(sd):
            raise RuntimeError(f"LxmertForQuestionAnswering.load_state_dict: missing {sorted(set(want) - set(sd))[:4]} "
                               f"unexpected {sorted(set(sd) - set(want))[:4]}")
        enc = self.lxmert
        enc.load_state_dict({k[len("lxmert."):]: v for k, v in sd.items() if k.startswith("lxmert.")}, strict)
        enc._lin["answer.fc0"] = enc._pack(sd["answer_head.logit_fc.0.weight"], sd["answer_head.logit_fc.0.bias"])
        enc._lin["answer.fc3"] = enc._pack(sd["answer_head.logit_fc.3.weight"], sd["answer_head.logit_fc.3.bias"])
        enc._ln["answer.ln"] = (torch.from_numpy(sd["answer_head.logit_fc.2.weight"]).to(enc.device),
                                torch.from_numpy(sd["answer_head.logit_fc.2.bias"]).to(enc.device))
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def __call__(self, input_ids, visual_feats, visual_pos, attention_mask=None, visual_attention_mask=None, token_type_ids=None):
        enc = self.lxmert
        _, _, pooled = enc._forward(input_ids, visual_feats, visual_pos, attention_mask, visual_attention_mask, token_type_ids)
        h = enc._linear(pooled, "answer.fc0", act=L.VK_ACT_GELU)
        h = enc._layernorm(h, "answer.ln")
        score = enc._linear(h, "answer.fc3")
        torch.cuda.synchronize(enc.device)
        return score
