"""TEST INFRASTRUCTURE ONLY -- CPU restatement of transformers' LxmertModel forward.

The reference (`vltk/legacy/legacy_train.py:30-39`) feeds the extractor's features to `transformers` LXMERT; that
library IS importable in this image (transformers 5.15), so this restatement is pinned by golden vectors generated
from `transformers.LxmertModel` itself (`tools/gen_golden_lxmert.py` -> `tests/golden/lxmert_small.npz`).
Follows transformers/models/lxmert/modeling_lxmert.py: LxmertEmbeddings :191-214, LxmertAttention :238-266,
LxmertAttentionOutput :276-280, LxmertIntermediate/Output :325-342, LxmertXLayer :417-449,
LxmertVisualFeatureEncoder :468-476, LxmertEncoder :498-557, LxmertPooler :566-572, LxmertModel :691-824.

`emulate="bf16"|"fp16"` restates the rounding points of the HIP path (vltk_amd/lxmert.py): every stored tensor is
rounded to the storage type; GEMMs accumulate in fp32 and add bias (+ residual) before the one rounding; LayerNorm,
soft-max and the attention sums are fp32.  Only tests/, smoke() and bench's cpu_baseline may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-12


class LxmertOracle:
    def __init__(self, cfg, state_dict, emulate=None):
        assert emulate in (None, "bf16", "fp16")
        self.cfg = cfg
        self.sd = {k: (torch.from_numpy(np.asarray(v)) if not isinstance(v, torch.Tensor) else v).float() for k, v in state_dict.items()}
        self.emulate = emulate
        self._t = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16}[emulate]

    def q(self, t):                      # storage rounding of the emulated mode
        return t if self._t is None else t.to(self._t).float()

    def linear(self, x, p, act=None, residual=None):
        y = F.linear(x, self.q(self.sd[p + ".weight"])) + self.sd[p + ".bias"]
        if residual is not None:
            y = y + residual
        if act == "gelu":
            y = F.gelu(y)                # erf form == transformers ACT2FN["gelu"]
        elif act == "tanh":
            y = torch.tanh(y)
        return self.q(y)

    def ln(self, x, p, scale=1.0):
        return F.layer_norm(x, (x.shape[-1],), self.sd[p + ".weight"], self.sd[p + ".bias"], LN_EPS) * scale

    def attention(self, x, ctx, mask, p):            # LxmertAttention.forward :238-266
        B, Lq, H = x.shape
        heads = self.cfg["num_attention_heads"]
        d = H // heads
        qh = self.linear(x, p + ".query").view(B, Lq, heads, d).transpose(1, 2)
        kh = self.linear(ctx, p + ".key").view(B, -1, heads, d).transpose(1, 2)
        vh = self.linear(ctx, p + ".value").view(B, -1, heads, d).transpose(1, 2)
        s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(d)
        if mask is not None:
            s = s + mask
        # the HIP path's 16-bit attention (csrc/lxmert.hip attention_mfma_kernel) keeps scores and soft-max in fp32 and rounds
        # the probabilities to the storage type before the second product, as the reference's own bf16 matmul sees them
        o = torch.matmul(self.q(F.softmax(s, dim=-1)), vh)
        return self.q(o.permute(0, 2, 1, 3).reshape(B, Lq, H))

    def att_block(self, x, ctx, mask, p_att, p_out):  # attention + LxmertAttentionOutput :276-280
        a = self.attention(x, ctx, mask, p_att)
        return self.q(self.ln(self.linear(a, p_out + ".dense", residual=x), p_out + ".LayerNorm"))

    def ffn(self, x, p_inter, p_out):
        h = self.linear(x, p_inter + ".dense", act="gelu")
        return self.q(self.ln(self.linear(h, p_out + ".dense", residual=x), p_out + ".LayerNorm"))

    def bert_layer(self, x, mask, p):                 # LxmertLayer :352-358
        a = self.att_block(x, x, mask, p + ".attention.self", p + ".attention.output")
        return self.ffn(a, p + ".intermediate", p + ".output")

    def forward(self, input_ids, visual_feats, visual_pos, attention_mask=None, visual_attention_mask=None, token_type_ids=None,
                return_stages=False):
        cfg, sd = self.cfg, self.sd
        ids = torch.as_tensor(input_ids).long()
        B, Lq = ids.shape
        tts = torch.zeros_like(ids) if token_type_ids is None else torch.as_tensor(token_type_ids).long()
        fmin = torch.finfo(torch.float32).min

        def ext(m):                                   # LxmertModel.forward :766-784
            return None if m is None else ((1.0 - torch.as_tensor(m).float()) * fmin)[:, None, None, :]
        lmask = ext(torch.ones((B, Lq)) if attention_mask is None else attention_mask)
        vmask = ext(visual_attention_mask)
        st = {}
        # LxmertEmbeddings :191-214
        e = (self.q(sd["embeddings.word_embeddings.weight"])[ids] + self.q(sd["embeddings.position_embeddings.weight"])[torch.arange(Lq)][None] +
             self.q(sd["embeddings.token_type_embeddings.weight"])[tts])
        lang = self.q(self.ln(e, "embeddings.LayerNorm"))
        st["embeddings"] = lang
        # LxmertVisualFeatureEncoder :468-476
        vf, vp = self.q(torch.as_tensor(visual_feats).float()), self.q(torch.as_tensor(visual_pos).float())
        x = self.q(self.ln(self.linear(vf, "encoder.visn_fc.visn_fc"), "encoder.visn_fc.visn_layer_norm", 0.5))
        y = self.ln(self.linear(vp, "encoder.visn_fc.box_fc"), "encoder.visn_fc.box_layer_norm", 0.5)
        visn = self.q(x + y)
        st["visual_embeddings"] = visn
        for i in range(cfg["l_layers"]):
            lang = self.bert_layer(lang, lmask, f"encoder.layer.{i}")
        st["lang_after_l"] = lang
        for i in range(cfg["r_layers"]):
            visn = self.bert_layer(visn, vmask, f"encoder.r_layers.{i}")
        st["visn_after_r"] = visn
        for i in range(cfg["x_layers"]):
            p = f"encoder.x_layers.{i}"
            l_att = self.att_block(lang, visn, vmask, p + ".visual_attention.att", p + ".visual_attention.output")
            v_att = self.att_block(visn, lang, lmask, p + ".visual_attention.att", p + ".visual_attention.output")
            l_att = self.att_block(l_att, l_att, lmask, p + ".lang_self_att.self", p + ".lang_self_att.output")
            v_att = self.att_block(v_att, v_att, vmask, p + ".visn_self_att.self", p + ".visn_self_att.output")
            lang = self.ffn(l_att, p + ".lang_inter", p + ".lang_output")
            visn = self.ffn(v_att, p + ".visn_inter", p + ".visn_output")
        pooled = self.linear(lang[:, 0], "pooler.dense", act="tanh")
        out = (lang, visn, pooled)
        return (out, st) if return_stages else out


    def qa_forward(self, *args, **kw):
        """LxmertForQuestionAnswering.forward :1255-1270 (state dict with the `lxmert.` prefix and `answer_head.logit_fc.*`):
        answer_head(pooled) = Linear -> GELU -> LayerNorm -> Linear (:602-614)."""
        full = self.sd
        self.sd = {k[len("lxmert."):]: v for k, v in full.items() if k.startswith("lxmert.")}
        try:
            _, _, pooled = self.forward(*args, **kw)
        finally:
            self.sd = full
        h = self.linear(pooled, "answer_head.logit_fc.0", act="gelu")
        h = self.q(self.ln(h, "answer_head.logit_fc.2"))
        return self.linear(h, "answer_head.logit_fc.3")
