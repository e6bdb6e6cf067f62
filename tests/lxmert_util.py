"""Rebuild the inputs of tests/golden/lxmert_small.npz from its seed (tools/gen_golden_lxmert.py draws them in this order)."""
import numpy as np

from vltk_amd.lxmert import lxmert_config, make_lxmert_state_dict


def golden_inputs(g):
    cfg = lxmert_config(**{k[4:]: int(g[k]) for k in g.files if k.startswith("cfg/")})
    seed = int(g["seed"])
    sd = make_lxmert_state_dict(cfg, seed)
    r = np.random.Generator(np.random.PCG64(seed))
    B, Lq = g["input_ids"].shape
    V = g["visual_pos"].shape[1]
    ids = r.integers(1, cfg["vocab_size"], (B, Lq))
    assert (ids == g["input_ids"]).all()
    r.integers(0, 2, (B, Lq))
    feats = np.maximum(r.standard_normal((B, V, cfg["visual_feat_dim"])), 0).astype(np.float32) * 2.0
    return cfg, sd, feats


def case_kwargs(g, tag):
    if tag == "plain":
        return {}
    return dict(attention_mask=g["attention_mask"], visual_attention_mask=g["visual_attention_mask"], token_type_ids=g["token_type_ids"])
