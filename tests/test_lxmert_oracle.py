"""CPU: the LXMERT oracle (oracle/lxmert_oracle.py) against vectors produced by transformers.LxmertModel itself, and the
build's parameter spec against that class's state_dict."""
import os

import numpy as np
import pytest
import torch

from oracle.lxmert_oracle import LxmertOracle
from vltk_amd.lxmert import lxmert_config, lxmert_param_spec

from lxmert_util import case_kwargs, golden_inputs


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "lxmert_small.npz"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


@pytest.mark.parametrize("tag", ["masked", "plain"])
def test_oracle_vs_transformers_golden(g, tag):
    cfg, sd, feats = golden_inputs(g)
    (lang, visn, pooled), st = LxmertOracle(cfg, sd).forward(g["input_ids"], feats, g["visual_pos"], return_stages=True,
                                                             **case_kwargs(g, tag))
    assert rel(st["lang_after_l"], g[f"{tag}/lang_after_l"]) <= 1e-5
    assert rel(st["visn_after_r"], g[f"{tag}/visn_after_r"]) <= 1e-5
    assert rel(lang, g[f"{tag}/language_output"]) <= 1e-5
    assert rel(visn, g[f"{tag}/vision_output"]) <= 1e-5
    assert rel(pooled, g[f"{tag}/pooled_output"]) <= 1e-5


def test_bf16_emulation_stays_near_fp32(g):
    cfg, sd, feats = golden_inputs(g)
    lang, visn, pooled = LxmertOracle(cfg, sd, emulate="bf16").forward(g["input_ids"], feats, g["visual_pos"], **case_kwargs(g, "masked"))
    assert rel(lang, g["masked/language_output"]) <= 6e-2 and rel(visn, g["masked/vision_output"]) <= 6e-2
    assert rel(pooled, g["masked/pooled_output"]) <= 6e-2


def test_param_spec_matches_transformers():
    tr = pytest.importorskip("transformers")
    for kw in (dict(vocab_size=100, hidden_size=64, num_attention_heads=2, intermediate_size=128, l_layers=1, x_layers=1, r_layers=1,
                    max_position_embeddings=16), {}):
        cfg = lxmert_config(**kw)
        with torch.device("meta"):
            m = tr.LxmertModel(tr.LxmertConfig(**cfg))
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in lxmert_param_spec(cfg)]


def test_qa_oracle_vs_transformers_golden(g):
    """LxmertForQuestionAnswering (encoder + answer head) vs the vectors transformers produced."""
    from vltk_amd.lxmert import make_lxmert_qa_state_dict
    cfg, _, feats = golden_inputs(g)
    sd = make_lxmert_qa_state_dict(cfg, int(g["qa/num_labels"]), int(g["seed"]))
    score = LxmertOracle(cfg, sd).qa_forward(g["input_ids"], feats, g["visual_pos"], **case_kwargs(g, "masked"))
    assert rel(score, g["qa/question_answering_score"]) <= 1e-5
