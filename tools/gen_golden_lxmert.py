#!/usr/bin/env python3
"""Golden vectors for N3 from `transformers.LxmertModel` itself (runs in the build container only).

transformers is a dependency of the reference (requirements.txt) and the consumer of the extractor's output
(vltk/legacy/legacy_train.py:30-39).  No checkpoint can be fetched offline, so the model is built from a small
config with build-owned seeded weights (vltk_amd.lxmert.make_lxmert_state_dict, regenerated from the seed by the
tests, never stored) and run on seeded inputs; inputs and outputs go to tests/golden/lxmert_small.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from vltk_amd.lxmert import lxmert_config, lxmert_param_spec, make_lxmert_state_dict  # noqa: E402

SMALL = dict(vocab_size=500, hidden_size=128, num_attention_heads=4, intermediate_size=256, l_layers=2, x_layers=2, r_layers=2,
             max_position_embeddings=32, type_vocab_size=2, visual_feat_dim=2048, visual_pos_dim=4)


def main():
    from transformers import LxmertConfig, LxmertModel
    cfg = lxmert_config(**SMALL)
    hf = LxmertModel(LxmertConfig(**cfg)).eval()
    ref_keys = [(k, tuple(v.shape)) for k, v in hf.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in lxmert_param_spec(cfg)], "param spec differs from transformers' state_dict"
    seed = 2024
    sd = make_lxmert_state_dict(cfg, seed)
    hf.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    g = np.random.Generator(np.random.PCG64(seed))
    B, Lq, V = 3, 11, 36
    ids = g.integers(1, cfg["vocab_size"], (B, Lq))
    tts = g.integers(0, 2, (B, Lq))
    amask = np.ones((B, Lq), np.float32)
    amask[1, 7:] = 0
    amask[2, 4:] = 0
    vmask = np.ones((B, V), np.float32)
    vmask[0, 30:] = 0
    feats = np.maximum(g.standard_normal((B, V, cfg["visual_feat_dim"])), 0).astype(np.float32) * 2.0     # ReLU'd RoI features
    x1y1 = g.uniform(0, 0.6, (B, V, 2))
    pos = np.concatenate([x1y1, x1y1 + g.uniform(0.05, 0.4, (B, V, 2))], -1).astype(np.float32)        # normalised boxes
    out = {"seed": np.asarray(seed), "input_ids": ids, "token_type_ids": tts, "attention_mask": amask, "visual_attention_mask": vmask,
           "visual_feats_seed": np.asarray(seed), "visual_pos": pos}
    out.update({f"cfg/{k}": np.asarray(v) for k, v in cfg.items()})
    with torch.no_grad():
        for tag, kw in (("masked", dict(attention_mask=torch.from_numpy(amask), visual_attention_mask=torch.from_numpy(vmask),
                                        token_type_ids=torch.from_numpy(tts))),
                        ("plain", dict())):
            o = hf(input_ids=torch.from_numpy(ids), visual_feats=torch.from_numpy(feats), visual_pos=torch.from_numpy(pos),
                   output_hidden_states=True, **kw)
            out[f"{tag}/language_output"] = o.language_output.numpy()
            out[f"{tag}/vision_output"] = o.vision_output.numpy()
            out[f"{tag}/pooled_output"] = o.pooled_output.numpy()
            out[f"{tag}/lang_after_l"] = o.language_hidden_states[cfg["l_layers"] - 1].numpy()
            out[f"{tag}/visn_after_r"] = o.vision_hidden_states[cfg["r_layers"] - 1].numpy()
            print(tag, "lang |max| %.3f visn |max| %.3f pooled |max| %.3f" % (np.abs(out[f"{tag}/language_output"]).max(),
                                                                             np.abs(out[f"{tag}/vision_output"]).max(),
                                                                             np.abs(out[f"{tag}/pooled_output"]).max()))
    # LxmertForQuestionAnswering (the VQA / GQA model the reference's training script instantiates): encoder + answer head
    from transformers import LxmertForQuestionAnswering
    from vltk_amd.lxmert import lxmert_qa_param_spec, make_lxmert_qa_state_dict
    NQA = 40
    qa = LxmertForQuestionAnswering(LxmertConfig(num_qa_labels=NQA, **cfg)).eval()
    assert [(k, tuple(v.shape)) for k, v in qa.state_dict().items()] == [(k, tuple(s_)) for k, s_ in lxmert_qa_param_spec(cfg, NQA)]
    qa.load_state_dict({k: torch.from_numpy(v) for k, v in make_lxmert_qa_state_dict(cfg, NQA, seed).items()}, strict=True)
    with torch.no_grad():
        o = qa(input_ids=torch.from_numpy(ids), visual_feats=torch.from_numpy(feats), visual_pos=torch.from_numpy(pos),
               attention_mask=torch.from_numpy(amask), visual_attention_mask=torch.from_numpy(vmask), token_type_ids=torch.from_numpy(tts))
    out["qa/num_labels"] = np.asarray(NQA)
    out["qa/question_answering_score"] = o.question_answering_score.numpy()
    print("qa score |max| %.3f argmax %s" % (np.abs(out["qa/question_answering_score"]).max(), out["qa/question_answering_score"].argmax(-1)))
    # the visual features are regenerated from the seed by the tests (3 x 36 x 2048 floats would dominate the file)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lxmert_small.npz"), **out)
    print("lxmert_small.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
