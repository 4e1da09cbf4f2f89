"""
Regenerate tests/golden/splade_tiny.npz from HF ``BertForMaskedLM`` and the reference's ``Splade.__call__``.
TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_splade        (build container; needs transformers + /root/reference, CPU only)

The reference encodes SPLADE vectors with ``AutoModelForMaskedLM.from_pretrained(...)`` and pools with
``max_t log(1 + relu(logits)) * mask`` (models/retrievers/splade.py:17,34-47).  No checkpoint is available
offline, so the pin is a seeded random-weight BertForMaskedLM (fp32, CPU, eval mode) run through the REAL HF
forward and the REAL, unmodified ``Splade.__call__`` (constructed without __init__, which would download a
checkpoint; its hard-coded ``.to('cuda')`` is neutralised by oracle/ref_import.py's Tensor.to patch).  Two
heads are stored: an untied decoder and a tied one (decoder = word embeddings, HF's default).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bert_oracle, ref_import  # noqa: E402

CFG = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
           max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")


def hf_model(sd_np, tied):
    from transformers import BertConfig, BertForMaskedLM
    model = BertForMaskedLM(BertConfig(**CFG, attn_implementation="eager", tie_word_embeddings=tied)).eval()
    sd = {"bert." + k if not k.startswith("cls.") else k: torch.from_numpy(v) for k, v in sd_np.items()}
    sd["cls.predictions.bias"] = sd["cls.predictions.decoder.bias"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m and "token_type_ids" not in m
               and not (tied and m == "cls.predictions.decoder.weight")]
    assert not missing, missing
    assert not unexpected, unexpected
    if tied:
        assert model.cls.predictions.decoder.weight.data_ptr() == model.bert.embeddings.word_embeddings.weight.data_ptr()
    return model


def main():
    assert ref_import.available(), "needs /root/reference"
    ref = ref_import.load()
    out = {}
    ids, mask, types = bert_oracle.random_batch(CFG, batch=9, max_len=41, seed=22)
    for tag, tied in (("untied", False), ("tied", True)):
        sd_np = bert_oracle.random_bert(CFG, seed=21)
        bert_oracle.random_mlm_head(CFG, seed=23, tied=tied, sd=sd_np)
        model = hf_model(sd_np, tied)
        sp = object.__new__(ref.splade.Splade)  # skip __init__ (it downloads a checkpoint)
        sp.model = model
        sp.query_encoder = model
        kwargs = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
                  "token_type_ids": torch.from_numpy(types)}
        with torch.no_grad():
            emb = sp("doc", kwargs)["embedding"]
            logits = model(**kwargs).logits
        out[f"ref_emb_{tag}"] = emb.numpy().astype(np.float32)
        out[f"hf_logits_{tag}"] = logits[:2].numpy().astype(np.float32)  # first two sequences only (fixture size)
        for k, v in sd_np.items():
            if k.startswith("cls."):
                out[f"w_{tag}::" + k] = v.astype(np.float16)
        if not tied:
            for k, v in sd_np.items():
                if not k.startswith("cls."):
                    out["w::" + k] = v.astype(np.float16)
    path = os.path.join(ROOT, "tests", "golden", "splade_tiny.npz")
    np.savez_compressed(path, cfg_keys=np.array(list(CFG.keys())), cfg_vals=np.array([str(v) for v in CFG.values()]),
                        input_ids=ids, attention_mask=mask, token_type_ids=types, **out)
    print("wrote", path, os.path.getsize(path), "bytes; density",
          {t: float((out[f'ref_emb_{t}'] > 0).mean()) for t in ("untied", "tied")})


if __name__ == "__main__":
    main()
