"""
Regenerate tests/golden/rerank_tiny.npz from HF ``BertForSequenceClassification`` and the reference's
``CrossEncoder.__call__`` / ``Rerank.sort_by_score_indexes``.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_rerank        (build container; needs transformers + /root/reference, CPU only)

No cross-encoder checkpoint is available offline, so the pin is a seeded random-weight BertForSequenceClassification
(fp32, CPU, eval mode) driven through the REAL, unmodified ``CrossEncoder.__call__`` (constructed without __init__,
which downloads a checkpoint; ``.to('cuda')`` neutralised by oracle/ref_import.py) on a batch padded to max_length
like the reference's collate_fn, and the REAL ``Rerank.sort_by_score_indexes`` on those scores.
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


def main():
    assert ref_import.available(), "needs /root/reference"
    ref = ref_import.load()
    from transformers import BertConfig, BertForSequenceClassification
    from transformers.tokenization_utils_base import BatchEncoding
    out = {}
    for labels in (1, 3):
        sd_np = bert_oracle.random_bert(CFG, seed=41)
        bert_oracle.random_cls_head(CFG, seed=42 + labels, num_labels=labels, sd=sd_np)
        model = BertForSequenceClassification(BertConfig(**CFG, attn_implementation="eager", num_labels=labels)).eval()
        sd = {k if k.startswith("classifier.") else "bert." + k: torch.from_numpy(v) for k, v in sd_np.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not [m for m in missing if "position_ids" not in m and "token_type_ids" not in m], missing
        assert not unexpected, unexpected
        ids, mask, types = bert_oracle.random_batch(CFG, batch=12, max_len=40, seed=44)
        # the reference pads to max_length (crossencoder.py:30): widen the batch to 48 columns of padding
        pad = 48 - ids.shape[1]
        ids, mask, types = (np.pad(a, ((0, 0), (0, pad))) for a in (ids, mask, types))
        ce = object.__new__(ref.crossencoder.CrossEncoder)
        ce.model = model
        enc = BatchEncoding({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
                             "token_type_ids": torch.from_numpy(types)})
        with torch.no_grad():
            score = ce(enc)["score"]
        out[f"ref_score_{labels}"] = score.numpy().astype(np.float32)
        for k, v in sd_np.items():
            if k.startswith("classifier.") or k.startswith("pooler."):
                out[f"w_{labels}::" + k] = v.astype(np.float16)
        if labels == 1:
            for k, v in sd_np.items():
                if not (k.startswith("classifier.") or k.startswith("pooler.")):
                    out["w::" + k] = v.astype(np.float16)
            out.update(input_ids=ids, attention_mask=mask, token_type_ids=types)
            # the reference's own per-query sort on these scores (ties included: two documents share a score)
            rr = object.__new__(ref.rerank.Rerank)
            flat = score.ravel().clone()
            flat[5] = flat[4]
            q_ids = [f"q{i // 4}" for i in range(12)]
            d_ids = [f"d{i}" for i in range(12)]
            qs, ds, ss = rr.sort_by_score_indexes(flat, q_ids, d_ids)
            out["sort_scores_in"] = flat.numpy()
            out["sort_q"] = np.array(qs)
            out["sort_d"] = np.array(ds)
            out["sort_s"] = np.stack([s.numpy() for s in ss])
    path = os.path.join(ROOT, "tests", "golden", "rerank_tiny.npz")
    np.savez_compressed(path, cfg_keys=np.array(list(CFG.keys())), cfg_vals=np.array([str(v) for v in CFG.values()]), **out)
    print("wrote", path, os.path.getsize(path), "bytes; scores", out["ref_score_1"].ravel()[:4])


if __name__ == "__main__":
    main()
