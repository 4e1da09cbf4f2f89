"""
Regenerate tests/golden/deberta_tiny.npz from HF ``DebertaV2ForSequenceClassification`` driven through the reference's
``CrossEncoder.__call__``.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_deberta        (build container; needs transformers + /root/reference, CPU only)

The reference's default reranker (config/reranker/debertav3.yaml:3, naver/trecdl22-crossencoder-debertav3) is not available
offline, so the pin is a seeded random-weight DebertaV2ForSequenceClassification with deberta-v3's architecture switches
(relative_attention, p2c|c2p, share_att_key, norm_rel_ebd = layer_norm, position_biased_input = False, type_vocab_size 0),
fp32, CPU, eval mode, driven through the REAL, unmodified ``CrossEncoder.__call__`` (constructed without __init__, which
downloads a checkpoint) on a right-padded batch.  16 position buckets and sequences of up to 60 tokens: both the linear and
the logarithmic part of the bucket function are exercised.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deberta_oracle, ref_import  # noqa: E402

CFG = dict(vocab_size=500, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
           max_position_embeddings=64, type_vocab_size=0, layer_norm_eps=1e-7, hidden_act="gelu", relative_attention=True,
           position_buckets=16, norm_rel_ebd="layer_norm", share_att_key=True, pos_att_type="p2c|c2p",
           position_biased_input=False, max_relative_positions=-1, pooler_hidden_size=128, pooler_hidden_act="gelu",
           pooler_dropout=0.0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=0)


def hf_model(sd_np, labels):
    from transformers import DebertaV2Config, DebertaV2ForSequenceClassification
    model = DebertaV2ForSequenceClassification(DebertaV2Config(**CFG, num_labels=labels)).eval()
    sd = {(k if k.startswith(("classifier.", "pooler.")) else "deberta." + k): torch.from_numpy(v) for k, v in sd_np.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "position_ids" not in m], missing
    assert not unexpected, unexpected
    return model


def batch(seed, n=10, max_len=60):
    rng = np.random.default_rng(seed)
    lens = rng.integers(3, max_len + 1, size=n)
    lens[0], lens[1] = max_len, 3
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, CFG["vocab_size"], size=(n, T)).astype(np.int64) * mask
    return ids, mask


def main():
    assert ref_import.available(), "needs /root/reference"
    ref = ref_import.load()
    from transformers.tokenization_utils_base import BatchEncoding
    out = {}
    ids, mask = batch(71)
    body = deberta_oracle.random_deberta(CFG, seed=71, num_labels=1)
    for labels in (1, 3):
        sd_np = dict(body)  # one encoder body, two heads
        if labels != 1:
            head = deberta_oracle.random_deberta(CFG, seed=70 + labels, num_labels=labels)
            sd_np.update({k: v for k, v in head.items() if k.startswith(("classifier.", "pooler."))})
        model = hf_model(sd_np, labels)
        ce = object.__new__(ref.crossencoder.CrossEncoder)
        ce.model = model
        enc = BatchEncoding({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)})
        with torch.no_grad():
            score = ce(enc)["score"]
            hidden = model.deberta(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))[0]
        out[f"ref_score_{labels}"] = score.numpy().astype(np.float32)
        want = deberta_oracle.cross_encode(sd_np, CFG, ids, mask)
        print(f"labels={labels}: max |oracle - HF| =", float(np.abs(want - score.numpy()).max()), "scores", score.numpy().ravel()[:3])
        if labels == 1:
            out["ref_hidden"] = hidden.numpy().astype(np.float32)
            for k, v in sd_np.items():
                out["w::" + k] = v.astype(np.float16)
            out.update(input_ids=ids, attention_mask=mask)
        else:
            for k, v in sd_np.items():
                if k.startswith(("classifier.", "pooler.")):
                    out[f"w_{labels}::" + k] = v.astype(np.float16)
    path = os.path.join(ROOT, "tests", "golden", "deberta_tiny.npz")
    np.savez_compressed(path, cfg_keys=np.array(list(CFG.keys())), cfg_vals=np.array([str(v) for v in CFG.values()]), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
