"""
Regenerate tests/golden/bert_tiny.npz from HF ``BertModel`` itself.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_encoder        (build container; needs transformers, CPU only)

The reference encodes with ``AutoModel.from_pretrained(...)`` + its own poolers
(models/retrievers/dense.py:16,37-47,64-75).  No checkpoint is available offline, so the pin is: a
seeded random-weight BertModel (fp32, CPU, eval mode) run through the REAL HF forward and — when
/root/reference is present — the REAL reference poolers, on a right-padded batch.  The fixture stores the
weights (fp16-rounded), the inputs and HF's outputs; tests check oracle/bert_oracle.py against it on CPU and
the HIP encoder against it on the GPU.
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
    from transformers import BertConfig, BertModel
    sd_np = bert_oracle.random_bert(CFG, seed=11)
    model = BertModel(BertConfig(**CFG, attn_implementation="eager"), add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert not [m for m in missing if "position_ids" not in m and "token_type_ids" not in m], missing
    assert not unexpected, unexpected
    ids, mask, types = bert_oracle.random_batch(CFG, batch=7, max_len=45, seed=12)
    with torch.no_grad():
        hidden = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                       token_type_ids=torch.from_numpy(types))[0]
    if ref_import.available():
        ref = ref_import.load()
        dense = ref.dense
        mean = dense.MeanPooler.pool(hidden, torch.from_numpy(mask))
        cls = dense.ClsPooler.pool(hidden)
        pooler_src = "reference models/retrievers/dense.py (imported unmodified)"
    else:  # same arithmetic, restated
        m = torch.from_numpy(mask)
        mean = hidden.masked_fill(~m[..., None].bool(), 0.).sum(1) / m.sum(1)[..., None]
        cls = hidden[:, 0]
        pooler_src = "restated poolers (/root/reference absent)"
    out = os.path.join(ROOT, "tests", "golden", "bert_tiny.npz")
    np.savez_compressed(
        out, cfg_keys=np.array(list(CFG.keys())), cfg_vals=np.array([str(v) for v in CFG.values()]),
        input_ids=ids, attention_mask=mask, token_type_ids=types,
        hf_hidden=hidden.numpy().astype(np.float32), ref_mean=mean.numpy().astype(np.float32),
        ref_cls=cls.numpy().astype(np.float32), pooler_src=np.array(pooler_src),
        **{"w::" + k: v.astype(np.float16) for k, v in sd_np.items()})
    print("wrote", out, os.path.getsize(out), "bytes;", pooler_src)


if __name__ == "__main__":
    main()
