"""Attention bucket threshold sweep (encoder option attn_short_len): forward time of the bench batch per threshold, and a
bit-equality check of the embeddings against the default (the threshold only changes which workgroup shape runs a sequence).
Run on the GPU box: python profiles/attn_bucket.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bergen_amd import BertEncoder, synth  # noqa: E402

cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
           max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
sd = synth.random_bert(cfg, seed=31)
enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
rng = np.random.default_rng(6)
B = 512
lens = np.clip(np.rint(rng.normal(130, 30, size=B)), 16, 256).astype(np.int64)
T = int(lens.max())
mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
ids = rng.integers(1, cfg["vocab_size"], size=(B, T)).astype(np.int64) * mask
kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
base = None
out = []
for thr in (128, 96, 160, 192, 256, 128):
    enc.set_option("attn_short_len", thr)
    e = enc.encode_pooled(kw, "cls")
    ms = []
    for _ in range(5):
        e = enc.encode_pooled(kw, "cls")
        ms.append(enc.counters()["forward_ms"])
    if base is None:
        base = e.clone()
    out.append({"attn_short_len": thr, "forward_ms_min": min(ms), "forward_ms_med": sorted(ms)[2],
                "same_bits_as_default": bool(torch.equal(e.view(torch.int16), base.view(torch.int16)))})
    print(out[-1], file=sys.stderr, flush=True)
print(json.dumps(out))
