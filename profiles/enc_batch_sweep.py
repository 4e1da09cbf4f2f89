#!/usr/bin/env python
"""Encoder forward pass (BERT-base, CLS pool) at several batch sizes: kernel time per passage and achieved TFLOP/s.
Question behind it: the 512-passage step moves ~2.8 GB of activations per layer through HBM (H alone is 421 MB); a smaller
step keeps them in the 256 MB Infinity Cache but quantises worse onto 256 CUs.

  python profiles/enc_batch_sweep.py [out.json] [batch sizes ...]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import BertEncoder, synth  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "enc_batch_sweep.json")
    sizes = [int(v) for v in sys.argv[2:]] or [64, 128, 192, 256, 384, 512, 768, 1024]
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = synth.random_bert(cfg, seed=31)
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
    rng = np.random.default_rng(6)
    lens_all = np.clip(np.rint(rng.normal(130, 30, size=max(sizes))), 16, 256).astype(np.int64)
    res = []
    for b in sizes:
        lens = lens_all[:b]
        T = int(lens.max())
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
        ids = rng.integers(1, cfg["vocab_size"], size=(b, T)).astype(np.int64) * mask
        kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        enc.encode_pooled(kw, "cls")
        best = 1e9
        for _ in range(5):
            enc.encode_pooled(kw, "cls")
            c = enc.counters()
            best = min(best, c["forward_ms"])
        row = {"batch": b, "packed_rows": int(c["packed_rows"]), "real_tokens": int(c["real_tokens"]), "forward_ms": best,
               "us_per_passage": best * 1e3 / b, "passages_per_s": b / (best * 1e-3), "tflops": c["flops"] / (best * 1e-3) / 1e12}
        res.append(row)
        print(json.dumps(row), flush=True)
    enc.close()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
