"""Where a rerank launch's wall time goes beyond its kernels: BertEncoder.classify on N pairs (BERT-large shape), wall per call vs the
library's HIP-event forward_ms.   python profiles/rerank_gap.py"""
import os, sys, time, statistics
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import BertEncoder, synth
cfg = dict(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
           max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
sd = synth.random_bert(cfg, seed=61)
synth.random_cls_head(cfg, seed=62, num_labels=1, sd=sd)
enc = BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)
rng = np.random.default_rng(1)
for n in (32, 256, 512, 1024):
    lens = np.clip(np.rint(rng.normal(180, 40, size=n)), 32, 256).astype(np.int64)
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(n, T)).astype(np.int64) * mask
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.zeros_like(torch.from_numpy(ids))}
    enc.classify(kw)
    walls, kern = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        out = enc.classify(kw)
        walls.append((time.perf_counter() - t0) * 1e3)
        kern.append(enc.counters()["forward_ms"])
    c = enc.counters()
    print(f"pairs {n}: wall {statistics.median(walls):.2f} ms, kernels {statistics.median(kern):.2f} ms, gap {statistics.median(walls) - statistics.median(kern):.2f} ms, "
          f"{n / statistics.median(walls) * 1e3:.0f} pairs/s wall, frac kernels {c['flops'] / (statistics.median(kern) * 1e-3) / 2.5e15:.3f}", flush=True)
