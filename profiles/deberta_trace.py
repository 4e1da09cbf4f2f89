"""DeBERTa-v3-large-shaped cross-encoder, N pairs (default 256) of ~180 tokens: three classify() calls — the command behind the kernel
trace of the rerank stage's launch size (profiles/r06_deberta256_kernel_stats.csv).  python profiles/deberta_trace.py [pairs] [bert]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import bench
n = int(argv[0]) if argv else 256
enc, cfg = bench.make_cross_encoder(not (len(argv) > 1 and argv[1] == "bert"), 0)
rng = np.random.default_rng(17)
lens = np.clip(np.rint(rng.normal(180, 40, size=n)), 32, 256).astype(np.int64)
T = int(lens.max())
mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
ids = rng.integers(1, cfg["vocab_size"], size=(n, T)).astype(np.int64) * mask
kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
enc.classify(kw)
ms = []
for _ in range(3):
    enc.classify(kw)
    ms.append(enc.counters()["forward_ms"])
c = enc.counters()
print(f"pairs {n}: forward ms {min(ms):.3f}; {n / min(ms) * 1e3:.0f} pairs/s; frac {c['flops'] / (min(ms) * 1e-3) / 2.5e15:.3f}")
