"""Where the fused rotary (GEMM epilogue) and the standalone kernel differ: hidden states of a 1-layer NomicBert-shaped encoder, both ways."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from bergen_amd import BertEncoder, _lib, synth
cfg = dict(vocab_size=3000, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, max_position_embeddings=512,
           type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="silu", rope_theta=1000.0, model_type="nomic_bert")
sd = synth.random_nomic(cfg, seed=3, scale=0.03)
enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
rng = np.random.default_rng(7)
B, T = 96, 200
lens = rng.integers(30, T + 1, size=B)
mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
ids = rng.integers(1, cfg["vocab_size"], size=(B, T)).astype(np.int64) * mask
kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
enc.set_option("micro_batches", 1)
outs = {}
for v in (1, 0):
    _lib.set_option("gemm_rotary_fused", v)
    outs[v] = enc(**kw)[0].float().cpu().numpy()
d = outs[1] != outs[0]
print("packed rows", enc.counters()["packed_rows"], "mismatching elements", int(d.sum()), "of", d.size, "max abs diff", float(np.abs(outs[1] - outs[0]).max()))
if d.any():
    b, t, c = np.nonzero(d)
    print("sequences with a mismatch", len(set(b.tolist())), "of", B, "; token positions (first 20)", sorted(set(t.tolist()))[:20], "; columns (first 20)", sorted(set(c.tolist()))[:20])
    print("mismatches per token position histogram (first 12):", np.bincount(t)[:12].tolist())
