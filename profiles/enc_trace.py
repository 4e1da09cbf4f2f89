"""The encoder-only command behind the rocprofv3 kernel traces of the encoder (profiles/r05*_encoder_kernel_stats.csv):
bench.py's encoder batch (512 synthetic passages, lengths ~ clipped-Normal(130, 30), random-init BERT-base; `e5_large`: bert-large
shape, mean pooling) through BertEncoder.encode_pooled, one warm-up + N forward passes; prints the forward time by the library's own
HIP events.  Options of the encoder / the GEMM can be set from the command line: name=value ...
  python profiles/enc_trace.py [bert|e5_large|nomic] [steps] [option=value ...]"""
import os, statistics, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import BertEncoder, _lib, synth
arch = sys.argv[1] if len(sys.argv) > 1 else "bert"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
           max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
if arch == "e5_large":
    cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
if arch == "nomic":
    cfg.update(model_type="nomic_bert", hidden_act="silu", vocab_size=30528, max_position_embeddings=2048, rope_theta=1000.0)
if arch == "gte":
    cfg = dict(model_type="new", vocab_size=30528, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               hidden_act="gelu", max_position_embeddings=8192, type_vocab_size=0, layer_norm_type="layer_norm", layer_norm_eps=1e-12,
               position_embedding_type="rope", rope_theta=500000.0, rope_scaling={"type": "ntk", "factor": 2.0})
sd = synth.random_nomic(cfg, seed=33, scale=0.02) if arch == "nomic" else synth.random_new(cfg, seed=37, scale=0.02) if arch == "gte" else synth.random_bert(cfg, seed=31)
enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
for kv in sys.argv[3:]:
    name, val = kv.split("=")
    (_lib.set_option if name.startswith(("gemm_", "ln_small")) else enc.set_option)(name, int(val))
rng = np.random.default_rng(6)
batch = 512
lens = np.clip(np.rint(rng.normal(130, 30, size=batch)), 16, 256).astype(np.int64)
T = int(lens.max())
mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
ids = rng.integers(1, cfg["vocab_size"], size=(batch, T)).astype(np.int64) * mask
kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
pool = "cls" if arch in ("bert", "gte") else "mean"
enc.encode_pooled(kw, pool)
ms = []
for _ in range(steps):
    enc.encode_pooled(kw, pool)
    ms.append(enc.counters()["forward_ms"])
c = enc.counters()
med = statistics.median(ms)
print(f"{arch}: forward ms median {med:.3f} min {min(ms):.3f} over {steps} steps; {batch / med * 1e3:.0f} passages/s; "
      f"{c['flops'] / (med * 1e-3) / 1e12:.1f} TFLOP/s = {c['flops'] / (med * 1e-3) / 1e12 / 2500:.4f} of MFMA peak; packed rows {int(c['packed_rows'])}")
