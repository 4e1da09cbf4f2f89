"""A/B of an encoder option inside one process: python profiles/enc_ab_option.py <option> <v0> <v1> [batch]
-> median forward ms (kernel time between the library's events) of bench.py's encoder leg for each value, alternating.
ENC_ARCH=nomic: the NomicBert leg's model (rotary positions, gated SiLU feed-forward), e.g. `ffn_fused 0 1`."""
import os, sys, statistics
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import BertEncoder, synth
opt, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 512
cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
           max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
nomic = os.environ.get("ENC_ARCH", "bert") == "nomic"
if nomic:
    cfg.update(model_type="nomic_bert", hidden_act="silu", vocab_size=30528, max_position_embeddings=2048, rope_theta=1000.0)
sd = synth.random_nomic(cfg, seed=33, scale=0.02) if nomic else synth.random_bert(cfg, seed=31)
enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
rng = np.random.default_rng(6)
lens = np.clip(np.rint(rng.normal(130, 30, size=batch)), 16, 256).astype(np.int64)
T = int(lens.max())
mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
ids = rng.integers(1, cfg["vocab_size"], size=(batch, T)).astype(np.int64) * mask
kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
res, outs = {v0: [], v1: []}, {}
for r in range(13):
    for v in (v0, v1):
        if opt.startswith("gemm_"):  # process-wide GEMM options (bh_set_option)
            from bergen_amd import _lib
            _lib.set_option(opt, v)
        else:
            enc.set_option(opt, v)
        out = enc.encode_pooled(kw, "mean" if nomic else "cls")
        if r:
            res[v].append(enc.counters()["forward_ms"])
        outs[v] = out.float().cpu()
same = bool(torch.equal(outs[v0], outs[v1]))
for v in (v0, v1):
    print(f"{opt}={v}: forward ms median {statistics.median(res[v]):.3f} min {min(res[v]):.3f}  ({batch / statistics.median(res[v]) * 1e3:.0f} passages/s)")
print("identical embeddings:", same, " max |difference|:", float((outs[v0] - outs[v1]).abs().max()))
