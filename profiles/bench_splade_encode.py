"""SPLADE encode micro-benchmark (GPU box): BERT-base + masked-LM head (30522 terms) + max pooling on 512 synthetic
passages; compares pool 3 (encode_splade) with pool 0 (CLS) on the same batch, so the head's cost is the difference.
Usage: python profiles/bench_splade_encode.py [batch] [steps] [bias_mean]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bergen_amd import BertEncoder, synth  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    bias_mean = float(sys.argv[3]) if len(sys.argv) > 3 else -3.0
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = synth.random_bert(cfg, seed=31)
    synth.random_mlm_head(cfg, seed=32, tied=True, sd=sd, bias_mean=bias_mean)
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
    rng = np.random.default_rng(6)
    lens = np.clip(np.rint(rng.normal(130, 30, size=batch)), 16, 256).astype(np.int64)
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(batch, T)).astype(np.int64) * mask
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    out = {}
    for name, fn in (("cls", lambda: enc.encode_pooled(kw, "cls")), ("splade", lambda: enc.encode_splade(kw))):
        fn()
        torch.cuda.synchronize()
        ms = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            e = fn()
            ms += enc.counters()["forward_ms"]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        c = enc.counters()
        out[name] = {"kernels_ms": ms / steps, "wall_ms": wall, "tflops": c["flops"] / (ms / steps * 1e-3) / 1e12,
                     "passages_per_s": batch / wall * 1e3}
        if name == "splade":
            out[name]["density"] = float((e > 0).float().mean())
            out[name]["real_tokens"] = int(c["real_tokens"])
    head_ms = out["splade"]["kernels_ms"] - out["cls"]["kernels_ms"]
    head_flops = out["splade"]["real_tokens"] * (2.0 * 768 * 768 + 2.0 * 768 * 30522)
    out["head"] = {"ms": head_ms, "tflops": head_flops / (head_ms * 1e-3) / 1e12,
                   "logits_bytes_not_written": out["splade"]["real_tokens"] * 30522 * 2}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
