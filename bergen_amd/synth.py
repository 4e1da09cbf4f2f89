"""
Synthetic inputs for benchmarks and tests (no checkpoint, dataset or HF hub exists offline — SURVEY §8c/§8d):
seeded random BertModel weights under HF state_dict names, and right-padded token batches shaped like an HF
tokenizer's ``padding="longest"`` output (reference models/retrievers/dense.py:57).  Pure numpy; none of the
retrieval arithmetic lives here.
"""
import numpy as np


def random_bert(cfg, seed, scale=None):
    """Seeded random BertModel weights (HF names, fp16-rounded values stored as float32), numpy only —
    runs on the GPU box where no checkpoint and no HF hub exist.  Initialisation mimics HF (N(0, 0.02),
    LayerNorm weight ~ 1) with wider biases / gains so that every term of the forward pass matters."""
    rng = np.random.default_rng(seed)
    d, dff, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    s = 0.02 if scale is None else scale
    sd = {}

    def put(name, shape, std, mean=0.0):
        sd[name] = (rng.standard_normal(shape) * std + mean).astype(np.float16).astype(np.float32)

    put("embeddings.word_embeddings.weight", (cfg["vocab_size"], d), s * 2)
    put("embeddings.position_embeddings.weight", (cfg["max_position_embeddings"], d), s * 2)
    put("embeddings.token_type_embeddings.weight", (cfg["type_vocab_size"], d), s * 2)
    put("embeddings.LayerNorm.weight", (d,), 0.1, 1.0)
    put("embeddings.LayerNorm.bias", (d,), 0.1)
    for l in range(L):
        p = f"encoder.layer.{l}."
        for n, (o, i) in {"attention.self.query": (d, d), "attention.self.key": (d, d), "attention.self.value": (d, d),
                          "attention.output.dense": (d, d), "intermediate.dense": (dff, d),
                          "output.dense": (d, dff)}.items():
            put(p + n + ".weight", (o, i), s * 2.5)
            put(p + n + ".bias", (o,), 0.05)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            put(p + n + ".weight", (d,), 0.1, 1.0)
            put(p + n + ".bias", (d,), 0.1)
    return sd


def random_batch(cfg, batch, max_len, seed, min_len=1):
    """Right-padded [B, T] ids / mask / types like an HF tokenizer with padding="longest"."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, max_len + 1, size=batch)
    lens[rng.integers(0, batch)] = max_len
    T = int(lens.max())
    ids = rng.integers(1, cfg["vocab_size"], size=(batch, T)).astype(np.int64)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = ids * mask  # pad id 0
    types = (rng.integers(0, cfg["type_vocab_size"], size=(batch, T)) * mask).astype(np.int64)
    return ids, mask, types
