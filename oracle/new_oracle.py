"""
new_oracle.py — numpy fp64 restatement of the "new" encoder architecture of Alibaba-NLP/gte-base-en-v1.5 / gte-large-en-v1.5
(config/retriever/gte-base-en-v1.5.yaml, gte-large-en-v1.5.yaml: Dense + ClsPooler + CosineSim, reached by the reference through
``AutoModel.from_pretrained(..., trust_remote_code=True)``, models/retrievers/dense.py:16).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

**PARITY UNPINNED.**  The architecture lives in the checkpoint's REMOTE modelling file (hub repository Alibaba-NLP/new-impl,
modeling.py / configuration.py: model_type "new"), a third-party dependency that is in neither /root/reference nor the transformers
package of this image (5.15.0 has no such model class), and there is no network.  What follows restates the PUBLISHED algorithm of that
file; it cannot be checked here against the file itself nor against a trained checkpoint.  What stands in for the pin:
  * the product never trusts this restatement alone: when bergen_amd converts such a model it holds the user's HF module (the remote
    code has been loaded by then) and runs a PROBE batch through both — `BertEncoder.self_check`; a mismatch keeps the run on the HF
    module, loudly (tests/test_gpu_gte.py exercises both outcomes with a torch re-implementation of the module);
  * tests/test_gte_oracle.py pins this file against an independent torch implementation written from the same description
    (tests/gte_torch_model.py) — a consistency check of two restatements, not a reference pin.

Restated (new-impl modeling.py, names as published):
    NewEmbeddings            word_embeddings (+ token_type_embeddings when type_vocab_size > 0) -> LayerNorm; NO position table when
                             position_embedding_type == "rope"; rope cos / sin from rotary_emb for positions arange(T)
    RotaryEmbedding          inv_freq[j] = base^(-2j / dim); emb = cat(freqs, freqs); rotate_half(x) = (-x2, x1)
    NTKScalingRotaryEmbedding (rope_scaling {"type": "ntk", "factor": f}, mixed_b None): the cos / sin cache is built once for
                             max_position_embeddings * f positions, i.e. always in the scaled regime:
                             base' = base * f;  inv_freq = base'^(-2j / dim) / f^(2 / dim)
    NewAttention             qkv_proj (Linear hidden -> 3 hidden, bias) split [q | k | v]; rotary on q, k; softmax(q k^T / sqrt(dim)) v;
                             o_proj (Linear, bias)
    NewGatedMLP              up_gate_proj (Linear hidden -> 2 intermediate, NO bias) split [up | gate]; down_proj(act(gate) * up) (bias),
                             act = erf-GELU (hidden_act "gelu")
    NewLayer                 x = attn_ln(x + attention(x));  x = mlp_ln(x + mlp(x))      (post-LN, like BERT)
"""
import math

import numpy as np

from .bert_oracle import _gelu, _ln
from .nomic_oracle import rotate_half


def ntk_rotary_tables(n_pos, head_dim, base, factor=None, dtype=np.float64):
    """cos, sin [n_pos, head_dim] of RotaryEmbedding (factor None) or NTKScalingRotaryEmbedding (mixed_b None) as described above."""
    j = np.arange(0, head_dim, 2, dtype=np.float64) / head_dim
    if factor is None:
        inv_freq = 1.0 / (float(base) ** j)
    else:
        inv_freq = 1.0 / ((float(base) * float(factor)) ** j) / (float(factor) ** (2.0 / head_dim))
    freqs = np.arange(n_pos, dtype=np.float64)[:, None] * inv_freq[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(dtype), np.sin(emb).astype(dtype)


def new_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, dtype=np.float64):
    """Last hidden state [B, T, d] of a NewModel with state dict `sd` (numpy, the remote file's names: encoder.layer.<l>.attention.qkv_proj /
    o_proj, mlp.up_gate_proj / down_proj, attn_ln, mlp_ln).  cfg: num_hidden_layers, num_attention_heads, intermediate_size,
    layer_norm_eps, rope_theta, rope_scaling (None or {"type": "ntk", "factor": f}), type_vocab_size."""
    W = lambda k: np.asarray(sd[k], dtype)
    ids = np.asarray(input_ids)
    B, T = ids.shape
    mask = np.ones((B, T), np.int64) if attention_mask is None else np.asarray(attention_mask)
    eps = cfg.get("layer_norm_eps", 1e-12)
    nh = cfg["num_attention_heads"]
    x = W("embeddings.word_embeddings.weight")[ids]
    if int(cfg.get("type_vocab_size", 0) or 0) > 0:
        types = np.zeros((B, T), np.int64) if token_type_ids is None else np.asarray(token_type_ids)
        x = x + W("embeddings.token_type_embeddings.weight")[types]
    x = _ln(x, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), eps)
    d = x.shape[-1]
    dh = d // nh
    f = cfg["intermediate_size"]
    scaling = cfg.get("rope_scaling") or None
    cos, sin = ntk_rotary_tables(T, dh, cfg.get("rope_theta", 10000.0), None if scaling is None else scaling["factor"], dtype)
    neg = np.where(mask[:, None, None, :] != 0, 0.0, -np.inf)
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        qkv = x @ W(p + "attention.qkv_proj.weight").T + W(p + "attention.qkv_proj.bias")
        q, k, v = (t.reshape(B, T, nh, dh).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, axis=-1))
        q = q * cos[None, None] + rotate_half(q) * sin[None, None]
        k = k * cos[None, None] + rotate_half(k) * sin[None, None]
        s = q @ k.transpose(0, 1, 3, 2) / math.sqrt(dh) + neg
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = _ln(ctx @ W(p + "attention.o_proj.weight").T + W(p + "attention.o_proj.bias") + x, W(p + "attn_ln.weight"), W(p + "attn_ln.bias"), eps)
        ug = x @ W(p + "mlp.up_gate_proj.weight").T
        up, gate = ug[..., :f], ug[..., f:]
        h = _gelu(gate) * up
        x = _ln(h @ W(p + "mlp.down_proj.weight").T + W(p + "mlp.down_proj.bias") + x, W(p + "mlp_ln.weight"), W(p + "mlp_ln.bias"), eps)
    return x


def encode(sd, cfg, input_ids, attention_mask, pooler="cls", l2_normalize=False):
    """Dense.__call__ (reference dense.py:37-47) with the ClsPooler of gte-*-en-v1.5.yaml -> [B, d] float64."""
    h = new_forward(sd, cfg, input_ids, attention_mask)
    if pooler == "cls":
        e = h[:, 0]
    else:
        m = np.asarray(attention_mask, np.float64)[..., None]
        e = (h * m).sum(1) / m.sum(1)
    if l2_normalize:
        e = e / np.linalg.norm(e, axis=-1, keepdims=True)
    return e


def geglu_ref(gu):
    """fp64 reference of bh_op_gated_act(act = 1): [rows][2 dff] of (gate, up) column PAIRS -> gelu(gate) * up [rows][dff]."""
    gu = np.asarray(gu, np.float64)
    return _gelu(gu[:, 0::2]) * gu[:, 1::2]


# ---- JinaBert (jinaai/jina-embeddings-v2-*: config/retriever/jina-embeddings-v2-base-en.yaml, Dense + MeanPooler + CosineSim) ----------
# The same situation as "new": the architecture is the checkpoint's REMOTE modelling file (hub repository
# jinaai/jina-bert-implementation, modeling_bert.py: JinaBertModel, model_type "bert"), absent offline -> PARITY UNPINNED; restated from its
# published description, guarded in the product by the conversion self-check:
#     JinaBertEmbeddings       word + token_type embeddings -> LayerNorm; no position table when position_embedding_type == "alibi"
#     JinaBertSelfAttention    BERT's biased query / key / value projections; scores = q k^T / sqrt(dim) + mask + alibi,
#                              alibi[h][i][j] = -slope_h |i - j| (the symmetric encoder form), slope_h = _get_alibi_head_slopes(n_heads)
#     JinaBertSelfOutput       dense + LayerNorm(hidden + input)
#     JinaBertGLUMLP           (feed_forward_type "geglu") gated_layers (Linear hidden -> 2 intermediate, NO bias) split [gated | non-gated];
#                              wo(gelu(gated) * non_gated) (bias); layernorm(out + residual)

def alibi_slopes(n_heads):
    """The standard ALiBi head slopes (Press et al.; JinaBert's _get_alibi_head_slopes)."""
    def pow2(n):
        start = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(n_heads).is_integer():
        return np.asarray(pow2(n_heads))
    closest = 2 ** math.floor(math.log2(n_heads))
    return np.asarray(pow2(closest) + list(alibi_slopes(2 * closest))[0::2][: n_heads - closest])


def jina_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, dtype=np.float64):
    """Last hidden state [B, T, d] of a JinaBertModel (alibi, geglu) with state dict `sd` (numpy, the remote file's names: BERT's for the
    embeddings and the attention, encoder.layer.<l>.mlp.gated_layers / wo / layernorm for the feed-forward)."""
    W = lambda k: np.asarray(sd[k], dtype)
    ids = np.asarray(input_ids)
    B, T = ids.shape
    mask = np.ones((B, T), np.int64) if attention_mask is None else np.asarray(attention_mask)
    types = np.zeros((B, T), np.int64) if token_type_ids is None else np.asarray(token_type_ids)
    eps = cfg.get("layer_norm_eps", 1e-12)
    nh, f = cfg["num_attention_heads"], cfg["intermediate_size"]
    x = W("embeddings.word_embeddings.weight")[ids] + W("embeddings.token_type_embeddings.weight")[types]
    x = _ln(x, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), eps)
    d = x.shape[-1]
    dh = d // nh
    pos = np.arange(T)
    bias = -alibi_slopes(nh)[:, None, None] * np.abs(pos[:, None] - pos[None, :])[None]  # [nh, T, T]
    neg = np.where(mask[:, None, None, :] != 0, 0.0, -np.inf)
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        lin = lambda t, n: t @ W(p + n + ".weight").T + W(p + n + ".bias")
        q, k, v = (lin(x, "attention.self." + n).reshape(B, T, nh, dh).transpose(0, 2, 1, 3) for n in ("query", "key", "value"))
        s = q @ k.transpose(0, 1, 3, 2) / math.sqrt(dh) + bias[None] + neg
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = _ln(lin(ctx, "attention.output.dense") + x, W(p + "attention.output.LayerNorm.weight"), W(p + "attention.output.LayerNorm.bias"), eps)
        gl = x @ W(p + "mlp.gated_layers.weight").T
        h = _gelu(gl[..., :f]) * gl[..., f:]
        x = _ln(lin(h, "mlp.wo") + x, W(p + "mlp.layernorm.weight"), W(p + "mlp.layernorm.bias"), eps)
    return x


# seeded synthetic weights live in the product package's bench helpers (no arithmetic of the path)
from bergen_amd.synth import random_jina, random_new  # noqa: E402,F401
