"""
nomic_oracle.py — numpy fp64 restatement of the NomicBert forward pass (rotary positions, gated SiLU feed-forward).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import it.

What it restates
  * the reference's encode step for `config/retriever/nomic-embed-text-v1.5.yaml` (Dense + MeanPooler + CosineSim,
    prompts "search_query: " / "search_document: "): models/retrievers/dense.py:37-47 — encoder(**kwargs)[0], then
    pooler.pool(hidden, attention_mask);
  * the arithmetic of the encoder the reference reaches through ``AutoModel.from_pretrained(..., trust_remote_code=True)``
    (dense.py:16).  The checkpoint's own modelling file is remote code that is NOT available offline; transformers 5.15.0
    (the version in this image; third party, unpinned in the reference's requirements.txt) carries the same architecture
    natively, transformers/models/nomic_bert/modeling_nomic_bert.py:
        NomicBertEmbeddings.forward        :49-92    word + token_type embeddings -> LayerNorm (NO position table)
        NomicBertRotaryEmbedding           :95-147   inv_freq[j] = theta^(-2j / head_dim), theta = rope_parameters.rope_theta
                                                     (default 1000); cos / sin of position x inv_freq, the 32 angles repeated
                                                     over both halves of a head
        rotate_half / apply_rotary_pos_emb :150-181  q' = q cos + rotate_half(q) sin, rotate_half(x) = (-x2, x1); same for k
        NomicBertAttention.forward         :214-263  bias-free q / k / v / o projections, softmax(q' k'^T / sqrt(head_dim)) v
        NomicBertMLP.forward               :266-279  down(silu(gate(x)) * up(x)), bias-free
        NomicBertLayer.forward             :282-316  x = LN(x + attn(x));  x = LN(x + mlp(x))   (post-LN, like BERT)
        NomicBertModel.forward             :position_ids = arange(T) for every row (right-padded batches)
    (dropout is inactive: the reference runs under torch.no_grad() on an eval-mode model.)

Parity status: PINNED against HF ``NomicBertModel`` itself run in this container on seeded random weights, loaded from a
checkpoint directory through the reference's unmodified ``Dense`` (oracle/make_golden_nomic.py -> tests/golden/nomic_tiny.npz;
tests/test_nomic_oracle.py).  The trained nomic-embed-text-v1.5 weights are not available offline, so parity on them — and
on the hub checkpoint's remote modelling code — is unpinned.
"""
import numpy as np

from .bert_oracle import _ln, mean_pool


def rotary_tables(n_pos, head_dim, theta, dtype=np.float64):
    """cos, sin [n_pos, head_dim]: the head_dim / 2 angles position * theta^(-2j / head_dim), laid out twice
    (NomicBertRotaryEmbedding.forward, modeling_nomic_bert.py:131-147: emb = cat(freqs, freqs))."""
    inv_freq = 1.0 / (float(theta) ** (np.arange(0, head_dim, 2, dtype=np.float64) / head_dim))
    freqs = np.arange(n_pos, dtype=np.float64)[:, None] * inv_freq[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(dtype), np.sin(emb).astype(dtype)


def rotate_half(x):
    """(-x2, x1) over the two halves of the last dimension (modeling_nomic_bert.py:150-154)."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def _silu(x):
    return x / (1.0 + np.exp(-x))


def nomic_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, dtype=np.float64):
    """Last hidden state [B, T, d] of a NomicBertModel with state_dict `sd` (name -> numpy array, HF names: layers.<l>.*).

    cfg: dict with num_hidden_layers, num_attention_heads, layer_norm_eps, rope_theta."""
    W = lambda k: np.asarray(sd[k], dtype)
    ids = np.asarray(input_ids)
    B, T = ids.shape
    mask = np.ones((B, T), np.int64) if attention_mask is None else np.asarray(attention_mask)
    types = np.zeros((B, T), np.int64) if token_type_ids is None else np.asarray(token_type_ids)
    eps = cfg.get("layer_norm_eps", 1e-12)
    nh = cfg["num_attention_heads"]
    x = W("embeddings.word_embeddings.weight")[ids] + W("embeddings.token_type_embeddings.weight")[types]
    x = _ln(x, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), eps)
    d = x.shape[-1]
    dh = d // nh
    cos, sin = rotary_tables(T, dh, cfg.get("rope_theta", 1000.0), dtype)  # position_ids = arange(T) for every row
    neg = np.where(mask[:, None, None, :] != 0, 0.0, -np.inf)  # additive mask on the keys
    for l in range(cfg["num_hidden_layers"]):
        p = f"layers.{l}."
        lin = lambda t, n: t @ W(p + n + ".weight").T  # every projection of the layer is bias-free
        q = lin(x, "self_attn.q_proj").reshape(B, T, nh, dh).transpose(0, 2, 1, 3)
        k = lin(x, "self_attn.k_proj").reshape(B, T, nh, dh).transpose(0, 2, 1, 3)
        v = lin(x, "self_attn.v_proj").reshape(B, T, nh, dh).transpose(0, 2, 1, 3)
        q = q * cos[None, None] + rotate_half(q) * sin[None, None]
        k = k * cos[None, None] + rotate_half(k) * sin[None, None]
        s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(dh) + neg
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = _ln(lin(ctx, "self_attn.o_proj") + x, W(p + "post_attention_layernorm.weight"), W(p + "post_attention_layernorm.bias"), eps)
        h = _silu(lin(x, "mlp.gate_proj")) * lin(x, "mlp.up_proj")
        x = _ln(lin(h, "mlp.down_proj") + x, W(p + "post_mlp_layernorm.weight"), W(p + "post_mlp_layernorm.bias"), eps)
    return x


def encode(sd, cfg, input_ids, attention_mask, token_type_ids=None, l2_normalize=False):
    """Dense.__call__ (reference dense.py:37-47) with the MeanPooler of nomic-embed-text-v1.5.yaml -> [B, d] float64."""
    e = mean_pool(nomic_forward(sd, cfg, input_ids, attention_mask, token_type_ids), attention_mask)
    if l2_normalize:
        e = e / np.linalg.norm(e, axis=-1, keepdims=True)
    return e


# ---- op-level references for the kernel parity tests -------------------------------------------------------

def rotary_ref(qk, pos, n_heads, theta):
    """fp64 reference of bh_op_rotary: rows of [Q | K] ([rows][2 * n_heads * 64]) rotated in place by their positions."""
    qk = np.asarray(qk, np.float64).copy()
    pos = np.asarray(pos)
    cos, sin = rotary_tables(int(pos.max()) + 1, 64, theta)
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    x = qk.reshape(qk.shape[0], 2 * n_heads, 64)
    return (x * c + rotate_half(x) * s).reshape(qk.shape)


def swiglu_ref(gu):
    """fp64 reference of bh_op_swiglu: [rows][2 dff] of (gate, up) column PAIRS -> silu(gate) * up [rows][dff]."""
    gu = np.asarray(gu, np.float64)
    return _silu(gu[:, 0::2]) * gu[:, 1::2]


# seeded synthetic weights live in the product package's bench helpers (no arithmetic of the path)
from bergen_amd.synth import random_nomic  # noqa: E402,F401
