"""
deberta_oracle.py — numpy fp64 restatement of the DeBERTa-v2 / v3 cross-encoder forward pass.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): only tests/ import it.

What it restates
  * the reference's rerank step, models/rerankers/crossencoder.py:34-38 (CrossEncoder.__call__):
    ``self.model(**kwargs).logits`` with the model of config/reranker/debertav3.yaml:3
    (naver/trecdl22-crossencoder-debertav3, a ``DebertaV2ForSequenceClassification``) loaded through
    AutoModelForSequenceClassification (crossencoder.py:18);
  * the arithmetic of HF ``DebertaV2ForSequenceClassification.forward``.  transformers is a third-party dependency that is
    NOT under /root/reference and is unpinned there; this restates the published architecture as implemented in
    transformers 5.15.0 (this image), transformers/models/deberta_v2/modeling_deberta_v2.py:
        DebertaV2Embeddings.forward          :518-563   word embeddings (no absolute positions: position_biased_input
                                                        = False; no token types: type_vocab_size = 0) -> LayerNorm
        make_log_bucket_position             :57-69     log-bucketed relative positions
        DisentangledSelfAttention.forward    :191-274   (Q K^T + c2p + p2c) / sqrt(3 d_head) -> softmax -> V
            .disentangled_attention_bias     :276-346   c2p[i, j] = Q_i . Kr[t(i - j)],  p2c[i, j] = K_j . Qr[t(i - j)]
                                                        with Kr = key_proj(rel), Qr = query_proj(rel) (share_att_key),
                                                        t(delta) = clamp(bucket(delta) + span, 0, 2 span - 1)
        DebertaV2Encoder.get_rel_embedding   :595-599   LayerNorm over the relative-position embedding table
        DebertaV2SelfOutput / Intermediate / Output     the usual post-LN block
        ContextPooler.forward                :988-996   gelu(dense(hidden[:, 0]))
        DebertaV2ForSequenceClassification   :1009-     classifier(pooled)
    (dropout is inactive: eval mode, torch.no_grad().)

Covered configuration (what deberta-v3-{small,base,large} ship): relative_attention, pos_att_type = p2c|c2p,
share_att_key, norm_rel_ebd = layer_norm, position_buckets > 0, position_biased_input = False, no conv layer.

Parity status: PINNED against HF ``DebertaV2ForSequenceClassification`` itself, driven through the reference's unmodified
``CrossEncoder.__call__`` in this container on seeded random weights (oracle/make_golden_deberta.py ->
tests/golden/deberta_tiny.npz; tests/test_deberta_oracle.py).  The trained checkpoint is not available offline.
"""
import math

import numpy as np

from .bert_oracle import _gelu, _ln


def relative_index_table(max_len, position_buckets, max_relative_positions):
    """t(delta) for delta = -(max_len - 1) .. max_len - 1 as an int array of 2 max_len - 1 entries (index delta + max_len - 1).
    The bucket function is evaluated with the very torch float32 operations of make_log_bucket_position
    (modeling_deberta_v2.py:57-69): its ceil() sits on float32 logarithms, so another libm could move a boundary."""
    import torch
    rel = torch.arange(-(max_len - 1), max_len, dtype=torch.long)
    mid = position_buckets // 2
    sign = torch.sign(rel)
    abs_pos = torch.where((rel < mid) & (rel > -mid), torch.tensor(mid - 1).type_as(rel), torch.abs(rel))
    log_pos = torch.ceil(torch.log(abs_pos / mid) / torch.log(torch.tensor((max_relative_positions - 1) / mid)) * (mid - 1)) + mid
    bucket = torch.where(abs_pos <= mid, rel.type_as(log_pos), log_pos * sign).to(torch.long)
    span = position_buckets
    return torch.clamp(bucket + span, 0, 2 * span - 1).numpy().astype(np.int64)


def deberta_forward(sd, cfg, input_ids, attention_mask=None, dtype=np.float64):
    """Last hidden state [B, T, d] of a DebertaV2Model with state dict `sd` (name -> numpy array, HF names without the
    'deberta.' prefix).  cfg: num_hidden_layers, num_attention_heads, layer_norm_eps, position_buckets,
    max_position_embeddings (= max_relative_positions when that is -1)."""
    W = lambda k: np.asarray(sd[k], dtype)
    ids = np.asarray(input_ids)
    B, T = ids.shape
    mask = np.ones((B, T), np.int64) if attention_mask is None else np.asarray(attention_mask)
    eps = cfg.get("layer_norm_eps", 1e-7)
    nh = cfg["num_attention_heads"]
    span = cfg["position_buckets"]
    max_rel = cfg.get("max_relative_positions", -1)
    if max_rel < 1:
        max_rel = cfg["max_position_embeddings"]
    x = _ln(W("embeddings.word_embeddings.weight")[ids], W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), eps)
    x = x * (mask != 0)[..., None]  # DebertaV2Embeddings.forward: embeddings * mask
    d = x.shape[-1]
    dh = d // nh
    rel = _ln(W("encoder.rel_embeddings.weight")[:2 * span], W("encoder.LayerNorm.weight"), W("encoder.LayerNorm.bias"), eps)
    table = relative_index_table(T, span, max_rel)                       # t(delta)
    t_ij = table[(np.arange(T)[:, None] - np.arange(T)[None, :]) + T - 1]  # [T, T]: t(i - j)
    # keys AND queries outside the mask are masked (attention_mask = m_i m_j); masked query rows are never read
    neg = np.where((mask[:, None, :, None] != 0) & (mask[:, None, None, :] != 0), 0.0, -np.inf)
    scale = math.sqrt(dh * 3.0)
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        lin = lambda t, n: t @ W(p + n + ".weight").T + W(p + n + ".bias")
        heads = lambda t: t.reshape(t.shape[0], -1, nh, dh).transpose(0, 2, 1, 3)
        q = heads(lin(x, "attention.self.query_proj"))
        k = heads(lin(x, "attention.self.key_proj"))
        v = heads(lin(x, "attention.self.value_proj"))
        qr = heads(lin(rel[None], "attention.self.query_proj"))[0]    # [nh, 2 span, dh]
        kr = heads(lin(rel[None], "attention.self.key_proj"))[0]
        s = q @ k.transpose(0, 1, 3, 2)
        c2p_full = np.einsum("bhid,hpd->bhip", q, kr)                  # Q_i . Kr[p]
        p2c_full = np.einsum("bhjd,hpd->bhjp", k, qr)                  # K_j . Qr[p]
        c2p = np.take_along_axis(c2p_full, np.broadcast_to(t_ij, (B, nh, T, T)), axis=-1)            # [.., i, j] = c2p_full[i, t(i-j)]
        p2c = np.take_along_axis(p2c_full, np.broadcast_to(t_ij.T, (B, nh, T, T)), axis=-1)          # [.., j, i] = p2c_full[j, t(i-j)]
        s = (s + c2p + p2c.transpose(0, 1, 3, 2)) / scale + neg
        with np.errstate(invalid="ignore"):
            s = s - np.where(np.isfinite(s.max(-1, keepdims=True)), s.max(-1, keepdims=True), 0.0)
        pr = np.exp(s)
        den = pr.sum(-1, keepdims=True)
        pr = pr / np.where(den > 0, den, 1.0)                           # (fully masked query rows: zeros, never read)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = _ln(lin(ctx, "attention.output.dense") + x, W(p + "attention.output.LayerNorm.weight"),
                W(p + "attention.output.LayerNorm.bias"), eps)
        h = _gelu(lin(x, "intermediate.dense"))
        x = _ln(lin(h, "output.dense") + x, W(p + "output.LayerNorm.weight"), W(p + "output.LayerNorm.bias"), eps)
    return x


def seqcls_logits(sd, hidden, dtype=np.float64):
    """ContextPooler (gelu(dense(hidden[:, 0]))) + classifier -> [B, num_labels]."""
    W = lambda k: np.asarray(sd[k], dtype)
    pooled = _gelu(hidden[:, 0] @ W("pooler.dense.weight").T + W("pooler.dense.bias"))
    return pooled @ W("classifier.weight").T + W("classifier.bias")


def cross_encode(sd, cfg, input_ids, attention_mask):
    """CrossEncoder.__call__ (reference crossencoder.py:34-38) on a DeBERTa-v2/v3 checkpoint -> [B, num_labels] float64."""
    return seqcls_logits(sd, deberta_forward(sd, cfg, input_ids, attention_mask))


def random_deberta(cfg, seed, num_labels=1):
    """Seeded random weights under HF's names (without the 'deberta.' prefix), fp16-exact values as float32 arrays."""
    rng = np.random.default_rng(seed)
    d, dff, V, span = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["position_buckets"]
    h16 = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    mat = lambda o, i, s=0.06: h16(rng.standard_normal((o, i)) * s)
    vec = lambda n, s=0.05: h16(rng.standard_normal(n) * s)
    gam = lambda n: h16(1.0 + 0.1 * rng.standard_normal(n))
    sd = {"embeddings.word_embeddings.weight": mat(V, d, 0.5), "embeddings.LayerNorm.weight": gam(d), "embeddings.LayerNorm.bias": vec(d),
          "encoder.rel_embeddings.weight": mat(2 * span, d, 0.5), "encoder.LayerNorm.weight": gam(d), "encoder.LayerNorm.bias": vec(d)}
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        for n in ("query_proj", "key_proj", "value_proj"):
            sd[p + f"attention.self.{n}.weight"] = mat(d, d)
            sd[p + f"attention.self.{n}.bias"] = vec(d)
        sd[p + "attention.output.dense.weight"] = mat(d, d)
        sd[p + "attention.output.dense.bias"] = vec(d)
        sd[p + "attention.output.LayerNorm.weight"] = gam(d)
        sd[p + "attention.output.LayerNorm.bias"] = vec(d)
        sd[p + "intermediate.dense.weight"] = mat(dff, d)
        sd[p + "intermediate.dense.bias"] = vec(dff)
        sd[p + "output.dense.weight"] = mat(d, dff)
        sd[p + "output.dense.bias"] = vec(d)
        sd[p + "output.LayerNorm.weight"] = gam(d)
        sd[p + "output.LayerNorm.bias"] = vec(d)
    sd["pooler.dense.weight"] = mat(d, d)
    sd["pooler.dense.bias"] = vec(d)
    sd["classifier.weight"] = mat(num_labels, d, 0.2)
    sd["classifier.bias"] = vec(num_labels)
    return sd
