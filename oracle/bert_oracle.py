"""
bert_oracle.py — numpy fp32/fp64 restatement of the bi-encoder forward pass.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

What it restates
  * the reference's encode step, models/retrievers/dense.py:37-47 (Dense.__call__): encoder(**kwargs)[0]
    then pooler.pool(hidden, attention_mask);
  * MeanPooler.pool (dense.py:64-69): masked sum over tokens / number of unmasked tokens;
    ClsPooler.pool (dense.py:71-75): hidden[:, 0];
  * the arithmetic of HF ``BertModel.forward`` that the reference reaches through ``AutoModel``
    (dense.py:16).  transformers is a third-party dependency that is NOT under /root/reference and is
    unpinned there (requirements.txt has no versions); this restates the published BERT encoder as
    implemented in transformers 5.15.0 (the version in this image),
    transformers/models/bert/modeling_bert.py:
        BertEmbeddings.forward      :70-110   word + token_type + position embeddings -> LayerNorm
        BertSelfAttention.forward   :139-     softmax(Q K^T / sqrt(d_head) + mask) V, mask = -inf-like on padding
        BertSelfOutput.forward      :282-     LayerNorm(dense(ctx) + input)
        BertIntermediate.forward    :325-     gelu(dense(x))   (erf GELU, config.hidden_act = "gelu")
        BertOutput.forward          :340-     LayerNorm(dense(h) + input)
        BertLayer / BertEncoder     :354-, :419-   the stack
    (dropout is inactive: the reference runs under torch.no_grad() on an eval-mode model.)

Parity status: PINNED against HF ``BertModel`` itself run in this container on seeded random weights
(oracle/make_golden_encoder.py -> tests/golden/bert_tiny.npz; tests/test_encoder_oracle.py).  Real retriever
checkpoints (RetroMAE, e5, ...) are not available offline, so parity on trained weights is unpinned.
"""
import math

import numpy as np


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)  # biased, like torch.nn.LayerNorm
    return (x - mu) / np.sqrt(var + eps) * g + b


_erf = np.vectorize(math.erf, otypes=[np.float64])


def _gelu(x):
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def bert_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, dtype=np.float64):
    """Last hidden state [B, T, d] of a BertModel with state_dict `sd` (name -> numpy array, HF names).

    cfg: dict with num_hidden_layers, num_attention_heads, layer_norm_eps.  Computed in `dtype` (fp64 by
    default: the oracle is the exact-arithmetic reading of the fp16-rounded weights).
    """
    W = lambda k: np.asarray(sd[k], dtype)
    ids = np.asarray(input_ids)
    B, T = ids.shape
    mask = np.ones((B, T), np.int64) if attention_mask is None else np.asarray(attention_mask)
    types = np.zeros((B, T), np.int64) if token_type_ids is None else np.asarray(token_type_ids)
    eps = cfg.get("layer_norm_eps", 1e-12)
    nh = cfg["num_attention_heads"]
    # BertEmbeddings.forward (modeling_bert.py:70-110): position_ids = arange(T)
    x = W("embeddings.word_embeddings.weight")[ids] + W("embeddings.token_type_embeddings.weight")[types] \
        + W("embeddings.position_embeddings.weight")[np.arange(T)][None]
    x = _ln(x, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), eps)
    d = x.shape[-1]
    dh = d // nh
    neg = np.where(mask[:, None, None, :] != 0, 0.0, -np.inf)  # additive mask on the keys
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        lin = lambda t, n: t @ W(p + n + ".weight").T + W(p + n + ".bias")
        q = lin(x, "attention.self.query").reshape(B, T, nh, dh).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(B, T, nh, dh).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(B, T, nh, dh).transpose(0, 2, 1, 3)
        s = q @ k.transpose(0, 1, 3, 2) / math.sqrt(dh) + neg
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = _ln(lin(ctx, "attention.output.dense") + x, W(p + "attention.output.LayerNorm.weight"),
                W(p + "attention.output.LayerNorm.bias"), eps)
        h = _gelu(lin(x, "intermediate.dense"))
        x = _ln(lin(h, "output.dense") + x, W(p + "output.LayerNorm.weight"), W(p + "output.LayerNorm.bias"), eps)
    return x


def mean_pool(hidden, mask):
    """MeanPooler.pool, reference models/retrievers/dense.py:64-69."""
    m = (np.asarray(mask) != 0)
    h = np.where(m[..., None], hidden, 0.0)
    return h.sum(1) / m.sum(1)[..., None]


def cls_pool(hidden, mask=None):
    """ClsPooler.pool, reference models/retrievers/dense.py:71-75."""
    return hidden[:, 0]


def encode(sd, cfg, input_ids, attention_mask, token_type_ids=None, pooler="cls", l2_normalize=False):
    """Dense.__call__ (reference dense.py:37-47) with the chosen pooler -> [B, d] float64."""
    h = bert_forward(sd, cfg, input_ids, attention_mask, token_type_ids)
    e = cls_pool(h) if pooler == "cls" else mean_pool(h, attention_mask)
    if l2_normalize:
        e = e / np.linalg.norm(e, axis=-1, keepdims=True)
    return e


def mlm_logits(sd, cfg, hidden, dtype=np.float64):
    """BertOnlyMLMHead on [B, T, d] hidden states -> [B, T, vocab] logits (transformers modeling_bert.py:
    BertPredictionHeadTransform :443-  dense -> gelu -> LayerNorm;  BertLMPredictionHead :462-  decoder(+bias);
    reached from the reference through AutoModelForMaskedLM, models/retrievers/splade.py:17,36-40).
    A state dict without cls.predictions.decoder.weight is a tied head: the word-embedding matrix."""
    W = lambda k: np.asarray(sd[k], dtype)
    eps = cfg.get("layer_norm_eps", 1e-12)
    t = _gelu(hidden @ W("cls.predictions.transform.dense.weight").T + W("cls.predictions.transform.dense.bias"))
    t = _ln(t, W("cls.predictions.transform.LayerNorm.weight"), W("cls.predictions.transform.LayerNorm.bias"), eps)
    wdec = W("cls.predictions.decoder.weight") if "cls.predictions.decoder.weight" in sd \
        else W("embeddings.word_embeddings.weight")
    bdec = W("cls.predictions.decoder.bias") if "cls.predictions.decoder.bias" in sd else 0.0
    return t @ wdec.T + bdec


def splade_pool(logits, mask):
    """SPLADE pooling, reference models/retrievers/splade.py:42-43:
    max over tokens of log(1 + relu(logits)) * attention_mask  -> [B, vocab]."""
    m = (np.asarray(mask) != 0)[..., None]
    return (np.log1p(np.maximum(logits, 0.0)) * m).max(1)


def encode_splade(sd, cfg, input_ids, attention_mask, token_type_ids=None):
    """Splade.__call__ (reference splade.py:34-47) -> [B, vocab] float64."""
    h = bert_forward(sd, cfg, input_ids, attention_mask, token_type_ids)
    return splade_pool(mlm_logits(sd, cfg, h), attention_mask)


def seqcls_logits(sd, cfg, hidden, dtype=np.float64):
    """BertForSequenceClassification head on [B, T, d] hidden states -> [B, num_labels] (transformers
    modeling_bert.py: BertPooler :425-  tanh(dense(hidden[:, 0]));  classifier = Linear(d, num_labels); dropout is
    inactive in eval mode; reached from the reference through AutoModelForSequenceClassification,
    models/rerankers/crossencoder.py:18,34-38)."""
    W = lambda k: np.asarray(sd[k], dtype)
    pooled = np.tanh(hidden[:, 0] @ W("pooler.dense.weight").T + W("pooler.dense.bias"))
    return pooled @ W("classifier.weight").T + W("classifier.bias")


def cross_encode(sd, cfg, input_ids, attention_mask, token_type_ids=None):
    """CrossEncoder.__call__ (reference crossencoder.py:34-38) -> [B, num_labels] float64 logits."""
    return seqcls_logits(sd, cfg, bert_forward(sd, cfg, input_ids, attention_mask, token_type_ids))


# ---- op-level references for the kernel parity tests -------------------------------------------------------

def gemm_ref(a, w, bias=None, bias_mode=1, residual=None, gelu=False):
    """fp64 reference of bh_op_gemm_f16: a[M,K] @ w[N,K]^T (+bias per column | per row) (+residual) (GELU)."""
    c = np.asarray(a, np.float64) @ np.asarray(w, np.float64).T
    if bias is not None:
        b = np.asarray(bias, np.float64)
        c = c + (b[None, :] if bias_mode == 1 else b[:, None])
    if residual is not None:
        c = c + np.asarray(residual, np.float64)
    if gelu:
        c = _gelu(c)
    return c


def attention_ref(qk, vt, seq_off, seq_len, n_heads):
    """fp64 reference of bh_op_attention over packed rows; rows outside every sequence stay 0."""
    qk = np.asarray(qk, np.float64)
    vt = np.asarray(vt, np.float64)
    d = n_heads * 64
    ctx = np.zeros((qk.shape[0], d))
    for off, n in zip(seq_off, seq_len):
        for h in range(n_heads):
            q = qk[off:off + n, h * 64:(h + 1) * 64]
            k = qk[off:off + n, d + h * 64:d + (h + 1) * 64]
            v = vt[h * 64:(h + 1) * 64, off:off + n].T
            s = q @ k.T / 8.0
            s = s - s.max(-1, keepdims=True)
            p = np.exp(s)
            p /= p.sum(-1, keepdims=True)
            ctx[off:off + n, h * 64:(h + 1) * 64] = p @ v
    return ctx


def layernorm_ref(x, g, b, eps):
    return _ln(np.asarray(x, np.float64), np.asarray(g, np.float64), np.asarray(b, np.float64), eps)


# seeded synthetic weights / batches live in the product package's bench helpers (no arithmetic of the path)
from bergen_amd.synth import random_batch, random_bert, random_cls_head, random_mlm_head  # noqa: E402,F401
