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


def random_mlm_head(cfg, seed, tied=False, sd=None, bias_mean=-1.5):
    """Seeded random BertOnlyMLMHead weights (HF names under cls.predictions.).  tied=True leaves out the decoder
    weight (HF ties it to the word embeddings).  The decoder bias is shifted negative so that, like a trained
    SPLADE model, most (token, term) logits are negative and the pooled vectors are sparse-ish."""
    rng = np.random.default_rng(seed)
    d, V = cfg["hidden_size"], cfg["vocab_size"]
    out = {}

    def put(name, shape, std, mean=0.0):
        out[name] = (rng.standard_normal(shape) * std + mean).astype(np.float16).astype(np.float32)

    put("cls.predictions.transform.dense.weight", (d, d), 0.05)
    put("cls.predictions.transform.dense.bias", (d,), 0.05)
    put("cls.predictions.transform.LayerNorm.weight", (d,), 0.1, 1.0)
    put("cls.predictions.transform.LayerNorm.bias", (d,), 0.1)
    if not tied:
        put("cls.predictions.decoder.weight", (V, d), 0.04)
    put("cls.predictions.decoder.bias", (V,), 0.3, bias_mean)
    if sd is not None:
        sd.update(out)
    return out


def random_cls_head(cfg, seed, num_labels=1, sd=None):
    """Seeded random BertPooler + classifier weights (HF BertForSequenceClassification names, `bert.` prefix
    stripped like the rest of this module's state dicts)."""
    rng = np.random.default_rng(seed)
    d = cfg["hidden_size"]
    out = {}

    def put(name, shape, std, mean=0.0):
        out[name] = (rng.standard_normal(shape) * std + mean).astype(np.float16).astype(np.float32)

    put("pooler.dense.weight", (d, d), 0.08)
    put("pooler.dense.bias", (d,), 0.05)
    put("classifier.weight", (num_labels, d), 0.1)
    put("classifier.bias", (num_labels,), 0.1)
    if sd is not None:
        sd.update(out)
    return out


def random_nomic(cfg, seed=0, scale=0.05):
    """Seeded random NomicBertModel state dict (transformers modeling_nomic_bert.py names: layers.<l>.self_attn.* / mlp.* /
    post_*_layernorm.*; float32): weights ~ N(0, scale), LayerNorm gains near 1, no biases (the architecture has none)."""
    rng = np.random.default_rng(seed)
    d, f = cfg["hidden_size"], cfg["intermediate_size"]
    n = lambda *shape: (rng.standard_normal(shape) * scale).astype(np.float32)
    g = lambda: (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    sd = {"embeddings.word_embeddings.weight": n(cfg["vocab_size"], d) * 4,
          "embeddings.token_type_embeddings.weight": n(cfg["type_vocab_size"], d) * 4,
          "embeddings.LayerNorm.weight": g(), "embeddings.LayerNorm.bias": n(d)}
    for l in range(cfg["num_hidden_layers"]):
        p = f"layers.{l}."
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{name}.weight"] = n(d, d) * 2
        sd[p + "post_attention_layernorm.weight"], sd[p + "post_attention_layernorm.bias"] = g(), n(d)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = n(f, d) * 2, n(f, d) * 2, n(d, f)
        sd[p + "post_mlp_layernorm.weight"], sd[p + "post_mlp_layernorm.bias"] = g(), n(d)
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


def random_sparse_corpus(n_docs, vocab, seed, mean_nnz=180, lo=16, hi=400, zipf_a=1.1):
    """Synthetic SPLADE-like document vectors: nnz ~ clipped-Poisson(mean_nnz) in [lo, hi], term
    ids Zipf(zipf_a)-distributed over the vocabulary without replacement inside a document, weights
    log1p(exponential) rounded to fp16.  Returns CSR (indptr int64, terms int32 sorted per row, weights float16)."""
    rng = np.random.default_rng(seed)
    nnz = np.clip(rng.poisson(mean_nnz, size=n_docs), lo, min(hi, vocab)).astype(np.int64)
    indptr = np.zeros(n_docs + 1, np.int64)
    np.cumsum(nnz, out=indptr[1:])
    p = 1.0 / np.arange(1, vocab + 1) ** zipf_a
    perm = rng.permutation(vocab)  # which term id has which popularity rank
    cdf = np.cumsum(p / p.sum())
    terms = np.empty(indptr[-1], np.int32)
    for r in range(n_docs):
        need = int(nnz[r])
        got = np.unique(perm[np.searchsorted(cdf, rng.random(need * 2)).clip(0, vocab - 1)])
        while got.size < need:
            got = np.unique(np.concatenate([got, perm[np.searchsorted(cdf, rng.random(need)).clip(0, vocab - 1)]]))
        terms[indptr[r]:indptr[r + 1]] = np.sort(rng.choice(got, size=need, replace=False))
    w = np.log1p(rng.exponential(1.0, size=indptr[-1])).astype(np.float16)
    w[w == 0] = np.float16(0.01)
    return indptr, terms, w


def csr_to_dense(indptr, terms, weights, vocab, dtype=np.float32):
    n = len(indptr) - 1
    out = np.zeros((n, vocab), dtype)
    for r in range(n):
        out[r, terms[indptr[r]:indptr[r + 1]]] = weights[indptr[r]:indptr[r + 1]]
    return out


def random_sparse_corpus_fast(n_docs, vocab, seed, mean_nnz=180, lo=16, hi=400, zipf_a=1.1):
    """Vectorised variant of random_sparse_corpus for benchmark-sized corpora: terms are drawn WITH replacement and
    duplicates inside a document are dropped (so documents come out slightly shorter than drawn)."""
    rng = np.random.default_rng(seed)
    nnz = np.clip(rng.poisson(mean_nnz, size=n_docs), lo, min(hi, vocab)).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1) ** zipf_a
    cdf = np.cumsum(p / p.sum())
    perm = rng.permutation(vocab)
    doc = np.repeat(np.arange(n_docs, dtype=np.int64), nnz)
    term = perm[np.searchsorted(cdf, rng.random(doc.size)).clip(0, vocab - 1)].astype(np.int64)
    key = np.unique(doc * 65536 + term)  # sorted by (doc, term), duplicates removed
    doc, term = key >> 16, (key & 65535).astype(np.int32)
    indptr = np.zeros(n_docs + 1, np.int64)
    np.cumsum(np.bincount(doc, minlength=n_docs), out=indptr[1:])
    w = np.log1p(rng.exponential(1.0, size=term.size)).astype(np.float16)
    w[w == 0] = np.float16(0.01)
    return indptr, term, w


def random_sparse_corpus_device(n_docs, vocab, seed, device, mean_nnz=180, lo=16, hi=400, zipf_a=1.1, chunk=8192):
    """SURVEY §8d S4 as stated, at benchmark size: document lengths ~ Poisson(mean_nnz) clipped to [lo, hi], term ids
    Zipf(zipf_a) over a permuted vocabulary drawn WITHOUT replacement, weights log1p(Exp(1)) in fp16.  Sampling nnz terms
    without replacement with probabilities p is taking the nnz smallest of E_i / p_i with E_i ~ Exp(1) (exponential
    clocks): one [chunk, vocab] draw and a top-`hi` per chunk of documents on the device (`device`: a torch device).
    Returns numpy CSR (indptr int64, terms int32 ascending inside a document, weights fp16) like random_sparse_corpus."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    rng = np.random.default_rng(seed)
    nnz = np.clip(rng.poisson(mean_nnz, size=n_docs), lo, min(hi, vocab)).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1) ** zipf_a
    perm = rng.permutation(vocab)
    p_of_term = np.empty(vocab)
    p_of_term[perm] = p / p.sum()
    inv_p = torch.from_numpy(1.0 / p_of_term).to(device=device, dtype=torch.float32)
    kmax = int(nnz.max())
    col = torch.arange(kmax, device=device)
    terms_out = []
    for c0 in range(0, n_docs, chunk):
        m = min(chunk, n_docs - c0)
        clock = torch.empty((m, vocab), device=device, dtype=torch.float32).exponential_(1.0, generator=g) * inv_p
        idx = clock.topk(kmax, dim=1, largest=False, sorted=True).indices  # the first nnz of a row: its sample, in draw order
        keep = col[None, :] < torch.from_numpy(nnz[c0:c0 + m]).to(device)[:, None]
        srt = idx.masked_fill(~keep, vocab).sort(dim=1).values              # ascending term ids, the dropped slots last
        terms_out.append(srt[srt < vocab].to(torch.int32).cpu().numpy())
        del clock, idx, keep, srt
    terms = np.concatenate(terms_out) if terms_out else np.zeros(0, np.int32)
    indptr = np.zeros(n_docs + 1, np.int64)
    np.cumsum(nnz, out=indptr[1:])
    assert terms.size == indptr[-1]
    w = np.log1p(rng.exponential(1.0, size=terms.size)).astype(np.float16)
    w[w == 0] = np.float16(0.01)
    return indptr, terms, w


def sparse_bench_blocks(n_docs, vocab, device, term_seeds=3, block=1_000_000):
    """The SPLADE corpus of BASELINE configs[3] at its stated size, block by block: DISTINCT blocks of `block`
    documents whose term sets come from `term_seeds` independent draws of the S4 recipe (random_sparse_corpus_device, seeds
    4, 1004, 2004, ...) used in turn; the first use of a draw keeps its own weights, every later block gets fresh weights
    log1p(Exp(1)) drawn on the device (seed 40 000 + block ordinal) — no two documents of the corpus share their scores.
    Yields (ordinal, first row, indptr int64 [m + 1], terms int32 [nnz], weights fp16 [nnz]) as numpy arrays; bench.py's SPLADE
    leg and tests/test_gpu_sparse.py::test_full_size_sparse iterate the SAME generator, so the test pins what the bench times."""
    import torch
    n_blocks = (n_docs + block - 1) // block
    sets = [random_sparse_corpus_device(min(block, n_docs), vocab, seed=4 + 1000 * j, device=device)
            for j in range(max(1, min(term_seeds, n_blocks)))]
    done = 0
    for b in range(n_blocks):
        indptr, terms, w0 = sets[b % len(sets)]
        m = min(len(indptr) - 1, n_docs - done)
        nnz_b = int(indptr[m])
        if b < len(sets):
            w = w0[:nnz_b]
        else:
            gw = torch.Generator(device=device).manual_seed(40_000 + b)
            w = torch.log1p(torch.empty(nnz_b, device=device).exponential_(1.0, generator=gw)).half().clamp_(min=0.01).cpu().numpy()
        yield b, done, indptr[:m + 1], terms[:nnz_b], w
        done += m


def random_new(cfg, seed=0, scale=0.05):
    """Seeded random NewModel state dict (numpy fp32, fp16-representable), the remote file's tensor names."""
    rng = np.random.default_rng(seed)
    d, f, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    r = lambda *s, sc=scale: (rng.standard_normal(s) * sc).astype(np.float16).astype(np.float32)
    g = lambda n: (1.0 + rng.standard_normal(n) * 0.05).astype(np.float16).astype(np.float32)
    sd = {"embeddings.word_embeddings.weight": r(V, d, sc=0.5), "embeddings.LayerNorm.weight": g(d), "embeddings.LayerNorm.bias": r(d)}
    if int(cfg.get("type_vocab_size", 0) or 0) > 0:
        sd["embeddings.token_type_embeddings.weight"] = r(cfg["type_vocab_size"], d, sc=0.5)
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        sd[p + "attention.qkv_proj.weight"] = r(3 * d, d)
        sd[p + "attention.qkv_proj.bias"] = r(3 * d)
        sd[p + "attention.o_proj.weight"] = r(d, d)
        sd[p + "attention.o_proj.bias"] = r(d)
        sd[p + "attn_ln.weight"] = g(d)
        sd[p + "attn_ln.bias"] = r(d)
        sd[p + "mlp.up_gate_proj.weight"] = r(2 * f, d)
        sd[p + "mlp.down_proj.weight"] = r(d, f)
        sd[p + "mlp.down_proj.bias"] = r(d)
        sd[p + "mlp_ln.weight"] = g(d)
        sd[p + "mlp_ln.bias"] = r(d)
    return sd


def random_jina(cfg, seed=0, scale=0.05):
    rng = np.random.default_rng(seed)
    d, f, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    r = lambda *s, sc=scale: (rng.standard_normal(s) * sc).astype(np.float16).astype(np.float32)
    g = lambda n: (1.0 + rng.standard_normal(n) * 0.05).astype(np.float16).astype(np.float32)
    sd = {"embeddings.word_embeddings.weight": r(V, d, sc=0.5), "embeddings.token_type_embeddings.weight": r(cfg.get("type_vocab_size", 2), d, sc=0.5),
          "embeddings.LayerNorm.weight": g(d), "embeddings.LayerNorm.bias": r(d)}
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = r(d, d), r(d)
        sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"] = g(d), r(d)
        sd[p + "mlp.gated_layers.weight"] = r(2 * f, d)
        sd[p + "mlp.wo.weight"], sd[p + "mlp.wo.bias"] = r(d, f), r(d)
        sd[p + "mlp.layernorm.weight"], sd[p + "mlp.layernorm.bias"] = g(d), r(d)
    return sd
