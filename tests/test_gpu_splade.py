"""GPU parity tests of the SPLADE encode path (masked-LM head + max-over-tokens pooling as hand-written HIP):
bergen_amd.BertEncoder.encode_splade through the C ABI against the HF-BertForMaskedLM / reference-Splade golden
fixture (tests/golden/splade_tiny.npz) and the fp64 oracle (oracle/bert_oracle.py).

Floating point (fp16 storage, fp32 accumulation).  Tolerance, written here:
  |emb - ref| <= 3e-2 absolute (values are log(1 + relu(logit)), O(1)); a term whose reference logit maximum is
  below -0.05 must be exactly 0, one above +0.05 must be > 0 (the support can only differ where the logit ~ 0).
"""
import numpy as np
import pytest
import torch

from oracle import bert_oracle

from test_splade_oracle import canonical_alt, load, load_alt

pytestmark = pytest.mark.gpu


def _native(cfg, sd):
    from bergen_amd import BertEncoder
    return BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)


def _kw(ids, mask, types):
    return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
            "token_type_ids": torch.from_numpy(types)}


def _check(got, ref_emb, ref_logit_max, what):
    got = got.float().cpu().numpy().astype(np.float64)
    assert got.shape == ref_emb.shape, (got.shape, ref_emb.shape)
    err = np.abs(got - ref_emb)
    assert err.max() <= 3e-2, f"{what}: max abs err {err.max():.4g}"
    assert np.all(got >= 0)
    assert np.all(got[ref_logit_max < -0.05] == 0), what
    assert np.all(got[ref_logit_max > 0.05] > 0), what
    return float(err.max())


def _logit_max(sd, cfg, ids, mask, types):
    h = bert_oracle.bert_forward(sd, cfg, ids, mask, types)
    lg = bert_oracle.mlm_logits(sd, cfg, h)
    return np.where((mask != 0)[..., None], lg, -np.inf).max(1), bert_oracle.splade_pool(lg, mask)


@pytest.mark.parametrize("tag", ["untied", "tied"])
def test_golden_fixture(tag):
    z, cfg, sd = load(tag)
    ids, mask, types = z["input_ids"], z["attention_mask"], z["token_type_ids"]
    enc = _native(cfg, sd)
    assert enc.has_mlm_head
    got = enc.encode_splade(_kw(ids, mask, types))
    lmax, emb = _logit_max(sd, cfg, ids, mask, types)
    _check(got, z[f"ref_emb_{tag}"].astype(np.float64), lmax, f"golden {tag}")
    _check(got, emb, lmax, f"oracle {tag}")
    # the hidden-state / pooled outputs of the same handle still work (shared layer stack)
    hid = enc(**_kw(ids, mask, types))[0]
    assert hid.shape == (ids.shape[0], ids.shape[1], cfg["hidden_size"])
    c = enc.counters()
    assert c["real_tokens"] == int(mask.sum())
    enc.close()


@pytest.mark.parametrize("kind", ["distilbert", "roberta"])
def test_golden_fixture_other_mlm_heads(kind):
    """DistilBertForMaskedLM (splade-efficient.yaml's checkpoints) and RobertaForMaskedLM: the HF state dict under HF's own names
    into BertEncoder, against the reference Splade.__call__'s output (tests/golden/splade_tiny_<kind>.npz) and the oracle."""
    from bergen_amd import BertEncoder
    z, cfg, sd = load_alt(kind)
    enc = BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)
    assert enc.has_mlm_head, "the masked-LM head tensors were dropped"
    ids, mask = z["input_ids"], z["attention_mask"]
    got = enc.encode_splade({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)})
    canon, csd = canonical_alt(cfg, sd)
    if "cls.predictions.bias" in csd:
        csd.setdefault("cls.predictions.decoder.bias", csd["cls.predictions.bias"])
    off = int(canon.get("position_offset", 0))
    if off:
        csd["embeddings.position_embeddings.weight"] = csd["embeddings.position_embeddings.weight"][off:]
    lmax, emb = _logit_max(csd, dict(canon, hidden_act="gelu"), ids, mask, np.zeros_like(ids))
    _check(got, z["ref_emb"].astype(np.float64), lmax, f"golden {kind}")
    _check(got, emb, lmax, f"oracle {kind}")
    enc.close()


def test_bert_base_shape_full_vocabulary():
    """12 x 768 with the 30522-term vocabulary of the naver/splade-* checkpoints (not a multiple of the 256-row
    tile: the decoder is zero-padded), ragged batch, tied decoder."""
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=256, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=31)
    bert_oracle.random_mlm_head(cfg, seed=32, tied=True, sd=sd, bias_mean=-3.0)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=5, max_len=60, seed=33)
    enc = _native(cfg, sd)
    got = enc.encode_splade(_kw(ids, mask, types))
    lmax, emb = _logit_max(sd, cfg, ids, mask, types)
    err = _check(got, emb, lmax, "bert-base")
    print(f"bert-base splade: max abs err {err:.4g}, density {(emb > 0).mean():.4f}")
    enc.close()


def test_batch_composition_invariance_and_determinism():
    """The running maximum is order-independent: same bits alone / in a batch / reversed / repeated."""
    z, cfg, sd = load("untied")
    ids, mask, types = z["input_ids"], z["attention_mask"], z["token_type_ids"]
    enc = _native(cfg, sd)
    full = enc.encode_splade(_kw(ids, mask, types)).cpu().numpy().view(np.uint16)
    again = enc.encode_splade(_kw(ids, mask, types)).cpu().numpy().view(np.uint16)
    assert np.array_equal(full, again)
    for b in (0, 4, ids.shape[0] - 1):
        alone = enc.encode_splade(_kw(ids[b:b + 1], mask[b:b + 1], types[b:b + 1])).cpu().numpy().view(np.uint16)
        assert np.array_equal(alone[0], full[b]), b
    rev = enc.encode_splade(_kw(ids[::-1].copy(), mask[::-1].copy(), types[::-1].copy())).cpu().numpy().view(np.uint16)
    assert np.array_equal(rev[::-1], full)
    enc.close()


def test_many_sequences_and_left_padding():
    """More sequences than fit one token tile, single-token sequences, and masks with holes / left padding
    (only attended tokens count: reference splade.py:43 multiplies by the mask)."""
    z, cfg, sd = load("untied")
    rng = np.random.default_rng(5)
    B, T = 300, 24
    ids = rng.integers(1, cfg["vocab_size"], size=(B, T)).astype(np.int64)
    mask = (rng.random((B, T)) < 0.7).astype(np.int64)
    mask[:, 0] = 1
    mask[:40, 1:] = 0  # single-token sequences
    mask[40:80] = np.flip(np.arange(T)[None] < rng.integers(1, T, size=(40, 1)), axis=1)  # left padding
    mask[40:80, -1] = 1
    types = np.zeros_like(ids)
    enc = _native(cfg, sd)
    got = enc.encode_splade(_kw(ids, mask, types))
    # positions follow the ORIGINAL column (HF position_ids = arange(T)), not the packed rank
    lmax, emb = _logit_max(sd, cfg, ids, mask, types)
    _check(got, emb, lmax, "ragged masks")
    enc.close()


def test_errors():
    from bergen_amd import BertEncoder
    z, cfg, sd = load("untied")
    plain = {k: v for k, v in sd.items() if not k.startswith("cls.")}
    enc = _native(cfg, plain)
    assert not enc.has_mlm_head
    with pytest.raises(RuntimeError):
        enc.encode_splade({"input_ids": torch.tensor([[3, 4]])})
    with pytest.raises(IOError):  # the C ABI itself refuses pool 3 without the head (BH_EINCOMPLETE)
        enc._forward(torch.tensor([[3, 4]]), None, None, 3)
    enc.close()
    partial = dict(plain)
    partial["cls.predictions.decoder.bias"] = sd["cls.predictions.decoder.bias"]
    with pytest.raises(IOError):  # head without its transform weights
        BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in partial.items()}, device=0)


def test_splade_plugin_end_to_end(tmp_path):
    """Splade + Retrieve on the native path: HIP encoder with MLM head -> sparse chunks -> resident CSR index ->
    fused sparse search; ranking must agree with an exact search over the oracle's SPLADE vectors."""
    import bergen_amd
    z, cfg, sd = load("untied")

    class ToyTokenizer:
        vocab = {}

        def __call__(self, texts, padding=None, truncation=None, max_length=None, return_tensors=None):
            rows = [[1] + [2 + (hash_(w) % (cfg["vocab_size"] - 2)) for w in t.split()][:max_length - 1] for t in texts]
            T = max(len(r) for r in rows)
            ids = torch.tensor([r + [0] * (T - len(r)) for r in rows])
            mask = torch.tensor([[1] * len(r) + [0] * (T - len(r)) for r in rows])
            return {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}

    def hash_(w):
        v = 0
        for ch in w:
            v = (v * 131 + ord(ch)) % 1000003
        return v

    class Col:
        def __init__(self, rows):
            self.rows = rows

        def __len__(self):
            return len(self.rows)

        def __getitem__(self, i):
            if isinstance(i, str):
                return [r[i] for r in self.rows]
            return self.rows[i]

        def remove_columns(self, cols):
            return Col([{k: v for k, v in r.items() if k not in cols} for r in self.rows])

    rng = np.random.default_rng(4)
    words = [f"w{i}" for i in range(300)]
    docs = [" ".join(rng.choice(words, size=int(rng.integers(3, 40)))) for _ in range(150)]
    queries = [" ".join(d.split()[:8]) for d in docs[:7]]
    enc = _native(cfg, sd)
    tok = ToyTokenizer()
    model = bergen_amd.Splade("toy/splade-tiny", max_len=48, model=enc, tokenizer=tok)
    assert model.model is enc and model.query_encoder is enc
    ds = {"doc": Col([{"id": str(i), "content": t} for i, t in enumerate(docs)]),
          "query": Col([{"id": f"q{i}", "generated_query": t} for i, t in enumerate(queries)])}
    r = bergen_amd.Retrieve(init_args=model, batch_size=64, batch_size_sim=4, num_workers=0)
    out = r.retrieve(ds, str(tmp_path / "q"), str(tmp_path / "d"), 10)
    r.close()
    bd, bq = tok(docs, max_length=48), tok(queries, max_length=48)
    ed = bert_oracle.encode_splade(sd, cfg, bd["input_ids"].numpy(), bd["attention_mask"].numpy())
    eq = bert_oracle.encode_splade(sd, cfg, bq["input_ids"].numpy(), bq["attention_mask"].numpy())
    want = np.argsort(-(eq @ ed.T), axis=1)[:, :10]
    got = np.array([[int(x) for x in row] for row in out["doc_id"]])
    overlap = np.mean([len(set(g) & set(w)) / 10 for g, w in zip(got, want)])
    assert overlap >= 0.9, overlap
    assert (got[:, 0] == want[:, 0]).mean() >= 0.8
