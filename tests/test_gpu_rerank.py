"""GPU parity tests of the cross-encoder rerank path: BertEncoder.classify (encoder forward + BertPooler / classifier
head kernel) through the C ABI against the HF / reference golden fixture and the fp64 oracle.
Floating point: |logit - ref| <= 2e-2 absolute (fp16 storage, fp32 accumulation; logits are O(1))."""
import numpy as np
import pytest
import torch

from oracle import bert_oracle

from test_rerank_oracle import load

pytestmark = pytest.mark.gpu


def _native(cfg, sd):
    from bergen_amd import BertEncoder
    return BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)


def _kw(ids, mask, types):
    return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
            "token_type_ids": torch.from_numpy(types)}


@pytest.mark.parametrize("labels", [1, 3])
def test_golden_fixture(labels):
    z, cfg, sd = load(labels)
    enc = _native(cfg, sd)
    assert enc.num_labels == labels
    got = enc.classify(_kw(z["input_ids"], z["attention_mask"], z["token_type_ids"]))
    assert got.dtype == torch.float32 and tuple(got.shape) == (12, labels)
    err = np.abs(got.cpu().numpy() - z[f"ref_score_{labels}"]).max()
    assert err <= 2e-2, err
    enc.close()


def test_bert_large_shape_against_oracle():
    """BAAI/bge-large-en shape (24 x 1024 x 16 heads), (query, passage) pairs with token types, max_length padding."""
    cfg = dict(vocab_size=3000, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
               max_position_embeddings=256, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=51)
    bert_oracle.random_cls_head(cfg, seed=52, num_labels=1, sd=sd)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=5, max_len=90, seed=53)
    pad = 128 - ids.shape[1]
    ids, mask, types = (np.pad(a, ((0, 0), (0, pad))) for a in (ids, mask, types))
    enc = _native(cfg, sd)
    got = enc.classify(_kw(ids, mask, types)).cpu().numpy()
    ref = bert_oracle.cross_encode(sd, cfg, ids, mask, types)
    err = np.abs(got - ref).max()
    print(f"bert-large cross-encoder: max abs err {err:.4g}, logits {ref.ravel()[:3]}")
    assert err <= 2e-2
    enc.close()


def test_errors_and_batch_invariance():
    z, cfg, sd = load(1)
    enc = _native(cfg, sd)
    kw = _kw(z["input_ids"], z["attention_mask"], z["token_type_ids"])
    full = enc.classify(kw).cpu().numpy()
    one = enc.classify({k: v[3:4] for k, v in kw.items()}).cpu().numpy()
    assert np.array_equal(one[0].view(np.uint32), full[3].view(np.uint32))
    with pytest.raises(ValueError):  # the head reads the first token: it must be attended
        enc.classify({"input_ids": torch.ones(1, 3, dtype=torch.long), "attention_mask": torch.tensor([[0, 1, 1]])})
    enc.close()
    plain = {k: v for k, v in sd.items() if not (k.startswith("classifier.") or k.startswith("pooler."))}
    enc = _native(cfg, plain)
    assert enc.num_labels == 0
    with pytest.raises(RuntimeError):
        enc.classify(kw)
    with pytest.raises(IOError):  # C ABI: BH_EINCOMPLETE
        enc._forward(z["input_ids"], z["attention_mask"], None, 4)
    enc.close()


def test_rerank_stage_end_to_end_on_the_native_cross_encoder():
    """Rerank + CrossEncoder on the HIP path: per-query order must agree with the oracle's scores."""
    import bergen_amd
    z, cfg, sd = load(1)

    class Tok:
        def __call__(self, a, b, padding=None, truncation=None, max_length=None, return_tensors=None):
            rows, types = [], []
            for qa, db in zip(a, b):
                qt = [1] + [2 + (hash_(w) % 900) for w in qa.split()] + [3]
                dt = [2 + (hash_(w) % 900) for w in db.split()][:max_length - len(qt) - 1] + [3]
                rows.append(qt + dt)
                types.append([0] * len(qt) + [1] * len(dt))
            ids = torch.tensor([r + [0] * (max_length - len(r)) for r in rows])
            mask = torch.tensor([[1] * len(r) + [0] * (max_length - len(r)) for r in rows])
            tt = torch.tensor([t + [0] * (max_length - len(t)) for t in types])
            return {"input_ids": ids, "attention_mask": mask, "token_type_ids": tt}

    def hash_(w):
        v = 0
        for ch in w:
            v = (v * 131 + ord(ch)) % 1000003
        return v

    rng = np.random.default_rng(9)
    words = [f"w{i}" for i in range(200)]
    data = []
    for qi in range(5):
        q = " ".join(rng.choice(words, size=6))
        for di in range(7):
            data.append({"query": q, "doc": " ".join(rng.choice(words, size=int(rng.integers(5, 50)))),
                         "q_id": f"q{qi}", "d_id": f"d{qi}_{di}"})
    enc = _native(cfg, sd)
    tok = Tok()
    ce = bergen_amd.CrossEncoder("toy/cross-encoder", max_len=48, model=enc, tokenizer=tok)
    assert ce.native
    stage = bergen_amd.Rerank(init_args=ce, batch_size=16)
    out = stage.eval(data)
    assert stage.last_eval_stats["launches"] == 1 and stage.last_eval_stats["pairs"] == 35
    # the pipeline's launch size does not change a bit of the result: one launch per yaml batch of 4 (9 launches) vs one of 35 pairs
    small = bergen_amd.Rerank(init_args=ce, batch_size=4, launch_pairs=1, num_workers=2)
    out_small = small.eval(data)
    assert small.last_eval_stats["launches"] == 9
    assert out_small["q_id"] == out["q_id"] and out_small["doc_id"] == out["doc_id"]
    for a, b in zip(out_small["score"], out["score"]):
        assert np.array_equal(a.numpy().view(np.uint32), b.numpy().view(np.uint32))
    # and it equals the reference-shaped loop (collate_fn padded to max_len, one __call__ and one copy per batch)
    loop = torch.cat([ce(ce.collate_fn(data[b0:b0 + 16]) and {k: v for k, v in ce.collate_fn(data[b0:b0 + 16]).items() if k not in ("q_id", "d_id")})["score"].cpu()
                      for b0 in range(0, len(data), 16)]).reshape(-1)
    _, _, s_loop = stage.sort_by_score_indexes(loop, [e["q_id"] for e in data], [e["d_id"] for e in data])
    for a, b in zip(s_loop, out["score"]):
        assert np.array_equal(a.numpy().view(np.uint32), b.numpy().view(np.uint32))
    assert out["q_id"] == [f"q{i}" for i in range(5)]
    b = tok([e["query"] for e in data], [e["doc"] for e in data], max_length=48)
    ref = bert_oracle.cross_encode(sd, cfg, b["input_ids"].numpy(), b["attention_mask"].numpy(),
                                   b["token_type_ids"].numpy()).ravel()
    for qi in range(5):
        want = sorted(range(7), key=lambda j: -ref[qi * 7 + j])
        got = [int(d.split("_")[1]) for d in out["doc_id"][qi]]
        s = out["score"][qi].numpy()
        assert np.all(np.diff(s) <= 0)
        assert np.abs(np.sort(ref[qi * 7:qi * 7 + 7])[::-1] - s).max() <= 2e-2
        # identical order wherever the oracle's adjacent scores differ by more than the tolerance
        gaps = -np.diff(np.sort(ref[qi * 7:qi * 7 + 7])[::-1])
        if gaps.min() > 4e-2:
            assert got == want
