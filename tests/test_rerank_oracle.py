"""CPU: the rerank stage's host logic and the numpy sequence-classification oracle against HF
BertForSequenceClassification + the reference's CrossEncoder.__call__ / Rerank.sort_by_score_indexes
(tests/golden/rerank_tiny.npz, made by oracle/make_golden_rerank.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import bert_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rerank_tiny.npz")


def load(labels):
    z = np.load(GOLD)
    cfg = {k: (float(v) if "." in v or "e-" in v else int(v)) if v.replace(".", "").replace("e-", "").isdigit() else v
           for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    sd = {k[3:]: z[k].astype(np.float32) for k in z.files if k.startswith("w::")}
    sd.update({k.split("::", 1)[1]: z[k].astype(np.float32) for k in z.files if k.startswith(f"w_{labels}::")})
    return z, cfg, sd


@pytest.mark.parametrize("labels", [1, 3])
def test_oracle_matches_hf_through_the_reference_call(labels):
    z, cfg, sd = load(labels)
    got = bert_oracle.cross_encode(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"])
    assert got.shape == (12, labels)
    np.testing.assert_allclose(got, z[f"ref_score_{labels}"], rtol=0, atol=2e-5)


def test_sort_by_score_indexes_matches_reference():
    import bergen_amd
    z, _, _ = load(1)
    rr = object.__new__(bergen_amd.Rerank)
    q_ids = [f"q{i // 4}" for i in range(12)]
    d_ids = [f"d{i}" for i in range(12)]
    qs, ds, ss = rr.sort_by_score_indexes(torch.from_numpy(z["sort_scores_in"]), q_ids, d_ids)
    assert qs == list(z["sort_q"])
    assert ds == [list(r) for r in z["sort_d"]]  # includes a tie: stable, retrieval order kept
    assert np.array_equal(np.stack([s.numpy() for s in ss]), z["sort_s"])


def test_rerank_stage_with_a_stub_model_and_config_aliases():
    import bergen_amd
    from bergen_amd import config

    class Stub(bergen_amd.Reranker):
        def __init__(self):
            super().__init__("org/stub-reranker")
            self.model = torch.nn.Identity()

        def collate_fn(self, examples, query_or_doc=None):
            return {"x": torch.tensor([[float(len(e["doc"]))] for e in examples]),
                    "q_id": [e["q_id"] for e in examples], "d_id": [e["d_id"] for e in examples]}

        def __call__(self, kwargs):
            return {"score": kwargs["x"]}

    data = [{"query": "q", "doc": "x" * n, "q_id": f"q{i % 2}", "d_id": f"d{i}"} for i, n in enumerate([3, 9, 5, 1, 7, 2])]
    r = bergen_amd.Rerank(init_args=Stub(), batch_size=4)
    out = r.eval(data)
    assert out["q_id"] == ["q0", "q1"]
    assert out["doc_id"] == [["d4", "d2", "d0"], ["d1", "d5", "d3"]]
    assert [s.tolist() for s in out["score"]] == [[7.0, 5.0, 3.0], [9.0, 2.0, 1.0]]
    assert r.get_clean_model_name() == "org_stub-reranker"
    assert config._locate("models.rerankers.crossencoder.CrossEncoder") is bergen_amd.CrossEncoder
    assert config._locate("modules.rerank.Rerank") is bergen_amd.Rerank


def test_cross_encoder_collate_matches_reference_contract():
    """padding='max_length', truncation='only_second', q_id / d_id carried beside the tensors (crossencoder.py:24-33)."""
    import bergen_amd
    seen = {}

    class Tok:
        def __call__(self, a, b, **kw):
            seen.update(kw, a=a, b=b)
            return {"input_ids": torch.zeros(len(a), kw["max_length"], dtype=torch.long)}

    ce = bergen_amd.CrossEncoder("org/m", max_len=32, model=torch.nn.Identity(), tokenizer=Tok())
    out = ce.collate_fn([{"query": "a", "doc": "b", "q_id": "q", "d_id": "d"}])
    assert seen["padding"] == "max_length" and seen["truncation"] == "only_second" and seen["max_length"] == 32
    assert seen["a"] == ["a"] and seen["b"] == ["b"] and out["q_id"] == ["q"] and out["d_id"] == ["d"]


@pytest.mark.parametrize("batch_size,launch_pairs,want_launches", [(4, 256, 1), (4, 8, 3), (4, 1, 5), (4, 6, 3), (32, 256, 1)])
def test_native_pipeline_coalesces_yaml_batches_and_keeps_the_reference_order(batch_size, launch_pairs, want_launches):
    """Rerank.eval on the native path (modules/rerank.py:24-48 as a pipeline): whole yaml batches coalesced into launches of
    >= launch_pairs pairs, tokenised ahead on threads, ONE copy of the scores back — pairs stay in dataset order, so the grouping
    and the per-query stable sort are the reference loop's whatever the launch size."""
    import bergen_amd

    launches = []

    class FakeEncoder:
        num_labels = 1

        def classify(self, batch):
            launches.append(int(batch["x"].shape[0]))
            return batch["x"].float() * 2.0

        def counters(self):
            return {"flops": 10.0, "forward_ms": 1.0}

    class FakeCE(bergen_amd.Reranker):
        native = True

        def __init__(self):
            super().__init__("org/fake-native")
            self.model = FakeEncoder()

        def collate_fn(self, examples, query_or_doc=None):
            raise AssertionError("the native pipeline tokenises with collate_packed")

        def collate_packed(self, examples):
            return {"x": torch.tensor([[float(len(e["doc"]))] for e in examples]),
                    "q_id": [e["q_id"] for e in examples], "d_id": [e["d_id"] for e in examples]}

        def __call__(self, kwargs):
            raise AssertionError("the native pipeline calls classify on the encoder")

    lens = [3, 9, 5, 1, 7, 2, 8, 8, 4, 6, 11, 10, 13, 12, 0, 15, 14]  # 17 pairs: a ragged last launch; a tie (8, 8)
    data = [{"query": "q", "doc": "x" * n, "q_id": f"q{i % 3}", "d_id": f"d{i}"} for i, n in enumerate(lens)]
    r = bergen_amd.Rerank(init_args=FakeCE(), batch_size=batch_size, launch_pairs=launch_pairs, num_workers=3)
    out = r.eval(data)
    assert len(launches) == want_launches and sum(launches) == len(lens)
    assert all(n % batch_size == 0 for n in launches[:-1]), "a launch is a whole number of yaml batches"
    ref = bergen_amd.Rerank.sort_by_score_indexes(r, torch.tensor([2.0 * n for n in lens]), [d["q_id"] for d in data], [d["d_id"] for d in data])
    assert out["q_id"] == ref[0] and out["doc_id"] == ref[1]
    assert [s.tolist() for s in out["score"]] == [s.tolist() for s in ref[2]]
    assert out["doc_id"][0] == ["d15", "d12", "d6", "d9", "d0", "d3"]  # q0 by descending length: 15, 13, 8, 6, 3, 1
    assert out["doc_id"][1].index("d7") < out["doc_id"][1].index("d4")   # q1: d7 (8) above d4 (7)
    st = r.last_eval_stats
    assert st["pairs"] == 17 and st["launches"] == want_launches and st["algorithmic_flops"] == 10.0 * want_launches


def test_fast_pair_tokenisation_equals_the_hf_call():
    """CrossEncoder.collate_packed's fast path (dense.fast_tokenize_pairs) against the HF call it replaces — same ids, type ids and mask,
    truncation of the SECOND sequence only, right padding to the longest pair — on a WordPiece tokenizer with BERT's pair template."""
    import sys
    sys.argv = [sys.argv[0]]
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_for_tok", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from bergen_amd.dense import fast_tokenize_pairs
    tok, words = bench.toy_wordpiece(2000, seed=3)
    rng = np.random.default_rng(4)
    qs = [" ".join(words[j] for j in rng.integers(0, len(words), size=int(rng.integers(1, 14)))) for _ in range(37)]
    ds = [" ".join(words[j] for j in rng.integers(0, len(words), size=int(rng.integers(1, 90)))) for _ in range(37)]
    ds[5] = ""  # an empty passage
    for max_len in (256, 48, 20):  # 48 / 20: the passages (and only they) are truncated
        want = tok(qs, ds, padding=True, truncation="only_second", max_length=max_len, return_tensors="pt")
        got = fast_tokenize_pairs(tok, qs, ds, max_len)
        assert got is not None
        for key in ("input_ids", "token_type_ids", "attention_mask"):
            assert torch.equal(got[key], want[key]), (max_len, key)
        assert int(want["input_ids"].shape[1]) <= max_len
    # the singles' fast path configures another truncation strategy on the same backend: each call sets its own
    from bergen_amd.dense import fast_tokenize
    single = fast_tokenize(tok, ds[:7], 32)
    assert torch.equal(single["input_ids"], tok(ds[:7], padding=True, truncation=True, max_length=32, return_tensors="pt")["input_ids"])
    again = fast_tokenize_pairs(tok, qs, ds, 48)
    assert torch.equal(again["input_ids"], tok(qs, ds, padding=True, truncation="only_second", max_length=48, return_tensors="pt")["input_ids"])

    class Slow:
        is_fast = False
    assert fast_tokenize_pairs(Slow(), ["a"], ["b"], 8) is None


def test_native_pipeline_on_an_hf_dataset_and_on_an_empty_one():
    """The rerank stage is handed an HF Dataset by BERGEN (modules/rag.py builds it from the retrieval run): columnar slices instead of
    row look-ups, same result; an empty dataset gives empty lists (the reference's torch.cat of nothing would raise: a query set whose
    retrieval came back empty should not take the stage down)."""
    import datasets
    import bergen_amd

    class FakeEncoder:
        num_labels = 1

        def classify(self, batch):
            return batch["x"].float()

        def counters(self):
            return {"flops": 1.0, "forward_ms": 0.1}

    class FakeCE(bergen_amd.Reranker):
        native = True

        def __init__(self):
            super().__init__("org/fake-native")
            self.model = FakeEncoder()

        def collate_fn(self, examples, query_or_doc=None):
            raise AssertionError

        def collate_packed(self, examples):
            assert all(set(e) >= {"query", "doc", "q_id", "d_id"} for e in examples)
            return {"x": torch.tensor([[float(len(e["doc"]))] for e in examples]),
                    "q_id": [e["q_id"] for e in examples], "d_id": [e["d_id"] for e in examples]}

        def __call__(self, kwargs):
            raise AssertionError

    rows = [{"query": "q", "doc": "x" * n, "q_id": f"q{i % 2}", "d_id": f"d{i}"} for i, n in enumerate([3, 9, 5, 1, 7, 2, 8])]
    stage = bergen_amd.Rerank(init_args=FakeCE(), batch_size=2, launch_pairs=4, num_workers=2)
    out_list = stage.eval(rows)
    out_ds = stage.eval(datasets.Dataset.from_list(rows))
    assert out_ds["q_id"] == out_list["q_id"] == ["q0", "q1"] and out_ds["doc_id"] == out_list["doc_id"] == [["d6", "d4", "d2", "d0"], ["d1", "d5", "d3"]]
    assert [s.tolist() for s in out_ds["score"]] == [s.tolist() for s in out_list["score"]]
    empty = stage.eval([])
    assert empty == {"score": [], "doc_id": [], "q_id": []} and stage.last_eval_stats["launches"] == 0
