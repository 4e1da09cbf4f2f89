"""CPU: bergen_amd/evaluation.py (the reference's utils.eval_retrieval_kilt, utils.py:263-300) against oracle/trec_eval_oracle.c — an
independent C restatement of the trec_eval rules behind pytrec_eval's P_1 / recall_k (form_res_rels.c ranking, m_P.c, m_recall.c).
pytrec_eval itself is not installable offline: parity with it stays formally unpinned (oracle header), but the two implementations
were written separately, in two languages, from the published algorithm, and must agree on every random case below."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from bergen_amd import evaluation
from oracle import c_oracle

DOCS = [f"{p}{i}" for p in ("", "wiki_", "Q") for i in (1, 2, 3, 10, 11, 20, 100)] + ["é", "z", "Z", "10_2"]
# scores that tie exactly, tie only after narrowing to a C float (trec_eval's `float sim`), and differ
SCORES = [0.0, 1.0, 1.0 + 1e-12, 1.0 + 2e-7, 84.8125, 8.769950866699219, -3.5, 1e-3, 7.0, 7.000000001]


@st.composite
def cases(draw):
    n_q = draw(st.integers(1, 6))
    run, qrel = {}, {}
    for q in range(n_q):
        q_id = f"q{q}"
        docs = draw(st.lists(st.sampled_from(DOCS), min_size=0, max_size=12, unique=True))
        if draw(st.booleans()) or q == 0:
            run[q_id] = {d: draw(st.sampled_from(SCORES)) for d in docs}
        if draw(st.integers(0, 4)) > 0:  # most topics are judged; some run topics are not
            judged = draw(st.lists(st.sampled_from(DOCS), min_size=0, max_size=8, unique=True))
            qrel[q_id] = {d: draw(st.sampled_from([-1, 0, 0, 1, 1, 2])) for d in judged}
    # topics of OTHER datasets in the qrels (scripts/kilt_generate_qrels.py:38,58-62 never resets its dict): must not count
    for extra in range(draw(st.integers(0, 3))):
        qrel[f"other{extra}"] = {"1": 1}
    return run, qrel, draw(st.sampled_from([1, 2, 5, 20]))


@settings(max_examples=300, deadline=None)
@given(cases())
def test_ranking_metrics_match_the_trec_eval_restatement(case):
    run, qrel, k = case
    got = evaluation.ranking_metrics(run, qrel, k)
    want, n = c_oracle.trec_eval_mean_metrics(run, qrel, k)
    assert set(got) == {"P_1", f"recall_{k}"}
    assert abs(got["P_1"] - want["P_1"]) < 1e-12 and abs(got[f"recall_{k}"] - want[f"recall_{k}"]) < 1e-12, (run, qrel, k, got, want)
    assert n == sum(1 for q in run if q in qrel)


def test_hand_cases_of_the_rules_that_differ_from_a_naive_reading():
    # ties by document id DESCENDING: of two top-scoring documents "b" outranks "a"
    run = {"q": {"a": 2.0, "b": 2.0, "c": 1.0}}
    assert c_oracle.trec_eval_mean_metrics(run, {"q": {"a": 1}}, 1)[0] == {"P_1": 0.0, "recall_1": 0.0}
    assert c_oracle.trec_eval_mean_metrics(run, {"q": {"b": 1}}, 1)[0] == {"P_1": 1.0, "recall_1": 1.0}
    assert evaluation.ranking_metrics(run, {"q": {"b": 1}}, 1) == {"P_1": 1.0, "recall_1": 1.0}
    # a tie that exists only in fp32: 1 + 1e-12 does not outrank 1 — the id decides ("x" > "a")
    run = {"q": {"a": 1.0 + 1e-12, "x": 1.0}}
    assert evaluation.ranking_metrics(run, {"q": {"a": 1}}, 1)["P_1"] == 0.0 == c_oracle.trec_eval_mean_metrics(run, {"q": {"a": 1}}, 1)[0]["P_1"]
    # a judged topic without a relevant document scores 0 and COUNTS in the mean; an unjudged topic does not
    run = {"q1": {"a": 1.0}, "q2": {"a": 1.0}, "q3": {"a": 1.0}}
    qrel = {"q1": {"a": 1}, "q2": {"a": 0}}
    assert evaluation.ranking_metrics(run, qrel, 5) == {"P_1": 0.5, "recall_5": 0.5} == c_oracle.trec_eval_mean_metrics(run, qrel, 5)[0]
    # recall's denominator is every relevant judged document, retrieved or not; negative relevance is not relevant
    run = {"q": {"a": 3.0, "b": 2.0}}
    qrel = {"q": {"a": 2, "zz": 1, "b": -1, "yy": 1}}
    assert c_oracle.trec_eval_mean_metrics(run, qrel, 5)[0]["recall_5"] == pytest.approx(1 / 3)
    assert evaluation.ranking_metrics(run, qrel, 5)["recall_5"] == pytest.approx(1 / 3)
    # nothing in common: 0 / max(1, 0)
    assert evaluation.ranking_metrics({"q": {"a": 1.0}}, {"p": {"a": 1}}, 5) == {"P_1": 0.0, "recall_5": 0.0} == \
        c_oracle.trec_eval_mean_metrics({"q": {"a": 1.0}}, {"p": {"a": 1}}, 5)[0]


def test_eval_retrieval_kilt_end_to_end_against_the_oracle(tmp_path):
    """The whole function on a qrels FILE with the reference's quirks: the doc_dataset_name key, topics of earlier datasets, several
    passages of one page (max-passage), page ids as strings."""
    rng = np.random.default_rng(4)
    q_ids = [f"nq{i}" for i in range(40)]
    pages = [str(p) for p in rng.integers(1, 60, size=(40, 20))]
    doc_ids = [[str(p) for p in rng.integers(1, 60, size=20)] for _ in q_ids]       # pages repeat inside a query's list
    scores = np.sort(rng.random((40, 20)).astype(np.float32), axis=1)[:, ::-1]
    qrel = {"doc_dataset_name": "kilt-100w"}
    for j in range(25):
        qrel[f"aidayago2_{j}"] = {str(int(p)): 1 for p in rng.integers(1, 60, size=2)}    # another dataset's topics
    for qi, q in enumerate(q_ids[:35]):
        qrel[q] = {str(int(p)): int(r) for p, r in zip(rng.integers(1, 60, size=4), rng.integers(0, 2, size=4))}
    os.makedirs(tmp_path / "qrels")
    json.dump(qrel, open(tmp_path / "qrels" / "qrel.kilt_nq.dev.json", "w"))
    got = evaluation.eval_retrieval_kilt(str(tmp_path), str(tmp_path / "qrels"), "kilt_nq", "kilt-100w", "dev", q_ids, doc_ids, scores, top_k=5)
    run = evaluation.max_passage_run(q_ids, doc_ids, scores.tolist())
    qrel.pop("doc_dataset_name")
    want, n = c_oracle.trec_eval_mean_metrics(run, qrel, 5)
    assert n == 35 and abs(got["P_1"] - want["P_1"]) < 1e-12 and abs(got["recall_5"] - want["recall_5"]) < 1e-12
    assert 0 < got["recall_5"] < 1
    assert json.load(open(tmp_path / "eval_dev_ranking_metrics.json")) == got
    del pages
