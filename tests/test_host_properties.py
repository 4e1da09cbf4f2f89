"""CPU property tests (hypothesis) of the host-side pieces around the hot path: id table, run-file round trip,
chunk ordering, rank-range partitions, the rerank grouping."""
import os

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import bergen_amd
from bergen_amd import utils

ids_strategy = st.lists(st.text(alphabet="abcdefXYZ0123456789_-", min_size=1, max_size=12), min_size=1, max_size=60)


@settings(max_examples=60, deadline=None)
@given(ids=ids_strategy, probes=st.lists(st.text(alphabet="abcdefXYZ0123456789_-", min_size=0, max_size=12), max_size=20))
def test_id_index_equals_dict(ids, probes):
    ref = {}
    for row, k in enumerate(ids):
        ref[k] = row
    table = utils.IdIndex(ids)
    ask = probes + ids[::3]
    rows, found = table.get_many(ask)
    assert [int(r) if f else None for r, f in zip(rows, found)] == [ref.get(k) for k in ask]
    assert len(table) == len(ref)


@settings(max_examples=40, deadline=None)
@given(nq=st.integers(1, 6), k=st.integers(1, 7), seed=st.integers(0, 10_000))
def test_trec_round_trip(tmp_path_factory, nq, k, seed):
    rng = np.random.default_rng(seed)
    scores = torch.from_numpy(np.sort(rng.standard_normal((nq, k)).astype(np.float32), axis=1)[:, ::-1].copy())
    q_ids = [f"q{i}" for i in range(nq)]
    d_ids = [[f"d{rng.integers(0, 1000)}_{j}" for j in range(k)] for _ in range(nq)]
    path = os.path.join(tmp_path_factory.mktemp("trec"), "run.trec")
    utils.write_trec(path, q_ids, d_ids, scores)
    lines = open(path).read().splitlines()
    assert len(lines) == nq * k and all(len(l.split("\t")) == 6 for l in lines)
    back_q, back_d, back_s = utils.load_trec(path)
    assert back_q == q_ids and back_d == d_ids
    assert np.array_equal(np.asarray(back_s, np.float32), scores.numpy())  # python float repr round-trips fp32 exactly


@settings(max_examples=80, deadline=None)
@given(n=st.integers(0, 10_000), world=st.integers(1, 16))
def test_shard_ranges_partition_exactly(n, world):
    spans = [bergen_amd.shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert all(lo <= hi for lo, hi in spans)
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) <= -(-n // world) if n else max(sizes) == 0


@settings(max_examples=60, deadline=None)
@given(data=st.lists(st.tuples(st.integers(0, 4), st.floats(-5, 5, allow_nan=False, width=32)), min_size=1, max_size=40))
def test_rerank_grouping_is_a_stable_descending_sort_per_query(data):
    rr = object.__new__(bergen_amd.Rerank)
    q_ids = [f"q{q}" for q, _ in data]
    d_ids = [f"d{i}" for i in range(len(data))]
    scores = torch.tensor([s for _, s in data], dtype=torch.float32)
    out_q, out_d, out_s = rr.sort_by_score_indexes(scores, q_ids, d_ids)
    assert out_q == list(dict.fromkeys(q_ids))
    for q, docs, sc in zip(out_q, out_d, out_s):
        rows = [i for i, x in enumerate(q_ids) if x == q]
        want = sorted(rows, key=lambda i: -float(scores[i]))  # python's sort is stable: ties keep incoming order
        assert docs == [d_ids[i] for i in want]
        assert sc.tolist() == [float(scores[i]) for i in want]


def test_chunk_files_sort_by_the_integer_in_the_path(tmp_path):
    names = [3, 292, 584, 10, 1]
    for n in names:
        torch.save(torch.zeros(1, 4, dtype=torch.float16), tmp_path / f"embedding_chunk_{n}.pt")
    got = [int(os.path.basename(p).split("_")[-1].split(".")[0]) for p in utils.sorted_chunk_files(str(tmp_path))]
    assert got == sorted(names)
