"""CPU: the oracle against itself (C vs numpy restatements) and against the committed golden
vectors produced by the REAL reference code (oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import c_oracle, compare, numpy_oracle, ref_port
from tests.conftest import gap_tolerance


def test_half_conversion_matches_numpy_exhaustively():
    # every fp16 bit pattern round-trips through the C decoder/encoder exactly like numpy
    bits = np.arange(65536, dtype=np.uint16)
    h = bits.view(np.float16)
    import ctypes
    out = np.empty(65536, np.float32)
    c_oracle.lib().oracle_halfs_to_floats(bits.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_int64(65536))
    want = h.astype(np.float32)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    # fp32 -> fp16 RNE on random + boundary values
    rng = np.random.default_rng(0)
    f = np.concatenate([(rng.standard_normal(200000) * 10.0 ** rng.integers(-8, 6, 200000)).astype(np.float32),
                        np.array([0.0, -0.0, 65504.0, 65520.0, 65519.99, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8,
                                  6.1e-5, np.inf, -np.inf], np.float32)])
    got = c_oracle.floats_to_halfs(f)
    with np.errstate(over="ignore"):
        want = f.astype(np.float16)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_c_and_numpy_restatements_agree_bitwise():
    rng = np.random.default_rng(5)
    q = rng.standard_normal((9, 96)).astype(np.float16)
    x = rng.standard_normal((700, 96)).astype(np.float16)
    x[10] = x[3]
    x[650] = x[3]
    cs, ci = c_oracle.canonical_search(q, x, 17, id_offset=1000)
    ns, ni = numpy_oracle.canonical_search(q, x, 17, id_offset=1000)
    compare.assert_bit_exact(cs, ci, ns, ni, "C vs numpy canonical")
    # fewer rows than k -> padded with (-inf, -1)
    cs, ci = c_oracle.canonical_search(q, x[:5], 8)
    ns, ni = numpy_oracle.canonical_search(q, x[:5], 8)
    compare.assert_bit_exact(cs, ci, ns, ni, "short index")
    assert (ci[:, 5:] == -1).all() and np.isneginf(cs[:, 5:]).all()


def test_canonical_order_on_exact_ties():
    # identical rows: ascending row index inside the tie group
    q = np.ones((1, 16), np.float16)
    x = np.zeros((40, 16), np.float16)
    x[[5, 9, 31, 2]] = 1.0
    s, i = c_oracle.canonical_search(q, x, 6)
    assert i[0].tolist() == [2, 5, 9, 31, 0, 1]
    assert s[0].tolist() == [16.0, 16.0, 16.0, 16.0, 0.0, 0.0]


def test_kat_small_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_small.npz"))
    q, x, k = g["q"], g["x"], int(g["k"])
    # canonical oracle reproduces the committed canonical answer bit-for-bit
    cs, ci = c_oracle.canonical_search(q, x, k)
    compare.assert_bit_exact(cs, ci, g["canon_scores"], g["canon_ids"], "kat canonical")
    # all scores in this KAT are exact in every precision: the reference's SCORES are identical,
    # only its order inside tie groups is arbitrary
    assert np.array_equal(cs, g["ref_scores"])
    st = compare.compare_near_tie(cs, ci, g["ref_scores"], g["ref_ids"], gap_tol=0.0)
    assert st["max_score_err"] == 0.0
    # the C restatement of the reference's chunk loop gives the same scores
    rs, ri = c_oracle.ref_chunked_search(q, x, g["chunk_rows"], 17, k)
    assert np.array_equal(rs, g["ref_scores"])
    compare.assert_bit_exact(rs, ri, cs, ci, "chunked restatement == canonical on exact data")


def test_config1_golden_subset(s1_inputs):
    """BASELINE configs[0]: oracle vs the real reference's outputs (first 64 queries, full corpus)."""
    q, d, gold = s1_inputs
    nq = 64
    qh, dh = q[:nq].half().numpy(), d.half().numpy()
    cs, ci = c_oracle.canonical_search(qh, dh, 50)
    compare.assert_bit_exact(cs, ci, gold["canon_h_scores"][:nq], gold["canon_h_ids"][:nq].astype(np.int64),
                             "canonical oracle vs committed golden")
    st = compare.compare_near_tie(cs, ci, gold["ref_h_scores"][:nq], gold["ref_h_ids"][:nq].astype(np.int64),
                                  gap_tol=gap_tolerance(qh, dh), score_tol=1e-3)
    assert st["exact_id_queries"] >= nq - 2, st


def test_config1_golden_reference_statistics(s1_inputs):
    """All 1000 queries: committed canonical vs committed reference outputs obey the near-tie rule."""
    q, d, gold = s1_inputs
    st = compare.compare_near_tie(gold["canon_h_scores"], gold["canon_h_ids"], gold["ref_h_scores"], gold["ref_h_ids"],
                                  gap_tol=gap_tolerance(q.numpy(), d[:2000].numpy()) * 1.2, score_tol=1e-3)
    assert st["exact_id_queries"] >= 990, st
    # raw-fp32 plumbing config: rounding the embeddings to fp16 moves scores by < 0.1 (|score| ~ 100)
    assert np.abs(gold["ref_fp32_scores"] - gold["ref_h_scores"]).max() < 0.5


def test_ref_port_matches_reference_outputs(s1_inputs):
    """The torch port used as cpu_baseline reproduces the real reference's output exactly (200 queries)."""
    import torch
    q, d, gold = s1_inputs
    qh, dh = q[:200].half().float(), d.half().float()
    s, i = ref_port.load_collection_and_retrieve(qh, list(torch.split(dh, [50000, 50000])), 50, 100000)
    assert np.array_equal(i.numpy(), gold["ref_h_ids"][:200].astype(np.int64))
    assert np.array_equal(s.numpy(), gold["ref_h_scores"][:200])


def test_cosine_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cosine_small.npz"))
    q, x, k = g["q"], g["x"], int(g["k"])
    xn, qn = c_oracle.l2_normalize_rows(x), c_oracle.l2_normalize_rows(q)
    assert np.array_equal(xn.view(np.uint16), numpy_oracle.l2_normalize_rows(x).view(np.uint16))
    cs, ci = c_oracle.canonical_search(qn, xn, k)
    compare.assert_bit_exact(cs, ci, g["canon_scores"], g["canon_ids"], "cosine canonical")
    # vs the real reference's CosineSim (fp32): normalised-fp16 storage moves scores by ~1e-4
    st = compare.compare_near_tie(cs, ci, g["ref_scores"], g["ref_ids"], gap_tol=5e-4, score_tol=1e-3)
    assert st["max_score_err"] < 1e-3


def test_incomplete_index_raises_like_reference():
    q = np.zeros((2, 8), np.float32)
    x = np.zeros((10, 8), np.float32)
    with pytest.raises(IOError, match=r"!!! Index is not complete. Please re-index. Missing 3 documents"):
        c_oracle.ref_chunked_search(q, x, [4, 6], 13, 2)
    with pytest.raises(IOError, match=r"Missing 3 documents"):
        numpy_oracle.ref_chunked_search(q, x, [4, 6], 13, 2)


def test_merge_topk_oracle_properties():
    rng = np.random.default_rng(3)
    q = rng.standard_normal((6, 32)).astype(np.float16)
    x = rng.standard_normal((999, 32)).astype(np.float16)
    k = 10
    full_s, full_i = c_oracle.canonical_search(q, x, k)
    for shards in (2, 3, 8):
        bounds = np.linspace(0, 999, shards + 1).astype(int)
        ps, pi = [], []
        for a, b in zip(bounds[:-1], bounds[1:]):
            s, i = c_oracle.canonical_search(q, x[a:b], k, id_offset=int(a))
            ps.append(s)
            pi.append(i)
        ms, mi = c_oracle.merge_topk(np.stack(ps), np.stack(pi))
        compare.assert_bit_exact(ms, mi, full_s, full_i, f"{shards} shards")
        ns, ni = numpy_oracle.merge_topk(np.stack(ps), np.stack(pi))
        compare.assert_bit_exact(ns, ni, full_s, full_i, f"numpy merge {shards} shards")


def test_reference_chunk_sizes():
    # first chunk 293 batches, later 292, last = remainder
    sizes = ref_port.reference_chunk_sizes(21_000_000, 512)
    assert sizes[0] == 293 * 512 and sizes[1] == 292 * 512 and sum(sizes) == 21_000_000
    assert ref_port.reference_chunk_sizes(30, 4, 12) == [16, 12, 2]   # matches tests/golden/ref_index
    assert ref_port.reference_chunk_sizes(5, 128) == [5]
