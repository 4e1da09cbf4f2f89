"""
Comparator between two top-k results.  TEST INFRASTRUCTURE ONLY.

Two modes:
  * ``assert_bit_exact``   ids AND fp32 score bits identical — the bar between the HIP path and
                           the canonical oracle (integer/index work is bit-exact).
  * ``compare_near_tie``   against the reference's own code (torch.mm fp32 + torch.topk), whose
                           summation order differs and whose tie order is arbitrary:
                           ids must be equal at every rank whose neighbouring score gaps exceed
                           ``gap_tol``; inside a near-tie group (consecutive scores closer than
                           gap_tol) only the SET of ids must agree; scores within ``score_tol``
                           (north star: 1e-3).
"""
import numpy as np


def assert_bit_exact(got_scores, got_ids, want_scores, want_ids, what=""):
    got_ids = np.asarray(got_ids)
    want_ids = np.asarray(want_ids)
    assert got_ids.shape == want_ids.shape, f"{what}: shape {got_ids.shape} vs {want_ids.shape}"
    bad = np.argwhere(got_ids != want_ids)
    assert bad.size == 0, (f"{what}: {len(bad)} id mismatches, first at {tuple(bad[0])}: "
                           f"got {got_ids[tuple(bad[0])]} want {want_ids[tuple(bad[0])]}")
    g = np.asarray(got_scores, np.float32).view(np.uint32)
    w = np.asarray(want_scores, np.float32).view(np.uint32)
    bad = np.argwhere(g != w)
    assert bad.size == 0, (f"{what}: {len(bad)} score-bit mismatches, first at {tuple(bad[0])}: "
                           f"got {np.asarray(got_scores)[tuple(bad[0])]!r} want {np.asarray(want_scores)[tuple(bad[0])]!r}")


def compare_near_tie(got_scores, got_ids, ref_scores, ref_ids, gap_tol, score_tol=1e-3):
    """Returns a dict of statistics; raises AssertionError when the rule is violated."""
    got_scores = np.asarray(got_scores, np.float64)
    ref_scores = np.asarray(ref_scores, np.float64)
    got_ids = np.asarray(got_ids)
    ref_ids = np.asarray(ref_ids)
    assert got_ids.shape == ref_ids.shape
    nq, k = got_ids.shape
    exact_queries = 0
    tie_swaps = 0
    boundary_diffs = 0
    max_score_err = 0.0
    for i in range(nq):
        if np.array_equal(got_ids[i], ref_ids[i]):
            exact_queries += 1
        err = np.abs(got_scores[i] - ref_scores[i])
        err = err[np.isfinite(err)]
        if err.size:
            max_score_err = max(max_score_err, float(err.max()))
        # near-tie groups over the reference scores
        j = 0
        while j < k:
            e = j + 1
            while e < k and abs(ref_scores[i, e - 1] - ref_scores[i, e]) <= gap_tol:
                e += 1
            gs, rs = set(got_ids[i, j:e].tolist()), set(ref_ids[i, j:e].tolist())
            if gs != rs:
                # the only legitimate difference: the group touches the cut at rank k, where a
                # near-tied candidate just outside one list may be just inside the other
                touches_cut = (e == k)
                assert touches_cut, (f"query {i}: ids differ in ranks [{j},{e}) outside a near-tie group "
                                     f"at the cut: got {sorted(gs - rs)} vs ref {sorted(rs - gs)}")
                # the swapped-in candidates must themselves be near-tied with the k-th score
                lo = ref_scores[i, k - 1]
                for pos in range(j, e):
                    assert abs(got_scores[i, pos] - lo) <= 4 * gap_tol + score_tol, (
                        f"query {i}: rank {pos} differs from the reference and is not a near tie")
                boundary_diffs += 1
            elif not np.array_equal(got_ids[i, j:e], ref_ids[i, j:e]):
                tie_swaps += 1
            j = e
    assert max_score_err <= score_tol, f"max |score - ref| = {max_score_err} > {score_tol}"
    return {"queries": nq, "exact_id_queries": exact_queries, "near_tie_reorders": tie_swaps,
            "cut_boundary_diffs": boundary_diffs, "max_score_err": max_score_err}
