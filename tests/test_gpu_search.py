"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the
committed golden fixtures.  Bit-exact for ids and canonical fp32 scores."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, compare
from tests.conftest import gap_tolerance

pytestmark = pytest.mark.gpu

SPLIT_DEFAULT = 1  # library default of "query_split" (restored after the option tests)
KERNEL_DEFAULT = 3  # library default of "scan_kernel" (256-query tile where it applies, else the 4-wave kernel)


@pytest.fixture(scope="module")
def amd():
    import bergen_amd
    from bergen_amd import _lib
    _lib.init(0)
    yield bergen_amd
    for name, val in (("query_tile", 128), ("share_threshold", 1), ("nontemporal", 1), ("query_split", SPLIT_DEFAULT),
                      ("scan_kernel", KERNEL_DEFAULT)):
        _lib.set_option(name, val)


@pytest.fixture
def no_tail_routing():
    """Tests written for the 256-query kernel with small query sets: keep a last pass of <= 128 queries on that kernel
    (the default routes it to the 128-query kernel; tests/test_gpu_search.py::test_tail_pass_on_the_128_query_kernel)."""
    from bergen_amd import _lib
    _lib.set_option("tail128", 0)
    yield
    _lib.set_option("tail128", 1)


def _search(amd, x, q, k, metric="ip", chunks=None):
    ix = amd.FlatIndex(x.shape[0], x.shape[1], metric=metric)
    try:
        if chunks is None:
            ix.upload(x)
        else:
            off = 0
            for c in chunks:
                ix.upload(x[off:off + c])
                off += c
        ix.finalize()
        return ix.search(q, k)
    finally:
        ix.close()


def test_kat_small_golden(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_small.npz"))
    s, i = _search(amd, g["x"], g["q"], int(g["k"]), chunks=[9, 8])
    compare.assert_bit_exact(s, i, g["canon_scores"], g["canon_ids"], "kat_small")
    assert np.array_equal(s, g["ref_scores"])  # the reference's scores, canonical order inside ties


@pytest.mark.parametrize("n,d,nq,k", [
    (1, 64, 1, 1), (31, 64, 3, 5), (32, 64, 128, 10), (33, 100, 129, 50), (1000, 128, 7, 56),
    (4097, 256, 130, 57), (3000, 384, 40, 120), (2500, 512, 33, 121), (7777, 768, 300, 50),
    (5000, 1024, 65, 200), (2000, 1024, 10, 248), (600, 16, 4, 50), (20000, 768, 64, 50),
])
def test_random_matches_oracle(amd, n, d, nq, k):
    rng = np.random.default_rng(n * 31 + d)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    s, i = _search(amd, x, q, k)
    ws, wi = c_oracle.canonical_search(q, x, k)
    compare.assert_bit_exact(s, i, ws, wi, f"n={n} d={d} nq={nq} k={k}")


def test_fp32_sources_are_rounded_like_torch_half(amd):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3000, 768)).astype(np.float32)
    q = rng.standard_normal((20, 768)).astype(np.float32)
    s, i = _search(amd, x, q, 50)
    ws, wi = c_oracle.canonical_search(q.astype(np.float16), x.astype(np.float16), 50)
    compare.assert_bit_exact(s, i, ws, wi, "fp32 sources")


def test_exact_ties_and_degenerate_corpora(amd):
    d = 64
    # (a) 3000 identical rows: every score ties -> ascending row ids, compaction on every buffer fill
    x = np.tile(np.arange(d, dtype=np.float16)[None] / 8, (3000, 1))
    q = np.ones((5, d), np.float16)
    s, i = _search(amd, x, q, 50)
    assert np.array_equal(i, np.tile(np.arange(50), (5, 1)))
    # (b) all-zero corpus and zero queries
    s, i = _search(amd, np.zeros((500, d), np.float16), np.zeros((3, d), np.float16), 20)
    assert np.array_equal(i, np.tile(np.arange(20), (3, 1))) and (s == 0).all()
    # (c) adversarial order: scores strictly increase with the row index, so EVERY row is a new best
    #     (worst case for the threshold filter: maximal append + compaction traffic)
    n = 40000
    x = np.zeros((n, d), np.float16)
    x[:, 0] = (np.arange(n) % 2048).astype(np.float16)     # fp16-exact integers
    x[:, 1] = (np.arange(n) // 2048).astype(np.float16)
    q = np.zeros((130, d), np.float16)
    q[:, 0] = 1.0
    q[:, 1] = 2048.0
    q[64:, :2] *= -1                                        # second half: strictly DEcreasing scores
    s, i = _search(amd, x, q, 50)
    ws, wi = c_oracle.canonical_search(q, x, 50)
    compare.assert_bit_exact(s, i, ws, wi, "monotone scores")
    assert i[0, 0] == n - 1 and i[64, 0] == 0
    # (d) fewer rows than k
    rng = np.random.default_rng(2)
    x = rng.standard_normal((7, d)).astype(np.float16)
    q = rng.standard_normal((2, d)).astype(np.float16)
    s, i = _search(amd, x, q, 10)
    ws, wi = c_oracle.canonical_search(q, x, 10)
    compare.assert_bit_exact(s, i, ws, wi, "short index")
    assert (i[:, 7:] == -1).all() and np.isneginf(s[:, 7:]).all()
    # (e) -inf scores / huge magnitudes stay ordered
    x = rng.standard_normal((300, d)).astype(np.float16)
    x[5] = 60000
    x[6] = -60000
    q = np.full((1, d), 60000, np.float16)
    s, i = _search(amd, x, q, 10)
    ws, wi = c_oracle.canonical_search(q, x, 10)
    compare.assert_bit_exact(s, i, ws, wi, "overflowing scores")


def test_config1_golden(amd, s1_inputs):
    """BASELINE configs[0] on the device: bit-exact vs the committed canonical golden; near-tie rule and
    1e-3 score tolerance vs the committed outputs of the REAL reference code."""
    q, d, gold = s1_inputs
    qh, dh = q.half().numpy(), d.half().numpy()
    s, i = _search(amd, dh, qh, 50, chunks=[50000, 50000])
    compare.assert_bit_exact(s, i, gold["canon_h_scores"], gold["canon_h_ids"].astype(np.int64), "config1 canonical")
    st = compare.compare_near_tie(s, i, gold["ref_h_scores"], gold["ref_h_ids"].astype(np.int64),
                                  gap_tol=gap_tolerance(qh, dh[:4000]) * 1.2, score_tol=1e-3)
    assert st["exact_id_queries"] >= 990, st


def test_options_do_not_change_results(amd):
    from bergen_amd import _lib
    rng = np.random.default_rng(8)
    x = rng.standard_normal((30011, 768)).astype(np.float16)
    q = rng.standard_normal((300, 768)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q[:40], x, 50)
    base = None
    try:
        for split in (1, 2):
            for tile in (128, 256):
                for share in (0, 1):
                    for nt in (0, 1):
                        _lib.set_option("query_split", split)
                        _lib.set_option("query_tile", tile)
                        _lib.set_option("share_threshold", share)
                        _lib.set_option("nontemporal", nt)
                        s, i = _search(amd, x, q, 50)
                        what = f"split={split} tile={tile} share={share} nt={nt}"
                        compare.assert_bit_exact(s[:40], i[:40], ws, wi, what)
                        if base is None:
                            base = (s, i)
                        compare.assert_bit_exact(s, i, base[0], base[1], "variant " + what)
    finally:
        _lib.set_option("query_tile", 128)
        _lib.set_option("share_threshold", 1)
        _lib.set_option("nontemporal", 1)
        _lib.set_option("query_split", SPLIT_DEFAULT)


@pytest.mark.parametrize("nq", [129, 256, 257, 385, 512])
def test_query_split_pass_boundaries(amd, nq):
    """query_split = 2: launches of 256 queries by paired workgroups, the last <= 128 queries unsplit."""
    from bergen_amd import _lib
    rng = np.random.default_rng(nq)
    x = rng.standard_normal((9001, 768)).astype(np.float16)
    q = rng.standard_normal((nq, 768)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, 50)
    try:
        _lib.set_option("query_split", 2)
        s, i = _search(amd, x, q, 50)
        compare.assert_bit_exact(s, i, ws, wi, f"split nq={nq}")
    finally:
        _lib.set_option("query_split", SPLIT_DEFAULT)


def test_shard_invariance_and_device_merge(amd):
    """Results identical for 1/2/4/8 row shards merged by the HIP merge kernel (SURVEY §8c KAT 3)."""
    rng = np.random.default_rng(4)
    n, d, nq, k = 10007, 384, 50, 50
    x = rng.standard_normal((n, d)).astype(np.float16)
    x[9000] = x[17]
    q = rng.standard_normal((nq, d)).astype(np.float16)
    full_s, full_i = _search(amd, x, q, k)
    ws, wi = c_oracle.canonical_search(q, x, k)
    compare.assert_bit_exact(full_s, full_i, ws, wi, "single shard")
    for shards in (2, 4, 8):
        ps, pi = [], []
        for r in range(shards):
            lo, hi = amd.shard_range(n, r, shards)
            ix = amd.FlatIndex(hi - lo, d)
            ix.upload(x[lo:hi])
            ix.finalize()
            s, i = ix.search(q, k, id_offset=lo)
            ix.close()
            ps.append(s)
            pi.append(i)
        ms, mi = amd.merge_topk(np.stack(ps), np.stack(pi))
        compare.assert_bit_exact(ms, mi, full_s, full_i, f"{shards} shards (host buffers)")
        ms, mi = amd.merge_topk(torch.from_numpy(np.stack(ps)).cuda(), torch.from_numpy(np.stack(pi)).cuda())
        compare.assert_bit_exact(ms.cpu().numpy(), mi.cpu().numpy(), full_s, full_i, f"{shards} shards (device)")
    # many lists (tree reduction path) + invalid entries
    many_s = np.stack([full_s] + [np.full_like(full_s, -np.inf)] * 99)
    many_i = np.stack([full_i] + [np.full_like(full_i, -1)] * 99)
    ms, mi = amd.merge_topk(many_s, many_i)
    compare.assert_bit_exact(ms, mi, full_s, full_i, "100 lists")


def test_cosine_golden(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "cosine_small.npz"))
    s, i = _search(amd, g["x"], g["q"], int(g["k"]), metric="cos")
    compare.assert_bit_exact(s, i, g["canon_scores"], g["canon_ids"], "cosine canonical")
    st = compare.compare_near_tie(s, i, g["ref_scores"], g["ref_ids"], gap_tol=5e-4, score_tol=1e-3)
    assert st["max_score_err"] < 1e-3


def test_device_resident_sources_and_queries(amd):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((6000, 768)).astype(np.float16)
    q = rng.standard_normal((70, 768)).astype(np.float16)
    ix = amd.FlatIndex(6000, 768)
    ix.upload(torch.from_numpy(x[:2500]).cuda())           # fp16 device rows
    ix.upload(torch.from_numpy(x[2500:]).cuda().float())   # fp32 device rows (exactly representable)
    ix.finalize()
    s, i = ix.search(torch.from_numpy(q).cuda(), 50, id_offset=1_000_000)
    assert s.is_cuda and i.is_cuda and i.dtype == torch.int64
    ws, wi = c_oracle.canonical_search(q, x, 50, id_offset=1_000_000)
    compare.assert_bit_exact(s.cpu().numpy(), i.cpu().numpy(), ws, wi, "device path")
    c = ix.counters()
    tile = c["query_tile"]  # 128: a search of at most 128 queries runs on the 128-query kernel whatever the default
    assert c["n_passes"] == 1 and tile == 128 and c["scan_ms"] > 0
    assert c["algorithmic_bytes"] == 6000 * 768 * 2 + tile * 768 * 2 + tile * 50 * 12
    ix.close()


def test_error_behaviour_matches_reference(amd):
    x = np.zeros((100, 64), np.float16)
    ix = amd.FlatIndex(100, 64)
    ix.upload(x[:60])
    with pytest.raises(IOError, match=r"!!! Index is not complete. Please re-index. Missing 40 documents in the index. !!!"):
        ix.finalize()
    with pytest.raises(IOError, match=r"Missing 40 documents"):
        ix.search(np.zeros((1, 64), np.float16), 5)
    with pytest.raises(ValueError):
        ix.upload(x, row0=50)                     # rows beyond the index
    ix.upload(x[60:])
    ix.finalize()
    with pytest.raises(Exception):
        ix.search(np.zeros((1, 64), np.float16), 5000)   # k > 4096 unsupported
    s, i = ix.search(np.zeros((0, 64), np.float16), 5)
    assert s.shape == (0, 5)
    ix.close()
    with pytest.raises(Exception):
        amd.FlatIndex(10, 5000)                   # dim > 1024 unsupported


def test_repeatability(amd):
    rng = np.random.default_rng(12)
    x = rng.standard_normal((50000, 768)).astype(np.float16)
    q = rng.standard_normal((256, 768)).astype(np.float16)
    ix = amd.FlatIndex(50000, 768)
    ix.upload(x)
    ix.finalize()
    a = ix.search(q, 50)
    for _ in range(3):
        b = ix.search(q, 50)
        compare.assert_bit_exact(b[0], b[1], a[0], a[1], "run-to-run")
    ix.close()


def _full_size_case(amd, n, d, nq, k, metric, checked, seed):
    """One full-size corpus, generated on the device in 1 M-row blocks and streamed through the oracle as it is generated.

    Size-independent properties: planted positives come back on top with their oracle scores, lists are canonically sorted,
    and a 2-way row split merged by the HIP kernel equals the single-index result.
    And the FULL lists of the `checked` queries against the oracle, bit for bit: every corpus block is copied to the host as
    it is generated, the oracle keeps a running top-k per block (`c_oracle.canonical_search` with the block's row offset;
    cosine: on the block normalised by the oracle's own `l2_normalize_rows`) and the per-block lists are merged by the
    oracle's own merge — ranks 1..k at full size are then the oracle's word, not the kernel's certificate vouching for
    itself.  `checked` names queries of EVERY pass of the search, the last (tail) pass included."""
    free, total = torch.cuda.mem_get_info()
    if free < (n * d * 2) * 2.2:
        pytest.skip("not enough free HBM for the full-size case")
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(seed)
    cos = metric == "cos"
    ix = amd.FlatIndex(n, d, metric=metric)
    half_lo = amd.FlatIndex(n // 2, d, metric=metric)
    half_hi = amd.FlatIndex(n - n // 2, d, metric=metric)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=gen, device=dev), dim=1)
    if cos:
        q = q * (0.5 + 3.0 * torch.rand(nq, 1, generator=gen, device=dev))  # cosine ignores the length: make it matter if it does not
    q = q.half()
    plant_rows = torch.randint(0, n, (nq, 5), generator=gen, device=dev)
    block = 1_000_000
    planted, row_owner = {}, {}
    checked = sorted(set(int(c) for c in checked))
    q_np = q.cpu().numpy()
    q_canon = c_oracle.l2_normalize_rows(q_np) if cos else q_np  # what the index scores with (bh_search normalises the queries)
    q_head = np.ascontiguousarray(q_canon[checked])
    part_s, part_i = [], []
    for b0 in range(0, n, block):
        m = min(block, n - b0)
        rows = torch.nn.functional.normalize(torch.randn(m, d, generator=gen, device=dev), dim=1)
        sel = ((plant_rows >= b0) & (plant_rows < b0 + m)).nonzero()
        for qi, j in sel.tolist():
            r = int(plant_rows[qi, j])
            noise = torch.randn(d, generator=gen, device=dev) * (0.3 / d ** 0.5)
            rows[r - b0] = torch.nn.functional.normalize(q[qi].float() + noise, dim=0)
        if cos:  # row lengths all over the place: the finalize-time normalisation has real work to do
            rows = rows * (0.25 + 4.0 * torch.rand(m, 1, generator=gen, device=dev))
        rows = rows.half()
        host = rows.cpu().numpy()
        canon = c_oracle.l2_normalize_rows(host) if cos else host
        for qi, j in sel.tolist():  # (two queries may plant on the same row: the later write is what the row holds)
            r = int(plant_rows[qi, j])
            if r in row_owner:
                del planted[(row_owner[r], r)]
            row_owner[r] = qi
            planted[(qi, r)] = canon[r - b0].copy()
        ix.upload(rows, row0=b0)
        bs, bi = c_oracle.canonical_search(q_head, canon, k, id_offset=b0)  # the streaming oracle's block step
        part_s.append(bs)
        part_i.append(bi)
        lo_n = n // 2
        if b0 + m <= lo_n:
            half_lo.upload(rows, row0=b0)
        elif b0 >= lo_n:
            half_hi.upload(rows, row0=b0 - lo_n)
        else:
            cut = lo_n - b0
            half_lo.upload(rows[:cut].contiguous(), row0=b0)
            half_hi.upload(rows[cut:].contiguous(), row0=0)
        del rows, host, canon
    for h in (ix, half_lo, half_hi):
        h.finalize()
    s, i = ix.search(q, k)
    counters = ix.counters()
    s_np, i_np = s.cpu().numpy(), i.cpu().numpy()
    # canonical sortedness
    assert (np.diff(s_np, axis=1) <= 0).all()
    ties = np.diff(s_np, axis=1) == 0
    assert (np.diff(i_np, axis=1)[ties] > 0).all()
    # planted positives: present, and their scores equal the oracle's canonical score
    for (qi, r), row in planted.items():
        pos = np.where(i_np[qi] == r)[0]
        assert len(pos) == 1, f"planted row {r} of query {qi} missing"
        want = c_oracle.canonical_scores(q_canon[qi:qi + 1], row[None], np.zeros((1, 1), np.int64))[0, 0]
        assert s_np[qi, pos[0]].view(np.uint32) == want.view(np.uint32)
    for qi in range(nq):
        mine = sorted(r for (a, r) in planted if a == qi)
        assert sorted(i_np[qi, :len(mine)].tolist()) == mine
    # the complete top-k lists of the checked queries vs the streaming oracle
    ws, wi = c_oracle.merge_topk(np.stack(part_s), np.stack(part_i))
    assert wi.min() >= 0 and len(set(wi[0].tolist())) == k
    compare.assert_bit_exact(s_np[checked], i_np[checked], ws, wi, f"full lists at {n} x {d} ({metric}, top-{k}) vs the streaming oracle")
    # shard invariance at full size
    s1, i1 = half_lo.search(q, k, id_offset=0)
    s2, i2 = half_hi.search(q, k, id_offset=n // 2)
    ms, mi = amd.merge_topk(torch.stack([s1, s2]), torch.stack([i1, i2]))
    compare.assert_bit_exact(ms.cpu().numpy(), mi.cpu().numpy(), s_np, i_np, "2 shards at full size")
    for h in (ix, half_lo, half_hi):
        h.close()
    return counters


def test_full_size_properties(amd):
    """BASELINE configs[1] size: 21M x 768 fp16 resident (32 GB), top-50; 612 queries = two passes of the 256-query kernel and
    a 100-query TAIL pass on the 128-query kernel.  Complete oracle lists of 32 queries: 12 of pass 0, 10 of pass 1 (both ends
    and the middle of each tile), 10 of the tail pass."""
    checked = [0, 1, 2, 3, 100, 101, 127, 128, 200, 253, 254, 255,          # pass 0
               256, 257, 258, 300, 383, 384, 400, 509, 510, 511,            # pass 1
               512, 513, 514, 550, 575, 576, 600, 609, 610, 611]            # tail pass (128-query kernel)
    c = _full_size_case(amd, 21_000_000, 768, 612, 50, "ip", checked, seed=1234)
    assert c["n_passes"] == 3 and c["query_tile"] == 256 and c["tail_query_tile"] == 128 and c["tail_scan_ms"] > 0


def test_full_size_config5_cosine(amd):
    """BASELINE configs[4] at its stated size: 21 M x 1024 fp16 (43 GB), top-200, COSINE (config/retriever/e5-large-v2.yaml:
    similarity CosineSim) — the d = 1024 instantiation of the scan (128 queries per pass, static tile distribution, candidate
    lists of 256) and the finalize-time row normalisation, both at full size.  300 queries = three passes; complete oracle
    lists of 14 queries from all three."""
    checked = [0, 1, 2, 64, 127, 128, 129, 200, 255, 256, 257, 280, 298, 299]
    c = _full_size_case(amd, 21_000_000, 1024, 300, 200, "cos", checked, seed=4321)
    assert c["n_passes"] == 3 and c["query_tile"] == 128 and c["k_padded"] == 256 and c["dim_padded"] == 1024


@pytest.mark.parametrize("kern", [0, 2, 3])
@pytest.mark.parametrize("n,nq,k", [(33, 1, 5), (9001, 191, 50), (9001, 192, 50), (9001, 193, 50), (70001, 400, 50), (12345, 600, 56),
                                    (9001, 255, 50), (9001, 256, 50), (9001, 257, 50)])
def test_query_tile_kernels_match_oracle(amd, no_tail_routing, kern, n, nq, k):
    """scan_kernel 0 (128-query tile, 32x32x16 MFMA, one wave per SIMD), 2 (192-query tile, 16x16x32 MFMA, one wave per SIMD)
    and 3 (256-query tile, 16x16x32 MFMA, two waves per SIMD) at d = 768: same bit-exact results, including the pass
    boundaries of the 192- and 256-query tiles."""
    from bergen_amd import _lib
    rng = np.random.default_rng(n + nq)
    x = rng.standard_normal((n, 768)).astype(np.float16)
    q = rng.standard_normal((nq, 768)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    try:
        _lib.set_option("scan_kernel", kern)
        ix = amd.FlatIndex(n, 768, metric="ip")
        ix.upload(x)
        ix.finalize()
        s, i = ix.search(q, k)
        tile = ix.counters()["query_tile"]
        ix.close()
        assert tile == {0: 128, 2: 192, 3: 256}[kern]
        compare.assert_bit_exact(s, i, ws, wi, f"kernel {kern} n={n} nq={nq} k={k}")
    finally:
        _lib.set_option("scan_kernel", KERNEL_DEFAULT)


@pytest.mark.parametrize("n,d,nq,k", [(30001, 384, 300, 50), (30001, 512, 300, 50), (30001, 500, 100, 120), (40001, 768, 300, 100),
                                      (40001, 768, 260, 200), (20001, 384, 257, 248), (300001, 768, 64, 50)])
def test_tile256_kernel_dims_and_list_lengths(amd, no_tail_routing, n, d, nq, k):
    """The 256-query kernel (default) at every dim it serves (384 / 512 / 768) and every candidate-list length (64 / 128 / 256),
    against the 4-wave kernel, whose results the oracle tests above pin."""
    from bergen_amd import _lib
    rng = np.random.default_rng(n + d + k)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    s, i = ix.search(q, k)
    tile = ix.counters()["query_tile"]
    ix.close()
    assert tile == 256
    compare.assert_bit_exact(s, i, ws, wi, f"256-query kernel n={n} d={d} nq={nq} k={k}")


@pytest.mark.parametrize("n,d,nq,k", [(300001, 768, 40, 50), (300001, 384, 40, 50), (300001, 512, 40, 120), (70001, 384, 300, 50),
                                      (70001, 512, 100, 200), (100003, 768, 257, 56), (340000, 768, 30, 200), (299617, 768, 64, 50),
                                      (300001, 1024, 40, 200), (320017, 1024, 130, 50)])
def test_dynamic_tile_distribution_matches_oracle(amd, no_tail_routing, n, d, nq, k):
    """scan_topk256's dynamic tile distribution (option dyn_tiles, default on): the first 7/8 of the corpus round robin and
    the tail in runs claimed from the pass's counter (>= 300 k rows on 256 workgroups; smaller corpora stay round robin).
    Every tile must be scanned exactly once whatever the claim order: bit-exact against the oracle with the option on and
    off, at every ring geometry that has it (d = 384 / 512 / 768) and with 64 / 128-entry candidate lists."""
    from bergen_amd import _lib
    rng = np.random.default_rng(n + d + k)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    # d = 1024: the four-stage ring with the claim word (ring_variant 5; the default of a shard of < 8 M rows) and the five-stage
    # ring that fills the LDS and stays static (ring_variant 7; the default of a whole corpus) — both on this corpus
    rings = (0, 5, 7) if d == 1024 else (0,)
    try:
        for ring in rings:
            _lib.set_option("ring_variant", ring)
            for dyn in (1, 0, 1):
                _lib.set_option("dyn_tiles", dyn)
                s, i = ix.search(q, k)
                compare.assert_bit_exact(s, i, ws, wi, f"ring_variant={ring} dyn_tiles={dyn} n={n} d={d} nq={nq} k={k}")
    finally:
        _lib.set_option("dyn_tiles", 1)
        _lib.set_option("ring_variant", 0)
        ix.close()


@pytest.mark.parametrize("n,d,nq,k", [(300001, 768, 600, 50), (70001, 768, 1300, 50), (340000, 768, 520, 10), (4999, 768, 512, 50),
                                      (100003, 768, 530, 100), (100003, 768, 515, 200), (320017, 1024, 300, 200), (70001, 1024, 650, 50),
                                      (90001, 1024, 257, 100)])
def test_paired_launches_match_oracle(amd, n, d, nq, k):
    """Option pair256 (scan_topk256.hip, ABL bit 128): two 256-query passes per launch, each on half the grid, partner
    workgroups on one XCD walking the same tiles (1 = paced through the progress words, 2 = free-running).  Whatever the
    pacing does, a pass of a paired launch is an ordinary pass on 128 workgroups: bit-exact against the oracle for two pairs +
    the 128-query tail (600), two pairs + an unpaired pass + the tail (1 300), one pair + 8 tail queries with the dynamic
    tile claims of half a grid (340 k rows), and a corpus with fewer tiles than workgroups; lists of 128 and 256; d = 1024 (128-query
    passes on the four-stage ring: three launches of 2 x 128 queries at 300 queries ...)."""
    rng = np.random.default_rng(n + nq + k + d)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    try:
        tile = 256 if d == 768 else 128
        for mode in (1, 2, 0, 3, 1):
            ix.set_option("pair256", mode)
            s, i = ix.search(q, k)
            compare.assert_bit_exact(s, i, ws, wi, f"pair256={mode} n={n} d={d} nq={nq} k={k}")
            assert ix.counters()["n_passes"] == (nq + tile - 1) // tile
    finally:
        ix.close()


@pytest.mark.parametrize("n,d,nq,k", [(300001, 768, 277, 50), (120001, 768, 512 + 277, 50), (70001, 768, 1024 + 300, 50), (70001, 768, 511, 100),
                                      (70001, 768, 257, 200), (4999, 768, 512 + 400, 50), (90001, 1024, 256 + 200, 50), (90001, 1024, 129, 200),
                                      (60001, 768, 600, 50), (60001, 768, 100, 50)])
def test_balanced_remainder_and_idle_waves_match_oracle(amd, n, d, nq, k):
    """Option balance_tail (index.hip, default on): the queries left behind the last full pair of passes — more than one tile, fewer than
    two: 277 of the headline's 2 837 — run as ONE more paired launch of two passes of about half each (the first a whole number of waves'
    queries; the second pass's tile starts where the first one's queries end, BhScanArgs::qtile2), in which the waves that hold no query
    (BhScanArgs::nq_valid) only keep the stage rendezvous and their lines of the refill.  Bit-exact against the oracle with the option
    on and off, for the headline remainder alone and behind one / two full pairs, lists of 128 / 256, d = 1024 (128-query tiles, 16
    queries per wave), a corpus with fewer tiles than workgroups; the counters say which routing ran; and the query counts the option
    does not apply to (an odd number of passes: 600; a single pass: 100 — where the idle waves of an ordinary pass still sit out)."""
    rng = np.random.default_rng(n + nq + k + d)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    x[n // 3] = x[5]  # a tie across workgroups
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    tile = 256 if d == 768 else 128
    n_pass = (nq + tile - 1) // tile
    rem = nq - (n_pass - 2) * tile
    applies = n_pass >= 2 and n_pass % 2 == 0 and tile < rem < 2 * tile
    try:
        for on in (1, 0, 1):
            ix.set_option("balance_tail", on)
            s, i = ix.search(q, k)
            c = ix.counters()
            compare.assert_bit_exact(s, i, ws, wi, f"balance_tail={on} n={n} d={d} nq={nq} k={k}")
            assert c["n_passes"] == n_pass
            if on and applies:
                assert c["balanced_queries"] == rem and c["balanced_scan_ms"] > 0 and c["tail_query_tile"] == 0 and c["paired_launches"] == n_pass // 2
            else:
                assert c["balanced_queries"] == 0 and c["balanced_scan_ms"] == 0
        # device-resident queries and the stage's pinned-host outputs take the same routing
        sd, idd = ix.search(torch.from_numpy(q).cuda(), k)
        compare.assert_bit_exact(sd.cpu().numpy(), idd.cpu().numpy(), ws, wi, "device queries")
    finally:
        ix.close()


@pytest.mark.parametrize("k", [50, 56, 120])
@pytest.mark.parametrize("d", [768, 1024])
def test_certificate_catches_a_near_tie_cluster_at_rank_k(amd, k, d):
    """More rows than the candidate margin (KP - k = 14 / 8 / 8) whose canonical scores TIE the k-th one exactly while
    their fp32 MFMA scores differ in the last bits (same products, summed in another order): the scan's top-KP by MFMA
    score then holds an arbitrary subset of the tied rows, the canonical order wants the lowest row indices.  The
    certificate must flag the query and the exact fall-back (MFMA filter pass + canonical re-scoring) must return the
    oracle's list bit for bit."""
    from bergen_amd import _lib
    rng = np.random.default_rng(1000 + k + d)
    n, cluster, better = 4000, 40, k - 15
    s = rng.choice(np.array([-1.0, 1.0]), size=d)
    v = (rng.standard_normal(d) * 0.5).astype(np.float16).astype(np.float64)
    q0 = (0.5 * s).astype(np.float16)                                   # |q_i| constant: q . (s * perm(v)) = 0.5 sum(v) for every perm
    x = np.empty((n, d), np.float16)
    for r in range(n):                                                    # the crowd: clearly lower scores
        x[r] = (s * (rng.permutation(v) - 0.25 * rng.random(d))).astype(np.float16)
    special = rng.choice(n, size=cluster + better, replace=False)
    for r in special[:cluster]:                                           # the tie cluster: the same multiset of products
        x[r] = (s * rng.permutation(v)).astype(np.float16)
    for r in special[cluster:]:                                           # clearly better rows
        x[r] = (s * (rng.permutation(v) + 0.25)).astype(np.float16)
    q = np.concatenate([q0[None, :], rng.standard_normal((6, d)).astype(np.float16)])
    ws, wi = c_oracle.canonical_search(q, x, k)
    tied = set(int(r) for r in special[:cluster])
    assert len(tied & set(wi[0].tolist())) == 15 and len(set(np.float32(ws[0, -15:]).tolist())) == 1   # the construction works
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    try:
        got_s, got_i = ix.search(q, k)
        c = ix.counters()
        compare.assert_bit_exact(got_s, got_i, ws, wi, f"near-tie cluster k={k} d={d}")
        assert c["uncertified_queries"] >= 1 and c["exact_ms"] > 0 and c["exact_passes"] == 1
        # an ordinary workload is certified from the candidate lists alone
        got_s, got_i = ix.search(q[1:], k)
        assert ix.counters()["uncertified_queries"] == 0
        compare.assert_bit_exact(got_s, got_i, ws[1:], wi[1:], "ordinary queries")
        # without the certificate the library still answers (results for the adversarial query are not guaranteed)
        _lib.set_option("certify", 0)
        ix.search(q, k)
        assert ix.counters()["uncertified_queries"] == 0
    finally:
        _lib.set_option("certify", 1)
        ix.close()


@pytest.mark.parametrize("filter256", [1, 0])
@pytest.mark.parametrize("n,d,nq,k", [(30_011, 768, 300, 50), (9_000, 1024, 130, 200), (5_000, 384, 70, 100), (3_000, 100, 40, 10),
                                      (2_000, 64, 9, 5), (4_000, 512, 129, 120), (4_001, 200, 5, 248), (20_000, 768, 600, 50),
                                      (6_000, 512, 257, 50)])
def test_fall_back_filter_pass_is_exact_for_every_query(amd, n, d, nq, k, filter256):
    """The exact fall-back on its own: with the certificate's error bound scaled up (test option
    certificate_error_scale) NO query can be certified, so every result comes from the MFMA filter pass (fixed threshold =
    k-th canonical score - bound) + canonical re-scoring of the rows it lets through + the host sort.  The filter pass takes
    256 queries on scan_topk256.hip (ABL bit 64) at d = 384 / 512 / 768 (option filter256, default) and 128 on scan_topk.hip
    (ABL 5) elsewhere or with the option off.  Bit-exact against the oracle at every padded dim, across the batches of the
    fall-back (600 queries = three batches of 256 / five of 128), with exact duplicates straddling the k-th rank."""
    from bergen_amd import _lib
    ftile = 256 if (filter256 and d in (384, 512, 768)) else 128
    rng = np.random.default_rng(n + d + k)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    x[n // 2: n // 2 + 30] = x[7]          # 31 exact duplicates: ties decided by row index
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    ix.set_option("filter256", filter256)
    try:
        _lib.set_option("certificate_error_scale", 32)   # a bound a little wider than the gaps: most queries, short lists
        s, i = ix.search(q, k)
        c = ix.counters()
        assert c["uncertified_queries"] <= nq and c["exact_passes"] == -(-c["uncertified_queries"] // ftile)  # (none on a small corpus: wide gaps)
        compare.assert_bit_exact(s, i, ws, wi, f"fall-back for most queries n={n} d={d} nq={nq} k={k}")
        _lib.set_option("certificate_error_scale", 1 << 12)  # a bound wider than the score range: every query, every row listed
        s, i = ix.search(q, k)
        c = ix.counters()
        # (a query whose per-workgroup lists never filled is certified whatever the bound — nothing was dropped —, and at small
        # d the scaled bound still sits below the score gaps: the "every query" claim is for the wide shapes)
        assert c["exact_passes"] == -(-c["uncertified_queries"] // ftile)
        if d >= 200:
            assert c["uncertified_queries"] == nq and c["exact_ms"] > 0 and c["exact_rows_rescored"] >= nq * k
        compare.assert_bit_exact(s, i, ws, wi, f"fall-back only n={n} d={d} nq={nq} k={k}")
        # device queries / device results, with a global row offset
        import torch
        sd, idd = ix.search(torch.from_numpy(q).cuda(), k, id_offset=1_000_000)
        compare.assert_bit_exact(sd.cpu().numpy(), idd.cpu().numpy() - 1_000_000, ws, wi, "fall-back only, device path")
        _lib.set_option("certificate_error_scale", 1)
        s, i = ix.search(q, k)
        assert ix.counters()["exact_passes"] == 0 or ix.counters()["uncertified_queries"] > 0
        compare.assert_bit_exact(s, i, ws, wi, "certificate back on")
    finally:
        _lib.set_option("certificate_error_scale", 1)
        ix.close()


@pytest.mark.parametrize("n,d,nq,k,cluster", [(50_000, 768, 40, 300, 0), (30_000, 384, 9, 1000, 0), (20_000, 128, 5, 2500, 0),
                                              (40_000, 768, 7, 600, 700), (900, 64, 3, 1000, 0), (5_000, 1024, 130, 249, 0)])
def test_large_k_is_searched_range_by_range(amd, n, d, nq, k, cluster):
    """k > 248 (the reference accepts any top_k_documents, modules/retrieve.py:157): the corpus is cut into row ranges, every
    range gives its exact top 248, the lists are merged, and a range that may have dropped a member of the top k is split and
    searched again (index.hip: search_large_k).  `cluster` consecutive copies of a row that scores at the very top of query 0
    put far more than 248 of its top k into ONE range: the split path must still return the oracle's list, ties by row index.
    Also k larger than the corpus (short lists padded with -inf / -1), host and device entry points."""
    import torch
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    if cluster:
        x[n // 3: n // 3 + cluster] = (q[0].astype(np.float32) * 0.5).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, min(k, n))
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    s, i = ix.search(q, k, id_offset=7)
    c = ix.counters()
    assert c["n_passes"] >= 2 and c["scan_ms"] > 0
    kk = min(k, n)
    compare.assert_bit_exact(s[:, :kk], i[:, :kk] - 7, ws, wi, f"large k={k} n={n} d={d}")
    if k > n:
        assert (i[:, n:] == -1).all() and np.isneginf(s[:, n:]).all()
    sd, idd = ix.search(torch.from_numpy(q).cuda(), k)
    compare.assert_bit_exact(sd.cpu().numpy()[:, :kk], idd.cpu().numpy()[:, :kk], ws, wi, "large k, device path")
    ix.close()


@pytest.mark.parametrize("nq", [1, 21, 128, 129, 256 + 21, 256 + 128, 256 + 129, 2 * 256 + 100])
def test_tail_pass_on_the_128_query_kernel(amd, nq):
    """scan_topk256 (d = 768): a last pass of at most 128 queries runs on the 128-query kernel, merged as a group of its own
    (option tail128, default on) — same bits as without it and as the oracle, around the routing boundaries.  (balance_tail off: it
    takes remainders of more than one tile away from this routing; test_balanced_remainder_and_idle_waves_match_oracle covers it.)"""
    from bergen_amd import _lib
    _lib.set_option("balance_tail", 0)
    rng = np.random.default_rng(nq)
    n, d, k = 20_000, 768, 50
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    try:
        for tail in (1, 0):
            _lib.set_option("tail128", tail)
            s, i = ix.search(q, k)
            c = ix.counters()
            assert c["n_passes"] == -(-nq // 256) and c["query_tile"] == (128 if tail and nq <= 128 else 256)
            assert (c["tail_query_tile"] == 128 and 0 < c["tail_scan_ms"] <= c["scan_ms"]) == bool(tail and 0 < nq % 256 <= 128)
            want_bytes = 0
            for p in range(c["n_passes"]):
                left = nq - 256 * p
                w = 128 if (tail and left <= 128) else 256
                want_bytes += n * d * 2 + w * d * 2 + w * k * 12
            assert c["algorithmic_bytes"] == want_bytes
            compare.assert_bit_exact(s, i, ws, wi, f"tail128={tail} nq={nq}")
    finally:
        _lib.set_option("tail128", 1)
        _lib.set_option("balance_tail", 1)
        ix.close()


@pytest.mark.parametrize("kern", [3, 0])
@pytest.mark.parametrize("nq", [127, 128, 129, 191, 192, 193, 600])
def test_config5_geometry_matches_oracle(amd, kern, nq):
    """BASELINE configs[4] geometry (e5-large-v2: d = 1024, top-200; reference config/retriever/e5-large-v2.yaml:5-10) on both
    kernels that serve it — the two-waves-per-SIMD kernel with one 16-query block per wave (128 queries per pass) and the
    4-wave kernel — across their pass boundaries."""
    from bergen_amd import _lib
    rng = np.random.default_rng(5000 + nq)
    n, d, k = 30011, 1024, 200
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    try:
        _lib.set_option("scan_kernel", kern)
        ix = amd.FlatIndex(n, d, metric="ip")
        ix.upload(x)
        ix.finalize()
        s, i = ix.search(q, k)
        c = ix.counters()
        ix.close()
        assert c["query_tile"] == 128 and c["k_padded"] == 256 and c["n_passes"] == (nq + 127) // 128
        compare.assert_bit_exact(s, i, ws, wi, f"config5 geometry, kernel {kern}, nq={nq}")
    finally:
        _lib.set_option("scan_kernel", KERNEL_DEFAULT)


def test_tile192_kernel_exact_ties(amd):
    from bergen_amd import _lib
    rng = np.random.default_rng(78)
    x = rng.integers(-2, 3, size=(5000, 768)).astype(np.float16)
    q = rng.integers(-2, 3, size=(70, 768)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, 50)
    try:
        for kern in (3, 2, 0):
            for share in (0, 1):
                _lib.set_option("scan_kernel", kern)
                _lib.set_option("share_threshold", share)
                s, i = _search(amd, x, q, 50)
                compare.assert_bit_exact(s, i, ws, wi, f"ties, kernel {kern}, share {share}")
    finally:
        _lib.set_option("scan_kernel", KERNEL_DEFAULT)


def test_results_written_to_pinned_host_memory(amd):
    """search(host=True): the merge kernel (and the exact fall-back) write the lists straight into pinned host memory."""
    import torch
    rng = np.random.default_rng(77)
    x = rng.standard_normal((20011, 768)).astype(np.float16)
    q = rng.standard_normal((300, 768)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, 50)
    ix = amd.FlatIndex(20011, 768, metric="ip")
    ix.upload(x)
    ix.finalize()
    qd = torch.from_numpy(q).cuda()
    s, i = ix.search(qd, 50, host=True)
    assert not s.is_cuda and s.is_pinned() and i.is_pinned()
    compare.assert_bit_exact(s.numpy(), i.numpy(), ws, wi, "host=True")
    s2, i2 = ix.search(qd[:40], 50, host=True, id_offset=1000)
    compare.assert_bit_exact(s2.numpy(), i2.numpy() - 1000, ws[:40], wi[:40], "host=True, second shape")
    with pytest.raises(ValueError):
        ix.search(q, 50, host=True)
    ix.close()


@pytest.mark.parametrize("n,d,nq,k", [(6000, 768, 8300, 10), (4000, 768, 2100, 200), (3000, 1024, 2200, 200)])
def test_merge_groups_split_at_one_gibibyte_of_lists(amd, n, d, nq, k):
    """The passes of a search are merged in groups whose per-workgroup lists fit 1 GiB (index.hip): 32 passes of 64-entry
    lists, 8 passes of 256-entry lists (16 at d = 1024, 128 queries per pass).  These query sets need two groups; the
    second group reuses the first one's list sets."""
    rng = np.random.default_rng(n + nq)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    ix = amd.FlatIndex(n, d, metric="ip")
    ix.upload(x)
    ix.finalize()
    s, i = ix.search(q, k)
    c = ix.counters()
    ix.close()
    per_pass = 256 * c["query_tile"] * c["k_padded"] * 8
    assert c["n_passes"] * per_pass > (1 << 30), "the case no longer needs two merge groups"
    compare.assert_bit_exact(s, i, ws, wi, f"two merge groups n={n} d={d} nq={nq} k={k}")


def test_sharded_searcher_buffers_on_the_device(amd, monkeypatch):
    """ShardedSearcher on device tensors (the path the 8-GPU bench runs): the local search writes into views of the packed
    send buffer (FlatIndex.search(out=)), the gathered lists are split into dense [G, nq, k] tensors and merged into
    preallocated outputs (merge_topk(out=)).  Two shards on one GPU, the collective replaced by a copy."""
    import torch
    from bergen_amd import sharded
    rng = np.random.default_rng(31)
    n, d, nq, k = 20001, 768, 77, 50      # nq * k odd: the int64 lists start at a padded offset
    x = rng.standard_normal((n, d)).astype(np.float16)
    x[15000] = x[100]                     # a tie across the shard boundary
    q = rng.standard_normal((nq, d)).astype(np.float16)
    ws, wi = c_oracle.canonical_search(q, x, k)
    qd = torch.from_numpy(q).cuda()
    shards = []
    for r in range(2):
        lo, hi = amd.shard_range(n, r, 2)
        ix = amd.FlatIndex(hi - lo, d, metric="ip")
        ix.upload(x[lo:hi])
        ix.finalize()
        shards.append((ix, lo))
    sent = {}

    def fake_gather(flat, packed, group=None):
        per = packed.numel()
        sent[fake_gather.rank] = packed.clone()
        for r, p in sent.items():
            flat[r * per:(r + 1) * per].copy_(p)
    monkeypatch.setattr(sharded.dist, "all_gather_into_tensor", fake_gather)
    s1 = amd.ShardedSearcher(shards[1][0], shards[1][1], rank=1, world_size=2)
    s0 = amd.ShardedSearcher(shards[0][0], shards[0][1], rank=0, world_size=2)
    for _ in range(2):                    # second round: the buffers are reused
        fake_gather.rank = 1
        assert s1.search(qd, k) is None
        fake_gather.rank = 0
        out_s, out_i = s0.search(qd, k)
        assert out_s.is_cuda and out_i.dtype == torch.int64
        compare.assert_bit_exact(out_s.cpu().numpy(), out_i.cpu().numpy(), ws, wi, "two shards, device buffers")
    assert len(s0._buf) == 1
    for ix, _ in shards:
        ix.close()


def test_merge_lists_ranking_paths_against_the_oracle(amd):
    """bh_merge_topk_device ranks by binary search when every list of a query is sorted in the canonical order (what searches
    produce) and by an all-pairs count otherwise (the entry point accepts any lists, like the reference's torch.cat + torch.topk,
    modules/retrieve.py:169-177): both against oracle/flat_ip_oracle.c's merge on lists with exact score ties inside and across
    lists, duplicate ids, short lists (padding at the tail) and — second half — shuffled entries."""
    import torch
    rng = np.random.default_rng(77)
    for n_lists, nq, k in ((8, 300, 200), (8, 257, 50), (3, 64, 7), (20, 33, 200), (1, 5, 50)):
        pool = rng.standard_normal((nq, 64)).astype(np.float32)  # few distinct scores: many ties
        s = pool[np.arange(nq)[None, :, None], rng.integers(0, 64, size=(n_lists, nq, k))]
        i = rng.integers(0, 5000, size=(n_lists, nq, k)).astype(np.int64)  # a small id range: duplicates across and inside lists
        n_valid = rng.integers(0, k + 1, size=(n_lists, nq))
        n_valid[0, 0] = k
        # canonical order inside every list: score descending, id ascending; padding (-inf, -1) behind the valid entries
        order = np.lexsort((i, -s), axis=-1)
        s, i = np.take_along_axis(s, order, -1), np.take_along_axis(i, order, -1)
        pad = np.arange(k)[None, None, :] >= n_valid[:, :, None]
        s[pad], i[pad] = -np.inf, -1
        for shuffled in (False, True):
            if shuffled:  # the same entries in arbitrary order (padding anywhere): the all-pairs path
                perm = np.argsort(rng.random(s.shape), axis=-1)
                s, i = np.take_along_axis(s, perm, -1), np.take_along_axis(i, perm, -1)
            ws, wi = c_oracle.merge_topk(np.ascontiguousarray(s), np.ascontiguousarray(i))
            ms, mi = amd.merge_topk(torch.from_numpy(s).cuda(), torch.from_numpy(i).cuda())
            compare.assert_bit_exact(ms.cpu().numpy(), mi.cpu().numpy(), ws, wi, f"{n_lists} lists x {nq} x {k}, shuffled={shuffled}")
