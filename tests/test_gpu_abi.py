"""GPU: the boundary's own promises (include/bergen_hip.h, BH_VERSION 141) — sized structs and per-handle options."""
import ctypes
import threading

import numpy as np
import pytest

from oracle import c_oracle, compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import bergen_amd
    from bergen_amd import _lib
    _lib.init(0)
    return bergen_amd


def _index(amd, n=9001, d=768, seed=5):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float16)
    ix = amd.FlatIndex(n, d)
    ix.upload(x)
    ix.finalize()
    return ix, x


def test_counters_never_write_beyond_the_callers_struct(amd):
    """A caller compiled against an older, shorter bh_counters passes its own sizeof in struct_size: the library fills that
    many bytes and leaves what follows alone (round 3 grew the struct under an unchanged version: VERDICT r3 weak #8)."""
    from bergen_amd import _lib
    ix, x = _index(amd)
    q = np.random.default_rng(1).standard_normal((5, 768)).astype(np.float16)
    ix.search(q, 10)
    full = _lib.bh_counters()
    _lib.check(_lib.lib().bh_bench_counters(ix._h, ctypes.byref(full)))
    assert full.n_rows == 9001 and full.struct_size == ctypes.sizeof(_lib.bh_counters)
    size = ctypes.sizeof(_lib.bh_counters)
    for short in (16, 48, 88, size, size + 64):
        buf = (ctypes.c_ubyte * (size + 128))(*([0xAB] * (size + 128)))
        ctypes.cast(buf, ctypes.POINTER(ctypes.c_int32))[0] = short
        rc = _lib.lib().bh_bench_counters(ix._h, ctypes.cast(buf, ctypes.POINTER(_lib.bh_counters)))
        assert rc == _lib.BH_OK
        raw = bytes(buf)
        written = min(short, size)
        assert raw[written:] == b"\xab" * (len(raw) - written), f"struct_size={short}: bytes beyond it were written"
        assert raw[:4] == int(short).to_bytes(4, "little")
        assert raw[4:written] == bytes(full)[4:written]
    for bad in (0, 8, -5):
        c = _lib.bh_counters()
        c.struct_size = bad
        assert _lib.lib().bh_bench_counters(ix._h, ctypes.byref(c)) == _lib.BH_EINVAL
        assert b"struct_size" in _lib.lib().bh_last_error()
    ix.close()


def test_per_handle_options_do_not_leak_between_indexes(amd):
    """`bh_index_set_option` changes ONE handle (independent handles per GPU / thread); the process-wide
    `bh_set_option` stays the default of the others.  Results are bit-identical either way."""
    from bergen_amd import _lib
    a, x = _index(amd, seed=7)
    b, _ = _index(amd, seed=7)
    q = np.random.default_rng(2).standard_normal((300, 768)).astype(np.float16)
    want = c_oracle.canonical_search(q, x, 50)
    a.set_option("scan_kernel", 0)          # the 128-query kernel for index a only
    a.set_option("tail128", 0)
    b.set_option("tail128", 0)
    ra, rb = a.search(q, 50), b.search(q, 50)
    assert a.counters()["query_tile"] == 128 and b.counters()["query_tile"] == 256
    compare.assert_bit_exact(ra[0], ra[1], want[0], want[1], "override: 128-query kernel")
    compare.assert_bit_exact(rb[0], rb[1], want[0], want[1], "no override: 256-query kernel")
    _lib.set_option("scan_kernel", 2)       # process-wide default -> b follows, a keeps its own value
    try:
        a.search(q, 50), b.search(q, 50)
        assert a.counters()["query_tile"] == 128 and b.counters()["query_tile"] == 192
        a.set_option("scan_kernel", None)   # drop the override: a follows the process-wide value again
        a.search(q, 50)
        assert a.counters()["query_tile"] == 192
    finally:
        _lib.set_option("scan_kernel", 3)
    with pytest.raises(ValueError):
        a.set_option("scan_kernel", 1)
    with pytest.raises(ValueError):
        a.set_option("sparse_kernel", 1)    # not a dense-search option
    a.close()
    b.close()


def test_two_threads_two_handles_different_options(amd):
    """One Python thread per handle (ctypes drops the GIL inside the call), each handle with its own kernel choice, searching
    at the same time: each gets its own kernel and the oracle's lists."""
    from bergen_amd import _lib
    q = np.random.default_rng(3).standard_normal((260, 768)).astype(np.float16)
    handles, wants = [], []
    for seed, kern in ((11, 0), (12, 3)):
        ix, x = _index(amd, seed=seed)
        ix.set_option("scan_kernel", kern)
        ix.set_option("tail128", 0)
        handles.append((ix, kern))
        wants.append(c_oracle.canonical_search(q, x, 50))
    out = [None, None]
    errors = []

    def work(j):
        try:
            _lib.init(0)
            ix, kern = handles[j]
            for _ in range(5):
                out[j] = ix.search(q, 50)
                assert ix.counters()["query_tile"] == {0: 128, 3: 256}[kern]
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(j,)) for j in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for j in range(2):
        compare.assert_bit_exact(out[j][0], out[j][1], wants[j][0], wants[j][1], f"thread {j}")
        handles[j][0].close()


def test_sparse_per_handle_option(amd):
    from bergen_amd import SparseIndex
    rng = np.random.default_rng(4)
    n, V = 3000, 2000
    indptr = np.zeros(n + 1, np.int64)
    lens = rng.integers(5, 40, n)
    indptr[1:] = np.cumsum(lens)
    terms = np.concatenate([np.sort(rng.choice(V, l, replace=False)) for l in lens]).astype(np.int32)
    vals = rng.random(terms.size).astype(np.float16) + np.float16(0.01)
    q = np.zeros((20, V), np.float16)
    for r in range(20):
        q[r, rng.choice(V, 12, replace=False)] = rng.random(12).astype(np.float16) + np.float16(0.01)
    res = []
    for head in (1, 0):
        ix = SparseIndex(n, V)
        ix.upload((indptr, terms, vals))
        ix.finalize()
        ix.set_option("sparse_head", head)
        res.append(ix.search(q, 10))
        with pytest.raises(ValueError):
            ix.set_option("scan_kernel", 0)
        ix.close()
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
