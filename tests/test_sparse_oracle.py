"""CPU: the sparse (SPLADE) oracle against the golden fixture produced by the reference's own
Splade.similarity_fn + Retrieve.load_collection_and_retrieve (oracle/make_golden_sparse.py), a scipy cross-check,
and host-side plumbing that needs no GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from bergen_amd import synth
from oracle import c_oracle
from oracle.compare import assert_bit_exact

from conftest import GOLDEN


def load_sparse_golden():
    z = np.load(os.path.join(GOLDEN, "sparse_small.npz"))
    V = int(z["vocab"])
    q = synth.csr_to_dense(z["q_indptr"], z["q_terms"], z["q_weights"], V).astype(np.float16)
    return z, V, q


def test_oracle_reproduces_golden_canonical_outputs():
    z, V, q = load_sparse_golden()
    s, i = c_oracle.sparse_canonical_search(z["d_indptr"], z["d_terms"], z["d_weights"], V, q, int(z["k"]))
    assert_bit_exact(s, i, z["canonical_scores"], z["canonical_ids"], "sparse oracle vs fixture")


def test_canonical_agrees_with_reference_code_up_to_ties():
    """Reference = torch.sparse.mm in fp32 + torch.topk (arbitrary tie order): scores equal, and ids equal wherever the
    score is not tied with a neighbour."""
    z, V, q = load_sparse_golden()
    rs, ri, cs, ci = z["ref_scores"], z["ref_ids"], z["canonical_scores"], z["canonical_ids"]
    assert np.allclose(rs, cs, rtol=2e-6, atol=1e-6)
    for a in range(rs.shape[0]):
        for j in range(rs.shape[1]):
            tied = (j > 0 and cs[a, j] == cs[a, j - 1]) or (j + 1 < cs.shape[1] and cs[a, j] == cs[a, j + 1])
            if not tied:
                assert ri[a, j] == ci[a, j], (a, j)
        assert set(ri[a].tolist()) - set(ci[a].tolist()) == set() or cs[a, -1] == rs[a, -1]


def test_oracle_against_scipy():
    import scipy.sparse as sp
    V = 997
    dp, dt, dw = synth.random_sparse_corpus(400, V, seed=1, mean_nnz=30, lo=0, hi=80)
    qp, qt, qw = synth.random_sparse_corpus(9, V, seed=2, mean_nnz=8, lo=1, hi=20)
    D = sp.csr_matrix((dw.astype(np.float64), dt, dp), shape=(400, V))
    Q = synth.csr_to_dense(qp, qt, qw, V, np.float64)
    want = np.asarray((D @ Q.T).T).astype(np.float32)
    s, i = c_oracle.sparse_canonical_search(dp, dt, dw, V, Q.astype(np.float16), 25)
    for a in range(9):
        order = np.lexsort((np.arange(400), -want[a]))[:25]
        assert np.array_equal(i[a], order)
        assert np.array_equal(s[a], want[a][order])


def test_csr_conversion_from_torch_numpy_scipy():
    from bergen_amd.sparse import _csr_from_any
    import scipy.sparse as sp
    V = 50
    a = np.zeros((4, V), np.float32)
    a[0, [3, 9]] = [1.5, 2.0]
    a[2, [0, 49, 7]] = [0.5, 4.0, 1.0]
    want = (np.array([0, 2, 2, 5]), np.array([3, 9, 0, 7, 49]), np.array([1.5, 2.0, 0.5, 1.0, 4.0], np.float32))
    for src in (a, torch.from_numpy(a), torch.from_numpy(a).to_sparse(), torch.from_numpy(a).half().to_sparse(), sp.csr_matrix(a)):
        p, t, v = _csr_from_any(src, V)
        assert np.array_equal(p[:4], want[0]) and p[4] == 5
        assert np.array_equal(t, want[1]) and np.array_equal(v.astype(np.float32), want[2])
    with pytest.raises(ValueError):
        _csr_from_any(a, V + 1)


def test_sparse_abi_argument_validation_needs_no_device():
    from bergen_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()
    assert lib.bh_sparse_create(ctypes.byref(h), 10, 70000) == _lib.BH_EUNSUPPORTED
    assert lib.bh_sparse_create(None, 10, 100) == _lib.BH_EINVAL
    assert lib.bh_sparse_finalize(None) == _lib.BH_EINVAL
    assert lib.bh_sparse_search(None, None, 0, 1, 1, 0, None, None) == _lib.BH_EINVAL
    assert lib.bh_sparse_rows_uploaded(None) == 0


def test_splade_plugin_surface():
    """Same attributes / call signatures as models/retrievers/splade.py; the selected encoder runs once."""
    import bergen_amd

    class FakeMLM:
        def __init__(self, scale):
            self.scale, self.calls = scale, 0

        def __call__(self, input_ids=None, attention_mask=None, **kw):
            self.calls += 1
            B, T = input_ids.shape
            logits = torch.zeros(B, T, 11)
            logits[:, :, 3] = self.scale * input_ids.float()
            logits[:, :, 5] = -1.0
            return type("O", (), {"logits": logits})()

        def eval(self):
            return self

        def to(self, *a, **k):
            return self

    doc_enc, q_enc = FakeMLM(1.0), FakeMLM(2.0)
    sp = bergen_amd.Splade("naver/splade-fake", max_len=8, model=doc_enc, query_encoder=q_enc, tokenizer=object())
    assert sp.sparse and sp.model_name == "naver/splade-fake"
    kw = {"input_ids": torch.tensor([[1, 4, 2], [3, 0, 0]]), "attention_mask": torch.tensor([[1, 1, 1], [1, 0, 0]])}
    e = sp("doc", kw)["embedding"]
    assert e.shape == (2, 11) and torch.allclose(e[:, 3], torch.log1p(torch.tensor([4.0, 3.0]))) and float(e[:, 5].abs().sum()) == 0
    e = sp("query", kw)["embedding"]
    assert torch.allclose(e[:, 3], torch.log1p(torch.tensor([8.0, 6.0])))
    assert doc_enc.calls == 1 and q_enc.calls == 1  # the reference runs the doc model a second time (splade.py:40)
    inst = bergen_amd.instantiate({"_target_": "models.retrievers.splade.Splade", "model_name": "naver/splade-fake",
                                   "max_len": 8}, model=doc_enc, tokenizer=object())
    assert isinstance(inst, bergen_amd.Splade)
