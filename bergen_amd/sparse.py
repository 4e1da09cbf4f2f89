"""
SparseIndex — Python handle of a resident CSR index of SPLADE document vectors (C ABI: bh_sparse_*).

Replaces the host list of sparse COO chunk tensors the reference keeps and re-uploads for every query chunk
(modules/retrieve.py:84-90,153) and its ``Splade.similarity_fn`` = ``torch.sparse.mm`` + ``torch.topk`` search
(models/retrievers/splade.py:55-56, modules/retrieve.py:157,169-177): the corpus is uploaded once, stays in HBM, and
one fused kernel streams it per tile of 64 queries.
"""
import ctypes

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _csr_from_any(rows, vocab):
    """-> (indptr int64 [n+1], terms int32 [nnz], values float16|float32 [nnz]) from a torch sparse COO / dense
    tensor, a numpy dense matrix, a scipy sparse matrix or an (indptr, terms, values) triple."""
    if isinstance(rows, tuple) and len(rows) == 3:
        indptr, terms, vals = rows
        vals = np.asarray(vals)
        if vals.dtype not in (np.float16, np.float32):
            vals = vals.astype(np.float32)
        return np.ascontiguousarray(indptr, np.int64), np.ascontiguousarray(terms, np.int32), np.ascontiguousarray(vals)
    if torch is not None and isinstance(rows, torch.Tensor):
        t = rows.detach().cpu()
        if not t.is_sparse:
            t = t.to_sparse()
        t = t.coalesce()  # sorted by (row, column), duplicates summed
        if t.shape[1] != vocab:
            raise ValueError(f"expected [n, {vocab}] rows, got {tuple(t.shape)}")
        idx = t.indices()
        n = t.shape[0]
        indptr = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(idx[0].numpy(), minlength=n), out=indptr[1:])
        vals = t.values()
        if vals.dtype not in (torch.float16, torch.float32):
            vals = vals.float()
        return indptr, np.ascontiguousarray(idx[1].numpy().astype(np.int32)), np.ascontiguousarray(vals.numpy())
    if hasattr(rows, "tocsr"):  # scipy.sparse
        m = rows.tocsr()
        m.sort_indices()
        vals = m.data if m.data.dtype in (np.float16, np.float32) else m.data.astype(np.float32)
        return m.indptr.astype(np.int64), m.indices.astype(np.int32), np.ascontiguousarray(vals)
    a = np.asarray(rows)
    if a.ndim != 2 or a.shape[1] != vocab:
        raise ValueError(f"expected [n, {vocab}] rows, got {a.shape}")
    nz = a != 0
    indptr = np.zeros(a.shape[0] + 1, np.int64)
    np.cumsum(nz.sum(1), out=indptr[1:])
    r, c = np.nonzero(nz)
    vals = a[r, c]
    if vals.dtype not in (np.float16, np.float32):
        vals = vals.astype(np.float32)
    return indptr, c.astype(np.int32), np.ascontiguousarray(vals)


class SparseIndex:
    """n_rows documents x vocab terms (<= 65535), fp16 weights, resident on one MI355X."""

    MAX_K = 4096  # k <= 120: one fused search (candidate lists of 64 / 128 entries with a margin of 8, csrc/sparse.hip:
    #               pick_kp_sparse); above that the documents are searched range by range (sparse_search_large_k)

    def __init__(self, n_rows, vocab, device=0):
        self._h = None
        _lib.init(device)
        self.device = device
        self.n_rows = int(n_rows)
        self.vocab = int(vocab)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().bh_sparse_create(ctypes.byref(h), self.n_rows, self.vocab))
        self._h = h

    def upload(self, rows, row0=None):
        """Append a block of rows (they must arrive in order).  Returns the number of rows appended."""
        indptr, terms, vals = _csr_from_any(rows, self.vocab)
        n = len(indptr) - 1
        if row0 is None:
            row0 = self.rows_uploaded
        code = _lib.BH_F16 if vals.dtype == np.float16 else _lib.BH_F32
        _lib.init(self.device)
        _lib.check(_lib.lib().bh_sparse_upload_csr(self._h, int(row0), n, ctypes.c_void_p(indptr.ctypes.data),
                                                   ctypes.c_void_p(terms.ctypes.data), ctypes.c_void_p(vals.ctypes.data), code))
        return n

    def finalize(self):
        _lib.init(self.device)
        _lib.check(_lib.lib().bh_sparse_finalize(self._h))
        return self

    @property
    def rows_uploaded(self):
        return int(_lib.lib().bh_sparse_rows_uploaded(self._h))

    @property
    def nnz(self):
        return int(_lib.lib().bh_sparse_nnz(self._h))

    def search(self, queries, k, id_offset=0):
        """Exact top-k for DENSE queries [nq, vocab] (numpy / CPU or device torch tensor, sparse tensors are
        densified).  Returns numpy (scores float32 [nq, k], ids int64 [nq, k]), canonical order."""
        if torch is not None and isinstance(queries, torch.Tensor):
            q = queries.detach()
            if q.is_sparse:
                q = q.to_dense()
            q = q.cpu()
            if q.dtype not in (torch.float16, torch.float32):
                q = q.float()
            q = q.contiguous().numpy()
        else:
            q = np.asarray(queries)
            if q.dtype not in (np.float16, np.float32):
                q = q.astype(np.float32)
            q = np.ascontiguousarray(q)
        if q.ndim != 2 or q.shape[1] != self.vocab:
            raise ValueError(f"expected [nq, {self.vocab}] queries, got {tuple(q.shape)}")
        nq = q.shape[0]
        code = _lib.BH_F16 if q.dtype == np.float16 else _lib.BH_F32
        out_s = np.empty((nq, k), np.float32)
        out_i = np.empty((nq, k), np.int64)
        _lib.init(self.device)
        _lib.check(_lib.lib().bh_sparse_search(self._h, ctypes.c_void_p(q.ctypes.data), code, nq, int(k), int(id_offset),
                                               ctypes.c_void_p(out_s.ctypes.data), ctypes.c_void_p(out_i.ctypes.data)))
        return out_s, out_i

    def counters(self):
        c = _lib.bh_counters()
        _lib.check(_lib.lib().bh_sparse_counters(self._h, ctypes.byref(c)))
        return {name: getattr(c, name) for name, _ in c._fields_}

    def set_option(self, name, value=None):
        """Override "sparse_kernel" / "sparse_head" / "sparse_ablate" for THIS index only; `value=None` drops the override."""
        _lib.check(_lib.lib().bh_sparse_set_option(self._h, name.encode(), _lib.BH_OPTION_INHERIT if value is None else int(value)))

    def close(self):
        if self._h is not None:
            _lib.lib().bh_sparse_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
