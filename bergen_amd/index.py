"""
FlatIndex — Python handle of a resident, HBM-only flat index (C ABI: bh_index_*).

Replaces the host-side list of chunk tensors the reference keeps and re-uploads for every query
chunk (modules/retrieve.py:84-90,153): the corpus is uploaded ONCE and stays in HBM.
Accepts numpy arrays and torch tensors (CPU or ROCm device); device tensors are consumed in
place through bh_index_upload_device / bh_search_device (no host round trip).
"""
import ctypes

import numpy as np

from . import _lib

try:  # torch is plumbing here (device memory / streams), never the compute path
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _dtype_code(x):
    if _is_torch(x):
        if x.dtype == torch.float16:
            return _lib.BH_F16
        if x.dtype == torch.float32:
            return _lib.BH_F32
    else:
        if x.dtype == np.float16:
            return _lib.BH_F16
        if x.dtype == np.float32:
            return _lib.BH_F32
    raise TypeError(f"embeddings must be float16 or float32, got {x.dtype}")


def _prepare(x):
    """-> (pointer, dtype_code, on_device, keepalive) for a 2-D row-major matrix."""
    if _is_torch(x):
        if x.is_sparse:
            x = x.to_dense()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        x = x.contiguous()
        if x.is_cuda:
            torch.cuda.current_stream(x.device).synchronize()  # library works on its own stream
            return ctypes.c_void_p(x.data_ptr()), _dtype_code(x), True, x
        return ctypes.c_void_p(x.data_ptr()), _dtype_code(x), False, x
    x = np.asarray(x)
    if x.dtype not in (np.float16, np.float32):
        x = x.astype(np.float32)
    x = np.ascontiguousarray(x)
    return ctypes.c_void_p(x.ctypes.data), _dtype_code(x), False, x


class FlatIndex:
    """n_rows x dim fp16 index resident on one MI355X.  metric: 'ip' | 'cos'."""

    MAX_K = 4096  # k <= 248: one fused search (candidate lists of 64 / 128 / 256 entries with a margin of 8, csrc/index.hip:
    #               pick_kp); larger k: the corpus is searched range by range and the lists merged (csrc/index.hip: search_large_k)

    def __init__(self, n_rows, dim, metric="ip", device=0):
        self._h = None
        self._host_out = None
        _lib.init(device)
        self.device = device
        self.n_rows = int(n_rows)
        self.dim = int(dim)
        self.metric = metric
        m = {"ip": _lib.BH_METRIC_IP, "dot": _lib.BH_METRIC_IP, "cos": _lib.BH_METRIC_COS,
             "cosine": _lib.BH_METRIC_COS}[metric]
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().bh_index_create(ctypes.byref(h), self.n_rows, self.dim, _lib.BH_F16, m))
        self._h = h
        self._next_row = 0

    # -- building -------------------------------------------------------------------------
    def upload(self, rows, row0=None):
        """Copy a [n, dim] block to rows [row0, row0+n) (default: append)."""
        if row0 is None:
            row0 = self._next_row
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] rows, got {tuple(rows.shape)}")
        ptr, code, on_dev, keep = _prepare(rows)
        n = int(rows.shape[0])
        _lib.init(self.device)
        fn = _lib.lib().bh_index_upload_device if on_dev else _lib.lib().bh_index_upload
        _lib.check(fn(self._h, int(row0), ptr, n, code))
        del keep
        self._next_row = max(self._next_row, int(row0) + n)
        return n

    def finalize(self):
        _lib.init(self.device)
        _lib.check(_lib.lib().bh_index_finalize(self._h))
        return self

    @property
    def rows_uploaded(self):
        return int(_lib.lib().bh_index_rows_uploaded(self._h))

    # -- searching ------------------------------------------------------------------------
    def search(self, queries, k, id_offset=0, out=None, host=False):
        """Exact top-k.  numpy / CPU tensor in -> numpy out; device tensor in -> device tensors out.

        Returns (scores float32 [nq, k], ids int64 [nq, k]) in the canonical order
        (score desc, row asc); ids are id_offset + row, -1 where the index has < k rows.
        out: optional (scores, ids) device tensors of those shapes to write into (device queries only).
        host=True (device queries): the kernels write the result lists straight into pinned host memory and CPU tensors
        come back — no device-side result buffers, no separate D2H (which costs ~1.2 ms behind a 90 ms search; the lists
        are 1.7 MB at 2 837 x 50).  The tensors are reused by the next host=True search of the same shape.
        """
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(queries.shape)}")
        nq = int(queries.shape[0])
        ptr, code, on_dev, keep = _prepare(queries)
        _lib.init(self.device)
        if out is not None and not on_dev:
            raise ValueError("out= needs device queries")
        if host and (out is not None or not on_dev):
            raise ValueError("host=True needs device queries and no out=")
        if on_dev and host:
            key = (nq, int(k))
            if self._host_out is None or self._host_out[0] != key:
                self._host_out = (key, torch.empty((nq, k), dtype=torch.float32, pin_memory=True),
                                  torch.empty((nq, k), dtype=torch.int64, pin_memory=True))
            _, out_s, out_i = self._host_out
            torch.cuda.current_stream(keep.device).synchronize()
            # (pinned host memory is device-accessible at its host address; bh_search_device returns after the stream drained)
            _lib.check(_lib.lib().bh_search_device(self._h, ptr, code, nq, int(k), int(id_offset),
                                                   ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr())))
            return out_s, out_i
        if on_dev:
            if out is not None:
                out_s, out_i = out
                if (tuple(out_s.shape) != (nq, k) or tuple(out_i.shape) != (nq, k) or out_s.dtype != torch.float32
                        or out_i.dtype != torch.int64 or not out_s.is_contiguous() or not out_i.is_contiguous()
                        or out_s.device != keep.device or out_i.device != keep.device):
                    raise ValueError(f"out= must be contiguous float32 / int64 [{nq}, {k}] tensors on {keep.device}")
            else:
                out_s = torch.empty((nq, k), dtype=torch.float32, device=keep.device)
                out_i = torch.empty((nq, k), dtype=torch.int64, device=keep.device)
            torch.cuda.current_stream(keep.device).synchronize()
            _lib.check(_lib.lib().bh_search_device(self._h, ptr, code, nq, int(k), int(id_offset),
                                                   ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr())))
            return out_s, out_i
        out_s = np.empty((nq, k), np.float32)
        out_i = np.empty((nq, k), np.int64)
        _lib.check(_lib.lib().bh_search(self._h, ptr, code, nq, int(k), int(id_offset),
                                        ctypes.c_void_p(out_s.ctypes.data), ctypes.c_void_p(out_i.ctypes.data)))
        del keep
        return out_s, out_i

    def counters(self):
        c = _lib.bh_counters()
        _lib.check(_lib.lib().bh_bench_counters(self._h, ctypes.byref(c)))
        return {name: getattr(c, name) for name, _ in c._fields_}

    def set_option(self, name, value=None):
        """Override a dense-search option for THIS index only (`_lib.set_option` sets the process-wide default every
        index without an override follows); `value=None` drops the override."""
        _lib.check(_lib.lib().bh_index_set_option(self._h, name.encode(), _lib.BH_OPTION_INHERIT if value is None else int(value)))

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if self._h is not None:
            _lib.lib().bh_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk(scores, ids, out=None):
    """Merge [n_lists, nq, k] partial top-k lists (one per shard / rank) on the device.

    numpy in -> numpy out; device tensors in -> device tensors out.  Canonical order.
    Replaces the host merge of reference modules/retrieve.py:169-177.
    out: optional (scores [nq, k] float32, ids [nq, k] int64) device tensors to write into.
    """
    if _is_torch(scores) and scores.is_cuda:
        scores = scores.contiguous().float()
        ids = ids.contiguous().long()
        n_lists, nq, k = scores.shape
        if out is not None:
            out_s, out_i = out
        else:
            out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
            out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
        torch.cuda.current_stream(scores.device).synchronize()
        _lib.init(scores.device.index if scores.device.index is not None else torch.cuda.current_device())  # (the lists' GPU: one process may hold several)
        _lib.check(_lib.lib().bh_merge_topk_device(ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                                   n_lists, nq, k, ctypes.c_void_p(out_s.data_ptr()),
                                                   ctypes.c_void_p(out_i.data_ptr())))
        return out_s, out_i
    if _is_torch(scores):
        scores, ids = scores.numpy(), ids.numpy()
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    n_lists, nq, k = scores.shape
    out_s = np.empty((nq, k), np.float32)
    out_i = np.empty((nq, k), np.int64)
    _lib.check(_lib.lib().bh_merge_topk(ctypes.c_void_p(scores.ctypes.data), ctypes.c_void_p(ids.ctypes.data),
                                        n_lists, nq, k, ctypes.c_void_p(out_s.ctypes.data),
                                        ctypes.c_void_p(out_i.ctypes.data)))
    return out_s, out_i
