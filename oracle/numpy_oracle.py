"""
numpy restatement of the exact-search contract.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Independent of liboracle.so, so the two restatements check each other.
Reference lines followed: modules/retrieve.py:146-185 (chunk loop, per-chunk topk, merge),
models/retrievers/dense.py:77-89 (DotProduct / CosineSim).
"""
import numpy as np


def canonical_score_matrix(q16, x16):
    """scores[i, r] = fp32( sum_j q[i,j]*x[r,j] accumulated sequentially in fp64, index order ).

    The j-loop is explicit so the summation ORDER is the contract's (numpy's own reductions use
    pairwise summation).  Vectorised over (i, r); use for small cases only.
    """
    q = np.asarray(q16, np.float16).astype(np.float64)
    x = np.asarray(x16, np.float16).astype(np.float64)
    acc = np.zeros((q.shape[0], x.shape[0]), np.float64)
    for j in range(q.shape[1]):
        acc += q[:, j:j + 1] * x[:, j][None, :]  # product exact in fp64; one rounding per add
    return acc.astype(np.float32)


def topk_canonical(scores, k, ids=None):
    """Per row: the k best columns in (score desc, id asc) order; pads with (-inf, -1)."""
    scores = np.asarray(scores, np.float32)
    nq, n = scores.shape
    if ids is None:
        ids = np.broadcast_to(np.arange(n, dtype=np.int64), (nq, n))
    out_s = np.full((nq, k), -np.inf, np.float32)
    out_i = np.full((nq, k), -1, np.int64)
    for i in range(nq):
        valid = ~np.isnan(scores[i]) & (ids[i] >= 0)
        s, d = scores[i][valid], ids[i][valid]
        order = np.lexsort((d, -s.astype(np.float64)))[:k]  # primary: -score, secondary: id
        out_s[i, :len(order)] = s[order]
        out_i[i, :len(order)] = d[order]
    return out_s, out_i


def canonical_search(q16, x16, k, id_offset=0):
    s = canonical_score_matrix(q16, x16)
    out_s, out_i = topk_canonical(s, k)
    out_i = np.where(out_i >= 0, out_i + id_offset, -1)
    return out_s, out_i


def l2_normalize_rows(x16):
    """Canonical cosine normalisation shared with convert.hip / flat_ip_oracle.c (dense.py:87-88)."""
    x = np.asarray(x16, np.float16).astype(np.float64)
    n2 = np.zeros(x.shape[0], np.float64)
    for j in range(x.shape[1]):
        n2 += x[:, j] * x[:, j]
    out = np.asarray(x16, np.float16).copy()
    nz = n2 > 0
    inv = np.zeros_like(n2)
    inv[nz] = 1.0 / np.sqrt(n2[nz])
    y = (x * inv[:, None]).astype(np.float32).astype(np.float16)
    out[nz] = y[nz]
    return out


def merge_topk(scores, ids):
    """[n_lists, nq, k] partial lists -> [nq, k] (restates retrieve.py:169-177, canonical ties)."""
    n_lists, nq, k = scores.shape
    s = np.transpose(scores, (1, 0, 2)).reshape(nq, n_lists * k)
    d = np.transpose(ids, (1, 0, 2)).reshape(nq, n_lists * k)
    return topk_canonical(s, k, d)


def ref_chunked_search(q32, x32, chunk_rows, dataset_size, k):
    """The reference's loop (retrieve.py:146-185) with numpy fp32 matmul and canonical ties."""
    if int(np.sum(chunk_rows)) != dataset_size:
        raise IOError(f'!!! Index is not complete. Please re-index. Missing {dataset_size - int(np.sum(chunk_rows))} '
                      f'documents in the index. !!!')
    q32 = np.asarray(q32, np.float32)
    parts_s, parts_i = [], []
    off = 0
    for n_c in chunk_rows:
        sc = q32 @ np.asarray(x32[off:off + n_c], np.float32).T           # similarity_fn, dense.py:81
        ps, pi = topk_canonical(sc, min(k, n_c))                          # torch.topk, retrieve.py:157
        parts_s.append(ps)
        parts_i.append(np.where(pi >= 0, pi + off, -1))                   # + num_emb, retrieve.py:159
        off += n_c
    cat_s = np.concatenate(parts_s, axis=1)                               # retrieve.py:169-170
    cat_i = np.concatenate(parts_i, axis=1)
    return topk_canonical(cat_s, k, cat_i)                                # retrieve.py:175-177
