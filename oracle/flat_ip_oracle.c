/*
 * flat_ip_oracle.c — CPU restatement of BERGEN's exact dense search.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product path (bergen_amd/) may call into this file: it is imported only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker.
 *
 * What it restates (naver/bergen @ 2026-01-30):
 *   - DotProduct.sim            models/retrievers/dense.py:77-81   q @ d^T
 *   - CosineSim.sim             models/retrievers/dense.py:83-89   row-normalise, then q @ d^T
 *   - Retrieve.load_collection_and_retrieve   modules/retrieve.py:146-185
 *        per chunk: scores -> topk(k) -> +offset (:152-164); size check -> IOError (:165-166);
 *        concat partial lists -> fp32 -> topk(k) -> gather (:169-177)
 * The arithmetic underneath (torch.mm, torch.topk) is third-party and unpinned in the reference
 * (requirements.txt has no versions), and torch.topk's tie order is arbitrary
 * (SURVEY §0 D4).  Two functions are therefore provided:
 *
 *   oracle_ref_chunked_search  — the reference's structure verbatim (chunk loop, per-chunk
 *       top-k, global merge) with fp32 dot products and the canonical tie order.  Its ids are
 *       what the reference returns wherever the reference's own result is well defined.
 *   oracle_canonical_search    — the contract of the new backend (include/bergen_hip.h,
 *       bh_search): score = fp32(RNE) of the fp64 sum  sum_{j=0}^{d-1} q[j]*x[j]  accumulated
 *       sequentially in index order over the fp16 values; order = (score desc, row asc).
 *       Products of two fp16 values are exact in fp64, so the result is independent of FMA
 *       contraction and bit-reproducible on any IEEE machine.
 *
 * Parity status: pinned against outputs of the reference's own code run in this container
 * (oracle/make_golden.py imports /root/reference unmodified; fixtures in tests/golden/).
 * The reference's tests hold no golden vectors for this path (tests/zeroshot_test.py:28 FIXME).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- fp16 <-> fp32 (bit-exact, no compiler _Float16 dependence: gcc 11 lacks it on x86) ---- */

float oracle_half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) {
            f = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while (!(man & 0x400u));
            man &= 0x3ffu;
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7f800000u | (man << 13);
    } else {
        f = sign | ((exp + 112) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &f, 4);
    return out;
}

/* round-to-nearest-even, like torch .half() / numpy astype(float16) / v_cvt_f16_f32 */
uint16_t oracle_float_to_half(float x) {
    uint32_t f;
    memcpy(&f, &x, 4);
    uint32_t sign = (f >> 16) & 0x8000u;
    uint32_t exp = (f >> 23) & 0xff;
    uint32_t man = f & 0x7fffffu;
    if (exp == 255) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0));
    int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        int shift = 14 - e; /* 14..24 */
        uint32_t half_man = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_man & 1))) ++half_man;
        return (uint16_t)(sign | half_man);
    }
    uint32_t half = (uint32_t)(e << 10) | (man >> 13);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half; /* may carry into exponent: correct */
    return (uint16_t)(sign | half);
}

/* ---- canonical order ---------------------------------------------------------------------- */

typedef struct {
    float score;
    int64_t id;
} pair_t;

/* returns 1 if a precedes b: score descending, id ascending */
static inline int precedes(float sa, int64_t ia, float sb, int64_t ib) {
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ia < ib;
}

/* keep the best k of a stream in a binary heap whose root is the WORST kept entry */
typedef struct {
    pair_t* h;
    int n, k;
} topk_t;

static void heap_sift_down(pair_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && precedes(h[w].score, h[w].id, h[l].score, h[l].id)) w = l;
        if (r < n && precedes(h[w].score, h[w].id, h[r].score, h[r].id)) w = r;
        if (w == i) return;
        pair_t t = h[i];
        h[i] = h[w];
        h[w] = t;
        i = w;
    }
}
static void heap_sift_up(pair_t* h, int i) {
    while (i > 0) {
        int p = (i - 1) / 2;
        if (precedes(h[p].score, h[p].id, h[i].score, h[i].id)) { /* parent better than child: swap */
            pair_t t = h[i];
            h[i] = h[p];
            h[p] = t;
            i = p;
        } else
            return;
    }
}
static void topk_push(topk_t* t, float s, int64_t id) {
    if (s != s) return; /* NaN scores are not ranked */
    if (t->n < t->k) {
        t->h[t->n].score = s;
        t->h[t->n].id = id;
        heap_sift_up(t->h, t->n);
        ++t->n;
    } else if (precedes(s, id, t->h[0].score, t->h[0].id)) {
        t->h[0].score = s;
        t->h[0].id = id;
        heap_sift_down(t->h, t->n, 0);
    }
}
static int cmp_pair(const void* a, const void* b) {
    const pair_t* x = (const pair_t*)a;
    const pair_t* y = (const pair_t*)b;
    if (precedes(x->score, x->id, y->score, y->id)) return -1;
    if (precedes(y->score, y->id, x->score, x->id)) return 1;
    return 0;
}
/* sorted output, padded with (-inf, -1) */
static void topk_emit(topk_t* t, float* out_s, int64_t* out_i) {
    qsort(t->h, (size_t)t->n, sizeof(pair_t), cmp_pair);
    for (int i = 0; i < t->k; ++i) {
        out_s[i] = i < t->n ? t->h[i].score : -INFINITY;
        out_i[i] = i < t->n ? t->h[i].id : -1;
    }
}

/* ---- canonical score ----------------------------------------------------------------------- */

double oracle_dot_f64_seq(const uint16_t* q, const uint16_t* x, int d) {
    double s = 0.0;
    for (int j = 0; j < d; ++j) s += (double)oracle_half_to_float(q[j]) * (double)oracle_half_to_float(x[j]);
    return s;
}

/* contract of bh_search: see header.  q: [nq, d] fp16 bits, x: [n, d] fp16 bits.
 * out_ids = id_offset + row.  Threads over queries when built with OpenMP. */
void oracle_canonical_search(const uint16_t* q, const uint16_t* x, int64_t nq, int64_t n, int d, int k,
                             int64_t id_offset, float* out_scores, int64_t* out_ids) {
    /* decode the corpus once */
    float* xf = (float*)malloc((size_t)n * d * sizeof(float) + 4);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n * (int64_t)d; ++i) xf[i] = oracle_half_to_float(x[i]);
    /* Tasks = (query, row range): a handful of queries over a million-row block (the streaming full-size check of
     * tests/test_gpu_search.py) still fills every core.  Each task keeps the best k of its range; the best k of the union
     * of the per-range lists is the best k of all rows (every row is offered to exactly one heap, and `precedes` is a
     * total order), so the result does not depend on the split. */
    const int64_t span = 16384;
    const int64_t n_spans = n > 0 ? (n + span - 1) / span : 1;
    pair_t* part = (pair_t*)malloc((size_t)nq * n_spans * k * sizeof(pair_t) + 8);
    int* part_n = (int*)malloc((size_t)nq * n_spans * sizeof(int) + 8);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t qi = 0; qi < nq; ++qi) {
        for (int64_t sp = 0; sp < n_spans; ++sp) {
            double* qd = (double*)malloc((size_t)d * sizeof(double) + 8);
            for (int j = 0; j < d; ++j) qd[j] = (double)oracle_half_to_float(q[qi * d + j]);
            topk_t t;
            t.h = part + (qi * n_spans + sp) * k;
            t.n = 0;
            t.k = k;
            int64_t r1 = (sp + 1) * span < n ? (sp + 1) * span : n;
            for (int64_t r = sp * span; r < r1; ++r) {
                const float* xr = xf + r * d;
                double s = 0.0;
                for (int j = 0; j < d; ++j) s += qd[j] * (double)xr[j]; /* sequential, index order */
                topk_push(&t, (float)s, id_offset + r);
            }
            part_n[qi * n_spans + sp] = t.n;
            free(qd);
        }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
        topk_t t;
        t.h = (pair_t*)malloc((size_t)k * sizeof(pair_t) + 8);
        t.n = 0;
        t.k = k;
        for (int64_t sp = 0; sp < n_spans; ++sp) {
            const pair_t* h = part + (qi * n_spans + sp) * k;
            for (int c = 0; c < part_n[qi * n_spans + sp]; ++c) topk_push(&t, h[c].score, h[c].id);
        }
        topk_emit(&t, out_scores + qi * k, out_ids + qi * k);
        free(t.h);
    }
    free(part_n);
    free(part);
    free(xf);
}

/* canonical scores of given (query, row) pairs: rows[qi*m + c] (row < 0 -> -inf) */
void oracle_canonical_scores(const uint16_t* q, const uint16_t* x, int64_t nq, int d, const int64_t* rows, int m,
                             float* out_scores) {
    for (int64_t qi = 0; qi < nq; ++qi)
        for (int c = 0; c < m; ++c) {
            int64_t r = rows[qi * m + c];
            out_scores[qi * m + c] = r < 0 ? -INFINITY : (float)oracle_dot_f64_seq(q + qi * d, x + r * d, d);
        }
}

/* ---- the reference's structure: modules/retrieve.py:146-185 -------------------------------- */

/* q: [nq, d] fp32, x: [n, d] fp32 (the reference-code oracle runs the fp16-valued data in fp32 on
 * CPU).  chunk_rows[0..n_chunks) are the chunk sizes (embedding_chunk_*.pt files,
 * retrieve.py:84-90); dataset_size is len(dataset['doc']).
 * Returns 0, or -4 when sum(chunk_rows) != dataset_size (the reference raises IOError,
 * retrieve.py:165-166), in which case *missing receives dataset_size - sum. */
int oracle_ref_chunked_search(const float* q, const float* x, int64_t nq, int d, const int64_t* chunk_rows,
                              int n_chunks, int64_t dataset_size, int k, float* out_scores, int64_t* out_ids,
                              int64_t* missing) {
    int64_t total = 0;
    for (int c = 0; c < n_chunks; ++c) total += chunk_rows[c];
    if (total != dataset_size) { /* retrieve.py:165-166 */
        if (missing) *missing = dataset_size - total;
        return -4;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float* qv = q + qi * d;
        /* partial lists: k per chunk (retrieve.py:157-159), concatenated (retrieve.py:169-170) */
        pair_t* cat = (pair_t*)malloc((size_t)n_chunks * k * sizeof(pair_t) + 16);
        int ncat = 0;
        int64_t num_emb = 0; /* running offset, retrieve.py:151,164 */
        topk_t t;
        t.h = (pair_t*)malloc((size_t)k * sizeof(pair_t));
        for (int c = 0; c < n_chunks; ++c) {
            t.n = 0;
            t.k = k < chunk_rows[c] ? k : (int)chunk_rows[c];
            for (int64_t r = 0; r < chunk_rows[c]; ++r) {
                const float* xr = x + (num_emb + r) * d;
                float s = 0.f; /* similarity_fn = torch.mm in fp32: dense.py:81 */
                for (int j = 0; j < d; ++j) s += qv[j] * xr[j];
                topk_push(&t, s, num_emb + r); /* indices + num_emb: retrieve.py:159 */
            }
            for (int i = 0; i < t.n; ++i) cat[ncat++] = t.h[i];
            num_emb += chunk_rows[c];
        }
        /* final top-k over the concatenation, then gather: retrieve.py:175-177 */
        t.n = 0;
        t.k = k;
        for (int i = 0; i < ncat; ++i) topk_push(&t, cat[i].score, cat[i].id);
        topk_emit(&t, out_scores + qi * k, out_ids + qi * k);
        free(t.h);
        free(cat);
    }
    return 0;
}

/* ---- cosine: canonical row normalisation (restates CosineSim.sim's x / ||x||, dense.py:87-88;
 * definition shared with bergen_amd/csrc/convert.hip) ---------------------------------------- */
void oracle_l2_normalize_rows(uint16_t* x, int64_t n, int d) {
    /* rows are independent (threads over rows change no arithmetic): the full-size cosine check normalises 21 M x 1024 */
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        uint16_t* xr = x + r * d;
        double n2 = 0.0;
        for (int j = 0; j < d; ++j) {
            double v = (double)oracle_half_to_float(xr[j]);
            n2 += v * v;
        }
        if (n2 > 0.0) {
            double inv = 1.0 / sqrt(n2);
            for (int j = 0; j < d; ++j) {
                float y = (float)((double)oracle_half_to_float(xr[j]) * inv); /* fp64 -> fp32 RNE */
                xr[j] = oracle_float_to_half(y);                               /* fp32 -> fp16 RNE */
            }
        }
    }
}

/* ---- merge of partial lists (shards / ranks): retrieve.py:169-177 with the canonical order --- */
void oracle_merge_topk(const float* scores, const int64_t* ids, int n_lists, int64_t nq, int k, float* out_scores,
                       int64_t* out_ids) {
    for (int64_t qi = 0; qi < nq; ++qi) {
        topk_t t;
        t.h = (pair_t*)malloc((size_t)k * sizeof(pair_t));
        t.n = 0;
        t.k = k;
        for (int l = 0; l < n_lists; ++l)
            for (int j = 0; j < k; ++j) {
                int64_t id = ids[((int64_t)l * nq + qi) * k + j];
                if (id >= 0) topk_push(&t, scores[((int64_t)l * nq + qi) * k + j], id);
            }
        topk_emit(&t, out_scores + qi * k, out_ids + qi * k);
        free(t.h);
    }
}

/* bulk conversions for the Python side */
void oracle_floats_to_halfs(const float* src, uint16_t* dst, int64_t n) {
    for (int64_t i = 0; i < n; ++i) dst[i] = oracle_float_to_half(src[i]);
}
void oracle_halfs_to_floats(const uint16_t* src, float* dst, int64_t n) {
    for (int64_t i = 0; i < n; ++i) dst[i] = oracle_half_to_float(src[i]);
}
