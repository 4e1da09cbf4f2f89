// sparse.hip — C ABI (include/bergen_hip.h, bh_sparse_*): resident CSR index of SPLADE document vectors and exact
// sparse search.
//
// Reference behaviour being replaced (naver/bergen):
//   modules/retrieve.py:135-141    chunks saved as sparse COO tensors when 'splade' is in the model name
//   modules/retrieve.py:84-90      every chunk loaded to HOST RAM (torch.load), re-uploaded per query chunk (:153)
//                                                                                    -> one CSR index resident in HBM
//   models/retrievers/splade.py:55-56   torch.sparse.mm(q.to_sparse(), d_chunk.t()).to_dense()   \
//   modules/retrieve.py:157,169-177     torch.topk per chunk, host cat/topk/gather               /  -> csr_topk.hip
//   modules/retrieve.py:165-166    IOError when rows are missing                     -> BH_EINCOMPLETE
// There is no CPU fallback: every entry point that computes needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <vector>

#include "bh_host.h"
#include "bh_kernels.h"

struct bh_sparse_index {
    int device = 0;
    int n_cu = 256;
    int64_t n_rows = 0;
    int vocab = 0;
    int64_t rows_have = 0;
    int64_t nnz = 0;
    bool finalized = false;
    hipStream_t stream = nullptr;
    hipStream_t merge_stream = nullptr;  // merge / re-score of tile p runs beside the scan of tile p + 1
    BhDevBuf<unsigned> entries;
    BhDevBuf<long long> row_ptr;
    BhDevBuf<bh_u64> cand, partial;
    BhDevBuf<unsigned> gthr, bitmap;
    bool nonneg_docs = true;  // no stored document weight is negative (SPLADE vectors are log(1 + relu(.)) >= 0)
    BhDevBuf<unsigned short> prefix;
    BhDevBuf<_Float16> W, qdense, WhT, WgT;
    // corpus-side head block (csr_head.hip / csr_mfma.hip): the 64 terms of largest document frequency as a dense tile per
    // 32-document group in front of the group's remaining entries; built at finalize, read by the MFMA scan only
    std::vector<unsigned> term_df;        // [vocab + 1] document frequency by stored id, counted while the rows are uploaded
    std::vector<int> head_terms;          // stored ids of the corpus-head terms, ascending
    std::vector<unsigned char> head_slot; // [vocab + 1] slot 0..63 of a corpus-head term, 0xff otherwise
    BhDevBuf<unsigned> stream2;           // tiles + tail entries
    BhDevBuf<long long> row_ptr2;         // [n_rows + 1] row pointers of the tail entries
    BhDevBuf<unsigned char> head_slot_dev;
    bool has_head = false;
    BhDevBuf<unsigned> sinfo, pairs;
    BhDevBuf<unsigned char> outbuf;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_b[4] = {nullptr, nullptr, nullptr, nullptr};  // timing events of odd query tiles (tiles are pipelined in pairs)
    bh_counters counters{};
    int opt_kernel = -1, opt_head = -1, opt_ablate = -1;  // per-handle overrides (bh_sparse_set_option); -1 = the process-wide value
};

namespace {

int g_sparse_ablate = 0;
int g_sparse_kernel = 1;  // 1 = csr_mfma.hip (head terms through MFMA), 0 = csr_topk.hip (broadcast per hit)
int g_sparse_head = 1;    // 1 = the MFMA scan reads the corpus-head tiles + tail stream built at finalize, 0 = the plain CSR

constexpr int kLdsBytes = 160 * 1024;
constexpr int kTileQ = 64;
constexpr int kSparseListK = 120;  // largest k of ONE fused sparse search (candidate lists of 128 with a margin of 8)

int grow_entries(bh_sparse_index* ix, size_t need) {
    if (need <= ix->entries.cap) return BH_OK;
    size_t cap = std::max<size_t>(need, ix->entries.cap * 2);
    cap = std::max<size_t>(cap, 1 << 20);
    unsigned* np = nullptr;
    BH_HIP_TRY(hipMalloc((void**)&np, cap * sizeof(unsigned)));
    if (ix->nnz > 0) {
        hipError_t e = hipMemcpyAsync(np, ix->entries.p, (size_t)ix->nnz * sizeof(unsigned), hipMemcpyDeviceToDevice,
                                      ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e != hipSuccess) {
            (void)hipFree(np);
            return bh_fail(BH_EHIP, "growing the entry buffer: %s", hipGetErrorString(e));
        }
    }
    if (ix->entries.p) (void)hipFree(ix->entries.p);
    ix->entries.p = np;
    ix->entries.cap = cap;
    return BH_OK;
}

inline unsigned short f32_to_f16_bits(float f) {
    const _Float16 h = (_Float16)f;  // round-to-nearest-even, like torch .half()
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}
inline float f16_bits_to_f32(unsigned short b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

int pick_kp_sparse(int k) {
    if (k <= 56) return 64;
    if (k <= 120) return 128;
    return -1;
}

}  // namespace

void bh_sparse_set_kernel(int which) { g_sparse_kernel = which ? 1 : 0; }
void bh_sparse_set_ablate(int bits) { g_sparse_ablate = bits; }
void bh_sparse_set_head(int on) { g_sparse_head = on ? 1 : 0; }

extern "C" {

int bh_sparse_create(bh_sparse_index** out, int64_t n_rows, int32_t vocab) {
    if (!out) return bh_fail(BH_EINVAL, "null out");
    *out = nullptr;
    if (n_rows < 0 || n_rows >= 0xffffffffll) return bh_fail(BH_EINVAL, "n_rows %lld out of range", (long long)n_rows);
    // (entries store term + 1 in 16 bits: id 0 is "no term" — what a load past a document group's end returns)
    if (vocab <= 0 || vocab > 65535) return bh_fail(BH_EUNSUPPORTED, "vocab %d unsupported (1..65535: stored term ids are 16-bit, 0 is reserved)", vocab);
    int dev = 0;
    BH_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    BH_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return bh_fail(BH_EUNSUPPORTED, "device %d is %s; gfx950 required", dev, prop.gcnArchName);
    bh_sparse_index* ix = new bh_sparse_index();
    ix->device = dev;
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ix->n_rows = n_rows;
    ix->vocab = vocab;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&ix->ev[i]);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&ix->ev_b[i]);
    int rc = e == hipSuccess ? ix->row_ptr.ensure((size_t)n_rows + 1) : bh_fail(BH_EHIP, "stream/event: %s", hipGetErrorString(e));
    if (rc == BH_OK) {
        const long long zero = 0;
        e = hipMemcpy(ix->row_ptr.p, &zero, sizeof zero, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = bh_fail(BH_EHIP, "row_ptr init: %s", hipGetErrorString(e));
    }
    if (rc != BH_OK) {
        bh_sparse_destroy(ix);
        return rc;
    }
    *out = ix;
    return BH_OK;
}

void bh_sparse_destroy(bh_sparse_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    ix->entries.release();
    ix->row_ptr.release();
    ix->cand.release();
    ix->partial.release();
    ix->gthr.release();
    ix->bitmap.release();
    ix->prefix.release();
    ix->W.release();
    ix->qdense.release();
    ix->WhT.release();
    ix->WgT.release();
    ix->stream2.release();
    ix->row_ptr2.release();
    ix->head_slot_dev.release();
    ix->sinfo.release();
    ix->pairs.release();
    ix->outbuf.release();
    for (auto& e : ix->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ix->ev_b)
        if (e) (void)hipEventDestroy(e);
    if (ix->merge_stream) (void)hipStreamDestroy(ix->merge_stream);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

int64_t bh_sparse_rows_uploaded(const bh_sparse_index* ix) { return ix ? ix->rows_have : 0; }
int64_t bh_sparse_nnz(const bh_sparse_index* ix) { return ix ? ix->nnz : 0; }

int bh_sparse_upload_csr(bh_sparse_index* ix, int64_t row0, int64_t n, const int64_t* indptr, const int32_t* terms,
                         const void* values, int32_t val_dtype) {
    if (!ix) return bh_fail(BH_EINVAL, "null index");
    if (ix->finalized) return bh_fail(BH_EINVAL, "index already finalized");
    if (n < 0 || row0 != ix->rows_have)
        return bh_fail(BH_EINVAL, "rows must be appended in order: expected row0=%lld, got %lld", (long long)ix->rows_have,
                       (long long)row0);
    if (row0 + n > ix->n_rows)
        return bh_fail(BH_EINVAL, "rows [%lld, %lld) outside the index (n_rows=%lld)", (long long)row0, (long long)(row0 + n),
                       (long long)ix->n_rows);
    if (val_dtype != BH_F16 && val_dtype != BH_F32) return bh_fail(BH_EINVAL, "bad val_dtype %d", val_dtype);
    if (n == 0) return BH_OK;
    if (!indptr || indptr[0] != 0) return bh_fail(BH_EINVAL, "indptr must start at 0");
    if (ix->term_df.empty()) ix->term_df.assign((size_t)ix->vocab + 1, 0u);
    const int64_t nnz_in = indptr[n];
    if (nnz_in < 0 || (nnz_in > 0 && (!terms || !values))) return bh_fail(BH_EINVAL, "null terms / values");
    BH_HIP_TRY(hipSetDevice(ix->device));
    // pack (term, fp16 weight), drop explicit zeros (to_sparse() stores none), sort each row by term id
    std::vector<unsigned> packed;
    packed.reserve((size_t)nnz_in);
    std::vector<long long> rp((size_t)n);
    const unsigned short* v16 = static_cast<const unsigned short*>(values);
    const float* v32 = static_cast<const float*>(values);
    for (int64_t r = 0; r < n; ++r) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        if (e < b || e > nnz_in) return bh_fail(BH_EINVAL, "indptr not monotone at row %lld", (long long)(row0 + r));
        const size_t start = packed.size();
        for (int64_t i = b; i < e; ++i) {
            const int32_t t = terms[i];
            if (t < 0 || t >= ix->vocab) return bh_fail(BH_EINVAL, "term id %d out of range at row %lld", t, (long long)(row0 + r));
            const unsigned short hb = val_dtype == BH_F16 ? v16[i] : f32_to_f16_bits(v32[i]);
            if ((hb & 0x7fffu) == 0) continue;  // +-0
            if (hb & 0x8000u) ix->nonneg_docs = false;
            packed.push_back((unsigned)(t + 1) | ((unsigned)hb << 16));  // stored id = term + 1
            ++ix->term_df[(size_t)t + 1];
        }
        std::sort(packed.begin() + start, packed.end(), [](unsigned x, unsigned y) { return (x & 0xffffu) < (y & 0xffffu); });
        for (size_t i = start + 1; i < packed.size(); ++i)
            if ((packed[i] & 0xffffu) == (packed[i - 1] & 0xffffu))
                return bh_fail(BH_EINVAL, "duplicate term %u in row %lld (coalesce the tensor first)", (packed[i] & 0xffffu) - 1u,
                               (long long)(row0 + r));
        rp[(size_t)r] = ix->nnz + (long long)packed.size();
    }
    int rc = grow_entries(ix, (size_t)ix->nnz + packed.size());
    if (rc) return rc;
    if (!packed.empty())
        BH_HIP_TRY(hipMemcpyAsync(ix->entries.p + ix->nnz, packed.data(), packed.size() * sizeof(unsigned),
                                  hipMemcpyHostToDevice, ix->stream));
    BH_HIP_TRY(hipMemcpyAsync(ix->row_ptr.p + row0 + 1, rp.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice,
                              ix->stream));
    BH_HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->nnz += (int64_t)packed.size();
    ix->rows_have += n;
    return BH_OK;
}

int bh_sparse_finalize(bh_sparse_index* ix) {
    if (!ix) return bh_fail(BH_EINVAL, "null index");
    if (ix->rows_have != ix->n_rows)
        return bh_fail(BH_EINCOMPLETE, "!!! Index is not complete. Please re-index. Missing %lld documents in the index. !!!",
                       (long long)(ix->n_rows - ix->rows_have));
    // ---- corpus-side head block: the 64 terms of largest document frequency become a dense tile per 32-document group
    // (csr_head.hip); the original CSR stays for the canonical re-score and the first-generation scan
    if (!ix->finalized && ix->nnz > 0 && ix->n_rows > 0) {
        BH_HIP_TRY(hipSetDevice(ix->device));
        const int V = ix->vocab + 1;
        std::vector<int> order;
        for (int t = 1; t < V; ++t)
            if (ix->term_df[(size_t)t] > 0) order.push_back(t);
        std::sort(order.begin(), order.end(), [&](int x, int y) {
            return ix->term_df[(size_t)x] != ix->term_df[(size_t)y] ? ix->term_df[(size_t)x] > ix->term_df[(size_t)y] : x < y;
        });
        order.resize(std::min<size_t>(order.size(), BH_CSR_HEAD_TERMS));
        std::sort(order.begin(), order.end());
        ix->head_terms = order;
        ix->head_slot.assign((size_t)V, (unsigned char)0xff);
        for (size_t i = 0; i < order.size(); ++i) ix->head_slot[(size_t)order[i]] = (unsigned char)i;
        int rc;
        if ((rc = ix->head_slot_dev.ensure((size_t)V))) return rc;
        if ((rc = ix->row_ptr2.ensure((size_t)ix->n_rows + 1))) return rc;
        BhDevBuf<unsigned> cnt;
        if ((rc = cnt.ensure((size_t)ix->n_rows))) return rc;
        hipStream_t st = ix->stream;
        BH_HIP_TRY(hipMemcpyAsync(ix->head_slot_dev.p, ix->head_slot.data(), (size_t)V, hipMemcpyHostToDevice, st));
        BH_HIP_TRY(bh_launch_csr_tail_count(ix->entries.p, ix->row_ptr.p, ix->n_rows, ix->head_slot_dev.p, cnt.p, st));
        std::vector<unsigned> cnt_h((size_t)ix->n_rows);
        BH_HIP_TRY(hipMemcpyAsync(cnt_h.data(), cnt.p, (size_t)ix->n_rows * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        BH_HIP_TRY(hipStreamSynchronize(st));
        cnt.release();
        std::vector<long long> rp2((size_t)ix->n_rows + 1);
        rp2[0] = 0;
        for (int64_t r = 0; r < ix->n_rows; ++r) rp2[(size_t)r + 1] = rp2[(size_t)r] + (long long)cnt_h[(size_t)r];
        const long long n_groups = (ix->n_rows + 31) / 32;
        const size_t stream_dwords = (size_t)rp2[(size_t)ix->n_rows] + (size_t)BH_CSR_HEAD_DWORDS * (size_t)n_groups;
        if ((rc = ix->stream2.ensure(stream_dwords + 64))) return rc;
        BH_HIP_TRY(hipMemcpyAsync(ix->row_ptr2.p, rp2.data(), rp2.size() * sizeof(long long), hipMemcpyHostToDevice, st));
        BH_HIP_TRY(bh_launch_csr_split(ix->entries.p, ix->row_ptr.p, ix->row_ptr2.p, ix->n_rows, ix->head_slot_dev.p, ix->stream2.p, st));
        BH_HIP_TRY(hipStreamSynchronize(st));
        ix->has_head = true;
    }
    ix->finalized = true;
    return BH_OK;
}

// One fused search over a VIEW of the corpus: documents [view_lo, view_lo + view_rows) (view_lo a multiple of 32, the scan's
// document group), k <= kSparseListK.  Row ids inside the kernels are relative to the view (the row-pointer arrays are
// passed from view_lo on, entry offsets stay absolute; the corpus-head tile of group g sits HD * g dwords into the tail
// stream, so that stream is passed from the view's first group on): the caller folds view_lo into id_offset.
// floor_flags (optional, [nq]): 1 for the queries whose tile took the non-negative path — their lists hold EVERY document of
// the view with a positive score first and continue with the view's lowest zero-score rows.
static int sparse_search_view(bh_sparse_index* ix, int64_t view_lo, int64_t view_rows, const void* q_host, int32_t q_dtype, int32_t nq,
                              int32_t k, int64_t id_offset, float* out_scores, int64_t* out_ids, char* floor_flags = nullptr) {
    if (!ix) return bh_fail(BH_EINVAL, "null index");
    if (!ix->finalized) {
        if (ix->rows_have != ix->n_rows)
            return bh_fail(BH_EINCOMPLETE,
                           "!!! Index is not complete. Please re-index. Missing %lld documents in the index. !!!",
                           (long long)(ix->n_rows - ix->rows_have));
        return bh_fail(BH_EINVAL, "index not finalized");
    }
    if (nq < 0 || k <= 0) return bh_fail(BH_EINVAL, "nq=%d k=%d", nq, k);
    if (q_dtype != BH_F16 && q_dtype != BH_F32) return bh_fail(BH_EINVAL, "bad q_dtype %d", q_dtype);
    const int kp = pick_kp_sparse(k);
    if (kp < 0) return bh_fail(BH_EUNSUPPORTED, "k=%d unsupported for one fused sparse search (max %d)", k, kSparseListK);
    if (nq == 0) return BH_OK;
    if (view_lo < 0 || (view_lo & 31) != 0 || view_rows < 0 || view_lo + view_rows > ix->n_rows)
        return bh_fail(BH_EINVAL, "internal: bad document view [%lld, +%lld)", (long long)view_lo, (long long)view_rows);
    const long long* const rp_v = ix->row_ptr.p + view_lo;
    const long long* const rp2_v = ix->row_ptr2.p ? ix->row_ptr2.p + view_lo : nullptr;
    const unsigned* const stream2_v = ix->stream2.p ? ix->stream2.p + (size_t)BH_CSR_HEAD_DWORDS * (size_t)(view_lo / 32) : nullptr;
    const int64_t n_rows_v = view_rows;
    if (!q_host || !out_scores || !out_ids) return bh_fail(BH_EINVAL, "null buffer");
    BH_HIP_TRY(hipSetDevice(ix->device));
    // Internally the vocabulary is the caller's shifted by one: stored id = term + 1, id 0 never occurs in a document or a
    // query.  A buffer load past the end of a document group returns 0, i.e. id 0, whose term-set bit is never set: the
    // scan needs no position test per entry (csr_mfma.hip).
    const int Vu = ix->vocab, V = Vu + 1, n_words = (V + 31) / 32;
    const int off_prefix = n_words * 4;
    const int off_w = (off_prefix + n_words * 2 + 15) / 16 * 16;
    const int max_slots = (kLdsBytes - 256 - off_w) / 128 - 1;
    if (max_slots < 8) return bh_fail(BH_EUNSUPPORTED, "vocab %d leaves no LDS for the tile weights", V);
    const int grid = ix->n_cu;
    int rc;
    const size_t partial_elems = (size_t)grid * 64 * kp;
    if ((rc = ix->partial.ensure(2 * partial_elems))) return rc;  // two sets: tile p is merged while tile p + 1 is scanned
    if ((rc = ix->gthr.ensure(64 * 64 + 8 + 64))) return rc;  // [64 queries][64 slots] (csr_mfma.hip); csr_topk.hip uses the first 64 words
    if ((rc = ix->bitmap.ensure((size_t)n_words))) return rc;
    if ((rc = ix->prefix.ensure((size_t)n_words))) return rc;
    if ((rc = ix->W.ensure((size_t)(max_slots + 1) * 64))) return rc;
    if ((rc = ix->qdense.ensure((size_t)nq * V))) return rc;  // every query, dense fp16 (the re-score kernel indexes it by term)
    const size_t out_bytes = (size_t)nq * k * (sizeof(float) + sizeof(long long));
    if ((rc = ix->outbuf.ensure(out_bytes + 16))) return rc;
    float* d_scores = reinterpret_cast<float*>(ix->outbuf.p);
    long long* d_ids = reinterpret_cast<long long*>(ix->outbuf.p + (((size_t)nq * k * sizeof(float) + 15) / 16 * 16));
    if ((size_t)((unsigned char*)d_ids - ix->outbuf.p) + (size_t)nq * k * 8 > ix->outbuf.cap) {
        if ((rc = ix->outbuf.ensure(out_bytes + 64))) return rc;
        d_scores = reinterpret_cast<float*>(ix->outbuf.p);
        d_ids = reinterpret_cast<long long*>(ix->outbuf.p + (((size_t)nq * k * sizeof(float) + 15) / 16 * 16));
    }
    hipStream_t st = ix->stream;
    if (!ix->merge_stream) BH_HIP_TRY(hipStreamCreateWithFlags(&ix->merge_stream, hipStreamNonBlocking));
    hipStream_t mst = ix->merge_stream;
    const unsigned short* q16 = static_cast<const unsigned short*>(q_host);
    const float* q32 = static_cast<const float*>(q_host);
    // non-zero terms of every query, once (fp16 bit patterns)
    std::vector<std::vector<std::pair<int, unsigned short>>> qnz((size_t)nq);
    for (int q = 0; q < nq; ++q) {
        auto& out = qnz[(size_t)q];
        if (q_dtype == BH_F16) {  // the rows are almost all zeros: skip them eight bytes at a time
            const unsigned short* row = q16 + (size_t)q * Vu;
            int t = 0;
            for (; t < Vu && ((uintptr_t)(row + t) & 7u); ++t)
                if (row[t] & 0x7fffu) out.emplace_back(t + 1, row[t]);
            for (; t + 4 <= Vu; t += 4) {
                unsigned long long w;
                memcpy(&w, row + t, 8);
                if ((w & 0x7fff7fff7fff7fffull) == 0) continue;
                for (int u = 0; u < 4; ++u)
                    if (row[t + u] & 0x7fffu) out.emplace_back(t + u + 1, row[t + u]);
            }
            for (; t < Vu; ++t)
                if (row[t] & 0x7fffu) out.emplace_back(t + 1, row[t]);
        } else {
            const float* row = q32 + (size_t)q * Vu;
            for (int t = 0; t < Vu; ++t) {
                if (row[t] == 0.0f) continue;  // (+-0)
                const unsigned short b = f32_to_f16_bits(row[t]);
                if (b & 0x7fffu) out.emplace_back(t + 1, b);
            }
        }
    }
    // options: read once per search, this handle's override first
    const int o_kernel = ix->opt_kernel >= 0 ? ix->opt_kernel : g_sparse_kernel;
    const int o_head = ix->opt_head >= 0 ? ix->opt_head : g_sparse_head;
    const int o_ablate = ix->opt_ablate >= 0 ? ix->opt_ablate : g_sparse_ablate;
    const bool mfma = o_kernel == 1;
    // the MFMA scan on the corpus-head tiles + tail stream: a query's weights of the 64 corpus-head terms go into WgT (matrix
    // cores), only its other terms enter the tile's term set
    const bool use_head = mfma && ix->has_head && o_head != 0;
    std::vector<std::vector<std::pair<int, unsigned short>>> qnz_tail;
    if (use_head) {
        qnz_tail.resize((size_t)nq);
        for (int q = 0; q < nq; ++q)
            for (auto& tv : qnz[(size_t)q])
                if (ix->head_slot[(size_t)tv.first] == 0xffu) qnz_tail[(size_t)q].push_back(tv);
    }
    const auto& qsrc = use_head ? qnz_tail : qnz;  // what the tile's term tables are built from
    const int waves_per_wg = mfma ? 8 : 16;
    if ((rc = ix->cand.ensure((size_t)grid * waves_per_wg * 64 * 2 * kp))) return rc;
    // LDS budget of the MFMA kernel: tables + 8 waves x (4 KiB D tile + 8 KiB S tile)
    const int mfma_fixed = n_words * 6 + 16 + 512 + 2 * 64 * 144 + 32 + 8 * BH_CSR_MFMA_WAVE_LDS;
    const int mfma_table_bytes = kLdsBytes - mfma_fixed;  // for sinfo (4 B per slot) + pairs (4 B per pair)
    if (mfma && mfma_table_bytes < 4096) return bh_fail(BH_EUNSUPPORTED, "vocab %d leaves no LDS for the tile tables", V);

    if (mfma) {  // full size up front: a growing buffer must not be reallocated under a tile that is still running
        if ((rc = ix->sinfo.ensure((size_t)mfma_table_bytes / 4 + 64))) return rc;
        if ((rc = ix->pairs.ensure((size_t)mfma_table_bytes / 4 + 64))) return rc;
        if ((rc = ix->WhT.ensure(64 * 64))) return rc;
        if ((rc = ix->WgT.ensure(2 * 64 * 64))) return rc;  // two sets, like the host tables
    }
    // Host tables come in two sets: while the GPU works on tile p the host builds tile p + 1 (its copies are enqueued
    // behind tile p's kernels); a set is reused only after the tile that used it has completed.
    std::vector<unsigned> bm_s[2] = {std::vector<unsigned>((size_t)n_words), std::vector<unsigned>((size_t)n_words)};
    std::vector<unsigned short> pf_s[2] = {std::vector<unsigned short>((size_t)n_words), std::vector<unsigned short>((size_t)n_words)};
    std::vector<unsigned short> Wh_s[2];
    if (!mfma) Wh_s[0].resize((size_t)(max_slots + 1) * 64), Wh_s[1].resize((size_t)(max_slots + 1) * 64);
    std::vector<unsigned> gt(64 * 64, 0x007fffffu);
    std::vector<int> term_cnt((size_t)V, 0);
    std::vector<unsigned> sinfo_s[2], pairs_s[2];
    std::vector<unsigned short> WhT_s[2] = {std::vector<unsigned short>((size_t)64 * 64), std::vector<unsigned short>((size_t)64 * 64)};
    std::vector<unsigned short> WgT_s[2] = {std::vector<unsigned short>((size_t)64 * 64), std::vector<unsigned short>((size_t)64 * 64)};
    // the dense fp16 query matrix for the re-score kernel, built ON the device from the non-zeros (memset + scatter)
    {
        std::vector<unsigned long long> qpos;
        std::vector<unsigned short> qval;
        for (int q = 0; q < nq; ++q)
            for (auto& tv : qnz[(size_t)q]) {
                qpos.push_back((unsigned long long)q * V + (unsigned long long)tv.first);
                qval.push_back(tv.second);
            }
        const size_t nz = qpos.size();
        if ((rc = ix->W.ensure(std::max((size_t)(max_slots + 1) * 64, (nz * 10 + 64) / 2)))) return rc;  // staging for (pos, val)
        unsigned long long* d_pos = reinterpret_cast<unsigned long long*>(ix->W.p);
        unsigned short* d_val = reinterpret_cast<unsigned short*>(d_pos + nz);
        BH_HIP_TRY(hipMemsetAsync(ix->qdense.p, 0, (size_t)nq * V * 2, st));
        if (nz) {
            BH_HIP_TRY(hipMemcpyAsync(d_pos, qpos.data(), nz * 8, hipMemcpyHostToDevice, st));
            BH_HIP_TRY(hipMemcpyAsync(d_val, qval.data(), nz * 2, hipMemcpyHostToDevice, st));
            BH_HIP_TRY(bh_launch_scatter_f16(reinterpret_cast<unsigned short*>(ix->qdense.p), d_pos, d_val, (int)nz, st));
        }
        BH_HIP_TRY(hipStreamSynchronize(st));  // (qpos / qval are locals; W is reused by the broadcast kernel's tiles)
    }
    struct Pending {
        bool on = false;
        int nt = 0;
    } pend[2];
    double scan_ms = 0, merge_ms = 0, bytes = 0;
    auto collect = [&](int b) -> int {  // wait for the tile that used set b, take its timings
        if (!pend[b].on) return BH_OK;
        hipEvent_t* ev = b ? ix->ev_b : ix->ev;
        BH_HIP_TRY(hipEventSynchronize(ev[3]));
        float ms = 0;
        BH_HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2]));
        scan_ms += ms;
        BH_HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3]));
        merge_ms += ms;
        pend[b].on = false;
        return BH_OK;
    };
    bh_counters& c = ix->counters;
    c = bh_counters{};
    c.n_rows = n_rows_v;
    c.dim = Vu;
    c.dim_padded = Vu;
    c.n_workgroups = grid;
    c.k_padded = kp;
    c.query_tile = kTileQ;
    int n_pass = 0;
    BH_HIP_TRY(hipEventRecord(ix->ev[0], st));
    int q0 = 0;
    while (q0 < nq) {
        const int par = n_pass & 1;
        if ((rc = collect(par))) return rc;
        hipEvent_t* tev = par ? ix->ev_b : ix->ev;
        auto& bm = bm_s[par];
        auto& pf = pf_s[par];
        auto& Wh = Wh_s[par];
        auto& sinfo_h = sinfo_s[par];
        auto& pairs_h = pairs_s[par];
        auto& WhT_h = WhT_s[par];
        // ---- tile: as many queries as fit (<= 64 queries; distinct terms / pairs within the kernel's LDS budget)
        std::fill(bm.begin(), bm.end(), 0u);
        int nt = 0, n_slots = 0, n_pairs_tot = 0;
        while (q0 + nt < nq && nt < kTileQ) {
            const auto& nz = qsrc[(size_t)(q0 + nt)];
            int add = 0;
            for (auto& tv : nz)
                if (!(bm[tv.first >> 5] >> (tv.first & 31) & 1u)) ++add;
            const bool fits = mfma ? ((n_slots + add) * 4 + (n_pairs_tot + (int)nz.size()) * 4 <= mfma_table_bytes)
                                   : (n_slots + add <= max_slots);
            if (!fits) {
                if (nt == 0)
                    return bh_fail(BH_EUNSUPPORTED, "query %d has %d non-zero terms: too many for the LDS tile", q0, (int)qnz[(size_t)q0].size());
                break;
            }
            for (auto& tv : nz) bm[tv.first >> 5] |= 1u << (tv.first & 31);
            n_slots += add;
            n_pairs_tot += (int)nz.size();
            ++nt;
        }
        int run = 0;
        for (int w = 0; w < n_words; ++w) {
            pf[w] = (unsigned short)run;
            run += __builtin_popcount(bm[w]);
        }
        auto slot_of = [&](int t) { return (int)pf[t >> 5] + __builtin_popcount(bm[t >> 5] & ((1u << (t & 31)) - 1u)); };
        BH_HIP_TRY(hipMemcpyAsync(ix->bitmap.p, bm.data(), (size_t)n_words * 4, hipMemcpyHostToDevice, st));
        BH_HIP_TRY(hipMemcpyAsync(ix->prefix.p, pf.data(), (size_t)n_words * 2, hipMemcpyHostToDevice, st));
        BH_HIP_TRY(hipMemcpyAsync(ix->gthr.p, gt.data(), 64 * 64 * 4, hipMemcpyHostToDevice, st));
        BH_HIP_TRY(hipEventRecord(tev[1], st));
        bool floor_zero = false;
        if (!mfma) {
            std::fill(Wh.begin(), Wh.begin() + (size_t)(n_slots + 1) * 64, (unsigned short)0);
            for (int j = 0; j < nt; ++j)
                for (auto& tv : qnz[(size_t)(q0 + j)]) Wh[(size_t)slot_of(tv.first) * 64 + j] = tv.second;
            BH_HIP_TRY(hipMemcpyAsync(ix->W.p, Wh.data(), (size_t)(n_slots + 1) * 128, hipMemcpyHostToDevice, st));
            BhCsrScanArgs sa{};
            sa.entries = ix->entries.p;
            sa.row_ptr = rp_v;
            sa.n_rows = n_rows_v;
            sa.bitmap = ix->bitmap.p;
            sa.prefix = ix->prefix.p;
            sa.W = ix->W.p;
            sa.n_words = n_words;
            sa.n_slots = n_slots;
            sa.off_prefix = off_prefix;
            sa.off_w = off_w;
            sa.off_thr = off_w + (n_slots + 1) * 128;
            sa.cand = ix->cand.p;
            sa.partial = ix->partial.p + (size_t)par * partial_elems;
            sa.gthr = ix->gthr.p;
            BH_HIP_TRY(hipEventRecord(tev[1], st));
            BH_HIP_TRY(bh_launch_csr_scan(sa, kp, grid, (size_t)sa.off_thr + 256, st));
        } else {
            // head = the (up to) 64 terms used by the most queries of the tile; the rest are tail terms with pair lists
            std::vector<int> terms;
            for (int j = 0; j < nt; ++j)
                for (auto& tv : qsrc[(size_t)(q0 + j)])
                    if (term_cnt[(size_t)tv.first]++ == 0) terms.push_back(tv.first);
            std::vector<int> order(terms);
            std::sort(order.begin(), order.end(), [&](int x, int y) {
                return term_cnt[(size_t)x] != term_cnt[(size_t)y] ? term_cnt[(size_t)x] > term_cnt[(size_t)y] : x < y;
            });
            const int n_head = (int)std::min<size_t>(64, order.size());
            sinfo_h.assign((size_t)std::max(1, n_slots), 0u);
            pairs_h.clear();
            std::fill(WhT_h.begin(), WhT_h.end(), (unsigned short)0);
            // tail pair lists: offsets by slot
            std::vector<int> head_idx_of_slot((size_t)std::max(1, n_slots), -1);
            for (int i = 0; i < n_head; ++i) head_idx_of_slot[(size_t)slot_of(order[(size_t)i])] = i;
            std::vector<unsigned> off((size_t)n_slots + 1, 0u);
            for (int t : terms) {
                const int sl = slot_of(t);
                if (head_idx_of_slot[(size_t)sl] < 0) off[(size_t)sl + 1] = (unsigned)term_cnt[(size_t)t];
            }
            for (int i2 = 0; i2 < n_slots; ++i2) off[(size_t)i2 + 1] += off[(size_t)i2];
            pairs_h.assign((size_t)std::max<unsigned>(1u, off[(size_t)n_slots]), 0u);
            std::vector<unsigned> fill(off.begin(), off.end() - 1);
            for (int j = 0; j < nt; ++j)
                for (auto& tv : qsrc[(size_t)(q0 + j)]) {
                    const int sl = slot_of(tv.first);
                    const int hi = head_idx_of_slot[(size_t)sl];
                    if (hi >= 0)
                        WhT_h[(size_t)j * 64 + hi] = tv.second;
                    else
                        pairs_h[(size_t)fill[(size_t)sl]++] = ((unsigned)tv.second << 16) | (unsigned)j;
                }
            for (int i2 = 0; i2 < n_slots; ++i2) {
                const int hi = head_idx_of_slot[(size_t)i2];
                sinfo_h[(size_t)i2] = hi >= 0 ? (0x80000000u | (unsigned)hi) : ((off[(size_t)i2] << 8) | (off[(size_t)i2 + 1] - off[(size_t)i2]));
            }
            for (int t : terms) term_cnt[(size_t)t] = 0;
            const int n_pairs = (int)off[(size_t)n_slots];
            if ((rc = ix->sinfo.ensure(sinfo_h.size()))) return rc;
            if ((rc = ix->pairs.ensure(pairs_h.size()))) return rc;
            if ((rc = ix->WhT.ensure(64 * 64))) return rc;
            BH_HIP_TRY(hipMemcpyAsync(ix->sinfo.p, sinfo_h.data(), sinfo_h.size() * 4, hipMemcpyHostToDevice, st));
            BH_HIP_TRY(hipMemcpyAsync(ix->pairs.p, pairs_h.data(), pairs_h.size() * 4, hipMemcpyHostToDevice, st));
            BH_HIP_TRY(hipMemcpyAsync(ix->WhT.p, WhT_h.data(), 64 * 64 * 2, hipMemcpyHostToDevice, st));
            BhCsrMfmaArgs ma2{};
            ma2.entries = ix->entries.p;
            ma2.row_ptr = rp_v;
            if (use_head) {
                auto& WgT_h = WgT_s[par];
                std::fill(WgT_h.begin(), WgT_h.end(), (unsigned short)0);
                for (int j = 0; j < nt; ++j)
                    for (auto& tv : qnz[(size_t)(q0 + j)]) {
                        const unsigned char hs = ix->head_slot[(size_t)tv.first];
                        if (hs != 0xffu) WgT_h[(size_t)j * 64 + hs] = tv.second;
                    }
                _Float16* wg_dev = ix->WgT.p + (size_t)par * 64 * 64;
                BH_HIP_TRY(hipMemcpyAsync(wg_dev, WgT_h.data(), 64 * 64 * 2, hipMemcpyHostToDevice, st));
                ma2.entries = stream2_v;
                ma2.row_ptr = rp2_v;
                ma2.WgT = wg_dev;
                ma2.head_dwords = BH_CSR_HEAD_DWORDS;
            }
            ma2.n_rows = n_rows_v;
            ma2.bitmap = ix->bitmap.p;
            ma2.prefix = ix->prefix.p;
            ma2.sinfo = ix->sinfo.p;
            ma2.pairs = ix->pairs.p;
            ma2.WhT = ix->WhT.p;
            ma2.n_words = n_words;
            ma2.n_slots = n_slots;
            ma2.n_pairs = n_pairs;
            ma2.off_prefix = n_words * 4;
            ma2.off_sinfo = (ma2.off_prefix + n_words * 2 + 15) / 16 * 16;
            ma2.off_pairs = ma2.off_sinfo + n_slots * 4;
            ma2.off_thr = (ma2.off_pairs + n_pairs * 4 + 15) / 16 * 16;
            ma2.off_tiles = (ma2.off_thr + 512 + 2 * 64 * 144 + 15) / 16 * 16;  // published[64] bound[64] | WhT image | WgT image | per-wave tiles
            ma2.cand = ix->cand.p;
            ma2.partial = ix->partial.p + (size_t)par * partial_elems;
            ma2.gthr = ix->gthr.p;
            ma2.ablate = o_ablate;
            // Non-negative corpus and tile: scores are >= 0, a zero-score document can only enter a top-k that has fewer
            // than k positive documents, and then it is simply one of the lowest row ids.  The scan then never collects
            // zero-score documents (threshold floor 0, exclusive) and the merge fills short lists with the lowest absent rows.
            bool nonneg_q = true;
            for (int j = 0; j < nt && nonneg_q; ++j)
                for (auto& tv : qnz[(size_t)(q0 + j)])
                    if (tv.second & 0x8000u) {
                        nonneg_q = false;
                        break;
                    }
            floor_zero = ix->nonneg_docs && nonneg_q && !getenv("BH_SPARSE_NO_FLOOR");
            ma2.floor_zero = floor_zero ? 1 : 0;
            static const int stats_mode = getenv("BH_SPARSE_STATS") ? atoi(getenv("BH_SPARSE_STATS")) : 0;  // diagnostics
            const bool want_stats = stats_mode != 0;
            ma2.stats_mode = stats_mode;
            ma2.stats = want_stats ? ix->gthr.p + 64 * 64 : nullptr;
            if (want_stats) BH_HIP_TRY(hipMemsetAsync(ma2.stats, 0, 72 * sizeof(unsigned), st));
            const size_t smem2 = (size_t)ma2.off_tiles + 8 * BH_CSR_MFMA_WAVE_LDS;
            if (smem2 > (size_t)kLdsBytes) return bh_fail(BH_EHIP, "internal: sparse tile exceeds LDS (%zu bytes)", smem2);
            BH_HIP_TRY(hipEventRecord(tev[1], st));
            // Pre-pass: the same kernel over a short prefix of the corpus with a small grid, only to fill the threshold
            // slot table (its candidate lists are overwritten by the main launch, which scans the prefix again).
            // 64 workgroups: one per slot of the table (a slot nobody wrote leaves the bound open)
            const long long pre_rows = 64LL * 8 * 2 * 32;  // 64 workgroups x 8 waves x 2 groups
            if (n_rows_v > 4 * pre_rows && grid >= 64) {
                BhCsrMfmaArgs pre = ma2;
                pre.n_rows = pre_rows;
                pre.skip_final = 1;
                BH_HIP_TRY(bh_launch_csr_scan_mfma(pre, kp, 64, smem2, st));
            }
            BH_HIP_TRY(bh_launch_csr_scan_mfma(ma2, kp, grid, smem2, st));
        }
        BH_HIP_TRY(hipEventRecord(tev[2], st));
        BhCsrMergeArgs ma{};
        ma.floor_zero = floor_zero ? 1 : 0;
        if (floor_flags)
            for (int j = 0; j < nt; ++j) floor_flags[q0 + j] = floor_zero ? 1 : 0;
        ma.partial = ix->partial.p + (size_t)par * partial_elems;
        ma.n_lists = grid;
        ma.entries = ix->entries.p;
        ma.row_ptr = rp_v;
        ma.n_rows = n_rows_v;
        ma.q_dense = ix->qdense.p + (size_t)q0 * V;
        ma.vocab = V;
        ma.k = k;
        ma.id_offset = id_offset;
        ma.out_scores = d_scores + (size_t)q0 * k;
        ma.out_ids = d_ids + (size_t)q0 * k;
        BH_HIP_TRY(hipStreamWaitEvent(mst, tev[2], 0));  // side stream: behind this tile's scan, beside the next tile's
        BH_HIP_TRY(bh_launch_csr_merge_rescore(ma, kp, nt, mst));
        BH_HIP_TRY(hipEventRecord(tev[3], mst));
        pend[par].on = true;
        pend[par].nt = nt;
        if (mfma && getenv("BH_SPARSE_STATS")) {
            BH_HIP_TRY(hipStreamSynchronize(st));
            unsigned stv[72];
            BH_HIP_TRY(hipMemcpy(stv, ix->gthr.p + 64 * 64, sizeof(stv), hipMemcpyDeviceToHost));
            fprintf(stderr, "[bh sparse stats] groups_with_hits=%u appended=%u compactions=%u polls=%u\n", stv[0], stv[1], stv[2], stv[3]);
            for (int w = 0; w < 8; ++w) {  // sampled waves (BH_CSR_TIMERS builds): s_memtime ticks
                const unsigned* o = stv + 8 + w * 8;
                if (o[0])
                    fprintf(stderr, "[bh sparse wave %d] total=%u entries=%u appends=%u (groups 0-3: %u, 4-15: %u) final thr q0=%g q5=%g\n",
                            w * 256, o[0], o[6], o[7], o[4], o[5], *reinterpret_cast<const float*>(&o[2]), *reinterpret_cast<const float*>(&o[3]));
            }
        }
        // nnz*(2+2) + (N+1)*8 per query-tile pass (+ the tile's results)
        // (a view's share of the entries is taken as proportional to its documents: the exact count is on the device)
        bytes += (double)ix->nnz * 4.0 * ((double)n_rows_v / (double)std::max<int64_t>(1, ix->n_rows)) + (double)(n_rows_v + 1) * 8.0 + (double)nt * k * 12.0;
        ++n_pass;
        q0 += nt;
    }
    if ((rc = collect(0))) return rc;
    if ((rc = collect(1))) return rc;
    BH_HIP_TRY(hipMemcpyAsync(out_scores, d_scores, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    BH_HIP_TRY(hipMemcpyAsync(out_ids, d_ids, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, st));
    BH_HIP_TRY(hipStreamSynchronize(st));
    c.n_passes = n_pass;
    c.scan_ms = scan_ms;
    c.merge_ms = merge_ms;
    c.total_ms = scan_ms + merge_ms;
    c.algorithmic_bytes = bytes;
    return BH_OK;
}

// k > 120 (the reference accepts any top_k_documents, modules/retrieve.py:157): as for the dense index (index.hip,
// search_large_k) the corpus is cut into contiguous RANGES of 32-document groups small enough that a range is expected to
// hold ~50 of the top k, every range is searched for its exact top 120 by the ordinary fused search over a view, and the
// ranges' lists are merged in canonical order (score descending, row ascending).  Exact iff no range holds more than 120 of
// the true top k — CHECKED: a range whose list came back full and whose last entry still belongs to the merged top k may
// have dropped a row; it is split in four and searched again, down to single groups (32 documents cannot fill 120 slots).
// Non-negative data: a range's list continues with its lowest zero-score rows, so a query with fewer than k matching
// documents drills into the FIRST ranges until the lowest absent rows of the whole corpus are all listed.
static int sparse_search_large_k(bh_sparse_index* ix, const void* q_host, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                                 float* out_scores, int64_t* out_ids) {
    constexpr int KK = kSparseListK;
    // (as index.hip search_large_k: once split, a range holds lists only for the queries that overflowed in its parent)
    struct Range {
        int64_t g0, g1;  // 32-document groups [g0, g1)
        std::vector<int> qs;    // the queries this range holds lists for (ascending)
        std::vector<int> qpos;  // [nq] position of query q in qs, -1 = none
        std::vector<float> s;
        std::vector<long long> i;
    };
    const int64_t n_rows = ix->n_rows, n_groups = (n_rows + 31) / 32;
    const size_t qrow_bytes = (size_t)ix->vocab * (q_dtype == BH_F16 ? 2 : 4);
    std::vector<unsigned char> qsub;  // gathered query rows of a subset search
    bh_counters total{};
    // Non-negative data (the usual SPLADE case; per query tile, see the view search): a range's list holds every document with
    // a POSITIVE score first and continues with its lowest zero-score rows.  Zero-score rows need no search — they are the
    // rows that are not positive, in row order — so such a list has dropped something only if its LAST entry is still
    // positive, and a query with fewer than k matching documents is completed here from the lowest absent rows of the whole
    // corpus (without this, every range would be drilled down to single groups looking for more zeros).
    std::vector<char> floor_q((size_t)nq, 0), floor_sub;
    auto search_range = [&](Range& r) -> int {
        const int nqs = (int)r.qs.size();
        const int64_t lo = r.g0 * 32, hi = std::min<int64_t>(n_rows, r.g1 * 32);
        r.s.assign((size_t)nqs * KK, -INFINITY);
        r.i.assign((size_t)nqs * KK, -1);
        const void* qptr = q_host;
        if (nqs != nq) {
            qsub.resize((size_t)nqs * qrow_bytes);
            for (int j = 0; j < nqs; ++j)
                memcpy(qsub.data() + (size_t)j * qrow_bytes, (const unsigned char*)q_host + (size_t)r.qs[(size_t)j] * qrow_bytes, qrow_bytes);
            qptr = qsub.data();
        }
        floor_sub.assign((size_t)nqs, 0);
        const int rc = sparse_search_view(ix, lo, hi - lo, qptr, q_dtype, nqs, KK, id_offset + lo, r.s.data(), reinterpret_cast<int64_t*>(r.i.data()),
                                          floor_sub.data());
        if (rc != BH_OK) return rc;
        for (int j = 0; j < nqs; ++j) floor_q[(size_t)r.qs[(size_t)j]] = floor_sub[(size_t)j];  // (a property of the query and the corpus: the same in every range)
        const bh_counters& c = ix->counters;
        total.scan_ms += c.scan_ms;
        total.merge_ms += c.merge_ms;
        total.total_ms += c.total_ms;
        total.algorithmic_bytes += c.algorithmic_bytes;
        total.n_passes += c.n_passes;
        total.n_workgroups = c.n_workgroups;
        total.k_padded = c.k_padded;
        total.query_tile = c.query_tile;
        total.dim = c.dim;
        total.dim_padded = c.dim_padded;
        return BH_OK;
    };
    auto make_range = [&](int64_t g0, int64_t g1, const std::vector<int>& qs) {
        Range r{g0, g1, qs, std::vector<int>((size_t)nq, -1), {}, {}};
        for (size_t j = 0; j < qs.size(); ++j) r.qpos[(size_t)qs[j]] = (int)j;
        return r;
    };
    std::vector<Range> ranges;
    {
        std::vector<int> all_q((size_t)nq);
        for (int q = 0; q < nq; ++q) all_q[(size_t)q] = q;
        const int64_t want = std::max<int64_t>(1, std::min<int64_t>(n_groups, (k + 49) / 50));
        const int64_t per = (n_groups + want - 1) / want;
        for (int64_t g = 0; g < n_groups; g += per) ranges.push_back(make_range(g, std::min(n_groups, g + per), all_q));
    }
    int rc = BH_OK;
    for (auto& r : ranges)
        if ((rc = search_range(r)) != BH_OK) return rc;
    struct Ent {
        float s;
        long long id;
    };
    std::vector<Ent> all;
    constexpr int kMaxRounds = 64;
    for (int round = 0;; ++round) {
        std::vector<std::vector<int>> over(ranges.size());
        bool any = false;
        for (int q = 0; q < nq; ++q) {
            const bool floor = floor_q[(size_t)q] != 0;
            all.clear();
            for (auto& r : ranges) {
                const int at = r.qpos[(size_t)q];
                if (at < 0) continue;
                for (int t = 0; t < KK; ++t) {
                    const long long id = r.i[(size_t)at * KK + t];
                    const float sc = r.s[(size_t)at * KK + t];
                    if (id >= 0 && !(floor && !(sc > 0.f))) all.push_back(Ent{sc, id});  // (floor: the zero-score fill is rebuilt below)
                }
            }
            std::sort(all.begin(), all.end(), [](const Ent& a, const Ent& b) { return a.s != b.s ? a.s > b.s : a.id < b.id; });
            const size_t take = std::min<size_t>(all.size(), (size_t)k);
            for (size_t t = 0; t < (size_t)k; ++t) {
                out_scores[(size_t)q * k + t] = t < take ? all[t].s : -INFINITY;
                out_ids[(size_t)q * k + t] = t < take ? all[t].id : -1;
            }
            if (floor && take < (size_t)k) {
                // fewer than k documents match: the canonical order continues with the zero-score rows by ascending row —
                // the lowest rows that are not among the positive ones (at most `take` of the first k rows are)
                std::vector<long long> pos;
                pos.reserve(take);
                for (size_t t = 0; t < take; ++t) pos.push_back(all[t].id - id_offset);
                std::sort(pos.begin(), pos.end());
                size_t filled = take, pi = 0;
                for (long long row = 0; row < n_rows && filled < (size_t)k; ++row) {
                    while (pi < pos.size() && pos[pi] < row) ++pi;
                    if (pi < pos.size() && pos[pi] == row) continue;
                    out_scores[(size_t)q * k + filled] = 0.f;
                    out_ids[(size_t)q * k + filled] = id_offset + row;
                    ++filled;
                }
            }
            for (size_t j = 0; j < ranges.size(); ++j) {
                const int at = ranges[j].qpos[(size_t)q];
                if (at < 0) continue;
                const long long last_id = ranges[j].i[(size_t)at * KK + KK - 1];
                if (last_id < 0) continue;  // not full: the list holds every document of the range that can matter
                const float last_s = ranges[j].s[(size_t)at * KK + KK - 1];
                if (floor && !(last_s > 0.f)) continue;  // its positive documents are all listed; zeros are rebuilt above
                const bool last_in_topk = take < (size_t)k || last_s > all[take - 1].s || (last_s == all[take - 1].s && last_id <= all[take - 1].id);
                if (last_in_topk && ranges[j].g1 - ranges[j].g0 > 1) {
                    over[j].push_back(q);
                    any = true;
                }
            }
        }
        if (!any) break;
        if (round + 1 >= kMaxRounds)  // (cannot happen: a range shrinks fourfold per round down to one group, which cannot overflow)
            return bh_fail(BH_EUNSUPPORTED, "k = %d: the sparse range search did not converge in %d rounds", k, kMaxRounds);
        std::vector<Range> next;
        for (size_t j = 0; j < ranges.size(); ++j) {
            if (over[j].empty()) {
                next.push_back(std::move(ranges[j]));
                continue;
            }
            const int64_t span = ranges[j].g1 - ranges[j].g0, per = (span + 3) / 4;
            for (int64_t g = ranges[j].g0; g < ranges[j].g1; g += per) {
                Range r = make_range(g, std::min(ranges[j].g1, g + per), over[j]);
                if ((rc = search_range(r)) != BH_OK) return rc;
                next.push_back(std::move(r));
            }
            for (int q : over[j]) ranges[j].qpos[(size_t)q] = -1;  // the parent keeps its lists for the queries it did not overflow for
            if (over[j].size() < ranges[j].qs.size()) next.push_back(std::move(ranges[j]));
        }
        ranges.swap(next);
    }
    total.n_rows = n_rows;
    ix->counters = total;
    return BH_OK;
}

int bh_sparse_search(bh_sparse_index* ix, const void* q_host, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                     float* out_scores, int64_t* out_ids) {
    if (!ix) return bh_fail(BH_EINVAL, "null index");
    if (k > BH_MAX_K) return bh_fail(BH_EUNSUPPORTED, "k=%d unsupported for sparse search (max %d)", k, BH_MAX_K);
    const bool valid = ix->finalized && nq > 0 && (q_dtype == BH_F16 || q_dtype == BH_F32) && q_host && out_scores && out_ids;
    if (k > kSparseListK && valid) return sparse_search_large_k(ix, q_host, q_dtype, nq, k, id_offset, out_scores, out_ids);
    // (k <= 120, or a call the view search rejects / answers trivially — incomplete index, bad arguments, nq = 0 — in the usual order)
    return sparse_search_view(ix, 0, ix->n_rows, q_host, q_dtype, nq, std::min<int32_t>(k, kSparseListK), id_offset, out_scores, out_ids);
}

int bh_sparse_set_option(bh_sparse_index* ix, const char* name, int64_t value) {
    if (!ix) return bh_fail(BH_EINVAL, "null index");
    if (!name) return bh_fail(BH_EINVAL, "null option name");
    const bool inherit = value == BH_OPTION_INHERIT;
    if (strcmp(name, "sparse_kernel") == 0) {
        if (!inherit && value != 0 && value != 1) return bh_fail(BH_EINVAL, "sparse_kernel must be 0 (broadcast) or 1 (mfma)");
        ix->opt_kernel = inherit ? -1 : (int)value;
    } else if (strcmp(name, "sparse_head") == 0) {
        if (!inherit && value != 0 && value != 1) return bh_fail(BH_EINVAL, "sparse_head must be 0 (plain CSR stream) or 1 (corpus-head tiles + tail stream)");
        ix->opt_head = inherit ? -1 : (int)value;
    } else if (strcmp(name, "sparse_ablate") == 0) {
        if (!inherit && (value < 0 || value > 2047)) return bh_fail(BH_EINVAL, "sparse_ablate must be 0..2047");
        ix->opt_ablate = inherit ? -1 : (int)value;
    } else {
        return bh_fail(BH_EINVAL, "unknown per-index sparse option '%s'", name);
    }
    return BH_OK;
}

int bh_sparse_counters(const bh_sparse_index* ix, bh_counters* out) {
    if (!ix || !out) return bh_fail(BH_EINVAL, "null argument");
    if (!bh_copy_sized(out, ix->counters, 16))
        return bh_fail(BH_EINVAL, "bh_counters.struct_size = %d: set it to sizeof(bh_counters) before the call (BH_VERSION %d)", out->struct_size, BH_VERSION);
    return BH_OK;
}

}  // extern "C"
