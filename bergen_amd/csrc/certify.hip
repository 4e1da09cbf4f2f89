// certify.hip — the exactness certificate's two helpers.
//
// The scan ranks rows by their fp32 MFMA score and keeps KP >= k + 8 candidates per query; merge_rescore.hip re-scores
// those canonically (sequential fp64 sum -> fp32) and cuts to k.  That is exact unless a row of the true top-k sits
// beyond rank KP in MFMA order, i.e. unless more than KP - k rows lie within the MFMA rounding error of the k-th
// score.  merge_rescore.hip certifies each query: a dropped row x has  mfma(x) <= mfma(KP-th kept row)  and
// |mfma(x) - canonical(x)| <= 2 d 2^-24 |q| |x| (fp32 accumulation of d exact products, any order, any tree), so its
// canonical score is below  mfma(KP-th) + 2 d 2^-24 |q| max|x|.  If that is still below the k-th canonical score the
// result is proven exact; otherwise the query goes through the fall-back below (an MFMA filter pass with a fixed threshold +
// canonical re-scoring of the few rows it lets through) — exact by construction, at the price of one more corpus pass per
// BH_EXACT_BATCH (128) uncertified queries.
//
// Reference lines this protects: torch.topk over the exact score matrix (modules/retrieve.py:157,175).
#include "bh_device.h"
#include "bh_kernels.h"

// max_r |x_r|^2 (fp32, rounded up a little) as float bits; non-negative floats order like their bit patterns
__global__ void __launch_bounds__(256) bh_row_norm_max_kernel(const _Float16* rows, long long n, int dim_padded, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int chunks = dim_padded >> 3;
    float best = 0.f;
    for (long long r = wave; r < n; r += n_waves) {
        const half8* x = reinterpret_cast<const half8*>(rows + (size_t)r * dim_padded);
        float s = 0.f;
        for (int c = lane; c < chunks; c += 64) {
            const half8 v = x[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        best = fmaxf(best, s);
    }
    if (lane == 0) atomicMax(out, __float_as_uint(best * 1.001f));
}

hipError_t bh_launch_row_norm_max(const _Float16* rows, long long n, int dim_padded, unsigned* out_max_bits, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n * 64 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(bh_row_norm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, n, dim_padded, out_max_bits);
    return hipGetLastError();
}

// ---- the fall-back for queries the certificate could not prove --------------------------------------------------
// A row x can belong to the true top-k of query q only if its canonical key reaches the k-th kept one, and
// canonical(x) <= mfma(x) + err_coef |q| (the certificate's own bound), so every such row has
//     mfma(x) >= score(k-th canonical key) - err_coef |q|  =: fix_thr[q].
// The FILTER PASS (scan_topk256.hip, ABL bit 64: the 256-query scan's stream + MFMA loop with that fixed threshold — 7.5 ms per
// pass at 21 M x 768; where that kernel's tile is no wider, d = 1024, and for the other dims: scan_topk.hip, ABL = 5, 128 queries) lists these
// rows — the top k and whatever lies within rounding error of the k-th score — for up to BH_EXACT_BATCH queries per
// corpus pass, at the speed of a normal pass; bh_exact_rescore_kernel then gives each listed row its canonical score
// (sequential fp64 sum in dimension order -> fp32: the arithmetic of merge_rescore.hip and of the oracle's plain C loop)
// and keeps those whose key reaches the k-th one; the host sorts.  Exact by construction, whatever the cluster size (up to
// BH_EXACT_CAP rows per query).  (Round 2 ran an fp64 scan of EVERY row for 8 queries at a time: one thread per row,
// uncoalesced, ~fp64-peak-bound at best — a corpus pass per 8 queries; the matrix cores do the same filtering for 128.)

// Gathers the batch: query rows, k-th keys, thresholds.  One workgroup per slot of the filter kernel's query tile (128 or 256).
__global__ void __launch_bounds__(256) bh_exact_prepare_kernel(const _Float16* qbuf, const int* todo, int nb, const bh_u64* kth_all,
                                                               float err_coef, int dim_padded, _Float16* q_out, bh_u64* kth_out,
                                                               float* thr_out) {
    __shared__ float part[4];
    const int j = blockIdx.x, tid = threadIdx.x;
    _Float16* dst = q_out + (size_t)j * dim_padded;
    if (j >= nb) {  // unused query of the tile: zero row, nothing qualifies
        for (int c = tid; c < dim_padded; c += 256) dst[c] = (_Float16)0.f;
        if (tid == 0) thr_out[j] = __builtin_inff();
        return;
    }
    const int q = todo[j];
    const _Float16* src = qbuf + (size_t)q * dim_padded;
    float s2 = 0.f;
    for (int c = tid; c < dim_padded; c += 256) {
        const _Float16 v = src[c];
        dst[c] = v;
        s2 += (float)v * (float)v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = s2;
    __syncthreads();
    if (tid == 0) {
        const bh_u64 kth = kth_all[q];
        kth_out[j] = kth;
        // the certificate's bound (merge_rescore.hip: same over-estimate of |q|), rounded towards -inf twice over
        const float qn = sqrtf(part[0] + part[1] + part[2] + part[3]) * 1.0001f;
        const float lim = bh_key_score(kth) - err_coef * qn * 1.0001f;
        // (next float below lim: floats order like the bh_ordf image of their bits)
        thr_out[j] = kth != 0ull ? bh_unordf(bh_ordf(lim) - 1u) : -__builtin_inff();
    }
}

hipError_t bh_launch_exact_prepare(const _Float16* qbuf, const int* todo, int nb, int tile, const bh_u64* kth_all, float err_coef, int dim_padded,
                                   _Float16* q_out, bh_u64* kth_out, float* thr_out, hipStream_t stream) {
    if (tile <= 0 || tile > BH_EXACT_BATCH || nb > tile) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bh_exact_prepare_kernel, dim3((unsigned)tile), dim3(256), 0, stream, qbuf, todo, nb, kth_all, err_coef, dim_padded,
                       q_out, kth_out, thr_out);
    return hipGetLastError();
}

// One thread per listed (query, row): canonical score, key, keep iff it reaches the query's k-th key.
__global__ void __launch_bounds__(256) bh_exact_rescore_kernel(BhExactArgs a) {
    const int qi = blockIdx.y;
    const unsigned n = min(a.cnt[qi], (unsigned)BH_EXACT_CAP);
    const half8* qv = reinterpret_cast<const half8*>(a.q + (size_t)qi * a.dim_padded);
    const bh_u64 kth = a.kth_key[qi];
    const int n8 = a.dim_padded >> 3;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned row = a.rows[(size_t)qi * BH_EXACT_CAP + i];
        const half8* x = reinterpret_cast<const half8*>(a.corpus + (size_t)row * a.dim_padded);
        double s = 0.0;
        for (int j = 0; j < n8; j += 4) {  // (dim_padded is a multiple of 32)
            half8 xv[4], qq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = x[j + u];
                qq[u] = qv[j + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) s = __builtin_fma((double)qq[u][e], (double)xv[u][e], s);
        }
        const u64 key = bh_make_key((float)s, row);
        a.out_keys[(size_t)qi * BH_EXACT_CAP + i] = key >= kth ? key : 0ull;
    }
}

hipError_t bh_launch_exact_rescore(const BhExactArgs& a, hipStream_t stream) {
    if (a.nqf <= 0 || a.nqf > BH_EXACT_BATCH || a.n_rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bh_exact_rescore_kernel, dim3(32, (unsigned)a.nqf), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// The fall-back's re-done lists -> their rows of the result buffers: list j of `src` goes to row todo[j] (the buffers may be
// device memory or pinned host memory: the kernel writes either).
__global__ void __launch_bounds__(256) bh_scatter_lists_kernel(const float* src_s, const long long* src_i, const int* todo, int k,
                                                               float* out_s, long long* out_i) {
    const int j = blockIdx.x, q = todo[j];
    for (int t = threadIdx.x; t < k; t += 256) {
        out_s[(size_t)q * k + t] = src_s[(size_t)j * k + t];
        out_i[(size_t)q * k + t] = src_i[(size_t)j * k + t];
    }
}

hipError_t bh_launch_scatter_lists(const float* src_s, const long long* src_i, const int* todo, int n, int k, float* out_s, long long* out_i,
                                   hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_scatter_lists_kernel, dim3((unsigned)n), dim3(256), 0, stream, src_s, src_i, todo, k, out_s, out_i);
    return hipGetLastError();
}
