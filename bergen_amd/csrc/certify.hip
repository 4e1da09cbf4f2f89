// certify.hip — the exactness certificate's two helpers.
//
// The scan ranks rows by their fp32 MFMA score and keeps KP >= k + 8 candidates per query; merge_rescore.hip re-scores
// those canonically (sequential fp64 sum -> fp32) and cuts to k.  That is exact unless a row of the true top-k sits
// beyond rank KP in MFMA order, i.e. unless more than KP - k rows lie within the MFMA rounding error of the k-th
// score.  merge_rescore.hip certifies each query: a dropped row x has  mfma(x) <= mfma(KP-th kept row)  and
// |mfma(x) - canonical(x)| <= 2 d 2^-24 |q| |x| (fp32 accumulation of d exact products, any order, any tree), so its
// canonical score is below  mfma(KP-th) + 2 d 2^-24 |q| max|x|.  If that is still below the k-th canonical score the
// result is proven exact; otherwise the query goes through bh_exact_scan_kernel, which computes the canonical score of
// EVERY row and returns all rows whose canonical key reaches the k-th kept one — exact by construction, at the price of
// one more corpus pass per BH_EXACT_BATCH uncertified queries.
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

// One thread per row, BH_EXACT_BATCH queries at a time; the queries' fp64 images sit in LDS (same address for every lane:
// broadcast reads).  Canonical score = fp32 of the sequential fp64 FMA sum in dimension order — the arithmetic of
// merge_rescore.hip and of the oracle's plain C loop.
__global__ void __launch_bounds__(256) bh_exact_scan_kernel(BhExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* qd = reinterpret_cast<double*>(smem_raw);  // [nqf][D]
    const int D = a.dim_padded;
    for (int i = threadIdx.x; i < a.nqf * D; i += blockDim.x) qd[i] = (double)a.q[i];
    __syncthreads();
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_rows; r += (long long)gridDim.x * blockDim.x) {
        const half8* x = reinterpret_cast<const half8*>(a.corpus + (size_t)r * D);
        double s[BH_EXACT_BATCH];
#pragma unroll
        for (int qi = 0; qi < BH_EXACT_BATCH; ++qi) s[qi] = 0.0;
        for (int c = 0; c < (D >> 3); ++c) {
            const half8 xv = x[c];
            double xd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xd[e] = (double)xv[e];
#pragma unroll
            for (int qi = 0; qi < BH_EXACT_BATCH; ++qi) {
                if (qi < a.nqf) {
                    const double* qq = qd + (size_t)qi * D + c * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[qi] = __builtin_fma(qq[e], xd[e], s[qi]);
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < BH_EXACT_BATCH; ++qi) {
            if (qi < a.nqf) {
                const u64 key = bh_make_key((float)s[qi], (unsigned)r);
                if (key >= a.kth_key[qi]) {
                    const unsigned slot = atomicAdd(a.out_cnt + qi, 1u);
                    if (slot < BH_EXACT_CAP) a.out_keys[(size_t)qi * BH_EXACT_CAP + slot] = key;
                }
            }
        }
    }
}

hipError_t bh_launch_exact_scan(const BhExactArgs& a, hipStream_t stream) {
    if (a.nqf <= 0 || a.nqf > BH_EXACT_BATCH || a.n_rows <= 0) return hipErrorInvalidValue;
    const size_t smem = (size_t)a.nqf * a.dim_padded * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bh_exact_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           BH_EXACT_BATCH * 1024 * (int)sizeof(double));
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    long long blocks = (a.n_rows + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(bh_exact_scan_kernel, dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return hipGetLastError();
}
