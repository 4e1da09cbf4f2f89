// convert.hip — upload-side helpers: dtype conversion + zero padding of rows to the kernel's
// row stride, one-off L2 normalisation for cosine indexes, u32 fill.
//
// bh_convert_rows:  [n, dim] (fp16 | fp32, dense) -> [n, dim_padded] fp16, RNE like torch .half()
//   (the reference stores fp16 embeddings: models/retrievers/dense.py:16).  HBM-bound, 8 halfs
//   (16 B) stored per thread.
// bh_l2_normalize_rows:  replaces the per-call renormalisation in CosineSim.sim (reference
//   models/retrievers/dense.py:87-88).  Canonical definition (restated by the oracle):
//   n2 = sequential fp64 sum of x_j^2; y_j = fp16(RNE( fp32(RNE( double(x_j) * (1/sqrt(n2)) )) )), zero rows
//   stay zero.  One thread per row so the fp64 summation order is the oracle's.
#include "bh_device.h"
#include "bh_kernels.h"

template <typename SRC>
__global__ void __launch_bounds__(256) bh_convert_rows_kernel(const SRC* __restrict__ src, long long n, int dim,
                                                               _Float16* __restrict__ dst, int dim_padded) {
    const int chunks = dim_padded >> 3;  // 8-half chunks per row
    const long long total = n * chunks;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / chunks;
        const int c = (int)(i - row * chunks) << 3;
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = c + e;
            v[e] = col < dim ? (_Float16)src[row * dim + col] : (_Float16)0.f;
        }
        *reinterpret_cast<half8*>(dst + row * dim_padded + c) = v;
    }
}

hipError_t bh_launch_convert_rows(const void* src, int src_dtype, long long n, int dim, _Float16* dst,
                                  int dim_padded, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const long long total = n * (dim_padded >> 3);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (src_dtype == 0)
        hipLaunchKernelGGL(bh_convert_rows_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const _Float16*)src, n, dim, dst, dim_padded);
    else if (src_dtype == 1)
        hipLaunchKernelGGL(bh_convert_rows_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const float*)src, n, dim, dst, dim_padded);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) bh_l2_normalize_rows_kernel(_Float16* rows, long long n, int dim,
                                                                    int dim_padded) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n;
         r += (long long)gridDim.x * blockDim.x) {
        _Float16* x = rows + r * dim_padded;
        double n2 = 0.0;
        for (int j = 0; j < dim; ++j) {
            const double v = (double)x[j];
            n2 = __builtin_fma(v, v, n2);  // v*v is exact in fp64
        }
        if (n2 > 0.0) {
            const double inv = 1.0 / __builtin_sqrt(n2);
            for (int j = 0; j < dim; ++j) {
                float y = (float)((double)x[j] * inv);  // fp64 -> fp32, RNE (v_cvt_f32_f64)
                // hipcc otherwise folds the two casts into ONE fp64->fp16 rounding (different bits in
                // ~2^-13 of the elements): keep the contract's two roundings
                asm volatile("" : "+v"(y));
                x[j] = (_Float16)y;  // fp32 -> fp16, RNE (v_cvt_f16_f32)
            }
        }
    }
}

hipError_t bh_launch_l2_normalize_rows(_Float16* rows, long long n, int dim, int dim_padded, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(bh_l2_normalize_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, n, dim,
                       dim_padded);
    return hipGetLastError();
}

// p[i] = v, except the word at offset `special_at` of every block of `period` words (period > 0), which gets v_special
__global__ void bh_fill_u32_kernel(unsigned* p, long long n, unsigned v, long long period, long long special_at, unsigned v_special) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = (period > 0 && i % period == special_at) ? v_special : v;
}

hipError_t bh_launch_fill_u32(unsigned* p, long long n, unsigned v, hipStream_t stream, long long period, long long special_at,
                              unsigned v_special) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bh_fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n, v, period, special_at, v_special);
    return hipGetLastError();
}
