// encoder_ops.hip — the HBM-bound row kernels of the bi-encoder forward pass.  One wave per token row,
// 16-byte accesses, fp32 math, fp16 storage.
//
//   bh_embed_ln_kernel   word + position + token-type embedding gather, sum, LayerNorm
//                        (BertEmbeddings.forward, transformers modeling_bert.py; reached from the
//                        reference through AutoModel, models/retrievers/dense.py:16)
//   bh_layernorm_kernel  LayerNorm(dense output + residual) (BertSelfOutput / BertOutput: LayerNorm(dense(x) +
//                        input)); the residual is added here, in fp32, rather than in the GEMM epilogue
//   bh_pool_kernel       ClsPooler.pool / MeanPooler.pool (reference models/retrievers/dense.py:64-75) over
//                        the packed rows of each sequence, optional L2 normalisation, fp16 [B][d] out
//   bh_unpack_kernel     packed rows -> padded [B][T][d] last_hidden_state (API compatibility: a stock
//                        pooler / stock BERGEN Retrieve can consume it); padding positions are zero
//   bh_rotary_kernel     rotary positions of NomicBert (transformers modeling_nomic_bert.py:150-181, apply_rotary_pos_emb with
//                        rotate_half): every 64-dim head slice of a token's query and key, in place in the Q | K buffer
//   bh_swiglu_kernel     gated feed-forward of NomicBert (NomicBertMLP.forward, :266-279): silu(gate) * up over the interleaved
//                        (gate, up) columns of ONE GEMM, fp32 math — the fallback of the GEMM's own fused fold (BH_EPI_SWIGLU)
#include "bh_device.h"
#include "bh_kernels.h"
#include "gemm_f16_kernel.h"  // bh_gemm::gelu_erf (the GEMM epilogues' GELU: the gated fold below must give the same bits)

namespace {

constexpr int MAXC = 4;  // 8-half chunks per lane: rows up to 64*8*4 = 2048 elements

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// x[c][e]: the lane's elements (chunk c = lane + 64 c).  Normalises in place.
__device__ __forceinline__ void row_layernorm(float (&x)[MAXC][8], int nchunk, int lane, int d, float eps,
                                              const _Float16* gamma, const _Float16* beta) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[c][e];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = x[c][e] - mean;
                q += t * t;
            }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nchunk) {
            const half8 g = *reinterpret_cast<const half8*>(gamma + (size_t)(lane + 64 * c) * 8);
            const half8 b = *reinterpret_cast<const half8*>(beta + (size_t)(lane + 64 * c) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] = (x[c][e] - mean) * rstd * (float)g[e] + (float)b[e];
        }
}

__device__ __forceinline__ void store_row(_Float16* dst, const float (&x)[MAXC][8], int nchunk, int lane) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nchunk) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)x[c][e];
            *reinterpret_cast<half8*>(dst + (size_t)(lane + 64 * c) * 8) = o;
        }
}

}  // namespace

__global__ void __launch_bounds__(256) bh_embed_ln_kernel(BhEmbedArgs a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int nchunk = a.d >> 3;
    const int tok = a.tok[row], pos = a.pos[row], typ = a.typ[row];
    const _Float16* w = a.word + (size_t)tok * a.d;
    const _Float16* p = a.position + (size_t)pos * a.d;
    const _Float16* t = a.type + (size_t)typ * a.d;
    float x[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nchunk) {
            const size_t o = (size_t)(lane + 64 * c) * 8;
            const half8 a8 = *reinterpret_cast<const half8*>(w + o);
            const half8 b8 = *reinterpret_cast<const half8*>(p + o);
            const half8 c8 = *reinterpret_cast<const half8*>(t + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] = (float)a8[e] + (float)c8[e] + (float)b8[e];
        }
    row_layernorm(x, nchunk, lane, a.d, a.eps, a.gamma, a.beta);
    store_row(a.out + (size_t)row * a.d, x, nchunk, lane);
}

__global__ void __launch_bounds__(256) bh_layernorm_kernel(BhLnArgs a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int nchunk = a.d >> 3;
    const _Float16* src = a.in + (size_t)row * a.d;
    const _Float16* res = a.residual ? a.residual + (size_t)row * a.d : nullptr;
    float x[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nchunk) {
            const half8 v = *reinterpret_cast<const half8*>(src + (size_t)(lane + 64 * c) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] = (float)v[e];
            if (res) {  // LayerNorm(dense output + layer input): the residual add of BertSelfOutput / BertOutput
                const half8 r = *reinterpret_cast<const half8*>(res + (size_t)(lane + 64 * c) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c][e] += (float)r[e];
            }
        }
    row_layernorm(x, nchunk, lane, a.d, a.eps, a.gamma, a.beta);
    store_row(a.out + (size_t)row * a.d, x, nchunk, lane);
}

// The same LayerNorm(dense output + residual) in 32 vector registers for 768-wide rows (row widths of 256 NJ; launched for NJ = 3 only).  Why (round 6): a persistent GEMM workgroup (gemm_f16_p16.h: 235-237 VGPRs, allocated as 240, two waves per SIMD = 480 of the
// 512-entry file) leaves 32 registers per SIMD lane and no LDS on its CU.  The 62-register kernel above therefore cannot start on a CU a
// GEMM workgroup of the OTHER micro-batch's stream occupies: the LayerNorm passes (12 % of the forward pass's kernel time, HBM-bound) wait
// for whole GEMMs to drain and the forward pass is the SUM of the isolated kernel times.  One wave of this kernel fits beside the two GEMM
// waves of a SIMD, so a LayerNorm of one micro-batch streams through HBM while the other micro-batch's GEMM owns the matrix pipes.
// A lane owns NJ groups of 4 consecutive elements (8-byte accesses, 512 contiguous bytes per wave instruction); the wave reductions are
// DPP adds inside a row of 16 lanes + four v_readlane (no ds_bpermute: the LDS crossbar is the GEMM's).  Same arithmetic order as
// row_layernorm is NOT kept (another summation tree): the two kernels agree to fp32 round-off, tests/test_gpu_encoder.py.
namespace {
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += ln_dpp<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v += ln_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v += ln_dpp<0x141>(v);  // row_half_mirror: lane i <-> 7 - i of its 8
    v += ln_dpp<0x140>(v);  // row_mirror: lane i <-> 15 - i of its 16 -> every lane of a row holds the row's sum
    // (the builtin takes and returns an int: bit casts, not value conversions — a float argument is silently TRUNCATED to an integer, which
    // cost this kernel a factor 2.4 in mean error until tests/test_gpu_encoder.py's accuracy comparison with the general kernel caught it)
    const unsigned u = __float_as_uint(v);
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u, 0)) + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u, 16)) +
           __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u, 32)) + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u, 48));
}
}  // namespace

template <int NJ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) bh_layernorm_small_kernel(BhLnArgs a) {
    const int lane = threadIdx.x & 63;
    // (the row is wave-uniform, and the compiler is told so: every row address is then a scalar base + ONE lane-offset register)
    const long long row = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= a.n_rows) return;
    constexpr int D = 256 * NJ;
    const _Float16* src = a.in + (size_t)row * D + lane * 4;
    float x[NJ][4];
    {
        half4 v[NJ], r[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const half4*>(src + j * 256);
        if (a.residual) {
            const _Float16* res = a.residual + (size_t)row * D + lane * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) r[j] = *reinterpret_cast<const half4*>(res + j * 256);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) x[j][e] = (float)v[j][e] + (float)r[j][e];
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) x[j][e] = (float)v[j][e];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += x[j][e];
    const float mean = wave_sum_dpp(s) * (1.0f / (float)D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[j][e] -= mean;
            q = fmaf(x[j][e], x[j][e], q);
        }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * (1.0f / (float)D) + a.eps);
    _Float16* dst = a.out + (size_t)row * D + lane * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {  // one group at a time: its gamma / beta registers are dead before the next group's arrive
        const half4 g = *reinterpret_cast<const half4*>(a.gamma + j * 256 + lane * 4);
        const half4 b = *reinterpret_cast<const half4*>(a.beta + j * 256 + lane * 4);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)(x[j][e] * rstd * (float)g[e] + (float)b[e]);
        *reinterpret_cast<half4*>(dst + j * 256) = o;
    }
}

// one wave per sequence
__global__ void __launch_bounds__(256) bh_pool_kernel(BhPoolArgs a) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= a.batch) return;
    const int nchunk = a.d >> 3;
    const long long t0 = a.seq_off[s];
    const int len = a.seq_len[s];
    float x[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[c][e] = 0.f;
    const int rows = a.mode == 0 ? 1 : len;  // 0 = CLS (first token), 1 = mean over the sequence's tokens
    for (int r = 0; r < rows; ++r) {
        const _Float16* src = a.x + (size_t)(t0 + r) * a.d;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (lane + 64 * c < nchunk) {
                const half8 v = *reinterpret_cast<const half8*>(src + (size_t)(lane + 64 * c) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c][e] += (float)v[e];
            }
    }
    if (a.mode == 1) {
        const float inv = 1.0f / (float)len;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] *= inv;
    }
    if (a.l2_normalize) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (lane + 64 * c < nchunk)
#pragma unroll
                for (int e = 0; e < 8; ++e) q += x[c][e] * x[c][e];
        const float n = sqrtf(wave_sum(q));
        const float inv = n > 0.f ? 1.0f / n : 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] *= inv;
    }
    store_row(a.out + (size_t)s * a.d, x, nchunk, lane);
}

// one wave per (sequence, position) of the padded output
__global__ void __launch_bounds__(256) bh_unpack_kernel(BhUnpackArgs a) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long long)a.batch * a.seq_len_padded) return;
    const int s = (int)(r / a.seq_len_padded), t = (int)(r % a.seq_len_padded);
    const int nchunk = a.d >> 3;
    const int src_row = a.slot[r];  // packed row of this (sequence, position), -1 for padding
    _Float16* dst = a.out + (size_t)r * a.d;
    (void)s;
    (void)t;
    for (int c = lane; c < nchunk; c += 64) {
        half8 v;
        if (src_row >= 0)
            v = *reinterpret_cast<const half8*>(a.x + (size_t)src_row * a.d + (size_t)c * 8);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
        *reinterpret_cast<half8*>(dst + (size_t)c * 8) = v;
    }
}

// SPLADE pooling, second half: emb[b][v] = log(1 + max_t relu(logit[b][t][v]))  (reference models/retrievers/splade.py:42-43;
// the max over tokens was taken by the decoder GEMM's epilogue)
__global__ void __launch_bounds__(256) bh_splade_finish_kernel(BhSpladeFinishArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)a.batch * a.vocab;
    if (i >= n) return;
    const int b = (int)(i / a.vocab), v = (int)(i % a.vocab);
    const float m = __uint_as_float(a.seg[(size_t)b * a.ld_seg + v]);
    a.out[i] = (_Float16)log1pf(m);
}

// Sequence-classification head of HF BertForSequenceClassification (the reference's cross-encoder reranker,
// models/rerankers/crossencoder.py:18,34-38 -> .logits): pooled = tanh(Wp h_cls + bp) (BertPooler), logits = Wc pooled + bc.
// Two launches, fp32 throughout.  (1) pooler: a wave per (output row j, sequence) pair group -- grid (d / 8, batch), four
// waves of two rows each, lanes striding over d with 16-byte loads (the [CLS] row and the weight rows are L2 hits after
// the first workgroup; one workgroup per SEQUENCE walking all d rows serially took 0.33 ms for 32 pairs at d = 1024,
// 4 % of a BERT-large forward).  (2) classifier: one workgroup per sequence, a wave per label.  The per-lane summation
// order is the one the single-kernel version had.
__global__ void __launch_bounds__(256) bh_cls_pool_kernel(BhClsHeadArgs a) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, nchunk = d >> 3;
    const _Float16* h = a.x + (size_t)a.seq_off[b] * d;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = blockIdx.x * 8 + wave * 2 + r;
        if (j >= d) break;
        const _Float16* w = a.wp + (size_t)j * d;
        float s = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            const half8 v = *reinterpret_cast<const half8*>(w + (size_t)c * 8);
            const half8 x = *reinterpret_cast<const half8*>(h + (size_t)c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)x[e];
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float z = s + (float)a.bp[j];
            a.pooled[(size_t)b * d + j] = a.activation == 1 ? 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)) : tanhf(z);
        }
    }
}

__global__ void __launch_bounds__(256) bh_cls_head_kernel(BhClsHeadArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, nchunk = d >> 3;
    const float* ps = a.pooled + (size_t)b * d;
    for (int l = wave; l < a.n_labels; l += 4) {
        const _Float16* w = a.wc + (size_t)l * d;
        float s = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            const half8 v = *reinterpret_cast<const half8*>(w + (size_t)c * 8);
            const float4 p0 = *reinterpret_cast<const float4*>(ps + (size_t)c * 8);
            const float4 p1 = *reinterpret_cast<const float4*>(ps + (size_t)c * 8 + 4);
            s += (float)v[0] * p0.x;
            s += (float)v[1] * p0.y;
            s += (float)v[2] * p0.z;
            s += (float)v[3] * p0.w;
            s += (float)v[4] * p1.x;
            s += (float)v[5] * p1.y;
            s += (float)v[6] * p1.z;
            s += (float)v[7] * p1.w;
        }
        s = wave_sum(s);
        if (lane == 0) a.out[(size_t)b * a.n_labels + l] = s + (float)a.bc[l];
    }
}

hipError_t bh_launch_cls_head(const BhClsHeadArgs& a, hipStream_t st) {
    if (a.batch <= 0) return hipSuccess;
    if (a.d > 2048 || (a.d & 7) || !a.pooled) return hipErrorInvalidValue;
    for (int b0 = 0; b0 < a.batch; b0 += 65535) {  // (grid.y limit)
        BhClsHeadArgs t = a;
        t.seq_off = a.seq_off + b0;
        t.pooled = a.pooled + (size_t)b0 * a.d;
        t.out = a.out + (size_t)b0 * a.n_labels;
        t.batch = a.batch - b0 < 65535 ? a.batch - b0 : 65535;
        hipLaunchKernelGGL(bh_cls_pool_kernel, dim3((unsigned)((a.d + 7) / 8), (unsigned)t.batch), dim3(256), 0, st, t);
        hipLaunchKernelGGL(bh_cls_head_kernel, dim3((unsigned)t.batch), dim3(256), 0, st, t);
    }
    return hipGetLastError();
}

hipError_t bh_launch_splade_finish(const BhSpladeFinishArgs& a, hipStream_t st) {
    const long long n = (long long)a.batch * a.vocab;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_splade_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t bh_launch_embed_ln(const BhEmbedArgs& a, hipStream_t st) {
    if (a.n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_embed_ln_kernel, dim3((unsigned)((a.n_rows + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}
// one wave per weight row: k is a multiple of 64 (bh_encoder_create)
__global__ void __launch_bounds__(256) bh_ln_fold_kernel(BhLnFoldArgs a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.n) return;
    const _Float16* w = a.w + (size_t)n * a.k;
    _Float16* wo = a.w_out + (size_t)n * a.k;
    float csum = 0.f, bsum = 0.f;
    for (int k = lane; k < a.k; k += 64) {
        const float wv = (float)w[k];
        const _Float16 f = (_Float16)(wv * (float)a.gamma[k]);
        wo[k] = f;
        csum += (float)f;
        bsum = fmaf(wv, (float)a.beta[k], bsum);
    }
    csum = wave_sum(csum);
    bsum = wave_sum(bsum);
    if (lane == 0) {
        a.c_out[n] = (_Float16)csum;
        a.bias_out[n] = (_Float16)((a.bias ? (float)a.bias[n] : 0.f) + bsum);
    }
}

__global__ void __launch_bounds__(256) bh_ln_finalize_kernel(BhLnFinalizeArgs a) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n_rows) return;
    const float2* p = reinterpret_cast<const float2*>(a.partial) + (size_t)row * a.n_part;
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < a.n_part; ++j) {
        const float2 v = p[j];
        s1 += v.x;
        s2 += v.y;
    }
    const float inv = 1.0f / (float)a.d;
    const float mean = s1 * inv;
    const float var = fmaxf(s2 * inv - mean * mean, 0.f);
    reinterpret_cast<float2*>(a.stats)[row] = make_float2(mean, 1.0f / sqrtf(var + a.eps));
}

hipError_t bh_launch_ln_fold(const BhLnFoldArgs& a, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_ln_fold_kernel, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t bh_launch_ln_finalize(const BhLnFinalizeArgs& a, hipStream_t st) {
    if (a.n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_ln_finalize_kernel, dim3((unsigned)((a.n_rows + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

static int g_ln_small = 1;
void bh_ln_set_small(int on) { g_ln_small = on ? 1 : 0; }

hipError_t bh_launch_layernorm(const BhLnArgs& a_in, hipStream_t st) {
    if (a_in.n_rows <= 0) return hipSuccess;
    BhLnArgs a = a_in;
    if (a.small_regs < 0) a.small_regs = g_ln_small;
    // the 32-register kernel (fits beside a persistent GEMM workgroup's waves) where the row width allows it and the caller asks for it
    // (768-wide rows only: the 1024-wide instantiation needs 40 registers, does not fit beside the GEMM and measured no gain — the large shapes
    // keep the general kernel and their round-5 bits)
    if (a.small_regs && a.d == 768)
        hipLaunchKernelGGL(bh_layernorm_small_kernel<3>, dim3((unsigned)((a.n_rows + 3) / 4)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(bh_layernorm_kernel, dim3((unsigned)((a.n_rows + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}
// One thread = 8 consecutive dims j .. j + 7 (j < 32) of one head slice and their partners j + 32 ..: two 16-byte loads, the 8
// angles' cosines and sines from the fp32 table (positions are token indices inside a sequence: the table has max_position
// rows, built on the host in double precision), fp32 rotation, two 16-byte stores.  x1' = x1 cos - x2 sin, x2' = x2 cos + x1 sin.
__global__ void __launch_bounds__(256) bh_rotary_kernel(BhRotaryArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per_row = a.n_heads * 8;  // 2 * n_heads slices x 4 chunks of 8 pairs
    const long long row = t / per_row;
    if (row >= a.n_rows) return;
    const int rem = (int)(t - row * per_row);
    const int slice = rem >> 2, c = rem & 3;
    int p = a.pos[row];
    p = p < 0 ? 0 : (p >= a.max_pos ? a.max_pos - 1 : p);
    _Float16* x = a.qk + (size_t)row * (size_t)(a.n_heads * 128) + (size_t)slice * 64 + c * 8;
    const float* cs = a.cos_sin + (size_t)p * 64 + c * 8;
    half8 x1 = *reinterpret_cast<const half8*>(x);
    half8 x2 = *reinterpret_cast<const half8*>(x + 32);
    const float4 c0 = *reinterpret_cast<const float4*>(cs), c1 = *reinterpret_cast<const float4*>(cs + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(cs + 32), s1 = *reinterpret_cast<const float4*>(cs + 36);
    const float co[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float si[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    half8 o1, o2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a1 = (float)x1[e], a2 = (float)x2[e];
        // (explicit fma: the GEMM epilogue that rotates the rows of whole tiles — gemm_f16_p16.h, BH_EPI_ROTARY — writes the same expression)
        // (and the fp32 results are rounded to fp16 by a conversion of their own — an opaque register keeps hipcc from fusing fma and
        // conversion into v_fma_mix*_f16, whose single rounding differs from fma-then-convert in rare cases: both kernels do the same)
        float r1 = __builtin_fmaf(-a2, si[e], a1 * co[e]);
        float r2 = __builtin_fmaf(a1, si[e], a2 * co[e]);
        asm volatile("" : "+v"(r1), "+v"(r2));
        o1[e] = (_Float16)r1;
        o2[e] = (_Float16)r2;
    }
    *reinterpret_cast<half8*>(x) = o1;
    *reinterpret_cast<half8*>(x + 32) = o2;
}

// One thread = 8 output columns = 16 consecutive inputs (gate, up, gate, up, ...): out = g / (1 + exp(-g)) * u  (v_exp_f32 of
// -g log2 e; |g| large: exp -> inf or 0, the quotient -> 0 or g, both finite).  The standalone form of the fold the persistent
// GEMM does in its epilogue (BH_EPI_SWIGLU): used where that kernel does not apply (small or ragged problems).
__global__ void __launch_bounds__(256) bh_swiglu_kernel(BhSwigluArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per_row = a.f >> 3;
    const long long row = t / per_row;
    if (row >= a.n_rows) return;
    const int c = (int)(t - row * per_row);
    const _Float16* g = a.gu + (size_t)row * (size_t)(2 * a.f) + (size_t)c * 16;
    const half8 lo = *reinterpret_cast<const half8*>(g);
    const half8 hi = *reinterpret_cast<const half8*>(g + 8);
    half8 o;
    if (a.act == 1) {  // erf-GELU gate (gte-*-en-v1.5): the GEMM epilogue's polynomial, expression for expression (same bits)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (_Float16)(bh_gemm::gelu_erf((float)lo[2 * e]) * (float)lo[2 * e + 1]);
            o[4 + e] = (_Float16)(bh_gemm::gelu_erf((float)hi[2 * e]) * (float)hi[2 * e + 1]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = (float)lo[2 * e], g1 = (float)hi[2 * e];
            o[e] = (_Float16)(g0 / (1.0f + __builtin_amdgcn_exp2f(-g0 * 1.4426950408889634f)) * (float)lo[2 * e + 1]);
            o[4 + e] = (_Float16)(g1 / (1.0f + __builtin_amdgcn_exp2f(-g1 * 1.4426950408889634f)) * (float)hi[2 * e + 1]);
        }
    }
    *reinterpret_cast<half8*>(a.out + (size_t)row * (size_t)a.f + (size_t)c * 8) = o;
}

hipError_t bh_launch_rotary(const BhRotaryArgs& a, hipStream_t st) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.n_heads <= 0 || a.max_pos <= 0 || !a.qk || !a.pos || !a.cos_sin) return hipErrorInvalidValue;
    const long long n = a.n_rows * (long long)a.n_heads * 8;
    hipLaunchKernelGGL(bh_rotary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t bh_launch_swiglu(const BhSwigluArgs& a, hipStream_t st) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.f <= 0 || (a.f & 7) || !a.gu || !a.out) return hipErrorInvalidValue;
    const long long n = a.n_rows * (long long)(a.f >> 3);
    hipLaunchKernelGGL(bh_swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t bh_launch_pool(const BhPoolArgs& a, hipStream_t st) {
    if (a.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_pool_kernel, dim3((unsigned)((a.batch + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t bh_launch_unpack(const BhUnpackArgs& a, hipStream_t st) {
    const long long n = (long long)a.batch * a.seq_len_padded;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_unpack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}
