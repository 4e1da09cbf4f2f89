// attention.hip — fused variable-length self-attention of the bi-encoder (softmax(Q K^T / sqrt(64)) V).
//
// Replaces BertSelfAttention's matmul -> +mask -> softmax -> matmul (transformers modeling_bert.py, reached
// from the reference through AutoModel, models/retrievers/dense.py:16,40-44).  Tokens are PACKED: sequence s
// owns rows [off[s], off[s]+len[s]) of the activation matrices, so the reference's additive padding mask
// becomes "keys >= len do not exist"; results for real tokens are identical (masked keys carry zero
// probability in the reference as well).
//
// One wave per (sequence, head, 32-query block); four such waves (4 query blocks) per workgroup share the
// sequence's K / V^T lines through the vector L1.  No LDS, no barriers.
//   S^T = K . Q^T          v_mfma_f32_32x32x16_f16, A = K rows (i = key), B = Q rows (j = query):
//                          lane l owns query l&31 and 16 of the 32 key scores -> the softmax reductions are
//                          in-lane plus ONE exchange between the two half-lanes of a query.
//   online softmax         fp32, v_exp_f32, scale 1/8 folded into the exponent.
//   O^T += V^T . P         A = V^T rows (i = head dim) read from the TRANSPOSED value matrix VT[d][tokens]
//                          (written that way by the QKV GEMM) as two 8-byte loads per fragment, B = P (the
//                          lane's own probabilities, already in operand layout).
// Head dim is 64 (bert-base 768/12, bert-large 1024/16, e5 / contriever / bge / RetroMAE alike).
// Roofline: MFMA (4 T^2 64 flop per head) — ~3 % of the encoder's flops at T = 128.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {
__device__ __forceinline__ float half_lanes_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));  // {own, partner} in either swap direction
}
__device__ __forceinline__ float half_lanes_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
}  // namespace

__global__ void __launch_bounds__(256) bh_attention_kernel(BhAttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.z, head = blockIdx.y;
    const int qb = blockIdx.x * 4 + wave;
    const int len = a.seq_len[s];
    const long long t0 = a.seq_off[s];
    if (qb * 32 >= len) return;
    const int ql = lane & 31, h = lane >> 5;
    const int q0 = qb * 32;

    // Q fragments (operand B): lane (query ql, half h), k-step s4 covers head dims 16 s4 + 8 h .. + 8
    half8 qf[4];
    {
        int qr = q0 + ql;
        qr = qr < len ? qr : len - 1;
        const _Float16* qp = a.qk + (size_t)(t0 + qr) * a.ldqk + head * 64 + 8 * h;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *reinterpret_cast<const half8*>(qp + 16 * s4);
    }
    const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    float m_run = -__builtin_inff();               // running max (identical in both half-lanes)
    float l_run = 0.f;                             // this half-lane's share of the running denominator
    floatx16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[dt][v] = 0.f;

    const _Float16* kbase = a.qk + (size_t)t0 * a.ldqk + a.d_model + head * 64 + 8 * h;
    const _Float16* vbase = a.vt + (size_t)(head * 64 + ql) * a.ldvt + t0 + 4 * h;
    const int nkb = (len + 31) >> 5;
    for (int kb = 0; kb < nkb; ++kb) {
        // K fragments (operand A): lane (key ql, half h)
        int kr = kb * 32 + ql;
        kr = kr < len ? kr : len - 1;
        const _Float16* kp = kbase + (size_t)kr * a.ldqk;
        half8 kf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) kf[s4] = *reinterpret_cast<const half8*>(kp + 16 * s4);
        // V^T fragments (operand A of the second product): lane (dim ql [+32], half h); key slot (h, e) of
        // step s2 is key kb*32 + 16 s2 + 8 (e>>2) + 4 h + (e&3)  — the order the probabilities sit in
        half4 vlo[2][2], vhi[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const _Float16* vp = vbase + (size_t)dt * 32 * a.ldvt + kb * 32 + 16 * s2;
                vlo[dt][s2] = *reinterpret_cast<const half4*>(vp);
                vhi[dt][s2] = *reinterpret_cast<const half4*>(vp + 8);
            }

        floatx16 sc;
#pragma unroll
        for (int v = 0; v < 16; ++v) sc[v] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s4], qf[s4], sc, 0, 0, 0);

        // mask keys beyond the sequence, block max
        const int key0 = kb * 32 + 4 * h;
        float bmax = -__builtin_inff();
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int key = key0 + (v & 3) + 8 * (v >> 2);
            sc[v] = key < len ? sc[v] : -__builtin_inff();
            bmax = fmaxf(bmax, sc[v]);
        }
        bmax = half_lanes_max(bmax);  // finite: key kb*32 < len always exists
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);  // first block: exp2(-inf) = 0
        const float mc = m_new * c;
        float psum = 0.f;
        half8 pf[2];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sc[v], c, -mc));
            psum += p;
            pf[v >> 3][v & 7] = (_Float16)p;
        }
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vf[e] = vlo[dt][s2][e];
                    vf[4 + e] = vhi[dt][s2][e];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], o[dt], 0, 0, 0);
            }
    }
    const float inv = 1.0f / half_lanes_sum(l_run);
    if (q0 + ql < len) {
        _Float16* op = a.ctx + (size_t)(t0 + q0 + ql) * a.ldc + head * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                half4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (_Float16)(o[dt][4 * gq + e] * inv);
                *reinterpret_cast<half4*>(op + dt * 32 + 8 * gq) = w;
            }
    }
}

hipError_t bh_launch_attention(const BhAttnArgs& a, int batch, int n_heads, int max_len, hipStream_t stream) {
    if (batch <= 0 || max_len <= 0) return hipSuccess;
    const int qblocks = (max_len + 31) / 32;
    hipLaunchKernelGGL(bh_attention_kernel, dim3((qblocks + 3) / 4, n_heads, batch), dim3(256), 0, stream, a);
    return hipGetLastError();
}
