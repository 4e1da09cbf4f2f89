// attention.hip — fused variable-length self-attention of the bi-encoder (softmax(Q K^T / sqrt(64)) V).
//
// Replaces BertSelfAttention's matmul -> +mask -> softmax -> matmul (transformers modeling_bert.py, reached
// from the reference through AutoModel, models/retrievers/dense.py:16,40-44).  Tokens are PACKED: sequence s
// owns rows [off[s], off[s]+len[s]) of the activation matrices, so the reference's additive padding mask
// becomes "keys >= len do not exist"; results for real tokens are identical (masked keys carry zero
// probability in the reference as well).
//
// One 8-wave workgroup per (sequence, head); wave w takes the 32-query blocks w, w+8, ...  The workgroup
// first stages the sequence's whole K slice (len x 64) and V^T slice (64 x len) of this head into LDS with
// coalesced loads — K in the XOR-permuted 128-byte-row image of the GEMM (conflict-free ds_read_b128 fragment
// reads), V^T in 64-byte rows with an 8-byte-chunk XOR — one barrier, then every wave runs its key loop out of LDS with
// no further global loads or barriers.  The XOR term of V^T row dd is (dd >> 1) & 7: hipcc pairs the reads of head dims
// ql and ql + 32 into ds_read2st64_b64, which the LDS serves in groups of 16 CONTIGUOUS lanes against 32 banks (a 128-byte
// window: two rows); 16 lanes = 2 row parities x 8 distinct terms.  (Until round 3 the term was (dd >> 2) & 7 — right
// for a plain ds_read_b64's 32-lane groups and 64 banks, two-way conflicts on the paired read and on the ds_write_b64 of
// the staging: SQ_LDS_BANK_CONFLICT was 38 % of SQ_LDS_IDX_ACTIVE.)
//   S^T = K . Q^T          v_mfma_f32_32x32x16_f16, A = K rows (i = key), B = Q rows (j = query):
//                          lane l owns query l&31 and 16 of the 32 key scores -> the softmax reductions are
//                          in-lane plus ONE exchange between the two half-lanes of a query.
//   online softmax         fp32, v_exp_f32, scale 1/8 folded into the exponent.
//   O^T += V^T . P         A = V^T rows (i = head dim) from the TRANSPOSED value matrix (written that way by the V
//                          GEMM, blocked by 64 tokens: [tokens/64][d][64], so that a head's slice of a token
//                          block is 8 KiB contiguous), two 8-byte LDS reads per fragment, B = P (the lane's own
//                          probabilities, already in operand layout).
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

// NWV = waves per workgroup: 8 in general, 4 for sequences of at most 128 tokens (half the LDS, twice the
// workgroups per CU, no idle waves behind the staging barrier).
// WIDE: the context rows leave as 16-byte stores (two neighbouring 4-column groups joined by one half-lane exchange: a store
// instruction then covers 32 bytes of each of its 32 rows instead of 16 — the same lesson as the GEMM's output path: the
// memory system pays per row piece, not per byte); needs the documented direction of v_permlane32_swap (probed by the GEMM).
// ALIBI: + the symmetric ALiBi bias -slope[head] |query - key| on every score (JinaBert: jina-embeddings-v2), applied with the
// padding mask in units of the raw dot product (the 1/8 scale is folded into the exponent: the bias is multiplied by 8).
template <int NWV, bool WIDE, bool ALIBI = false>
__global__ void __launch_bounds__(64 * NWV, 4) bh_attention_kernel(BhAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 64 * NWV;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = a.seq_idx ? a.seq_idx[blockIdx.y] : (int)blockIdx.y, head = blockIdx.x;
    const int len = a.seq_len[s];
    const long long t0 = a.seq_off[s];
    const int nkb = (len + 31) >> 5;
    unsigned char* smK = smem;                // nkb pieces of 32 keys x 128 B (XOR-permuted, see gemm_f16_kernel.h)
    unsigned char* smV = smem + a.v_lds_off;  // nkb tiles of 64 dims x 32 keys (64-byte rows, 8-byte chunk XOR)

    // Q fragments (operand B) of the wave's first query block: lane (query ql, half h), k-step s4 covers head dims
    // 16 s4 + 8 h .. + 8.  Requested before the staging so that their latency overlaps it.
    const int ql = lane & 31, h = lane >> 5;
    auto load_q = [&](half8 (&qf)[4], int q0) {
        int qr = q0 + ql;
        qr = qr < len ? qr : len - 1;
        const _Float16* qp = a.qk + (size_t)(t0 + qr) * a.ldqk + head * 64 + 8 * h;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *reinterpret_cast<const half8*>(qp + 16 * s4);
    };
    half8 qf[4];
    load_q(qf, wave * 32 < len ? wave * 32 : 0);

    // ---- stage the sequence's K rows and V^T rows of this head into LDS, once, coalesced; all loads of a
    // batch are issued before the first LDS store so that their latencies overlap
    {
        const _Float16* kg = a.qk + (size_t)t0 * a.ldqk + a.d_model + head * 64;
        const _Float16* vg = a.vt + (size_t)(head * 64) * a.ldvt + t0;  // row-major V^T
        const _Float16* vb = a.vt + (size_t)(head * 64) * 64;           // blocked V^T: + block * d_model * 64
        constexpr int UB = 4;  // items per thread per batch: 4 K chunks + 4 V chunks in flight
        const int n_items = nkb * 256;
        for (int base = 0; base < n_items; base += NT * UB) {
            half8 kv[UB], vv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * NT + tid;
                if (idx < n_items) {
                    const int row = idx >> 3, c = idx & 7;  // key row, 16-byte chunk of its 128-byte head slice
                    const int kr = row < len ? row : len - 1;
                    kv[u] = *reinterpret_cast<const half8*>(kg + (size_t)kr * a.ldqk + c * 8);
                    const int kbi = idx >> 8, dd = (idx >> 2) & 63, c16 = idx & 3;  // key block, head dim, 8-key chunk
                    if (a.vt_blocked) {  // an 8-token chunk never straddles a 64-token block (t0 % 8 == 0)
                        const long long tok = t0 + kbi * 32 + c16 * 8;
                        vv[u] = *reinterpret_cast<const half8*>(vb + (size_t)(tok >> 6) * a.d_model * 64 + dd * 64 + (tok & 63));
                    } else {
                        vv[u] = *reinterpret_cast<const half8*>(vg + (size_t)dd * a.ldvt + kbi * 32 + c16 * 8);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * NT + tid;
                if (idx < n_items) {
                    const int row = idx >> 3, c = idx & 7;
                    const int r = row & 31;
                    const int g = ((r >> 1) & 1) | ((r >> 3) << 1);
                    *reinterpret_cast<half8*>(smK + (row >> 5) * 4096 + (r >> 3) * 1024 + (r & 7) * 128 + ((c ^ g) << 4)) =
                        kv[u];
                    const int kbi = idx >> 8, dd = (idx >> 2) & 63, c16 = idx & 3;
                    half4 lo, hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = vv[u][e];
                        hi[e] = vv[u][4 + e];
                    }
                    const int g8 = (dd >> 1) & 7;
                    unsigned char* vb = smV + kbi * 4096 + dd * 64;
                    *reinterpret_cast<half4*>(vb + (((2 * c16) ^ g8) << 3)) = lo;
                    *reinterpret_cast<half4*>(vb + (((2 * c16 + 1) ^ g8) << 3)) = hi;
                }
            }
        }
    }
    __syncthreads();
    const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)

    // LDS fragment offsets: K as the conflict-free ds_read_b128 pattern, V^T conflict-free for ds_read2st64_b64 (see the header)
    unsigned k_off[4];
    {
        const int g = ((ql >> 1) & 1) | ((ql >> 3) << 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            k_off[s4] = (unsigned)((ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * s4 + h) ^ g) << 4));
    }
    unsigned v_off[2][2];  // [s2][lo|hi] for head dim ql; head dim 32 + ql sits 2048 bytes further (same XOR term)
    {
        const int g8 = (ql >> 1) & 7;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int c8 = 4 * s2 + h;  // keys 16 s2 + 4 h + 0..3; the second half of the fragment sits 8 keys later
            v_off[s2][0] = (unsigned)(ql * 64 + ((c8 ^ g8) << 3));
            v_off[s2][1] = (unsigned)(ql * 64 + (((c8 + 2) ^ g8) << 3));
        }
    }

    // wave w takes query blocks w, w + NWV, ... of the sequence
    for (int qb = wave; qb * 32 < len; qb += NWV) {
    const int q0 = qb * 32;
    if (qb != wave) load_q(qf, q0);
    float m_run = -__builtin_inff();  // running max (identical in both half-lanes)
    float l_run = 0.f;                // this half-lane's share of the running denominator
    floatx16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[dt][v] = 0.f;

    for (int kb = 0; kb < nkb; ++kb) {
        const unsigned char* kt = smK + kb * 4096;
        const unsigned char* vt = smV + kb * 4096;
        half8 kf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) kf[s4] = *reinterpret_cast<const half8*>(kt + k_off[s4]);
        // V^T fragments: key slot (h, e) of step s2 is key kb*32 + 16 s2 + 8 (e>>2) + 4 h + (e&3) — the order the
        // probabilities sit in the lane's registers
        half4 vlo[2][2], vhi[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                vlo[dt][s2] = *reinterpret_cast<const half4*>(vt + dt * 2048 + v_off[s2][0]);
                vhi[dt][s2] = *reinterpret_cast<const half4*>(vt + dt * 2048 + v_off[s2][1]);
            }

        floatx16 sc;
#pragma unroll
        for (int v = 0; v < 16; ++v) sc[v] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s4], qf[s4], sc, 0, 0, 0);

        // mask keys beyond the sequence, block max
        const int key0 = kb * 32 + 4 * h;
        float bmax = -__builtin_inff();
        float slope8 = 0.f;
        if constexpr (ALIBI) slope8 = 8.0f * a.alibi[head];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int key = key0 + (v & 3) + 8 * (v >> 2);
            if constexpr (ALIBI) sc[v] = fmaf(-slope8, fabsf((float)(q0 + ql - key)), sc[v]);
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
    if constexpr (WIDE) {
        // lane (ql, h) holds columns 8 gq + 4 h + 0..3 of every 8-column group gq of a 32-column half dt.  Groups 2 gp and 2 gp + 1:
        // the lower half-lane hands its part of group 2 gp + 1 to the upper one and takes the upper one's part of group 2 gp ->
        // lane h owns the 8 consecutive columns 16 gp + 8 h + 0..7.  (Executed by all lanes: the exchange is outside the branch.)
        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        uint4v outv[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                half4 wa, wb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wa[e] = (_Float16)(o[dt][4 * (2 * gp) + e] * inv);
                    wb[e] = (_Float16)(o[dt][4 * (2 * gp + 1) + e] * inv);
                }
                uint2v ua = __builtin_bit_cast(uint2v, wa), ub = __builtin_bit_cast(uint2v, wb);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    auto r = __builtin_amdgcn_permlane32_swap(ua[w], ub[w], false, false);
                    ua[w] = r[0];
                    ub[w] = r[1];
                }
                outv[dt][gp] = uint4v{ua[0], ua[1], ub[0], ub[1]};
            }
        if (q0 + ql < len) {
            _Float16* op = a.ctx + (size_t)(t0 + q0 + ql) * a.ldc + head * 64 + 8 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) *reinterpret_cast<uint4v*>(op + dt * 32 + 16 * gp) = outv[dt][gp];
        }
    } else if (q0 + ql < len) {
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
    }  // query blocks
}

namespace {
template <int NWV, bool WIDE, bool ALIBI>
hipError_t launch_attn_w(const BhAttnArgs& a_in, int n_seq, int n_heads, int max_len, hipStream_t stream) {
    if (n_seq <= 0) return hipSuccess;
    const int nkb = (max_len + 31) / 32;
    if (nkb * 8192 > 160 * 1024) return hipErrorInvalidValue;  // sequences longer than 640 tokens
    BhAttnArgs a = a_in;
    a.v_lds_off = nkb * 4096;
    const size_t smem = (size_t)nkb * 8192;
    static size_t attr_smem = 0;
    if (smem > attr_smem) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bh_attention_kernel<NWV, WIDE, ALIBI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_smem = smem;
    }
    hipLaunchKernelGGL((bh_attention_kernel<NWV, WIDE, ALIBI>), dim3(n_heads, n_seq), dim3(64 * NWV), smem, stream, a);
    return hipGetLastError();
}
template <int NWV>
hipError_t launch_attn(const BhAttnArgs& a, int n_seq, int n_heads, int max_len, hipStream_t stream) {
    // 16-byte context stores where v_permlane32_swap has its documented direction (the GEMM's probe), 8-byte stores otherwise
    const bool wide = n_seq > 0 && bh_gemm_probe_permlane(stream) == hipSuccess && bh_gemm_swap_mode() == 0;
    if (a.alibi) return wide ? launch_attn_w<NWV, true, true>(a, n_seq, n_heads, max_len, stream) : launch_attn_w<NWV, false, true>(a, n_seq, n_heads, max_len, stream);
    return wide ? launch_attn_w<NWV, true, false>(a, n_seq, n_heads, max_len, stream) : launch_attn_w<NWV, false, false>(a, n_seq, n_heads, max_len, stream);
}
}  // namespace

// All sequences in one launch (8-wave workgroups, LDS sized for max_len).
hipError_t bh_launch_attention(const BhAttnArgs& a, int batch, int n_heads, int max_len, hipStream_t stream) {
    if (batch <= 0 || max_len <= 0) return hipSuccess;
    if (max_len <= 128) return launch_attn<4>(a, batch, n_heads, max_len, stream);
    return launch_attn<8>(a, batch, n_heads, max_len, stream);
}

// Two launches over a length-bucketed batch: seq_idx_dev holds the n_short sequences of at most 128 tokens first,
// then the n_long longer ones.
// stream_long (optional): the launch over the long sequences goes there instead — the two launches touch disjoint sequences,
// the caller orders both streams against the producers and consumers of Q | K, V^T and the context rows.
hipError_t bh_launch_attention_bucketed(const BhAttnArgs& a_in, const int* seq_idx_dev, int n_short, int n_long,
                                        int max_len_long, int n_heads, hipStream_t stream, int short_max, hipStream_t stream_long) {
    BhAttnArgs a = a_in;
    a.seq_idx = seq_idx_dev;
    hipError_t e = launch_attn<4>(a, n_short, n_heads, short_max, stream);
    if (e != hipSuccess) return e;
    a.seq_idx = seq_idx_dev + n_short;
    return launch_attn<8>(a, n_long, n_heads, max_len_long > short_max + 1 ? max_len_long : short_max + 1, stream_long ? stream_long : stream);
}
