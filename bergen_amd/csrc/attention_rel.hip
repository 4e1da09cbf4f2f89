// attention_rel.hip — fused variable-length DISENTANGLED self-attention (DeBERTa-v2 / v3):
//
//     score[i][j] = ( Q_i . K_j  +  Q_i . Kr[t(i - j)]  +  K_j . Qr[t(i - j)] ) / sqrt(3 * 64)        softmax over j, times V
//
// Replaces DisentangledSelfAttention.forward + disentangled_attention_bias (transformers modeling_deberta_v2.py:191-346),
// which the reference reaches through AutoModelForSequenceClassification for its default reranker
// (models/rerankers/crossencoder.py:18, config/reranker/debertav3.yaml:3).  Kr / Qr are the key / query projections of the
// (layer-normed) relative-position embedding table (share_att_key), t(delta) = clamp(bucket(delta) + span, 0, 2 span - 1)
// with the log-bucket function of make_log_bucket_position; the bucket function is odd, which is why ONE index t(i - j)
// serves the content->position term (row i of c2p) and the position->content term (row j of p2c): HF's
// clamp(-bucket(j - i) + span) is the same number.
//
// The two position terms arrive as matrices c2p[head][token][p] = Q_i . Kr[p] and p2c[head][token][p] = K_j . Qr[p]
// (2 span columns), produced per layer and head by the MFMA GEMM (encoder.hip); this kernel is bh_attention_kernel
// (attention.hip: K and V^T of the (sequence, head) staged in LDS once, S^T = K Q^T on the matrix cores, a lane owns one
// query and 16 of a block's 32 key scores, online softmax in fp32, O^T += V^T P) with the two gathered terms added to the
// scores before the softmax.  Round 3 gathered them straight from L2 (32 two-byte loads per lane and 32 x 32 block, every one of
// a wave's load instructions touching up to 64 cache lines: 127 us per launch at DeBERTa-v3-large's shape, plain attention 46-77).
// Round 4 (WIN = true, sequences up to 320 tokens): t(delta) is monotone, so a (query block, key block) pair only needs the
// columns [t(delta_min), t(delta_max)] — at most 63 — of 32 rows of c2p and 32 rows of p2c: two 32 x 72 windows (the range
// aligned down to 8 columns) are fetched with coalesced 16-byte loads issued BEFORE the block's QK^T MFMAs, parked in the wave's
// own 9 KiB of LDS behind them, and the per-score lookups become two-byte LDS reads.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {
__device__ __forceinline__ float rel_half_lanes_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rel_half_lanes_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
}  // namespace

template <bool WIN>
__global__ void __launch_bounds__(512, 2) bh_attention_rel_kernel(BhAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWV = 8, NT = 512;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = (int)blockIdx.y, head = blockIdx.x;
    const int len = a.seq_len[s];
    const long long t0 = a.seq_off[s];
    const int nkb = (len + 31) >> 5;
    unsigned char* smK = smem;
    unsigned char* smV = smem + a.v_lds_off;
    int* ridx = reinterpret_cast<int*>(smem + a.rel_lds_off);  // t(delta) for |delta| < len, at index delta + len - 1
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
    for (int i = tid; i < 2 * len - 1; i += NT) ridx[i] = a.rel_idx[i - (len - 1) + a.rel_center];

    // ---- stage K rows and V^T rows of this (sequence, head) into LDS (same images as attention.hip)
    {
        const _Float16* kg = a.qk + (size_t)t0 * a.ldqk + a.d_model + head * 64;
        const _Float16* vg = a.vt + (size_t)(head * 64) * a.ldvt + t0;
        const _Float16* vb = a.vt + (size_t)(head * 64) * 64;
        const int n_items = nkb * 256;
        for (int idx = tid; idx < n_items; idx += NT) {
            const int row = idx >> 3, c = idx & 7;
            const int kr = row < len ? row : len - 1;
            const half8 kv = *reinterpret_cast<const half8*>(kg + (size_t)kr * a.ldqk + c * 8);
            const int kbi = idx >> 8, dd = (idx >> 2) & 63, c16 = idx & 3;
            half8 vv;
            if (a.vt_blocked) {
                const long long tok = t0 + kbi * 32 + c16 * 8;
                vv = *reinterpret_cast<const half8*>(vb + (size_t)(tok >> 6) * a.d_model * 64 + dd * 64 + (tok & 63));
            } else {
                vv = *reinterpret_cast<const half8*>(vg + (size_t)dd * a.ldvt + kbi * 32 + c16 * 8);
            }
            const int r = row & 31;
            const int g = ((r >> 1) & 1) | ((r >> 3) << 1);
            *reinterpret_cast<half8*>(smK + (row >> 5) * 4096 + (r >> 3) * 1024 + (r & 7) * 128 + ((c ^ g) << 4)) = kv;
            half4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = vv[e];
                hi[e] = vv[4 + e];
            }
            const int g8 = (dd >> 1) & 7;
            unsigned char* vbl = smV + kbi * 4096 + dd * 64;
            *reinterpret_cast<half4*>(vbl + (((2 * c16) ^ g8) << 3)) = lo;
            *reinterpret_cast<half4*>(vbl + (((2 * c16 + 1) ^ g8) << 3)) = hi;
        }
    }
    __syncthreads();
    const float c = a.rel_scale * 1.4426950408889634f;  // scale * log2(e)

    unsigned k_off[4];
    {
        const int g = ((ql >> 1) & 1) | ((ql >> 3) << 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) k_off[s4] = (unsigned)((ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * s4 + h) ^ g) << 4));
    }
    unsigned v_off[2][2];
    {
        const int g8 = (ql >> 1) & 7;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int c8 = 4 * s2 + h;
            v_off[s2][0] = (unsigned)(ql * 64 + ((c8 ^ g8) << 3));
            v_off[s2][1] = (unsigned)(ql * 64 + (((c8 + 2) ^ g8) << 3));
        }
    }
    const _Float16* c2p_h = a.c2p + (size_t)head * a.rel_head_stride;
    const _Float16* p2c_h = a.p2c + (size_t)head * a.rel_head_stride;
    // position windows of this wave: [0] c2p rows of the query block, [1] p2c rows of the key block; 32 rows x 72 halfs each
    constexpr int WCOLS = 72, WCH = WCOLS / 8;           // 9 chunks of 16 bytes per row
    constexpr int WITEMS = 32 * WCH, WIT = (WITEMS + 63) / 64;  // 288 (row, chunk) items per window: 5 rounds of 64 lanes
    _Float16* win = reinterpret_cast<_Float16*>(smem + a.win_lds_off + wave * (2 * 32 * WCOLS * 2));

    for (int qb = wave; qb * 32 < len; qb += NWV) {
        const int q0 = qb * 32;
        if (qb != wave) load_q(qf, q0);
        const int qrow = q0 + ql < len ? q0 + ql : len - 1;
        const _Float16* c2p_row = c2p_h + (size_t)(t0 + qrow) * a.rel_ld;
        float m_run = -__builtin_inff();
        float l_run = 0.f;
        floatx16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] = 0.f;

        for (int kb = 0; kb < nkb; ++kb) {
            const unsigned char* kt = smK + kb * 4096;
            const unsigned char* vt = smV + kb * 4096;
            // ---- position windows: columns [tlo8, tlo8 + 72) of the block pair's rows, loads issued now, parked in LDS behind
            // the QK^T MFMAs (t is monotone in delta: every t of the pair lies in [t(delta_min), t(delta_max)], <= 63 wide)
            int tlo8 = 0;
            half8 wreg[2][WIT];
            if constexpr (WIN) {
                int dmin = q0 - (kb * 32 + 31), dmax = q0 + 31 - kb * 32;
                dmin = dmin > -(len - 1) ? dmin : -(len - 1);
                dmax = dmax < len - 1 ? dmax : len - 1;
                tlo8 = __builtin_amdgcn_readfirstlane(ridx[dmin + len - 1]) & ~7;
                // (rows are rel_ld = 2 span columns, a multiple of 8 like tlo8: a window chunk lies inside the row or wholly beyond it
                // — then it is fetched from the row's last chunk instead and never asked for, t <= 2 span - 1)
                const int last = a.rel_ld - 8;
#pragma unroll
                for (int it = 0; it < WIT; ++it) {
                    const int item = it * 64 + lane;
                    const int r = item / WCH, ch = item - r * WCH;
                    int col = tlo8 + ch * 8;
                    col = col < last ? col : last;
                    if (item < WITEMS) {
                        int qr = q0 + r;
                        qr = qr < len ? qr : len - 1;
                        int kr = kb * 32 + r;
                        kr = kr < len ? kr : len - 1;
                        wreg[0][it] = *reinterpret_cast<const half8*>(c2p_h + (size_t)(t0 + qr) * a.rel_ld + col);
                        wreg[1][it] = *reinterpret_cast<const half8*>(p2c_h + (size_t)(t0 + kr) * a.rel_ld + col);
                    }
                }
                (void)dmax;
            }
            half8 kf[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) kf[s4] = *reinterpret_cast<const half8*>(kt + k_off[s4]);
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

            if constexpr (WIN) {  // windows -> this wave's LDS (wave-private: its own waits order the stores and the reads below)
#pragma unroll
                for (int it = 0; it < WIT; ++it) {
                    const int item = it * 64 + lane;
                    if (item < WITEMS) {
                        *reinterpret_cast<half8*>(win + item * 8) = wreg[0][it];
                        *reinterpret_cast<half8*>(win + 32 * WCOLS + item * 8) = wreg[1][it];
                    }
                }
            }
            // + c2p[i][t(i - j)] + p2c[j][t(i - j)], mask keys beyond the sequence, block max
            const int key0 = kb * 32 + 4 * h;
            float bmax = -__builtin_inff();
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key = key0 + (v & 3) + 8 * (v >> 2);
                const int kk = key < len ? key : len - 1;
                const int t = ridx[qrow - kk + len - 1];
                float bias;
                if constexpr (WIN) {
                    const int wc = t - tlo8;  // 0 .. 69: t lies in [t(delta_min), t(delta_min) + 62], tlo8 at most 7 below
                    bias = (float)win[ql * WCOLS + wc] + (float)win[32 * WCOLS + (kk - kb * 32) * WCOLS + wc];
                } else {
                    bias = (float)c2p_row[t] + (float)p2c_h[(size_t)(t0 + kk) * a.rel_ld + t];
                }
                sc[v] = key < len ? sc[v] + bias : -__builtin_inff();
                bmax = fmaxf(bmax, sc[v]);
            }
            bmax = rel_half_lanes_max(bmax);
            const float m_new = fmaxf(m_run, bmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
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
        const float inv = 1.0f / rel_half_lanes_sum(l_run);
        if (a.wide_stores) {
            // 16-byte context stores, as in attention.hip (option attention_rel_wide_stores; the exchange runs outside the row branch)
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
    }
}

namespace {
int g_rel_wide_stores = 1;  // bh_set_option "attention_rel_wide_stores" (default on since round 5: bit-identical to the 8-byte stores, tests/test_gpu_store_paths.py)
}
void bh_attention_rel_set_wide_stores(int on) { g_rel_wide_stores = on != 0; }

hipError_t bh_launch_attention_rel(const BhAttnArgs& a_in, int batch, int n_heads, int max_len, hipStream_t stream) {
    if (batch <= 0 || max_len <= 0) return hipSuccess;
    if (!a_in.c2p || !a_in.p2c || !a_in.rel_idx || a_in.rel_ld <= 0 || max_len - 1 > a_in.rel_center) return hipErrorInvalidValue;
    const int nkb = (max_len + 31) / 32;
    const size_t table = ((size_t)(2 * max_len) * sizeof(int) + 15) / 16 * 16;
    const size_t base = (size_t)nkb * 8192 + table;
    const size_t windows = (size_t)8 * 2 * 32 * 72 * 2;  // eight waves x two 32 x 72 fp16 windows
    // the LDS windows of the position terms where they fit beside the sequence's K / V^T (up to 320 tokens), rows of at least
    // one window chunk and 16-byte aligned; else the round-3 gathers from L2
    const bool win = base + windows <= 160 * 1024 && a_in.rel_ld >= 8 && a_in.rel_ld % 8 == 0 && a_in.rel_head_stride % 8 == 0;
    const size_t smem = base + (win ? windows : 0);
    if (smem > 160 * 1024) return hipErrorInvalidValue;  // sequences longer than ~600 tokens
    BhAttnArgs a = a_in;
    a.v_lds_off = nkb * 4096;
    a.rel_lds_off = nkb * 8192;
    a.win_lds_off = (int)base;
    a.wide_stores = (g_rel_wide_stores && bh_gemm_probe_permlane(stream) == hipSuccess && bh_gemm_swap_mode() == 0) ? 1 : 0;
    static size_t attr_smem[2] = {0, 0};
    const void* fn = win ? reinterpret_cast<const void*>(bh_attention_rel_kernel<true>) : reinterpret_cast<const void*>(bh_attention_rel_kernel<false>);
    if (smem > attr_smem[win ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_smem[win ? 1 : 0] = smem;
    }
    if (win)
        hipLaunchKernelGGL(bh_attention_rel_kernel<true>, dim3(n_heads, batch), dim3(512), smem, stream, a);
    else
        hipLaunchKernelGGL(bh_attention_rel_kernel<false>, dim3(n_heads, batch), dim3(512), smem, stream, a);
    return hipGetLastError();
}
