// gemm_f16_kernel.h — the MFMA GEMM kernel template of the bi-encoder forward pass (roofline: MFMA).
// Instantiated by gemm_f16.hip / gemm_f16_b.hip; see gemm_f16.hip for the dispatch and the design notes.
#pragma once
#include "bh_device.h"
#include "bh_kernels.h"

// epilogue feature bits (compile-time)
#define BH_EPI_BIAS_COL 1  // + bias[n]
#define BH_EPI_BIAS_ROW 2  // + bias[m]
#define BH_EPI_RESIDUAL 4  // + residual[m][n]
#define BH_EPI_GELU 8      // erf-GELU
#define BH_EPI_SEGMAX 16   // persistent kernel only: no store; relu + per-sequence max into seg_out (SPLADE head)
#define BH_EPI_BATCHED 32  // persistent kernel only: a.batch problems of one shape in one launch (BhGemmArgs::batch_stride_*)
#define BH_EPI_SWIGLU 64   // persistent kernel only: the N columns are (gate, up) PAIRS — weight rows interleaved g0, u0, g1, u1, ... —
                           // and C [M][N / 2] (row stride ldc) = silu(gate) * up: the gated feed-forward of NomicBert folded in the
                           // epilogue (a lane's 8 consecutive columns are 4 pairs -> one 8-byte store)

#define BH_EPI_LNA 128     // persistent kernel, full-line stores: the token operand is a pre-LayerNorm tensor, the LayerNorm is applied
                           // algebraically in the epilogue (gemm_f16_persist.h; BhGemmArgs::ln_stats / ln_c, folded weights and bias)
#define BH_EPI_RESLN 256   // persistent kernel, full-line stores: + residual rows (normalised on the fly when res_stats is set), and the
                           // per-row (sum, sum of squares) of the stored outputs into BhGemmArgs::stats_out

#define BH_EPI_ROTARY 512  // 16x16x32 kernel only: rotate-half RoPE on the (bias-added) output by BhGemmArgs::rot_pos / rot_cs

namespace bh_gemm {

// erf-GELU, 0.5 x (1 + erf(x / sqrt 2)) = x Phi(x), without a transcendental instruction (round 5):
//     Phi(x) - 1/2 ~= xc R(t),  xc = x clamped to [-4.5, 4.5],  t = xc^2 (2 / 4.5^2) - 1 in [-1, 1],  R a degree-9 polynomial (Horner in t)
//     gelu(x) ~= max(x, -4.5) (1/2 + xc R(t))
// R = weighted minimax fit of (Phi(x) - 1/2) / x over [0, 4.5] under the constraint 4.5 R(1) = 1/2 (so that the bracket is 1 / 0 at the clamp,
// up to rounding; profiles/fit_gelu.py, tests/test_gelu_formula.py): max |error| 1.6e-5 in this fp32 arithmetic over all x (|x| <= 12 dense grid + +-10^0..4.8), 1.7e-7 for
// x < -6, relative 6e-8 for x > 6 — inside what the sigmoid form it replaces had (2.55e-5), a thirtieth of an fp16 half-ulp at |y| ~ 1 and far
// below what rounding the pre-activation to fp16 (which the reference's fp16 forward does before its GELU) moves the result by.
// Why: the FFN-up GEMM's epilogue is VALU-bound (210 M outputs per launch; both waves of a SIMD are in it at the same time) and the two
// transcendentals of the sigmoid form (v_exp_f32, v_rcp_f32: quarter rate) were half of its ~66 cycles per output.  Every operation here has a
// packed form (v_pk_mul_f32 / v_pk_fma_f32: two outputs per instruction) except the two clamps: gelu_erf2 is the same arithmetic on a pair, bit for
// bit (IEEE fma / mul per lane), so kernels may use either and all agree.
constexpr float GELU_C = 4.5f, GELU_TA = 2.0f / (4.5f * 4.5f);
constexpr float GELU_R[10] = {1.569042617e-01f, -7.717613388e-02f, 5.471959913e-02f, -4.042935046e-02f, 2.830150923e-02f,
                              -1.751939744e-02f, 1.080767254e-02f, -7.782468650e-03f, 4.237509094e-03f, -9.520901429e-04f};
__device__ __forceinline__ float gelu_erf(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -GELU_C, GELU_C);
    const float t = __builtin_fmaf(xc * xc, GELU_TA, -1.0f);
    float r = GELU_R[9];
#pragma unroll
    for (int k = 8; k >= 0; --k) r = __builtin_fmaf(r, t, GELU_R[k]);
    const float p = __builtin_fmaf(xc, r, 0.5f);
    return fmaxf(x, -GELU_C) * p;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    f32x2 xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -GELU_C, GELU_C);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -GELU_C, GELU_C);
    const f32x2 t = __builtin_elementwise_fma(xc * xc, f32x2{GELU_TA, GELU_TA}, f32x2{-1.0f, -1.0f});
    f32x2 r = {GELU_R[9], GELU_R[9]};
#pragma unroll
    for (int k = 8; k >= 0; --k) r = __builtin_elementwise_fma(r, t, f32x2{GELU_R[k], GELU_R[k]});
    const f32x2 p = __builtin_elementwise_fma(xc, r, f32x2{0.5f, 0.5f});
    f32x2 xm;
    xm[0] = fmaxf(x[0], -GELU_C);
    xm[1] = fmaxf(x[1], -GELU_C);
    return xm * p;
}
// two pairs in lockstep (a dependent packed instruction costs a wait state; the other pair's fills it)
__device__ __forceinline__ void gelu_erf2x2(f32x2& a, f32x2& b) {
    f32x2 ac, bc;
    ac[0] = __builtin_amdgcn_fmed3f(a[0], -GELU_C, GELU_C);
    ac[1] = __builtin_amdgcn_fmed3f(a[1], -GELU_C, GELU_C);
    bc[0] = __builtin_amdgcn_fmed3f(b[0], -GELU_C, GELU_C);
    bc[1] = __builtin_amdgcn_fmed3f(b[1], -GELU_C, GELU_C);
    const f32x2 ta = __builtin_elementwise_fma(ac * ac, f32x2{GELU_TA, GELU_TA}, f32x2{-1.0f, -1.0f});
    const f32x2 tb = __builtin_elementwise_fma(bc * bc, f32x2{GELU_TA, GELU_TA}, f32x2{-1.0f, -1.0f});
    f32x2 ra = {GELU_R[9], GELU_R[9]}, rb = ra;
#pragma unroll
    for (int k = 8; k >= 0; --k) {
        ra = __builtin_elementwise_fma(ra, ta, f32x2{GELU_R[k], GELU_R[k]});
        rb = __builtin_elementwise_fma(rb, tb, f32x2{GELU_R[k], GELU_R[k]});
    }
    const f32x2 pa = __builtin_elementwise_fma(ac, ra, f32x2{0.5f, 0.5f});
    const f32x2 pb = __builtin_elementwise_fma(bc, rb, f32x2{0.5f, 0.5f});
    a[0] = fmaxf(a[0], -GELU_C);
    a[1] = fmaxf(a[1], -GELU_C);
    b[0] = fmaxf(b[0], -GELU_C);
    b[1] = fmaxf(b[1], -GELU_C);
    a *= pa;
    b *= pb;
}

}  // namespace bh_gemm

// BK     K-slice per pipeline stage: 64 (128-byte row segments) or 32 (64-byte row segments)
// WM,WN  waves per block along M / N;  TM,TN  32x32 MFMA tiles per wave along M / N
// R      LDS ring depth (stages);  EPI  epilogue bits;  OCC  blocks-per-CU * waves-per-block / 4
// GEN    generic epilogue: bounds checks on every element, run-time epilogue flags, no lane exchange
// ABL    bench-only ablation bits (0 in production): 1 no LDS-DMA in the main loop, 2 no MFMA, 4 no fragment
//        reads, 8 no epilogue, 16 epilogue math without its stores.  Results are garbage when ABL != 0.
//
// LDS image of a 32-row piece (bank-conflict-free ds_read_b128 for the MFMA fragment pattern "lane l reads row
// l&31, 16-byte chunk 2j + (l>>5)"; the permutation is applied to the per-lane SOURCE address of the LDS-DMA):
//   BK = 64: chunk c of row r at (r>>3)*1024 + (r&7)*128 + ((c ^ g)<<4), g = ((r>>1)&1) | ((r>>3)<<1)
//   BK = 32: chunk c of row r at  r*64 + ((c ^ g)<<4),                   g = (r>>2)&3
template <int BK, int WM, int WN, int TM, int TN, int R, int EPI, int OCC, bool GEN = false, int ABL = 0>
__global__ void __launch_bounds__(64 * WM * WN, OCC) bh_gemm_f16_kernel(BhGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(BK == 64 || BK == 32, "BK");
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int PIECE = 32 * BK * 2;       // bytes per 32-row piece
    constexpr int SUBS = PIECE / 1024;       // LDS-DMA instructions per piece (1 KiB each)
    constexpr int CPR = BK / 8;              // 16-byte chunks per row segment
    constexpr int RPI = 64 / CPR;            // rows per LDS-DMA instruction
    constexpr int KS = BK / 16;              // MFMA k-steps per stage
    constexpr int STAGE_BYTES = (PA + PB) * PIECE;
    static_assert(((PA + PB) * SUBS) % NW == 0, "stage pieces must divide over the waves");
    constexpr int NL = (PA + PB) * SUBS / NW;  // LDS-DMA instructions per wave per stage
    static_assert((R - 2) * NL <= 63, "vmcnt range");
    constexpr int NMF = TM * TN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ql = lane & 31, h = lane >> 5;

    // ---- optional start stagger (first round of blocks only): spreads the blocks' epilogues (VALU + store
    // bursts) over time instead of all CUs hitting the memory system in the same microseconds
    if (a.stagger_phases > 1 && (int)blockIdx.x < a.stagger_first_round) {
        const int phase = (int)((blockIdx.x >> 3) % (unsigned)a.stagger_phases);
        for (int i = 0; i < phase * a.stagger_unit; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // ---- XCD-aware tile assignment (bijective for any grid size): block b runs on XCD b % 8
    const int tiles_n = (a.N + BN - 1) / BN;
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    // ---- per-lane LDS-DMA sources (one per instruction of a stage) and wave-uniform destinations
    const unsigned char* src[NL];
    int dst[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = wave + NW * i;
        const int piece = idx / SUBS, sub = idx % SUBS;
        const int row = RPI * sub + lane / CPR;  // row inside the 32-row piece
        const int g = BK == 64 ? (((row >> 1) & 1) | ((row >> 3) << 1)) : ((row >> 2) & 3);
        const int chunk = (lane % CPR) ^ g;
        dst[i] = piece * PIECE + sub * 1024;
        if (piece < PA) {
            int rr = m0 + piece * 32 + row;
            rr = rr < a.M ? rr : a.M - 1;
            src[i] = reinterpret_cast<const unsigned char*>(a.A + (size_t)rr * a.lda) + chunk * 16;
        } else {
            int rr = n0 + (piece - PA) * 32 + row;
            rr = rr < a.N ? rr : a.N - 1;
            src[i] = reinterpret_cast<const unsigned char*>(a.B + (size_t)rr * a.ldb) + chunk * 16;
        }
    }
    // fragment read offsets inside a piece, one per k-step of a stage
    unsigned rd_off[KS];
    {
        const int g = BK == 64 ? (((ql >> 1) & 1) | ((ql >> 3) << 1)) : ((ql >> 2) & 3);
#pragma unroll
        for (int j = 0; j < KS; ++j)
            rd_off[j] = BK == 64 ? (unsigned)((ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * j + h) ^ g) << 4))
                                 : (unsigned)(ql * 64 + (((2 * j + h) ^ g) << 4));
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[tm][tn][v] = 0.f;

    const int KT = a.K / BK;
    int ikt = 0, islot = 0;
    bool dma_on = true;
    // one LDS-DMA instruction of the stage being issued (i = 0 .. NL-1); issue_advance moves the cursor
    auto issue_piece = [&](int i) {
        if constexpr ((ABL & 1) != 0) {
            if (!dma_on) return;
        }
        const int kk = ikt < KT ? ikt : KT - 1;  // past the end: harmless re-fetch keeps vmcnt uniform
        unsigned char* sb = smem + islot * STAGE_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kk * (BK * 2)),
                                         (__attribute__((address_space(3))) void*)(sb + dst[i]), 16, 0, 0);
    };
    auto issue_advance = [&]() {
        ++ikt;
        if (++islot == R) islot = 0;
    };
    auto issue_stage = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_piece(i);
        issue_advance();
    };
#pragma unroll
    for (int p = 0; p < R - 1; ++p) issue_stage();

    // Fragment registers are double-buffered over the k-steps of a stage; the reads of k-step j+1 are issued
    // after the wait for k-step j's fragments and ahead of its MFMAs.  The stage hand-over (wait for the next
    // stage's DMA, barrier, first fragment reads of the next stage, re-issue into the slot just drained) sits in
    // front of the LAST k-step's MFMAs, which cover its latency; the LDS-DMA issue is spread over their gaps.
    half8 xa[2][TM], wb[2][TN];
    auto read_frags = [&](int buf, const unsigned char* st, int j) {
        if constexpr ((ABL & 4) != 0) return;
        const unsigned char* sa = st + (wm * TM) * PIECE + rd_off[j];
        const unsigned char* sw = st + (PA + wn * TN) * PIECE + rd_off[j];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) wb[buf][tn] = *reinterpret_cast<const half8*>(sw + tn * PIECE);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xa[buf][tm] = *reinterpret_cast<const half8*>(sa + tm * PIECE);
    };
    auto wait_frags = [&](int buf) {
        // fake use: makes hipcc place its lgkmcnt wait for THIS buffer here, before the next reads are issued
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(wb[buf][tn]));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(xa[buf][tm]));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](int buf) {
        if constexpr ((ABL & 2) != 0) return;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[buf][tn], xa[buf][tm], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * NL) : "memory");  // stage 0 landed
    issue_stage();                                                                       // stage R-1
    int cslot = 0;
    if constexpr ((ABL & 4) != 0) {  // fragments never read: give the MFMAs defined operands
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int e = 0; e < 8; ++e) wb[b2][tn][e] = (_Float16)(0.01f * (lane + e));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int e = 0; e < 8; ++e) xa[b2][tm][e] = (_Float16)(0.02f * (lane - e));
        }
    }
    dma_on = false;  // (ablation 1: only the prologue stages were fetched)
    read_frags(0, smem, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* st = smem + cslot * STAGE_BYTES;
        if (++cslot == R) cslot = 0;
#pragma unroll
        for (int j = 0; j < KS - 1; ++j) {
            wait_frags(j & 1);
            read_frags((j + 1) & 1, st, j + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(j & 1);
        }
        wait_frags(1);  // every read of this stage has returned: its slot may be overwritten
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * NL) : "memory");  // stage kt+1 landed
        read_frags(0, smem + cslot * STAGE_BYTES, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NMF; ++t) {
            if constexpr ((ABL & 2) == 0)
                acc[t / TN][t % TN] =
                    __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[1][t % TN], xa[1][t / TN], acc[t / TN][t % TN], 0, 0, 0);
#pragma unroll
            for (int i = t * NL / NMF; i < (t + 1) * NL / NMF; ++i) issue_piece(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail re-fetches before the block may exit

    // ---- epilogue ------------------------------------------------------------------------------------------
    if constexpr ((ABL & 8) != 0) {
        if (a.bias_mode == 12345) {  // never true: keeps the accumulators live without an epilogue
            float t = 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int v = 0; v < 16; ++v) t += acc[tm][tn][v];
            a.C[tid] = (_Float16)t;
        }
        return;
    }
    if constexpr (!GEN) {
        // Interior tile (the launcher guarantees it): no bounds checks.  The two half-lanes of a row exchange
        // register quads (v_permlane32_swap, documented direction vdst[32:63] <-> vsrc[0:31]; the launcher routes
        // devices that swap the other way to the generic kernel) so that each lane holds 8 consecutive columns
        // per 16-column half: registers 8u .. 8u+7 <-> columns nt + 8*(2u + h) + 0..7, one 16-byte store each.
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = m0 + (wm * TM + tm) * 32 + ql;
            float bias_row = 0.f;
            if constexpr ((EPI & BH_EPI_BIAS_ROW) != 0) bias_row = (float)a.bias[m];
            // row-major output, or (c_block_rows != 0) blocked by 64 columns: (m, n) at
            // C[(n / 64) * c_block_rows * 64 + m * 64 + n % 64]; a 16-byte store never straddles a block
            _Float16* crow = a.c_block_rows ? a.C + (size_t)m * 64 : a.C + (size_t)m * a.ldc;
            const size_t cblk = (size_t)a.c_block_rows * 64;
            const _Float16* rrow = nullptr;
            if constexpr ((EPI & BH_EPI_RESIDUAL) != 0) rrow = a.residual + (size_t)m * a.ldr;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nt = n0 + (wn * TN + tn) * 32;
                floatx16 c = acc[tm][tn];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[8 * u + e]),
                                                                  __float_as_uint(c[8 * u + 4 + e]), false, false);
                        c[8 * u + e] = __uint_as_float(r[0]);
                        c[8 * u + 4 + e] = __uint_as_float(r[1]);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = nt + 8 * (2 * u + h);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = c[8 * u + e] + bias_row;
                    if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) {
                        const half8 b8 = *reinterpret_cast<const half8*>(a.bias + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)b8[e];
                    }
                    if constexpr ((EPI & BH_EPI_RESIDUAL) != 0) {
                        const half8 r8 = *reinterpret_cast<const half8*>(rrow + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                    }
                    if constexpr ((EPI & BH_EPI_GELU) != 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = bh_gemm::gelu_erf(v[e]);
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
                    if constexpr ((ABL & 16) != 0) {  // ablation: epilogue math without the stores
                        if (a.bias_mode == 12345) *reinterpret_cast<half8*>(crow + n) = o;
                    } else if (a.c_block_rows) {
                        *reinterpret_cast<half8*>(crow + (size_t)(n >> 6) * cblk + (n & 63)) = o;
                    } else {
                        *reinterpret_cast<half8*>(crow + n) = o;
                    }
                }
            }
        }
    } else {
        // Generic epilogue (edge strips; devices with the other swap direction; unusual epilogue combinations):
        // straight from the accumulator layout — lane (row m = l&31, half h) register v holds column
        // nt + (v&3) + 8*(v>>2) + 4*h — element-wise, every access bounds-checked, flags read at run time.
        const bool has_res = a.residual != nullptr;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = m0 + (wm * TM + tm) * 32 + ql;
            if (m >= a.M) continue;
            const float bias_row = a.bias_mode == 2 ? (float)a.bias[m] : 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nt = n0 + (wn * TN + tn) * 32 + 4 * h;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int n = nt + (v & 3) + 8 * (v >> 2);
                    if (n < a.N) {
                        float x = acc[tm][tn][v] + bias_row;
                        if (a.bias_mode == 1) x += (float)a.bias[n];
                        if (has_res) x += (float)a.residual[(size_t)m * a.ldr + n];
                        if (a.gelu) x = bh_gemm::gelu_erf(x);
                        a.C[(size_t)m * a.ldc + n] = (_Float16)x;
                    }
                }
            }
        }
    }
}

// One launch of one configuration over the tile grid covering M x N.
template <int BK, int WM, int WN, int TM, int TN, int R, int EPI, int OCC, bool GEN = false, int ABL = 0>
hipError_t bh_gemm_launch_cfg(const BhGemmArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem = (size_t)R * (BM + BN) / 32 * (32 * BK * 2);
    static_assert(smem <= 160 * 1024, "LDS ring exceeds the CU");
    auto kern = bh_gemm_f16_kernel<BK, WM, WN, TM, TN, R, EPI, OCC, GEN, ABL>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), smem, stream, a);
    return hipGetLastError();
}

// Dispatch over the epilogue bits that the encoder uses: none, bias-col, bias-row, bias-col+residual,
// bias-col+GELU.  (Other combinations are served by the generic configuration.)
#define BH_GEMM_DISPATCH_EPI(epi, CALL)                                            \
    switch (epi) {                                                                 \
        case 0: return CALL(0);                                                    \
        case BH_EPI_BIAS_COL: return CALL(BH_EPI_BIAS_COL);                        \
        case BH_EPI_BIAS_ROW: return CALL(BH_EPI_BIAS_ROW);                        \
        case BH_EPI_BIAS_COL | BH_EPI_RESIDUAL: return CALL(BH_EPI_BIAS_COL | BH_EPI_RESIDUAL); \
        case BH_EPI_BIAS_COL | BH_EPI_GELU: return CALL(BH_EPI_BIAS_COL | BH_EPI_GELU);         \
        default: return hipErrorNotSupported;                                      \
    }

// configuration entry points (defined in gemm_f16.hip / gemm_f16_b.hip)
hipError_t bh_gemm_cfg1(const BhGemmArgs& a, int epi, hipStream_t s);  // BK64 128x128  4 waves ring 2, 2 blocks/CU
hipError_t bh_gemm_cfg2(const BhGemmArgs& a, int epi, hipStream_t s);  // BK32 256x128  4 waves ring 3, 2 blocks/CU
hipError_t bh_gemm_cfg3(const BhGemmArgs& a, int epi, hipStream_t s);  // BK32 256x256  8 waves ring 4
hipError_t bh_gemm_cfg4(const BhGemmArgs& a, int epi, hipStream_t s);  // BK64 256x128  8 waves ring 3
hipError_t bh_gemm_cfg5(const BhGemmArgs& a, int epi, hipStream_t s);  // BK64 256x256  8 waves ring 2
hipError_t bh_gemm_generic(const BhGemmArgs& a, int epi, hipStream_t s);  // BK64 128x128, bounds-checked, any epi
hipError_t bh_gemm_ablate(const BhGemmArgs& a, int which, hipStream_t s);  // bench-only
