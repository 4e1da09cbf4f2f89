// gemm_f16.hip — MFMA GEMM of the bi-encoder forward pass (roofline: MFMA, fp16 dense).
//
//   C[M][N] = A[M][K] . B[N][K]^T  (+ bias)  (+ residual)  (GELU)      fp16 in / fp32 accumulate / fp16 out
//
// Replaces the torch.nn.Linear calls inside HF BertModel.forward that the reference reaches through
// AutoModel (models/retrievers/dense.py:16,40-44): QKV / attention-output / FFN projections
// (transformers modeling_bert.py: BertSelfAttention.query/key/value, BertSelfOutput.dense,
// BertIntermediate.dense + GELU, BertOutput.dense), with the bias add, the residual add and the erf-GELU
// fused into the epilogue instead of running as separate HBM passes.
//
// Both operands are K-contiguous — activations [tokens][d] and HF weights [out][in] — i.e. both are
// "rows of 128-byte lines", so both stream HBM/L2 -> LDS by LDS-DMA (`global_load_lds_dwordx4`, no VGPR
// round trip) in full lines, into the same XOR-permuted image the scan kernel uses (scan_topk.hip header:
// the permutation is applied to the per-lane SOURCE address; MFMA fragment reads are then conflict-free
// ds_read_b128).  A stage is one BK = 64 slice of the block tile: (BM + BN)/32 pieces of 32 rows x 128 B.
//
// Pipeline: R-deep ring of stages (R-1 stages in flight behind the one being consumed); per K-step ONE raw
// s_barrier with a counted `s_waitcnt vmcnt((R-2)*NL)` — loads stay in flight across barriers, never
// vmcnt(0) in the main loop for R >= 3 — placed in front of the stage's last 16-deep MFMA group, which also
// covers the first fragment reads of the next stage; fragment reads run one k-step ahead of the MFMAs.
//
// MFMA orientation: v_mfma_f32_32x32x16_f16 with the WEIGHT fragment as operand A (i = n) and the ACTIVATION
// fragment as operand B (j = m): lane l ends with token row m = l & 31 and output columns
// n = (v&3) + 8*(v>>2) + 4*(l>>5) — four consecutive columns per register quad; the two half-lanes of a row
// exchange quads (v_permlane32_swap) so that every lane stores 8 consecutive columns (16 bytes).
//
// Block -> tile map is XCD-aware: the 8 XCDs (block b runs on XCD b % 8) get contiguous ranges of tiles in
// (m-panel major, n minor) order, so one XCD's L2 sees whole A panels and the weight matrix once.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {

__device__ __forceinline__ float gelu_erf(float x) {
    // 0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below fp16 ulp)
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    return fmaf(0.5f * fabsf(x), erf_abs, 0.5f * x);
}

}  // namespace

// WM x WN waves; each wave owns a (TM*32) x (TN*32) sub-tile; R = ring depth; WIDE = 16-byte epilogue
// stores through the half-lane exchange (else 8-byte stores straight from the accumulator layout).
template <int WM, int WN, int TM, int TN, int R, bool WIDE, int OCC>
__global__ void __launch_bounds__(64 * WM * WN, OCC) bh_gemm_f16_kernel(BhGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int STAGE_BYTES = (PA + PB) * 4096;
    static_assert(((PA + PB) * 4) % NW == 0, "stage pieces must divide over the waves");
    constexpr int NL = (PA + PB) * 4 / NW;  // LDS-DMA instructions per wave per stage
    static_assert((R - 2) * NL <= 63, "vmcnt range");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ql = lane & 31, h = lane >> 5;

    // ---- XCD-aware tile assignment (bijective for any grid size)
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
        const int piece = idx >> 2, sub = idx & 3;
        const int row = 8 * sub + (lane >> 3);               // row inside the 32-row piece
        const int g = ((lane >> 4) & 1) | (sub << 1);        // g(row) of the XOR permutation
        const int chunk = (lane & 7) ^ g;                    // 16-byte chunk of the 128-byte line
        dst[i] = piece * 4096 + sub * 1024;
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
    // fragment read offsets inside a piece (scan_topk.hip: conflict-free ds_read_b128 pattern)
    const int rd_g = ((ql >> 1) & 1) | ((ql >> 3) << 1);
    unsigned rd_off[4];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
        rd_off[j4] = (unsigned)((ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * j4 + h) ^ rd_g) << 4));

    floatx16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[tm][tn][v] = 0.f;

    const int KT = a.K >> 6;
    int ikt = 0, islot = 0;
    // one LDS-DMA instruction of the stage being issued (i = 0 .. NL-1), then the cursor advance
    auto issue_piece = [&](int i) {
        const int kk = ikt < KT ? ikt : KT - 1;  // past the end: harmless re-fetch keeps vmcnt uniform
        unsigned char* sb = smem + islot * STAGE_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kk * 128),
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

    // Fragment registers are double-buffered over the four k-steps of a stage; the reads of k-step j+1 are
    // issued after the wait for k-step j's fragments and ahead of its MFMAs.  The stage hand-over (wait for
    // the next stage's DMA, barrier, re-issue into the slot just drained, first fragment reads of the next
    // stage) sits in front of the LAST k-step's MFMAs, which cover its latency.
    half8 xa[2][TM], wb[2][TN];
    auto read_frags = [&](int buf, const unsigned char* st, int j4) {
        const unsigned char* sa = st + (wm * TM) * 4096 + rd_off[j4];
        const unsigned char* sw = st + (PA + wn * TN) * 4096 + rd_off[j4];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) wb[buf][tn] = *reinterpret_cast<const half8*>(sw + tn * 4096);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xa[buf][tm] = *reinterpret_cast<const half8*>(sa + tm * 4096);
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
    read_frags(0, smem, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* st = smem + cslot * STAGE_BYTES;
        if (++cslot == R) cslot = 0;
#pragma unroll
        for (int j4 = 0; j4 < 3; ++j4) {
            wait_frags(j4 & 1);
            read_frags((j4 + 1) & 1, st, j4 + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(j4 & 1);
        }
        wait_frags(1);  // every read of this stage has returned: its slot may be overwritten
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * NL) : "memory");  // stage kt+1 landed
        read_frags(0, smem + cslot * STAGE_BYTES, 0);
        __builtin_amdgcn_sched_barrier(0);
        // last k-step: the LDS-DMA issue of stage kt+R is spread over the gaps between its MFMAs
#pragma unroll
        for (int t = 0; t < TM * TN; ++t) {
            acc[t / TN][t % TN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[1][t % TN], xa[1][t / TN], acc[t / TN][t % TN], 0, 0, 0);
#pragma unroll
            for (int i = t * NL / (TM * TN); i < (t + 1) * NL / (TM * TN); ++i) issue_piece(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail re-fetches before the block may exit

    // ---- epilogue
    const bool has_res = a.residual != nullptr;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + (wm * TM + tm) * 32 + ql;
        const bool m_ok = m < a.M;
        float bias_row = 0.f;
        if (a.bias_mode == 2 && m_ok) bias_row = (float)a.bias[m];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nt = n0 + (wn * TN + tn) * 32;
            floatx16 c = acc[tm][tn];
            if constexpr (WIDE) {
                // exchange register quads between the half-lanes of a row: afterwards registers
                // 8u .. 8u+7 hold columns nt + 8*(2u + h) + 0..7
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned x = __float_as_uint(c[8 * u + e]), y = __float_as_uint(c[8 * u + 4 + e]);
                        unsigned nx, ny;
                        if (a.swap_b) {  // hardware swaps vdst[0:31] <-> vsrc[32:63]
                            auto r = __builtin_amdgcn_permlane32_swap(y, x, false, false);
                            ny = r[0];
                            nx = r[1];
                        } else {  // hardware swaps vdst[32:63] <-> vsrc[0:31]
                            auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
                            nx = r[0];
                            ny = r[1];
                        }
                        c[8 * u + e] = __uint_as_float(nx);
                        c[8 * u + 4 + e] = __uint_as_float(ny);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = nt + 8 * (2 * u + h);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = c[8 * u + e] + bias_row;
                    if (n + 7 < a.N) {
                        if (a.bias_mode == 1) {
                            const half8 b8 = *reinterpret_cast<const half8*>(a.bias + n);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)b8[e];
                        }
                        if (has_res && m_ok) {
                            const half8 r8 = *reinterpret_cast<const half8*>(a.residual + (size_t)m * a.ldr + n);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                        }
                        if (a.gelu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
                        }
                        if (m_ok) {
                            half8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
                            *reinterpret_cast<half8*>(a.C + (size_t)m * a.ldc + n) = o;
                        }
                    } else if (m_ok) {  // ragged right edge: element-wise
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (n + e < a.N) {
                                float x = v[e];
                                if (a.bias_mode == 1) x += (float)a.bias[n + e];
                                if (has_res) x += (float)a.residual[(size_t)m * a.ldr + n + e];
                                if (a.gelu) x = gelu_erf(x);
                                a.C[(size_t)m * a.ldc + n + e] = (_Float16)x;
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = nt + 8 * gq + 4 * h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = c[4 * gq + e] + bias_row;
                    if (n + 3 < a.N) {
                        if (a.bias_mode == 1) {
                            const half4 b4 = *reinterpret_cast<const half4*>(a.bias + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)b4[e];
                        }
                        if (has_res && m_ok) {
                            const half4 r4 = *reinterpret_cast<const half4*>(a.residual + (size_t)m * a.ldr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
                        }
                        if (a.gelu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        if (m_ok) {
                            half4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                            *reinterpret_cast<half4*>(a.C + (size_t)m * a.ldc + n) = o;
                        }
                    } else if (m_ok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < a.N) {
                                float x = v[e];
                                if (a.bias_mode == 1) x += (float)a.bias[n + e];
                                if (has_res) x += (float)a.residual[(size_t)m * a.ldr + n + e];
                                if (a.gelu) x = gelu_erf(x);
                                a.C[(size_t)m * a.ldc + n + e] = (_Float16)x;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Probe of v_permlane32_swap's direction (documented: vdst[32:63] <-> vsrc[0:31]); the GEMM epilogue
// adapts to what the hardware does instead of trusting the reading of the manual.
__global__ void bh_permlane_probe_kernel(unsigned* out) {
    const unsigned lane = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}

// ------------------------------------------------------------------------------------------------
namespace {

template <int WM, int WN, int TM, int TN, int R, bool WIDE, int OCC>
hipError_t launch_cfg(const BhGemmArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem = (size_t)R * (BM + BN) / 32 * 4096;
    static_assert(smem <= 160 * 1024, "LDS ring exceeds the CU");
    auto kern = bh_gemm_f16_kernel<WM, WN, TM, TN, R, WIDE, OCC>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), smem, stream, a);
    return hipGetLastError();
}

int g_swap_b = -1;  // -1 unknown, 0 documented direction, 1 the other one

}  // namespace

hipError_t bh_gemm_probe_permlane(hipStream_t stream) {
    if (g_swap_b >= 0) return hipSuccess;
    unsigned* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, 128 * sizeof(unsigned));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bh_permlane_probe_kernel, dim3(1), dim3(64), 0, stream, d);
    unsigned hbuf[128];
    e = hipMemcpyAsync(hbuf, d, sizeof hbuf, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    // documented: r0 (vdst) = {lanes 0-31: own lane id, lanes 32-63: src of lane-32 = 100 + (lane-32)}
    if (hbuf[0] == 0u && hbuf[32] == 100u && hbuf[64 + 0] == 32u && hbuf[64 + 32] == 132u)
        g_swap_b = 0;
    else if (hbuf[0] == 132u && hbuf[32] == 32u && hbuf[64 + 0] == 100u && hbuf[64 + 32] == 0u)
        g_swap_b = 1;
    else
        return hipErrorUnknown;
    return hipSuccess;
}

int bh_gemm_swap_mode() { return g_swap_b; }

// variant: 0 = auto; 1..N = explicit tile configurations (bench / tests)
//   1: 128x128 tile, 4 waves (64x64 each), ring 2, two blocks per CU
//   2: 256x128 tile, 4 waves (128x64 each), ring 3
//   3: 128x128 tile, 4 waves, ring 4
//   4: 256x128 tile, 8 waves (64x64 each), ring 3
//   5: 256x256 tile, 8 waves (128x64 each), ring 2
//   6: as 2 with 8-byte epilogue stores (no half-lane exchange)
//   7: 128x256 tile, 4 waves (64x128 each), ring 3
hipError_t bh_launch_gemm_f16(const BhGemmArgs& a_in, int variant, hipStream_t stream) {
    BhGemmArgs a = a_in;
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (a.K <= 0 || (a.K & 63) || (a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (a.residual && (a.ldr & 7)))
        return hipErrorInvalidValue;
    hipError_t e = bh_gemm_probe_permlane(stream);
    if (e != hipSuccess) return e;
    a.swap_b = g_swap_b;
    if (variant == 0) variant = (a.M >= 256) ? 2 : 1;
    switch (variant) {
        case 1: return launch_cfg<2, 2, 2, 2, 2, true, 2>(a, stream);
        case 2: return launch_cfg<2, 2, 4, 2, 3, true, 1>(a, stream);
        case 3: return launch_cfg<2, 2, 2, 2, 4, true, 1>(a, stream);
        case 4: return launch_cfg<4, 2, 2, 2, 3, true, 2>(a, stream);
        case 5: return launch_cfg<2, 4, 4, 2, 2, true, 2>(a, stream);
        case 6: return launch_cfg<2, 2, 4, 2, 3, false, 1>(a, stream);
        case 7: return launch_cfg<2, 2, 2, 4, 3, true, 1>(a, stream);
    }
    return hipErrorInvalidValue;
}
