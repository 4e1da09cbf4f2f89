// gemm_f16_p16.h — the persistent encoder GEMM on v_mfma_f32_16x16x32_f16 (round 5; option gemm_mfma16).
//
// Same problem, tile, ring and store path as gemm_f16_persist.h's production configuration — C[M][N] = A[M][K] . B[N][K]^T + bias[n]
// (+ GELU), 256 x 256 tile, 8 waves (2 x 4, 128 tokens x 64 features each), BK = 64, LDS-DMA ring of two stages, XCD-aware
// persistent tile order, outputs through a wave-private 4 KiB of LDS as whole 128-byte lines — with the OTHER matrix instruction:
// the board sustains 1.88 PFLOP/s of 16x16x32 against 1.64 PFLOP/s of 32x32x16 on random fp16 data (701 vs 791 J per PFLOP:
// HISTORY.md "The power ceiling"), which is why the scan kernels use it; the GEMM had stayed on 32x32x16 since round 1.
//
// What changes against gemm_f16_persist.h:
//   * LDS image: chunk c of row r of a 32-row piece at (r >> 3) * 1024 + (r & 7) * 128 + ((c ^ g) << 4) with g = (r >> 1) & 7 — the
//     permutation the scan kernels use for this fragment pattern (lane (q16 = l & 15, lg = l >> 4) reads row q16 (+ 16), chunk
//     4 ks + lg: conflict-free, profiles/lds_swizzle_search.py); applied, as there, to the per-lane SOURCE address of the LDS-DMA.
//   * fragments: per k-step of 32 a wave reads 4 feature fragments (operand a: the weights, so that a lane ends with ONE token and 4
//     consecutive features) and 8 token fragments (operand b), 32 MFMAs on 32 independent accumulator quads (128 registers, as
//     before); feature fragments double-buffered over the k-steps, token fragments refilled rolling behind their 4 MFMAs.
//   * epilogue: lane (q16, lg) of block (tb, fb) holds token 16 tb + q16, features 16 fb + 4 lg + 0..3: one 8-byte LDS write per block
//     into the wave's 32-row x 128-byte staging image (16-byte chunks XORed with the row), read back and stored exactly as before.
// Epilogues: bias per column, optionally GELU (the Q | K, attention-output, FFN-up and FFN-down projections: 92 % of the encoder's
// GEMM time); everything else stays on gemm_f16_persist.h.
#pragma once
#include "gemm_f16_persist.h"

// SCHED bit 1: token fragments double-buffered in registers, all reads of the next k-step issued in the first six MFMA groups of the current one;
// bit 2: the stage refill issued two LDS-DMA instructions per group in the first four groups (instead of one in each of the eight)
template <int EPI, bool NT, int SCHED>
__global__ void __launch_bounds__(512, 2) bh_gemm_f16_p16kernel(BhGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert((EPI & ~(BH_EPI_BIAS_COL | BH_EPI_GELU)) == 0 && (EPI & BH_EPI_BIAS_COL) != 0, "bias per column, optional GELU");
    constexpr int BK = 64, WM = 2, WN = 4, R = 2;
    constexpr int NW = WM * WN;
    constexpr int BM = 256, BN = 256;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int PIECE = 4096, SUBS = 4;
    constexpr int STAGE_BYTES = (PA + PB) * PIECE;
    constexpr int NL = (PA + PB) * SUBS / NW;  // 8 LDS-DMA instructions per wave per stage
    constexpr int TB = 8, FB = 4;              // 16-token blocks / 16-feature blocks per wave
    constexpr int NLA = PA * SUBS / NW;        // instructions 0..3 fetch token rows, 4..7 feature rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q16 = lane & 15, lg = lane >> 4;

    // ---- this block's tiles (gemm_f16_persist.h: XCD x = block % 8 owns a contiguous range of the m-major tile order)
    const int tiles_n = a.N / BN;
    const int n_tiles = (a.M / BM) * tiles_n;
    int t_first, t_step, n_my;
    {
        const int G8 = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = n_tiles >> 3, r = n_tiles & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        t_first = start + j;
        t_step = G8;
        n_my = cnt > j ? (cnt - j + G8 - 1) / G8 : 0;
    }
    if (n_my == 0) return;

    // ---- per-lane LDS-DMA source offsets: instruction i of a wave fetches 8 rows of piece 2 i + (wave >> 2)
    unsigned offA, offB;
    int dst0;
    {
        const int sub = wave & 3, p0 = wave >> 2;
        const int row = 8 * sub + (lane >> 3);
        const int g = (row >> 1) & 7;
        const int chunk = (lane & 7) ^ g;
        offA = (unsigned)((p0 * 32 + row) * a.lda * 2 + chunk * 16);
        offB = (unsigned)((p0 * 32 + row) * a.ldb * 2 + chunk * 16);
        dst0 = p0 * PIECE + sub * 1024;
    }
    const unsigned strideA = (unsigned)(64 * a.lda * 2), strideB = (unsigned)(64 * a.ldb * 2);
    // fragment read offsets inside a 16-row block of a piece, one per k-step (32 dims) of a stage
    unsigned rd_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        rd_off[ks] = (unsigned)((q16 >> 3) * 1024 + (q16 & 7) * 128 + (((4 * ks + lg) ^ ((q16 >> 1) & 7)) << 4));

    const int KT = a.K / BK;
    const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.A);
    const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.B);

    // ---- issue cursor (runs one stage ahead of the consumer, across tile boundaries)
    int it = 0, ikt = 0, islot = 0;
    const unsigned char *curA, *curB;
    auto set_issue_tile = [&](int ord) {
        const int t = t_first + ord * t_step;
        const int tm0 = (t / tiles_n) * BM, tn0 = (t % tiles_n) * BN;
        curA = baseA + (size_t)tm0 * a.lda * 2;
        curB = baseB + (size_t)tn0 * a.ldb * 2;
    };
    set_issue_tile(0);
    auto issue_piece = [&](int i) {
        const unsigned char* ub = (i < NLA ? curA : curB) + (size_t)ikt * (BK * 2);
        unsigned off = i < NLA ? offA : offB;
        asm volatile("" : "+v"(off));  // ONE live offset register (gemm_f16_persist.h)
        off += i < NLA ? (unsigned)i * strideA : (unsigned)(i - NLA) * strideB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + off),
                                         (__attribute__((address_space(3))) void*)(smem + islot * STAGE_BYTES + dst0 + i * 2 * PIECE), 16, 0, 0);
    };
    auto issue_advance = [&]() {
        if (++islot == R) islot = 0;
        if (++ikt == KT) {
            if (it + 1 < n_my) {
                ++it;
                ikt = 0;
                set_issue_tile(it);
            } else {
                ikt = KT - 1;  // no further tile: harmless re-fetch keeps vmcnt uniform
            }
        }
    };
    auto issue_stage = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_piece(i);
        issue_advance();
    };

    floatx4 acc[TB][FB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) acc[tb][fb] = floatx4{0.f, 0.f, 0.f, 0.f};
    constexpr bool DBX = (SCHED & 1) != 0, DMA2 = (SCHED & 2) != 0;
    half8 xt[DBX ? 2 : 1][TB], wf[2][FB];
    // token block tb of the wave: piece wm * 4 + (tb >> 1), 16-row block tb & 1; feature block fb: piece PA + wn * 2 + (fb >> 1), block fb & 1
    auto read_wf = [&](int buf, const unsigned char* st, int ks) {
        const unsigned char* sw = st + (PA + wn * 2) * PIECE + rd_off[ks];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) wf[buf][fb] = *reinterpret_cast<const half8*>(sw + (fb >> 1) * PIECE + (fb & 1) * 2048);
    };
    auto read_xt = [&](int tb, const unsigned char* st, int ks) {
        xt[DBX ? ks : 0][tb] = *reinterpret_cast<const half8*>(st + (wm * 4 + (tb >> 1)) * PIECE + (tb & 1) * 2048 + rd_off[ks]);
    };

    int cslot = 0;
    auto stage = [&]() {
        const unsigned char* st = smem + cslot * STAGE_BYTES;
        if (++cslot == R) cslot = 0;
        // k-step 0 (fragments wf[0], xt[] are in registers: the wait hipcc puts at the loop head covers exactly them, so no read is
        // issued before the first MFMAs); the reads of k-step 1 follow the MFMA group that frees their registers
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
                acc[tb][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][fb], xt[0][tb], acc[tb][fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DBX) {  // 12 reads over groups 0..5
                if (tb == 0) read_wf(1, st, 1);
                if (tb >= 1 && tb <= 4) { read_xt(2 * tb - 2, st, 1); read_xt(2 * tb - 1, st, 1); }
            } else {
                if (tb == 0) read_wf(1, st, 1);
                read_xt(tb, st, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 1.  Every read of this stage must have returned before its slot is handed back:
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) asm volatile("" : "+v"(wf[1][fb]));
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) asm volatile("" : "+v"(xt[DBX ? 1 : 0][tb]));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // next stage landed (ring 2)
        const unsigned char* nst = smem + cslot * STAGE_BYTES;
        if constexpr (!DBX) read_wf(0, nst, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
                acc[tb][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][fb], xt[DBX ? 1 : 0][tb], acc[tb][fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the refill goes into the slot everybody just left
            if constexpr (DMA2) {
                if (tb < 4) { issue_piece(2 * tb); issue_piece(2 * tb + 1); }
            } else {
                issue_piece(tb);  // NL == TB: one LDS-DMA instruction per group of four MFMAs
            }
            if constexpr (DBX) {
                if (tb == 0) read_wf(0, nst, 0);
                if (tb >= 1 && tb <= 4) { read_xt(2 * tb - 2, nst, 0); read_xt(2 * tb - 1, nst, 0); }
            } else {
                read_xt(tb, nst, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_advance();
    };
    static_assert(NL == TB, "issue schedule");

    // ---- pipeline start
    issue_stage();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    issue_stage();
    read_wf(0, smem, 0);
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) read_xt(tb, smem, 0);

    for (int ti = 0; ti < n_my; ++ti) {
        const int t = t_first + ti * t_step;
        const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
        for (int kt = 0; kt < KT; ++kt) stage();  // (the accumulators are zero: the epilogue leaves them so — a separate first stage with a
                                                  // zero C operand made hipcc spill six accumulator quads per tile)

        // ---- epilogue: bias (+ GELU), fp16, through the wave's staging image, out as whole 128-byte lines
        unsigned char* stg = smem + R * STAGE_BYTES + wave * 4096;  // 32 token rows x 128 bytes
        const int rrow = lane >> 3, rch = lane & 7;                  // read-back: instruction i takes rows 8 i + rrow, 16-byte chunk rch
        _Float16* gptr = a.C + (size_t)(m0 + wm * 128 + rrow) * a.ldc + n0 + wn * 64 + rch * 8;
        half4 bias4[FB];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) bias4[fb] = *reinterpret_cast<const half4*>(a.bias + n0 + wn * 64 + fb * 16 + 4 * lg);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {  // 32-token parts of the wave's 128 tokens
#pragma unroll
            for (int half_ = 0; half_ < 2; ++half_) {
                const int tb = 2 * tp + half_;
                const int tr = half_ * 16 + q16;  // token row inside the part
#pragma unroll
                for (int fb = 0; fb < FB; ++fb) {
                    half4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[tb][fb][r] + (float)bias4[fb][r];
                        if constexpr ((EPI & BH_EPI_GELU) != 0) v = bh_gemm::gelu_erf(v);
                        o[r] = (_Float16)v;
                    }
                    acc[tb][fb] = floatx4{0.f, 0.f, 0.f, 0.f};
                    const int chunk = fb * 2 + (lg >> 1);  // the 16-byte chunk of the row's 128 bytes that holds features 16 fb + 4 lg ..
                    *reinterpret_cast<half4*>(stg + tr * 128 + ((chunk ^ (tr & 7)) << 4) + (lg & 1) * 8) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's own writes; no other wave touches stg)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const half8 v = *reinterpret_cast<const half8*>(stg + (8 * i + rrow) * 128 + ((rch ^ rrow) << 4));
                half8* p = reinterpret_cast<half8*>(gptr + (size_t)(tp * 32 + 8 * i) * a.ldc);
                if constexpr (NT)
                    __builtin_nontemporal_store(v, p);
                else
                    *p = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads have landed before the next part overwrites stg)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, bool NT, int SCHED>
hipError_t bh_gemm_launch_p16(const BhGemmArgs& a, int n_cu, hipStream_t stream) {
    constexpr size_t smem = 2 * 16 * 4096 + 8 * 4096;
    auto kern = bh_gemm_f16_p16kernel<EPI, NT, SCHED>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if ((a.M & 255) || (a.N & 255) || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    int grid = n_cu / 8 * 8;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, a);
    return hipGetLastError();
}

hipError_t bh_gemm_p16(const BhGemmArgs& a, int epi, bool nontemporal, int sched, hipStream_t s);  // gemm_f16_d.hip
