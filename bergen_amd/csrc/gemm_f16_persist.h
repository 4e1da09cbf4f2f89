// gemm_f16_persist.h — persistent form of the encoder GEMM (256x256 tile, 8 waves, BK = 64, ring 2).
//
// Why (profiles/README.md, GEMM ablations): with one 128-KiB block per CU, a K = 768 tile spends as long in its
// serialised tail — draining 128 KiB of output through the store queue at the CU's share of HBM write bandwidth,
// plus the pipeline refill of the next block — as in its 12-stage main loop.  Here a workgroup stays resident
// and walks its tiles:
//   * the LDS-DMA pipeline never drains: while a tile's last stages are consumed, the stages issued behind them
//     already belong to the NEXT tile;
//   * a finished tile's accumulators go through the epilogue math (bias / GELU / fp16 convert) into 64 packed
//     registers, and the 16-byte stores are issued two per stage inside the NEXT tile's main loop, between its
//     MFMAs — the store queue drains while the matrix pipe works.
// Everything else (LDS image, fragment pattern, MFMA orientation, XCD-aware tile order) is gemm_f16_kernel.h's.
#pragma once
#include <type_traits>

#include "gemm_f16_kernel.h"

// PST bits: 1 = the stores of a tile are issued in one burst right after its epilogue math (production: measured
// faster — on gfx950 stores share the vmcnt counter with the LDS-DMA loads, so stores spread over the stages make
// every stage hand-over wait for store acknowledgements); 0 = stores deferred two per stage into the next tile's
// main loop (kept as an ablation); 2 = non-temporal stores; 4 / 8 = bench-only (results invalid): epilogue math
// without its stores / no epilogue at all.
// PST bit 32 (with bit 1 set; row-major C only) = FULL-LINE STORES: a lane leaves the MFMA layout with 8 consecutive columns of ONE
// row, so a burst store instruction touches 32 rows with 32 bytes each — 32 partial cache lines.  Here every wave passes each
// 32-row x 64-column part of its output through 4 KiB of LDS of its own (the 32 KiB the ring leaves free; 16-byte chunks XORed with
// the row: conflict-free both ways; no barrier, a wave's LDS instructions execute in order) and stores it back out as 8 rows x 128
// bytes per instruction: whole lines.  Same values, same bits.
// PST bit 16 (with bit 1 clear) = ALTERNATING LOADER TEAMS: the eight waves form two teams of four (one wave per SIMD each);
// in even stages team 0 issues the WHOLE stage refill (16 LDS-DMA instructions per wave) and team 1 issues four of its
// deferred stores, in odd stages the roles swap.  Why: the stage hand-over needs "my LDS-DMA loads have landed" =
// s_waitcnt vmcnt(0) on a ring of two, and vmcnt counts stores too — with every wave loading in every stage, a deferred
// store had to be acknowledged within the stage it was issued in (variant 8: slower than a burst).  Here a wave waits only
// before the barrier that follows a stage in which it LOADED; the stores it issued one stage earlier have had two whole
// stages (~3 us) to be acknowledged, and a wave that only stored does not wait at all.  The stores of a tile (128 KiB per
// workgroup, 5 us as a burst with every CU bursting at the same time) then drain under the next tile's MFMAs.
// Needs an even number of stages >= 8 per tile (K = 768: 12, K = 3072: 48): the team roles are then the same in every tile.
namespace bh_gemm {
// v = the value of `x` in the lane DPP control CTRL names (all rows, all banks, bound_ctrl: no lane is without a source here)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, true));
}
}  // namespace bh_gemm
using bh_gemm::dpp_f32;

template <int EPI, int PST = 0>
__global__ void __launch_bounds__(512, 2) bh_gemm_f16_pkernel(BhGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = 64, WM = 2, WN = 4, TM = 4, TN = 2, R = 2;
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int PIECE = 4096, SUBS = 4, KS = 4;
    constexpr int STAGE_BYTES = (PA + PB) * PIECE;
    constexpr int NL = (PA + PB) * SUBS / NW;  // 8 LDS-DMA instructions per wave per stage
    constexpr int NMF = TM * TN;               // 8 MFMAs per k-step
    constexpr int NST = TM * TN * 2;           // 16 deferred 16-byte stores per lane per tile
    constexpr int SLOTS = 8;                   // main-loop stages that carry deferred stores (2 each)
    static_assert(NST == 2 * SLOTS, "store schedule");
    constexpr bool ALT = (PST & 16) != 0 && (PST & 1) == 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ql = lane & 31, h = lane >> 5;
    // ---- this block's tiles: XCD x (= block % 8) owns a contiguous range of the (m-major, n-minor) tile order;
    // its blocks j = block / 8 take tiles start + j, start + j + G/8, ... so that the XCD's CUs work on
    // neighbouring tiles at the same time
    constexpr bool BATCHED = (EPI & BH_EPI_BATCHED) != 0;
    const int tiles_n = a.N / BN;
    const int tiles_1 = (a.M / BM) * tiles_n;                   // tiles of one problem
    const int n_tiles = BATCHED ? tiles_1 * a.batch : tiles_1;  // batched: problem-major tile order
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
    // optional start stagger (bench knob): spreads the workgroups' store bursts over time
    if (a.stagger_phases > 1) {
        const int phase = (int)((blockIdx.x >> 3) % (unsigned)a.stagger_phases);
        for (int i = 0; i < phase * a.stagger_unit; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- per-lane LDS-DMA source offsets inside a tile.  Instruction i of a wave fetches 8 rows of piece
    // 2 i + (wave >> 2) (i < 4: A pieces, i >= 4: B pieces), so its offset is a per-lane base plus a uniform stride.
    constexpr int NLA = PA * SUBS / NW;  // 4: instructions 0..3 fetch A rows, 4..7 fetch B rows
    static_assert(NLA * NW == PA * SUBS && NW == 8 && SUBS == 4, "piece schedule");
    // ALT: a team's wave fetches sub-piece (wave & 3) of EVERY piece: 16 instructions, pieces 0..7 = A rows, 8..15 = B rows
    // (p0 = 0, piece stride 32 rows instead of 64)
    const int team = wave >> 2;
    unsigned offA, offB;  // byte offsets of the lane's source inside the tile's A rows / B rows for i = 0 / i = NLA
    int dst0;             // wave-uniform LDS destination of instruction 0; instruction i adds i * 2 * PIECE
    {
        const int sub = wave & 3, p0 = ALT ? 0 : wave >> 2;
        const int row = 8 * sub + (lane >> 3);
        const int g = ((row >> 1) & 1) | ((row >> 3) << 1);
        const int chunk = (lane & 7) ^ g;
        offA = (unsigned)((p0 * 32 + row) * a.lda * 2 + chunk * 16);
        offB = (unsigned)((p0 * 32 + row) * a.ldb * 2 + chunk * 16);
        dst0 = p0 * PIECE + sub * 1024;
    }
    const unsigned strideA = (unsigned)((ALT ? 32 : 64) * a.lda * 2), strideB = (unsigned)((ALT ? 32 : 64) * a.ldb * 2);
    unsigned rd_off[KS];
    {
        const int g = ((ql >> 1) & 1) | ((ql >> 3) << 1);
#pragma unroll
        for (int j = 0; j < KS; ++j)
            rd_off[j] = (unsigned)((ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * j + h) ^ g) << 4));
    }

    const int KT = a.K / BK;
    const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.A);
    const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.B);

    // ---- issue cursor (runs R-1 stages ahead of the consumer, across tile boundaries)
    int it = 0, ikt = 0, islot = 0;  // tile ordinal, k-stage inside it, ring slot
    const unsigned char *curA, *curB;  // wave-uniform bases of the issue tile's A rows / B rows
    auto set_issue_tile = [&](int ord) {
        int t = t_first + ord * t_step;
        const unsigned char *pa = baseA, *pb = baseB;
        if constexpr (BATCHED) {
            const int bz = t / tiles_1;
            t -= bz * tiles_1;
            pa += (size_t)bz * (size_t)a.batch_stride_a * 2;
            pb += (size_t)bz * (size_t)a.batch_stride_b * 2;
        }
        const int tm0 = (t / tiles_n) * BM, tn0 = (t % tiles_n) * BN;
        curA = pa + (size_t)tm0 * a.lda * 2;
        curB = pb + (size_t)tn0 * a.ldb * 2;
    };
    set_issue_tile(0);
    auto issue_piece = [&](int i) {
        // uniform 64-bit base + per-lane 32-bit offset (saddr form of global_load_lds)
        const unsigned char* ub = (i < NLA ? curA : curB) + (size_t)ikt * (BK * 2);
        unsigned off = i < NLA ? offA : offB;
        asm volatile("" : "+v"(off));  // keep ONE live offset register: stops hipcc from hoisting (and spilling)
                                       // eight precomputed 64-bit per-lane addresses
        off += i < NLA ? (unsigned)i * strideA : (unsigned)(i - NLA) * strideB;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ub + off),
            (__attribute__((address_space(3))) void*)(smem + islot * STAGE_BYTES + dst0 + i * 2 * PIECE), 16, 0, 0);
    };
    auto issue_piece_alt = [&](int p) {  // ALT: piece p of the stage at the issue cursor, this wave's quarter of it
        const unsigned char* ub = (p < PA ? curA : curB) + (size_t)ikt * (BK * 2);
        unsigned off = p < PA ? offA : offB;
        asm volatile("" : "+v"(off));
        off += p < PA ? (unsigned)p * strideA : (unsigned)(p - PA) * strideB;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ub + off),
            (__attribute__((address_space(3))) void*)(smem + islot * STAGE_BYTES + dst0 + p * PIECE), 16, 0, 0);
    };
    auto issue_advance = [&]() {
        if (++islot == R) islot = 0;
        if (++ikt == KT) {
            if (it + 1 < n_my) {  // the pipeline rolls straight into the next tile
                ++it;
                ikt = 0;
                set_issue_tile(it);
            } else {
                ikt = KT - 1;  // no further tile: harmless re-fetch of the last stage keeps vmcnt uniform
            }
        }
    };
    auto issue_stage = [&]() {  // pipeline start: every wave takes part
        if constexpr (ALT) {
#pragma unroll
            for (int p = 0; p < PA + PB; ++p)
                if ((p & 1) == team) issue_piece_alt(p);
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) issue_piece(i);
        }
        issue_advance();
    };

    floatx16 acc[TM][TN];
    half8 pend[NST];             // previous tile's outputs, packed, waiting to be stored
    _Float16* pend_ptr = a.C;    // per-lane address of the pending tile's (tm = 0, tn = 0, u = 0) store
    bool pend_valid = false;
    // output addressing: row-major (row stride ldc), or — c_block_rows != 0 — blocked by 64 columns:
    // element (m, n) at C[(n / 64) * c_block_rows * 64 + m * 64 + n % 64]  (the attention kernel's V^T layout)
    const long long ldc_eff = a.c_block_rows ? 64 : a.ldc;
    const long long ldc32 = ldc_eff * 32;

    // Fragments: the weight fragments wb are double-buffered over the k-steps; the activation fragments xa are
    // single-buffered and refilled "rolling": xa[tm] of the next k-step is read right after the MFMAs that
    // consume the current xa[tm] have been issued (6 MFMAs of cover before it is needed again).
    half8 xa[TM], wb[2][TN];
    auto read_wb = [&](int buf, const unsigned char* st, int j) {
        const unsigned char* sw = st + (PA + wn * TN) * PIECE + rd_off[j];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) wb[buf][tn] = *reinterpret_cast<const half8*>(sw + tn * PIECE);
    };
    auto read_xa = [&](int tm, const unsigned char* st, int j) {
        xa[tm] = *reinterpret_cast<const half8*>(st + (wm * TM + tm) * PIECE + rd_off[j]);
    };
    auto store_pending = [&](int q) {  // q = (tm * TN + tn) * 2 + u
        const int tm = q >> 2, tn = (q >> 1) & 1, u = q & 1;
        half8* p = reinterpret_cast<half8*>(pend_ptr + tm * ldc32 + tn * 32 + u * 16);
        if constexpr ((PST & 2) != 0)
            __builtin_nontemporal_store(pend[q], p);
        else
            *p = pend[q];
    };

    int cslot = 0;
    bool everyone_loaded = true;  // ALT: the stage in flight was issued by all eight waves (pipeline start only)
    // one pipeline stage of the consumed tile; SLOT >= 0: this stage also issues deferred stores 2*SLOT, 2*SLOT+1
    // (ALT: stage parity par = stage ordinal & 1; team == par loads the refill, the other team stores 4 * (SLOT / 2) ...)
    auto stage = [&](auto slot_c, bool first, int par) {
        constexpr int SLOT = decltype(slot_c)::value;
        const unsigned char* st = smem + cslot * STAGE_BYTES;
        if (++cslot == R) cslot = 0;
#pragma unroll
        for (int j = 0; j < KS - 1; ++j) {
            read_wb((j + 1) & 1, st, j + 1);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if (j == 0 && first) {  // first MFMA group of a tile starts from zero accumulators
                        floatx16 z;
#pragma unroll
                        for (int v = 0; v < 16; ++v) z[v] = 0.f;
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[0][tn], xa[tm], z, 0, 0, 0);
                    } else {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j & 1][tn], xa[tm], acc[tm][tn], 0, 0, 0);
                    }
                }
                read_xa(tm, st, j + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // last k-step.  Every read of this stage must have returned before its slot is handed back:
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(wb[1][tn]));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(xa[tm]));
        if constexpr (ALT) {
            // the refill in flight was issued by the team that loaded in the PREVIOUS stage (parity par ^ 1): only its waves
            // have LDS-DMA loads outstanding; the others may still have stores in flight and do not wait for them
            if (everyone_loaded || team != par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            everyone_loaded = false;
            asm volatile("s_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // next stage landed (ring 2)
        }
        const unsigned char* nst = smem + cslot * STAGE_BYTES;
        read_wb(0, nst, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[1][tn], xa[tm], acc[tm][tn], 0, 0, 0);
                if constexpr (ALT) {
                    if (team == par) {  // this team's turn to refill the slot everybody just left: two per MFMA gap
                        issue_piece_alt(2 * (tm * TN + tn));
                        issue_piece_alt(2 * (tm * TN + tn) + 1);
                    }
                } else {
                    issue_piece(tm * TN + tn);  // NL == NMF: one LDS-DMA instruction per MFMA gap
                }
            }
            read_xa(tm, nst, 0);
            if constexpr (SLOT >= 0 && (PST & 1) == 0) {
                if constexpr (ALT) {  // the other team: four of the previous tile's stores per stage
                    if (team != par && pend_valid) store_pending(4 * (SLOT >> 1) + tm);
                } else {
                    if (tm == 1 && pend_valid) store_pending(2 * SLOT);
                    if (tm == 3 && pend_valid) store_pending(2 * SLOT + 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_advance();
    };
    static_assert(NL == NMF, "issue schedule");

    // ---- pipeline start
#pragma unroll
    for (int p = 0; p < R - 1; ++p) issue_stage();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    issue_stage();
    read_wb(0, smem, 0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) read_xa(tm, smem, 0);

    for (int ti = 0; ti < n_my; ++ti) {
        int t = t_first + ti * t_step;
        _Float16* c_base = a.C;
        if constexpr (BATCHED) {
            const int bz = t / tiles_1;
            t -= bz * tiles_1;
            c_base += (size_t)bz * (size_t)a.batch_stride_c;
        }
        const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
        // ---- main loop: the first SLOTS stages are unrolled (static register indices for the deferred stores)
        {
            int kt = 0;
#define BH_ST(S)                                                          \
    if (kt < KT) {                                                        \
        stage(std::integral_constant<int, S>{}, S == 0, S & 1);           \
        ++kt;                                                             \
    }
            BH_ST(0) BH_ST(1) BH_ST(2) BH_ST(3) BH_ST(4) BH_ST(5) BH_ST(6) BH_ST(7)
#undef BH_ST
            for (; kt < KT; ++kt) stage(std::integral_constant<int, -1>{}, false, kt & 1);
            // K so short that some deferred stores found no stage: flush them now
            if (pend_valid && (PST & 1) == 0) {
#pragma unroll
                for (int s2 = 0; s2 < SLOTS; ++s2)
                    if (s2 >= KT) {
                        store_pending(2 * s2);
                        store_pending(2 * s2 + 1);
                    }
            }
        }
        // ---- epilogue math of this tile -> pend (stores are issued inside the next tile's main loop)
        if constexpr ((PST & 8) != 0) {  // ablation: no epilogue at all
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) asm volatile("" ::"v"(acc[tm][tn]));
        } else {
            _Float16* tile_ptr;  // per-lane address of this tile's (tm = 0, tn = 0, u = 0) store
            if constexpr ((EPI & BH_EPI_SWIGLU) != 0)  // C is [M][N / 2]: 16 folded columns per (tn), 8 per lane after the exchange below
                tile_ptr = c_base + (size_t)(m0 + wm * TM * 32 + ql) * a.ldc + ((n0 + wn * TN * 32) >> 1) + 8 * h;
            else if (a.c_block_rows)  // a wave's 64 output columns are exactly one 64-column block
                tile_ptr = c_base + (size_t)((n0 + wn * TN * 32) >> 6) * a.c_block_rows * 64 +
                           (size_t)(m0 + wm * TM * 32 + ql) * 64 + 8 * h;
            else
                tile_ptr = c_base + (size_t)(m0 + wm * TM * 32 + ql) * a.ldc + n0 + wn * TN * 32 + 8 * h;
            float bias_row[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                bias_row[tm] = 0.f;
                if constexpr ((EPI & BH_EPI_BIAS_ROW) != 0) bias_row[tm] = (float)a.bias[m0 + (wm * TM + tm) * 32 + ql];
            }
            if constexpr ((PST & 32) != 0 && (EPI & BH_EPI_SWIGLU) != 0) {
                // The gated fold through LDS (level 2 of option gemm_full_line_stores, the default since round 5): a wave's folded part of a tile
                // is 32 rows x 32 columns = 64-byte row pieces; straight from the registers a store instruction covers 32 rows with
                // 32 bytes each, through the wave's LDS buffer 16 rows with their whole 64 bytes.
                static_assert((PST & 32) == 0 || (PST & 1) != 0, "burst stores");
                unsigned char* stg = smem + R * STAGE_BYTES + wave * 4096;  // 16 windows of 128 bytes = 32 rows x 64 bytes
                typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                half8 bb[TN][2];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        bb[tn][u] = *reinterpret_cast<const half8*>(a.bias + n0 + (wn * TN + tn) * 32 + 8 * (2 * u + h));
                const int rrow = lane >> 2, rch = lane & 3;  // read-back: instruction i takes rows 16 i + rrow, 16-byte chunk rch of 4
                _Float16* gptr = c_base + (size_t)(m0 + wm * TM * 32 + rrow) * a.ldc + ((n0 + wn * TN * 32) >> 1) + rch * 8;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
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
                        half4 fold[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float g = c[8 * u + 2 * e] + bias_row[tm] + (float)bb[tn][u][2 * e];
                                const float up = c[8 * u + 2 * e + 1] + bias_row[tm] + (float)bb[tn][u][2 * e + 1];
                                fold[u][e] = (_Float16)(g / (1.0f + __builtin_amdgcn_exp2f(-g * 1.4426950408889634f)) * up);
                            }
                        uint2v lo = __builtin_bit_cast(uint2v, fold[0]), hi = __builtin_bit_cast(uint2v, fold[1]);
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            auto r = __builtin_amdgcn_permlane32_swap(lo[w], hi[w], false, false);
                            lo[w] = r[0];
                            hi[w] = r[1];
                        }
                        const int chunk = tn * 2 + h;  // the lane's 8 folded columns inside the row's 32
                        *reinterpret_cast<uintx4*>(stg + (ql >> 1) * 128 + ((((ql & 1) << 2) + (chunk ^ ((ql >> 1) & 3))) << 4)) =
                            uintx4{lo[0], lo[1], hi[0], hi[1]};
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = 16 * i + rrow;
                        const uintx4 v = *reinterpret_cast<const uintx4*>(stg + (row >> 1) * 128 + ((((row & 1) << 2) + (rch ^ ((row >> 1) & 3))) << 4));
                        uintx4* p = reinterpret_cast<uintx4*>(gptr + (size_t)(tm * 32 + 16 * i) * a.ldc);
                        if constexpr ((PST & 2) != 0)
                            __builtin_nontemporal_store(v, p);
                        else
                            *p = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else if constexpr ((PST & 32) != 0) {
                static_assert((PST & 32) == 0 || ((PST & 1) != 0 && (EPI & BH_EPI_SEGMAX) == 0), "full-line stores: burst, plain outputs");
                // FUSED LayerNorm (encoder.hip, option ln_fused; the two bits are independent):
                //   BH_EPI_LNA    the token operand of this GEMM is an UN-NORMALISED pre-LayerNorm tensor z and the weight operand was
                //                 folded with the LayerNorm's gain (W' = W o gamma): LN(z) W^T + b = rstd (z W'^T - mu c) + b', c = sum_k W'
                //                 (a.ln_c), b' = b + W beta (a.bias), (mu, rstd) per token (a.ln_stats: float2 per token).  Tokens are
                //                 the C rows (bias per column) or, for the transposed V projection, the C columns (bias per row).
                //   BH_EPI_RESLN  C = fp16(A B^T + bias) + R, R = the residual rows a.residual, normalised on the fly when a.res_stats
                //                 is set: (r - mu) rstd gamma + beta; the sum and the sum of squares of every output row over this
                //                 wave's 64 columns go to a.stats_out[row][N / 64 slots] (fixed slots, no atomics: bit-reproducible).
                //                 Done on the READ-BACK side of the LDS transposition: whole 128-byte lines of residual and output.
                constexpr bool LNA = (EPI & BH_EPI_LNA) != 0, RESLN = (EPI & BH_EPI_RESLN) != 0;
                constexpr bool TOK_COLS = LNA && (EPI & BH_EPI_BIAS_ROW) != 0;  // tokens along the columns (V^T projection)
                static_assert(!LNA || (EPI & (BH_EPI_BIAS_COL | BH_EPI_BIAS_ROW)) != 0, "the folded bias comes with the fold");
                static_assert(!RESLN || ((EPI & BH_EPI_BIAS_ROW) == 0 && (EPI & BH_EPI_GELU) == 0), "residual epilogue: row-major, bias per column");
                unsigned char* stg = smem + R * STAGE_BYTES + wave * 4096;  // this wave's 32 rows x 128 bytes
                // (the per-column vectors: hoisted out of the tile's loops in the plain epilogues — 16 registers —; the fused-LayerNorm
                // epilogues, which also hold statistics and residual rows, re-read theirs per 32 x 32 part from L1: 128 accumulator
                // registers of 256 leave no room for 32 more)
                constexpr bool HOIST = !LNA && !RESLN;
                half8 bb[TN][2];
                if constexpr ((EPI & BH_EPI_BIAS_COL) != 0 && HOIST) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            bb[tn][u] = *reinterpret_cast<const half8*>(a.bias + n0 + (wn * TN + tn) * 32 + 8 * (2 * u + h));
                }
                (void)bb;
                // (fused-LayerNorm epilogues: the read-back lane coordinates come from an OPAQUE copy of the lane id — otherwise hipcc
                // computes the per-lane residual / statistics addresses before the main loop and carries them across it in scratch)
                // (... and the lane id itself is re-derived with v_mbcnt: two instructions instead of a register kept — spilled — across the main loop)
                int lane_e = lane;
                if constexpr (!HOIST) {
                    lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    asm volatile("" : "+v"(lane_e));
                }
                const int rrow = lane_e >> 3, rch = lane_e & 7;  // read-back: instruction i takes rows 8 i + rrow, 16-byte chunk rch
                const int qlx = lane_e & 31, hx = lane_e >> 5;    // (= ql, h; opaque in the fused-LayerNorm epilogues, see above)
                // (blocked output — c_block_rows, the V^T layout; level 2 of the option —: a row's 64 columns are one
                // 128-byte line and consecutive rows are adjacent: 8 rows = 1 KiB contiguous per instruction)
                const long long ldrow = a.c_block_rows ? 64 : a.ldc;
                _Float16* gptr = a.c_block_rows
                                     ? c_base + (size_t)((n0 + wn * TN * 32) >> 6) * a.c_block_rows * 64 + (size_t)(m0 + wm * TM * 32 + rrow) * 64 + rch * 8
                                     : c_base + (size_t)(m0 + wm * TM * 32 + rrow) * a.ldc + n0 + wn * TN * 32 + rch * 8;
                // TOK_COLS: the statistics of this wave's 64 tokens (columns) sit one token per LANE (one coalesced 512-byte load per
                // tile); an element's (mean, rstd) comes out of its token's lane with v_readlane — the loads of 4 x 64 bytes per 8
                // columns that this replaces were L2 round trips in the innermost loop (measured: the V^T GEMM 38 % slower)
                float lane_mu = 0.f, lane_rstd = 1.f;
                if constexpr (TOK_COLS) {
                    const float2 st2 = reinterpret_cast<const float2*>(a.ln_stats)[n0 + wn * TN * 32 + lane_e];
                    lane_mu = st2.x;
                    lane_rstd = st2.y;
                }
                (void)lane_mu;
                (void)lane_rstd;
                // RESLN: the residual rows (and their statistics) of the NEXT 32-row part are requested while this part is converted,
                // transposed and stored: 4 row groups x (16-byte residual piece + float2) per lane in flight, double-buffered over tm
                half8 rres[2][4];
                auto res_prefetch = [&](int buf, int tm_) {
                    if constexpr (RESLN) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = m0 + wm * TM * 32 + tm_ * 32 + 8 * i + rrow;
                            rres[buf][i] = *reinterpret_cast<const half8*>(a.residual + (size_t)row * a.ldr + n0 + wn * TN * 32 + rch * 8);
                        }
                    }
                };
                res_prefetch(0, 0);
                (void)rres;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    float ln_mu = 0.f, ln_rstd = 1.f, ln_cm = 0.f;
                    if constexpr (LNA && !TOK_COLS) {
                        const float2 st2 = reinterpret_cast<const float2*>(a.ln_stats)[m0 + (wm * TM + tm) * 32 + qlx];
                        ln_mu = st2.x;
                        ln_rstd = st2.y;
                    }
                    if constexpr (TOK_COLS) ln_cm = (float)a.ln_c[m0 + (wm * TM + tm) * 32 + qlx];
                    (void)ln_mu;
                    (void)ln_rstd;
                    (void)ln_cm;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
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

                            half8 o, bcol, ccol;
                            if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) {
                                if constexpr (HOIST)
                                    bcol = bb[tn][u];
                                else
                                    bcol = *reinterpret_cast<const half8*>(a.bias + n0 + (wn * TN + tn) * 32 + 8 * (2 * u + hx));
                                if constexpr (LNA) ccol = *reinterpret_cast<const half8*>(a.ln_c + n0 + (wn * TN + tn) * 32 + 8 * (2 * u + hx));
                            }
                            (void)bcol;
                            (void)ccol;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float v = c[8 * u + e];
                                if constexpr (TOK_COLS) {
                                    // this lane's token: column tn*32 + 16 u + 8 hx + e of the wave's 64 = lane t0 (+ 8 for the upper half-lanes)
                                    const int t0 = tn * 32 + 16 * u + e;
                                    const float m_lo = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lane_mu), t0));
                                    const float m_hi = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lane_mu), t0 + 8));
                                    const float r_lo = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lane_rstd), t0));
                                    const float r_hi = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lane_rstd), t0 + 8));
                                    v = (hx ? r_hi : r_lo) * (v - (hx ? m_hi : m_lo) * ln_cm) + bias_row[tm];
                                }
                                else if constexpr (LNA)
                                    v = ln_rstd * (v - ln_mu * (float)ccol[e]) + (float)bcol[e];
                                else {
                                    v += bias_row[tm];
                                    if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) v += (float)bcol[e];
                                }
                                if constexpr ((EPI & BH_EPI_GELU) != 0) v = bh_gemm::gelu_erf(v);
                                o[e] = (_Float16)v;
                            }
                            const int chunk = tn * 4 + u * 2 + hx;  // the lane's 8 columns inside the row's 64
                            *reinterpret_cast<half8*>(stg + qlx * 128 + ((chunk ^ (qlx & 7)) << 4)) = o;
                        }
                    }
                    // (the next part's residual rows: requested here, where this part's 32 accumulator registers have just died)
                    if (tm + 1 < TM) res_prefetch((tm + 1) & 1, tm + 1);
                    float2 rstt[4];  // (the residual rows' statistics: a 270 KB array three tiles of a row strip share — L2 hits, requested together)
                    if constexpr (RESLN) {
                        if (a.res_stats) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) rstt[i] = reinterpret_cast<const float2*>(a.res_stats)[m0 + wm * TM * 32 + tm * 32 + 8 * i + rrow];
                        }
                    }
                    (void)rstt;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's own writes; no other wave touches stg)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        half8 v = *reinterpret_cast<const half8*>(stg + (8 * i + rrow) * 128 + ((rch ^ rrow) << 4));
                        half8* p = reinterpret_cast<half8*>(gptr + (size_t)(tm * 32 + 8 * i) * ldrow);
                        if constexpr (RESLN) {
                            const int row = m0 + wm * TM * 32 + tm * 32 + 8 * i + rrow;
                            const half8 r8 = rres[tm & 1][i];
                            float rr[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) rr[e] = (float)r8[e];
                            if (a.res_stats) {
                                // (gain and shift of the residual's LayerNorm for this lane's 8 read-back columns: re-read from L1 per
                                // row group rather than held in 8 registers across the tile)
                                const float2 rs = rstt[i];
                                const half8 rg = *reinterpret_cast<const half8*>(a.res_gamma + n0 + wn * TN * 32 + rch * 8);
                                const half8 rbeta = *reinterpret_cast<const half8*>(a.res_beta + n0 + wn * TN * 32 + rch * 8);
#pragma unroll
                                for (int e = 0; e < 8; ++e) rr[e] = (rr[e] - rs.x) * rs.y * (float)rg[e] + (float)rbeta[e];
                            }
                            float s1 = 0.f, s2 = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                v[e] = (_Float16)((float)v[e] + rr[e]);
                                const float zf = (float)v[e];  // the statistics describe what is STORED
                                s1 += zf;
                                s2 = fmaf(zf, zf, s2);
                            }
                            // the 8 lanes of a row (rch = 0 .. 7) are neighbours: two quad butterflies and one half-row mirror, all DPP
                            // (VALU; the ds_bpermute form of __shfl_xor cost ~3 us per tile in dependent LDS round trips)
                            s1 += dpp_f32<0xB1>(s1);   // quad_perm [1, 0, 3, 2]
                            s2 += dpp_f32<0xB1>(s2);
                            s1 += dpp_f32<0x4E>(s1);   // quad_perm [2, 3, 0, 1]
                            s2 += dpp_f32<0x4E>(s2);
                            s1 += dpp_f32<0x141>(s1);  // row_half_mirror: lane i <-> 7 - i of its 8
                            s2 += dpp_f32<0x141>(s2);
                            if (rch == 0)
                                reinterpret_cast<float2*>(a.stats_out)[(size_t)row * (a.N >> 6) + ((n0 + wn * TN * 32) >> 6)] = make_float2(s1, s2);
                        }
                        if constexpr ((PST & 2) != 0)
                            __builtin_nontemporal_store(v, p);
                        else
                            *p = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads have landed before the next part overwrites stg)
                }
            } else
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                half8 b8[2];
                if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        b8[u] = *reinterpret_cast<const half8*>(a.bias + n0 + (wn * TN + tn) * 32 + 8 * (2 * u + h));
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
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
                    half4 fold[2];  // (BH_EPI_SWIGLU) this lane's 4 folded outputs of u = 0 and of u = 1
                    (void)fold;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        half8 o;
                        float vv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = c[8 * u + e] + bias_row[tm];
                            if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) v += (float)b8[u][e];
                            if constexpr ((EPI & BH_EPI_GELU) != 0) v = bh_gemm::gelu_erf(v);
                            o[e] = (_Float16)v;
                            vv[e] = v;
                        }
                        (void)vv;
                        if constexpr ((EPI & BH_EPI_SWIGLU) != 0) {
                            // gated feed-forward: (vv[2 j], vv[2 j + 1]) = (gate, up) of output column j of this lane's four;
                            // folded here, exchanged and stored below the u loop
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float g = vv[2 * e];
                                fold[u][e] = (_Float16)(g / (1.0f + __builtin_amdgcn_exp2f(-g * 1.4426950408889634f)) * vv[2 * e + 1]);
                            }
                        } else if constexpr ((EPI & BH_EPI_SEGMAX) != 0) {
                            // SPLADE head (C rows = vocabulary terms, C columns = packed tokens): the lane's 8 values
                            // are 8 consecutive tokens of ONE sequence (sequences start at multiples of 8 rows), so
                            // the max over tokens is taken in registers; seg_grp[token / 8] = sequence << 4 | valid
                            // tokens in the group (0 = gap rows only).  max commutes with the monotone
                            // log(1 + relu(.)), applied once per (sequence, term) by bh_splade_finish_kernel.
                            const int g = a.seg_grp[(n0 + (wn * TN + tn) * 32 + u * 16 + 8 * h) >> 3];
                            const int cnt = g & 15;
                            float mx = 0.f;  // relu
#pragma unroll
                            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, e < cnt ? vv[e] : 0.f);
                            if (cnt > 0 && mx > 0.f) {
                                // non-negative floats order like their bit patterns; read first: once a few tokens of
                                // the sequence have been seen almost nothing beats the running maximum
                                unsigned* sp = a.seg_out + (size_t)(g >> 4) * a.ld_seg + m0 + (wm * TM + tm) * 32 + ql;
                                const unsigned bits = __float_as_uint(mx);
                                if (bits > __builtin_nontemporal_load(sp)) atomicMax(sp, bits);
                            }
                        } else if constexpr ((PST & 4) != 0) {  // ablation: epilogue math without the stores
                            asm volatile("" ::"v"(o));
                        } else if constexpr ((PST & 1) != 0) {  // burst: store at once, nothing stays live
                            half8* p = reinterpret_cast<half8*>(tile_ptr + tm * ldc32 + tn * 32 + u * 16);
                            if constexpr ((PST & 2) != 0)
                                __builtin_nontemporal_store(o, p);
                            else
                                *p = o;
                        } else {
                            pend[(tm * TN + tn) * 2 + u] = o;
                        }
                    }
                    if constexpr ((EPI & BH_EPI_SWIGLU) != 0) {
                        // A lane holds output columns 4 h + 0..3 (u = 0) and 8 + 4 h + 0..3 (u = 1) of this 16-column group; one
                        // half-lane exchange (the lower half's u = 1 part against the upper half's u = 0 part) leaves 8 consecutive
                        // columns per lane — 8 h + 0..7 — for ONE 16-byte store (8-byte stores, a row's 32 bytes in four pieces from
                        // two instructions, measured 12 % slower than not fusing at all).
                        static_assert((EPI & BH_EPI_SWIGLU) == 0 || (PST & 1) != 0, "the gated fold stores at once (burst)");
                        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                        typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                        uint2v lo = __builtin_bit_cast(uint2v, fold[0]), hi = __builtin_bit_cast(uint2v, fold[1]);
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            auto r = __builtin_amdgcn_permlane32_swap(lo[w], hi[w], false, false);
                            lo[w] = r[0];
                            hi[w] = r[1];
                        }
                        uintx4 o8 = {lo[0], lo[1], hi[0], hi[1]};
                        uintx4* p = reinterpret_cast<uintx4*>(tile_ptr + tm * ldc32 + tn * 16);
                        if constexpr ((PST & 2) != 0)
                            __builtin_nontemporal_store(o8, p);
                        else
                            *p = o8;
                    }
                }
            }
            if constexpr ((PST & 1) == 0) {
                pend_ptr = tile_ptr;
                pend_valid = true;
            }
        }
    }
    // ---- drain: the last tile's outputs
    if constexpr ((PST & 1) == 0) {
#pragma unroll
        for (int q = 0; q < NST; ++q) store_pending(q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, int PST = 0>
hipError_t bh_gemm_launch_persist(const BhGemmArgs& a, int n_cu, hipStream_t stream) {
    constexpr size_t smem = 2 * 16 * 4096 + ((PST & 32) != 0 ? 8 * 4096 : 0);  // ring 2 x (8 + 8) pieces (+ 4 KiB per wave: full-line stores)
    auto kern = bh_gemm_f16_pkernel<EPI, PST>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = (a.M / 256) * (a.N / 256) * ((EPI & BH_EPI_BATCHED) != 0 ? a.batch : 1);
    if (tiles <= 0) return hipSuccess;
    int grid = n_cu / 8 * 8;  // one resident workgroup per CU, a multiple of the XCD count
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, a);
    return hipGetLastError();
}

hipError_t bh_gemm_persist(const BhGemmArgs& a, int epi, int pst, hipStream_t s);  // gemm_f16_c.hip
