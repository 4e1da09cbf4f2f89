// scan_topk.hip — fused inner-product + running top-k scan over a resident fp16 corpus.
//
// Replaces, in one kernel, the reference's per-chunk  torch.mm  (models/retrievers/dense.py:81)
// + torch.topk (modules/retrieve.py:157): the [Bq, n] score matrix is never materialised.
//
// Roofline: HBM.  One launch reads every corpus row exactly once (N*d*2 bytes) for a tile of
// BQ = 128*QW queries; arithmetic intensity = BQ flop/byte, below the gfx950 ridge (~310).
//
// Work decomposition (gfx950: 256 CUs, wave64, 160 KiB LDS, 512 regs/lane at 1 wave/SIMD)
//   * persistent grid, one 256-thread workgroup (4 waves, one per SIMD) per CU;
//     workgroup b walks row tiles t = b, b+G, b+2G, ... (32 rows each, ascending).
//   * QUERIES LIVE IN REGISTERS: wave w keeps queries [32w*QW, 32w*QW+32*QW) of the tile as
//     MFMA B-fragments, pinned in the ACCUMULATOR half of the register file (AGPRs; MFMA reads
//     B straight from them) for the whole launch — d=768: 192 regs/lane per 32 queries.  The
//     register file is the largest on-chip store (512 KiB/CU) and this costs zero re-reads.
//   * CORPUS STREAMS THROUGH LDS by LDS-DMA (`global_load_lds_dwordx4`, no VGPR round trip)
//     into an R-deep ring of stages; a stage = 32 rows x LS 128-byte lines.  All 4 waves read
//     every stage (each against its own queries) with conflict-free ds_read_b128, software
//     pipelined one line (4 fragments) ahead of the MFMAs.
//   * v_mfma_f32_32x32x16_f16, A = 32 corpus rows, B = 32 queries, fp32 accumulate: lane l ends
//     up with 16 scores of ONE query (l & 31)  ->  one threshold register per lane.
//   * top-k: per-query threshold compare (15 v_max + 1 v_cmp per tile); survivors are
//     appended to a per-(workgroup, query) candidate buffer in global memory (L2-resident),
//     slot counters live in registers (both half-lanes of a query keep identical copies);
//     when a buffer nears capacity the owning wave bitonic-sorts it, keeps the best KP and
//     raises the threshold.
//   * thresholds are SHARED between workgroups through a 64-slot table per query: workgroup b
//     atomicMax-es the (KP/64)-th best score it has appended into slot b % 64; the minimum over
//     the 64 slots is a score that at least 64 * (KP/64) = KP distinct rows reach (distinct
//     slots <-> distinct workgroups), hence a valid lower bound of the final KP-th best.  A wave
//     re-reads its queries' slots only after it had hits.  The table is a pure filter hint: a
//     stale value only means less filtering, never a wrong result.
//
// LDS image / bank conflicts.  One LDS-DMA instruction writes 1 KiB lane-linearly
// (dest = base + lane*16).  Lanes 8j..8j+7 fetch the eight 16-byte chunks of ONE 128-byte
// line of row 8*rg+j (full-line coalescing), with the chunk order XOR-permuted by
// g(row) = ((row>>1)&1) | ((row>>3)<<1).  The MFMA A-fragment read (lane l: row l&31, chunk
// 2*j4 + (l>>5)) then hits 16 distinct 16-byte slots in each of ds_read_b128's four 16-lane
// service groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) => conflict-free.
//
// Exactness.  The scan ranks by the fp32 MFMA score and keeps KP >= k+8 candidates per
// workgroup; merge_rescore.hip re-scores the merged best KP in sequential fp64 (the
// canonical score) and cuts to k.  The threshold logic can only drop a row that at least KP
// better rows (score desc, row asc) beat, so the merged top-KP is independent of timing.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {

template <int EPL>
__device__ __forceinline__ void load_list(u64 (&e)[EPL], const u64* buf, unsigned n, int lane) {
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const unsigned idx = r * 64 + lane;
        e[r] = idx < n ? buf[idx] : 0ull;
    }
}

// Sort one query's candidate buffer (n <= 2*KP entries) with the whole wave; on return
// e[0 .. KP/64) hold the best KP keys, sorted descending (element i = r*64 + lane).
template <int KP>
__device__ __forceinline__ void sort_candidates(u64 (&e)[2 * KP / 64], const u64* buf, unsigned n, int lane) {
    load_list<2 * KP / 64>(e, buf, n, lane);
    bh_wave_sort_desc<2 * KP / 64>(e, lane);
}

}  // namespace

// NK = padded dim / 16 (MFMA k-steps per row); KP = candidate list length (64|128|256);
// LS = 128-byte lines per stage; R = ring depth in stages; QW = 32-query blocks per wave;
// NT = non-temporal cache policy on the corpus stream.
// ABL (ablation, bench-only, 0 in production): 1 = no threshold epilogue, 2 = stream only (no LDS
// reads, no MFMA), 3 = LDS reads without MFMA, 4 = MFMA without LDS reads.
// ABL = 5 is not an ablation but the FILTER PASS of the exactness fall-back (certify.hip, index.hip): the same stream and
// MFMA loop against the queries the certificate could not prove, with a FIXED per-query threshold (their k-th canonical
// score minus the MFMA error bound) instead of the running top-k: every row whose MFMA score reaches it is appended to
// the query's list a.fix_rows (few rows: the top k and whatever lies within rounding error of the k-th score); no
// candidate buffers, no bounds exchange, no final sort.
template <int NK, int KP, int LS, int R, int QW, bool NT, int ABL = 0, bool IL = true>
__global__ void __launch_bounds__(256, 1) bh_scan_topk_kernel(BhScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NK * 16;
    constexpr int LINES = D / 64;
    static_assert(LINES % LS == 0, "stage must divide the row");
    constexpr int S = LINES / LS;  // stages per 32-row tile
    constexpr int STAGE_BYTES = 32 * LS * 128;
    constexpr int CAP = 2 * KP;
    constexpr int EPLC = CAP / 64;
    constexpr int EPLK = KP / 64;
    constexpr int BQ = 128 * QW;
    constexpr int ROW_BYTES = D * 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Query split (a.qsplit = 2): workgroups xb and xb ^ 8 — dispatched to the SAME XCD (workgroup id mod 8) —
    // walk the same row tiles b, b+G, ... in step, each against its own BQ queries, so one of them takes every
    // corpus line from HBM and the other finds it in that XCD's L2: HBM traffic per query halves while the
    // per-workgroup register budget (the query fragments) stays that of a BQ-query tile.
    const int QS = a.qsplit;
    const int xb = blockIdx.x;
    const int G = QS == 1 ? (int)gridDim.x : (int)gridDim.x / QS;         // row-tile walkers
    const int qh = QS == 1 ? 0 : (xb >> 3) % QS;                          // which BQ queries of the launch
    const int b = QS == 1 ? xb : (((xb >> 3) / QS) << 3) | (xb & 7);      // row-tile walker id
    const int ql = lane & 31, h = lane >> 5;

    const long long my_tiles = (a.n_tiles > b) ? (a.n_tiles - b + G - 1) / G : 0;
    u64* cand_wg = a.cand + (size_t)xb * BQ * CAP;
    u64* part_wg = a.partial + ((size_t)b * QS + qh) * BQ * KP;
    a.qtile += (size_t)qh * BQ * D;
    a.gthr += (size_t)qh * BQ * 64;

    // ---- queries -> registers (B fragments).  Lane (ql, h) holds, for k-step s, the 8 halfs
    // at k = 16 s + 8 h of query  (wave*QW + w2)*32 + ql.  All loads first, then pin to AGPRs.
    half8 qf[QW][NK];
#pragma unroll
    for (int w2 = 0; w2 < QW; ++w2) {
        const _Float16* qrow = a.qtile + (size_t)((wave * QW + w2) * 32 + ql) * D;
#pragma unroll
        for (int s = 0; s < NK; ++s) qf[w2][s] = *reinterpret_cast<const half8*>(qrow + (2 * s + h) * 8);
    }
    // Pin fragments into the accumulator file, but never more than it can hold next to the MFMA
    // accumulators (256 AGPRs - 16*QW): over-subscribing the "a" constraint makes hipcc (ROCm 7.2)
    // split-spill AGPR tuples and mis-reload them (observed: one dword of a fragment left stale).
    constexpr int PIN_MAX = (256 - 16 * QW - 16) / 4;
#pragma unroll
    for (int w2 = 0; w2 < QW; ++w2)
#pragma unroll
        for (int s = 0; s < NK; ++s)
            if (w2 * NK + s < PIN_MAX) asm volatile("" : "+a"(qf[w2][s]));

    constexpr int RB = KP / 64;  // a workgroup publishes its RB-th best appended score
    float thr[QW];       // candidate iff score > thr   (per lane = per query)
    unsigned cnt[QW];    // entries in the query's candidate buffer (identical in both half-lanes)
    float best[QW][RB];  // this half-lane's RB best appended scores, descending
    float pub[QW];       // last value published to the slot table
    long long next_poll = 0;  // tile ordinal of the next slot-table exchange (same in every wave)
#pragma unroll
    for (int w2 = 0; w2 < QW; ++w2) {
        thr[w2] = -__builtin_inff();
        if constexpr (ABL == 5) thr[w2] = a.fix_thr[(wave * QW + w2) * 32 + ql];  // (+inf for the tile's unused queries)
        cnt[w2] = 0;
        pub[w2] = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < RB; ++r) best[w2][r] = -__builtin_inff();
    }

    // ---- per-lane constants of the LDS-DMA source pattern and of the fragment reads
    const int ld_row = 8 * wave + (lane >> 3);         // row inside the tile this lane fetches
    const int ld_g = ((lane >> 4) & 1) | (wave << 1);  // g(ld_row)
    const unsigned ld_off = (unsigned)ld_row * ROW_BYTES + (unsigned)(((lane & 7) ^ ld_g) << 4);
    const int rd_g = ((ql >> 1) & 1) | ((ql >> 3) << 1);  // g(row = ql)
    unsigned rd_off[4];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
        rd_off[j4] = (unsigned)((ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * j4 + h) ^ rd_g) << 4));

    // Compact query qq (of block w2) of this wave: sort, keep best KP, raise its threshold.
    auto compact = [&](int w2, int qq) {
        const int qi = (wave * QW + w2) * 32 + qq;
        const unsigned n = __builtin_amdgcn_readlane(cnt[w2], qq);
        u64* buf = cand_wg + (size_t)qi * CAP;
        u64 e[EPLC];
        sort_candidates<KP>(e, buf, n, lane);
#pragma unroll
        for (int r = 0; r < EPLK; ++r) buf[r * 64 + lane] = e[r];
        if (ql == qq) cnt[w2] = n < (unsigned)KP ? n : (unsigned)KP;
        const u64 kth = bh_shfl64(e[EPLK - 1], 63);
        if (kth != 0ull) {
            // Rows arrive in ascending order inside a workgroup, so a later row that merely
            // TIES the KP-th best loses on row index: the exclusive compare is exact.
            const float nt = bh_key_score(kth);
            if (ql == qq) thr[w2] = fmaxf(thr[w2], nt);
        }
    };

    if (my_tiles > 0) {
        const unsigned char* corpus = reinterpret_cast<const unsigned char*>(a.corpus);
        // issue cursor: (tile ordinal it, part ip) of the next stage to fetch, and its ring slot
        long long it = 0;
        int ip = 0;
        int islot = 0;
        // one LDS-DMA instruction: line j of the stage at the issue cursor
        auto issue_line = [&](int j) {
            const long long itc = it < my_tiles ? it : my_tiles - 1;  // past the end: harmless re-fetch,
            const long long tile = b + itc * G;                       // keeps the vmcnt arithmetic uniform
            const unsigned char* src = corpus + (size_t)tile * 32 * ROW_BYTES + (size_t)ip * LS * 128 + ld_off;
            unsigned char* dst = smem + islot * STAGE_BYTES + wave * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 128),
                                             (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0,
                                             NT ? 2 : 0);
        };
        auto advance_cursor = [&]() {
            if (++ip == S) { ip = 0; ++it; }
            if (++islot == R) islot = 0;
        };
        auto issue_stage = [&]() {
#pragma unroll
            for (int j = 0; j < LS; ++j) issue_line(j);
            advance_cursor();
        };
#pragma unroll
        for (int p = 0; p < R - 1; ++p) issue_stage();

        int cslot = 0;  // ring slot of the stage being consumed
        bool paced = QS > 1 && a.pair_window > 0;
        for (long long i = 0; i < my_tiles; ++i) {
            // Pairing only pays while the partners stay within the L2 residency of a line (a few tiles): wave 0
            // publishes the tile it starts and holds the workgroup (the others wait at the stage barrier) while
            // it is more than pair_window tiles ahead of its partner.  The partner's counter is read with a
            // SCALAR load (glc: from the XCD's L2, which both partners share) — a vector load would queue behind
            // the ring's LDS-DMA requests, which return in order.  Pure pacing hint: a stale or missing value can
            // only change timing; after one timeout the workgroup stops pacing for the rest of the launch.
            if (paced && wave == 0) {
                if (lane == 0) *(volatile unsigned*)(a.progress + xb) = (unsigned)i + 1u;
                const unsigned* pp = a.progress + (xb ^ 8);
                int spins = 0;
                for (;;) {
                    unsigned pv;
                    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(pv) : "s"(pp) : "memory");
                    if ((int)((unsigned)i + 1u - pv) <= a.pair_window) break;
                    if (++spins > 2048) {
                        paced = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            floatx16 acc[QW];
#pragma unroll
            for (int w2 = 0; w2 < QW; ++w2)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[w2][v] = 0.f;

#pragma unroll
            for (int part = 0; part < S; ++part) {
                // Own LDS-DMA of this stage landed (R-2 younger stages may stay in flight), then
                // rendezvous: everybody's pieces landed AND everybody is done reading the slot
                // that the next issue overwrites.
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * LS) : "memory");
                // The refill of the slot everybody just left: either all LS instructions here (the matrix pipe is idle
                // while they issue, ~100 cycles each under load), or one after every line's MFMAs (dma_interleave).
                constexpr bool spread = (ABL == 0 || ABL == 5) && IL;
                if (!spread) issue_stage();
                const unsigned char* st = smem + cslot * STAGE_BYTES;
                // fragment reads run one 128-byte line (4 k-steps) ahead of the MFMAs
                half8 af[2][4];
                if constexpr (ABL == 2) {
                    // stream only
                } else if constexpr (ABL == 4) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) af[0][j4] = af[1][j4] = qf[0][j4];
#pragma unroll
                    for (int jl = 0; jl < LS; ++jl)
#pragma unroll
                        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                            for (int w2 = 0; w2 < QW; ++w2)
                                acc[w2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    af[jl & 1][j4], qf[w2][(part * LS + jl) * 4 + j4], acc[w2], 0, 0, 0);
                } else {
                    // GL lines (4*GL fragments) per group; reads of group g+1 are issued after the
                    // wait for group g and ahead of group g's MFMAs (>= 128*GL cycles of cover)
                    constexpr int GL = (LS % 2 == 0) ? 2 : 1;
                    constexpr int NG = LS / GL;
                    half8 ag[2][4 * GL];
#pragma unroll
                    for (int f = 0; f < 4 * GL; ++f)
                        ag[0][f] = *reinterpret_cast<const half8*>(st + (f >> 2) * 4096 + rd_off[f & 3]);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        __builtin_amdgcn_sched_barrier(0);
                        // fake use: makes hipcc wait for THIS group's fragments here (it waits
                        // lgkmcnt(0), which would otherwise also cover the reads issued next)
#pragma unroll
                        for (int f = 0; f < 4 * GL; ++f) asm volatile("" : "+v"(ag[g & 1][f]));
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 1 < NG) {
#pragma unroll
                            for (int f = 0; f < 4 * GL; ++f)
                                ag[(g + 1) & 1][f] = *reinterpret_cast<const half8*>(
                                    st + ((g + 1) * GL + (f >> 2)) * 4096 + rd_off[f & 3]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (ABL == 3) {
#pragma unroll
                            for (int f = 0; f < 4 * GL; ++f) asm volatile("" ::"v"(ag[g & 1][f]));
                        } else {
#pragma unroll
                            for (int f = 0; f < 4 * GL; ++f) {
#pragma unroll
                                for (int w2 = 0; w2 < QW; ++w2)
                                    acc[w2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                        ag[g & 1][f], qf[w2][(part * LS + g * GL) * 4 + f], acc[w2], 0, 0, 0);
                                if ((f & 3) == 3) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (spread) issue_line(g * GL + (f >> 2));
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                    }
                    if (spread) advance_cursor();
                }
                __builtin_amdgcn_sched_barrier(0);
                if (++cslot == R) cslot = 0;
            }

            // ---- epilogue: threshold filter ------------------------------------------------
            const long long row0 = (b + i * G) * 32;
#pragma unroll
            for (int w2 = 0; w2 < QW; ++w2) {
                float m = acc[w2][0];
#pragma unroll
                for (int v = 1; v < 16; ++v) m = fmaxf(m, acc[w2][v]);
                if constexpr (ABL == 5) {
                    if (__builtin_amdgcn_ballot_w64(m >= thr[w2]) != 0ull) {
                        const int qi = (wave * QW + w2) * 32 + ql;
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const long long row = row0 + (v & 3) + 8 * (v >> 2) + 4 * h;
                            if (acc[w2][v] >= thr[w2] && row < a.n_rows) {
                                const unsigned slot = atomicAdd(a.fix_cnt + qi, 1u);
                                if (slot < a.fix_cap) a.fix_rows[(size_t)qi * a.fix_cap + slot] = (unsigned)row;
                            }
                        }
                    }
                    continue;
                }
                if constexpr (ABL != 0) {
                    asm volatile("" ::"v"(m));
                    m = -__builtin_inff();
                }
                if (__builtin_amdgcn_ballot_w64(m > thr[w2]) != 0ull) {
                    // (1) make room: a tile adds at most 32 entries per query
                    u64 need = __builtin_amdgcn_ballot_w64(cnt[w2] > (unsigned)(CAP - 32)) & 0xffffffffull;
                    while (need) {
                        const int qq = __builtin_ctzll(need);
                        need &= need - 1;
                        compact(w2, qq);
                    }
                    // (2) append survivors; slot = cnt + (hits of the same query in lower lanes)
                    u64* buf = cand_wg + (size_t)((wave * QW + w2) * 32 + ql) * CAP;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const long long row = row0 + (v & 3) + 8 * (v >> 2) + 4 * h;
                        const bool hit = (acc[w2][v] > thr[w2]) && (row < a.n_rows);
                        const u64 hm = __builtin_amdgcn_ballot_w64(hit);
                        if (hm != 0ull) {
                            const unsigned hl = ((unsigned)hm >> ql) & 1u;
                            const unsigned hh = ((unsigned)(hm >> 32) >> ql) & 1u;
                            if (hit) {
                                buf[cnt[w2] + (h ? hl : 0u)] = bh_make_key(acc[w2][v], (unsigned)row);
                                float x = acc[w2][v];  // insert into this half-lane's RB best
#pragma unroll
                                for (int r = 0; r < RB; ++r) {
                                    const float hi = fmaxf(best[w2][r], x);
                                    x = fminf(best[w2][r], x);
                                    best[w2][r] = hi;
                                }
                            }
                            cnt[w2] += hl + hh;
                        }
                    }
                }
            }
            // ---- threshold exchange through the slot table (filter hint only) ---------------
            // Exchanges run on a geometric schedule of the tile ordinal (0,1,2,4,7,11,17,...): the
            // bound tightens like 1/i, so this keeps it within ~1.5x of what continuous polling
            // would give at ~20 exchanges per launch; every wave of the workgroup exchanges at the
            // same tiles so the L2 round trips overlap instead of adding up across waves.
            if (a.share && i >= next_poll) {
                next_poll = i + 1 + (i >> 1);
#pragma unroll
                for (int w2 = 0; w2 < QW; ++w2) {
                    // publish: this half-lane's RB-th best appended score -> slot b % 64
                    const int q = (wave * QW + w2) * 32 + ql;
                    const float mine = best[w2][RB - 1];
                    if (mine > pub[w2]) {
                        pub[w2] = mine;
                        __hip_atomic_fetch_max(a.gthr + (size_t)q * 64 + (b & 63), bh_ordf(mine), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int w2 = 0; w2 < QW; ++w2) {
                    // poll: 16 lanes x 4 slots cover one query; 4 queries per load instruction
                    uint4 sl[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int q = (wave * QW + w2) * 32 + it * 4 + (lane >> 4);
                        const unsigned* src = a.gthr + (size_t)q * 64 + (lane & 15) * 4;
                        sl[it].x = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sl[it].y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sl[it].z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sl[it].w = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        unsigned mn = min(min(sl[it].x, sl[it].y), min(sl[it].z, sl[it].w));
#pragma unroll
                        for (int o = 8; o >= 1; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64));
                        // lanes 16g..16g+15 now hold the minimum of query it*4+g
                        const unsigned mine = (unsigned)__shfl((int)mn, (ql & 3) * 16, 64);
                        // a row that TIES the bound may still win on row index: inclusive compare,
                        // i.e. exclusive against the next lower float
                        if ((ql >> 2) == it && mine > BH_ORD_NEG_INF) thr[w2] = fmaxf(thr[w2], bh_unordf(mine - 1u));
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail re-fetches
    }

    if constexpr (ABL == 5) return;
    // ---- final: every wave sorts its queries' buffers and publishes the best KP -------------
#pragma unroll
    for (int w2 = 0; w2 < QW; ++w2) {
        for (int qq = 0; qq < 32; ++qq) {
            const int qi = (wave * QW + w2) * 32 + qq;
            const unsigned n = __builtin_amdgcn_readlane(cnt[w2], qq);
            u64 e[EPLC];
            sort_candidates<KP>(e, cand_wg + (size_t)qi * CAP, n, lane);
#pragma unroll
            for (int r = 0; r < EPLK; ++r) part_wg[(size_t)qi * KP + r * 64 + lane] = e[r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch

template <int NK, int KP, int LS, int R, int QW, bool NT, int ABL = 0, bool IL = true>
static hipError_t launch_one(const BhScanArgs& a, int grid, hipStream_t stream) {
    constexpr size_t smem = (size_t)R * 32 * LS * 128;
    static bool attr_done = false;
    auto kern = bh_scan_topk_kernel<NK, KP, LS, R, QW, NT, ABL, IL>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <int NK, int LS, int R, int QW>
static hipError_t launch_kp(const BhScanArgs& a, int kp, int grid, hipStream_t stream) {
    const bool nt = a.nontemporal != 0;
    if constexpr (NK == 48 && QW == 1 && LS == 6 && R == 6) {
        if (a.ablate != 0 && kp == 64) {  // bench-only ablations of the d=768 kernel
            switch (a.ablate) {
                case 1: return launch_one<NK, 64, LS, R, QW, true, 1>(a, grid, stream);
                case 2: return launch_one<NK, 64, LS, R, QW, true, 2>(a, grid, stream);
                case 3: return launch_one<NK, 64, LS, R, QW, true, 3>(a, grid, stream);
                case 4: return launch_one<NK, 64, LS, R, QW, true, 4>(a, grid, stream);
            }
        }
    }
    if constexpr (NK == 48 && QW == 1 && LS == 6 && R == 6) {
        // (bench option "dma_interleave" 0: the refill issued in one block after the barrier, d = 768 / top-50 only)
        if (a.dma_interleave == 0 && kp == 64)
            return nt ? launch_one<NK, 64, LS, R, QW, true, 0, false>(a, grid, stream)
                      : launch_one<NK, 64, LS, R, QW, false, 0, false>(a, grid, stream);
    }
    switch (kp) {
        case 64:
            return nt ? launch_one<NK, 64, LS, R, QW, true>(a, grid, stream)
                      : launch_one<NK, 64, LS, R, QW, false>(a, grid, stream);
        case 128:
            return nt ? launch_one<NK, 128, LS, R, QW, true>(a, grid, stream)
                      : launch_one<NK, 128, LS, R, QW, false>(a, grid, stream);
        case 256:
            if constexpr (QW == 1)
                return nt ? launch_one<NK, 256, LS, R, QW, true>(a, grid, stream)
                          : launch_one<NK, 256, LS, R, QW, false>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}

template <int NK, int LS, int R>
static hipError_t launch_qw(const BhScanArgs& a, int kp, int qw, int grid, hipStream_t stream) {
    if (qw == 1) return launch_kp<NK, LS, R, 1>(a, kp, grid, stream);
    if constexpr (NK == 24) {
        if (qw == 2) return launch_kp<NK, LS, R, 2>(a, kp, grid, stream);
    }
    return hipErrorInvalidValue;
}

// dim_padded in {64,128,256,384,512,768,1024}
hipError_t bh_launch_scan(const BhScanArgs& a, int dim_padded, int kp, int qw, int grid, hipStream_t stream) {
    switch (dim_padded) {
        case 64: return launch_qw<4, 1, 6>(a, kp, qw, grid, stream);
        case 128: return launch_qw<8, 2, 6>(a, kp, qw, grid, stream);
        case 256: return launch_qw<16, 4, 6>(a, kp, qw, grid, stream);
        case 384: return launch_qw<24, 6, 6>(a, kp, qw, grid, stream);
        case 512: return launch_qw<32, 8, 4>(a, kp, qw, grid, stream);
        case 768:
            // ring geometry variants (bench option "ring_variant"): lines per stage x ring depth
            if (a.ring_variant == 1 && kp == 64) return launch_kp<48, 12, 3, 1>(a, kp, grid, stream);
            if (a.ring_variant == 2 && kp == 64) return launch_kp<48, 4, 9, 1>(a, kp, grid, stream);
            if (a.ring_variant == 3 && kp == 64) return launch_kp<48, 3, 12, 1>(a, kp, grid, stream);
            if (a.ring_variant == 4 && kp == 64) return launch_kp<48, 6, 5, 1>(a, kp, grid, stream);
            return launch_qw<48, 6, 6>(a, kp, qw, grid, stream);
        case 1024: return launch_qw<64, 8, 4>(a, kp, qw, grid, stream);
    }
    return hipErrorInvalidValue;
}

// The fall-back's filter pass (ABL = 5 above): 128 queries per launch, every padded dim.
hipError_t bh_launch_filter_scan(const BhScanArgs& a, int dim_padded, int grid, hipStream_t stream) {
    switch (dim_padded) {
        case 64: return launch_one<4, 64, 1, 6, 1, true, 5>(a, grid, stream);
        case 128: return launch_one<8, 64, 2, 6, 1, true, 5>(a, grid, stream);
        case 256: return launch_one<16, 64, 4, 6, 1, true, 5>(a, grid, stream);
        case 384: return launch_one<24, 64, 6, 6, 1, true, 5>(a, grid, stream);
        case 512: return launch_one<32, 64, 8, 4, 1, true, 5>(a, grid, stream);
        case 768: return launch_one<48, 64, 6, 6, 1, true, 5>(a, grid, stream);
        case 1024: return launch_one<64, 64, 8, 4, 1, true, 5>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}

// Query-tile widths: 128 everywhere; 256 (two 32-query blocks per wave) only where both blocks'
// fragments fit the register file without spilling: d = 384 with candidate lists up to 128.
// (d = 768 needs 384 + 32 + 64 registers before any state: hipcc spills, and its reload of
// split AGPR tuples was observed to be wrong — not shipped.)
bool bh_scan_supports(int dim_padded, int kp, int qw) {
    if (qw == 1) return true;
    return qw == 2 && dim_padded == 384 && kp <= 128;
}
