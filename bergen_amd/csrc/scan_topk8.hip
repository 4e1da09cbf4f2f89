// scan_topk8.hip — second-generation fused inner-product + running top-k scan: 8 waves per CU, the row's
// dimensions split between the two waves of each SIMD.
//
// Same contract, candidate scheme and canonical re-score as scan_topk.hip (read that header first); same
// reference lines replaced: torch.mm (models/retrievers/dense.py:81) + torch.topk (modules/retrieve.py:157).
//
// Why a second kernel.  In the 4-wave kernel one wave per SIMD does everything in sequence — issue the stage's
// LDS-DMA loads (each issue blocks the wave while the memory pipeline is saturated), 48 MFMAs per 32-row tile,
// the threshold filter — so the matrix pipe idles whenever its only wave is busy with something else (measured:
// stream alone 4.5 ms, stream + MFMA + filter 5.8 ms per launch; with paired workgroups the sum, not the
// maximum, of stream and compute).  Two waves per SIMD fix that, but a wave's 32 queries x 768 dims are 192
// registers of fragments and two such waves do not fit the 512-register lane.  So the K dimension is split:
//
//   wave (r, s)   r = 0 | 1 (which half of every row's dimensions),  s = 0..3 (SIMD = 32-query block)
//     * keeps the fragments of queries 32 s .. 32 s + 31 for ITS half of the dims (96 registers at d = 768),
//     * fetches (LDS-DMA) and reads ONLY its half's 128-byte lines of each 32-row tile,
//     * accumulates a partial inner product per (row, query).
//   r = 1 hands its 32 x 32 partial sums to r = 0 through a 4 KiB LDS mailbox per SIMD once per tile; r = 0 adds
//   them (one tile later, behind the next stage barrier) and runs the threshold filter / candidate append /
//   compaction / threshold exchange exactly as the 4-wave kernel does.  While one wave of a SIMD is blocked in
//   a DMA issue, waits at a barrier or runs filter VALU code, the other one feeds the matrix pipe.
//
// The fp32 score is now (sum over the first half) + (sum over the second half) instead of one running sum; it
// only ranks candidates (KP >= k + 8 kept), results are the canonical fp64 re-scores of merge_rescore.hip, so
// ids and scores stay bit-identical to the 4-wave kernel and to the oracle.
//
// LDS: R stages x (32 rows x LS lines x 128 B) ring, lines [0, LS/2) of a stage belong to half 0 and
// [LS/2, LS) to half 1, each in the XOR-permuted image of scan_topk.hip; + the mailboxes (4 KiB per SIMD, two
// sets when a tile is a single stage).  d = 768: 6 x 24 KiB + 16 KiB = 160 KiB, all of the CU's LDS.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {

template <int EPL>
__device__ __forceinline__ void load_list8(u64 (&e)[EPL], const u64* buf, unsigned n, int lane) {
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const unsigned idx = r * 64 + lane;
        e[r] = idx < n ? buf[idx] : 0ull;
    }
}

template <int KP>
__device__ __forceinline__ void sort_candidates8(u64 (&e)[2 * KP / 64], const u64* buf, unsigned n, int lane) {
    load_list8<2 * KP / 64>(e, buf, n, lane);
    bh_wave_sort_desc<2 * KP / 64>(e, lane);
}

}  // namespace

// NK = padded dim / 16; KP = candidate list length; LS = lines per stage (both halves together); R = ring depth;
// NT = non-temporal policy on the corpus stream; ABL (bench-only): 1 = no filter, 2 = stream only, 4 = MFMA
// without LDS reads.
template <int NK, int KP, int LS, int R, bool NT, int ABL = 0>
__global__ void __launch_bounds__(512, 1) bh_scan_topk8_kernel(BhScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NK * 16;
    constexpr int LINES = D / 64;
    static_assert(LINES % 2 == 0 && LS % 2 == 0 && LINES % LS == 0, "both halves need whole lines in every stage");
    constexpr int S = LINES / LS;    // stages per 32-row tile
    constexpr int HLS = LS / 2;      // lines of ONE half per stage
    constexpr int HL = LINES / 2;    // lines of one half per row
    constexpr int NKH = NK / 2;      // k-steps (fragments) per half
    constexpr int STAGE_BYTES = 32 * LS * 128;
    constexpr int NBOX = S == 1 ? 2 : 1;  // mailbox sets (see the hand-over below)
    constexpr int CAP = 2 * KP;
    constexpr int EPLC = CAP / 64;
    constexpr int EPLK = KP / 64;
    constexpr int BQ = 128;
    constexpr int ROW_BYTES = D * 2;
    constexpr int RB = KP / 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;  // 0: first half of the dims + filter, 1: second half
    const int sq = wave & 3;     // 32-query block (= the SIMD the two waves of a block share, dispatch order permitting)
    const int ql = lane & 31, h = lane >> 5;

    // query split (see scan_topk.hip): workgroups xb and xb ^ 8 walk the same row tiles for different queries
    const int QS = a.qsplit;
    const int xb = blockIdx.x;
    const int G = QS == 1 ? (int)gridDim.x : (int)gridDim.x / QS;
    const int qh = QS == 1 ? 0 : (xb >> 3) % QS;
    const int b = QS == 1 ? xb : (((xb >> 3) / QS) << 3) | (xb & 7);

    const long long my_tiles = (a.n_tiles > b) ? (a.n_tiles - b + G - 1) / G : 0;
    u64* cand_wg = a.cand + (size_t)xb * BQ * CAP;
    u64* part_wg = a.partial + ((size_t)b * QS + qh) * BQ * KP;
    a.qtile += (size_t)qh * BQ * D;
    a.gthr += (size_t)qh * BQ * 64;

    // ---- this wave's query fragments: k-steps [role*NKH, role*NKH + NKH) of queries 32 sq + ql
    half8 qf[NKH];
    {
        const _Float16* qrow = a.qtile + (size_t)(sq * 32 + ql) * D + (size_t)role * NKH * 16;
#pragma unroll
        for (int t = 0; t < NKH; ++t) qf[t] = *reinterpret_cast<const half8*>(qrow + (2 * t + h) * 8);
    }
    // pin into the accumulator file next to the MFMA accumulators and the held partial sums (hipcc splits the 256
    // registers of a wave 128 / 128 here; never over-subscribe the "a" constraint, see scan_topk.hip)
    constexpr int PIN_MAX = (128 - 32) / 4;
    static_assert(NKH <= PIN_MAX, "query fragments + two accumulator tiles must fit the 128 AGPRs");
#pragma unroll
    for (int t = 0; t < NKH; ++t)
        if (t < PIN_MAX) asm volatile("" : "+a"(qf[t]));

    // filter state (used by role 0 only)
    float thr = -__builtin_inff();
    unsigned cnt = 0;
    float best[RB];
    float pub = -__builtin_inff();
    long long next_poll = 0;
#pragma unroll
    for (int r = 0; r < RB; ++r) best[r] = -__builtin_inff();

    // ---- per-lane constants of the LDS-DMA source pattern and of the fragment reads (as in scan_topk.hip)
    const int ld_row = 8 * sq + (lane >> 3);
    const int ld_g = ((lane >> 4) & 1) | (sq << 1);
    const unsigned ld_off = (unsigned)ld_row * ROW_BYTES + (unsigned)(((lane & 7) ^ ld_g) << 4) + (unsigned)role * HL * 128;
    const int rd_g = ((ql >> 1) & 1) | ((ql >> 3) << 1);
    unsigned rd_off[4];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
        rd_off[j4] = (unsigned)(role * HLS * 4096 + (ql >> 3) * 1024 + (ql & 7) * 128 + (((2 * j4 + h) ^ rd_g) << 4));
    unsigned char* box = smem + R * STAGE_BYTES + sq * 4096 + lane * 16;  // + set * 16384 + 1024 * (v / 4)

    auto compact = [&](int qq, int sqc) {
        const int qi = sqc * 32 + qq;
        const unsigned n = __builtin_amdgcn_readlane(cnt, qq);
        u64* buf = cand_wg + (size_t)qi * CAP;
        u64 e[EPLC];
        sort_candidates8<KP>(e, buf, n, lane);
#pragma unroll
        for (int r = 0; r < EPLK; ++r) buf[r * 64 + lane] = e[r];
        if (ql == qq) cnt = n < (unsigned)KP ? n : (unsigned)KP;
        const u64 kth = bh_shfl64(e[EPLK - 1], 63);
        if (kth != 0ull) {
            const float nt = bh_key_score(kth);
            if (ql == qq) thr = fmaxf(thr, nt);
        }
    };

    // Threshold filter of one finished tile (role 0): tot = this wave's partial sums + the mailbox.
    auto filter_tile = [&](const floatx16& tot, long long tile_ord) {
        // hipcc hoists the cold path's address arithmetic out of the tile loop and then spills it; an opaque copy
        // of the wave's block id keeps those computations (and their registers) inside the rarely taken branches
        int sqc = sq;
        asm volatile("" : "+s"(sqc));
        const long long row0 = (b + tile_ord * G) * 32;
        float m = tot[0];
#pragma unroll
        for (int v = 1; v < 16; ++v) m = fmaxf(m, tot[v]);
        if constexpr (ABL != 0) {
            asm volatile("" ::"v"(m));
            m = -__builtin_inff();
        }
        if (__builtin_amdgcn_ballot_w64(m > thr) != 0ull) {
            u64 need = __builtin_amdgcn_ballot_w64(cnt > (unsigned)(CAP - 32)) & 0xffffffffull;
            while (need) {
                const int qq = __builtin_ctzll(need);
                need &= need - 1;
                compact(qq, sqc);
            }
            u64* buf = cand_wg + (size_t)(sqc * 32 + ql) * CAP;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long row = row0 + (v & 3) + 8 * (v >> 2) + 4 * h;
                const bool hit = (tot[v] > thr) && (row < a.n_rows);
                const u64 hm = __builtin_amdgcn_ballot_w64(hit);
                if (hm != 0ull) {
                    const unsigned hl = ((unsigned)hm >> ql) & 1u;
                    const unsigned hh = ((unsigned)(hm >> 32) >> ql) & 1u;
                    if (hit) {
                        buf[cnt + (h ? hl : 0u)] = bh_make_key(tot[v], (unsigned)row);
                        float x = tot[v];
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            const float hi = fmaxf(best[r], x);
                            x = fminf(best[r], x);
                            best[r] = hi;
                        }
                    }
                    cnt += hl + hh;
                }
            }
        }
        // threshold exchange through the slot table, geometric schedule (scan_topk.hip)
        if (a.share && tile_ord >= next_poll) {
            next_poll = tile_ord + 1 + (tile_ord >> 1);
            const int q = sqc * 32 + ql;
            const float mine = best[RB - 1];
            if (mine > pub) {
                pub = mine;
                __hip_atomic_fetch_max(a.gthr + (size_t)q * 64 + (b & 63), bh_ordf(mine), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            uint4 sl[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int q2 = sqc * 32 + it * 4 + (lane >> 4);
                const unsigned* src = a.gthr + (size_t)q2 * 64 + (lane & 15) * 4;
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
                const unsigned mine2 = (unsigned)__shfl((int)mn, (ql & 3) * 16, 64);
                if ((ql >> 2) == it && mine2 > BH_ORD_NEG_INF) thr = fmaxf(thr, bh_unordf(mine2 - 1u));
            }
        }
    };

    if (my_tiles > 0) {
        const unsigned char* corpus = reinterpret_cast<const unsigned char*>(a.corpus);
        long long it = 0;
        int ip = 0;
        int islot = 0;
        auto issue_stage = [&]() {
            const long long itc = it < my_tiles ? it : my_tiles - 1;  // past the end: harmless re-fetch
            const long long tile = b + itc * G;
            const unsigned char* src = corpus + (size_t)tile * 32 * ROW_BYTES + (size_t)ip * HLS * 128 + ld_off;
            unsigned char* dst = smem + islot * STAGE_BYTES + role * HLS * 4096 + sq * 1024;
#pragma unroll
            for (int j = 0; j < HLS; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0,
                                                 NT ? 2 : 0);
            if (++ip == S) { ip = 0; ++it; }
            if (++islot == R) islot = 0;
        };
#pragma unroll
        for (int p = 0; p < R - 1; ++p) issue_stage();

        int cslot = 0;
        floatx16 prev;  // role 0: its partial sums of the previous tile, waiting for the mailbox
#pragma unroll
        for (int v = 0; v < 16; ++v) prev[v] = 0.f;

        for (long long i = 0; i <= my_tiles; ++i) {
            const bool live = i < my_tiles;  // the extra iteration only drains the last hand-over
            floatx16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
            for (int part = 0; part < S; ++part) {
                // own LDS-DMA of this stage landed, everybody's landed, everybody left the slot the next issue
                // overwrites — and (part 0) the mailbox of the previous tile is complete
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((R - 2) * HLS) : "memory");
                if (live) issue_stage();
                const unsigned char* st = smem + cslot * STAGE_BYTES;
                const bool take = role == 0 && part == 0 && i > 0;
                if (live) {
                    if constexpr (ABL == 2) {
                        // stream only
                    } else if constexpr (ABL == 4) {
#pragma unroll
                        for (int jl = 0; jl < HLS; ++jl)
#pragma unroll
                            for (int j4 = 0; j4 < 4; ++j4)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[j4], qf[(part * HLS + jl) * 4 + j4], acc, 0, 0, 0);
                    } else {
                        // GF fragments per group, reads one group ahead of the MFMAs (the SIMD's other wave covers
                        // what is left of the LDS latency)
                        constexpr int GF = 2;
                        constexpr int NG = HLS * 4 / GF;
                        half8 ag[2][GF];
                        auto frag_ptr = [&](int gi, int f) {
                            const int k4 = gi * GF + f;  // fragment ordinal inside this half's lines of the stage
                            return reinterpret_cast<const half8*>(st + (k4 >> 2) * 4096 + rd_off[k4 & 3]);
                        };
#pragma unroll
                        for (int f = 0; f < GF; ++f) ag[0][f] = *frag_ptr(0, f);
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int f = 0; f < GF; ++f) asm volatile("" : "+v"(ag[g & 1][f]));
                            __builtin_amdgcn_sched_barrier(0);
                            if (g + 1 < NG) {
#pragma unroll
                                for (int f = 0; f < GF; ++f) ag[(g + 1) & 1][f] = *frag_ptr(g + 1, f);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int f = 0; f < GF; ++f)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag[g & 1][f], qf[part * HLS * 4 + g * GF + f], acc, 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (take) {
                    // the filter of tile i-1 runs here, beside the partner wave's MFMAs of this stage
                    const unsigned char* bx = box + (NBOX == 2 ? ((i - 1) & 1) * 16384 : 0);
                    floatx16 tot;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const floatx4 t4 = *reinterpret_cast<const floatx4*>(bx + g * 1024);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[4 * g + e] = t4[e] + prev[4 * g + e];
                    }
                    filter_tile(tot, i - 1);
                }
                if (live && ++cslot == R) cslot = 0;
            }
            if (live) {
                if (role == 1) {
                    // hand-over: partial sums of tile i -> mailbox (read by role 0 behind the next barrier; with a
                    // single stage per tile that read overlaps the next write, hence two sets)
                    unsigned char* bx = box + (NBOX == 2 ? (i & 1) * 16384 : 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        floatx4 t4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) t4[e] = acc[4 * g + e];
                        *reinterpret_cast<floatx4*>(bx + g * 1024) = t4;
                    }
                } else {
                    prev = acc;
                    asm volatile("" : "+a"(prev));  // parked in the accumulator file until the mailbox arrives
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- final: role-0 waves sort their queries' buffers and publish the best KP
    if (role == 0) {
        for (int qq = 0; qq < 32; ++qq) {
            const int qi = sq * 32 + qq;
            const unsigned n = __builtin_amdgcn_readlane(cnt, qq);
            u64 e[EPLC];
            sort_candidates8<KP>(e, cand_wg + (size_t)qi * CAP, n, lane);
#pragma unroll
            for (int r = 0; r < EPLK; ++r) part_wg[(size_t)qi * KP + r * 64 + lane] = e[r];
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int NK, int KP, int LS, int R, bool NT, int ABL = 0>
static hipError_t launch8_one(const BhScanArgs& a, int grid, hipStream_t stream) {
    constexpr int S = (NK / 4) / LS;
    constexpr size_t smem = (size_t)R * 32 * LS * 128 + (S == 1 ? 2 : 1) * 16384;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    auto kern = bh_scan_topk8_kernel<NK, KP, LS, R, NT, ABL>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, a);
    return hipGetLastError();
}

template <int NK, int LS, int R>
static hipError_t launch8_kp(const BhScanArgs& a, int kp, int grid, hipStream_t stream) {
    const bool nt = a.nontemporal != 0;
    if constexpr (NK == 48) {
        if (a.ablate != 0 && kp == 64) {
            switch (a.ablate) {
                case 1: return launch8_one<NK, 64, LS, R, true, 1>(a, grid, stream);
                case 2: return launch8_one<NK, 64, LS, R, true, 2>(a, grid, stream);
                case 4: return launch8_one<NK, 64, LS, R, true, 4>(a, grid, stream);
            }
            return hipErrorInvalidValue;
        }
    }
    switch (kp) {
        case 64: return nt ? launch8_one<NK, 64, LS, R, true>(a, grid, stream) : launch8_one<NK, 64, LS, R, false>(a, grid, stream);
        case 128: return nt ? launch8_one<NK, 128, LS, R, true>(a, grid, stream) : launch8_one<NK, 128, LS, R, false>(a, grid, stream);
        case 256: return nt ? launch8_one<NK, 256, LS, R, true>(a, grid, stream) : launch8_one<NK, 256, LS, R, false>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}

bool bh_scan8_supports(int dim_padded) { return dim_padded >= 128 && dim_padded <= 768; }

// dim_padded in {128,256,384,512,768} (d = 64 has a single line per row, d = 1024 does not fit the
// accumulator file: both stay on the 4-wave kernel); ring geometry: lines per stage x depth (+ mailboxes) <= 160 KiB
hipError_t bh_launch_scan8(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream) {
    switch (dim_padded) {
        case 128: return launch8_kp<8, 2, 6>(a, kp, grid, stream);
        case 256: return launch8_kp<16, 4, 6>(a, kp, grid, stream);
        case 384: return launch8_kp<24, 6, 5>(a, kp, grid, stream);   // one stage per tile: two mailbox sets
        case 512: return launch8_kp<32, 8, 4>(a, kp, grid, stream);
        case 768: return launch8_kp<48, 6, 6>(a, kp, grid, stream);
    }
    return hipErrorInvalidValue;
}
