// scan_topk192.hip — fused inner-product + running top-k scan with 192 queries per corpus pass (option scan_kernel 2).
//
// Same contract, ring, candidate scheme and canonical re-score as scan_topk.hip (read that header first); same
// reference lines replaced: torch.mm (models/retrievers/dense.py:81) + torch.topk (modules/retrieve.py:157).
//
// Why.  A corpus pass costs the same HBM bytes whatever the number of queries riding on it; the 128-query kernel is
// bounded by that stream.  256 queries per workgroup need 384 of the 512 registers per lane for fragments and hipcc
// spills them into the main loop; with `v_mfma_f32_16x16x32_f16` the natural unit is a block of 16 queries, and THREE
// blocks per wave (48 queries, 288 fragment registers, 24 accumulators for a 32-row tile) leave a comfortable
// remainder: 1.5x the queries per pass for 1.5x the MFMA work.
//
//   * wave w of the 4-wave workgroup keeps queries 48 w .. 48 w + 47 as MFMA B fragments: lane (q = l & 15, g = l >> 4),
//     k-step s covers dims 32 s + 8 g .. + 8;
//   * A fragments (corpus rows) per k-step: lane reads row rb * 16 + (l & 15), 16 bytes at dim 32 s + 8 g: one
//     ds_read_b128 serves three MFMAs; the LDS image is scan_topk.hip's with the chunk permutation g(row) =
//     (row >> 1) & 7 (conflict-free for this read pattern: profiles/lds_swizzle_search.py);
//   * C: lane l holds rows 4 (l >> 4) + v (v = 0..3) of query l & 15: lane <-> query as in the other kernels, four lanes
//     per query; the candidate slot of a hit is the query's count plus the hits of the lower lane groups.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {

template <int EPL>
__device__ __forceinline__ void load_list192(u64 (&e)[EPL], const u64* buf, unsigned n, int lane) {
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const unsigned idx = r * 64 + lane;
        e[r] = idx < n ? buf[idx] : 0ull;
    }
}

template <int KP>
__device__ __forceinline__ void sort_candidates192(u64 (&e)[2 * KP / 64], const u64* buf, unsigned n, int lane) {
    load_list192<2 * KP / 64>(e, buf, n, lane);
    bh_wave_sort_desc<2 * KP / 64>(e, lane);
}

}  // namespace

// NK32 = padded dim / 32 (k-steps per row); KP = candidate list length; LS = 128-byte lines per stage; R = ring depth.
template <int NK32, int KP, int LS, int R, bool NT>
__global__ void __launch_bounds__(256, 1) bh_scan_topk192_kernel(BhScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NK32 * 32;
    constexpr int LINES = D / 64;
    static_assert(LINES % LS == 0, "stage must divide the row");
    constexpr int S = LINES / LS;
    constexpr int STAGE_BYTES = 32 * LS * 128;
    constexpr int CAP = 2 * KP;
    constexpr int EPLC = CAP / 64;
    constexpr int EPLK = KP / 64;
    constexpr int NB = 3;          // 16-query blocks per wave
    constexpr int BQ = 64 * NB;    // 192 queries per workgroup
    constexpr int ROW_BYTES = D * 2;
    constexpr int RB = KP / 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int q16 = lane & 15, lg = lane >> 4;

    const long long my_tiles = (a.n_tiles > b) ? (a.n_tiles - b + G - 1) / G : 0;
    u64* cand_wg = a.cand + (size_t)b * BQ * CAP;
    u64* part_wg = a.partial + (size_t)b * BQ * KP;

    // ---- queries -> registers (B fragments)
    half8 qf[NB][NK32];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const _Float16* qrow = a.qtile + (size_t)((wave * NB + nb) * 16 + q16) * D;
#pragma unroll
        for (int s = 0; s < NK32; ++s) qf[nb][s] = *reinterpret_cast<const half8*>(qrow + 32 * s + 8 * lg);
    }
    constexpr int PIN_MAX = (256 - 8 * NB - 16) / 4;  // never over-subscribe the "a" constraint (scan_topk.hip)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < NK32; ++s)
            if (nb * NK32 + s < PIN_MAX) asm volatile("" : "+a"(qf[nb][s]));

    float thr[NB];       // candidate iff score > thr (per lane = per query)
    unsigned cnt[NB];    // entries in the query's candidate buffer (identical in the four lanes of a query)
    float best[NB][RB];  // this lane's RB best appended scores, descending
    float pub[NB];
    long long next_poll = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        thr[nb] = -__builtin_inff();
        cnt[nb] = 0;
        pub[nb] = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < RB; ++r) best[nb][r] = -__builtin_inff();
    }

    // ---- LDS-DMA source pattern: lanes 8j..8j+7 of wave w fetch the eight chunks of row 8 w + j, permuted by g(row)
    const int ld_row = 8 * wave + (lane >> 3);
    const int ld_g = (ld_row >> 1) & 7;
    const unsigned ld_off = (unsigned)ld_row * ROW_BYTES + (unsigned)(((lane & 7) ^ ld_g) << 4);
    // ---- fragment read offsets: row block rb, k-step parity sb (two k-steps per 128-byte line)
    unsigned rd_off[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int row = rb * 16 + q16;
            rd_off[rb][sb] = (unsigned)((row >> 3) * 1024 + (row & 7) * 128 + (((4 * sb + lg) ^ ((row >> 1) & 7)) << 4));
        }

    auto compact = [&](int nb, int qq) {
        const int qi = (wave * NB + nb) * 16 + qq;
        const unsigned n = __builtin_amdgcn_readlane(cnt[nb], qq);
        u64* buf = cand_wg + (size_t)qi * CAP;
        u64 e[EPLC];
        sort_candidates192<KP>(e, buf, n, lane);
#pragma unroll
        for (int r = 0; r < EPLK; ++r) buf[r * 64 + lane] = e[r];
        if (q16 == qq) cnt[nb] = n < (unsigned)KP ? n : (unsigned)KP;
        const u64 kth = bh_shfl64(e[EPLK - 1], 63);
        if (kth != 0ull) {
            // rows arrive in ascending order inside a workgroup: a later row that merely TIES the KP-th best loses on
            // row index, so the exclusive compare against the workgroup's own bound is exact
            const float nt = bh_key_score(kth);
            if (q16 == qq) thr[nb] = fmaxf(thr[nb], nt);
        }
    };

    if (my_tiles > 0) {
        const unsigned char* corpus = reinterpret_cast<const unsigned char*>(a.corpus);
        long long it = 0;
        int ip = 0;
        int islot = 0;
        auto issue_line = [&](int j) {
            const long long itc = it < my_tiles ? it : my_tiles - 1;  // past the end: harmless re-fetch
            const long long tile = b + itc * G;
            const unsigned char* src = corpus + (size_t)tile * 32 * ROW_BYTES + (size_t)ip * LS * 128 + ld_off;
            unsigned char* dst = smem + islot * STAGE_BYTES + wave * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 128),
                                             (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0, NT ? 2 : 0);
        };
        auto advance_cursor = [&]() {
            if (++ip == S) { ip = 0; ++it; }
            if (++islot == R) islot = 0;
        };
#pragma unroll
        for (int p = 0; p < R - 1; ++p) {
#pragma unroll
            for (int j = 0; j < LS; ++j) issue_line(j);
            advance_cursor();
        }

        int cslot = 0;
        for (long long i = 0; i < my_tiles; ++i) {
            floatx4 acc[2][NB];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[rb][nb][v] = 0.f;
#pragma unroll
            for (int part = 0; part < S; ++part) {
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * LS) : "memory");
                const unsigned char* st = smem + cslot * STAGE_BYTES;
                // one line (two k-steps x two row blocks = four fragments) per group, reads one group ahead of the MFMAs;
                // the stage's refill instruction for line g follows the MFMAs of line g
                half8 ag[2][4];
#pragma unroll
                for (int f = 0; f < 4; ++f) ag[0][f] = *reinterpret_cast<const half8*>(st + rd_off[f & 1][f >> 1]);
#pragma unroll
                for (int g = 0; g < LS; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(ag[g & 1][f]));
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 1 < LS) {
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            ag[(g + 1) & 1][f] = *reinterpret_cast<const half8*>(st + (g + 1) * 4096 + rd_off[f & 1][f >> 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < 4; ++f) {  // f = rb + 2 sb
                        const int ks = (part * LS + g) * 2 + (f >> 1);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[f & 1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ag[g & 1][f], qf[nb][ks], acc[f & 1][nb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    issue_line(g);
                    __builtin_amdgcn_sched_barrier(0);
                }
                advance_cursor();
                if (++cslot == R) cslot = 0;
            }

            // ---- threshold filter
            const long long row0 = (b + i * G) * 32;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float m = acc[0][nb][0];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) m = fmaxf(m, acc[rb][nb][v]);
                if (__builtin_amdgcn_ballot_w64(m > thr[nb]) != 0ull) {
                    // (1) make room: a tile adds at most 32 entries per query
                    u64 need = __builtin_amdgcn_ballot_w64(cnt[nb] > (unsigned)(CAP - 32)) & 0xffffull;
                    while (need) {
                        const int qq = __builtin_ctzll(need);
                        need &= need - 1;
                        compact(nb, qq);
                    }
                    // (2) append survivors; slot = count + hits of the same query in the lower lane groups
                    u64* buf = cand_wg + (size_t)((wave * NB + nb) * 16 + q16) * CAP;
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const long long row = row0 + rb * 16 + 4 * lg + v;
                            const float sv = acc[rb][nb][v];
                            const bool hit = (sv > thr[nb]) && (row < a.n_rows);
                            const u64 hm = __builtin_amdgcn_ballot_w64(hit);
                            if (hm != 0ull) {
                                const unsigned h0 = (unsigned)(hm >> q16) & 1u, h1 = (unsigned)(hm >> (q16 + 16)) & 1u;
                                const unsigned h2 = (unsigned)(hm >> (q16 + 32)) & 1u, h3 = (unsigned)(hm >> (q16 + 48)) & 1u;
                                const unsigned below = lg == 0 ? 0u : lg == 1 ? h0 : lg == 2 ? h0 + h1 : h0 + h1 + h2;
                                if (hit) {
                                    buf[cnt[nb] + below] = bh_make_key(sv, (unsigned)row);
                                    float x = sv;
#pragma unroll
                                    for (int r = 0; r < RB; ++r) {
                                        const float hi = fmaxf(best[nb][r], x);
                                        x = fminf(best[nb][r], x);
                                        best[nb][r] = hi;
                                    }
                                }
                                cnt[nb] += h0 + h1 + h2 + h3;
                            }
                        }
                }
            }
            // ---- threshold exchange through the slot table (scan_topk.hip), geometric schedule
            if (a.share && i >= next_poll) {
                next_poll = i + 1 + (i >> 1);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int q = (wave * NB + nb) * 16 + q16;
                    const float mine = best[nb][RB - 1];
                    if (mine > pub[nb]) {
                        pub[nb] = mine;
                        __hip_atomic_fetch_max(a.gthr + (size_t)q * 64 + (b & 63), bh_ordf(mine), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    uint4 sl[4];  // 16 lanes x 4 slots cover one query; 4 queries per load instruction
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int q = (wave * NB + nb) * 16 + t4 * 4 + (lane >> 4);
                        const unsigned* src = a.gthr + (size_t)q * 64 + (lane & 15) * 4;
                        sl[t4].x = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sl[t4].y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sl[t4].z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sl[t4].w = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        unsigned mn = min(min(sl[t4].x, sl[t4].y), min(sl[t4].z, sl[t4].w));
#pragma unroll
                        for (int o = 8; o >= 1; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64));
                        // lanes 16 g .. 16 g + 15 hold the minimum of query t4 * 4 + g of this block
                        const unsigned got = (unsigned)__shfl((int)mn, (q16 & 3) * 16, 64);
                        // a row that TIES the bound may still win on row index: inclusive compare
                        if ((q16 >> 2) == t4 && got > BH_ORD_NEG_INF) thr[nb] = fmaxf(thr[nb], bh_unordf(got - 1u));
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- final: every wave sorts its queries' buffers and publishes the best KP
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        for (int qq = 0; qq < 16; ++qq) {
            const int qi = (wave * NB + nb) * 16 + qq;
            const unsigned n = __builtin_amdgcn_readlane(cnt[nb], qq);
            u64 e[EPLC];
            sort_candidates192<KP>(e, cand_wg + (size_t)qi * CAP, n, lane);
#pragma unroll
            for (int r = 0; r < EPLK; ++r) part_wg[(size_t)qi * KP + r * 64 + lane] = e[r];
        }
    }
}

template <int NK32, int KP, int LS, int R>
static hipError_t launch192_one(const BhScanArgs& a, int grid, hipStream_t stream) {
    constexpr size_t smem = (size_t)R * 32 * LS * 128;
    const bool nt = a.nontemporal != 0;
    static bool attr_done[2] = {false, false};
    auto kern = nt ? bh_scan_topk192_kernel<NK32, KP, LS, R, true> : bh_scan_topk192_kernel<NK32, KP, LS, R, false>;
    if (!attr_done[nt ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done[nt ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, a);
    return hipGetLastError();
}

// the headline geometry only: d = 768 (padded), candidate lists of 64 (k <= 56)
bool bh_scan192_supports(int dim_padded, int kp) { return dim_padded == 768 && kp == 64; }

hipError_t bh_launch_scan192(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream) {
    if (!bh_scan192_supports(dim_padded, kp) || a.qsplit != 1) return hipErrorInvalidValue;
    return launch192_one<24, 64, 6, 6>(a, grid, stream);
}
