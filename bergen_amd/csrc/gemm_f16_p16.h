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
//     before); the fragments of both k-steps of a stage double-buffered in registers, all reads of the next k-step issued behind the first five
//     MFMA groups of the current one.
//   * epilogue: lane (q16, lg) of block (tb, fb) holds token 16 tb + q16, features 16 fb + 4 lg + 0..3: one 8-byte LDS write per block
//     into the wave's 32-row x 128-byte staging image (16-byte chunks XORed with the row), read back and stored exactly as before.
//   * TAIL SPLIT (BhGemmArgs::tail_split): an XCD's 32 workgroups walk its tiles in rounds, and a last round with r <= 16 tiles left 16
//     or more of them idle for a whole tile time (the bench batch's micro-batch: 780 tiles of the Q | K projection = 3.05 rounds, paid
//     as 4).  Here such a remainder is cut along the TOKENS into 2 (r <= 16) or 4 (r <= 8) sub-tiles of 128 or 64 tokens x 256 features,
//     one per workgroup, all eight waves on each (64 or 32 tokens x 64 features per wave); same pipeline, same stage count, a half or a
//     quarter of the MFMAs per stage.  Every output element still sums the same k-steps in the same order: same bits.  OFF by default: +8 % on
//     a projection launched alone, -1 % inside the encoder, whose second micro-batch stream already runs in a launch's idle tail.
// Epilogues: none, bias per column (optionally + GELU), bias per row; row-major output or the attention kernel's blocked V^T layout — every
// projection of a BERT layer — and the gated (SwiGLU) fold of NomicBert's feed-forward.  The fused-LayerNorm, segmented-max and batched
// epilogues stay on gemm_f16_persist.h (the SPLADE head's segmented max was built here too and measured: with 4 tokens per lane instead of 8 it
// needs a ds_bpermute and twice the predicated reads of the running maxima per group — head 3.35 -> 3.58 ms per 512 passages; removed).
#pragma once
#include "gemm_f16_persist.h"

// ABL (bench only, results invalid): 1 no LDS-DMA after the pipeline start, 2 no MFMA, 4 no fragment reads, 8 no epilogue, 16 epilogue math without stores
template <int EPI, bool NT, int ABL = 0>
__global__ void __launch_bounds__(512, 2) bh_gemm_f16_p16kernel(BhGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(EPI == 0 || EPI == BH_EPI_BIAS_COL || EPI == (BH_EPI_BIAS_COL | BH_EPI_GELU) || EPI == BH_EPI_BIAS_ROW ||
                      EPI == (BH_EPI_BIAS_COL | BH_EPI_SWIGLU) || EPI == (BH_EPI_BIAS_COL | BH_EPI_SWIGLU | BH_EPI_GELU) ||
                      EPI == (BH_EPI_BIAS_COL | BH_EPI_ROTARY),
                  "epilogues of this kernel");
    constexpr int BK = 64, WN = 4, R = 2;
    constexpr int NW = 8;
    constexpr int BM = 256, BN = 256;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int PIECE = 4096, SUBS = 4;
    constexpr int STAGE_BYTES = (PA + PB) * PIECE;
    constexpr int NL = (PA + PB) * SUBS / NW;  // 8 LDS-DMA instructions per wave per stage
    constexpr int TB = 8, FB = 4;              // 16-token blocks / 16-feature blocks per wave (a whole tile)
    constexpr int NLA = PA * SUBS / NW;        // instructions 0..3 fetch token rows, 4..7 feature rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q16 = lane & 15, lg = lane >> 4;

    // ---- this block's work (gemm_f16_persist.h: XCD x = block % 8 owns a contiguous range of the m-major tile order; its G8 blocks walk
    // it in rounds).  n_full whole tiles t_first + k G8, then — tail split — one sub-tile: part sub_part of 1 << sub_log of tile sub_tile.
    const int tiles_n = a.N / BN;
    const int n_tiles = (a.M / BM) * tiles_n;
    int t_first, n_full, sub_tile = 0, sub_part = 0, sub_log = 0;
    const int G8 = gridDim.x >> 3;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = n_tiles >> 3, r = n_tiles & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        const int rounds = cnt / G8, rem = cnt - rounds * G8;
        t_first = start + j;
        n_full = rounds + (j < rem ? 1 : 0);
        if (a.tail_split && rem > 0 && 2 * rem <= G8) {
            sub_log = 4 * rem <= G8 ? 2 : 1;
            n_full = rounds;
            if ((j >> sub_log) < rem) {
                sub_tile = start + rounds * G8 + (j >> sub_log);
                sub_part = j & ((1 << sub_log) - 1);
            } else {
                sub_log = 0;
            }
        }
    }
    const int n_items = n_full + (sub_log ? 1 : 0);
    if (n_items == 0) return;

    // ---- per-lane LDS-DMA source offsets: instruction i of a wave fetches 8 rows of piece 2 i + (wave >> 2)
    unsigned offA, offB;
    int dst0;
    const int p0 = wave >> 2;
    {
        const int sub = wave & 3;
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

    // ---- issue cursor (runs one stage ahead of the consumer, across item boundaries)
    int it = 0, ikt = 0, islot = 0;
    int ipa = PA;  // token pieces of the item at the cursor: 8, or 4 / 2 for a sub-tile
    const unsigned char *curA, *curB;
    auto set_issue_item = [&](int ord) {
        const bool is_sub = ord >= n_full;
        const int t = is_sub ? sub_tile : t_first + ord * G8;
        const int tm0 = (t / tiles_n) * BM + (is_sub ? sub_part * (BM >> sub_log) : 0), tn0 = (t % tiles_n) * BN;
        ipa = is_sub ? PA >> sub_log : PA;
        curA = baseA + (size_t)tm0 * a.lda * 2;
        curB = baseB + (size_t)tn0 * a.ldb * 2;
    };
    set_issue_item(0);
    bool abl_started = false;
    auto issue_piece = [&](int i) {
        if (i < NLA && 2 * i + p0 >= ipa) return;  // (a sub-tile has fewer token rows)
        if constexpr ((ABL & 1) != 0) {
            if (abl_started) return;
        }
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
            if (it + 1 < n_items) {
                ++it;
                ikt = 0;
                set_issue_item(it);
            } else {
                ikt = KT - 1;  // nothing further: a harmless re-fetch
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
    // Fragments, double-buffered over the k-steps: all reads of the next k-step are issued behind the first five MFMA groups of the current
    // one (measured against single-buffered token fragments refilled rolling: BERT-base forward 13.94 -> 13.80 ms).
    half8 xt[2][TB], wf[2][FB];
    // token block tb of a wave that owns TBS blocks: rows wm * 16 TBS + 16 tb of the (sub-)tile; feature block fb: piece PA + wn * 2 + (fb >> 1)
    auto read_wf = [&](int buf, const unsigned char* st, int ks) {
        const unsigned char* sw = st + (PA + wn * 2) * PIECE + rd_off[ks];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) wf[buf][fb] = *reinterpret_cast<const half8*>(sw + (fb >> 1) * PIECE + (fb & 1) * 2048);
    };
    auto read_xt = [&](auto tbs_c, int tb, const unsigned char* st, int ks) {
        constexpr int TBS = decltype(tbs_c)::value;
        const int row0 = wm * 16 * TBS + 16 * tb;
        xt[ks][tb] = *reinterpret_cast<const half8*>(st + (row0 >> 5) * PIECE + ((row0 >> 4) & 1) * 2048 + rd_off[ks]);
    };
    // reads of k-step ks behind MFMA group tb of the other k-step: the four feature fragments behind group 0, two token fragments behind each
    // of the groups 1 .. 4 (what a short sub-tile schedule has no group for follows its last group)
    auto reads_behind = [&](auto tbs_c, int tb, const unsigned char* st, int ks) {
        constexpr int TBS = decltype(tbs_c)::value;
        if constexpr ((ABL & 4) != 0) return;
        if (tb == 0) read_wf(ks, st, ks);
        if (tb >= 1) {
            if (2 * tb - 2 < TBS) read_xt(tbs_c, 2 * tb - 2, st, ks);
            if (2 * tb - 1 < TBS) read_xt(tbs_c, 2 * tb - 1, st, ks);
        }
        if (tb == TBS - 1) {
#pragma unroll
            for (int u = 2 * tb; u < TBS; ++u) read_xt(tbs_c, u, st, ks);
        }
    };

    int cslot = 0;
    auto stage = [&](auto tbs_c) {
        constexpr int TBS = decltype(tbs_c)::value;
        const unsigned char* st = smem + cslot * STAGE_BYTES;
        if (++cslot == R) cslot = 0;
        // k-step 0 (its fragments are in registers: the wait hipcc puts at the loop head covers exactly them, so no read is issued before the
        // first MFMAs)
#pragma unroll
        for (int tb = 0; tb < TBS; ++tb) {
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
                if constexpr ((ABL & 2) == 0) acc[tb][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][fb], xt[0][tb], acc[tb][fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            reads_behind(tbs_c, tb, st, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 1.  Every read of this stage must have returned before its slot is handed back:
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) asm volatile("" : "+v"(wf[1][fb]));
#pragma unroll
        for (int tb = 0; tb < TBS; ++tb) asm volatile("" : "+v"(xt[1][tb]));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // next stage landed (ring 2)
        const unsigned char* nst = smem + cslot * STAGE_BYTES;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tb = 0; tb < TBS; ++tb) {
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
                if constexpr ((ABL & 2) == 0) acc[tb][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][fb], xt[1][tb], acc[tb][fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the refill goes into the slot everybody just left, one LDS-DMA instruction per MFMA group of a whole tile (also measured, BERT-base
            // forward: two per group in the first four groups 13.80 -> 14.13 ms; the refill spread over both halves of a stage — token rows behind
            // this k-step, feature rows behind the next stage's first — 13.79 -> 14.12 ms: profiles/README.md, round 5)
#pragma unroll
            for (int u = 0; u < NL / TBS; ++u) issue_piece(tb * (NL / TBS) + u);
            reads_behind(tbs_c, tb, nst, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_advance();
    };

    // ---- epilogue of a (sub-)tile whose rows start at m0: bias (+ GELU), fp16, through the wave's staging image, out as whole 128-byte
    // lines; leaves the accumulators zero (a first stage with a zero C operand instead made hipcc spill six accumulator quads per tile)
    unsigned char* stg = smem + R * STAGE_BYTES + wave * 4096;  // 32 token rows x 128 bytes
    auto epilogue = [&](auto tbs_c, int m0, int n0) {
        constexpr int TBS = decltype(tbs_c)::value;
        if constexpr ((ABL & 8) != 0) return;
        if constexpr ((EPI & BH_EPI_SWIGLU) != 0) {
            // Gated feed-forward (NomicBertMLP): the columns are (gate, up) pairs, C is [M][N / 2] = silu(gate) * up — the arithmetic of
            // gemm_f16_persist.h's fold, expression for expression (same bits).  A lane's four columns are two pairs = two outputs = 4 bytes; the
            // wave's 64 columns fold to 64-byte row pieces, staged as 32 rows x 64 bytes and stored 16 rows per instruction.
            const int rrow = lane >> 2, rch = lane & 3;
            _Float16* gptr = a.C + (size_t)(m0 + wm * 16 * TBS + rrow) * a.ldc + ((n0 + wn * 64) >> 1) + rch * 8;
            half4 bias4[FB];
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) bias4[fb] = *reinterpret_cast<const half4*>(a.bias + n0 + wn * 64 + fb * 16 + 4 * lg);
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int tp = 0; tp < (TBS + 1) / 2; ++tp) {
#pragma unroll
                for (int half_ = 0; half_ < (TBS > 1 ? 2 : 1); ++half_) {
                    const int tb = 2 * tp + half_;
                    const int tr = half_ * 16 + q16;
#pragma unroll
                    for (int fb = 0; fb < FB; ++fb) {
                        half2v o;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float g = acc[tb][fb][2 * e] + 0.f + (float)bias4[fb][2 * e];
                            const float up = acc[tb][fb][2 * e + 1] + 0.f + (float)bias4[fb][2 * e + 1];
                            if constexpr ((EPI & BH_EPI_GELU) != 0)  // GELU gate (gte-*-en-v1.5): bh_swiglu_kernel's act = 1, same bits
                                o[e] = (_Float16)(bh_gemm::gelu_erf(g) * up);
                            else
                                o[e] = (_Float16)(g / (1.0f + __builtin_amdgcn_exp2f(-g * 1.4426950408889634f)) * up);
                        }
                        acc[tb][fb] = floatx4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<half2v*>(stg + tr * 64 + ((fb ^ ((tr >> 1) & 3)) << 4) + lg * 4) = o;
                    }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < (TBS > 1 ? 2 : 1); ++i) {
                    const int row = 16 * i + rrow;
                    const half8 v = *reinterpret_cast<const half8*>(stg + row * 64 + ((rch ^ ((row >> 1) & 3)) << 4));
                    half8* p = reinterpret_cast<half8*>(gptr + (size_t)(tp * 32 + 16 * i) * a.ldc);
                    if constexpr (NT)
                        __builtin_nontemporal_store(v, p);
                    else
                        *p = v;
                }
                asm volatile("" ::: "memory");
            }
            return;
        }
        if constexpr ((EPI & BH_EPI_ROTARY) != 0) {
            // Q | K projection with rotary positions (NomicBert, gte): the wave's 64 columns are ONE head slice; a lane holds, per 16-token block,
            // columns 16 fb + 4 lg + e of it (fb < 4, e < 4): the rotate-half partner of column j < 32 is j + 32 = the SAME lane's value of
            // feature block fb + 2.  x1' = x1 cos - x2 sin, x2' = x2 cos + x1 sin; cos / sin of the token's position from the fp32 table (L1 / L2 hits: a batch spans a few hundred positions),
            // all of a part's loads issued before its arithmetic.  Stores as in the plain epilogue below.
            const int rrow = lane >> 3, rch = lane & 7;
            _Float16* gptr = a.C + (size_t)(m0 + wm * 16 * TBS + rrow) * a.ldc + rch * 8 + (size_t)(n0 + wn * 64);
            half4 bias4[FB];
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) bias4[fb] = *reinterpret_cast<const half4*>(a.bias + n0 + wn * 64 + fb * 16 + 4 * lg);
#pragma unroll
            for (int tp = 0; tp < (TBS + 1) / 2; ++tp) {
                floatx4 co[2][2], si[2][2];  // [half_][fb]
#pragma unroll
                for (int half_ = 0; half_ < (TBS > 1 ? 2 : 1); ++half_) {
                    int pos = a.rot_pos[m0 + wm * 16 * TBS + (2 * tp + half_) * 16 + q16];
                    pos = pos < 0 ? 0 : pos < a.rot_max_pos ? pos : a.rot_max_pos - 1;
                    const float* cs = a.rot_cs + (size_t)pos * 64 + 4 * lg;
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb) {
                        co[half_][fb] = *reinterpret_cast<const floatx4*>(cs + fb * 16);
                        si[half_][fb] = *reinterpret_cast<const floatx4*>(cs + 32 + fb * 16);
                    }
                }
#pragma unroll
                for (int half_ = 0; half_ < (TBS > 1 ? 2 : 1); ++half_) {
                    const int tb = 2 * tp + half_;
                    const int tr = half_ * 16 + q16;
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb) {
                        half4 o1, o2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // (the projection is ROUNDED to fp16 before it is rotated, and the rotation is bh_rotary_kernel's expression, fma for
                            // fma: rows of whole tiles and rows the standalone kernel rotates — edge strips, small batches — carry the same
                            // bits, so a sequence's embedding does not depend on the batch it is encoded in)
                            const float x1 = (float)(_Float16)(acc[tb][fb][e] + (float)bias4[fb][e]), x2 = (float)(_Float16)(acc[tb][fb + 2][e] + (float)bias4[fb + 2][e]);
                            // (the fp32 results pass through an opaque register: hipcc otherwise selects v_fma_mix*_f16, which rounds the exact
                            // fma ONCE to fp16 — the standalone kernel rounds to fp32 first: one double-rounding case in a few thousand, enough
                            // to move 10 % of a layer's outputs by an ulp)
                            float r1 = __builtin_fmaf(-x2, si[half_][fb][e], x1 * co[half_][fb][e]);
                            float r2 = __builtin_fmaf(x1, si[half_][fb][e], x2 * co[half_][fb][e]);
                            asm volatile("" : "+v"(r1), "+v"(r2));
                            o1[e] = (_Float16)r1;
                            o2[e] = (_Float16)r2;
                        }
                        acc[tb][fb] = floatx4{0.f, 0.f, 0.f, 0.f};
                        acc[tb][fb + 2] = floatx4{0.f, 0.f, 0.f, 0.f};
                        const int c1 = fb * 2 + (lg >> 1), c2 = (fb + 2) * 2 + (lg >> 1);
                        *reinterpret_cast<half4*>(stg + tr * 128 + ((c1 ^ ((tr >> 1) & 7)) << 4) + (lg & 1) * 8) = o1;
                        *reinterpret_cast<half4*>(stg + tr * 128 + ((c2 ^ ((tr >> 1) & 7)) << 4) + (lg & 1) * 8) = o2;
                    }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < (TBS > 1 ? 4 : 2); ++i) {
                    const half8 v = *reinterpret_cast<const half8*>(stg + (8 * i + rrow) * 128 + ((rch ^ (((8 * i + rrow) >> 1) & 7)) << 4));
                    *reinterpret_cast<half8*>(gptr + (size_t)(tp * 32 + 8 * i) * a.ldc) = v;
                }
                asm volatile("" ::: "memory");
            }
            return;
        }
        const int rrow = lane >> 3, rch = lane & 7;  // read-back: instruction i takes rows 8 i + rrow, 16-byte chunk rch
        // output addressing: row-major, or — c_block_rows != 0 — blocked by 64 columns (the attention kernel's V^T layout: element (m, n) at
        // C[(n / 64) * c_block_rows * 64 + m * 64 + n % 64]): the wave's 64 columns are one block, its rows 128-byte lines either way
        const long long ldc_eff = a.c_block_rows ? 64 : a.ldc;
        _Float16* gptr = a.C + (size_t)(m0 + wm * 16 * TBS + rrow) * ldc_eff + rch * 8 +
                         (a.c_block_rows ? (size_t)((n0 >> 6) + wn) * (size_t)a.c_block_rows * 64 : (size_t)(n0 + wn * 64));
        // (fetching the column biases ahead of the tile's last stage, out of the epilogue's open latency, was tried: 8 registers held through a
        // stage push hipcc over the register files — 20 to 50 VGPR spills, SGPR spills)
        half4 bias4[FB];
        if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) {
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) bias4[fb] = *reinterpret_cast<const half4*>(a.bias + n0 + wn * 64 + fb * 16 + 4 * lg);
        }
#pragma unroll
        for (int tp = 0; tp < (TBS + 1) / 2; ++tp) {  // 32-token parts of the wave's tokens
#pragma unroll
            for (int half_ = 0; half_ < (TBS > 1 ? 2 : 1); ++half_) {
                const int tb = 2 * tp + half_;
                const int tr = half_ * 16 + q16;  // token row inside the part
                float brow = 0.f;
                if constexpr ((EPI & BH_EPI_BIAS_ROW) != 0) brow = (float)a.bias[m0 + wm * 16 * TBS + tb * 16 + q16];
#pragma unroll
                for (int fb = 0; fb < FB; ++fb) {
                    // pairs: the bias add, the GELU polynomial and the final product are packed fp32 instructions
                    bh_gemm::f32x2 v0 = {acc[tb][fb][0], acc[tb][fb][1]}, v1 = {acc[tb][fb][2], acc[tb][fb][3]};
                    if constexpr ((EPI & BH_EPI_BIAS_COL) != 0) {
                        v0 += bh_gemm::f32x2{(float)bias4[fb][0], (float)bias4[fb][1]};
                        v1 += bh_gemm::f32x2{(float)bias4[fb][2], (float)bias4[fb][3]};
                    }
                    if constexpr ((EPI & BH_EPI_BIAS_ROW) != 0) {
                        v0 += bh_gemm::f32x2{brow, brow};
                        v1 += bh_gemm::f32x2{brow, brow};
                    }
                    if constexpr ((EPI & BH_EPI_GELU) != 0) bh_gemm::gelu_erf2x2(v0, v1);
                    const half4 o = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v1[0], (_Float16)v1[1]};
                    acc[tb][fb] = floatx4{0.f, 0.f, 0.f, 0.f};
                    const int chunk = fb * 2 + (lg >> 1);  // the 16-byte chunk of the row's 128 bytes that holds features 16 fb + 4 lg ..
                    // (chunk order XOR-permuted by (row >> 1) & 7: the 16 rows a half-wave's ds_write_b64 touches land in 16 distinct bank
                    // groups — rows r and r + 1 in the two 128-byte halves of the 256-byte bank window, the eight row pairs on eight chunk
                    // positions; with (row & 7), until round 6, rows r and r + 8 shared one.  Measured: the kernel's SQ_LDS_BANK_CONFLICT
                    // share (5-9 % of SQ_LDS_IDX_ACTIVE) and the forward pass did not move with it: those conflicts are not these writes)
                    *reinterpret_cast<half4*>(stg + tr * 128 + ((chunk ^ ((tr >> 1) & 7)) << 4) + (lg & 1) * 8) = o;
                }
            }
            asm volatile("" ::: "memory");  // (a wave's LDS instructions execute in order, and no other wave touches stg)
#pragma unroll
            for (int i = 0; i < (TBS > 1 ? 4 : 2); ++i) {
                const half8 v = *reinterpret_cast<const half8*>(stg + (8 * i + rrow) * 128 + ((rch ^ (((8 * i + rrow) >> 1) & 7)) << 4));
                half8* p = reinterpret_cast<half8*>(gptr + (size_t)(tp * 32 + 8 * i) * ldc_eff);
                if constexpr ((ABL & 16) != 0) {
                    if (v[0] != (_Float16)12345.f) continue;  // (never true for the bench data: keeps the math alive)
                }
                if constexpr (NT)
                    __builtin_nontemporal_store(v, p);
                else
                    *p = v;
            }
            asm volatile("" ::: "memory");
        }
    };

    // ---- pipeline start
    issue_stage();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    issue_stage();
    abl_started = true;
    using Whole = std::integral_constant<int, TB>;
    auto first_fragments = [&](auto tbs_c) {  // k-step 0 of the stage at the consumer's slot (landed)
        constexpr int TBS = decltype(tbs_c)::value;
        const unsigned char* st = smem + cslot * STAGE_BYTES;
        read_wf(0, st, 0);
#pragma unroll
        for (int tb = 0; tb < TBS; ++tb) read_xt(tbs_c, tb, st, 0);
    };
    if (n_full > 0) first_fragments(Whole{});

    for (int ti = 0; ti < n_full; ++ti) {
        const int t = t_first + ti * G8;
        for (int kt = 0; kt < KT; ++kt) stage(Whole{});
        epilogue(Whole{}, (t / tiles_n) * BM, (t % tiles_n) * BN);
    }
    if (sub_log) {
        const int n0 = (sub_tile % tiles_n) * BN;
        const int m0 = (sub_tile / tiles_n) * BM + sub_part * (BM >> sub_log);
        // (the fragments pre-read behind the last whole tile's last stage followed the whole tile's row map)
        if (sub_log == 1) {
            using Sub = std::integral_constant<int, TB / 2>;
            first_fragments(Sub{});
            for (int kt = 0; kt < KT; ++kt) stage(Sub{});
            epilogue(Sub{}, m0, n0);
        } else {
            using Sub = std::integral_constant<int, TB / 4>;
            first_fragments(Sub{});
            for (int kt = 0; kt < KT; ++kt) stage(Sub{});
            epilogue(Sub{}, m0, n0);
        }
    }
    if constexpr ((ABL & 8) != 0) {  // keep the accumulators alive
        float sum = 0.f;
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) sum += acc[tb][fb][0] + acc[tb][fb][1] + acc[tb][fb][2] + acc[tb][fb][3];
        if (sum == 12345.678f) a.C[tid] = (_Float16)sum;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, bool NT, int ABL = 0>
hipError_t bh_gemm_launch_p16(const BhGemmArgs& a, int n_cu, hipStream_t stream) {
    constexpr size_t smem = 2 * 16 * 4096 + 8 * 4096;
    auto kern = bh_gemm_f16_p16kernel<EPI, NT, ABL>;
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

hipError_t bh_gemm_p16(const BhGemmArgs& a, int epi, bool nontemporal, int mode, hipStream_t s);  // gemm_f16_d.hip
