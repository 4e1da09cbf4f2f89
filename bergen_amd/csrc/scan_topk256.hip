// scan_topk256.hip — fused inner-product + running top-k scan, 256 queries per corpus pass, two waves per SIMD
// (option scan_kernel 3; the default for d in {384, 512, 768, 1024} and candidate lists of 64 / 128 / 256 entries, i.e.
// k <= 248; at d = 1024 a wave carries one 16-query block instead of two: 128 queries per pass).
//
// Same contract, candidate keys and canonical re-score as scan_topk.hip (read that header first); same reference lines
// replaced: torch.mm (models/retrievers/dense.py:81) + torch.topk (modules/retrieve.py:157).
//
// Why.  With ONE wave per SIMD (scan_topk.hip, scan_topk192.hip) everything a wave does besides MFMAs — the LDS-DMA
// issue (the wave is blocked ~100 cycles per instruction while the memory pipeline is saturated), the LDS latency behind
// every stage barrier, the threshold filter — leaves the SIMD's matrix pipe idle: a pass costs the SUM of the stream and
// the matrix work.  Here the workgroup has 8 waves, two per SIMD, each with the query fragments of 32 queries and at most
// 256 registers:
//   * 256 queries ride on one pass of the corpus (HBM bytes per query: 1/2 of the 128-query kernel's, 3/4 of the
//     192-query kernel's); every wave reads every stage from LDS against its own queries;
//   * whatever blocks one wave (DMA issue, rendezvous, filter, candidate code) is covered by its SIMD partner's MFMAs;
//   * waves 0-3 issue the LDS-DMA refill of a stage in one block right behind the stage's rendezvous, at raised priority
//     (the matrix pipe serves the two waves of a SIMD by priority, then age);
//   * the rendezvous of stage s is taken in the MIDDLE of the stage (fragment NF / 2) and waits for the pieces of stage
//     s + 1: a wave runs from one stage into the next without stopping, fragment reads are a rolling pipeline PD deep
//     that never drains (inline-asm ds_read_b128 + counted `s_waitcnt lgkmcnt(PD - 1)`; hipcc waits lgkmcnt(0) for
//     builtin loads in this loop shape).
//
// MFMA: v_mfma_f32_16x16x32_f16, accumulators updated in place by inline asm (the builtin lets hipcc put the result into
// a fresh tuple).  A wave owns NB 16-query blocks x 2 row blocks of a 32-row tile = 2 NB independent accumulator chains:
// two dependent MFMAs on one accumulator lose the back-to-back forwarding as soon as ANY instruction sits between them
// (measured: 48 instead of 32 cycles per 32x32x16 in this loop shape), independent chains do not care.  The wait states
// between the tile's last MFMAs and the first VALU read of their results are an asm statement TIED to the accumulators
// (a bare `s_nop` asm is not ordered against plain register reads: hipcc sank it behind them).
//
// Register budget (launch bounds 512 x 2 -> 256 per lane; hipcc splits the file 128 VGPR + 128 AGPR as soon as an AGPR
// is used): NB * NK32 query fragments of 4 registers (48 at d = 768): the first 32 pinned in the 128 AGPRs (MFMA reads
// operand B from either file), the rest in VGPRs; 4 NB accumulator registers, PD x 4 fragment-read registers, addresses
// and filter state in what is left of the 128 VGPRs.  The scalar file is as tight (102): cold-path addresses are derived
// from OPAQUE copies of the lane / wave index so that hipcc does not hoist ~40 vector and ~20 scalar registers of
// loop-invariant cold-path values across the tile loop.  tests/test_scan256_isa.py checks the generated code: no
// scratch and no vector-register moves in the tile loop, nothing touches a fragment register while its read is in flight.
//
// Thresholds (filter hints: stale or low means less filtering, never a wrong result — but every bound must be VALID):
//   * own bound: the workgroup's KP-th best so far (exclusive compare: its rows come in ascending order);
//   * shared bound of a query: ONE slot per workgroup holds the best score the workgroup has published (the RB-th best of a
//     lane: RB = 1 for lists of 64 / 128, 2 for lists of 256); any T that NS = KP / RB slots reach is reached by KP distinct rows of the
//     corpus.  At every exchange (geometric schedule) a wave REFINES one of its queries — loads the query's 256 slots,
//     builds the largest such T bit by bit (radix select on ballots), publishes it with atomicMax to the query's bound
//     word — and reads the bound words of all its queries (the 256 workgroups cover all queries of the pass at every
//     exchange).  64th largest of 256 per-workgroup maxima ~ the 74th best row seen so far (the 64-slot min-of-max
//     table of scan_topk.hip: ~300th);
//   * bootstrap: before scanning, a workgroup runs BH_BOOT_TILES tiles of its own first run for their maxima only,
//     with two exchanges: the first scanned tile already meets a bound (otherwise every query's buffer fills and is
//     sorted several times at the start: 0.25 ms per pass, independent of the corpus size).
// Candidate path: a tile whose maximum beats the threshold appends its survivors to the per-(workgroup, query) buffer
// (slot = count + hits of the lower lane groups), compaction (bitonic sort, keep KP) when a buffer nears CAP = 2 KP.
//
// Tile distribution: round robin over the first 7/8 of the corpus, the tail in runs claimed from a per-pass counter (an
// atomic add; run lengths halve down to 8 tiles) — the XCDs clock differently under the power limit (3-4 % between
// them) and a launch ends with its slowest workgroup.  The claim is made by wave 7 (no LDS-DMA: its vmcnt is its own),
// waited for on the spot (an answer on its way into a compiler-allocated register gets copied while in flight), parked
// in LDS and picked up by all waves a tile later, before the LDS-DMA issue cursor (up to LEAD tiles ahead) needs it.
// Not at d = 1024 (the ring takes all 160 KiB of LDS).
//
// End of the launch: every wave sorts its queries' buffers (eight 64-key sorts side by side: the shuffles are a latency
// chain) and writes the workgroup's best KP per query; merge_rescore.hip merges the 256 lists of every query of a GROUP
// of passes in one launch (index.hip).  Diagnostics: a.clk gets per-workgroup phase stamps (profiles/scan_phases.py) and,
// in the ABL & 32 instantiation, per-stage stamps of workgroup 0's waves (profiles/scan_timeline.py).
//
// Inline-asm VMEM with an "s" pointer operand starts with s_nop 4: the pointer may come out of v_readfirstlane, a VMEM
// instruction that reads an SGPR written by a VALU instruction needs 5 wait states, and the hazard recognizer does not
// look into inline asm (this read a stale pointer — address 0 — in one instantiation).
#include "bh_device.h"
#include "bh_kernels.h"

// paired launches: tiles between two checkpoints of partner workgroups (a power of two >= 4)
#ifndef BH_PAIR_CKPT
#define BH_PAIR_CKPT 16
#endif

namespace {

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int EPL>
__device__ __forceinline__ void load_list256(u64 (&e)[EPL], const u64* buf, unsigned n, int lane) {
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const unsigned idx = r * 64 + lane;
        e[r] = idx < n ? buf[idx] : 0ull;
    }
}

template <int KP>
__device__ __forceinline__ void sort_candidates256(u64 (&e)[2 * KP / 64], const u64* buf, unsigned n, int lane) {
    load_list256<2 * KP / 64>(e, buf, n, lane);
    bh_wave_sort_desc<2 * KP / 64>(e, lane);
}

// one A fragment (32 rows x 16 dims: 16 bytes per lane) from the LDS ring; completion is tracked by hand
// (s_waitcnt lgkmcnt(n) below), the compiler only sees a register that becomes defined here
__device__ __forceinline__ void lds_read_frag(half8& dst, unsigned addr, int imm) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
}

// acc += A x B on the matrix pipe, accumulator updated IN PLACE (the builtin lets hipcc put the result into a fresh tuple:
// 8 more registers in this kernel, which it took from the pinned query fragments).  The accumulators live in VGPRs: the
// filter reads them with plain VALU instructions (from AGPRs every score costs a v_accvgpr_read first).  `first` starts a
// tile: C = 0 as an inline constant, no zeroing pass.  Hazards are the caller's: consecutive calls use different
// accumulators (>= 3 MFMAs between two on the same one), wait states before the first VALU read of a result.
__device__ __forceinline__ void mfma16_inplace(floatx4& acc, const half8& a, const half8& b, bool b_in_agpr, bool first) {
    // (both flags are compile-time constants once the caller's loops are unrolled)
    if (first) {
        if (b_in_agpr)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b));
        else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
    } else {
        if (b_in_agpr)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
        else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}

// max of three without the NaN canonicalisation fmaxf() drags in (one v_max_f32 x, x, x per input): scores are finite
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

}  // namespace

// NK32 = padded dim / 32 (k-steps of the 16x16x32 MFMA per row); KP = candidate list length; LS = 128-byte lines per
// stage; R = ring depth in stages; PD = fragment reads in flight per wave; NT = non-temporal policy on the corpus stream;
// ABL (bench-only, results invalid unless 0 or 32): bit flags  1 = no filter / candidate code, 2 = no LDS fragment reads,
// 4 = no MFMA, 8 = no LDS-DMA refill inside the tile loop (the ring keeps its first contents), 16 = no stage barrier;
// 32 = production path + s_memtime stamps of workgroup 0's waves (timeline diagnostics into a.clk, see BH_TL_*).
// LM (who issues the LDS-DMA refill): 0 = all eight waves, half a stage's lines each; 1 = waves 0-3 only, at raised
// priority; 2 = waves 4-7 only, at raised priority; 3 = waves 0-3, no priority change.
// SCHED: 0 = rendezvous at the top of a stage, refill spread over the stage; 1 = rendezvous in the middle of the stage,
// refill in one block right behind it; 2 = rendezvous in the middle, refill spread over the second half.
template <int NK32, int KP, int LS, int R, int PD, bool NT, int ABL = 0, int LM = 1, int SCHED = 1, int NBUF = PD, int NB = 2>
__global__ void __launch_bounds__(512, 2) bh_scan_topk256_kernel(BhScanArgs a) {
    // diagnostics block of this workgroup (a.clk, 8 words: 100 MHz ticks at entry / loop start / loop end / exit, shader
    // cycles at loop start / end, candidates wave 0 holds at the end); every stamp is stored where it is taken
    if (a.clk != nullptr && threadIdx.x == 0) a.clk[8 * blockIdx.x + 0] = wall_clock64();  // (paired launch: the block of the REAL index; the other stamps go by b below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NK32 * 32;
    constexpr int LINES = D / 64;
    static_assert(LINES % LS == 0 && LS % 2 == 0, "a stage is a whole number of line pairs and divides the row");
    static_assert(R >= 3, "early rendezvous needs three ring slots");
    constexpr int S = LINES / LS;          // stages per 32-row tile
    constexpr int NF = LS * 4;             // A fragments per stage: (two k-steps) x (two 16-row blocks) per line
    constexpr int BF = SCHED == 0 ? 0 : NF / 2;  // fragment of a stage at which its rendezvous is taken
    constexpr bool SPLIT = LM == 0;
    constexpr int DPW = SPLIT ? LS / 2 : LS;  // LDS-DMA instructions per (loading) wave and stage
    static_assert(PD <= NF / 2 && NF % NBUF == 0 && NBUF >= PD, "fragment pipeline depth divides the stage; reads of the next stage start behind the rendezvous");
    constexpr int STAGE_BYTES = 32 * LS * 128;
    // dynamic tile distribution needs one word of LDS behind the ring (the d = 1024 ring takes all 160 KiB: static there)
    constexpr bool DYN = R * STAGE_BYTES + 16 <= 160 * 1024;
    // the LDS-DMA issue cursor runs LEAD tiles ahead of the MFMAs: the run after the current one must be known by then
    constexpr int LEAD = (R - 1 + S - 1) / S;
    constexpr int CH = bh_scan256_chunk_tiles(D);
    static_assert(!DYN || CH >= LEAD + 3, "a run is claimed LEAD + 2 tiles before the current one ends");
    constexpr int CAP = 2 * KP;
    constexpr int EPLC = CAP / 64;
    constexpr int EPLK = KP / 64;
    // NB = 16-query blocks per wave: 2 (256 queries per workgroup) wherever their fragments fit, 1 at d = 1024
    constexpr int BQ = 8 * 16 * NB;
    constexpr int ROW_BYTES = D * 2;
    // shared bound: NS slots (= workgroups) that each vouch for RB rows at or above T give NS * RB = KP distinct rows.  A lane
    // tracks its RB best appended scores in registers: lists of 256 use (128 slots, RB = 2), lists of 128 (128 slots, RB = 1:
    // nothing to track, the slot is fed at append time) — until round 3 both used 64 slots with RB = KP / 64, whose eight
    // tracking registers pushed the d = 768 instantiations' cold paths into scratch
    constexpr int RB = KP >= 256 ? KP / 128 : 1;
    constexpr int NS = KP / RB;
    static_assert(NS * RB == KP && NS <= BH_SLOTS256, "slots x rows per slot = list length");
    // fragments in the accumulator half of the register file (all 128 of its registers; the MFMA accumulators are VGPRs)
    constexpr int NPIN_A = NB * NK32 < 32 ? NB * NK32 : 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3;   // row group of the LDS-DMA pattern (rows 8 wr .. 8 wr + 7)
    const int wl = wave >> 2;  // LM 0: which half of a stage's lines this wave fetches; else: loader half or not
    const bool loader = SPLIT || (LM == 2 ? wl == 1 : wl == 0);
    const int line0 = SPLIT ? wl * DPW : 0;  // first line of a stage this wave fetches
    // The matrix pipe of a SIMD serves its two waves by priority, then age; the wave that also issues the refill (each
    // issue blocks it for ~100 cycles while the memory pipeline is saturated) gets the priority.
    if (LM == 1 || LM == 2) {
        if (loader) __builtin_amdgcn_s_setprio(1);
    }
    // ABL bit 128 is not an ablation either: the PAIRED launch (index.hip option pair256).  One launch serves TWO 256-query passes
    // with the same grid: workgroup B works for pass (B >> 3) & 1 as workgroup (B & 7) | (B >> 4) << 3 of a launch of half the
    // grid — its own queries, thresholds, candidate buffers and lists, exactly as if the two passes had been launched one after
    // the other on half the chip each — and B ^ 8, which the dispatcher puts on the same XCD, walks the same round-robin tiles
    // for the other pass: whichever of the two asks for a line first takes it from HBM, the partner finds it in the XCD's L2.
    // That only holds while the two stay within the L2's reach of each other (a few tiles): every 16 tiles they wait for each
    // other through a progress word (pure pacing: a stale or missing value only changes timing, a timeout switches it off).
    constexpr bool PAIRED = (ABL & 128) != 0;
    static_assert(!PAIRED || R == 3 || LM == 1, "the pacing load of wave 7 must not disturb a loader's vmcnt arithmetic");
    // Inside the kernel a paired workgroup goes by b = pass * 128 + its index in the pass's half grid (the launcher insists on a
    // grid of 256): the buffers of the two passes lie behind each other in exactly that order, the slots 128 .. 255 of the
    // second pass's tables are as good as 0 .. 127, and the only things that need the pass are the tile distribution's start,
    // the query pointer (both before the loop) and the pass's threshold block.
    int G = (int)gridDim.x, b = (int)blockIdx.x;
    int b_pass = b;  // index in the pass's grid: the tile distribution
    if constexpr (PAIRED) {
        const int bc = b, h = (bc >> 3) & 1;
        G >>= 1;
        b_pass = (bc & 7) | ((bc >> 4) << 3);
        b = h * G + b_pass;
        a.qtile = (h != 0 && a.qtile2 != nullptr) ? a.qtile2 : a.qtile + (size_t)h * (8 * 16 * NB) * (NK32 * 32);
        a.gthr += (size_t)h * ((size_t)(8 * 16 * NB) * (BH_SLOTS256 + 1) + 4);
        if (h != 0) a.nq_valid[0] = a.nq_valid[1];
    }
    // A wave without a single existing query (a tile of fewer than 256 queries: the remainder of a query set, balanced over the two
    // passes of a paired launch by index.hip) leaves the matrix pipe and the LDS read ports to the waves that have some: no fragment
    // reads, no MFMAs, no filter.  It still issues its lines of every refill, meets every rendezvous and (wave 7) paces and claims.
    // (compiled out — every wave scans — in the filter pass of the exact fall-back, whose unused queries carry a +inf threshold)
    constexpr bool IDLE_WAVES = (ABL & 64) == 0;
    const int nq_here = __builtin_amdgcn_readfirstlane(a.nq_valid[0]);
    const bool wave_on = !IDLE_WAVES || nq_here <= 0 || wave * (16 * NB) < nq_here;  // wave-uniform; tested ONCE, in front of the tile loop
    auto gthr_pass = [&]() -> unsigned* { return a.gthr; };  // (paired: moved to the pass's block above)
    const int q16 = lane & 15, lg = lane >> 4;

    const int n_tiles = (int)a.n_tiles;  // < 2^27 (n_rows < 2^32)
    // The workgroup's tiles, one RUN at a time: tile j of a run is run_base + j * run_stride.
    //   static  : one run, tiles b, b + G, b + 2 G, ... (round robin)
    //   dynamic : the first 7/8 of the corpus round robin (one long run), the rest in runs of consecutive tiles handed out
    //             by the pass's claim counter (an atomic add of the run length; the lengths halve from half a workgroup's
    //             share of the tail down to CH tiles).  Workgroups do not run at one speed (the XCDs clock differently
    //             under the power limit: 3-4 % between them, plus the luck of the candidate path: +-5 % over an eighth
    //             of the corpus), and a launch ends with its slowest workgroup.  A corpus too small for a round-robin
    //             run of 4 CH tiles (< 300 k rows on 256 workgroups) stays static: chunks of CH tiles would balance it
    //             worse than round robin does.
    const int rr = DYN && a.dyn_tiles != 0 ? bh_scan256_round_robin_tiles(n_tiles, G, D) : 0;  // (the host starts the claim counter behind them)
    const bool dyn = rr > 0;
    int run_base = b_pass, run_stride = G, run_len = n_tiles > b_pass ? (n_tiles - b_pass + G - 1) / G : 0;
    int nxt_base = 0, nxt_len = 0;
    int claim_len = CH;  // tiles the next claim asks for
    if (dyn) {
        run_len = rr;
        claim_len = (n_tiles - rr * G) / (2 * G);
        if (claim_len < CH) claim_len = CH;
    }
    const unsigned n_rows32 = (unsigned)a.n_rows;
    u64* cand_wg = a.cand + (size_t)b * BQ * CAP;
    u64* part_wg = a.partial + (size_t)b * BQ * KP;

    // ---- queries -> registers (B fragments): lane (q16, lg), block nb, k-step s: the 8 halfs at k = 32 s + 8 lg of
    // query (wave*2 + nb)*16 + q16
    half8 qf[NB][NK32];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const _Float16* qrow = a.qtile + (size_t)((wave * NB + nb) * 16 + q16) * D;
#pragma unroll
        for (int s = 0; s < NK32; ++s) qf[nb][s] = *reinterpret_cast<const half8*>(qrow + 32 * s + 8 * lg);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < NK32; ++s) {
            if (nb * NK32 + s < NPIN_A)
                asm volatile("" : "+a"(qf[nb][s]));
            else
                asm volatile("" : "+v"(qf[nb][s]));
        }

    float thr[NB];       // candidate iff score > thr (per lane and block = per query)
    unsigned cnt[NB];    // entries in the query's candidate buffer (identical in the four lanes of a query)
    float best[NB][RB];  // this lane's RB best appended scores, descending
    int next_poll = 0;
    // ABL bit 64 is not an ablation but the FILTER PASS of the exactness fall-back (certify.hip, index.hip; the 128-query
    // kernel's ABL = 5 on this kernel's 256-query tile): the same stream, fragment pipeline and MFMA loop, but a FIXED
    // threshold per query (a.fix_thr: a row qualifies iff its MFMA score >= it) and no top-k state — no bootstrap, no bounds,
    // no candidate buffers, no lists: every qualifying row index is appended to the query's list a.fix_rows (few rows: the
    // top k and whatever lies within rounding error of the k-th score).
    constexpr bool FILT = (ABL & 64) != 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        thr[nb] = -__builtin_inff();
        if constexpr (FILT) {
            // the scan's compare is exclusive: the next float below the inclusive threshold (+inf = unused query of the tile:
            // nothing qualifies; -inf = the query's list was not full: every row does)
            const float x = a.fix_thr[(wave * NB + nb) * 16 + q16];
            thr[nb] = (x > -__builtin_inff() && x < __builtin_inff()) ? bh_unordf(bh_ordf(x) - 1u) : x;
        }
        cnt[nb] = 0;
#pragma unroll
        for (int r = 0; r < RB; ++r) best[nb][r] = -__builtin_inff();
    }

    // ---- LDS-DMA source pattern: lanes 8j..8j+7 fetch the eight 16-byte chunks of one line of row 8 wr + j, chunk
    // order XOR-permuted by g(row) = (row >> 1) & 7 (conflict-free for the 16x16x32 fragment reads, scan_topk192.hip)
    const int ld_row = 8 * wr + (lane >> 3);
    const int ld_g = (ld_row >> 1) & 7;
    const unsigned ld_off = (unsigned)ld_row * ROW_BYTES + (unsigned)(((lane & 7) ^ ld_g) << 4) + (unsigned)(line0 * 128);
    // fragment read addresses in the stage the READ cursor is in: fragment f of a stage = line f >> 2, k-step parity
    // sb = (f >> 1) & 1, row block rb = f & 1; lane reads row rb*16 + q16, chunk 4 sb + lg.  g(row) does not depend on rb:
    // the second row block is the first one's address + 2048 (an immediate), one register per parity.
    unsigned rdb[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
        rdb[sb] = (unsigned)((q16 >> 3) * 1024 + (q16 & 7) * 128 + (((4 * sb + lg) ^ ((q16 >> 1) & 7)) << 4));

    // Everything the cold paths derive from the lane id is computed from an OPAQUE copy of it: otherwise hipcc hoists the
    // shuffle partners, sort directions and per-query predicates out of the tile loop and keeps ~40 vector and ~30
    // scalar registers alive across the hot path.
    auto opaque_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    // (same for the wave index: the per-wave base addresses of the candidate buffers, slot tables and bounds would
    // otherwise sit in ~20 scalar registers across the hot path — past the 102 there are, into lanes of a vector register)
    auto opaque_wave = [&]() {
        int w = wave;
        asm volatile("" : "+s"(w));
        return w;
    };
    // a pointer for an "s" operand of inline asm: uniform by construction, made uniform for the compiler too.  The
    // v_readfirstlane pair may sit right in front of the asm: a VMEM instruction that reads an SGPR a VALU instruction wrote
    // needs 5 wait states, and the hazard recognizer does not look into inline asm — such asm starts with s_nop 4
    auto sgpr_ptr = [](const unsigned* p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const unsigned*)(((unsigned long long)hi << 32) | lo);
    };
    auto compact = [&](int nb, int qq) {
        const int lane = opaque_lane();
        const int wave = opaque_wave();
        const int q16 = lane & 15;
        const int qi = (wave * NB + nb) * 16 + qq;
        const unsigned n = __builtin_amdgcn_readlane(cnt[nb], qq);
        u64* buf = cand_wg + (size_t)qi * CAP;
        u64 e[EPLC];
        sort_candidates256<KP>(e, buf, n, lane);
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

    if (a.clk != nullptr && tid == 0) {
        a.clk[8 * b + 1] = wall_clock64();
        a.clk[8 * b + 4] = __builtin_readcyclecounter();
    }
    // Bootstrap of the thresholds.  Without a bound every score of a workgroup's first tiles is a candidate: the buffers
    // of all its queries fill and are sorted several times before the shared bounds take hold (~0.25 ms per pass, a third
    // of a pass over one of eight shards).  So the workgroup first runs nboot tiles for their per-query maxima only
    // (published to the slot table, no candidates), exchanges bounds, and then scans its tiles.  The bootstrap tiles are
    // tiles of the workgroup's OWN first run, spread evenly over it (0.15 % more corpus bytes at 21 M rows; 4 tiles measured best: 2 / 3 / 4 / 8 / 16 give 1.043 / 1.034 / 1.040 / 1.071 /
    // 1.166 ms per pass over an eighth of the corpus, 7.22 / - / 7.21 / 7.23 / 7.29 ms over all of it): a slot's value
    // is the score of a row that this workgroup — and no other — appends when the scan proper comes by, so distinct slots
    // still stand for distinct rows and the bound stays valid.
    const int nboot = (!a.share || FILT) ? 0 : run_len < BH_BOOT_TILES ? run_len : BH_BOOT_TILES;
    const int boot_step = (run_len / BH_BOOT_TILES > 1 ? run_len / BH_BOOT_TILES : 1) * run_stride;  // in tiles
    unsigned* lds_claim = reinterpret_cast<unsigned*>(smem + R * STAGE_BYTES);
    if constexpr (PAIRED) {
        static_assert(DYN, "the partner's progress word lives behind the claim word");
        if (tid == 448) {  // (wave 7 reads these: its own writes, in program order)
            lds_claim[1] = 0u;
            *reinterpret_cast<unsigned long long*>(lds_claim + 2) =
                (a.pair_window != 0 && a.progress != nullptr) ? (unsigned long long)(a.progress + blockIdx.x) : 0ull;
        }
    }
    if (run_len > 0) {
        const unsigned char* corpus = reinterpret_cast<const unsigned char*>(a.corpus);
        // issue cursor: the tile it stands on, the tile step and the tiles left of ITS run (phase 0: the bootstrap tiles;
        // 1: the run the MFMAs are in; 2: the run after it, known by then), stage of the tile, ring slot.  Past the last
        // run it stays on the last tile: harmless re-fetches, uniform vmcnt arithmetic.
        int it_phase, it_tile, it_step, it_left;
        if (nboot > 0) {
            it_phase = 0;
            it_tile = run_base;
            it_step = boot_step;
            it_left = nboot;
        } else {
            it_phase = 1;
            it_tile = run_base;
            it_step = run_stride;
            it_left = run_len;
        }
        int ip = 0;
        int islot = 0;
        auto issue_line = [&](int j) {
            const unsigned char* src = corpus + (size_t)(unsigned)it_tile * (size_t)(32 * ROW_BYTES) + (size_t)(ip * LS * 128) + ld_off;
            unsigned char* dst = smem + islot * STAGE_BYTES + line0 * 4096 + wr * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 128),
                                             (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0, NT ? 2 : 0);
        };
        auto advance_cursor = [&]() {
            if (++ip == S) {
                ip = 0;
                if (it_left > 1) {
                    --it_left;
                    it_tile += it_step;
                } else if (it_phase == 0) {  // bootstrap done: the workgroup's first run
                    it_phase = 1;
                    it_tile = run_base;
                    it_step = run_stride;
                    it_left = run_len;
                } else if (it_phase == 1 && nxt_len > 0) {
                    it_phase = 2;
                    it_tile = nxt_base;
                    it_step = 1;
                    it_left = nxt_len;
                }
            }
            if (++islot == R) islot = 0;
        };
#pragma unroll
        for (int p = 0; p < R - 1; ++p) {
            if (loader) {
#pragma unroll
                for (int j = 0; j < DPW; ++j) issue_line(j);
            }
            advance_cursor();
        }

        // ---- stage 0 landed (own pieces, then everybody's); start the fragment pipeline
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * DPW) : "memory");
        int cslot = 0;
        half8 ag[NBUF];  // fragment g of a stage lives in ag[g % NBUF]; NBUF = PD + 1 leaves one step between a buffer's last MFMA and its refill
        if constexpr (!(ABL & 2)) {
#pragma unroll
            for (int f = 0; f < PD; ++f) lds_read_frag(ag[f], rdb[(f >> 1) & 1], (f >> 2) * 4096 + (f & 1) * 2048);
        } else {
#pragma unroll
            for (int f = 0; f < NBUF; ++f) ag[f] = qf[0][f];
        }

        bool boot = nboot > 0;
        int tj = 0;  // tile of the bootstrap / of the current run
        int i = 0;   // tiles scanned so far (threshold exchange schedule)
        // the bookkeeping between two tiles (bootstrap / run / exchange counters, the paired launch's checkpoints, the claim of the next
        // run, the switch to it): shared by the scanning loop and the idle loop below; true = the workgroup's tiles are done
        auto next_tile = [&]() -> bool {
            ++tj;
            if (boot) {
                if (tj == nboot) {
                    boot = false;
                    tj = 0;
                }
                return false;
            }
            ++i;
            if (i > n_tiles) return true;  // (cannot happen: a workgroup never scans more tiles than there are; keeps a logic error from hanging the GPU)
            if constexpr (PAIRED) {
                // Checkpoint every BH_PAIR_CKPT (16) tiles of the round-robin run: wave 7 publishes the count and starts the load of the
                // partner's word (LDS-DMA into the word behind the claim word: no register is in flight; the rendezvous of
                // the next tile waits for it with the refill); one tile later it reads the word and, if the partner has not
                // reached the checkpoint, polls until it has.  The other waves wait at the next rendezvous meanwhile.
                // Nothing of this lives in registers across the tile loop: the workgroup's progress pointer is parked in LDS
                // (null = not paced), the partner's word is 32 bytes away from it (bc ^ 8; the block is 64-byte aligned).
                if (run_stride != 1 && wave == 7 && (i & (BH_PAIR_CKPT - 2)) == 0 && i > 1) {
                    const unsigned long long ppv = *(volatile unsigned long long*)(lds_claim + 2);
                    unsigned pv = *(volatile unsigned*)(lds_claim + 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the fragment reads in flight land too: no register moves)
                    const unsigned long long pp = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ppv >> 32)) << 32) |
                                                  (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ppv);
                    if (pp != 0ull) {
                        unsigned* mine = reinterpret_cast<unsigned*>(pp);
                        const unsigned* theirs = reinterpret_cast<const unsigned*>(pp ^ 32ull);
                        if ((i & 1) == 0) {
                            if (opaque_lane() == 0) {
                                __hip_atomic_fetch_max(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)theirs,
                                                                 (__attribute__((address_space(3))) void*)(lds_claim + 1), 4, 0, 16);
                            }
                        } else {
                            int spins = 0;
                            while ((unsigned)__builtin_amdgcn_readfirstlane(pv) + 1u < (unsigned)i) {
                                if (++spins > 8192) {  // the partner is not coming (not resident?): stop pacing
                                    if (opaque_lane() == 0) *(volatile unsigned long long*)(lds_claim + 2) = 0ull;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(8);
                                if (opaque_lane() == 0)
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)theirs,
                                                                     (__attribute__((address_space(3))) void*)(lds_claim + 1), 4, 0, 16);
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                pv = *(volatile unsigned*)(lds_claim + 1);
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            }
                        }
                    }
                }
            }
            if constexpr (DYN) {
                // The run after the current one: claimed LEAD + 2 tiles before the current run's end by wave 7 (it does not
                // issue LDS-DMA, its vmcnt is its own; it waits for the answer on the spot — an answer on its way into a
                // register the compiler allocates is not safe — which holds the workgroup up for ~1.5 us, a handful of
                // times per launch), parked in LDS, read by every wave one tile later (the rendezvous between make it
                // visible) — a tile before the issue cursor, up to LEAD tiles ahead, leaves the run.
                if (dyn && run_len >= CH) {  // (a shorter run is the corpus's last, clipped: nothing follows)
                    const int left = run_len - tj;
                    if (left == LEAD + 2 && wave == 7) {
                        unsigned* ctr = gthr_pass() + (size_t)BQ * (BH_SLOTS256 + 1);
                        unsigned got = (unsigned)claim_len;
                        const unsigned off = 0u;
                        if (opaque_lane() == 0) {
                            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %0, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "+v"(got) : "v"(off), "s"(sgpr_ptr(ctr)) : "memory");
                            *(volatile unsigned*)lds_claim = got;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    } else if (left == LEAD + 1) {
                        const unsigned got = *(volatile unsigned*)lds_claim;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the fragment reads in flight land too: no register moves)
                        // (the host starts the counter at the first tile of the claimed part of the corpus)
                        const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane(got);
                        if (t0 < (unsigned)n_tiles) {
                            nxt_base = (int)t0;
                            nxt_len = n_tiles - nxt_base < claim_len ? n_tiles - nxt_base : claim_len;
                        }
                        claim_len = claim_len >= 2 * CH ? claim_len >> 1 : CH;
                    }
                }
            }
            if (tj == run_len) {
                if constexpr (PAIRED) {  // the partner must not wait for a workgroup that has left the shared tiles
                    if (run_stride != 1 && wave == 7) {
                        const unsigned long long ppv = *(volatile unsigned long long*)(lds_claim + 2);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        const unsigned long long pp = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ppv >> 32)) << 32) |
                                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ppv);
                        if (pp != 0ull && opaque_lane() == 0)
                            __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(pp), 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (nxt_len == 0) return true;
                run_base = nxt_base;
                run_stride = 1;
                run_len = nxt_len;
                nxt_len = 0;
                tj = 0;
                if (it_phase == 2) it_phase = 1;  // (the issue cursor is ahead of the tile just finished: it stands in the new run)
            }
            return false;
        };
        if (wave_on) {
        for (;;) {
            const int tile_id = run_base + tj * (boot ? boot_step : run_stride);
            // four independent accumulator chains (row block x query block): consecutive MFMAs never share one, so the
            // stream does not depend on the back-to-back forwarding of a single chain (any instruction between two
            // dependent MFMAs costs ~60 cycles: measured 48 instead of 32 cycles per 32x32x16 MFMA in this loop shape)
            floatx4 acc[2][NB];  // (every element is first written by the tile's first MFMA on it, with C = 0)
            static_assert(NB == 1 || NB == 2, "one or two query blocks per wave");

#pragma unroll
            for (int part = 0; part < S; ++part) {
                unsigned long long tl0 = 0, tl1 = 0, tl2 = 0, tl_dma = 0;
                const bool tl_on = (ABL & 32) != 0 && b == 0 && i >= BH_TL_TILE0 && i < BH_TL_TILE0 + BH_TL_TILES;
                const int nslot = cslot + 1 == R ? 0 : cslot + 1;
                auto rendezvous = [&]() {
                        // Rendezvous of stage s, taken at its fragment BF.  Own pieces of stage s+1 landed (R-3 younger
                        // stages stay in flight), then everybody's did, and everybody has left stage s-1, whose slot this
                        // stage's refill overwrites.  With BF in the middle of the stage a wave runs from one stage into
                        // the next without stopping: the filter / candidate code at a tile's end overlaps the other waves'
                        // MFMAs, up to half a stage of skew is absorbed.
                        if (tl_on) tl0 = __builtin_readcyclecounter();
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * DPW) : "memory");
                        if (tl_on) tl1 = __builtin_readcyclecounter();
                        if constexpr (!(ABL & 16)) asm volatile("s_barrier" ::: "memory");
                        if (tl_on) tl2 = __builtin_readcyclecounter();
                        if (SCHED == 1 && loader && !(ABL & 8)) {
                            // the whole refill in one block right behind the barrier: the loading wave stalls here while its
                            // SIMD partner has the most MFMA work ahead of it
#pragma unroll
                            for (int j = 0; j < DPW; ++j) issue_line(j);
                            if (tl_on) tl_dma = __builtin_readcyclecounter() - tl2;
                        }
                };
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int ks = (part * LS + (f >> 2)) * 2 + ((f >> 1) & 1);  // k-step of 32 dims
                    const int rb = f & 1;
                    if (f == BF) rendezvous();
                    if constexpr (!(ABL & 2)) {
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PD - 1));
                        asm volatile("" : "+v"(ag[f % NBUF]));
                    }
                    if constexpr ((ABL & 4) != 0) {
                        if constexpr (!(ABL & 2)) asm volatile("" ::"v"(ag[f % NBUF]));
                    } else {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            mfma16_inplace(acc[rb][nb], ag[f % NBUF], qf[nb][ks], nb * NK32 + ks < NPIN_A, part == 0 && f < 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(ABL & 2)) {
                        const int nf = f + PD;
                        if (nf == NF) {  // the read cursor enters the next stage
                            const int delta = nslot == 0 ? -(R - 1) * STAGE_BYTES : STAGE_BYTES;
#pragma unroll
                            for (int x = 0; x < 2; ++x) rdb[x] += (unsigned)delta;
                        }
                        const int rf = nf < NF ? nf : nf - NF;
                        lds_read_frag(ag[nf % NBUF], rdb[(rf >> 1) & 1], (rf >> 2) * 4096 + (rf & 1) * 2048);
                    }
                    if constexpr (SCHED != 1) {
                        // the stage's refill spread behind the barrier: one instruction every SP fragments
                        constexpr int SP = (NF - BF) / DPW;
                        if (f >= BF && (f - BF) % SP == SP / 2 && (f - BF) / SP < DPW) {
                            if (loader && !(ABL & 8)) issue_line((f - BF) / SP);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                advance_cursor();
                cslot = nslot;
                if (tl_on && lane == 0) {
                    bh_u64* rec = a.clk + 8 * G + ((size_t)(wave * BH_TL_TILES + (i - BH_TL_TILE0)) * S + part) * 5;
                    rec[0] = tl0;
                    rec[1] = tl1;
                    rec[2] = tl2;
                    rec[3] = __builtin_readcyclecounter();
                    rec[4] = tl_dma;
                }
            }

            // ---- threshold filter: lane (q16, lg) holds rows 16 rb + 4 lg + v of queries 16 nb + q16
            const unsigned row0 = (unsigned)tile_id * 32u;  // n_rows < 2^32 (bh_index_create)
            // last MFMAs -> VALU reads of their results: the hardware does not interlock these, and the wait must be TIED to the
            // accumulators (a bare asm nop is not ordered against plain register reads: hipcc sank it behind them and the
            // filter saw the last block's scores one k-step short)
            if constexpr (!(ABL & 4))
            {
                if constexpr (NB == 2)
                    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
                else
                    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]));
            }
            else {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[rb][nb][v] = 0.f;
            }
            bool over = false;
            float tile_max[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float m = max3_raw(acc[0][nb][0], acc[0][nb][1], acc[0][nb][2]);
                m = max3_raw(m, acc[0][nb][3], acc[1][nb][0]);
                m = max3_raw(m, acc[1][nb][1], acc[1][nb][2]);
                m = max3_raw(m, acc[1][nb][3], acc[1][nb][3]);
                if constexpr ((ABL & 1) != 0) {
                    asm volatile("" ::"v"(m));
                    m = -__builtin_inff();
                }
                tile_max[nb] = m;
                over = over || (m > thr[nb]);
            }
            // ---- cold paths.  The fragment reads of the next tile's first k-steps are in flight and the compiler does not
            // know it: the cold block starts by letting them land (code in there may move registers around).  The append of
            // a tile's few survivors keeps the fragments alive and costs a few hundred cycles, which the half stage of slack
            // between two rendezvous absorbs; the rare heavy parts (compaction of a full candidate buffer, the threshold
            // exchange) run with the fragments DEAD and read them again at the end (they are still in the ring): 16
            // registers more for that code.
            const bool any_hit = !boot && __builtin_amdgcn_ballot_w64(over) != 0ull;
            // (bootstrap: two exchanges, two tiles apart — the first publishes this wave's refinement, the second reads what the
            // other workgroups refined meanwhile, so that the first scanned tile already meets a bound)
            const bool do_poll = a.share && (boot ? (tj == nboot - 1 || tj == nboot - 3) : i >= next_poll);
            if (__builtin_expect(any_hit || do_poll || boot, 0)) {
                if constexpr (!(ABL & 2)) {
                    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
                    for (int f = 0; f < NBUF; ++f) asm volatile("" : "+v"(ag[f]));
                }
                const int lane_c = opaque_lane();
                const int wave = opaque_wave();  // (shadows the kernel's: cold-path addresses are not hoisted)
                const int q16 = lane_c & 15, lg = lane_c >> 4;
                if (boot && row0 + 32u <= n_rows32 && !(ABL & 1)) {
                    // bootstrap tile: this lane's best score of its 8 rows goes to the slot table, nothing else happens
                    // (a tile holding padding rows is skipped: their zero scores are no rows of the corpus)
                    // (candidate lists of NS * RB entries: the slot needs a score that RB rows of this lane reach)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        float pubv = tile_max[nb];
                        if constexpr (RB > 1) {
                            float top[RB];
#pragma unroll
                            for (int r = 0; r < RB; ++r) top[r] = -__builtin_inff();
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    float x = acc[rb][nb][v];
#pragma unroll
                                    for (int r = 0; r < RB; ++r) {
                                        const float hi = fmaxf(top[r], x);
                                        x = fminf(top[r], x);
                                        top[r] = hi;
                                    }
                                }
                            pubv = top[RB - 1];
                        }
                        __hip_atomic_fetch_max(gthr_pass() + (size_t)((wave * NB + nb) * 16 + q16) * BH_SLOTS256 + (b & (BH_SLOTS256 - 1)), bh_ordf(pubv),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if constexpr (FILT) {
                    if (any_hit) {  // filter pass: list the qualifying rows of this tile (rare: one atomic per listed row)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int qi = (wave * NB + nb) * 16 + q16;
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const float sv = acc[rb][nb][v];
                                    const unsigned row = row0 + (unsigned)(rb * 16 + v) + 4u * (unsigned)lg;
                                    if (sv > thr[nb] && row < n_rows32) {
                                        const unsigned slot = atomicAdd(a.fix_cnt + qi, 1u);
                                        if (slot < a.fix_cap) a.fix_rows[(size_t)qi * a.fix_cap + slot] = row;
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                        }
                    }
                } else if (any_hit) {
                    // (1) append survivors (room for 32 entries per query is guaranteed by (2) of the previous visit);
                    //     slot = count + hits of the same query in the lower lane groups.  Hits are rare: the per-row test is
                    //     a fall-through branch, the work sits out of line.
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        u64* cand_blk = cand_wg + (size_t)(wave * NB + nb) * 16 * CAP;  // uniform base, 32-bit per-lane index
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float sv = acc[rb][nb][v];
                                if (__builtin_expect(__builtin_amdgcn_ballot_w64(sv > thr[nb]) != 0ull, 0)) {
                                    const unsigned row = row0 + (unsigned)(rb * 16 + v) + 4u * (unsigned)lg;
                                    const bool hit = (sv > thr[nb]) && (row < n_rows32);
                                    const u64 hm = __builtin_amdgcn_ballot_w64(hit);
                                    const unsigned h0 = (unsigned)(hm >> q16) & 1u, h1 = (unsigned)(hm >> (q16 + 16)) & 1u;
                                    const unsigned h2 = (unsigned)(hm >> (q16 + 32)) & 1u, h3 = (unsigned)(hm >> (q16 + 48)) & 1u;
                                    const unsigned below = lg == 0 ? 0u : lg == 1 ? h0 : lg == 2 ? h0 + h1 : h0 + h1 + h2;
                                    if (hit) {
                                        cand_blk[(unsigned)q16 * CAP + cnt[nb] + below] = bh_make_key(sv, row);
                                        if constexpr (RB == 1) {
                                            // lists of 64: the slot holds the best score of its workgroups; published right
                                            // here (fire and forget) instead of tracked in a register
                                            if (a.share)
                                                __hip_atomic_fetch_max(gthr_pass() + (size_t)((wave * NB + nb) * 16 + q16) * BH_SLOTS256 + (b & (BH_SLOTS256 - 1)), bh_ordf(sv),
                                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        } else {
                                            float x = sv;
#pragma unroll
                                            for (int r = 0; r < RB; ++r) {
                                                const float hi = fmaxf(best[nb][r], x);
                                                x = fminf(best[nb][r], x);
                                                best[nb][r] = hi;
                                            }
                                        }
                                    }
                                    cnt[nb] += h0 + h1 + h2 + h3;
                                }
                                __builtin_amdgcn_sched_barrier(0);  // one row at a time: keeps the live ranges of this cold code short
                            }
                    }
                }
                // (2) make room for the next visit: a tile adds at most 32 entries per query
                u64 need[NB];
                bool heavy = do_poll;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    need[nb] = __builtin_amdgcn_ballot_w64(cnt[nb] > (unsigned)(CAP - 32)) & 0xffffull;
                    heavy = heavy || need[nb] != 0ull;
                }
                if (__builtin_expect(heavy, 0)) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        while (need[nb] != 0ull) {
                            const int qq = __builtin_ctzll(need[nb]);
                            need[nb] &= need[nb] - 1;
                            compact(nb, qq);
                        }
                    }
                // ---- threshold exchange through the slot table (filter hint only), geometric schedule (scan_topk.hip)
                if (do_poll) {
                    if (!boot) next_poll = i + 1 + (i >> 1);
                    if constexpr (RB > 1) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int q = (wave * NB + nb) * 16 + q16;
                            const float mine = best[nb][RB - 1];
                            if (mine > -__builtin_inff()) {  // (re-published at every exchange: ~20 atomics per lane and pass)
                                __hip_atomic_fetch_max(gthr_pass() + (size_t)q * BH_SLOTS256 + (b & (BH_SLOTS256 - 1)), bh_ordf(mine), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                    // (a) refine the bound of ONE of this wave's queries (they rotate; the 256 workgroups refine all of them
                    //     at every exchange): the 256 slots of the query are one 16-byte agent-scope load per lane (the
                    //     per-XCD L2s are not coherent: sc1), and the largest T that NS slots reach is built bit by bit
                    //     (bits 31..8: a bound a little low is still a bound).  NS slots = NS workgroups = NS * RB = KP distinct
                    //     rows at or above T.  (b) the bounds the other workgroups refined: one word per query.
                    const int sel = (b + i + tj) & (16 * NB - 1);
                    const int qsel = wave * (16 * NB) + sel;
                    unsigned* gbound = gthr_pass() + (size_t)(128 * NB) * BH_SLOTS256;
                    uintx4 sl;
                    unsigned gb[NB];
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1"
                                 : "=v"(sl)
                                 : "v"((unsigned)lane_c * 16u), "s"(sgpr_ptr(gthr_pass() + (size_t)qsel * BH_SLOTS256))
                                 : "memory");
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1"
                                     : "=v"(gb[nb])
                                     : "v"((unsigned)q16 * 4u), "s"(sgpr_ptr(gbound + (wave * NB + nb) * 16))
                                     : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(sl));
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(gb[nb]));
                    unsigned T = 0u;
                    for (int bit = 31; bit >= 8; --bit) {
                        const unsigned c = T | (1u << bit);
                        const int n = __builtin_popcountll(__builtin_amdgcn_ballot_w64(sl.x >= c)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(sl.y >= c)) +
                                      __builtin_popcountll(__builtin_amdgcn_ballot_w64(sl.z >= c)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(sl.w >= c));
                        if (n >= NS) T = c;
                    }
                    if (T > BH_ORD_NEG_INF && lane_c == 0)
                        __hip_atomic_fetch_max(gbound + qsel, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        unsigned got = gb[nb];
                        if (sel == nb * 16 + q16) got = max(got, T);
                        // a row that TIES the bound may still win on row index: inclusive compare
                        if (got > BH_ORD_NEG_INF) thr[nb] = fmaxf(thr[nb], bh_unordf(got - 1u));
                    }
                }
                    // read the next tile's first fragments again (the read cursor already stands behind them)
                    if constexpr (!(ABL & 2)) {
#pragma unroll
                        for (int f = 0; f < PD; ++f) lds_read_frag(ag[f], rdb[(f >> 1) & 1], (f >> 2) * 4096 + (f & 1) * 2048);
                        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
                        for (int f = 0; f < NBUF; ++f) asm volatile("" : "+v"(ag[f]));
                    }
                }
            }

            if (next_tile()) break;
        }
        } else {
            // A wave none of whose queries exists (BhScanArgs::nq_valid): its tile is the stage rendezvous, its lines of the refill
            // and the bookkeeping — no fragment reads, no MFMAs, no filter.  (The choice is made HERE, once: a flag tested inside
            // the tile loop would need a scalar register across it, and the scalar file is full.)
            for (;;) {
#pragma unroll
                for (int part = 0; part < S; ++part) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * DPW) : "memory");
                    if constexpr (!(ABL & 16)) asm volatile("s_barrier" ::: "memory");
                    if (loader && !(ABL & 8)) {
#pragma unroll
                        for (int j = 0; j < DPW; ++j) issue_line(j);
                    }
                    advance_cursor();
                    cslot = cslot + 1 == R ? 0 : cslot + 1;
                }
                if (next_tile()) break;
            }
        }
        // drain: tail re-fetches and the fragment reads still in flight (their registers are dead to the compiler)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int f = 0; f < NBUF; ++f) asm volatile("" : "+v"(ag[f]));
    }

    if (a.clk != nullptr && tid == 0) {
        a.clk[8 * b + 2] = wall_clock64();
        a.clk[8 * b + 5] = __builtin_readcyclecounter();
    }

    if constexpr (FILT) return;  // (a filter pass has no lists to publish)

    // ---- final: every wave sorts its queries' buffers and publishes the best KP.  Once the bounds work a buffer holds a
    // handful of candidates: eight buffers of up to 64 entries are loaded together and sorted side by side (the 21 stages
    // of a 64-key sort are a chain of LDS-crossbar shuffles; eight independent chains hide each other's latency)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        for (int q0 = 0; q0 < 16; q0 += 8) {  // (rolled: one copy of the sort per query block)
            u64 e8[8];
            bool big = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const unsigned n = __builtin_amdgcn_readlane(cnt[nb], q0 + t);
                big = big || n > 64u;
                const unsigned nn = n > 64u ? 0u : n;
                load_list256<1>(*reinterpret_cast<u64(*)[1]>(&e8[t]), cand_wg + (size_t)((wave * NB + nb) * 16 + q0 + t) * CAP, nn, lane);
            }
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) bh_bitonic_stage<8>(e8, lane, j, (k == 64) ? 0 : k);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int qi = (wave * NB + nb) * 16 + q0 + t;
                part_wg[(size_t)qi * KP + lane] = e8[t];
#pragma unroll
                for (int r = 1; r < EPLK; ++r) part_wg[(size_t)qi * KP + r * 64 + lane] = 0ull;
            }
            if (__builtin_expect(big, 0)) {
                // longer buffers, one at a time, with the smallest sort that holds them: lists of 256 end a pass with ~60
                // (a shard of an eighth of the corpus) to ~150 candidates per query and workgroup, and the 512-key sort of a
                // full buffer for each of them was 0.11 - 0.24 ms per pass (profiles/scan_phases.py, BH_K=200)
#pragma unroll 1
                for (int t = 0; t < 8; ++t) {
                    const int qi = (wave * NB + nb) * 16 + q0 + t;
                    const unsigned n = __builtin_amdgcn_readlane(cnt[nb], q0 + t);
                    if (n <= 64u) continue;
                    const u64* src = cand_wg + (size_t)qi * CAP;
                    u64* dst = part_wg + (size_t)qi * KP;
                    if (EPLC >= 4 && n <= 128u) {
                        u64 e[2];
                        load_list256<2>(e, src, n, lane);
                        bh_wave_sort_desc<2>(e, lane);
#pragma unroll
                        for (int r = 0; r < EPLK; ++r) dst[r * 64 + lane] = r < 2 ? e[r < 2 ? r : 0] : 0ull;
                    } else if (EPLC >= 8 && n <= 256u) {
                        u64 e[4];
                        load_list256<4>(e, src, n, lane);
                        bh_wave_sort_desc<4>(e, lane);
#pragma unroll
                        for (int r = 0; r < EPLK; ++r) dst[r * 64 + lane] = r < 4 ? e[r < 4 ? r : 0] : 0ull;
                    } else {
                        u64 e[EPLC];
                        sort_candidates256<KP>(e, src, n, lane);
#pragma unroll
                        for (int r = 0; r < EPLK; ++r) dst[r * 64 + lane] = e[r];
                    }
                }
            }
        }
    }
    if (a.clk != nullptr && tid == 0) {
        unsigned tot = 0;
        for (int nb = 0; nb < NB; ++nb)
            for (int qq = 0; qq < 16; ++qq) tot += __builtin_amdgcn_readlane(cnt[nb], qq);
        a.clk[8 * b + 6] = tot;
        a.clk[8 * b + 3] = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch

template <int NK32, int KP, int LS, int R, int PD, int ABL = 0, int LM = 1, int SCHED = 1, int NBUF = PD, int NB = 2>
static hipError_t launch256_one(const BhScanArgs& a, int grid, hipStream_t stream) {
    constexpr size_t ring = (size_t)R * 32 * LS * 128;
    constexpr size_t smem = ring + 16 <= 160 * 1024 ? ring + 16 : ring;  // + the chunk claim word (dynamic tile distribution)
    constexpr bool kProduction = (ABL == 0 || ABL == 128) && (LM == 1 || LM == 0) && SCHED == 1 && NBUF == PD;
    // the bench-only instantiations exist with the non-temporal stream policy only (compile time)
    // (... except the ablations of the PAIRED launch, which exist with the cached policy only: partner workgroups share the
    // stream through L2, which a non-temporal first reader would defeat — the production paired launch runs cached too)
    constexpr bool kPairedAblation = (ABL & 128) != 0 && ABL != 128;
    const bool nt = !kPairedAblation && (a.nontemporal != 0 || !kProduction);
    static bool attr_done[2] = {false, false};
    void (*kern)(BhScanArgs) = bh_scan_topk256_kernel<NK32, KP, LS, R, PD, !kPairedAblation, ABL, LM, SCHED, NBUF, NB>;
    if constexpr (kProduction) {
        if (!nt) kern = bh_scan_topk256_kernel<NK32, KP, LS, R, PD, false, ABL, LM, SCHED, NBUF, NB>;
    }
    if (!attr_done[nt ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done[nt ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, a);
    return hipGetLastError();
}

template <int NK32, int LS, int R, int PD, int NB = 2, int LM = 1>
static hipError_t launch256_kp(const BhScanArgs& a, int kp, int grid, hipStream_t stream) {
    switch (kp) {
        case 64: return launch256_one<NK32, 64, LS, R, PD, 0, LM, 1, PD, NB>(a, grid, stream);
        case 128: return launch256_one<NK32, 128, LS, R, PD, 0, LM, 1, PD, NB>(a, grid, stream);
        case 256: return launch256_one<NK32, 256, LS, R, PD, 0, LM, 1, PD, NB>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}

// Two waves per SIMD need the fragments of a wave's queries in 192 registers: 32 queries up to d = 768 (256 queries per
// workgroup), 16 queries at d = 1024 (128 registers; 128 queries per workgroup)
bool bh_scan256_supports(int dim_padded, int kp) {
    return (dim_padded == 1024 || dim_padded == 768 || dim_padded == 512 || dim_padded == 384) && (kp == 64 || kp == 128 || kp == 256);
}
int bh_scan256_tile(int dim_padded) { return dim_padded == 1024 ? 128 : 256; }

// the exact fall-back's filter pass on the 256-query tile (d = 384 / 512 / 768; at d = 1024 this kernel's tile is 128 queries
// wide, no wider than scan_topk.hip's filter pass, which serves it)
bool bh_filter256_supports(int dim_padded) { return dim_padded == 768 || dim_padded == 512 || dim_padded == 384; }
hipError_t bh_launch_filter_scan256(const BhScanArgs& a, int dim_padded, int grid, hipStream_t stream) {
    if (!bh_filter256_supports(dim_padded) || a.qsplit != 1 || !a.fix_thr || !a.fix_cnt || !a.fix_rows || a.share != 0 || (a.dyn_tiles != 0 && !a.gthr))
        return hipErrorInvalidValue;
    switch (dim_padded) {
        case 384: return launch256_one<12, 64, 6, 6, 4, 64>(a, grid, stream);
        case 512: return launch256_one<16, 64, 4, 9, 4, 64>(a, grid, stream);
        case 768: return launch256_one<24, 64, 12, 3, 4, 64, 0>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}

// the paired launch (two passes of one launch on partner workgroups, see the kernel): d = 768 on the one-rendezvous ring, d = 1024
// on the four-stage ring (the five-stage one has no room for the claim and progress words)
bool bh_scan256_pair_supports(int dim_padded, int kp) { return (dim_padded == 768 || dim_padded == 1024) && (kp == 64 || kp == 128 || kp == 256); }
template <int NK32, int LS, int R, int PD, int NB, int LM>
static hipError_t launch256_paired_kp(const BhScanArgs& a, int kp, int grid, hipStream_t stream) {
    switch (kp) {
        case 64: return launch256_one<NK32, 64, LS, R, PD, 128, LM, 1, PD, NB>(a, grid, stream);
        case 128: return launch256_one<NK32, 128, LS, R, PD, 128, LM, 1, PD, NB>(a, grid, stream);
        case 256: return launch256_one<NK32, 256, LS, R, PD, 128, LM, 1, PD, NB>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}
hipError_t bh_launch_scan256_paired(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream) {
    if (!bh_scan256_pair_supports(dim_padded, kp) || a.qsplit != 2 || grid != 256 || !a.progress) return hipErrorInvalidValue;
    if (a.ablate != 0) {
        // bench-only (profiles/ablate_paired.py; results invalid): the in-kernel ablation ladder of the PAIRED headline instantiation
        if (dim_padded != 768 || kp != 64) return hipErrorInvalidValue;
        switch (a.ablate) {
            case 1: return launch256_one<24, 64, 12, 3, 4, 128 | 1, 0, 1, 4, 2>(a, grid, stream);  // no filter / candidate path
            case 5: return launch256_one<24, 64, 12, 3, 4, 128 | 5, 0, 1, 4, 2>(a, grid, stream);  // + no MFMA: stream + fragment reads
            case 7: return launch256_one<24, 64, 12, 3, 4, 128 | 7, 0, 1, 4, 2>(a, grid, stream);  // + no fragment reads: stream + rendezvous (+ pacing)
            case 9: return launch256_one<24, 64, 12, 3, 4, 128 | 9, 0, 1, 4, 2>(a, grid, stream);  // no filter, no refill: MFMA + fragment reads + rendezvous
        }
        return hipErrorInvalidValue;
    }
    if (dim_padded == 1024) return launch256_paired_kp<32, 8, 4, 4, 1, 1>(a, kp, grid, stream);
    return launch256_paired_kp<24, 12, 3, 4, 2, 0>(a, kp, grid, stream);
}

hipError_t bh_launch_scan256(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream) {
    if (!bh_scan256_supports(dim_padded, kp) || a.qsplit != 1) return hipErrorInvalidValue;
    switch (dim_padded) {
        case 384: return launch256_kp<12, 6, 6, 4>(a, kp, grid, stream);
        case 512: return launch256_kp<16, 4, 9, 4>(a, kp, grid, stream);
        case 1024:
            if (a.ring_variant == 6) return launch256_kp<32, 4, 10, 4, 1>(a, kp, grid, stream);  // 16 KiB stages x 10: four rendezvous per tile
            // 32 KiB stages x 4 = 128 KiB: one stage less in flight, but room for the claim word — the DYNAMIC tile distribution
            // (the five-stage ring takes all 160 KiB and stays static: 79 us between the first and the last workgroup's end on
            // an eighth of the 21 M x 1024 corpus, profiles/r04e_scan_phases_d1024.txt)
            // Same-box A/B (profiles/README.md round 4): +0.7 % per pass over all 21 M rows, -2.0 % over an eighth of them — so a
            // SHARD (< 8 M rows: one of several GPUs' share) takes the dynamic ring, a whole corpus the five-stage one;
            // ring_variant 5 / 7 force either (tests run both on small corpora)
            if (a.ring_variant == 5 || (a.ring_variant == 0 && a.n_rows < 8000000ll)) return launch256_kp<32, 8, 4, 4, 1>(a, kp, grid, stream);
            return launch256_kp<32, 8, 5, 4, 1>(a, kp, grid, stream);  // 32 KiB stages x 5 = all 160 KiB of LDS, two rendezvous per tile
        case 768:
            if (kp == 64) {
                // bench-only: ablations (bit flags, see the kernel) and schedule variants of the headline geometry
                switch (a.ablate) {
                    case 1: return launch256_one<24, 64, 12, 3, 4, 1>(a, grid, stream);    // no filter
                    case 3: return launch256_one<24, 64, 12, 3, 4, 3>(a, grid, stream);    // no filter, no fragment reads
                    case 7: return launch256_one<24, 64, 12, 3, 4, 7>(a, grid, stream);    // stream only
                    case 9: return launch256_one<24, 64, 12, 3, 4, 9>(a, grid, stream);    // no filter, no refill
                    case 11: return launch256_one<24, 64, 12, 3, 4, 11>(a, grid, stream);  // MFMA + rendezvous only
                    case 32: return launch256_one<24, 64, 12, 3, 4, 32>(a, grid, stream);  // production + timeline stamps
                    case 33: return launch256_one<24, 64, 12, 3, 4, 33>(a, grid, stream);  // no filter + timeline stamps
                    case 0: break;
                    default: return hipErrorInvalidValue;
                }
                switch (a.ring_variant) {
                    case 1: return launch256_one<24, 64, 12, 3, 4, 0, 1, 1>(a, grid, stream);     // waves 0-3 load, at raised priority
                    case 2: return launch256_one<24, 64, 12, 3, 4, 0, 1, 0>(a, grid, stream);     // rendezvous at the stage top, refill spread
                    case 3: return launch256_one<24, 64, 6, 6, 3, 0, 1, 1, 4>(a, grid, stream);  // prefetch distance 3, four buffers
                    case 4: return launch256_one<24, 64, 4, 9, 4>(a, grid, stream);              // 4 lines x 9 stages
                    case 7: return launch256_one<24, 64, 12, 3, 4, 0, 0, 2>(a, grid, stream);     // refill spread over the second half of the stage
                    case 5: return launch256_one<24, 64, 6, 6, 4>(a, grid, stream);              // 6 lines x 6 stages: two rendezvous per tile (the geometry until mid round 2)
                }
            }
            // a whole 32-row tile per stage, three stages: ONE rendezvous per tile (two with 6 lines x 6 stages: +2.9 % time);
            // with 12 LDS-DMA instructions per stage all eight waves issue six each (four loader waves x 12: +0.9 % time)
            return launch256_kp<24, 12, 3, 4, 2, 0>(a, kp, grid, stream);
    }
    return hipErrorInvalidValue;
}
