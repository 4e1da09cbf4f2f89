// gemm_f16.hip — MFMA GEMM of the bi-encoder forward pass (roofline: MFMA, fp16 dense 2.5 PFLOP/s).
//
//   C[M][N] = A[M][K] . B[N][K]^T  (+ bias)  (+ residual)  (GELU)      fp16 in / fp32 accumulate / fp16 out
//
// Replaces the torch.nn.Linear calls inside HF BertModel.forward that the reference reaches through
// AutoModel (models/retrievers/dense.py:16,40-44): QKV / attention-output / FFN projections
// (transformers modeling_bert.py: BertSelfAttention.query/key/value, BertSelfOutput.dense,
// BertIntermediate.dense + GELU, BertOutput.dense), with the bias add, the residual add and the erf-GELU
// fused into the epilogue instead of running as separate HBM passes.
//
// Kernel (gemm_f16_kernel.h).  Both operands are K-contiguous — activations [tokens][d] and HF weights
// [out][in] — so both stream L2 -> LDS by LDS-DMA (`global_load_lds_dwordx4`, no VGPR round trip) into an
// XOR-permuted image whose MFMA fragment reads are conflict-free ds_read_b128 (the permutation is applied to
// the per-lane SOURCE address).  R-deep ring of BK-wide stages, ONE raw s_barrier per stage with a counted
// `s_waitcnt vmcnt` (loads stay in flight across barriers), fragment reads one k-step ahead of the MFMAs,
// LDS-DMA issue spread over the gaps of the stage's last MFMA group.  v_mfma_f32_32x32x16_f16 with the WEIGHT
// fragment as operand A (i = n) and the ACTIVATION fragment as operand B (j = m): a lane ends with one token
// row and, after a half-lane exchange, 8 consecutive output columns -> 16-byte stores.  XCD-aware block->tile
// map.  Epilogue features are compile-time (no branches in the hot path); ragged edges go to a separate,
// bounds-checked launch over the edge strips.
//
// Measured (profiles/README.md): for K = 768 the epilogue (bias/GELU/convert/store of 64 Ki outputs per tile)
// costs as much as the 12-stage main loop when it runs serialised behind it, so the production configuration
// keeps TWO independent 4-wave blocks per CU (256x128 tile, BK = 32, ring 3 = 72 KiB each): one block's
// VALU/store epilogue overlaps the other block's MFMA main loop.
#include "gemm_f16_persist.h"
hipError_t bh_gemm_p16(const BhGemmArgs& a, int epi, bool nontemporal, int mode, hipStream_t s);  // gemm_f16_d.hip

// Probe of v_permlane32_swap's direction (documented: vdst[32:63] <-> vsrc[0:31]).
__global__ void bh_permlane_probe_kernel(unsigned* out) {
    const unsigned lane = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}

namespace {
int g_swap_b = -1;  // -1 unknown, 0 documented direction, 1 the other one
int g_stagger_phases = 0, g_stagger_pct = 100;  // bench knobs (bh_set_option "gemm_stagger_phases" / "_pct")
int g_gelu_nontemporal = 1;  // bh_set_option "gemm_gelu_nontemporal": the FFN-up (bias + GELU) output bypasses the caches (measured +8 % on that
                             // GEMM in round 1, when one batch's 421 MB were written at once; an A/B knob since the micro-batches halved that)
int g_full_line_stores = 2;  // bh_set_option "gemm_full_line_stores" (default 2 = every output path; 1 = the row-major outputs only, 0 = off): the persistent kernel's row-major outputs leave as whole
                             // 128-byte lines through LDS (gemm_f16_persist.h PST bit 32) — bit-identical, BERT-base forward 15.60 -> 14.55 ms
                             // per 512 passages in the same process (profiles/r04t_ab_full_line_stores.txt)
}

void bh_gemm_set_gelu_nontemporal(int on) { g_gelu_nontemporal = on != 0; }
int g_mfma16 = 1;  // bh_set_option "gemm_mfma16" (default 1 since round 5): the bias (+ GELU) projections on gemm_f16_p16.h (v_mfma_f32_16x16x32_f16) instead of gemm_f16_persist.h
void bh_gemm_set_mfma16(int on) { g_mfma16 = on; }  // (16 x ablation bits + 1: profiles/gemm_p16_ablate.py)
int g_tail_split = 0;  // bh_set_option "gemm_tail_split": gemm_f16_p16.h cuts a short last round of tiles into sub-tiles (same bits; default off: faster
                       // for a GEMM alone, 1 % slower inside the encoder, whose two micro-batch streams already fill a launch's idle tail)
void bh_gemm_set_tail_split(int on) { g_tail_split = on != 0; }
void bh_gemm_set_full_line_stores(int level) { g_full_line_stores = level < 0 ? 0 : level > 2 ? 2 : level; }

void bh_gemm_set_stagger(int phases, int pct) {
    if (phases >= 0) g_stagger_phases = phases;
    if (pct >= 0) g_stagger_pct = pct;
}

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

namespace {
struct TileDim {
    int bm, bn;
};
const TileDim kTile[6] = {{0, 0}, {128, 128}, {256, 128}, {256, 256}, {256, 128}, {256, 256}};
}  // namespace

#define CFG1(E) bh_gemm_launch_cfg<64, 2, 2, 2, 2, 2, E, 2>(a, s)
hipError_t bh_gemm_cfg1(const BhGemmArgs& a, int epi, hipStream_t s) { BH_GEMM_DISPATCH_EPI(epi, CFG1) }
#define CFG5(E) bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2>(a, s)
hipError_t bh_gemm_cfg5(const BhGemmArgs& a, int epi, hipStream_t s) { BH_GEMM_DISPATCH_EPI(epi, CFG5) }
hipError_t bh_gemm_generic(const BhGemmArgs& a, int, hipStream_t s) {
    return bh_gemm_launch_cfg<64, 2, 2, 2, 2, 2, 0, 2, true>(a, s);
}

namespace {

hipError_t run_cfg(int cfg, const BhGemmArgs& a, int epi, hipStream_t s) {
    switch (cfg) {
        case 1: return bh_gemm_cfg1(a, epi, s);
        case 2: return bh_gemm_cfg2(a, epi, s);
        case 3: return bh_gemm_cfg3(a, epi, s);
        case 4: return bh_gemm_cfg4(a, epi, s);
        case 5: return bh_gemm_cfg5(a, epi, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t bh_launch_gemm_f16_batched(const BhGemmArgs& a_in, hipStream_t stream) {
    BhGemmArgs a = a_in;
    if (a.batch <= 0 || a.M <= 0 || a.N <= 0) return hipSuccess;
    if ((a.M & 255) || (a.N & 255) || a.K <= 0 || (a.K & 63) || (a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (a.batch_stride_a & 7) ||
        (a.batch_stride_b & 7) || (a.batch_stride_c & 7) || a.bias || a.residual || a.gelu || a.seg_out || a.c_block_rows)
        return hipErrorInvalidValue;
    hipError_t e = bh_gemm_probe_permlane(stream);
    if (e != hipSuccess) return e;
    if (g_swap_b != 0) return hipErrorNotSupported;  // (the caller falls back to one launch per problem)
    a.swap_b = g_swap_b;
    a.stagger_phases = 0;
    return bh_gemm_persist(a, BH_EPI_BATCHED, 1, stream);
}

static int gemm_cu_count();
bool bh_gemm_ln_fusable(int M, int N, bool blocked_out) {
    // the auto dispatch of bh_launch_gemm_f16: persistent kernel for more than half a round of whole 256 x 256 tiles, documented
    // swap direction, full-line stores on (level 2 for the blocked V^T output)
    if (M <= 0 || N <= 0 || (M & 255) || (N & 255) || g_swap_b != 0) return false;
    if (g_full_line_stores < (blocked_out ? 2 : 1)) return false;
    return (long long)(M / 256) * (N / 256) * 2 > gemm_cu_count();
}

static int gemm_cu_count() {
    static int cached = 0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (cached == 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cached > 0 ? cached : 256;
}

// variant: 0 = auto; 1..5 = explicit tile configuration (gemm_f16_kernel.h); 6 = generic bounds-checked kernel
// for everything; 7 = persistent 256x256 kernel (gemm_f16_persist.h; burst stores); 8 = 7 with stores deferred into
// the next tile's main loop, 9 = 7 with non-temporal stores (both valid results; ablations); 11..28 = bench-only
// ablations (results invalid); 33 = 7 with deferred stores and alternating loader teams; 10 = 7 with the last partial round of tiles re-cut into 128x128 tiles (experiment, measured neutral).
bool bh_gemm_geglu_fusable() { return g_mfma16 != 0 && g_mfma16 < 16 && g_full_line_stores >= 2; }

static hipError_t launch_gemm_impl(const BhGemmArgs& a_in, int variant, hipStream_t stream, long long* rot_rows);

// BhGemmArgs::rot_pos set: the projection comes back with rotary positions applied — fused into the 16x16x32 kernel's epilogue for the rows
// that kernel computes (launch_gemm_impl reports how many), by the standalone kernel for the rest (edge strips, small problems, other routes)
hipError_t bh_launch_gemm_f16(const BhGemmArgs& a_in, int variant, hipStream_t stream) {
    if (!a_in.rot_pos) return launch_gemm_impl(a_in, variant, stream, nullptr);
    if (!a_in.rot_cs || a_in.rot_max_pos <= 0 || !a_in.bias || a_in.bias_mode != 1 || a_in.N % 128 != 0 || a_in.ldc != a_in.N || a_in.gelu || a_in.swiglu ||
        a_in.residual || a_in.c_block_rows || a_in.seg_out || a_in.ln_stats || a_in.stats_out)
        return hipErrorInvalidValue;
    if (a_in.M <= 0) return hipSuccess;
    long long done = 0;
    hipError_t e = launch_gemm_impl(a_in, variant, stream, &done);
    if (e != hipSuccess) return e;
    if (done < a_in.M) {
        BhRotaryArgs ra{};
        ra.qk = a_in.C + (size_t)done * a_in.ldc;
        ra.pos = a_in.rot_pos + done;
        ra.cos_sin = a_in.rot_cs;
        ra.n_rows = a_in.M - done;
        ra.n_heads = a_in.N / 128;
        ra.max_pos = a_in.rot_max_pos;
        e = bh_launch_rotary(ra, stream);
    }
    return e;
}

static int g_rotary_fused = 1;
void bh_gemm_set_rotary_fused(int on) { g_rotary_fused = on ? 1 : 0; }

static hipError_t launch_gemm_impl(const BhGemmArgs& a_in, int variant, hipStream_t stream, long long* rot_rows) {
    BhGemmArgs a = a_in;
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (a.K <= 0 || (a.K & 63) || (a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (a.residual && (a.ldr & 7)))
        return hipErrorInvalidValue;
    hipError_t e = bh_gemm_probe_permlane(stream);
    if (e != hipSuccess) return e;
    a.swap_b = g_swap_b;
    a.stagger_phases = 0;
    if (g_stagger_phases > 1) {
        // spread the first round of blocks over ~pct % of one tile time (estimated at 60 % MFMA utilisation)
        const int cfgv = variant >= 21 ? 2 : variant >= 11 ? 5 : (variant == 0 || (variant >= 7 && variant <= 9) || variant >= 31) ? 5 : variant;
        if (cfgv >= 1 && cfgv <= 5) {
            const int bm = kTile[cfgv].bm, bn = kTile[cfgv].bn;
            const int per_cu = (cfgv == 1 || cfgv == 2) ? 2 : 1;
            const double tile_cycles = (double)(bm / 32) * (bn / 32) * (a.K / 16) * 32.0 / 4.0 * per_cu / 0.6;
            a.stagger_phases = g_stagger_phases;
            a.stagger_unit = (int)(tile_cycles * g_stagger_pct / 100.0 / g_stagger_phases / 8128.0 + 0.5);
            if (a.stagger_unit < 1) a.stagger_unit = 1;
            a.stagger_first_round = 256 * per_cu;
        }
    }
    if (variant >= 11 && variant < 31) return bh_gemm_ablate(a, variant, stream);
    int epi = 0;
    if (a.bias && a.bias_mode == 1) epi |= BH_EPI_BIAS_COL;
    if (a.bias && a.bias_mode == 2) epi |= BH_EPI_BIAS_ROW;
    if (a.residual) epi |= BH_EPI_RESIDUAL;
    if (a.gelu) epi |= BH_EPI_GELU;
    if (a.seg_out) {  // SPLADE head: whole 256x256 tiles on the persistent kernel, bias per row (vocabulary term)
        if (!a.seg_grp || a.M % 256 || a.N % 256 || a.residual || a.gelu || !(a.bias && a.bias_mode == 2) || g_swap_b != 0)
            return hipErrorInvalidValue;
        return bh_gemm_persist(a, BH_EPI_BIAS_ROW | BH_EPI_SEGMAX, 1, stream);
    }
    if (a.swiglu) {  // gated feed-forward fold: whole 256x256 tiles on the persistent kernel, bias per (interleaved) column
        if (a.M % 256 || a.N % 256 || a.residual || a.gelu || !(a.bias && a.bias_mode == 1) || a.c_block_rows || g_swap_b != 0)
            return hipErrorInvalidValue;
        // (non-temporal burst, like the GELU output; level 2 of gemm_full_line_stores: 64-byte row pieces through LDS)
        if (g_mfma16 && g_mfma16 < 16 && g_full_line_stores >= 2) {  // (the 16x16x32 kernel has the through-LDS route only)
            BhGemmArgs t = a;
            t.tail_split = g_tail_split;
            return bh_gemm_p16(t, BH_EPI_BIAS_COL | BH_EPI_SWIGLU | (a.swiglu == 2 ? BH_EPI_GELU : 0), true, g_mfma16, stream);
        }
        if (a.swiglu == 2) return hipErrorNotSupported;  // (the GELU gate exists on the 16x16x32 kernel only: bh_gemm_geglu_fusable())
        return bh_gemm_persist(a, BH_EPI_BIAS_COL | BH_EPI_SWIGLU, g_full_line_stores >= 2 ? 35 : 3, stream);
    }
    if (a.ln_stats || a.stats_out) {
        // fused LayerNorm: whole 256 x 256 tiles on the persistent kernel's full-line-store path, nothing else (the caller asks
        // bh_gemm_ln_fusable first and keeps the separate LayerNorm kernel otherwise)
        if (!bh_gemm_ln_fusable(a.M, a.N, a.c_block_rows != 0) || !a.bias || (variant != 0 && variant != 7))
            return hipErrorNotSupported;
        if (a.stats_out) {
            if (a.ln_stats || a.bias_mode != 1 || !a.residual || a.gelu || a.c_block_rows || (a.res_stats && (!a.res_gamma || !a.res_beta)))
                return hipErrorInvalidValue;
            return bh_gemm_persist(a, BH_EPI_BIAS_COL | BH_EPI_RESLN, 33, stream);
        }
        if (!a.ln_c || a.residual) return hipErrorInvalidValue;
        if (a.bias_mode == 2) return a.gelu ? hipErrorNotSupported : bh_gemm_persist(a, BH_EPI_BIAS_ROW | BH_EPI_LNA, 33, stream);
        if (a.gelu) return bh_gemm_persist(a, BH_EPI_BIAS_COL | BH_EPI_GELU | BH_EPI_LNA, g_gelu_nontemporal ? 35 : 33, stream);
        return bh_gemm_persist(a, BH_EPI_BIAS_COL | BH_EPI_LNA, 33, stream);
    }
    const bool epi_fast = epi == 0 || epi == BH_EPI_BIAS_COL || epi == BH_EPI_BIAS_ROW ||
                          epi == (BH_EPI_BIAS_COL | BH_EPI_RESIDUAL) || epi == (BH_EPI_BIAS_COL | BH_EPI_GELU);
    const bool auto_variant = variant == 0;
    const bool variant_balanced = variant == 10;  // 7 + balanced remainder, explicitly
    if (variant == 10) variant = 7;
    if (variant == 0) variant = (a.M >= 256 && a.N >= 256) ? ((epi & BH_EPI_RESIDUAL) ? 5 : 7) : (a.M >= 256 && a.N >= 128) ? 2 : 1;
    // A small problem leaves most of the chip idle under 256x256 tiles (a rerank batch of 32 pairs: 5 632 packed rows x
    // N = 1024 = 88 tiles on 256 CUs): with at most half a round of them, 128x128 tiles (two workgroups per CU) put four
    // times as many workgroups on the chip.  Measured (profiles/r04i_gemm_small_m.jsonl): 0.0282 -> 0.0218 ms at
    // K = 1024 and 0.0852 -> 0.0637 ms at K = 4096 for 88 tiles; at 176 tiles the 256x256 kernel is still ahead.
    if (auto_variant && variant == 7 && (!a.c_block_rows || (a.M % 128 == 0 && a.N % 128 == 0)) &&
        (long long)(a.M / 256) * (a.N / 256) * 2 <= gemm_cu_count())
        variant = 1;
    if (variant == 6 || !epi_fast || g_swap_b != 0) return bh_gemm_generic(a, epi, stream);
    const bool persist = (variant >= 7 && variant <= 9) || variant == 31 || variant == 32 || variant == 33;
    if (a.c_block_rows && !(((persist && a.M % 256 == 0 && a.N % 256 == 0) || (variant == 1 && a.M % 128 == 0 && a.N % 128 == 0)) &&
                            !(epi & BH_EPI_RESIDUAL)))
        return hipErrorInvalidValue;  // blocked output: whole tiles only (fast epilogues)
    // burst stores; non-temporal for the GELU (FFN-up) output, which is far larger than the caches and is read
    // back only by the next kernel (measured: +8 % on that GEMM, -7 % on the others)
    // 33: deferred stores with alternating loader teams (gemm_f16_persist.h PST 16) where the stage count allows it
    const int kt_ = a.K / 64;
    const bool alt_ok = (kt_ & 1) == 0 && kt_ >= 8;
    const int pst = variant == 31 ? 5 : variant == 32 ? 9 : variant == 8 ? 0 : variant == 9 ? 3 : (variant == 33 && alt_ok) ? 16
                    : (auto_variant && (epi & BH_EPI_GELU) && g_gelu_nontemporal) ? 3 : 1;
    if (persist) {
        if (epi & BH_EPI_RESIDUAL) return bh_gemm_generic(a, epi, stream);  // (the encoder adds residuals in LayerNorm)
        variant = 5;  // same tile geometry
    }
    int pst_eff = pst;
    if (persist && g_full_line_stores && !a.c_block_rows && (pst == 1 || pst == 3) && (epi == 0 || epi == BH_EPI_BIAS_COL || epi == (BH_EPI_BIAS_COL | BH_EPI_GELU)))
        pst_eff = pst | 32;
    // level 2 (the default since round 5: bit-identical to level 1, tests/test_gpu_store_paths.py; BERT-base forward 14.40 -> 14.29 ms, NomicBert
    // 20.49 -> 19.15 ms, profiles/r05a_ab_full_line_level2.txt): the blocked V^T output too — bias per row, rows of 64 columns
    // that are whole 128-byte lines already, 8 of them = 1 KiB contiguous per store instruction
    if (persist && g_full_line_stores >= 2 && a.c_block_rows && pst == 1 && (epi == 0 || epi == BH_EPI_BIAS_ROW)) pst_eff = pst | 32;
    if (variant < 1 || variant > 5) return hipErrorInvalidValue;
    // interior region with the fast kernel, edge strips with the generic one
    const int bm = kTile[variant].bm, bn = kTile[variant].bn;
    const int mi = a.M / bm * bm, ni = a.N / bn * bn;
    if (mi > 0 && ni > 0) {
        BhGemmArgs t = a;
        t.M = mi;
        t.N = ni;
        // Wave quantisation: T tiles on n_cu persistent workgroups take ceil(T / n_cu) tile times (804 tiles of an
        // N = 768 projection = 3.14 -> 4 rounds).  In auto mode the tiles beyond the last FULL round go to a second
        // launch with 128x128 tiles (two blocks per CU): a quarter of the work per tile, spread over the whole chip.
        // Variant 10 only: measured neutral (a small tile still pays the full K-loop latency), kept as an experiment.
        const int n_cu = gemm_cu_count();
        const int tm_ = mi / 256, tn_ = ni / 256;
        const long long T = (long long)tm_ * tn_;
        const long long full = T / n_cu * n_cu;
        if (persist && variant_balanced && full > 0 && full < T) {  // (measured neutral: the re-cut tiles keep the
                                                                    // K-loop latency; not used by auto)
            BhGemmArgs r = t;  // remainder
            if (tm_ >= tn_) {  // split along M: whole M panels of tn_ tiles
                const int pm = (int)(full / tn_);
                t.M = pm * 256;
                r.A = t.A + (size_t)t.M * t.lda;
                r.M = mi - t.M;
                if (t.c_block_rows)
                    r.C = t.C + (size_t)t.M * 64;  // blocked layout: rows are 64 elements apart inside a block
                else
                    r.C = t.C + (size_t)t.M * t.ldc;
                if (t.bias && t.bias_mode == 2) r.bias = t.bias + t.M;
            } else {  // split along N: whole N panels of tm_ tiles
                const int pn = (int)(full / tm_);
                t.N = pn * 256;
                r.B = t.B + (size_t)t.N * t.ldb;
                r.N = ni - t.N;
                if (t.c_block_rows)
                    r.C = t.C + (size_t)(t.N / 64) * t.c_block_rows * 64;
                else
                    r.C = t.C + t.N;
                if (t.bias && t.bias_mode == 1) r.bias = t.bias + t.N;
            }
            if ((e = bh_gemm_persist(t, epi, pst_eff, stream)) != hipSuccess) return e;
            if ((e = run_cfg(1, r, epi, stream)) != hipSuccess) return e;
        } else {
            // every full-line-store case of the plain epilogues runs on the 16x16x32 kernel (gemm_f16_p16.h) unless gemm_mfma16 is 0
            if (persist && g_mfma16 && (pst_eff == 33 || pst_eff == 35) &&
                (epi == 0 || epi == BH_EPI_BIAS_COL || epi == (BH_EPI_BIAS_COL | BH_EPI_GELU) || (epi == BH_EPI_BIAS_ROW && t.c_block_rows))) {
                t.tail_split = g_tail_split;
                // rotary positions in the epilogue (BhGemmArgs::rot_pos; option rotary_fused): all columns must be this kernel's
                int epi_l = epi;
                if (rot_rows && a.rot_pos && g_rotary_fused && epi == BH_EPI_BIAS_COL && !t.c_block_rows && ni == a.N) epi_l |= BH_EPI_ROTARY;
                // (gemm_mfma16 >= 16 are the bench-only ablation modes of the bias + GELU instantiation, profiles/gemm_p16_ablate.py:
                // every other epilogue runs the production kernel — mode 1 — so that a forward pass never fails on them)
                e = bh_gemm_p16(t, epi_l, pst_eff == 35, (g_mfma16 >= 16 && epi != (BH_EPI_BIAS_COL | BH_EPI_GELU)) ? 1 : g_mfma16, stream);
                if (e == hipSuccess && (epi_l & BH_EPI_ROTARY)) *rot_rows = t.M;
            } else
                e = persist ? bh_gemm_persist(t, epi, pst_eff, stream) : run_cfg(variant, t, epi, stream);
            if (e != hipSuccess) return e;
        }
    }
    if (ni < a.N) {  // right strip: all rows, columns [ni, N)
        BhGemmArgs t = a;
        t.B = a.B + (size_t)ni * a.ldb;
        t.C = a.C + ni;
        t.N = a.N - ni;
        if (a.bias && a.bias_mode == 1) t.bias = a.bias + ni;
        if (a.residual) t.residual = a.residual + ni;
        if ((e = bh_gemm_generic(t, epi, stream)) != hipSuccess) return e;
    }
    if (mi < a.M && ni > 0) {  // bottom strip: rows [mi, M), interior columns
        BhGemmArgs t = a;
        t.A = a.A + (size_t)mi * a.lda;
        t.C = a.C + (size_t)mi * a.ldc;
        t.M = a.M - mi;
        t.N = ni;
        if (a.bias && a.bias_mode == 2) t.bias = a.bias + mi;
        if (a.residual) t.residual = a.residual + (size_t)mi * a.ldr;
        if ((e = bh_gemm_generic(t, epi, stream)) != hipSuccess) return e;
    }
    return hipSuccess;
}
