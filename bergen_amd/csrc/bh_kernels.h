// bh_kernels.h — host-visible launch interfaces of the gfx950 kernels (internal to the .so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long bh_u64;
#define BH_MAX_LIST_K 248   /* largest k one fused search serves (candidate lists of 256 entries with a margin of 8) */
#define BH_MAX_K 4096       /* largest k bh_search* accepts: above BH_MAX_LIST_K the corpus is searched range by range (index.hip) */

struct BhScanArgs {
    const _Float16* corpus;  // [n_tiles*32][D] fp16, rows >= n_rows are zero padding
    long long n_rows;        // valid rows
    long long n_tiles;       // ceil(n_rows / 32)
    const _Float16* qtile;   // [qsplit*BQ][D] fp16 query tile (zero rows beyond the valid queries)
    bh_u64* cand;            // [grid][BQ][2*KP] candidate buffers (scratch)
    bh_u64* partial;         // [grid/qsplit][qsplit*BQ][KP]   out: per-workgroup sorted best-KP keys
    unsigned* gthr;          // [qsplit*BQ][64] threshold slot table (ordf), initialised to BH_ORD_NEG_INF
    int share;               // share thresholds between workgroups
    int nontemporal;         // nt cache policy on the corpus stream
    int ablate;              // bench-only kernel ablation (0 = production kernel)
    int ring_variant;        // bench-only LDS ring geometry selector for d=768 (0 = default 6 lines x 6)
    int qsplit;              // 1 | 2: workgroups sharing each row tile, each with its own BQ queries (see scan_topk.hip)
    unsigned* progress;      // [grid] qsplit = 2: tiles started per workgroup (zeroed by the host), pacing hint only
    int dma_interleave;      // 1: LDS-DMA refills issued one per line between the MFMAs instead of all after the barrier
    int pair_window;         // qsplit = 2: a workgroup may run at most this many tiles ahead of its partner (0 = free-running)
    int dyn_tiles;           // scan_topk256: tiles handed out in chunks by a claim counter instead of round robin
    bh_u64* clk;             // optional diagnostics [grid][8] phase stamps + BH_TL_WORDS timeline words (scan_topk256.hip)
    // scan_topk256: queries of the tile that EXIST (0 = all of them).  A wave whose queries all lie beyond that count neither reads
    // fragments nor issues MFMAs nor filters — it keeps its share of the LDS-DMA refill, the rendezvous and the pacing / claim duties.
    // [1] = the second pass of a paired launch; qtile2 (optional) = that pass's query tile when it does not sit 256 rows behind qtile.
    int nq_valid[2] = {0, 0};
    const _Float16* qtile2 = nullptr;
    // filter pass of the exactness fall-back (bh_launch_filter_scan; unused by the top-k scans)
    const float* fix_thr = nullptr;  // [128] a row qualifies for query q iff its MFMA score >= fix_thr[q] (+inf: unused query)
    unsigned* fix_cnt = nullptr;     // [128] qualifying rows per query, zeroed by the caller; may exceed fix_cap (overflow)
    unsigned* fix_rows = nullptr;    // [128][fix_cap] their row indices
    unsigned fix_cap = 0;
};

// scan_topk.hip
hipError_t bh_launch_scan(const BhScanArgs& a, int dim_padded, int kp, int qw, int grid, hipStream_t stream);
hipError_t bh_launch_filter_scan(const BhScanArgs& a, int dim_padded, int grid, hipStream_t stream);
bool bh_scan_supports(int dim_padded, int kp, int qw);
// scan_topk192.hip (192 queries per pass on v_mfma_f32_16x16x32_f16; d = 768, k <= 56 only)
hipError_t bh_launch_scan192(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream);
bool bh_scan192_supports(int dim_padded, int kp);
// scan_topk256.hip in filter mode (the exact fall-back's filter pass, 256 queries per corpus pass)
hipError_t bh_launch_filter_scan256(const BhScanArgs& a, int dim_padded, int grid, hipStream_t stream);
bool bh_filter256_supports(int dim_padded);
// timeline diagnostics of scan_topk256.hip (ablate 5): workgroup 0 records, for BH_TL_TILES tiles from ordinal BH_TL_TILE0,
// per wave and stage five s_memtime values (before the vmcnt wait, after it, after the barrier, stage end, cycles spent in
// LDS-DMA issue) behind the [grid][8] phase stamps of BhScanArgs::clk
#define BH_TL_TILE0 1200
#define BH_TL_TILES 6
#define BH_TL_WORDS (8 * BH_TL_TILES * 2 * 5)
#define BH_BOOT_TILES 4  /* tiles a workgroup of scan_topk256.hip scans twice to seed the shared score bounds */
// scan_topk256: threshold slots per query (one per workgroup; the bound of a query is the 64th largest) and, behind the
// slot tables of a pass, one refined bound per query
#define BH_SLOTS256 256
// scan_topk256, dynamic tile distribution: smallest run of consecutive tiles a claim hands out (the LDS ring holds 2.5 tiles
// in flight at d = 768, 4 at d = 512, 5 at d = 384: a run must outlast the look-ahead by two tiles); the pass's claim counter
// sits behind its bounds
__host__ __device__ constexpr int bh_scan256_chunk_tiles(int /*dim_padded*/) { return 8; }
// tiles per workgroup that go round robin before the claimed part of the corpus starts (0: the corpus is too small, everything
// stays round robin); shared by the kernel and the host
__host__ __device__ inline int bh_scan256_round_robin_tiles(int n_tiles, int grid, int dim_padded) {
    const int rr = (int)(((long long)n_tiles * 7 / 8) / grid);
    return rr >= 4 * bh_scan256_chunk_tiles(dim_padded) ? rr : 0;
}
__host__ __device__ inline unsigned bh_scan256_first_claimed_tile(int n_tiles, int grid, int dim_padded) {
    const int rr = bh_scan256_round_robin_tiles(n_tiles, grid, dim_padded);
    return rr > 0 ? (unsigned)rr * (unsigned)grid : 0u;  // (0: never read)
}
// scan_topk256.hip (8 waves, two per SIMD, 256 queries per pass; d in {384, 512, 768})
hipError_t bh_launch_scan256(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream);
bool bh_scan256_supports(int dim_padded, int kp);
bool bh_scan256_pair_supports(int dim_padded, int kp);
hipError_t bh_launch_scan256_paired(const BhScanArgs& a, int dim_padded, int kp, int grid, hipStream_t stream);  // qsplit = 2: two passes, see scan_topk256.hip
int bh_scan256_tile(int dim_padded);  // queries per pass: 256, or 128 at d = 1024

struct BhMergeArgs {
    const bh_u64* partial;   // [G][BQ][KP] (x passes, pass_stride keys apart)
    long long pass_stride;   // 0: the launch merges one pass; else keys between the list sets of consecutive passes
    int n_lists;             // G
    int bq;                  // BQ (stride between lists = BQ*KP)
    const _Float16* corpus;  // [*, D]
    long long n_rows;
    const _Float16* qtile;   // [BQ][D]
    int dim_padded;          // D
    int k;                   // results per query
    long long id_offset;
    float* out_scores;       // [nq_total][k] (already offset to this tile's first query)
    long long* out_ids;      // [nq_total][k]
    // exactness certificate (certify.hip): a row the scan did not keep has an MFMA score <= the KP-th kept one, hence a
    // canonical score <= that + err_coef * |q|; the query is certified when this stays below its k-th canonical score
    float err_coef;          // 2 d 2^-24 max|x| (bound of |MFMA fp32 score - canonical score| per unit |q|), 0 = no certificate
    unsigned* uncert;        // [nq_tile] out: 1 = not certified (already offset to this tile's first query), or null
    bh_u64* kth_key;         // [nq_tile] out: canonical key (score, row) of the k-th result, or null
    unsigned* n_uncert;      // host-mapped count of uncertified queries (system-scope atomic), or null
};
// merge_rescore.hip: one workgroup per query of the tile
hipError_t bh_launch_merge_rescore(const BhMergeArgs& a, int kp, int nq_tile, hipStream_t stream);

// merge_topk.hip: merge [n_lists][nq][k] (score, id) lists in canonical order
hipError_t bh_launch_merge_lists(const float* scores, const long long* ids, int n_lists, int nq, int k,
                                 float* out_scores, long long* out_ids, hipStream_t stream);

// certify.hip: largest row norm of the corpus (for the certificate's error bound) and the exact fall-back scan
hipError_t bh_launch_row_norm_max(const _Float16* rows, long long n, int dim_padded, unsigned* out_max_bits, hipStream_t stream);
#define BH_EXACT_BATCH 256      /* most uncertified queries per filter pass: one query tile of the 256-query kernel (scan_topk256.hip, d <= 768); 128 on scan_topk.hip */
#define BH_EXACT_CAP 65536      /* qualifying rows kept per query */
struct BhExactArgs {
    const _Float16* corpus;   // [n_rows][D]
    long long n_rows;
    int dim_padded;
    const _Float16* q;        // [nqf][D] the uncertified queries, gathered
    int nqf;                  // <= BH_EXACT_BATCH
    const bh_u64* kth_key;    // [nqf] a row qualifies iff its canonical key >= this one
    const unsigned* rows;     // [nqf][BH_EXACT_CAP] rows the filter pass let through
    const unsigned* cnt;      // [nqf] how many (values above the cap are clamped)
    bh_u64* out_keys;         // [nqf][BH_EXACT_CAP] out: canonical key of rows[q][i] if it qualifies, else 0
};
// canonical (sequential fp64) score of every row the filter pass let through
hipError_t bh_launch_exact_rescore(const BhExactArgs& a, hipStream_t stream);
// gathers a batch of uncertified queries: rows -> q_out [tile][D] (zero beyond nb), k-th keys, filter thresholds (tile = the
// filter kernel's query tile, 128 or 256 <= BH_EXACT_BATCH)
hipError_t bh_launch_exact_prepare(const _Float16* qbuf, const int* todo, int nb, int tile, const bh_u64* kth_all, float err_coef, int dim_padded,
                                   _Float16* q_out, bh_u64* kth_out, float* thr_out, hipStream_t stream);

// list j of (src_s, src_i) [n][k] -> row todo[j] of (out_s, out_i) [.][k] (device or pinned host memory)
hipError_t bh_launch_scatter_lists(const float* src_s, const long long* src_i, const int* todo, int n, int k, float* out_s, long long* out_i,
                                   hipStream_t stream);

// convert.hip: dtype conversion / padding / normalisation
hipError_t bh_launch_convert_rows(const void* src, int src_dtype /*0=f16,1=f32*/, long long n, int dim,
                                  _Float16* dst, int dim_padded, hipStream_t stream);
hipError_t bh_launch_l2_normalize_rows(_Float16* rows, long long n, int dim, int dim_padded, hipStream_t stream);
hipError_t bh_launch_fill_u32(unsigned* p, long long n, unsigned v, hipStream_t stream, long long period = 0, long long special_at = 0,
                              unsigned v_special = 0);

// ---------------------------------------------------------------------------------------------
// bi-encoder forward pass (gemm_f16.hip, attention.hip, encoder_ops.hip; orchestrated by encoder.hip)

struct BhGemmArgs {
    const _Float16* A;  // [M][K], row stride lda (activations; K contiguous)
    long long lda;
    const _Float16* B;  // [N][K], row stride ldb (weights as HF stores them: [out][in])
    long long ldb;
    _Float16* C;  // [M][N], row stride ldc
    long long ldc;
    const _Float16* bias;      // bias_mode 1: [N] added per column; 2: [M] added per row; 0: none
    const _Float16* residual;  // [M][N] row stride ldr, or null
    long long ldr;
    int M, N, K;  // K % 64 == 0; lda, ldb, ldc, ldr % 8 == 0
    int bias_mode;
    int gelu;    // erf-GELU on the result
    int swiglu;  // persistent kernel, whole 256 x 256 tiles, bias per column: columns are (gate, up) pairs, C is [M][N / 2] = silu(gate) * up;
                 // 2 = the same fold with an erf-GELU gate (16x16x32 kernel only: ask bh_gemm_geglu_fusable() first)
    int swap_b;  // filled by the launcher: direction of v_permlane32_swap on this device
    // segmented-max epilogue (persistent kernel, SPLADE head): C is not stored; relu(C + bias) is max-reduced over the
    // COLUMNS (packed tokens) of each sequence into seg_out[sequence][m] (uint32 view of non-negative floats, zeroed
    // by the caller).  seg_grp[n / 8] = sequence << 4 | number of valid tokens among columns 8(n/8) .. 8(n/8)+7.
    const int* seg_grp;
    unsigned* seg_out;
    long long ld_seg;
    long long c_block_rows;  // != 0: C is stored blocked by 64 columns with this many rows per block (persistent kernel only)
    int stagger_phases, stagger_unit, stagger_first_round;  // start stagger of the first round of blocks (0 = off)
    // bh_launch_gemm_f16_batched: `batch` independent problems of one shape in ONE launch of the persistent kernel; problem b
    // reads A + b * batch_stride_a, B + b * batch_stride_b and writes C + b * batch_stride_c (strides in elements)
    int batch;
    long long batch_stride_a, batch_stride_b, batch_stride_c;
    // Fused LayerNorm (persistent kernel with full-line stores only; encoder.hip option ln_fused):
    //   ln_stats != null (BH_EPI_LNA): the token operand (A rows with bias_mode 1, B rows = C columns with bias_mode 2) is a
    //     pre-LayerNorm tensor; ln_stats = float2 (mean, 1 / sqrt(var + eps)) per token, ln_c[f] = sum_k W'[f][k] per output feature of
    //     the FOLDED weight operand W' = W o gamma, `bias` = the folded bias b + W beta.
    //   stats_out != null (BH_EPI_RESLN; bias_mode 1, `residual` set): C = fp16(A B^T + bias) + R with R = residual, or — res_stats
    //     set — (residual - mean) rstd res_gamma + res_beta; stats_out[row][N / 64] float2 (sum, sum of squares) of the stored row
    //     over each 64-column slice.
    const float* ln_stats;
    const _Float16* ln_c;
    const float* res_stats;
    const _Float16 *res_gamma, *res_beta;
    float* stats_out;
    int tail_split;  // gemm_f16_p16.h: a last round of at most half the workgroups' worth of tiles is cut into 2 or 4 sub-tiles along M
    // Rotary positions applied to the OUTPUT (the Q | K projection of NomicBert / gte: every 64-column slice is one head of Q or K): set
    // rot_pos (position of every output row) and rot_cs ([rot_max_pos][64] fp32: 32 cosines | 32 sines) and bh_launch_gemm_f16 returns the
    // ROTATED projection — in the 16x16x32 kernel's epilogue for the rows of whole tiles (a lane's partner column, 32 further, is its own
    // accumulator two feature blocks on: no exchange), by bh_launch_rotary for whatever rows another kernel computed — the same bits either way.
    // Needs bias_mode 1, N % 64 == 0, ldc == N, no other epilogue.
    const int* rot_pos = nullptr;
    const float* rot_cs = nullptr;
    int rot_max_pos = 0;
};
// whether bh_launch_gemm_f16 would run an M x N problem on the persistent kernel's full-line-store path (the fused-LayerNorm epilogues exist there only)
bool bh_gemm_ln_fusable(int M, int N, bool blocked_out);
// variant 0 = auto; see gemm_f16.hip for the explicit tile configurations
hipError_t bh_launch_gemm_f16(const BhGemmArgs& a, int variant, hipStream_t stream);
// C_b = A_b . B_b^T for b < a.batch, whole 256 x 256 tiles only (M, N % 256 == 0, K % 64 == 0), no bias / residual / GELU:
// the per-head position GEMMs of the disentangled attention (encoder.hip) — 16 heads x 44 tiles fill the chip, one head's
// 44 tiles (K = 64: a single pipeline stage each) are a launch-latency-bound 8.7 us
hipError_t bh_launch_gemm_f16_batched(const BhGemmArgs& a, hipStream_t stream);
hipError_t bh_gemm_probe_permlane(hipStream_t stream);
int bh_gemm_swap_mode();
void bh_gemm_set_stagger(int phases, int pct);  // bench knob: start stagger of the first round of blocks
void bh_gemm_set_gelu_nontemporal(int on);      // A/B knob: non-temporal stores of the bias + GELU output (default on)
void bh_gemm_set_mfma16(int on);                // bias (+ GELU) projections on the 16x16x32 persistent kernel (gemm_f16_p16.h)
void bh_gemm_set_tail_split(int on);            // gemm_f16_p16.h: sub-tiles for a short last round (default on)
void bh_gemm_set_rotary_fused(int on);          // BhGemmArgs::rot_pos: rotary positions in the 16x16x32 kernel's epilogue (default on; 0 = standalone kernel for all rows)
void bh_gemm_set_full_line_stores(int on);      // persistent kernel: outputs through LDS as whole 128-byte lines (gemm_f16_persist.h PST bit 32)

struct BhAttnArgs {
    const _Float16* qk;  // [tokens][2*d_model]: queries in columns [0, d), keys in [d, 2d)
    long long ldqk;
    const _Float16* vt;  // values, transposed: vt_blocked == 0: [d_model][ldvt] (column = token);
                         // vt_blocked != 0: [tokens / 64][d_model][64] (64-token blocks, each d_model x 64)
    long long ldvt;
    int vt_blocked;
    _Float16* ctx;  // [tokens][d_model]
    long long ldc;
    const long long* seq_off;  // [batch] first row of each sequence (multiple of 8)
    const int* seq_len;        // [batch] tokens per sequence (>= 1)
    int d_model;
    int v_lds_off;       // filled by the launcher: byte offset of the V^T image in LDS
    const int* seq_idx;  // filled by the launcher: optional indirection blockIdx.y -> sequence (length buckets)
    const float* alibi = nullptr;  // [n_heads] ALiBi slope per head (attention.hip only; null = none): score -= slope |i - j|
    // disentangled (DeBERTa-v2/v3) attention, attention_rel.hip only: score[i][j] = (Q_i.K_j + c2p[i][t(i-j)] + p2c[j][t(i-j)]) * scale
    const _Float16* c2p = nullptr;  // [n_heads][tokens][rel_ld]: Q_i . Kr[p]  (Kr = key_proj(rel_embeddings))
    const _Float16* p2c = nullptr;  // [n_heads][tokens][rel_ld]: K_j . Qr[p]  (Qr = query_proj(rel_embeddings))
    long long rel_ld = 0;           // 2 * span
    long long rel_head_stride = 0;  // tokens * rel_ld
    const int* rel_idx = nullptr;   // [2 * rel_center + 1] t(delta) at index delta + rel_center
    int rel_center = 0;             // max sequence length - 1
    float rel_scale = 0.f;          // 1 / sqrt(3 * head_dim)
    int rel_lds_off = 0;            // filled by the launcher: byte offset of the index table in LDS
    int win_lds_off = 0;            // filled by the launcher: byte offset of the waves' position windows in LDS (attention_rel.hip, WIN)
    int wide_stores = 0;            // filled by the launcher (attention_rel.hip): context rows as 16-byte stores
};
void bh_attention_rel_set_wide_stores(int on);  // option "attention_rel_wide_stores" (default on since round 5; 0 = 8-byte stores)
hipError_t bh_launch_attention(const BhAttnArgs& a, int batch, int n_heads, int max_len, hipStream_t stream);
hipError_t bh_launch_attention_bucketed(const BhAttnArgs& a, const int* seq_idx_dev, int n_short, int n_long,
                                        int max_len_long, int n_heads, hipStream_t stream, int short_max = 128,
                                        hipStream_t stream_long = nullptr);
// attention_rel.hip: DeBERTa-v2/v3 disentangled attention (one launch, 8-wave workgroups)
hipError_t bh_launch_attention_rel(const BhAttnArgs& a, int batch, int n_heads, int max_len, hipStream_t stream);

struct BhEmbedArgs {
    const int* tok;  // [n_rows] token id / position id / token-type id of each packed row
    const int* pos;
    const int* typ;
    long long n_rows;
    int d;
    float eps;
    const _Float16 *word, *position, *type, *gamma, *beta;
    _Float16* out;  // [n_rows][d]
};
hipError_t bh_launch_embed_ln(const BhEmbedArgs& a, hipStream_t stream);

struct BhLnArgs {
    const _Float16* in;
    const _Float16* residual;  // optional second addend (same shape), may alias out
    _Float16* out;             // may alias in
    long long n_rows;
    int d;
    float eps;
    const _Float16 *gamma, *beta;
    int small_regs = -1;  // 1: the 32-register kernel (768-wide rows: fits beside a persistent GEMM workgroup); 0: the general
                          // kernel; -1: the process default (option ln_small, default 1)
};
hipError_t bh_launch_layernorm(const BhLnArgs& a, hipStream_t stream);
void bh_ln_set_small(int on);  // process default of BhLnArgs::small_regs = -1 (bh_set_option "ln_small")

// Fused LayerNorm (encoder.hip, option ln_fused): the two small kernels beside the GEMM epilogues of gemm_f16_persist.h.
//   bh_launch_ln_fold      once per weight, at commit: W'[n][k] = fp16(W[n][k] gamma[k]), c[n] = fp16(sum_k W'[n][k]) (the sum of the
//                          ROUNDED values the MFMA multiplies), b'[n] = fp16(bias[n] + sum_k W[n][k] beta[k]) (fp32 sums)
//   bh_launch_ln_finalize  per producing GEMM: stats[row] = (mean, 1 / sqrt(var + eps)) from the row's n_part (sum, sum of squares)
//                          slices, added in slot order (bit-reproducible); var = E[x^2] - mean^2, clamped at 0
struct BhLnFoldArgs {
    const _Float16* w;      // [n][k]
    const _Float16 *gamma, *beta, *bias;  // [k], [k], [n]
    _Float16* w_out;        // [n][k]
    _Float16 *c_out, *bias_out;  // [n]
    int n, k;
};
hipError_t bh_launch_ln_fold(const BhLnFoldArgs& a, hipStream_t stream);
struct BhLnFinalizeArgs {
    const float* partial;   // [n_rows][n_part] float2
    float* stats;           // [n_rows] float2
    long long n_rows;
    int n_part, d;
    float eps;
};
hipError_t bh_launch_ln_finalize(const BhLnFinalizeArgs& a, hipStream_t stream);

// rotary positions (NomicBert; bh_encoder_config.rotary_theta): rows of [Q | K] rotated in place by their token index
struct BhRotaryArgs {
    _Float16* qk;          // [n_rows][2 * n_heads * 64]: n_heads query slices, then n_heads key slices, 64 dims each
    const int* pos;        // [n_rows] token index of the row inside its sequence (alignment rows: 0 = identity)
    const float* cos_sin;  // [max_pos][64]: 32 cosines, then 32 sines, of pos * theta^(-2j / 64)
    long long n_rows;
    int n_heads;
    int max_pos;           // rows of the table (positions beyond it are clamped: the caller checks seq_len against it)
};
hipError_t bh_launch_rotary(const BhRotaryArgs& a, hipStream_t stream);

// gated feed-forward (NomicBertMLP; bh_encoder_config.ffn_gated): out = silu(gate) * up
struct BhSwigluArgs {
    const _Float16* gu;  // [n_rows][2 f]: (gate, up) pairs — column 2 j = gate j, column 2 j + 1 = up j
    _Float16* out;       // [n_rows][f]
    long long n_rows;
    int f;               // multiple of 8
    int act = 0;         // the gate's activation: 0 = SiLU (NomicBert), 1 = erf-GELU (the "new" architecture of gte-*-en-v1.5)
};
hipError_t bh_launch_swiglu(const BhSwigluArgs& a, hipStream_t stream);

struct BhPoolArgs {
    const _Float16* x;  // [tokens][d]
    _Float16* out;      // [batch][d]
    const long long* seq_off;
    const int* seq_len;
    int batch, d;
    int mode;  // 0 = CLS (first token), 1 = mean
    int l2_normalize;
};
hipError_t bh_launch_pool(const BhPoolArgs& a, hipStream_t stream);

struct BhUnpackArgs {
    const _Float16* x;  // [tokens][d]
    _Float16* out;      // [batch][seq_len_padded][d]
    const int* slot;    // [batch*seq_len_padded] packed row or -1
    int batch, seq_len_padded, d;
};
hipError_t bh_launch_unpack(const BhUnpackArgs& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// sparse (SPLADE) retrieval: csr_topk.hip, orchestrated by sparse.hip

struct BhCsrScanArgs {
    const unsigned* entries;    // [nnz] uint16 term id | fp16 weight << 16, rows sorted by term id
    const long long* row_ptr;   // [n_rows + 1]
    long long n_rows;
    const unsigned* bitmap;     // [n_words] bit t set <=> some query of the tile uses term t
    const unsigned short* prefix;  // [n_words] number of set bits in the preceding words (= slot of the word's first term)
    const _Float16* W;          // [(n_slots + 1)][64] tile weights: W[slot][query]; the last row is zeros
    int n_words, n_slots;
    int off_prefix, off_w, off_thr;  // byte offsets of the LDS images
    bh_u64* cand;               // [grid * 16][64][2 * KP] per-wave candidate buffers (scratch)
    bh_u64* partial;            // [grid][64][KP] out: per-workgroup sorted best-KP keys
    unsigned* gthr;             // [64] chip-wide score bounds (ordf), initialised to BH_ORD_NEG_INF
};
hipError_t bh_launch_csr_scan(const BhCsrScanArgs& a, int kp, int grid, size_t smem, hipStream_t stream);

struct BhCsrMergeArgs {
    const bh_u64* partial;  // [n_lists][64][KP]
    int n_lists;
    const unsigned* entries;
    const long long* row_ptr;
    long long n_rows;
    const _Float16* q_dense;  // [tile queries][vocab] fp16
    int vocab;
    int k;
    long long id_offset;
    float* out_scores;    // already offset to the tile's first query
    long long* out_ids;
    int floor_zero;                // fill lists shorter than k with the lowest absent rows at score 0 (non-negative data)
};
hipError_t bh_launch_csr_merge_rescore(const BhCsrMergeArgs& a, int kp, int nq_tile, hipStream_t stream);

#define BH_CSR_MFMA_QUEUE 256       /* pending-hit ring entries per wave (8 bytes each) */
#define BH_CSR_MFMA_WAVE_LDS 14336  /* per wave: D tile 4 KiB + S tile 8 KiB + hit queue 2 KiB */
#define BH_CSR_HEAD_TERMS 64        /* corpus-head terms stored as a dense tile per 32-document group */
#define BH_CSR_HEAD_DWORDS 1024     /* = 32 documents x 64 terms x 2 bytes */
struct BhCsrMfmaArgs {
    const unsigned* entries;
    const long long* row_ptr;
    long long n_rows;
    const unsigned* bitmap;        // [n_words]
    const unsigned short* prefix;  // [n_words]
    const unsigned* sinfo;         // [n_slots] head: 0x80000000 | head index; tail: pair offset << 8 | pair count
    const unsigned* pairs;         // [n_pairs] tail pairs: fp16 weight << 16 | query
    const _Float16* WhT;           // [64 queries][64 head terms] weights of the head terms
    const _Float16* WgT;           // [64 queries][64 corpus-head terms] (head_dwords != 0)
    int head_dwords;               // 0: plain CSR stream; BH_CSR_HEAD_DWORDS: every 32-document group's entries are preceded by its
                                   // dense corpus-head tile and `entries` / `row_ptr` are the TAIL stream (csr_mfma.hip header)
    int n_words, n_slots, n_pairs;
    int off_prefix, off_sinfo, off_pairs, off_thr, off_tiles;  // byte offsets of the LDS images
    bh_u64* cand;                  // [grid * 8][64][2 * KP]
    bh_u64* partial;               // [grid][64][KP]
    unsigned* gthr;                // [64 queries][64 slots] threshold slot table (ordf), initialised to BH_ORD_NEG_INF
    int floor_zero;                // non-negative data: candidates need score > 0 (zero-score rows are filled in by the merge)
    int skip_final;                // pre-pass launch: fill the slot table only, leave no candidate lists
    int stats_mode;                // BH_SPARSE_STATS: 1 = count events (global atomics: distorts timing), 2 = phase timers of sampled waves
    unsigned* stats;               // optional diagnostics (BH_SPARSE_STATS): [0] groups with hits [1] appended [2] compactions [3] polls
    int ablate;                    // bench-only: 1 = no scatter, 2 = no candidate handling (results invalid)
};
hipError_t bh_launch_csr_scan_mfma(const BhCsrMfmaArgs& a, int kp, int grid, size_t smem, hipStream_t stream);
// csr_head.hip: split a CSR corpus into the corpus-head tiles + tail stream the MFMA scan reads
hipError_t bh_launch_csr_tail_count(const unsigned* entries, const long long* row_ptr, long long n_rows, const unsigned char* head_slot,
                                    unsigned* tail_cnt, hipStream_t stream);
hipError_t bh_launch_csr_split(const unsigned* entries, const long long* row_ptr, const long long* row_ptr2, long long n_rows,
                               const unsigned char* head_slot, unsigned* stream_out, hipStream_t stream);
void bh_sparse_set_kernel(int which);  // 1 = csr_mfma.hip (default), 0 = csr_topk.hip
void bh_sparse_set_ablate(int bits);   // bench-only
void bh_sparse_set_head(int on);       // 1 = MFMA scan on the corpus-head tiles + tail stream (default), 0 = plain CSR

struct BhSpladeFinishArgs {
    const unsigned* seg;  // [batch][ld_seg] max over the sequence's tokens of relu(logit), as uint32 bit patterns
    long long ld_seg;
    _Float16* out;        // [batch][vocab] = log(1 + max)
    int batch, vocab;
};
hipError_t bh_launch_splade_finish(const BhSpladeFinishArgs& a, hipStream_t stream);

struct BhClsHeadArgs {
    const _Float16* x;         // packed hidden states [rows][d]
    const long long* seq_off;  // [batch] first packed row of each sequence (= its [CLS] token)
    const _Float16* wp;        // [d][d]  BertPooler.dense.weight
    const _Float16* bp;        // [d]
    const _Float16* wc;        // [n_labels][d]  classifier.weight
    const _Float16* bc;        // [n_labels]
    float* out;                // [batch][n_labels] logits, fp32
    float* pooled;             // [batch][d] scratch: the pooler's output, fp32
    int batch, d, n_labels;
    int activation = 0;        // 0 = tanh (BertPooler), 1 = erf-GELU (DeBERTa's ContextPooler)
};
hipError_t bh_launch_cls_head(const BhClsHeadArgs& a, hipStream_t stream);

hipError_t bh_launch_scatter_f16(unsigned short* dst, const unsigned long long* pos, const unsigned short* val, int n,
                                 hipStream_t stream);

// whether bh_launch_gemm_f16 can fold a GELU-gated feed-forward in its epilogue under the current process options (BhGemmArgs::swiglu = 2
// exists on the 16x16x32 kernel's through-LDS store route only); false = the caller runs the plain GEMM + bh_launch_swiglu (act = 1)
bool bh_gemm_geglu_fusable();
