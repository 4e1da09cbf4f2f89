/*
 * bergen_hip.h — C ABI of the MI355X-native dense-retrieval backend for BERGEN.
 *
 * The reference (naver/bergen) has no FFI: its retrieval hot path is Python calling
 * torch ops.  This header is the boundary a replacement exports so that the reference's
 * `modules.retrieve.Retrieve` (stage seam, reference modules/retrieve.py:20-108) can be
 * backed by hand-written gfx950 kernels.  Each entry point cites the reference code it
 * replaces.  INTEGRATION.md shows the ctypes binding a BERGEN maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative bh_status; it never throws;
 *     a thread-local message is available from bh_last_error().
 *   - host buffers are owned by the caller and must stay alive for the duration of the
 *     call; all calls are synchronous (they return after the device work has finished).
 *   - device memory and the opaque bh_index are owned by the library.
 *   - matrices are row-major, C-contiguous.
 *   - there is NO CPU fallback in this library: without a HIP device bh_init fails.
 */
#ifndef BERGEN_HIP_H
#define BERGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = major * 1000 + minor * 10 + patch.  It is bumped whenever a struct of this header changes size or layout or an
 * entry point changes meaning; a caller compares bh_version() with the BH_VERSION it was compiled against before any other
 * call.  History: 130 = 0.1.3 (round 2: 256-query-tile scan, clock / timeline diagnostics); 140 = 0.1.4 (round 4): every
 * struct the library WRITES or READS through a caller's pointer now starts with `struct_size` (the caller's sizeof — the
 * library never touches bytes beyond it, so a caller built against an older, shorter struct stays memory-safe and a newer,
 * longer one gets zeros in the fields this build does not know); bh_counters grew the fall-back / tail-pass fields (they were
 * added under 130 without a bump: VERDICT r3); + bh_index_set_option / bh_sparse_set_option (per-handle options),
 * bh_encoder_set_rel_index (DeBERTa); 141 = 0.1.4.1 (round 4): bh_counters grew paired_scan_ms / paired_launches at its END (option
 * pair256) and bh_encoder_config grew rotary_theta / ffn_gated at its END (NomicBert: rotary positions, gated SiLU feed-forward;
 * a shorter struct reads as plain BERT) + bh_op_rotary / bh_op_swiglu; a caller built against 140 keeps working memory-wise
 * (struct_size) but must be rebuilt to pass the version check; 142 = 0.1.4.2 (round 5): bh_encoder_counters grew ln_fused at its END
 * (whether the last forward pass ran with the LayerNorms fused into the GEMM epilogues: encoder option "ln_fused") and bh_counters grew
 * balanced_scan_ms / balanced_queries at its END (option balance_tail); 143 = 0.1.4.3 (round 6): bh_encoder_config grew rotary_scale at its
 * END and accepts activation 0 with ffn_gated 1 (the GELU-gated feed-forward + NTK-scaled rotary positions of Alibaba-NLP/gte-*-en-v1.5,
 * config/retriever/gte-base-en-v1.5.yaml), alibi (jina-embeddings-v2) + bh_op_gated_act (bh_op_swiglu with the gate's activation as an argument). */
#define BH_VERSION 143

typedef enum bh_status {
    BH_OK = 0,
    BH_EINVAL = -1,      /* invalid argument */
    BH_EHIP = -2,        /* HIP runtime error */
    BH_ENOMEM = -3,      /* out of device memory */
    BH_EINCOMPLETE = -4, /* index rows missing (reference: IOError at retrieve.py:165-166) */
    BH_EUNSUPPORTED = -5 /* unsupported k / dim / dtype */
} bh_status;

typedef enum bh_dtype { BH_F16 = 0, BH_F32 = 1 } bh_dtype;
typedef enum bh_metric {
    BH_METRIC_IP = 0, /* DotProduct.sim, reference models/retrievers/dense.py:77-81 */
    BH_METRIC_COS = 1 /* CosineSim.sim,  reference models/retrievers/dense.py:83-89 */
} bh_metric;

typedef struct bh_index bh_index;

/* Kernel-time / traffic counters of the most recent bh_search* call on an index
 * (consumed by bench.py's `roofline` object). */
typedef struct bh_counters {
    int32_t struct_size;      /* IN: sizeof(bh_counters) as the CALLER was compiled; the library writes at most that many bytes */
    int32_t reserved_head;
    int64_t n_rows;           /* rows resident in this index */
    int32_t dim;              /* logical dim */
    int32_t dim_padded;       /* row stride in elements (multiple of 64) */
    int32_t query_tile;       /* queries per corpus pass (Bq_tile) */
    int32_t n_passes;         /* corpus passes of the last search = ceil(nq / query_tile) */
    int32_t n_workgroups;     /* persistent workgroups of the scan kernel */
    int32_t k_padded;         /* candidate list length carried through the scan (k + margin) */
    double scan_ms;           /* sum of scan-kernel durations (HIP events on the launch stream) */
    double merge_ms;          /* sum of merge+rescore kernel durations */
    double total_ms;          /* first launch -> last kernel done (HIP events), excludes H2D/D2H */
    double algorithmic_bytes; /* sum over passes of N*d*2 + Bq*d*2 + Bq*k*12 */
    double shader_mhz;        /* effective shader clock during the last scan launch (s_memtime per 100 MHz tick); 0 = not measured */
    int64_t uncertified_queries; /* queries whose top-k the exactness certificate could not prove from the scan's candidate
                                    lists (more than KP - k rows within MFMA rounding error of the k-th score); they were
                                    answered by the exact fall-back: an MFMA filter pass with a fixed threshold (one more
                                    corpus pass per 128 such queries) + canonical re-scoring of the rows it lets through */
    double exact_ms;          /* time spent in that fall-back (wall, included in total_ms) */
    double tail_scan_ms;      /* part of scan_ms spent in a last pass that ran on the 128-query kernel (<= 128 queries left), else 0 */
    int32_t tail_query_tile;  /* 128 when the last pass ran on the 128-query kernel, else 0 */
    int32_t reserved0;
    int64_t exact_passes;     /* filter passes (corpus passes) the fall-back of the last search took */
    int64_t exact_rows_rescored; /* rows the filter passes let through, summed over the uncertified queries */
    double paired_scan_ms;    /* part of scan_ms spent in PAIRED launches (option pair256: two query-tile passes per launch, the
                                 corpus stream of the second served from the L2 of the first's XCD), else 0 */
    int32_t paired_launches;  /* number of such launches in the last search (each counts as 2 in n_passes) */
    int32_t reserved1;
    /* ---- since BH_VERSION 142 ---- */
    double balanced_scan_ms;  /* part of paired_scan_ms spent in the BALANCED launch: the queries left behind the last full pair of
                                 passes (more than one tile, fewer than two) cut into two passes of about half each, run as one more
                                 paired launch in which waves without any query sit the tiles out (option balance_tail); else 0 */
    int32_t balanced_queries; /* queries that launch served (0 = no balanced launch in the last search) */
    int32_t reserved2;
} bh_counters;

/* Library / device lifecycle ------------------------------------------------------- */

/* Select the HIP device for the calling thread and verify it is a gfx950 part.
 * Replaces: `tensor.to('cuda')` device selection, reference modules/retrieve.py:76,124. */
int bh_init(int device_id);
int bh_version(void);
const char* bh_last_error(void);
/* Number of HIP devices visible; 0 (not an error) when there is none. */
int bh_device_count(void);

/* Flat index (resident corpus) ----------------------------------------------------- */

/* Allocate an n_rows x dim index in HBM.  dtype is the storage type (BH_F16 only in this
 * version: the reference stores fp16 embeddings, dense.py:16).  Replaces the host-side
 * list of chunk tensors built at reference modules/retrieve.py:84-90. */
int bh_index_create(bh_index** out, int64_t n_rows, int32_t dim, int32_t dtype, int32_t metric);

/* Copy rows [row0, row0+n) from a HOST buffer of `src_dtype` (converted to fp16 with
 * round-to-nearest-even, like torch .half()).  Replaces the per-query-chunk
 * `emb_chunk.to('cuda')` of reference modules/retrieve.py:153 (done once, not per chunk). */
int bh_index_upload(bh_index* ix, int64_t row0, const void* host_rows, int64_t n, int32_t src_dtype);
/* Same, from a DEVICE buffer (e.g. encoder output that never left HBM). */
int bh_index_upload_device(bh_index* ix, int64_t row0, const void* dev_rows, int64_t n, int32_t src_dtype);

/* Declare the index complete.  Fails with BH_EINCOMPLETE if fewer than n_rows rows were
 * uploaded (reference raises IOError, retrieve.py:165-166).  For BH_METRIC_COS every row is
 * L2-normalised once here (reference renormalises every chunk on every call, dense.py:87-88). */
int bh_index_finalize(bh_index* ix);

/* Number of rows uploaded so far. */
int64_t bh_index_rows_uploaded(const bh_index* ix);

void bh_index_destroy(bh_index* ix);

/* Exact brute-force search: for each of nq queries the k rows with the largest inner
 * product, in the canonical total order (score descending, row index ascending).
 * Scores are the canonical fp32 scores: round-to-nearest fp32 of the fp64 sum
 * sum_{j=0..dim-1} q[j]*x[j] taken in index order over the fp16-stored values.
 * out_ids = id_offset + row index (int64); slots beyond the number of rows hold id -1
 * and score -inf.  Replaces reference Retrieve.load_collection_and_retrieve
 * (modules/retrieve.py:146-185): similarity_fn + torch.topk per chunk + CPU merge. */
int bh_search(bh_index* ix, const void* q_host, int32_t q_dtype, int32_t nq, int32_t k,
              int64_t id_offset, float* out_scores, int64_t* out_ids);
/* Same with DEVICE-resident queries and outputs (feeds the RCCL all-gather without a host
 * round trip).  The output buffers may also be PINNED HOST memory (hipHostMalloc / torch pin_memory): the merge kernel
 * then writes the result lists there itself and no device-to-host copy follows.  Synchronous: the lists are complete on
 * return.  Every query's top-k is proven exact from the scan's candidate lists by a certificate; queries it cannot prove
 * are re-done exactly inside the call (MFMA filter pass + fp64 re-scoring) (bh_counters.uncertified_queries / exact_ms; option "certify"). */
int bh_search_device(bh_index* ix, const void* q_dev, int32_t q_dtype, int32_t nq, int32_t k,
                     int64_t id_offset, float* out_scores_dev, int64_t* out_ids_dev);

/* Merge n_lists partial top-k lists per query ([n_lists, nq, k] scores + ids, e.g. one per
 * shard / rank) into one [nq, k] list in the canonical order; entries with id < 0 are
 * ignored.  Replaces the host `torch.cat` + `torch.topk` + `gather` merge of reference
 * modules/retrieve.py:169-177.  Runs on the device. */
int bh_merge_topk(const float* scores, const int64_t* ids, int32_t n_lists, int32_t nq,
                  int32_t k, float* out_scores, int64_t* out_ids);
int bh_merge_topk_device(const float* scores_dev, const int64_t* ids_dev, int32_t n_lists,
                         int32_t nq, int32_t k, float* out_scores_dev, int64_t* out_ids_dev);

/* Counters of the last search on this index.  `out->struct_size` must hold the caller's sizeof(bh_counters) on entry
 * (BH_EINVAL when it is smaller than the round-2 prefix of the struct, 16 bytes); the library fills min(that, its own
 * sizeof) bytes and leaves the rest untouched. */
int bh_bench_counters(const bh_index* ix, bh_counters* out);

/* Diagnostics: copies up to max_words 64-bit words of the last scan launch's record (scan_topk256.hip: 8 words per
 * workgroup — 100 MHz ticks at entry / loop start / loop end / exit, shader cycles at loop start / end, candidates held —
 * then, with option "ablate" 32, workgroup 0's per-wave stage stamps; layout in bergen_amd/csrc/bh_kernels.h) into
 * `out`; returns the number of words written or a negative error code. */
int64_t bh_debug_scan_timeline(const bh_index* ix, uint64_t* out, int64_t max_words);

/* Tuning knobs; results are identical for every valid setting.  TWO LEVELS:
 *   bh_set_option        PROCESS-WIDE defaults.  Every handle that has no override of its own sees the new value from its
 *                        next search on; a running search is not affected (each search reads the options once, up front).
 *                        Not meant to be toggled concurrently with searches of OTHER threads' handles that rely on a
 *                        particular value: give those handles their own value with the per-handle call.
 *   bh_index_set_option  (and bh_sparse_set_option): PER-HANDLE override of the dense-search (sparse-search) options, visible to
 *                        that handle only — the independence of handles per GPU / thread SURVEY section 8b promises.
 *                        BH_OPTION_INHERIT as the value drops the override.
 * The encoder's options have always been per handle (bh_encoder_set_option); "gemm_stagger_*" are process-wide only.
 * Names (bench sweeps and A/B comparisons):
 * Dense scan: "query_tile" (128|256), "share_threshold" (0|1), "nontemporal" (0|1), "dma_interleave" (0|1, default 1),
 * "query_split" (1|2: paired workgroups share the corpus stream through L2), "pair_window" (0..64), "scan_kernel"
 * (3 = 256-query tile [128 at d = 1024], two waves per SIMD, where it applies [d in {384, 512, 768, 1024}] else the
 * 4-wave kernel, default; 2 = 192-query tile where it applies [d = 768, k <= 56]; 0 = 4-wave kernel, 128-query tile),
 * "dyn_tiles" (0|1, default 1: the last eighth of the corpus is handed out by a claim counter), "certify" (0|1, default
 * 1: exactness certificate + exact fall-back scan), "ring_variant" (0..7: bench-only variants of the selected kernel),
 * "tail128" (0|1, default 1), "filter256" (0|1, default 1: the exact fall-back's filter passes take 256 queries on the
 * 256-query kernel where it applies), "pair256" (0|1: paired workgroups for the 256-query kernel), "certificate_error_scale" (tests).
 * Sparse scan: "sparse_kernel"
 * (1 = csr_mfma.hip, default; 0 = csr_topk.hip).  Encoder GEMM: "gemm_stagger_phases", "gemm_stagger_pct".
 * "ablate" / "sparse_ablate" switch parts of the kernels OFF for profiling: results are INVALID while they are set. */
int bh_set_option(const char* name, int64_t value);
#define BH_OPTION_INHERIT INT64_MIN
int bh_index_set_option(bh_index* ix, const char* name, int64_t value);

/* Bi-encoder forward pass (BERT-architecture dense retrievers) ------------------------------ */

/* Architecture of HF `BertModel` as the reference loads it through AutoModel
 * (models/retrievers/dense.py:16): post-LN transformer encoder, absolute position embeddings,
 * head dim 64.  RetroMAE / contriever / e5 / bge checkpoints are all of this class. */
typedef struct bh_encoder_config {
    int32_t struct_size;     /* sizeof(bh_encoder_config) as the caller was compiled; fields beyond it read as 0 */
    int32_t n_layers;        /* num_hidden_layers */
    int32_t hidden;          /* hidden_size (multiple of 64; = n_heads * head_dim) */
    int32_t n_heads;         /* num_attention_heads */
    int32_t intermediate;    /* intermediate_size */
    int32_t vocab_size;
    int32_t max_position;    /* max_position_embeddings */
    int32_t type_vocab_size;
    int32_t activation;      /* 0 = erf-GELU ("gelu"); 1 = SiLU, with ffn_gated = 1 only.  With ffn_gated = 1 it is the GATE's activation */
    float ln_eps;            /* layer_norm_eps */
    int32_t head_dim;        /* 0 or 64: hidden / n_heads = 64.  8..56 (e5-small, bge-small, MiniLM: 32): the attention kernel
                                works on 64-wide heads, the CALLER stores query / key / value weights and biases zero-padded to
                                64 rows per head ([n_heads*64, hidden]; query rows scaled by sqrt(64 / head_dim), which turns
                                the kernel's 1/sqrt(64) into 1/sqrt(head_dim)) and attention.output.dense.weight zero-padded
                                to 64 columns per head ([hidden, n_heads*64]) */
    int32_t position_offset; /* added to the token index to form the position id: 0 = BERT, padding_idx + 1 (= 2) = RoBERTa /
                                XLM-R with right-padded inputs */
    /* ---- since BH_VERSION 141 (a shorter struct reads as zeros here: plain BERT) ---- */
    float rotary_theta;      /* 0: learned absolute positions (the position table is added to the embeddings).  > 0: rotary
                                positions (NomicBert: transformers modeling_nomic_bert.py:95-181, theta 1000): after the Q | K
                                projection every 64-dim head slice of a token's query and key is rotated by its token index t,
                                x' = x cos + rotate_half(x) sin with angles t * theta^(-2j / 64), j < 32; the caller uploads a
                                zero position table.  head_dim must be 64 */
    int32_t ffn_gated;       /* 0: H = act(X W1^T + b1), W1 = [intermediate][hidden].  1 (activation 1 = SiLU): gated feed-forward
                                H = silu(X Wg^T) * (X Wu^T) (NomicBertMLP, modeling_nomic_bert.py:266-279): the tensor
                                "intermediate.dense.weight" holds gate and up rows INTERLEAVED — row 2 j = gate row j, row
                                2 j + 1 = up row j, [2 * intermediate][hidden] —, "intermediate.dense.bias" likewise (2 *
                                intermediate entries): one GEMM yields (gate, up) column pairs, which the GEMM's epilogue folds.
                                activation 0 with ffn_gated 1 (since 143): H = gelu(X Wg^T + bg) * (X Wu^T + bu), the GEGLU feed-forward of
                                the "new" architecture (Alibaba-NLP/new-impl NewGatedMLP: up_gate_proj's first half is up, its second half
                                the gate; the CALLER interleaves them as above) */
    /* ---- since BH_VERSION 143 (a shorter struct reads as 0 = 1.0 here) ---- */
    float rotary_scale;      /* 0 or 1: plain rotary angles t * theta^(-2j / 64).  Else every angle is multiplied by it: NTK-scaled RoPE
                                (new-impl NTKScalingRotaryEmbedding: base' = base * factor goes into rotary_theta, the inverse frequencies
                                are divided by factor^(2 / 64) — rotary_scale = factor^(-2 / 64)) */
    int32_t alibi;           /* 1: ALiBi attention biases (jinaai/jina-embeddings-v2-*: remote JinaBert, position_embedding_type "alibi"):
                                score[i][j] = q_i . k_j / sqrt(head_dim) - slope_h |i - j|, the SYMMETRIC encoder form, slope_h the standard
                                ALiBi head slopes (2^(-8 (h + 1) / n) for n a power of two, interleaved from 2 n otherwise); the caller
                                uploads a zero position table.  0: none */
} bh_encoder_config;

typedef struct bh_encoder bh_encoder;

/* Counters of the most recent bh_encoder_forward (encoder roofline = MFMA). */
typedef struct bh_encoder_counters {
    int32_t struct_size;  /* IN: sizeof(bh_encoder_counters) as the caller was compiled; the library writes at most that many bytes */
    int32_t batch;        /* sequences */
    int32_t seq_len;      /* padded length of the input matrix */
    int64_t real_tokens;  /* tokens with attention_mask != 0 */
    int64_t packed_rows;  /* rows the kernels ran over (real tokens + alignment padding) */
    double forward_ms;    /* embedding -> pooled output, HIP events on the encoder's stream */
    double flops;         /* ALGORITHMIC flops over real tokens:
                             n_layers * (T*(8 d^2 + 4 d d_ff) + 4 d sum(len_s^2))
                             (+ T*(2 d^2 + 2 d vocab) for pool 3) */
    /* ---- since BH_VERSION 142 ---- */
    int32_t ln_fused;     /* 1 = the last forward pass ran WITHOUT standalone LayerNorm passes between its GEMMs (residual add +
                             row statistics in the producing GEMM's epilogue, normalisation folded into the consuming GEMM: option
                             "ln_fused" = 1 — off by default: measured 2 % slower —, BERT-type stacks, batches whose GEMMs fill the chip);
                             0 = one LayerNorm kernel per LayerNorm */
    int32_t reserved0;
} bh_encoder_counters;

/* Allocate an encoder (weights + workspace live in HBM, owned by the library).  Replaces
 * `AutoModel.from_pretrained(model_name, torch_dtype=float16)`, reference dense.py:16-20. */
int bh_encoder_create(bh_encoder** out, const bh_encoder_config* cfg);
/* Copy one weight tensor from HOST memory, addressed by its HF `BertModel.state_dict()` name
 * (e.g. "encoder.layer.3.attention.self.query.weight"); fp32 sources are rounded to fp16 like
 * `.half()`.  Shapes are HF's ([out, in] for Linear weights).
 * Optional masked-LM head (HF `BertForMaskedLM`, which the reference loads through
 * AutoModelForMaskedLM for SPLADE, models/retrievers/splade.py:17): the six
 * "cls.predictions.{transform.dense.{weight,bias}, transform.LayerNorm.{weight,bias},
 * decoder.{weight,bias}}" tensors ("cls.predictions.bias" is accepted as the decoder bias).  A head
 * without decoder.weight is tied to the word embeddings; a missing decoder bias is zero.
 * Optional sequence-classification head (HF `BertForSequenceClassification`, which the reference
 * loads through AutoModelForSequenceClassification for its cross-encoder reranker,
 * models/rerankers/crossencoder.py:18): "pooler.dense.{weight,bias}" and
 * "classifier.{weight,bias}" ([num_labels, hidden], num_labels <= 16). */
int bh_encoder_set_tensor(bh_encoder* enc, const char* name, const void* host, int32_t dtype, int64_t numel);
/* Check that every tensor of the architecture has been set (BH_EINCOMPLETE otherwise; when any
 * cls.predictions.* tensor was given, the head's transform weights must be complete too). */
int bh_encoder_commit(bh_encoder* enc);
/* name in {"gemm_variant" (0 = auto, 1..5 explicit tile configurations, 6 = generic bounds-checked kernel;
 * bench sweeps), "attn_short_len" (32..512, multiple of 32; default 128: longest sequence whose attention runs in a
 * 4-wave workgroup), "micro_batches" (1..4, default 2: batches of >= 8192 packed rows run their layer stack as that many
 * micro-batches on as many streams — results are bit-identical for every value), "vt_side_stream", "attn_side_stream",
 * "rel_batched_gemm"}. */
int bh_encoder_set_option(bh_encoder* enc, const char* name, int64_t value);
/* DeBERTa-v2 / v3 encoders (the reference's default reranker, config/reranker/debertav3.yaml:3, loaded through
 * AutoModelForSequenceClassification, models/rerankers/crossencoder.py:18): set option "rel_attention_span" (=
 * config.position_buckets, 2 * span relative positions; BEFORE the weights — it adds the tensors
 * "encoder.rel_embeddings.weight" [2 * span, hidden], "encoder.LayerNorm.{weight,bias}"), option "cls_activation" 1 (the
 * ContextPooler's erf-GELU instead of BertPooler's tanh), and hand over the relative index table
 * t(delta) = clamp(bucket(delta) + span, 0, 2 * span - 1) for delta = -(L - 1) .. L - 1 (n = 2 L - 1 entries, L = longest
 * sequence the encoder will see; bucket = make_log_bucket_position of transformers' modeling_deberta_v2.py:57-69).  The
 * attention is then the disentangled one: (Q K^T + c2p + p2c) / sqrt(3 * 64), modeling_deberta_v2.py:191-346.  Word
 * embeddings only: pass zero position / token-type tables (position_biased_input = false, type_vocab_size = 0). */
int bh_encoder_set_rel_index(bh_encoder* enc, const int32_t* table, int32_t n);

/* One forward pass over a HOST batch in the layout of an HF BatchEncoding (row-major
 * [batch, seq_len] int64; attention_mask / token_type_ids may be NULL = all ones / all zeros).
 * pool: 0 = ClsPooler (reference dense.py:71-75), 1 = MeanPooler (dense.py:64-69) -> out is
 * [batch, hidden] fp16; 2 = no pooling -> out is the padded last_hidden_state
 * [batch, seq_len, hidden] fp16 (zeros at padding); 3 = SPLADE: masked-LM head, then
 * max over the attended tokens of log(1 + relu(logit)) -> out is [batch, vocab_size] fp16
 * (replaces Splade.__call__, reference models/retrievers/splade.py:34-47; BH_EINCOMPLETE if the
 * head was not set; the [batch, seq_len, vocab] logits are never materialised).
 * 4 = sequence classification: tanh(BertPooler) on the first token, then the classifier ->
 * out is [batch, num_labels] FP32 logits (replaces CrossEncoder.__call__'s
 * `self.model(**kwargs).logits`, reference models/rerankers/crossencoder.py:34-38).
 * l2_normalize applies to pooled outputs (pool 0 / 1).
 * out is a device pointer when out_on_device != 0 (e.g. the rows of a bh_index, or a torch
 * tensor), else a host pointer.  Replaces Dense.__call__, reference dense.py:37-47. */
int bh_encoder_forward(bh_encoder* enc, const int64_t* input_ids, const int64_t* attention_mask,
                       const int64_t* token_type_ids, int32_t batch, int32_t seq_len, int32_t pool,
                       int32_t l2_normalize, void* out, int32_t out_on_device);
int bh_encoder_counters_get(const bh_encoder* enc, bh_encoder_counters* out);
void bh_encoder_destroy(bh_encoder* enc);

/* Op-level entry points over DEVICE pointers (parity tests, kernel micro-benchmarks).
 * bh_op_gemm_f16: C[M][N] = A[M][K] . B[N][K]^T (+bias: mode 1 per column [N], 2 per row [M])
 * (+residual[M][N]) (erf-GELU), fp16 in/out, fp32 accumulate; K % 64 == 0, ld* % 8 == 0.
 * repeats > 1 times the launches after the first (avg_ms, HIP events on the null stream). */
/* (gelu = 2: the gated fold instead — B's rows are (gate, up) pairs, C is [M][N / 2] = silu(gate) * up; M, N multiples of 256,
 * bias per column required) */
int bh_op_gemm_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                   const void* bias, int32_t bias_mode, const void* residual, int64_t ldr, int32_t M,
                   int32_t N, int32_t K, int32_t gelu, int32_t variant, int32_t repeats, float* avg_ms);
/* Packed variable-length self-attention, head dim 64: qk [tokens][2*64*n_heads] (queries | keys),
 * vt [64*n_heads][ldvt] (values, transposed), ctx [tokens][64*n_heads]. */
int bh_op_attention(const void* qk, int64_t ldqk, const void* vt, int64_t ldvt, void* ctx, int64_t ldc,
                    const int64_t* seq_off_dev, const int32_t* seq_len_dev, int32_t batch,
                    int32_t n_heads, int32_t max_len);
/* bh_op_rotary: rows of [Q | K] ([n_rows][2 * n_heads * 64] fp16, device) rotated IN PLACE by pos[row] (int32, device) — the
 * rotate-half RoPE of bh_encoder_config.rotary_theta.  bh_op_swiglu: out[n_rows][f] = silu(gu[row][2 j]) * gu[row][2 j + 1] over
 * gu [n_rows][2 f] fp16 (device; (gate, up) column pairs).  Kernel-level entry points for the parity tests, like the three around them. */
int bh_op_rotary(void* qk, int64_t n_rows, int32_t n_heads, const int32_t* pos, float theta, int32_t max_pos);
int bh_op_swiglu(const void* gu, void* out, int64_t n_rows, int32_t f);
/* the same fold with the gate's activation as an argument: act 0 = SiLU (bh_op_swiglu), 1 = erf-GELU (since BH_VERSION 143) */
int bh_op_gated_act(const void* gu, void* out, int64_t n_rows, int32_t f, int32_t act);
int bh_op_layernorm(const void* in, void* out, int64_t n_rows, int32_t d, float eps, const void* gamma,
                    const void* beta);
/* 0 / 1 = direction of v_permlane32_swap found on the device (diagnostic), -1 on failure. */
int bh_gemm_permlane_mode(void);

/* Sparse (SPLADE) index: resident CSR corpus + exact sparse search ------------------------- */

typedef struct bh_sparse_index bh_sparse_index;

/* n_rows documents over a vocabulary of `vocab` terms (<= 65535: stored ids are 16-bit, 0 is reserved), weights stored as fp16 (the
 * reference stores fp16 sparse COO chunks, modules/retrieve.py:138-139).  Replaces the host
 * list of sparse chunk tensors of reference modules/retrieve.py:84-90. */
int bh_sparse_create(bh_sparse_index** out, int64_t n_rows, int32_t vocab);
/* Append rows [row0, row0+n) given as a HOST CSR block: indptr[n+1] (starting at 0), term ids,
 * weights (fp16 or fp32; fp32 is rounded like .half()).  Rows must arrive in order
 * (row0 == rows uploaded so far); explicit zeros are dropped, rows are sorted by term id. */
int bh_sparse_upload_csr(bh_sparse_index* ix, int64_t row0, int64_t n, const int64_t* indptr,
                         const int32_t* terms, const void* values, int32_t val_dtype);
/* BH_EINCOMPLETE if fewer than n_rows rows were uploaded (reference IOError, retrieve.py:165-166). */
int bh_sparse_finalize(bh_sparse_index* ix);
int64_t bh_sparse_rows_uploaded(const bh_sparse_index* ix);
int64_t bh_sparse_nnz(const bh_sparse_index* ix);
/* Exact search with DENSE host queries [nq, vocab] (what the reference holds after
 * `load_embeddings(...).to_dense()`, retrieve.py:75-76): for each query the k documents with
 * the largest sparse dot product, canonical order (score desc, row asc).  Canonical score =
 * fp32(RNE) of the fp64 sum over the document's terms, in increasing term id, of
 * q[term] * weight (both as fp16 values).  k <= 4096 (k <= 120: one fused search; above that the
 * documents are searched range by range and the ranges' exact top-120 lists merged, as for the dense index).  Replaces Splade.similarity_fn
 * (models/retrievers/splade.py:55-56) + torch.topk + host merge (retrieve.py:146-185). */
int bh_sparse_search(bh_sparse_index* ix, const void* q_host, int32_t q_dtype, int32_t nq, int32_t k,
                     int64_t id_offset, float* out_scores, int64_t* out_ids);
/* Counters of the last bh_sparse_search (dim = vocab; algorithmic_bytes = passes * (nnz*4 + (N+1)*8)). */
int bh_sparse_counters(const bh_sparse_index* ix, bh_counters* out);
/* Per-handle override of "sparse_kernel" / "sparse_head" / "sparse_ablate" (see bh_set_option). */
int bh_sparse_set_option(bh_sparse_index* ix, const char* name, int64_t value);
void bh_sparse_destroy(bh_sparse_index* ix);

#ifdef __cplusplus
}
#endif
#endif /* BERGEN_HIP_H */
