// bh_kernels.h — host-visible launch interfaces of the gfx950 kernels (internal to the .so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long bh_u64;

struct BhScanArgs {
    const _Float16* corpus;  // [n_tiles*32][D] fp16, rows >= n_rows are zero padding
    long long n_rows;        // valid rows
    long long n_tiles;       // ceil(n_rows / 32)
    const _Float16* qtile;   // [BQ][D] fp16 query tile (zero rows beyond the valid queries)
    bh_u64* cand;            // [G][BQ][2*KP] candidate buffers (scratch)
    bh_u64* partial;         // [G][BQ][KP]   out: per-workgroup sorted best-KP keys
    unsigned* gthr;          // [BQ][64] threshold slot table (ordf), initialised to BH_ORD_NEG_INF
    int share;               // share thresholds between workgroups
    int nontemporal;         // nt cache policy on the corpus stream
    int ablate;              // bench-only kernel ablation (0 = production kernel)
    int ring_variant;        // bench-only LDS ring geometry selector for d=768 (0 = default 6 lines x 6)
};

// scan_topk.hip
hipError_t bh_launch_scan(const BhScanArgs& a, int dim_padded, int kp, int qw, int grid, hipStream_t stream);
bool bh_scan_supports(int dim_padded, int kp, int qw);

struct BhMergeArgs {
    const bh_u64* partial;   // [G][BQ][KP]
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
};
// merge_rescore.hip: one workgroup per query of the tile
hipError_t bh_launch_merge_rescore(const BhMergeArgs& a, int kp, int nq_tile, hipStream_t stream);

// merge_topk.hip: merge [n_lists][nq][k] (score, id) lists in canonical order
hipError_t bh_launch_merge_lists(const float* scores, const long long* ids, int n_lists, int nq, int k,
                                 float* out_scores, long long* out_ids, hipStream_t stream);

// convert.hip: dtype conversion / padding / normalisation
hipError_t bh_launch_convert_rows(const void* src, int src_dtype /*0=f16,1=f32*/, long long n, int dim,
                                  _Float16* dst, int dim_padded, hipStream_t stream);
hipError_t bh_launch_l2_normalize_rows(_Float16* rows, long long n, int dim, int dim_padded, hipStream_t stream);
hipError_t bh_launch_fill_u32(unsigned* p, long long n, unsigned v, hipStream_t stream);
