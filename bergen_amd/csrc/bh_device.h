// bh_device.h — device-side helpers shared by the gfx950 kernels.
//
// Candidate keys.  A (score, row) pair travels through the scan / merge kernels as one
// u64 key whose unsigned order IS the canonical retrieval order (score descending, row
// index ascending — SURVEY §0 D4):  key = ordf(score) << 32 | ~row.  key 0 = empty slot.
//
// Wave-level sorting.  All lists handled here have 64*EPL elements held EPL-per-lane by one
// 64-wide wavefront, element index i = r*64 + lane.  Compare-exchange partners at distance
// < 64 are fetched with a cross-lane shuffle, at distance >= 64 they are another register of
// the same lane, so every register index is a compile-time constant (no scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define BH_WAVE 64

// Monotone map float -> uint32 (a < b  <=>  ordf(a) < ordf(b) for non-NaN).
__device__ __forceinline__ unsigned bh_ordf(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bh_unordf(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
#define BH_ORD_NEG_INF 0x007fffffu /* bh_ordf(-inf) */

__device__ __forceinline__ u64 bh_make_key(float score, unsigned row) {
    return ((u64)bh_ordf(score) << 32) | (u64)(~row);
}
__device__ __forceinline__ float bh_key_score(u64 key) { return bh_unordf((unsigned)(key >> 32)); }
__device__ __forceinline__ unsigned bh_key_row(u64 key) { return ~(unsigned)key; }

__device__ __forceinline__ u64 bh_shfl64(u64 v, int src_lane) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
    return ((u64)hi << 32) | lo;
}

// One compare-exchange stage of a bitonic network over n = 64*EPL elements.
//   j    : partner distance (power of two)
//   kblk : size of the bitonic block being built; element i sorts "descending" inside its
//          block when (i & kblk) == 0.  Pass kblk = 0 for a pure descending merge stage.
template <int EPL>
__device__ __forceinline__ void bh_bitonic_stage(u64 (&e)[EPL], int lane, int j, int kblk) {
    if (j >= 64) {
        const int dr = j >> 6;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            if ((r & dr) == 0) {  // r is the lower index of the pair (r, r | dr)
                const int i = r * 64 + lane;
                const bool desc = (kblk == 0) || ((i & kblk) == 0);
                u64 a = e[r], b = e[r | dr];
                u64 hi = a > b ? a : b, lo = a > b ? b : a;
                e[r] = desc ? hi : lo;
                e[r | dr] = desc ? lo : hi;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int i = r * 64 + lane;
            const bool desc = (kblk == 0) || ((i & kblk) == 0);
            const bool lower = (lane & j) == 0;  // this lane holds the lower index of the pair
            u64 a = e[r];
            u64 b = bh_shfl64(a, lane ^ j);
            u64 hi = a > b ? a : b, lo = a > b ? b : a;
            e[r] = (lower == desc) ? hi : lo;
        }
    }
}

// Full sort, descending by key (element 0 = best).
template <int EPL>
__device__ __forceinline__ void bh_wave_sort_desc(u64 (&e)[EPL], int lane) {
    constexpr int N = 64 * EPL;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            // the last pass (k == N) is a pure descending merge
            bh_bitonic_stage<EPL>(e, lane, j, (k == N) ? 0 : k);
        }
    }
}

// acc, b: both sorted descending (64*EPL each).  acc <- the best 64*EPL of the union, sorted.
template <int EPL>
__device__ __forceinline__ void bh_wave_merge_top(u64 (&acc)[EPL], const u64 (&b)[EPL], int lane) {
    constexpr int N = 64 * EPL;
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        // element i of acc against element N-1-i of b  ->  bitonic sequence of the top N
        u64 x = bh_shfl64(b[EPL - 1 - r], 63 - lane);
        acc[r] = acc[r] > x ? acc[r] : x;
    }
#pragma unroll
    for (int j = N >> 1; j > 0; j >>= 1) bh_bitonic_stage<EPL>(acc, lane, j, 0);
}
