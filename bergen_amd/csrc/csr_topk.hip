// csr_topk.hip — exact sparse retrieval (SPLADE): sparse-query x sparse-document scores + running top-k over a
// resident CSR corpus, and the merge / canonical re-score of the candidates.
//
// Replaces the reference's  Splade.similarity_fn  = torch.sparse.mm(q.to_sparse(), d_chunk.t()).to_dense()
// (models/retrievers/splade.py:55-56) + torch.topk per chunk (modules/retrieve.py:157) + host merge
// (modules/retrieve.py:169-177): the dense [Bq, n] score matrix is never materialised.
//
// Roofline: HBM.  One launch streams the whole CSR corpus once for a tile of up to 64 queries:
// algorithmic bytes = nnz*4 (one 4-byte entry = uint16 term id | fp16 weight) + (N+1)*8 (row pointers).
//
// Work decomposition
//   * persistent grid, one 1024-thread workgroup (16 waves) per CU; wave g walks documents g, g+W, g+2W, ...
//     (neighbouring documents — contiguous entries — are read by neighbouring waves at about the same time).
//   * LANE = QUERY.  A wave reads 64 consecutive entries of its document with one coalesced load (lane i = entry
//     i), looks every term up in the tile's term set — a bitmap + rank table in LDS: two LDS gathers and a popcount
//     give "is any query of the tile using this term, and in which slot" — and ballots the hits.  For each hit
//     entry the (slot, weight) pair is broadcast with v_readlane and lane q adds  weight * W[slot][q]  from the
//     tile's dense slot x query weight table in LDS (conflict-free 128-byte rows).  Misses cost nothing per lane.
//     Hits are processed four at a time so that their LDS reads overlap.  Sums run in entry (= term id) order in
//     fp32: deterministic, no atomics.
//   * end of document: lane q holds the document's score for query q -> the same threshold / candidate-buffer /
//     wave-bitonic-compaction scheme as the dense scan (per-wave buffers in global memory, count and threshold in
//     registers).  A wave's KP-th best score is a valid lower bound of the final KP-th best, so waves share
//     thresholds through LDS (workgroup) and one global table (chip) with plain atomicMax — a pure filter hint.
//   * end of launch: the 16 waves' sorted lists are merged per query into one list per workgroup.
// Exactness: candidates are ranked by the fp32 score; bh_csr_merge_rescore_kernel re-scores the merged best KP in
// fp64 in term order (the canonical score the oracle's plain C loop reproduces bit for bit), sorts by
// (score desc, row asc) and cuts to k.
#include "bh_device.h"
#include "bh_kernels.h"

namespace {

template <int EPL>
__device__ __forceinline__ void load_keys(u64 (&e)[EPL], const u64* buf, unsigned n, int lane) {
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const unsigned idx = r * 64 + lane;
        e[r] = idx < n ? buf[idx] : 0ull;
    }
}

}  // namespace

template <int KP>
__global__ void __launch_bounds__(1024) bh_csr_scan_topk_kernel(BhCsrScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CAP = 2 * KP;
    constexpr int EPLC = CAP / 64, EPLK = KP / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NWV = 16;
    const long long gw = (long long)blockIdx.x * NWV + wave, TW = (long long)gridDim.x * NWV;

    // ---- LDS: bitmap[n_words] u32 | prefix[n_words] u16 | W[(n_slots + 1)][64] fp16 (last row = zeros) | thr[64] u32
    unsigned* bitmap = reinterpret_cast<unsigned*>(smem);
    unsigned short* prefix = reinterpret_cast<unsigned short*>(smem + a.off_prefix);
    const _Float16* W = reinterpret_cast<const _Float16*>(smem + a.off_w);
    unsigned* thr_lds = reinterpret_cast<unsigned*>(smem + a.off_thr);
    for (int i = tid; i < a.n_words; i += 1024) {
        bitmap[i] = a.bitmap[i];
        prefix[i] = a.prefix[i];
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.W);
        uint4* dst = reinterpret_cast<uint4*>(smem + a.off_w);
        const int n16 = (a.n_slots + 1) * 8;  // 128-byte rows
        for (int i = tid; i < n16; i += 1024) dst[i] = src[i];
    }
    if (tid < 64) thr_lds[tid] = BH_ORD_NEG_INF;
    __syncthreads();
    const int zero_slot = a.n_slots;

    float thr = -__builtin_inff();  // candidate iff score > thr (lane = query)
    unsigned cnt = 0;               // entries in this (wave, query) candidate buffer
    u64* cand_w = a.cand + (size_t)gw * 64 * CAP;
    u64* mybuf = cand_w + (size_t)lane * CAP;

    auto compact = [&](int qq) {  // sort query qq's buffer with the whole wave, keep the best KP, raise its threshold
        const unsigned n = __builtin_amdgcn_readlane(cnt, qq);
        u64* buf = cand_w + (size_t)qq * CAP;
        u64 e[EPLC];
        load_keys<EPLC>(e, buf, n, lane);
        bh_wave_sort_desc<EPLC>(e, lane);
#pragma unroll
        for (int r = 0; r < EPLK; ++r) buf[r * 64 + lane] = e[r];
        const u64 kth = bh_shfl64(e[EPLK - 1], 63);
        if (lane == qq) {
            cnt = n < (unsigned)KP ? n : (unsigned)KP;
            if (kth != 0ull) {
                // documents arrive in ascending order inside a wave: a later document that merely ties the KP-th
                // best loses on row index, so the exclusive compare against the wave's own bound is exact
                thr = fmaxf(thr, bh_key_score(kth));
                atomicMax(&thr_lds[qq], bh_ordf(bh_key_score(kth)));
            }
        }
    };

    // Software pipeline over documents: while document d is processed, the first PF chunks (64 entries each) of the
    // wave's NEXT document are already in flight and the row pointers of the one after that are being fetched, so
    // the dependent chain row pointer -> entries -> LDS lookups never sits exposed.
    constexpr int PF = 2;
    long long n_seen = 0;
    long long d = gw;
    long long e0 = 0, e1 = 0, ne0 = 0, ne1 = 0;
    if (d < a.n_rows) {
        e0 = a.row_ptr[d];
        e1 = a.row_ptr[d + 1];
    }
    if (d + TW < a.n_rows) {
        ne0 = a.row_ptr[d + TW];
        ne1 = a.row_ptr[d + TW + 1];
    }
    unsigned cur[PF], nxt[PF];
#pragma unroll
    for (int c = 0; c < PF; ++c) cur[c] = (e0 + c * 64 + lane < e1) ? a.entries[e0 + c * 64 + lane] : 0u;
    for (; d < a.n_rows; d += TW, ++n_seen) {
        // prefetch: entries of document d + TW, row pointers of document d + 2 TW
        long long nne0 = 0, nne1 = 0;
        if (d + 2 * TW < a.n_rows) {
            nne0 = a.row_ptr[d + 2 * TW];
            nne1 = a.row_ptr[d + 2 * TW + 1];
        }
#pragma unroll
        for (int c = 0; c < PF; ++c) nxt[c] = (ne0 + c * 64 + lane < ne1) ? a.entries[ne0 + c * 64 + lane] : 0u;

        float acc = 0.f;
        auto chunk = [&](unsigned ent, bool valid) {
            const unsigned term = ent & 0xffffu;
            const float val = (float)__builtin_bit_cast(_Float16, (unsigned short)(ent >> 16));
            const unsigned word = bitmap[term >> 5];
            const unsigned bit = 1u << (term & 31);
            const bool hit = valid && (word & bit);
            const int slot = (int)prefix[term >> 5] + __builtin_popcount(word & (bit - 1u));
            u64 mask = __builtin_amdgcn_ballot_w64(hit);
            while (mask) {  // four hits per round: their LDS reads are issued together
                int sl[4];
                float vv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (mask) {
                        const int i = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        sl[t] = __builtin_amdgcn_readlane(slot, i);
                        vv[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), i));
                    } else {
                        sl[t] = zero_slot;
                        vv[t] = 0.f;
                    }
                }
                float w[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) w[t] = (float)W[sl[t] * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = fmaf(vv[t], w[t], acc);
            }
        };
#pragma unroll
        for (int c = 0; c < PF; ++c)
            if (e0 + c * 64 < e1) chunk(cur[c], e0 + c * 64 + lane < e1);
        for (long long eb = e0 + PF * 64; eb < e1; eb += 64) {  // unusually long documents: the rest on demand
            const bool valid = eb + lane < e1;
            chunk(valid ? a.entries[eb + lane] : 0u, valid);
        }
        // ---- threshold filter (lane = query; queries beyond the tile hold score 0 and are ignored downstream)
        if (acc > thr) {
            mybuf[cnt] = bh_make_key(acc, (unsigned)d);
            ++cnt;
        }
        u64 need = __builtin_amdgcn_ballot_w64(cnt >= (unsigned)CAP);
        while (need) {
            const int qq = __builtin_ctzll(need);
            need &= need - 1;
            compact(qq);
        }
        // ---- pick up bounds published by the other waves (workgroup: LDS; chip: global), now and then
        if ((n_seen & 31) == 31) {
            unsigned b = thr_lds[lane];
            if ((n_seen & 255) == 255) {
                const unsigned gl = __hip_atomic_load(a.gthr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b > gl) __hip_atomic_fetch_max(a.gthr + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = b > gl ? b : gl;
            }
            // a document that TIES another wave's bound may still win on row index: inclusive compare,
            // i.e. exclusive against the next lower float
            if (b > BH_ORD_NEG_INF) thr = fmaxf(thr, bh_unordf(b - 1u));
        }
        // rotate the pipeline
        e0 = ne0;
        e1 = ne1;
        ne0 = nne0;
        ne1 = nne1;
#pragma unroll
        for (int c = 0; c < PF; ++c) cur[c] = nxt[c];
    }

    // ---- final: every wave sorts its buffers (best KP first), then the workgroup merges its 16 lists per query
    for (int qq = 0; qq < 64; ++qq) {
        const unsigned n = __builtin_amdgcn_readlane(cnt, qq);
        u64* buf = cand_w + (size_t)qq * CAP;
        u64 e[EPLC];
        load_keys<EPLC>(e, buf, n, lane);
        bh_wave_sort_desc<EPLC>(e, lane);
#pragma unroll
        for (int r = 0; r < EPLK; ++r) buf[r * 64 + lane] = e[r];
    }
    __syncthreads();  // (workgroup-scope release/acquire: the lists were written by waves of this CU)
    for (int qq = wave; qq < 64; qq += NWV) {
        u64 acc[EPLK];
#pragma unroll
        for (int r = 0; r < EPLK; ++r) acc[r] = 0ull;
        for (int w2 = 0; w2 < NWV; ++w2) {  // fold the 16 sorted lists one at a time (bitonic merge, best KP kept)
            const u64* lst = a.cand + ((size_t)((long long)blockIdx.x * NWV + w2) * 64 + qq) * CAP;
            u64 bb[EPLK];
#pragma unroll
            for (int r = 0; r < EPLK; ++r) bb[r] = lst[r * 64 + lane];
            bh_wave_merge_top<EPLK>(acc, bb, lane);
        }
        u64* out = a.partial + ((size_t)blockIdx.x * 64 + qq) * KP;
#pragma unroll
        for (int r = 0; r < EPLK; ++r) out[r * 64 + lane] = acc[r];
    }
}

// One 256-thread workgroup per query of the tile: fold the per-workgroup lists, canonical fp64 re-score of the
// merged best KP (sum over the document's entries in term order of  q[term] * weight), final sort, cut to k.
template <int KP>
__global__ void __launch_bounds__(256) bh_csr_merge_rescore_kernel(BhCsrMergeArgs a) {
    constexpr int EPL = KP / 64;
    __shared__ u64 lds_keys[4 * KP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;
    const size_t list_stride = (size_t)64 * KP;
    const u64* base = a.partial + (size_t)q * KP;
    u64 acc[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) acc[r] = 0ull;
    for (int g = wave; g < a.n_lists; g += 4) {
        const u64* lst = base + (size_t)g * list_stride;
        const u64 best = lst[0];
        const u64 worst = bh_shfl64(acc[EPL - 1], 63);
        if (best <= worst) continue;
        u64 b[EPL];
#pragma unroll
        for (int r = 0; r < EPL; ++r) b[r] = lst[r * 64 + lane];
        bh_wave_merge_top<EPL>(acc, b, lane);
    }
#pragma unroll
    for (int r = 0; r < EPL; ++r) lds_keys[wave * KP + r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            u64 b[EPL];
#pragma unroll
            for (int r = 0; r < EPL; ++r) b[r] = lds_keys[w * KP + r * 64 + lane];
            bh_wave_merge_top<EPL>(acc, b, lane);
        }
#pragma unroll
        for (int r = 0; r < EPL; ++r) lds_keys[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (tid < KP) {
        const u64 key = lds_keys[tid];
        u64 out = 0ull;
        const unsigned row = bh_key_row(key);
        if (key != 0ull && (long long)row < a.n_rows) {
            const _Float16* qv = a.q_dense + (size_t)q * a.vocab;
            double s = 0.0;
            for (long long e = a.row_ptr[row]; e < a.row_ptr[row + 1]; ++e) {
                const unsigned ent = a.entries[e];
                const _Float16 w = __builtin_bit_cast(_Float16, (unsigned short)(ent >> 16));
                s = __builtin_fma((double)qv[ent & 0xffffu], (double)w, s);
            }
            out = bh_make_key((float)s, row);
        }
        lds_keys[KP + tid] = out;
    }
    __syncthreads();
    if (wave == 0) {
        u64 e[EPL];
#pragma unroll
        for (int r = 0; r < EPL; ++r) e[r] = lds_keys[KP + r * 64 + lane];
        bh_wave_sort_desc<EPL>(e, lane);
        unsigned n_valid = 0;
#pragma unroll
        for (int r = 0; r < EPL; ++r) n_valid += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(e[r] != 0ull));
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int i = r * 64 + lane;
            if (i < a.k) {
                const bool valid = e[r] != 0ull;
                a.out_scores[(size_t)q * a.k + i] = valid ? bh_key_score(e[r]) : -__builtin_inff();
                a.out_ids[(size_t)q * a.k + i] = valid ? a.id_offset + (long long)bh_key_row(e[r]) : -1ll;
            }
        }
        // Non-negative data (floor_zero): the scan collected positive-score documents only.  A list shorter than k then
        // holds EVERY positive document of the corpus, and the canonical order continues with the zero-score documents
        // by ascending row: the lowest rows that are not in the list.  At most n_valid of the rows [0, k) are in it, so
        // the k - n_valid fill rows are all found below k.
        if (a.floor_zero && n_valid < (unsigned)a.k) {
#pragma unroll
            for (int r = 0; r < EPL; ++r) lds_keys[r * 64 + lane] = e[r];  // (same wave wrote and reads: no barrier needed)
            unsigned filled = n_valid;
            for (int rb = 0; rb < a.k && filled < (unsigned)a.k; rb += 64) {
                const long long row = rb + lane;
                bool absent = row < a.k && row < a.n_rows;
                for (unsigned c = 0; c < n_valid && absent; ++c)
                    if ((long long)bh_key_row(lds_keys[c]) == row) absent = false;
                const u64 am = __builtin_amdgcn_ballot_w64(absent);
                const unsigned pos = filled + __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
                if (absent && pos < (unsigned)a.k) {
                    a.out_scores[(size_t)q * a.k + pos] = 0.f;
                    a.out_ids[(size_t)q * a.k + pos] = a.id_offset + row;
                }
                filled += (unsigned)__builtin_popcountll(am);
            }
        }
    }
}

// dst[pos[i]] = val[i]: builds the dense fp16 query matrix on the device from the queries' non-zeros (the matrix is
// mostly zeros: a memset plus a few thousand scattered halves instead of a 15 MB host-to-device copy)
__global__ void __launch_bounds__(256) bh_scatter_f16_kernel(unsigned short* dst, const unsigned long long* pos,
                                                             const unsigned short* val, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[pos[i]] = val[i];
}

hipError_t bh_launch_scatter_f16(unsigned short* dst, const unsigned long long* pos, const unsigned short* val, int n,
                                 hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_scatter_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, pos, val, n);
    return hipGetLastError();
}

hipError_t bh_launch_csr_scan(const BhCsrScanArgs& a, int kp, int grid, size_t smem, hipStream_t stream) {
    static size_t attr[2] = {0, 0};
    const void* fn = kp == 64 ? reinterpret_cast<const void*>(bh_csr_scan_topk_kernel<64>)
                              : reinterpret_cast<const void*>(bh_csr_scan_topk_kernel<128>);
    size_t& done = attr[kp == 64 ? 0 : 1];
    if (smem > done) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        done = smem;
    }
    if (kp == 64)
        hipLaunchKernelGGL(bh_csr_scan_topk_kernel<64>, dim3(grid), dim3(1024), smem, stream, a);
    else if (kp == 128)
        hipLaunchKernelGGL(bh_csr_scan_topk_kernel<128>, dim3(grid), dim3(1024), smem, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t bh_launch_csr_merge_rescore(const BhCsrMergeArgs& a, int kp, int nq_tile, hipStream_t stream) {
    if (nq_tile <= 0) return hipSuccess;
    if (kp == 64)
        hipLaunchKernelGGL(bh_csr_merge_rescore_kernel<64>, dim3(nq_tile), dim3(256), 0, stream, a);
    else if (kp == 128)
        hipLaunchKernelGGL(bh_csr_merge_rescore_kernel<128>, dim3(nq_tile), dim3(256), 0, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
