// merge_topk.hip — merge n_lists partial top-k lists per query into one, canonical order
// (score descending, id ascending).  This is the reduction step after the per-shard searches
// (one list per GPU / per rank after the RCCL all-gather) and replaces the reference's host
// merge, modules/retrieve.py:169-177 (torch.cat + torch.topk + torch.gather on CPU).
//
// One workgroup per query.  M = n_lists*k entries (<= 4096) are staged in LDS and RANKED: rank(i) = #{j : entry j precedes
// entry i}; entries with rank < k are scattered to their output slot.
//   * The lists a search produces are SORTED in the canonical order (valid entries first, padding — id < 0 — at the tail): the
//     entries of list l that precede entry (l, j) are then exactly its first j, and those of another list l' a PREFIX of it, found
//     by binary search: (n_lists - 1) * log2(k) steps per entry.  (Round 6: the one-GPU proxy of the whole 8-GPU search showed the
//     all-pairs count below taking 2.45 ms for 8 lists of 200 on rank 0's critical path — 28 % of the step it follows.)
//   * The entry point accepts ANY lists (the reference's host merge, torch.cat + torch.topk, does): every workgroup first checks
//     that its query's lists are sorted, and sorts them in LDS (rank inside the own list: k compares per entry) when one is not.
#include "bh_device.h"
#include "bh_kernels.h"

#define BH_MERGE_MAX 4096

__global__ void __launch_bounds__(256) bh_merge_lists_kernel(const float* __restrict__ scores,
                                                              const long long* __restrict__ ids, int n_lists,
                                                              int nq, int k, float* __restrict__ out_scores,
                                                              long long* __restrict__ out_ids) {
    __shared__ float s_sc[BH_MERGE_MAX];
    __shared__ long long s_id[BH_MERGE_MAX];
    const int q = blockIdx.x;
    const int M = n_lists * k;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int l = i / k, j = i - l * k;
        const size_t src = ((size_t)l * nq + q) * k + j;
        s_sc[i] = scores[src];
        s_id[i] = ids[src];
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        out_scores[(size_t)q * k + i] = -__builtin_inff();
        out_ids[(size_t)q * k + i] = -1ll;
    }
    __syncthreads();
    // entry a precedes entry b (canonical order: score descending, id ascending, then position; padding never precedes)
    auto precedes = [&](int a, float sc, long long id, int b) {
        const long long ida = s_id[a];
        const float sca = s_sc[a];
        return (ida >= 0) && ((sca > sc) || (sca == sc && (ida < id || (ida == id && a < b))));
    };
    // sorted?  (every entry against its successor in the same list: a valid entry must precede the next valid one, padding must
    // not be followed by a valid entry)
    __shared__ int s_unsorted;
    if (threadIdx.x == 0) s_unsorted = 0;
    __syncthreads();
    bool bad = false;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int j = i % k;
        if (j + 1 < k) {
            const long long id1 = s_id[i + 1];
            if (id1 >= 0 && !precedes(i, s_sc[i + 1], id1, i + 1)) bad = true;
        }
    }
    if (bad) s_unsorted = 1;
    __syncthreads();
    if (s_unsorted != 0) {
        // Arbitrary input (an external caller's lists): sort every list in place first — an entry's position inside its own list by counting
        // that list's entries that precede it (k compares per entry instead of the M an all-pairs rank over all lists takes: 8 x 200 entries
        // 0.68 -> ~0.1 ms for the one workgroup that needs it), padding behind the valid entries in index order.  Every thread holds its
        // entries in registers across the barrier, so the permutation needs no second buffer.
        constexpr int PER = (BH_MERGE_MAX + 255) / 256;
        float rs[PER];
        long long ri[PER];
        int rp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = threadIdx.x + u * 256;
            rp[u] = -1;
            if (i < M) {
                const int l = i / k, base = l * k;
                rs[u] = s_sc[i];
                ri[u] = s_id[i];
                int pos = 0;
                if (ri[u] >= 0) {
                    for (int j = base; j < base + k; ++j) pos += precedes(j, rs[u], ri[u], i) ? 1 : 0;
                } else {
                    for (int j = base; j < base + k; ++j) pos += (s_id[j] >= 0 || j < i) ? 1 : 0;  // every valid entry, and the padding before it
                }
                rp[u] = base + pos;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (rp[u] >= 0) {
                s_sc[rp[u]] = rs[u];
                s_id[rp[u]] = ri[u];
            }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const long long id = s_id[i];
        if (id < 0) continue;
        const float sc = s_sc[i];
        const int l = i / k;
        int rank = i - l * k;  // its own list's entries before it
        for (int lo = 0; lo < n_lists && rank < k; ++lo) {  // (rank >= k: not in the result, whatever the other lists add)
            if (lo == l) continue;
            // the prefix of list lo that precedes entry i: first position whose entry does not.  Only prefixes shorter than
            // k - rank matter (a longer one pushes the entry out of the result all the same): the search is bounded by it
            int a = 0, b = k - rank;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (precedes(lo * k + mid, sc, id, i))
                    a = mid + 1;
                else
                    b = mid;
            }
            rank += a;
        }
        if (rank < k) {
            out_scores[(size_t)q * k + rank] = sc;
            out_ids[(size_t)q * k + rank] = id;
        }
    }
}

hipError_t bh_launch_merge_lists(const float* scores, const long long* ids, int n_lists, int nq, int k,
                                 float* out_scores, long long* out_ids, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if ((long long)n_lists * k > BH_MERGE_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bh_merge_lists_kernel, dim3(nq), dim3(256), 0, stream, scores, ids, n_lists, nq, k,
                       out_scores, out_ids);
    return hipGetLastError();
}
