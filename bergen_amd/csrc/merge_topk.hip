// merge_topk.hip — merge n_lists partial top-k lists per query into one, canonical order
// (score descending, id ascending).  This is the reduction step after the per-shard searches
// (one list per GPU / per rank after the RCCL all-gather) and replaces the reference's host
// merge, modules/retrieve.py:169-177 (torch.cat + torch.topk + torch.gather on CPU).
//
// One workgroup per query.  M = n_lists*k entries (<= 4096) are staged in LDS and ranked by
// counting: rank(i) = #{j : entry j precedes entry i}; entries with rank < k are scattered to
// their output slot.  O(M^2) compares per query, but M is a few hundred to ~1600 here
// (8 shards x k<=200) and the whole step is latency-bound next to the scan.
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
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const long long id = s_id[i];
        if (id < 0) continue;
        const float sc = s_sc[i];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
            const long long idj = s_id[j];
            const float scj = s_sc[j];
            const bool before = (idj >= 0) && ((scj > sc) || (scj == sc && (idj < id || (idj == id && j < i))));
            rank += before ? 1 : 0;
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
