// csr_head.hip — builds the stream the MFMA sparse scan reads (csr_mfma.hip, "corpus-side head block") from the CSR corpus.
//
// The BH_CSR_HEAD_TERMS (64) terms with the largest document frequency of the index are stored per 32-document group as a
// dense fp16 tile, the other ("tail") entries stay a CSR stream; group g's part of the stream is
//     [ tile: BH_CSR_HEAD_DWORDS dwords ][ tail entries of documents 32 g .. 32 g + 31, in document order ]
// at dword offset  row_ptr2[32 g] + BH_CSR_HEAD_DWORDS * g,  row_ptr2 = row pointers of the tail entries alone.
// Tile layout = the order in which the scan's register buffer receives it: dword 256 s + 4 lane + j holds halves 2 j, 2 j + 1
// of the 8-half MFMA A fragment of k-step s for lane = doc + 32 h, i.e. head slots 16 s + 8 h + {2 j, 2 j + 1} of that document.
// The original CSR stays resident: the canonical re-score (csr_topk.hip) and the first-generation scan read it.
// One-time cost at bh_sparse_finalize (a pass over the corpus); reference: the reference keeps sparse COO chunks on the host
// and re-uploads them per query chunk (modules/retrieve.py:84-90,153) — there is no index structure to mirror.
#include "bh_device.h"
#include "bh_kernels.h"

// tail entries per document
__global__ void __launch_bounds__(256) bh_csr_tail_count_kernel(const unsigned* entries, const long long* row_ptr, long long n_rows,
                                                                const unsigned char* head_slot, unsigned* tail_cnt) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    unsigned n = 0;
    for (long long i = row_ptr[r], e = row_ptr[r + 1]; i < e; ++i) n += head_slot[entries[i] & 0xffffu] == 0xffu ? 1u : 0u;
    tail_cnt[r] = n;
}

// one thread per document: its tail entries in order, its 64 head weights into the group's tile
__global__ void __launch_bounds__(256) bh_csr_split_kernel(const unsigned* entries, const long long* row_ptr, const long long* row_ptr2,
                                                           long long n_rows, const unsigned char* head_slot, unsigned* out) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_pad = (n_rows + 31) / 32 * 32;
    if (r >= n_pad) return;
    const long long grp = r >> 5;
    const int dd = (int)(r & 31);
    const long long g0 = grp * 32;
    unsigned* tile = out + row_ptr2[g0 < n_rows ? g0 : n_rows] + (long long)BH_CSR_HEAD_DWORDS * grp;
    unsigned short hw[BH_CSR_HEAD_TERMS];
#pragma unroll
    for (int i = 0; i < BH_CSR_HEAD_TERMS; ++i) hw[i] = 0;
    if (r < n_rows) {
        unsigned* tail = out + row_ptr2[r] + (long long)BH_CSR_HEAD_DWORDS * (grp + 1);
        for (long long i = row_ptr[r], e = row_ptr[r + 1]; i < e; ++i) {
            const unsigned ent = entries[i];
            const unsigned hs = head_slot[ent & 0xffffu];
            if (hs == 0xffu) {
                *tail++ = ent;
            } else {
#pragma unroll
                for (int u = 0; u < BH_CSR_HEAD_TERMS; ++u)  // (static indices: the array stays in registers)
                    if (u == (int)hs) hw[u] = (unsigned short)(ent >> 16);
            }
        }
    }
    // (rows of the last group beyond n_rows write zeros: the tile is complete whatever n_rows is)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hx = 16 * s + 8 * h + 2 * j;
                tile[256 * s + 4 * (dd + 32 * h) + j] = (unsigned)hw[hx] | ((unsigned)hw[hx + 1] << 16);
            }
}

hipError_t bh_launch_csr_tail_count(const unsigned* entries, const long long* row_ptr, long long n_rows, const unsigned char* head_slot,
                                    unsigned* tail_cnt, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(bh_csr_tail_count_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, entries, row_ptr, n_rows,
                       head_slot, tail_cnt);
    return hipGetLastError();
}

hipError_t bh_launch_csr_split(const unsigned* entries, const long long* row_ptr, const long long* row_ptr2, long long n_rows,
                               const unsigned char* head_slot, unsigned* stream_out, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    const long long n_pad = (n_rows + 31) / 32 * 32;
    hipLaunchKernelGGL(bh_csr_split_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, stream, entries, row_ptr, row_ptr2, n_rows,
                       head_slot, stream_out);
    return hipGetLastError();
}
