// merge_rescore.hip — per-query merge of the scan's per-workgroup candidate lists, canonical
// fp64 re-scoring of the merged best KP, final sort and cut to k.
//
// Replaces the reference's host merge `torch.cat` -> `.cpu().float()` -> `torch.topk` ->
// `torch.gather` (modules/retrieve.py:169-177), which orders ties arbitrarily.
//
// One 256-thread workgroup per query:
//   1. each of the 4 waves folds every 4th workgroup list (sorted, KP keys) into a running
//      best-KP with a bitonic merge held in registers: the list heads are loaded 64 at a time (one
//      per lane), the lists whose head can still enter are fetched four at a time, and the waves
//      share their running KP-th best through LDS (a scan workgroup fills the whole register file
//      of its CU, so this kernel cannot run beside the next pass's scan: its duration is per-pass
//      fixed cost, and the dependent-load chain of a list-by-list fold was most of it);
//   2. the 4 running lists are combined through LDS by wave 0;
//   3. thread i re-scores candidate i: canonical score = fp32(sum_{j=0..d-1} q[j]*x[j]) with
//      the sum taken sequentially in fp64 (products of fp16 values are exact in fp64, so the
//      result does not depend on FMA contraction and is reproducible bit-for-bit by the
//      oracle's plain C loop);
//   4. wave 0 sorts the KP canonical keys (score desc, row asc) and writes the first k.
// Roofline: latency / L2 (KP x d x 2 bytes gathered per query); negligible next to the scan.
#include "bh_device.h"
#include "bh_kernels.h"

template <int KP>
__global__ void __launch_bounds__(256) bh_merge_rescore_kernel(BhMergeArgs a) {
    constexpr int EPL = KP / 64;
    __shared__ u64 lds_keys[4 * KP];
    __shared__ float lds_qn[4];
    __shared__ u64 lds_worst;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;  // query of the launch: query q % bq of the launch's pass q / bq (a launch merges one pass, or
                               // — pass_stride > 0 — a group of passes whose list sets lie pass_stride keys apart)
    const size_t list_stride = (size_t)a.bq * KP;
    const int pass = a.pass_stride > 0 ? q / a.bq : 0;
    const u64* base = a.partial + (size_t)pass * (size_t)a.pass_stride + (size_t)(q - pass * a.bq) * KP;

    // ---- 1. fold lists wave, wave+4, ...: the heads of 64 lists sit one per lane; the lists whose head can still enter the
    // running best-KP are fetched PF at a time (their latencies overlap) and merged; the running KP-th best of every wave
    // is a lower bound of the query's KP-th best, so the four waves share the largest one through LDS
    constexpr int PF = 4;
    u64 acc[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) acc[r] = 0ull;
    if (tid == 0) lds_worst = 0ull;
    __syncthreads();
    for (int c0 = 0; c0 < a.n_lists; c0 += 256) {
        const int g_lane = c0 + 4 * lane + wave;
        const u64 head = g_lane < a.n_lists ? base[(size_t)g_lane * list_stride] : 0ull;
        u64 todo = __ballot(head != 0ull);
        while (todo) {
            u64 worst = bh_shfl64(acc[EPL - 1], 63);
            const u64 shared = *(volatile u64*)&lds_worst;
            worst = worst > shared ? worst : shared;
            u64 mask = __ballot(head > worst) & todo;
            todo = mask;  // (bounds only rise: a list that fails now fails later too)
            if (!mask) break;
            int pick[PF];
            u64 b[PF][EPL];
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                pick[p] = -1;
                if (mask) {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    todo &= ~(1ull << l);
                    pick[p] = l;
                    const u64* lst = base + (size_t)(c0 + 4 * l + wave) * list_stride;
#pragma unroll
                    for (int r = 0; r < EPL; ++r) b[p][r] = lst[r * 64 + lane];
                }
            }
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                if (pick[p] < 0) continue;
                const u64 cur = bh_shfl64(acc[EPL - 1], 63);
                if (bh_shfl64(b[p][0], 0) <= cur) continue;
                bh_wave_merge_top<EPL>(acc, b[p], lane);
                if (lane == 63 && acc[EPL - 1] != 0ull) atomicMax((unsigned long long*)&lds_worst, (unsigned long long)acc[EPL - 1]);
            }
        }
    }
    // ---- 2. combine the 4 waves
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
    // (certificate) |q|^2 in fp32, slightly over-estimated below: every thread takes a strided part of the query
    {
        const _Float16* qv = a.qtile + (size_t)q * a.dim_padded;
        float s2 = 0.f;
        for (int j = tid; j < a.dim_padded; j += 256) {
            const float v = (float)qv[j];
            s2 += v * v;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        if (lane == 0) lds_qn[wave] = s2;
    }
    const u64 approx_kp_key = lds_keys[KP - 1];  // KP-th best by MFMA score (0: the lists were not full, nothing was dropped)
    // ---- 3. canonical re-scoring, one candidate per thread
    if (tid < KP) {
        const u64 key = lds_keys[tid];
        u64 out = 0ull;
        const unsigned row = bh_key_row(key);
        if (key != 0ull && (long long)row < a.n_rows) {
            const half8* x = reinterpret_cast<const half8*>(a.corpus + (size_t)row * a.dim_padded);
            const half8* qv = reinterpret_cast<const half8*>(a.qtile + (size_t)q * a.dim_padded);
            double s = 0.0;
            const int n8 = a.dim_padded >> 3;
            for (int j = 0; j < n8; j += 4) {  // (dim_padded is a multiple of 32) four 16-byte loads in flight per candidate
                half8 xv[4], qq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = x[j + u];
                    qq[u] = qv[j + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) s = __builtin_fma((double)qq[u][e], (double)xv[u][e], s);
            }
            out = bh_make_key((float)s, row);
        }
        lds_keys[KP + tid] = out;
    }
    __syncthreads();
    // ---- 4. final sort + output
    if (wave == 0) {
        u64 e[EPL];
#pragma unroll
        for (int r = 0; r < EPL; ++r) e[r] = lds_keys[KP + r * 64 + lane];
        bh_wave_sort_desc<EPL>(e, lane);
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int i = r * 64 + lane;
            if (i < a.k) {
                const bool valid = e[r] != 0ull;
                a.out_scores[(size_t)q * a.k + i] = valid ? bh_key_score(e[r]) : -__builtin_inff();
                a.out_ids[(size_t)q * a.k + i] = valid ? a.id_offset + (long long)bh_key_row(e[r]) : -1ll;
            }
        }
        if (a.uncert != nullptr) {
            // k-th canonical key (element k - 1 of the sorted list)
            u64 kth = 0ull;
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const u64 v = bh_shfl64(e[r], (a.k - 1) & 63);
                if (r == ((a.k - 1) >> 6)) kth = v;
            }
            if (lane == 0) {
                bool certified = true;
                if (approx_kp_key != 0ull && a.err_coef > 0.f) {
                    const float qn = sqrtf(lds_qn[0] + lds_qn[1] + lds_qn[2] + lds_qn[3]) * 1.0001f;
                    // everything the scan dropped has an MFMA score <= the KP-th kept one (exclusive thresholds drop ties too)
                    const float limit = bh_key_score(approx_kp_key) + a.err_coef * qn;
                    certified = kth != 0ull && bh_key_score(kth) > limit;
                }
                a.uncert[q] = certified ? 0u : 1u;
                a.kth_key[q] = kth;
                // (host-mapped counter: the host learns "nothing to re-do" from the search's one synchronisation)
                if (!certified && a.n_uncert != nullptr) __hip_atomic_fetch_add(a.n_uncert, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

hipError_t bh_launch_merge_rescore(const BhMergeArgs& a, int kp, int nq_tile, hipStream_t stream) {
    if (nq_tile <= 0) return hipSuccess;
    switch (kp) {
        case 64: hipLaunchKernelGGL(bh_merge_rescore_kernel<64>, dim3(nq_tile), dim3(256), 0, stream, a); break;
        case 128: hipLaunchKernelGGL(bh_merge_rescore_kernel<128>, dim3(nq_tile), dim3(256), 0, stream, a); break;
        case 256: hipLaunchKernelGGL(bh_merge_rescore_kernel<256>, dim3(nq_tile), dim3(256), 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
