// ubench_mfma.hip — issue-rate microbenchmark behind the scan kernel's design questions (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mfma profiles/ubench_mfma.hip && /tmp/ubench_mfma
// For each (waves per SIMD, independent accumulator chains per wave, MFMA shape) it reports shader cycles per MFMA per
// SIMD (s_memtime) and the effective shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz) on a full chip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int CH, bool BIG>
__global__ void __launch_bounds__(512) mfma_chain(unsigned long long* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.001f + i);
        b[i] = (_Float16)(threadIdx.x * 0.002f - i);
    }
    floatx16 acc[CH];
    floatx4 acc4[CH];
    for (int c = 0; c < CH; ++c) {
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        for (int v = 0; v < 4; ++v) acc4[c][v] = 0.f;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (BIG)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
                else
                    acc4[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[c], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) {
        for (int v = 0; v < 16; ++v) s += acc[c][v];
        for (int v = 0; v < 4; ++v) s += acc4[c][v];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
        out[blockIdx.x * 3 + 2] = (unsigned long long)(s != 12345.f);
    }
}

template <int CH, bool BIG>
void run(const char* name, int threads, int blocks, int iters, unsigned long long* dbuf) {
    std::vector<unsigned long long> h(blocks * 3);
    hipLaunchKernelGGL((mfma_chain<CH, BIG>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_chain<CH, BIG>), dim3(blocks), dim3(threads), 0, 0, dbuf, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int b = 0; b < blocks; ++b) {
        cyc += (double)h[b * 3];
        rt += (double)h[b * 3 + 1];
    }
    cyc /= blocks;
    rt /= blocks;
    const double waves_per_simd = threads / 256.0;
    const double n_per_wave = (double)iters * 8 * CH;
    const double flop = (BIG ? 32768.0 : 16384.0) * n_per_wave * (threads / 64) * blocks;
    printf("%-44s blocks %4d  cycles/MFMA/SIMD %6.2f  memtime/realtime %7.3f  kernel %.3f ms  %.0f TFLOP/s\n", name, blocks,
           cyc / (n_per_wave * waves_per_simd), cyc / rt, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned long long* dbuf;
    hipMalloc(&dbuf, 4096 * 3 * 8);
    const int it = 20000;
    for (int blocks : {1, 256}) {
        run<1, true>("32x32x16 1 wave/SIMD, 1 chain", 256, blocks, it, dbuf);
        run<2, true>("32x32x16 1 wave/SIMD, 2 chains", 256, blocks, it / 2, dbuf);
        run<1, true>("32x32x16 2 waves/SIMD, 1 chain each", 512, blocks, it, dbuf);
        run<2, true>("32x32x16 2 waves/SIMD, 2 chains each", 512, blocks, it / 2, dbuf);
        run<1, false>("16x16x32 1 wave/SIMD, 1 chain", 256, blocks, it, dbuf);
        run<2, false>("16x16x32 1 wave/SIMD, 2 chains", 256, blocks, it / 2, dbuf);
        run<4, false>("16x16x32 1 wave/SIMD, 4 chains", 256, blocks, it / 4, dbuf);
        run<1, false>("16x16x32 2 waves/SIMD, 1 chain each", 512, blocks, it, dbuf);
        run<2, false>("16x16x32 2 waves/SIMD, 2 chains each", 512, blocks, it / 2, dbuf);
        run<4, false>("16x16x32 2 waves/SIMD, 4 chains each", 512, blocks, it / 4, dbuf);
    }
    return 0;
}
