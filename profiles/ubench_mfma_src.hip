// ubench_mfma_src.hip — does the source register file of the MFMA B operand matter?  16x16x32 f16, four accumulator chains per
// wave (in the accumulator file, updated in place), two waves per SIMD, 8 distinct B fragments either in AGPRs or in VGPRs.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/bin/ubench_mfma_src profiles/ubench_mfma_src.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: B in VGPRs, 1: B in AGPRs, 2: B in VGPRs and accumulators in VGPRs too, 3: half / half
__global__ void __launch_bounds__(512, 2) k(unsigned long long* out, int iters) {
    half8 a[2], b[8];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
            b[j][i] = (_Float16)(threadIdx.x * 0.002f - i + j);
            a[j & 1][i] = (_Float16)(threadIdx.x * 0.001f + i - j);
        }
    floatx4 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int v = 0; v < 4; ++v) acc[c][v] = 0.f;
    for (int j = 0; j < 8; ++j) {
        if (MODE == 1 || (MODE == 3 && j < 4)) asm volatile("" : "+a"(b[j]));
        else asm volatile("" : "+v"(b[j]));
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ch = (j & 1) * 2 + c;
                if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[ch]) : "v"(a[j & 1]), "v"(b[(j + c * 4) & 7]));
                else if (MODE == 1 || (MODE == 3 && ((j + c * 4) & 7) < 4))
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[ch]) : "v"(a[j & 1]), "a"(b[(j + c * 4) & 7]));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[ch]) : "v"(a[j & 1]), "v"(b[(j + c * 4) & 7]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int v = 0; v < 4; ++v) s += acc[c][v];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)(s != 1.25f); }
}

template <int MODE>
void run(const char* name, int blocks, int iters, unsigned long long* d) {
    std::vector<unsigned long long> h(blocks * 2);
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) cyc += (double)h[b * 2];
    cyc /= blocks;
    printf("%-52s blocks %3d: %.2f cycles per MFMA per SIMD (wave 0's clock; 2 waves x 16 MFMAs per iteration)\n", name, blocks, cyc / (iters * 16.0 * 2));
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 4096 * 16);
    for (int blocks : {1, 256}) {
        run<0>("B in VGPRs, accumulators in AGPRs", blocks, 20000, d);
        run<1>("B in AGPRs, accumulators in AGPRs", blocks, 20000, d);
        run<2>("B in VGPRs, accumulators in VGPRs", blocks, 20000, d);
        run<3>("B half AGPR half VGPR, accumulators in AGPRs", blocks, 20000, d);
    }
    return 0;
}
