// ubench_mfma_ops.hip — MFMA issue rate with DISTINCT operand registers per instruction (what a real kernel has), by
// shape, by register file of the accumulator (V/A) and of the B operand (V/A); A operand always in VGPRs (it comes from
// LDS).  Kernel time on a full chip -> cycles per MFMA per SIMD at the measured clock, and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/bin/ubench_mfma_ops profiles/ubench_mfma_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// BIG: 0 = 16x16x32 (four chains), 1 = 32x32x16 (two chains); ACCA / BA: accumulator / B operand in AGPRs; NA = distinct A
// registers in rotation (1 = the same A every time), NBR = distinct B registers
template <int THREADS, int BIG, int ACCA, int BA, int NA, int NBR>
__global__ void __launch_bounds__(THREADS) k(unsigned long long* out, int iters) {
    half8 a[4], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            b[j][i] = (_Float16)(threadIdx.x * 0.002f - i + j);
            if (j < 4) a[j][i] = (_Float16)(threadIdx.x * 0.001f + i - j);
        }
    floatx4 acc4[4];
    floatx16 acc16[2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc4[c][v] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc16[c][v] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (BA) asm volatile("" : "+a"(b[j]));
        else asm volatile("" : "+v"(b[j]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(a[j]));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            const int j = m >> 1, c = m & 1;
            const half8& av = a[j % NA];
            const half8& bv = b[(j + c * 4) % NBR];
            if (BIG == 0) {
                floatx4& x = acc4[(j & 1) * 2 + c];
                if (ACCA && BA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(x) : "v"(av), "a"(bv));
                else if (ACCA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(x) : "v"(av), "v"(bv));
                else if (BA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(av), "a"(bv));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(av), "v"(bv));
            } else {
                floatx16& x = acc16[c];
                if (ACCA && BA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(x) : "v"(av), "a"(bv));
                else if (ACCA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(x) : "v"(av), "v"(bv));
                else if (BA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x) : "v"(av), "a"(bv));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x) : "v"(av), "v"(bv));
            }
        }
        asm volatile("s_barrier");
    }
    float s = 0.f;
    if (BIG == 0) {
        if (ACCA) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc4[0]), "+a"(acc4[1]), "+a"(acc4[2]), "+a"(acc4[3]));
        else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]));
        for (int c = 0; c < 4; ++c)
            for (int v = 0; v < 4; ++v) s += acc4[c][v];
    } else {
        if (ACCA) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(acc16[0]), "+a"(acc16[1]));
        else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc16[0]), "+v"(acc16[1]));
        for (int c = 0; c < 2; ++c)
            for (int v = 0; v < 16; ++v) s += acc16[c][v];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)(s != 1.25f); }
}

template <int THREADS, int BIG, int ACCA, int BA, int NA, int NBR>
void run(unsigned long long* d) {
    const int blocks = 256, iters = 3000;
    std::vector<unsigned long long> h(blocks * 3);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<THREADS, BIG, ACCA, BA, NA, NBR>), dim3(blocks), dim3(THREADS), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<THREADS, BIG, ACCA, BA, NA, NBR>), dim3(blocks), dim3(THREADS), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)h[b * 3]; rt += (double)h[b * 3 + 1]; }
    const double mhz = 100.0 * cyc / rt;
    const double n_simd = (double)iters * 48 * (THREADS / 256);
    const double flop = (BIG ? 32768.0 : 16384.0) * n_simd * 4 * blocks;
    printf("%s  %d wave/SIMD  acc %c  B %c  A regs %d  B regs %d : %5.2f cycles/MFMA/SIMD (ideal %d)  %4.0f MHz  %4.0f TFLOP/s\n",
           BIG ? "32x32x16" : "16x16x32", THREADS / 256, ACCA ? 'A' : 'V', BA ? 'A' : 'V', NA, NBR, ms * 1e-3 * mhz * 1e6 / n_simd, BIG ? 32 : 16, mhz,
           flop / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 4096 * 24);
    run<256, 0, 0, 0, 1, 1>(d);
    run<256, 0, 0, 0, 4, 8>(d);
    run<256, 0, 1, 0, 4, 8>(d);
    run<256, 0, 0, 1, 4, 8>(d);
    run<256, 0, 1, 1, 4, 8>(d);
    run<256, 0, 1, 1, 1, 8>(d);
    run<256, 0, 1, 1, 4, 1>(d);
    run<512, 0, 1, 1, 4, 8>(d);
    run<512, 0, 0, 1, 4, 8>(d);
    run<256, 1, 0, 0, 1, 1>(d);
    run<256, 1, 0, 0, 4, 8>(d);
    run<256, 1, 1, 0, 4, 8>(d);
    run<256, 1, 0, 1, 4, 8>(d);
    run<256, 1, 1, 1, 4, 8>(d);
    run<512, 1, 1, 1, 4, 8>(d);
    run<512, 1, 0, 1, 4, 8>(d);
    return 0;
}
