// ubench_coresident.hip — does a small-register streaming kernel start on CUs that a persistent, register- and LDS-filling kernel occupies?
// (Round 6: the question behind bh_layernorm_small_kernel.)  K1 ("gemm-like"): 256 workgroups x 512 threads, REGS vector registers per lane
// (launch bounds 512 x 2 waves per SIMD), 160 KiB of LDS, spins for a given number of microseconds.  K2 ("layernorm-like"): one wave per 1.5 KiB
// row, loads two rows and stores one (8-byte accesses), V2 vector registers.  K2 is launched on a second stream 200 us after K1; reported: K2's
// duration alone, and — launched under K1 — whether it finishes before K1 does (co-resident) and how long it takes.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/bin/ubench_coresident profiles/ubench_coresident.hip && profiles/bin/ubench_coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int REGS>
__global__ void __launch_bounds__(512, 2) hog_kernel(float* out, long long spin_ticks, int lds_bytes) {
    extern __shared__ unsigned char smem[];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x + i);
    if (lds_bytes > 0) smem[threadIdx.x] = (unsigned char)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) {  // 100 MHz ticks
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    if (s == 12345.678f) out[blockIdx.x] = s + smem[0];
}

__global__ void __launch_bounds__(256) stream_kernel(const _Float16* a, const _Float16* b, _Float16* c, long long n_rows) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= n_rows) return;
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 v[3], w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const half4*>(a + row * 768 + j * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = *reinterpret_cast<const half4*>(b + row * 768 + j * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<half4*>(c + row * 768 + j * 256 + lane * 4) = v[j] + w[j];
}

int main() {
    const long long rows = 34432;  // one micro-batch of the bench's encoder batch
    _Float16 *a, *b, *c;
    float* out;
    CHECK(hipMalloc(&a, rows * 768 * 2)); CHECK(hipMalloc(&b, rows * 768 * 2)); CHECK(hipMalloc(&c, rows * 768 * 2)); CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(a, 0, rows * 768 * 2)); CHECK(hipMemset(b, 0, rows * 768 * 2));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, g0, g1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&g0)); CHECK(hipEventCreate(&g1));
    const int lds = 160 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel<234>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel<100>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    auto stream_alone = [&]() -> float {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, s2);
            hipLaunchKernelGGL(stream_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s2, a, b, c, rows);
            hipEventRecord(e1, s2);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        return best;
    };
    printf("stream kernel alone: %.1f us (%lld rows x 768 fp16: 2 reads + 1 write = %.0f MB)\n", stream_alone() * 1e3, rows, rows * 768 * 2 * 3 / 1e6);
    for (int variant = 0; variant < 3; ++variant) {
        // 0: hog with ~240 registers + all LDS (the GEMM's footprint); 1: ~240 registers, no LDS; 2: ~120 registers + all LDS
        const long long spin = 200000;  // 2 ms
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(g0, s1);
            if (variant == 0) hipLaunchKernelGGL(hog_kernel<234>, dim3(256), dim3(512), lds, s1, out, spin, lds);
            if (variant == 1) hipLaunchKernelGGL(hog_kernel<234>, dim3(256), dim3(512), 0, s1, out, spin, 0);
            if (variant == 2) hipLaunchKernelGGL(hog_kernel<100>, dim3(256), dim3(512), lds, s1, out, spin, lds);
            hipEventRecord(g1, s1);
            // give the hog 200 us to occupy the chip, then the streaming kernel on the other stream
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 200.0) {}
            hipEventRecord(e0, s2);
            hipLaunchKernelGGL(stream_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s2, a, b, c, rows);
            hipEventRecord(e1, s2);
            hipEventSynchronize(e1);
            const bool hog_done = hipEventQuery(g1) == hipSuccess;
            hipEventSynchronize(g1);
            float ms_s, ms_g; hipEventElapsedTime(&ms_s, e0, e1); hipEventElapsedTime(&ms_g, g0, g1);
            printf("variant %d rep %d: stream kernel under the hog took %.1f us, finished %s the hog (hog %.2f ms)\n", variant, rep, ms_s * 1e3,
                   hog_done ? "AFTER" : "BEFORE", ms_g);
        }
    }
    return 0;
}
