// ubench_mfma_power.hip — energy per flop of the two fp16 MFMA shapes on a full chip (gfx950), from the board's power sensor.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mfma_power profiles/ubench_mfma_power.hip && /tmp/ubench_mfma_power
// The dense scan runs at the board's power cap (profiles/r03_power_clock.json): what bounds it is joules per pass.  This
// asks how many of those joules the matrix pipe needs, and whether the shape of the MFMA matters:
//   v_mfma_f32_16x16x32_f16 (what scan_topk256.hip issues)  vs  v_mfma_f32_32x32x16_f16, both with 2 waves per SIMD and
//   4 independent accumulator chains per wave, operands = random fp16 values (the multipliers' switching activity depends
//   on the data: an all-zero leg is there for contrast).
// Each leg runs ~2 s while the host samples power1_input / freq1_input of this GPU's hwmon node (found through
// hipDeviceGetPCIBusId) every 20 ms.  Prints one JSON line per leg.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include <dirent.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <bool BIG>
__global__ void __launch_bounds__(512, 2) mfma_loop(const half8* __restrict__ src, float* out, int iters) {
    // 8 distinct A and B fragments per lane, rotated through: operands change every instruction as they do in a GEMM
    half8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(blockIdx.x * 512 + threadIdx.x) * 16 + i];
        b[i] = src[(blockIdx.x * 512 + threadIdx.x) * 16 + 8 + i];
    }
    floatx16 acc[4];
    floatx4 acc4[4];
    for (int c = 0; c < 4; ++c) {
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        for (int v = 0; v < 4; ++v) acc4[c][v] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (BIG)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + c) & 7], acc[c], 0, 0, 0);
                else
                    acc4[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[(u + c) & 7], acc4[c], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) {
        for (int v = 0; v < 16; ++v) s += acc[c][v];
        for (int v = 0; v < 4; ++v) s += acc4[c][v];
    }
    if (s == 12345.678f) out[0] = s;
}

static std::string hwmon_dir() {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) != hipSuccess) return "";
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    const std::string base = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon";
    DIR* d = opendir(base.c_str());
    if (!d) return "";
    std::string res;
    while (dirent* e = readdir(d))
        if (strncmp(e->d_name, "hwmon", 5) == 0) res = base + "/" + e->d_name;
    closedir(d);
    return res;
}

static double read_num(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

template <bool BIG>
static void leg(const char* name, const half8* src, float* out, const std::string& hw, double seconds) {
    const int iters = 400;  // 400 x 32 MFMAs per wave and launch (~0.3 s per launch)
    const double flop_per_launch = (BIG ? 32768.0 : 16384.0) * 32.0 * iters * 8 * 256;
    hipLaunchKernelGGL((mfma_loop<BIG>), dim3(256), dim3(512), 0, 0, src, out, iters);
    (void)hipDeviceSynchronize();
    std::atomic<bool> stop{false};
    std::vector<double> w, f;
    std::thread sampler([&] {
        while (!stop.load()) {
            if (!hw.empty()) {
                w.push_back(read_num(hw + "/power1_input") / 1e6);
                f.push_back(read_num(hw + "/freq1_input") / 1e6);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((mfma_loop<BIG>), dim3(256), dim3(512), 0, 0, src, out, iters);
        (void)hipDeviceSynchronize();
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    stop.store(true);
    sampler.join();
    // skip the first 0.5 s of samples (the sensor averages)
    double ws = 0, fs = 0;
    int n = 0;
    for (size_t i = 25; i < w.size(); ++i) {
        ws += w[i];
        fs += f[i];
        ++n;
    }
    const double watts = n ? ws / n : -1, mhz = n ? fs / n : -1;
    const double tflops = flop_per_launch * launches / el / 1e12;
    printf("{\"leg\": \"%s\", \"tflops\": %.1f, \"watts\": %.1f, \"sclk_mhz\": %.0f, \"joule_per_pflop\": %.3f, \"samples\": %d, \"seconds\": %.2f}\n", name, tflops, watts, mhz,
           watts > 0 ? watts / tflops * 1e3 : -1.0, n, el);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const std::string hw = hwmon_dir();
    printf("{\"hwmon\": \"%s\", \"idle_watts\": %.1f, \"cap_watts\": %.0f}\n", hw.c_str(), hw.empty() ? -1.0 : read_num(hw + "/power1_input") / 1e6,
           hw.empty() ? -1.0 : read_num(hw + "/power1_cap") / 1e6);
    const size_t n = (size_t)256 * 512 * 16;
    std::vector<_Float16> h(n * 8);
    srand(7);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);  // unit-norm-like magnitudes
    half8 *rnd, *zero;
    float* out;
    (void)hipMalloc(&rnd, n * sizeof(half8));
    (void)hipMalloc(&zero, n * sizeof(half8));
    (void)hipMalloc(&out, 64);
    (void)hipMemcpy(rnd, h.data(), n * sizeof(half8), hipMemcpyHostToDevice);
    (void)hipMemset(zero, 0, n * sizeof(half8));
    leg<false>("16x16x32_f16 random operands", rnd, out, hw, seconds);
    leg<true>("32x32x16_f16 random operands", rnd, out, hw, seconds);
    leg<false>("16x16x32_f16 zero operands", zero, out, hw, seconds);
    leg<true>("32x32x16_f16 zero operands", zero, out, hw, seconds);
    leg<false>("16x16x32_f16 random operands (again)", rnd, out, hw, seconds);
    return 0;
}
