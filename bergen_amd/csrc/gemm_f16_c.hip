// gemm_f16_c.hip — instantiations of the persistent encoder GEMM (gemm_f16_persist.h).
#include "gemm_f16_persist.h"

namespace {
int g_n_cu = 0;
int n_cu() {
    if (g_n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_n_cu = prop.multiProcessorCount;
        if (g_n_cu <= 0) g_n_cu = 256;
    }
    return g_n_cu;
}
}  // namespace

// epi in {0, bias-col, bias-row, bias-col + GELU}; pst: 1 = burst stores (production), 0 = deferred stores,
// 3 = burst + non-temporal stores (gemm_f16_persist.h)
#define BH_PERSIST_EPI(P)                                                                                         \
    switch (epi) {                                                                                                \
        case 0: return bh_gemm_launch_persist<0, P>(a, n_cu(), s);                                                \
        case BH_EPI_BIAS_COL: return bh_gemm_launch_persist<BH_EPI_BIAS_COL, P>(a, n_cu(), s);                    \
        case BH_EPI_BIAS_ROW: return bh_gemm_launch_persist<BH_EPI_BIAS_ROW, P>(a, n_cu(), s);                    \
        case BH_EPI_BIAS_COL | BH_EPI_GELU: return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_GELU, P>(a, n_cu(), s); \
        case BH_EPI_BIAS_ROW | BH_EPI_SEGMAX:                                                                     \
            if (P == 1) return bh_gemm_launch_persist<BH_EPI_BIAS_ROW | BH_EPI_SEGMAX, 1>(a, n_cu(), s);          \
            break;                                                                                                \
        case BH_EPI_BATCHED:                                                                                      \
            if (P == 1) return bh_gemm_launch_persist<BH_EPI_BATCHED, 1>(a, n_cu(), s);                           \
            break;                                                                                                \
        case BH_EPI_BIAS_COL | BH_EPI_LNA:                                                                        \
            if (P == 33) return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_LNA, 33>(a, n_cu(), s);           \
            break;                                                                                                \
        case BH_EPI_BIAS_ROW | BH_EPI_LNA:                                                                        \
            if (P == 33) return bh_gemm_launch_persist<BH_EPI_BIAS_ROW | BH_EPI_LNA, 33>(a, n_cu(), s);           \
            break;                                                                                                \
        case BH_EPI_BIAS_COL | BH_EPI_GELU | BH_EPI_LNA:                                                          \
            if (P == 35) return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_GELU | BH_EPI_LNA, 35>(a, n_cu(), s); \
            if (P == 33) return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_GELU | BH_EPI_LNA, 33>(a, n_cu(), s); \
            break;                                                                                                \
        case BH_EPI_BIAS_COL | BH_EPI_RESLN:                                                                      \
            if (P == 33) return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_RESLN, 33>(a, n_cu(), s);         \
            break;                                                                                                \
        case BH_EPI_BIAS_COL | BH_EPI_SWIGLU:                                                                     \
            if (P == 3) return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_SWIGLU, 3>(a, n_cu(), s);          \
            if (P == 35) return bh_gemm_launch_persist<BH_EPI_BIAS_COL | BH_EPI_SWIGLU, 35>(a, n_cu(), s);        \
            break;                                                                                                \
    }                                                                                                             \
    return hipErrorNotSupported;

hipError_t bh_gemm_persist(const BhGemmArgs& a, int epi, int pst, hipStream_t s) {
    if (pst == 1) { BH_PERSIST_EPI(1) }
    if (pst == 0) { BH_PERSIST_EPI(0) }
    if (pst == 3) { BH_PERSIST_EPI(3) }
    if (pst == 33) { BH_PERSIST_EPI(33) }  // 1 + full-line stores through LDS (gemm_f16_persist.h PST bit 32)
    if (pst == 35) { BH_PERSIST_EPI(35) }  // 3 + the same
    if (pst == 16) { BH_PERSIST_EPI(16) }  // deferred stores + alternating loader teams (needs an even number of stages >= 8)
    if (pst == 5) { BH_PERSIST_EPI(5) }  // bench-only: math, no stores
    if (pst == 9) { BH_PERSIST_EPI(9) }  // bench-only: no epilogue
    return hipErrorNotSupported;
}
