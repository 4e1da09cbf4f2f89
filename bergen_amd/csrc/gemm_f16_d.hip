// gemm_f16_d.hip — instantiations of the 16x16x32 persistent encoder GEMM (gemm_f16_p16.h; option gemm_mfma16).
#include "gemm_f16_p16.h"

namespace {
int g_n_cu = 0;
int n_cu() {
    if (g_n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_n_cu = prop.multiProcessorCount;
        if (g_n_cu <= 0) g_n_cu = 256;
    }
    return g_n_cu;
}
}  // namespace

template <int SCHED>
hipError_t p16_sched(const BhGemmArgs& a, int epi, bool nontemporal, hipStream_t s) {
    if (epi == BH_EPI_BIAS_COL)
        return nontemporal ? bh_gemm_launch_p16<BH_EPI_BIAS_COL, true, SCHED>(a, n_cu(), s) : bh_gemm_launch_p16<BH_EPI_BIAS_COL, false, SCHED>(a, n_cu(), s);
    if (epi == (BH_EPI_BIAS_COL | BH_EPI_GELU))
        return nontemporal ? bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_GELU, true, SCHED>(a, n_cu(), s)
                           : bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_GELU, false, SCHED>(a, n_cu(), s);
    return hipErrorNotSupported;
}

hipError_t bh_gemm_p16(const BhGemmArgs& a, int epi, bool nontemporal, int sched, hipStream_t s) {
    switch (sched) {
        case 0: return p16_sched<0>(a, epi, nontemporal, s);
        case 1: return p16_sched<1>(a, epi, nontemporal, s);
        case 2: return p16_sched<2>(a, epi, nontemporal, s);
        case 3: return p16_sched<3>(a, epi, nontemporal, s);
    }
    return hipErrorNotSupported;
}
