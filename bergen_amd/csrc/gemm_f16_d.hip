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

hipError_t p16_v(const BhGemmArgs& a, int epi, bool nontemporal, hipStream_t s) {
    switch (epi) {
        case 0: return bh_gemm_launch_p16<0, false>(a, n_cu(), s);
        case BH_EPI_BIAS_COL: return bh_gemm_launch_p16<BH_EPI_BIAS_COL, false>(a, n_cu(), s);
        case BH_EPI_BIAS_ROW: return bh_gemm_launch_p16<BH_EPI_BIAS_ROW, false>(a, n_cu(), s);
        case BH_EPI_BIAS_COL | BH_EPI_SWIGLU:
            return nontemporal ? bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_SWIGLU, true>(a, n_cu(), s)
                               : bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_SWIGLU, false>(a, n_cu(), s);
        case BH_EPI_BIAS_COL | BH_EPI_ROTARY: return bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_ROTARY, false>(a, n_cu(), s);
        case BH_EPI_BIAS_COL | BH_EPI_SWIGLU | BH_EPI_GELU:
            return nontemporal ? bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_SWIGLU | BH_EPI_GELU, true>(a, n_cu(), s)
                               : bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_SWIGLU | BH_EPI_GELU, false>(a, n_cu(), s);
        case BH_EPI_BIAS_COL | BH_EPI_GELU:
            return nontemporal ? bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_GELU, true>(a, n_cu(), s)
                               : bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_GELU, false>(a, n_cu(), s);
    }
    return hipErrorNotSupported;
}
// mode: 1 = production, 16 * ABL + 1 = bench-only ablations of the GELU kernel
hipError_t bh_gemm_p16(const BhGemmArgs& a, int epi, bool nontemporal, int mode, hipStream_t s) {
    if (mode >= 16) {
        if (epi != (BH_EPI_BIAS_COL | BH_EPI_GELU)) return hipErrorNotSupported;
        switch (mode >> 4) {
#define BH_P16_ABL(X) \
    case X: return bh_gemm_launch_p16<BH_EPI_BIAS_COL | BH_EPI_GELU, true, X>(a, n_cu(), s);
            BH_P16_ABL(1) BH_P16_ABL(2) BH_P16_ABL(4) BH_P16_ABL(8) BH_P16_ABL(16) BH_P16_ABL(9) BH_P16_ABL(10) BH_P16_ABL(12) BH_P16_ABL(14) BH_P16_ABL(13) BH_P16_ABL(11)
#undef BH_P16_ABL
        }
        return hipErrorNotSupported;
    }
    return p16_v(a, epi, nontemporal, s);
}
