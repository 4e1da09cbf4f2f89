// gemm_f16_b.hip — further instantiations of the encoder GEMM template (split from gemm_f16.hip so that the
// two translation units compile in parallel): the BK = 32 configurations, the 8-wave 256x128 configuration
// and the bench-only ablations.  Design notes: gemm_f16.hip, gemm_f16_kernel.h.
#include "gemm_f16_kernel.h"

#define CFG2(E) bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2>(a, s)
hipError_t bh_gemm_cfg2(const BhGemmArgs& a, int epi, hipStream_t s) { BH_GEMM_DISPATCH_EPI(epi, CFG2) }
#define CFG3(E) bh_gemm_launch_cfg<32, 2, 4, 4, 2, 4, E, 2>(a, s)
hipError_t bh_gemm_cfg3(const BhGemmArgs& a, int epi, hipStream_t s) { BH_GEMM_DISPATCH_EPI(epi, CFG3) }
#define CFG4(E) bh_gemm_launch_cfg<64, 4, 2, 2, 2, 3, E, 2>(a, s)
hipError_t bh_gemm_cfg4(const BhGemmArgs& a, int epi, hipStream_t s) { BH_GEMM_DISPATCH_EPI(epi, CFG4) }

// bench-only ablations (ABL bits: 1 no LDS-DMA in loop, 2 no MFMA, 4 no fragment reads, 8 no epilogue), all
// with the bias+GELU epilogue: 11-16 on configuration 5 (256x256, 8 waves), 21-26 on configuration 2.
hipError_t bh_gemm_ablate(const BhGemmArgs& a, int which, hipStream_t s) {
    constexpr int E = BH_EPI_BIAS_COL | BH_EPI_GELU;
    switch (which) {
        case 11: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 1>(a, s);
        case 12: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 2 | 4>(a, s);
        case 13: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 4>(a, s);
        case 14: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 8>(a, s);
        case 15: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 1 | 4>(a, s);
        case 16: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 1 | 4 | 8>(a, s);
        case 17: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 16>(a, s);
        case 18: return bh_gemm_launch_cfg<64, 2, 4, 4, 2, 2, E, 2, false, 2 | 4 | 8>(a, s);
        case 21: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 1>(a, s);
        case 22: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 2 | 4>(a, s);
        case 23: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 4>(a, s);
        case 24: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 8>(a, s);
        case 25: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 1 | 4>(a, s);
        case 26: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 1 | 4 | 8>(a, s);
        case 27: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 16>(a, s);
        case 28: return bh_gemm_launch_cfg<32, 2, 2, 4, 2, 3, E, 2, false, 2 | 4 | 8>(a, s);
    }
    return hipErrorInvalidValue;
}
