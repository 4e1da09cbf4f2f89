// bh_host.h — host-side helpers shared by the translation units of libbergen_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bergen_hip.h"

// Records a thread-local error message (bh_last_error) and returns `code`.  Defined in index.hip.
int bh_fail(int code, const char* fmt, ...);

#define BH_HIP_TRY(expr)                                                                                   \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            return bh_fail(_e == hipErrorOutOfMemory ? BH_ENOMEM : BH_EHIP, "%s failed: %s (%s:%d)", #expr, \
                           hipGetErrorString(_e), __FILE__, __LINE__);                                     \
    } while (0)

// Grow-only device buffer.
template <typename T>
struct BhDevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n, bool zero_new = false, hipStream_t st = nullptr) {
        if (n <= cap) return BH_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        BH_HIP_TRY(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        if (zero_new) BH_HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), st));
        return BH_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
