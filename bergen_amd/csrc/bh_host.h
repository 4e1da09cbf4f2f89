// bh_host.h — host-side helpers shared by the translation units of libbergen_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

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

// Copy a library-side struct into a caller's struct whose first field is `int32_t struct_size` = the caller's sizeof
// (include/bergen_hip.h, BH_VERSION 141): at most that many bytes are written, the caller's struct_size stays as it was.
// Returns false when the caller's size is implausible (smaller than `min_size`).
template <typename T>
inline bool bh_copy_sized(T* out, const T& src, size_t min_size) {
    const int32_t want = out->struct_size;
    if (want < (int32_t)min_size) return false;
    const size_t n = (size_t)want < sizeof(T) ? (size_t)want : sizeof(T);
    unsigned char tmp[sizeof(T)];
    memcpy(tmp, &src, sizeof(T));
    memcpy(tmp, &want, sizeof want);
    memcpy(out, tmp, n);
    return true;
}

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
