// Shared host-side helpers of the library (error string, HIP error check, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/rho2sdf_hip.h"

inline thread_local std::string g_err;
inline int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(R2S_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc(&p, bytes) != hipSuccess) return -1;
            want = bytes;
        }
        cap = want;
        return 0;
    }
    int ensure_exact(size_t bytes)   // no growth margin (very large buffers)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        cap = bytes;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() { return reinterpret_cast<T*>(p); }
};


inline int check_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(R2S_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    if (device >= n) return fail(R2S_ERR_ARG, "device %d out of range (%d devices)", device, n);
    return 0;
}

// resolve `device` (-1 = current) and make it current
inline int use_device(int device)
{
    int rc = check_device(device < 0 ? 0 : device);
    if (rc) return rc;
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    return 0;
}

#define ENSURE(buf, bytes)                                                       \
    do {                                                                         \
        if ((buf).ensure(bytes)) return fail(R2S_ERR_NOMEM, "hipMalloc of %zu bytes failed", (size_t)(bytes)); \
    } while (0)
