// Shared helpers for the gfx950 kernels (device + host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/rdpn6d.h"

extern "C" void rdpn6d_set_error(const char* fmt, ...);

#define RD_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            rdpn6d_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return RDPN6D_EHIP;                                                              \
        }                                                                                    \
    } while (0)

#define RD_REQUIRE(cond, msg)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            rdpn6d_set_error("invalid argument: %s (%s) (%s:%d)", msg, #cond, __FILE__, __LINE__); \
            return RDPN6D_EINVAL;                                                            \
        }                                                                                    \
    } while (0)

#define RD_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            rdpn6d_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return RDPN6D_EHIP;                                                              \
        }                                                                                    \
    } while (0)

static inline int rd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
