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

// bf16 storage helpers (bf16 = the upper 16 bits of an fp32; conversion rounds to nearest even)
typedef unsigned short rd_bf16_t;
typedef unsigned rd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rd_bf16_t rd_f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (rd_bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (rd_bf16_t)(u >> 16);
}
__device__ __forceinline__ float rd_bf2f(rd_bf16_t h) { return __uint_as_float((unsigned)h << 16); }
// 8 packed bf16 (one 16-byte access) <-> 8 floats
__device__ __forceinline__ void rd_unpack8(const rd_u32x4 p, float (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(p[i] << 16);
        v[2 * i + 1] = __uint_as_float(p[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ rd_u32x4 rd_pack8(const float (&v)[8])
{
    rd_u32x4 p;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = (unsigned)rd_f2bf(v[2 * i]) | ((unsigned)rd_f2bf(v[2 * i + 1]) << 16);
    return p;
}
