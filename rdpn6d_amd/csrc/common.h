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

// Opt-in to more than 64 KiB of dynamic LDS for `kern`.  hipFuncSetAttribute acts on the CURRENT device's copy of the function, so the
// "done" state is kept per device (one bit each, set after the call succeeded): a process that drives several GPUs configures
// every one of them, and two host threads building plans at once can at worst both make the (idempotent) call.
#include <atomic>
struct RdLdsOptIn {
    std::atomic<unsigned long long> done[4];  // devices 0..255
};
static inline hipError_t rd_lds_opt_in(RdLdsOptIn& st, const void* kern, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<unsigned long long>& word = st.done[(dev >> 6) & 3];
    const unsigned long long bit = 1ull << (dev & 63);
    if (word.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) word.fetch_or(bit, std::memory_order_release);
    return e;
}
#define RD_LDS_OPT_IN(kern, bytes)                                                           \
    do {                                                                                     \
        static RdLdsOptIn _st;                                                               \
        RD_CHECK_HIP(rd_lds_opt_in(_st, reinterpret_cast<const void*>(kern), (int)(bytes))); \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-bit storage helpers.  The "low-precision" kernels (conv_igemm_bf16*.hip, pointwise_bf16.hip, the 16-bit forms in
// pointwise.hip / train_*.hip) are written once against these helpers and compiled TWICE (rdpn6d_amd/build.py):
//   default            bf16 (the upper 16 bits of an fp32, round to nearest even)   -> entry points  rdpn6d_*_bf16
//   -DRDPN6D_LP_FP16   IEEE fp16 (the reference's AMP dtype: torch.cuda.amp.autocast + GradScaler, engine.py:279-309)
//                                                                                   -> entry points  rdpn6d_*_fp16
// (the second set of objects is merged with `ld -r`, its rdpn6d_*_bf16 symbols are renamed and everything else is made local
// with llvm-objcopy, so that the two builds of the same source never meet at link time).  The type name rd_bf16_t and the
// helper names keep their bf16 spelling in both builds: they mean "the 16-bit storage format of this translation unit".
typedef unsigned short rd_bf16_t;
typedef unsigned rd_u32x4 __attribute__((ext_vector_type(4)));
typedef float rd_f32x2 __attribute__((ext_vector_type(2)));
#ifdef RDPN6D_LP_FP16
typedef _Float16 rd_lp_hw __attribute__((ext_vector_type(1)));
typedef _Float16 rd_lp_x2 __attribute__((ext_vector_type(2)));
typedef _Float16 rd_lp_x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned rd_f2bf_pk(float lo, float hi)
{
    const rd_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rd_lp_x2));  // v_cvt_f16_f32 x2: round to nearest even
}
__device__ __forceinline__ rd_bf16_t rd_f2bf(float f) { return __builtin_bit_cast(rd_bf16_t, (_Float16)f); }
__device__ __forceinline__ float rd_bf2f(rd_bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ void rd_unpack8(const rd_u32x4 p, float (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = p[i];  // (a scalar copy first: bit-casting the vector ELEMENT expression reads element 0 every time - clang 22)
        const rd_f32x2 t = __builtin_convertvector(__builtin_bit_cast(rd_lp_x2, u), rd_f32x2);
        v[2 * i] = t[0];
        v[2 * i + 1] = t[1];
    }
}
#define RD_LP_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rd_lp_x8, a), __builtin_bit_cast(rd_lp_x8, b), c, 0, 0, 0)
#else
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN) - one instruction per PAIR
// instead of the ~7-instruction integer sequence per value
typedef __bf16 rd_bf16x2_hw __attribute__((ext_vector_type(2)));
typedef __bf16 rd_lp_x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned rd_f2bf_pk(float lo, float hi)
{
    const rd_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rd_bf16x2_hw));
}
__device__ __forceinline__ rd_bf16_t rd_f2bf(float f) { return __builtin_bit_cast(rd_bf16_t, (__bf16)f); }
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
#define RD_LP_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_lp_x8, a), __builtin_bit_cast(rd_lp_x8, b), c, 0, 0, 0)
#endif
__device__ __forceinline__ rd_u32x4 rd_pack8(const float (&v)[8])
{
    rd_u32x4 p;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = rd_f2bf_pk(v[2 * i], v[2 * i + 1]);
    return p;
}

// 4 consecutive channels of an fp32 or bf16 activation <-> f32x4 (the training kernels are templated on the storage type)
template <typename T> __device__ __forceinline__ f32x4 rd_ld4(const T* p);
template <> __device__ __forceinline__ f32x4 rd_ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 rd_ld4<rd_bf16_t>(const rd_bf16_t* p)
{
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 v;
#ifdef RDPN6D_LP_FP16
    const unsigned ux = u.x, uy = u.y;  // (scalar copies first, see rd_unpack8)
    const rd_f32x2 a = __builtin_convertvector(__builtin_bit_cast(rd_lp_x2, ux), rd_f32x2);
    const rd_f32x2 b = __builtin_convertvector(__builtin_bit_cast(rd_lp_x2, uy), rd_f32x2);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
#else
    v[0] = __uint_as_float(u.x << 16);
    v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16);
    v[3] = __uint_as_float(u.y & 0xffff0000u);
#endif
    return v;
}
template <typename T> __device__ __forceinline__ void rd_st4(T* p, const f32x4 v);
template <> __device__ __forceinline__ void rd_st4<float>(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void rd_st4<rd_bf16_t>(rd_bf16_t* p, const f32x4 v)
{
    uint2 u;
    u.x = rd_f2bf_pk(v[0], v[1]);
    u.y = rd_f2bf_pk(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
template <typename T> __device__ __forceinline__ float rd_ld1(const T* p);
template <> __device__ __forceinline__ float rd_ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float rd_ld1<rd_bf16_t>(const rd_bf16_t* p) { return rd_bf2f(*p); }
template <typename T> __device__ __forceinline__ void rd_st1(T* p, float v);
template <> __device__ __forceinline__ void rd_st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void rd_st1<rd_bf16_t>(rd_bf16_t* p, float v) { *p = rd_f2bf(v); }
// V = 4 (16-byte fp32 or 8-byte 16-bit accesses) or 8 (16-byte accesses of a 16-bit tensor) consecutive channels as floats
template <typename T, int V> __device__ __forceinline__ void rd_ldv(const T* p, float (&v)[V])
{
    if constexpr (V == 4) {
        const f32x4 t = rd_ld4<T>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
        static_assert(sizeof(T) == 2 && V == 8, "8 channels per thread is the bf16 form");
        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(p), v);
    }
}

// streaming form of rd_ldv<T, 8>: the line is not kept in the caches (an operand that is not read again before it has left them anyway)
template <typename T> __device__ __forceinline__ void rd_ldv8_nt(const T* p, float (&v)[8])
{
    static_assert(sizeof(T) == 2, "16-bit tensors");
    rd_unpack8(__builtin_nontemporal_load(reinterpret_cast<const rd_u32x4*>(p)), v);
}

template <typename T, int V> __device__ __forceinline__ void rd_stv(T* p, const float (&v)[V])
{
    if constexpr (V == 4) rd_st4<T>(p, f32x4{v[0], v[1], v[2], v[3]});
    else *reinterpret_cast<rd_u32x4*>(p) = rd_pack8(v);
}

// The normalised value of the forward pass, ONE expression for bn_apply_kernel and for the backward kernels that re-derive the ReLU
// mask from x instead of reading the stored activation (relu == 2): the same operations in the same order, so the same sign.
__device__ __forceinline__ float bn_fwd_value(float x, float mean, float invstd, float gamma, float beta)
{
    return __builtin_fmaf((x - mean) * invstd, gamma, beta);
}
// y > 0 for the activation as it was STORED (type T): a positive fp32 value that rounds to zero in 16 bits has a zero mask
template <typename T> __device__ __forceinline__ bool bn_stored_positive(float v)
{
    if constexpr (sizeof(T) == 2) return rd_bf2f(rd_f2bf(v)) > 0.f;
    else return v > 0.f;
}


