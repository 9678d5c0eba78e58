// The fp32 -> h2 (two fp16 planes: hi = fp16(s), lo = fp16(s - hi), s = 16 x value; conv_igemm_h2.hip) split of eight values, shared by
// every kernel that writes h2 records.  It is the VALU load of their epilogues (probe build of the eight-phase kernel: 18 000 - 19 000
// cycles of epilogue per 256x256 tile, two wavefronts per SIMD each converting 8 192 values), so it is written instruction by instruction:
//   range check   one unsigned max of the |bit patterns| + ONE compare for the eight (|s| > 65504, inf and NaN all order above 0x477fe000)
//   clamp         v_med3_f32 (no NaN canonicalisation in front of it)
//   hi            v_cvt_pk_f16_f32, two values per instruction (round to nearest even)
//   lo            v_fma_mixlo_f16 / v_fma_mixhi_f16: fp16(c - fp32(hi)) in ONE instruction per value - the fp16 hi half is an operand
//                 of the fp32 fma (exact difference: |c - hi| <= half an fp16 ulp of c), its result is rounded to fp16 once;
//                 bit-identical to (_Float16)(c - (float)hi) (tests/test_gpu_h2.py::test_split_h2_is_the_exact_two_term_split)
// ~4 VALU instructions per value instead of ~8 from the plain C form.
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 rd_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bool rd_h2_split8(const float (&s)[8], rd_h8& hi, rd_h8& lo)
{
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    unsigned m = 0;
    u4_t hp, lp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float s0 = s[2 * q], s1 = s[2 * q + 1];
        const unsigned u0 = __float_as_uint(s0) & 0x7fffffffu, u1 = __float_as_uint(s1) & 0x7fffffffu;
        const unsigned u01 = u0 > u1 ? u0 : u1;
        m = m > u01 ? m : u01;
        const f2_t c = {__builtin_amdgcn_fmed3f(s0, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(s1, -65504.f, 65504.f)};
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(c, h2_t));
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(c[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(c[1]));
        hp[q] = h;
        lp[q] = l;
    }
    hi = __builtin_bit_cast(rd_h8, hp);
    lo = __builtin_bit_cast(rd_h8, lp);
    return m > 0x477fe000u;  // the bits of 65504.f
}
