// Shared pieces of the h2 (two-plane fp16, fp32-accurate) convolution kernels: conv_igemm_h2.hip (256x256 eight-phase kernel, 2x2-wave
// tile kernel) and conv_igemm_h2_pp.hip (8-wave ping-pong tile kernel).  Format, range and error analysis: header of conv_igemm_h2.hip.
#pragma once
#include "conv_bf16_common.h"
#include "h2_format.h"

#include <type_traits>

struct ConvH2Args {
    ConvBArgs b;         // d.x / d.w: h2 tensors; d.y: fp32 output or null; d.res: fp32 residual or null; channel counts REAL
    void* y_h2;          // optional: the result as an h2 tensor (geometry d.out_cs / d.out_co, multiples of 32)
    const void* res_h2;  // optional: the residual as an h2 tensor (geometry d.res_cs / d.res_co), used when d.res is null
    int* overflow_flag;  // set to 1 when an output had to be clamped to the fp16 range (may be null)
    // per-crop bias [B][4][Npad] added after scale / shift: row (b, variant) with variant = (last output row) * 2 + (last output column)
    // - the contribution of a spatially constant input slice to a ConvTranspose phase (pointwise_h2.hip); null = none
    const float* crop_bias;
    // split-K of the tile kernel (per-image batches: layer3 / layer4 of one crop are 256 / 64 rows x 2304 / 4608 reductions on a
    // handful of workgroups): gridDim.y K-slices of b.kper chunks each write raw fp32 partial tiles to `partial`
    // ([slice][mtiles*BM][Npad]); h2_splitk_reduce_kernel adds the slices in slice order (deterministic) and runs the epilogue
    float* partial;
    int nsplit, mpad;
    // Fused 1x1 output convolution (eight-phase kernel, one N tile = all channels of a pixel in the workgroup): out[pix][n] =
    // fuse_scale[n] * sum_c relu(...)[pix][c] * fuse_w[n][c] + fuse_bias[n], n < fuse_n <= 64 - the activation tile is multiplied by the
    // 1x1 weights (h2 records [64][N/32][hi|lo]) straight out of the epilogue and never written (cdpn_rot_head_region.py:130-138)
    const void* fuse_w;
    const float* fuse_scale;
    const float* fuse_bias;
    float* fuse_out;
    int fuse_cs, fuse_n;
    // the weights once more, FRAGMENT-MAJOR ([Npad/32][ntaps][Cin/32][slot 0..7][row 0..31][8 halfs]: rdpn6d_h2_weight_frag), for the kernels
    // that load their weight fragments straight from L2 instead of staging the weight tile through LDS (conv_igemm_h2_pp.hip, BFG); null = none
    const void* w_frag;
    // measurement only (rdpn6d_conv_h2_set_clock_probe; null = off): workgroup 0 of the 256x256 eight-phase kernel leaves
    // {s_memtime, s_memrealtime} at its start and its end here - shader-clock ticks against the constant 100 MHz counter = the clock the
    // power-limited kernel actually ran at (bench.py: roofline.clock_ghz; boxes of the pool differ by 7 % on exactly this)
    unsigned long long* clk_probe;
};


namespace {

constexpr float H2_SCALE = 16.f, H2_INV_SCALE = 1.f / 16.f, H2_MAX = 65504.f;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int V>
using ic = std::integral_constant<int, V>;

// fp32 value (already multiplied by the tensor scale) -> hi, lo; returns true when it had to be clamped
__device__ __forceinline__ bool h2_split(float s, _Float16& hi, _Float16& lo)
{
    const bool over = !(fabsf(s) <= H2_MAX);  // also catches NaN
    s = fminf(fmaxf(s, -H2_MAX), H2_MAX);
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
    return over;
}
// eight values at once (the epilogues' unit): h2_format.h
__device__ __forceinline__ bool h2_split8(const float (&s)[8], f16x8& hi, f16x8& lo) { return rd_h2_split8(s, hi, lo); }

// Tail of the h2 epilogues for 8 consecutive channels [ch, ch+8) of output pixel `pix`: v = scale*acc + shift on entry;
// residual (fp32 tensor or h2 record), activation, fp32 store and / or the h2 record of the result.
// PRE: the h2 residual record of these 8 channels was loaded by the caller (rh / rl) - kernels that issue every residual load of
// their tile up front instead of one dependent load per row group.
template <bool PRE>
__device__ __forceinline__ void h2_finish_row8_t(const ConvH2Args& ax, float (&v)[8], const long long pix, const int ch, const f16x8 rh_pre,
                                                 const f16x8 rl_pre)
{
    const rdpn6d_conv_desc& d = ax.b.d;
    if (ax.crop_bias) {
        const int ohw = d.OH * d.OW;
        const int b = (int)(pix / ohw);
        const int r = (int)(pix - (long long)b * ohw);
        const int oy = r / d.OW, ox = r - oy * d.OW;
        const float* cb = ax.crop_bias + (long long)(b * 4 + (oy == d.OH - 1 ? 2 : 0) + (ox == d.OW - 1 ? 1 : 0)) * d.Npad + ch;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb), c1 = *reinterpret_cast<const f32x4*>(cb + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] += c0[q];
            v[4 + q] += c1[q];
        }
    }
    if (d.res) {
        const float* rp = d.res + pix * d.res_cs + d.res_co + ch;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] += r0[q];
            v[4 + q] += r1[q];
        }
    } else if (ax.res_h2) {
        f16x8 rh, rl;
        if constexpr (PRE) {
            rh = rh_pre;
            rl = rl_pre;
        } else {
            const int c = d.res_co + ch;
            const _Float16* rp = reinterpret_cast<const _Float16*>(ax.res_h2) + pix * (2 * d.res_cs) + (c >> 5) * 64 + (c & 31);
            rh = *reinterpret_cast<const f16x8*>(rp);
            rl = *reinterpret_cast<const f16x8*>(rp + 32);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += ((float)rh[q] + (float)rl[q]) * H2_INV_SCALE;  // the activation as the h2 tensor holds it (22 significand bits)
    }
    conv_bf16_act(v, d.act, d.slope);
    if (d.y) {
        float* op = d.y + pix * d.out_cs + d.out_co + ch;
        const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(op) = o0;
        *reinterpret_cast<f32x4*>(op + 4) = o1;
    }
    if (ax.y_h2) {
        f16x8 hi, lo;
        float sv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sv[q] = v[q] * H2_SCALE;
        const bool over = h2_split8(sv, hi, lo);
        const int c = d.out_co + ch;
        _Float16* pp = reinterpret_cast<_Float16*>(ax.y_h2) + pix * (2 * d.out_cs) + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<f16x8*>(pp) = hi;
        *reinterpret_cast<f16x8*>(pp + 32) = lo;
        if (over && ax.overflow_flag) *ax.overflow_flag = 1;
    }
}
__device__ __forceinline__ void h2_finish_row8(const ConvH2Args& ax, float (&v)[8], const long long pix, const int ch)
{
    const f16x8 z = {};
    h2_finish_row8_t<false>(ax, v, pix, ch, z, z);
}

__device__ __forceinline__ f32x16 h2_mfma(const u32x4 a, const u32x4 b, const f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the six partial products of a 128-byte row pair, smallest terms first: (slot pair of A, slot pair of B)
#define H2_PAIRS constexpr int H2_PA[6] = {2, 0, 0, 3, 1, 1}, H2_PB[6] = {0, 2, 0, 1, 3, 1}

__device__ __forceinline__ long long h2_pixel_of(const ConvBArgs& a, const long long m)
{
    const rdpn6d_conv_desc& d = a.d;
    if (a.linear_out) return m;
    const int mm = (int)m;
    const int b = mm / a.HoWo;
    const int rem = mm - b * a.HoWo;
    const int oy = rem / d.Wo;
    const int ox = rem - oy * d.Wo;
    return ((long long)b * d.OH + (oy * d.osy + d.ooy)) * d.OW + (ox * d.osx + d.oox);
}

}  // namespace
