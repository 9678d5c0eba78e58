// Launch arguments of the bf16x3 (fp32-accurate, three bf16 planes per operand) convolution kernels.
#pragma once
#include "conv_bf16_common.h"

struct ConvX3Args {
    ConvBArgs b;             // d.x / d.w = plane 0 of the bf16 planes; d.y fp32 output or null; d.res fp32 residual or null
    unsigned x_plane_bytes;  // distance between activation planes
    unsigned w_plane_bytes;  // distance between weight planes
    void* y_planes;          // optional: the result as three bf16 planes (input of the next bf16x3 layer) or null
    long long y_plane_elems;
    const void* res_planes;  // optional: the residual as three bf16 planes (summed exactly in fp32), used when d.res is null
    long long res_plane_elems;
};

// fp32 -> three bf16 terms (round to nearest even each; the remainders are exact in fp32)
__device__ __forceinline__ void rd_split3(const float v, bf16_t& t1, bf16_t& t2, bf16_t& t3)
{
    t1 = f2bf(v);
    const float r1 = v - bf2f(t1);
    t2 = f2bf(r1);
    const float r2 = r1 - bf2f(t2);
    t3 = f2bf(r2);
}

// Tail of the bf16x3 epilogues for 8 consecutive channels [ch, ch+8) of output pixel `pix`: v = scale*acc + shift on entry;
// adds the residual (fp32 tensor, or three planes summed exactly), applies the activation, stores fp32 and / or the three
// bf16 planes of the result.
__device__ __forceinline__ void x3_finish_row8(const ConvX3Args& ax, float (&v)[8], const long long pix, const int ch)
{
    const rdpn6d_conv_desc& d = ax.b.d;
    if (d.res) {
        const float* rp = d.res + pix * d.res_cs + d.res_co + ch;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] += r0[q];
            v[4 + q] += r1[q];
        }
    } else if (ax.res_planes) {
        const bf16_t* rp = reinterpret_cast<const bf16_t*>(ax.res_planes) + pix * d.res_cs + d.res_co + ch;
        float r[3][8];
#pragma unroll
        for (int p = 0; p < 3; ++p) rd_unpack8(*reinterpret_cast<const rd_u32x4*>(rp + p * ax.res_plane_elems), r[p]);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (r[0][q] + r[1][q]) + r[2][q];  // exact: the planes re-assemble the fp32 value
    }
    conv_bf16_act(v, d.act, d.slope);
    if (d.y) {
        float* op = d.y + pix * d.out_cs + d.out_co + ch;
        const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(op) = o0;
        *reinterpret_cast<f32x4*>(op + 4) = o1;
    }
    if (ax.y_planes) {
        bf16_t t[3][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rd_split3(v[q], t[0][q], t[1][q], t[2][q]);
        bf16_t* pp = reinterpret_cast<bf16_t*>(ax.y_planes) + pix * d.out_cs + d.out_co + ch;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            rd_u32x4 u;
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = (unsigned)t[p][2 * q] | ((unsigned)t[p][2 * q + 1] << 16);
            *reinterpret_cast<rd_u32x4*>(pp + p * ax.y_plane_elems) = u;
        }
    }
}
