// The kernels between the two-plane fp16 ("h2") convolutions of the point-wise fusion branch (resnet_backbone.py:303-340,
// md_pointnet :39-54), reading and writing h2 tensors ([pixels][C/32][2][32] fp16 holding 16*a = hi + lo, see
// conv_igemm_h2.hip) so that no fp32 copy of those activations is ever made:
//   upsample_bilinear_h2   UpsamplingBilinear2d(scale_factor=f), align_corners=True (:280), fp32 interpolation of the values
//   xyz_subsample_h2       nearest 1/step subsample of the crop's depth-xyz channels into one 32-channel group [x y z 0 ...]
//   global_max_concat_h2   max over the pixels of channels [0,C), broadcast into channels [C,2C) (:51-52)
#include "common.h"
#include <float.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float H2_SCALE = 16.f, H2_INV_SCALE = 1.f / 16.f, H2_MAX = 65504.f;

__device__ __forceinline__ bool h2_split1(float s, _Float16& hi, _Float16& lo)
{
    const bool over = !(fabsf(s) <= H2_MAX);
    s = fminf(fmaxf(s, -H2_MAX), H2_MAX);
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
    return over;
}
__device__ __forceinline__ long long h2_off(long long pix, int cs, int c) { return pix * (2 * (long long)cs) + (c >> 5) * 64 + (c & 31); }

__global__ void upsample_bilinear_h2_kernel(const _Float16* __restrict__ x, int B, int H, int W, int C, int f, _Float16* __restrict__ y,
                                            int* __restrict__ overflow_flag)
{
    const int Ho = H * f, Wo = W * f, C8 = C / 8;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * C8;
    bool over = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        long long p = i / C8;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const long long pb = (long long)b * H * W;
        float v[4][8];
        const long long src[4] = {pb + (long long)y0 * W + x0, pb + (long long)y0 * W + x1, pb + (long long)y1 * W + x0, pb + (long long)y1 * W + x1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const _Float16* sp = x + h2_off(src[k], C, c);
            const f16x8 h = *reinterpret_cast<const f16x8*>(sp), l = *reinterpret_cast<const f16x8*>(sp + 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = ((float)h[e] + (float)l[e]) * H2_INV_SCALE;
        }
        f16x8 oh, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float o = hy * (hx * v[0][e] + lx * v[1][e]) + ly * (hx * v[2][e] + lx * v[3][e]);  // as upsample_bilinear_kernel
            _Float16 h, l;
            over |= h2_split1(o * H2_SCALE, h, l);
            oh[e] = h;
            ol[e] = l;
        }
        _Float16* dp = y + h2_off(((long long)b * Ho + oy) * Wo + ox, C, c);
        *reinterpret_cast<f16x8*>(dp) = oh;
        *reinterpret_cast<f16x8*>(dp + 32) = ol;
    }
    if (over && overflow_flag) *overflow_flag = 1;
}

__global__ void xyz_subsample_h2_kernel(const float* __restrict__ x, int B, int xc, int R, int step, _Float16* __restrict__ y, int out_cs,
                                        int out_co, int* __restrict__ overflow_flag)
{
    const int Ro = R / step;
    const long long total = (long long)B * Ro * Ro;
    bool over = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Ro);
        const int oy = (int)((i / Ro) % Ro);
        const int b = (int)(i / ((long long)Ro * Ro));
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            _Float16 hh, ll;
            over |= h2_split1(x[(((long long)b * xc + 3 + c) * R + oy * step) * R + ox * step] * H2_SCALE, hh, ll);
            h[c] = hh;
            l[c] = ll;
        }
        _Float16* dp = y + h2_off(i, out_cs, out_co);  // the whole 32-channel group: [x y z 0 ... 0]
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        *reinterpret_cast<f16x8*>(dp) = h;
        *reinterpret_cast<f16x8*>(dp + 32) = l;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            *reinterpret_cast<f16x8*>(dp + 8 * q) = z;
            *reinterpret_cast<f16x8*>(dp + 32 + 8 * q) = z;
        }
    }
    if (over && overflow_flag) *overflow_flag = 1;
}

// grid = (C/64, B); block = 1024 = 16 pixel lanes x 64 channels.  (hi, lo) is a canonical form of the stored value, so the
// pixel with the largest reconstructed value is found exactly and its pair is what gets broadcast.
__global__ __launch_bounds__(1024) void global_max_concat_h2_kernel(_Float16* __restrict__ buf, int HW, int C, int cs)
{
    __shared__ float s_m[16][64];
    __shared__ _Float16 s_h[16][64], s_l[16][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, pl = threadIdx.x >> 6;
    const long long p0 = (long long)b * HW;
    float m = -FLT_MAX;
    _Float16 mh = (_Float16)0.f, ml = (_Float16)0.f;
    for (int p = pl; p < HW; p += 16) {
        const _Float16* sp = buf + h2_off(p0 + p, cs, c);
        const _Float16 h = sp[0], l = sp[32];
        const float v = (float)h + (float)l;
        if (v > m) { m = v; mh = h; ml = l; }
    }
    s_m[pl][cl] = m; s_h[pl][cl] = mh; s_l[pl][cl] = ml;
    __syncthreads();
    m = s_m[0][cl]; mh = s_h[0][cl]; ml = s_l[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (s_m[k][cl] > m) { m = s_m[k][cl]; mh = s_h[k][cl]; ml = s_l[k][cl]; }
    for (int p = pl; p < HW; p += 16) {
        _Float16* dp = buf + h2_off(p0 + p, cs, C + c);
        dp[0] = mh;
        dp[32] = ml;
    }
}

}  // namespace

extern "C" int rdpn6d_upsample_bilinear_h2(const void* x, int B, int H, int W, int C, int factor, void* y, int* overflow_flag, void* stream)
{
    RD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && factor >= 1, "shape (C % 32)");
    const long long total = (long long)B * H * factor * W * factor * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    hipLaunchKernelGGL(upsample_bilinear_h2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, B, H, W, C, factor,
                       (_Float16*)y, overflow_flag);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_xyz_subsample_h2(const float* x, int B, int xc, int R, int step, void* y, int out_cs, int out_co, int* overflow_flag,
                                       void* stream)
{
    RD_REQUIRE(x && y && B > 0 && xc >= 6 && R > 0 && step > 0 && R % step == 0, "shape");
    RD_REQUIRE(out_cs % 32 == 0 && out_co % 32 == 0 && out_co + 32 <= out_cs, "output slice: one whole 32-channel group");
    const long long total = (long long)B * (R / step) * (R / step);
    hipLaunchKernelGGL(xyz_subsample_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, xc, R, step,
                       (_Float16*)y, out_cs, out_co, overflow_flag);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_global_max_concat_h2(void* buf, int B, int HW, int C, int cs, void* stream)
{
    RD_REQUIRE(buf && B > 0 && HW > 0 && C > 0 && C % 64 == 0 && 2 * C <= cs && cs % 32 == 0, "shape");
    hipLaunchKernelGGL(global_max_concat_h2_kernel, dim3(C / 64, B), dim3(1024), 0, (hipStream_t)stream, (_Float16*)buf, HW, C, cs);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
