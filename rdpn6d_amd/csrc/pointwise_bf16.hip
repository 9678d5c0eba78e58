// bf16 forms of the bandwidth-bound kernels between the bf16 convolutions (reduced-precision inference mode, see
// conv_igemm_bf16.hip).  NHWC bf16 activations, 16 bytes (8 channels) per lane; arithmetic in fp32, one rounding
// on the store.  Max-pooling and the global max are exact in bf16 (max commutes with rounding).
#include "common.h"
#include <float.h>

// MaxPool2d(kernel 3, stride 2, pad 1)  (resnet_backbone.py:275)
__global__ void maxpool3x3s2_bf16_kernel(const rd_bf16_t* __restrict__ x, int B, int H, int W, int C, rd_bf16_t* __restrict__ y)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, C8 = C / 8;
    const long long total = (long long)B * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long p = i / C8;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -FLT_MAX;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                float v[8];
                rd_unpack8(*reinterpret_cast<const rd_u32x4*>(x + (((long long)b * H + iy) * W + ix) * C + c8 * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *reinterpret_cast<rd_u32x4*>(y + (((long long)b * Ho + oy) * Wo + ox) * C + c8 * 8) = rd_pack8(m);
    }
}

extern "C" int rdpn6d_maxpool3x3s2_bf16(const void* x, int B, int H, int W, int C, void* y, void* stream)
{
    RD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const rd_bf16_t*)x, B, H, W, C, (rd_bf16_t*)y);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// Bilinear upsampling, align_corners=True (resnet_backbone.py:283 nn.UpsamplingBilinear2d)
__global__ void upsample_bilinear_bf16_kernel(const rd_bf16_t* __restrict__ x, int B, int H, int W, int C, int f,
                                              rd_bf16_t* __restrict__ y)
{
    const int Ho = H * f, Wo = W * f, C8 = C / 8;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long p = i / C8;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const rd_bf16_t* base = x + (long long)b * H * W * C + c8 * 8;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(base + ((long long)y0 * W + x0) * C), v00);
        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(base + ((long long)y0 * W + x1) * C), v01);
        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(base + ((long long)y1 * W + x0) * C), v10);
        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(base + ((long long)y1 * W + x1) * C), v11);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        *reinterpret_cast<rd_u32x4*>(y + (((long long)b * Ho + oy) * Wo + ox) * C + c8 * 8) = rd_pack8(o);
    }
}

extern "C" int rdpn6d_upsample_bilinear_bf16(const void* x, int B, int H, int W, int C, int factor, void* y, void* stream)
{
    RD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && factor >= 1, "shape");
    const long long total = (long long)B * H * factor * W * factor * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(upsample_bilinear_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const rd_bf16_t*)x, B, H, W, C, factor, (rd_bf16_t*)y);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// Nearest-neighbour subsample of the depth-xyz channels (3..5) of the fp32 NCHW crop into a bf16 NHWC slice
// (resnet_backbone.py:325 F.interpolate(xyz, size=(R/8, R/8)) with the default "nearest")
__global__ void xyz_subsample_bf16_kernel(const float* __restrict__ x, int B, int xc, int R, int step, rd_bf16_t* __restrict__ y,
                                          int out_cs, int out_co)
{
    const int Ro = R / step;
    const long long total = (long long)B * Ro * Ro;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Ro);
        const int oy = (int)((i / Ro) % Ro);
        const int b = (int)(i / ((long long)Ro * Ro));
#pragma unroll
        for (int c = 0; c < 3; ++c)
            y[i * out_cs + out_co + c] = rd_f2bf(x[(((long long)b * xc + 3 + c) * R + oy * step) * R + ox * step]);
    }
}

extern "C" int rdpn6d_xyz_subsample_bf16(const float* x, int B, int xc, int R, int step, void* y, int out_cs, int out_co,
                                         void* stream)
{
    RD_REQUIRE(x && y && B > 0 && xc >= 6 && R > 0 && step > 0 && R % step == 0, "shape");
    RD_REQUIRE(out_co + 3 <= out_cs, "output slice");
    const long long total = (long long)B * (R / step) * (R / step);
    hipLaunchKernelGGL(xyz_subsample_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, B, xc, R, step, (rd_bf16_t*)y, out_cs, out_co);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// Global max over the HW pixels of channels [0,C) broadcast into channels [C,2C) of the same NHWC buffer
// (resnet_backbone.py:210-214).  grid = (C/64, B); block = 256 = 32 pixel lanes x 8 lanes of 8 channels.
__global__ __launch_bounds__(256) void global_max_concat_bf16_kernel(rd_bf16_t* __restrict__ buf, int HW, int C, int cs)
{
    __shared__ float s_m[32][64];
    const int b = blockIdx.y, cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cl * 8;
    rd_bf16_t* base = buf + (long long)b * HW * cs;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -FLT_MAX;
    for (int p = pl; p < HW; p += 32) {
        float v[8];
        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(base + (long long)p * cs + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_m[pl][cl * 8 + e] = m[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = -FLT_MAX;
        for (int i = 0; i < 32; ++i) t = fmaxf(t, s_m[i][cl * 8 + e]);
        m[e] = t;
    }
    const rd_u32x4 pk = rd_pack8(m);
    for (int p = pl; p < HW; p += 32) *reinterpret_cast<rd_u32x4*>(base + (long long)p * cs + C + c) = pk;
}

extern "C" int rdpn6d_global_max_concat_bf16(void* buf, int B, int HW, int C, int cs, void* stream)
{
    RD_REQUIRE(buf && B > 0 && HW > 0 && C > 0 && C % 64 == 0 && 2 * C <= cs && cs % 8 == 0, "shape");
    hipLaunchKernelGGL(global_max_concat_bf16_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream,
                       (rd_bf16_t*)buf, HW, C, cs);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// fp32 NHWC channel slice -> compact bf16 NHWC, zero-padded to dst_cs channels (one lane = 8 output channels)
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, int src_cs, int src_co, int C, rd_bf16_t* __restrict__ dst,
                                     int dst_cs, long long npix, int vec)
{
    const int C8 = dst_cs / 8;
    const long long total = npix * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / C8;
        const int c0 = (int)(i - p * C8) * 8;
        const float* sp = src + p * src_cs + src_co + c0;
        float v[8];
        if (vec && c0 + 8 <= C) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? sp[e] : 0.f;
        }
        *reinterpret_cast<rd_u32x4*>(dst + p * dst_cs + c0) = rd_pack8(v);
    }
}

extern "C" int rdpn6d_cast_f32_bf16(const float* src, int src_cs, int src_co, int C, void* dst, int dst_cs, long long npix,
                                    void* stream)
{
    RD_REQUIRE(src && dst && npix > 0 && C > 0 && src_co >= 0 && src_co + C <= src_cs, "source slice");
    RD_REQUIRE(dst_cs >= C && dst_cs % 8 == 0, "dst_cs must be a multiple of 8 >= C");
    const int vec = (src_cs % 4 == 0 && src_co % 4 == 0) ? 1 : 0;
    const long long total = npix * (dst_cs / 8);
    const int blocks = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src_cs, src_co, C,
                       (rd_bf16_t*)dst, dst_cs, npix, vec);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
