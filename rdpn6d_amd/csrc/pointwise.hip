// Bandwidth-bound kernels of the RDPN6D forward path (everything that is not a GEMM).
// All activations are NHWC fp32; loads/stores are 16 bytes per lane wherever the layout allows.
#include "common.h"
#include "h2_format.h"
#include <float.h>

// ------------------------------------------------------------------------------------------------
// Stem: conv 7x7 stride 2 pad 3 on channels 0..2 of the NCHW crop + folded BN + ReLU -> NHWC 64.
// (resnet_backbone.py:272,:321-323).  K = 147 is too ragged/small for the MFMA path and the layer
// is 0.7 % of the FLOPs: a direct LDS-tiled VALU kernel.  The input patch and the 147x64 weights sit in
// LDS; weights are wave-uniform LDS broadcasts, 16 B per read.
// OUT_BF16: the 64 output channels are rounded (RNE) to bf16 - the input of the bf16 trunk (conv_igemm_bf16.hip).
// Workgroup = 16 x 32 output pixels, two pixels (16 columns apart) per thread: every 16-byte weight read from LDS
// feeds 8 FMAs instead of 4 - the kernel is bound by the LDS weight broadcasts, not by the FMA rate.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void stem_conv7x7_kernel(const float* __restrict__ x, int xc, int R,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, void* __restrict__ yv, int relu)
{
    constexpr int TY = 16, TX = 32, PY = 2 * TY + 5, PX = 2 * TX + 5;  // 37 x 69 input patch
    __shared__ float s_in[3][PY][PX + 1];
    __shared__ __attribute__((aligned(16))) float s_w[147 * 64];  // [tap(ky,kx,c)][64]
    const int Ro = R / 2;
    const int b = blockIdx.z, ty0 = blockIdx.y * TY, tx0 = blockIdx.x * TX;
    const int tid = threadIdx.x;
    // weights: global layout [64][7][7][3] -> LDS [147][64]
    for (int i = tid; i < 147 * 64; i += 256) {
        const int n = i / 147, k = i - n * 147;
        s_w[k * 64 + n] = w[i];
    }
    const int iy0 = ty0 * 2 - 3, ix0 = tx0 * 2 - 3;
    for (int i = tid; i < 3 * PY * PX; i += 256) {
        const int c = i / (PY * PX), rem = i - c * PY * PX, py = rem / PX, px = rem - py * PX;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)R && (unsigned)ix < (unsigned)R)
            v = x[(((long long)b * xc + c) * R + iy) * R + ix];
        s_in[c][py][px] = v;
    }
    __syncthreads();
    const int ly = tid / 16, lx = tid - ly * 16;
    const int oy = ty0 + ly;
    // two channels per instruction (v_pk_fma_f32): the same IEEE fma per channel in the same order
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[2][32];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int n = 0; n < 32; ++n) acc2[p][n] = f32x2{0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v0 = s_in[c][ly * 2 + ky][lx * 2 + kx];
                const float v1 = s_in[c][ly * 2 + ky][(lx + 16) * 2 + kx];
                const f32x2 vv0 = {v0, v0}, vv1 = {v1, v1};
                const f32x4* wp = reinterpret_cast<const f32x4*>(&s_w[((ky * 7 + kx) * 3 + c) * 64]);
#pragma unroll
                for (int n4 = 0; n4 < 16; ++n4) {
                    const f32x4 wv = wp[n4];
                    const f32x2 wlo = {wv[0], wv[1]}, whi = {wv[2], wv[3]};
                    acc2[0][n4 * 2 + 0] = __builtin_elementwise_fma(vv0, wlo, acc2[0][n4 * 2 + 0]);
                    acc2[0][n4 * 2 + 1] = __builtin_elementwise_fma(vv0, whi, acc2[0][n4 * 2 + 1]);
                    acc2[1][n4 * 2 + 0] = __builtin_elementwise_fma(vv1, wlo, acc2[1][n4 * 2 + 0]);
                    acc2[1][n4 * 2 + 1] = __builtin_elementwise_fma(vv1, whi, acc2[1][n4 * 2 + 1]);
                }
            }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int ox = tx0 + lx + 16 * p;
        if (oy < Ro && ox < Ro) {
            const long long pix = ((long long)b * Ro + oy) * Ro + ox;
#pragma unroll
            for (int n4 = 0; n4 < 16; ++n4) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n4 * 4 + e;
                    const float a = acc2[p][n >> 1][n & 1];
                    const float v = scale ? a * scale[n] + shift[n] : a;
                    o[e] = (v > 0.f || !relu) ? v : 0.f;
                }
                if constexpr (OUT_BF16) {
                    uint2 pk;
                    pk.x = (unsigned)rd_f2bf(o[0]) | ((unsigned)rd_f2bf(o[1]) << 16);
                    pk.y = (unsigned)rd_f2bf(o[2]) | ((unsigned)rd_f2bf(o[3]) << 16);
                    reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yv) + pix * 64)[n4] = pk;
                } else {
                    reinterpret_cast<f32x4*>(reinterpret_cast<float*>(yv) + pix * 64)[n4] = o;
                }
            }
        }
    }
}

static int stem_launch(const float* x, int B, int xc, int R, const float* w, const float* scale, const float* shift,
                       void* y, int relu, void* stream, bool out_bf16 = false)
{
    RD_REQUIRE(x && w && y && (!scale == !shift), "null pointer");
    RD_REQUIRE(B > 0 && xc >= 3 && R > 0 && R % 2 == 0, "shape");
    const int Ro = R / 2;
    dim3 grid(rd_cdiv(Ro, 32), rd_cdiv(Ro, 16), B);
    if (out_bf16) hipLaunchKernelGGL(stem_conv7x7_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, xc, R, w, scale, shift, y, relu);
    else hipLaunchKernelGGL(stem_conv7x7_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, xc, R, w, scale, shift, y, relu);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_stem_conv7x7_bf16(const float* x, int B, int xc, int R, const float* w, const float* scale,
                                        const float* shift, void* y, void* stream)
{
    RD_REQUIRE(scale && shift, "null pointer");
    return stem_launch(x, B, xc, R, w, scale, shift, y, 1, stream, true);
}

extern "C" int rdpn6d_stem_conv7x7_f32(const float* x, int B, int xc, int R, const float* w, const float* scale,
                                       const float* shift, float* y, void* stream)
{
    RD_REQUIRE(scale && shift, "null pointer");
    return stem_launch(x, B, xc, R, w, scale, shift, y, 1, stream);
}

// training form: raw convolution output (no folded BN, no ReLU)
extern "C" int rdpn6d_stem_conv7x7_raw_f32(const float* x, int B, int xc, int R, const float* w, float* y, void* stream)
{
    return stem_launch(x, B, xc, R, w, nullptr, nullptr, y, 0, stream);
}
extern "C" int rdpn6d_stem_conv7x7_raw_bf16(const float* x, int B, int xc, int R, const float* w, void* y, void* stream)
{
    return stem_launch(x, B, xc, R, w, nullptr, nullptr, y, 0, stream, true);
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, pad 1): one thread per (output pixel, 4 channels).
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, int B, int H, int W, int C, float* __restrict__ y)
{
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, C4 = C / 4;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long p = i / C4;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        f32x4 m = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long long)b * H + iy) * W + ix) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *reinterpret_cast<f32x4*>(y + (((long long)b * Ho + oy) * Wo + ox) * C + c4 * 4) = m;
    }
}

extern "C" int rdpn6d_maxpool3x3s2_f32(const float* x, int B, int H, int W, int C, float* y, void* stream)
{
    RD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C, y);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// ------------------------------------------------------------------------------------------------
// Bilinear upsampling, align_corners=True (nn.UpsamplingBilinear2d).
__global__ void upsample_bilinear_kernel(const float* __restrict__ x, int B, int H, int W, int C, int f,
                                         float* __restrict__ y)
{
    const int Ho = H * f, Wo = W * f, C4 = C / 4;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long p = i / C4;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* base = x + (long long)b * H * W * C + c4 * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * W + x0) * C);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * W + x1) * C);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * W + x0) * C);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * W + x1) * C);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        *reinterpret_cast<f32x4*>(y + (((long long)b * Ho + oy) * Wo + ox) * C + c4 * 4) = o;
    }
}

extern "C" int rdpn6d_upsample_bilinear_f32(const float* x, int B, int H, int W, int C, int factor, float* y,
                                            void* stream)
{
    RD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && factor >= 1, "shape");
    const long long total = (long long)B * H * factor * W * factor * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C,
                       factor, y);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// ------------------------------------------------------------------------------------------------
// Nearest-neighbour subsample of the depth-xyz channels (3..5) of the NCHW crop into an NHWC slice.
__global__ void xyz_subsample_kernel(const float* __restrict__ x, int B, int xc, int R, int step, float* __restrict__ y,
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
            y[i * out_cs + out_co + c] = x[(((long long)b * xc + 3 + c) * R + oy * step) * R + ox * step];
    }
}

extern "C" int rdpn6d_xyz_subsample_f32(const float* x, int B, int xc, int R, int step, float* y, int out_cs,
                                        int out_co, void* stream)
{
    RD_REQUIRE(x && y && B > 0 && xc >= 6 && R > 0 && step > 0 && R % step == 0, "shape");
    RD_REQUIRE(out_co + 3 <= out_cs, "output slice");
    const long long total = (long long)B * (R / step) * (R / step);
    hipLaunchKernelGGL(xyz_subsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, B, xc, R, step, y, out_cs, out_co);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// ------------------------------------------------------------------------------------------------
// Global max over the HW pixels of channels [0,C) and broadcast into channels [C,2C) of the same
// NHWC buffer (md_pointnet's adaptive_max_pool2d + adaptive_avg_pool2d-broadcast + cat).
// grid = (C/64, B); block = 1024 = 16 pixel lanes x 64 channels (256-byte coalesced rows).
__global__ __launch_bounds__(1024) void global_max_concat_kernel(float* __restrict__ buf, int HW, int C, int cs)
{
    __shared__ float s_m[16][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    float* base = buf + (long long)b * HW * cs;
    float m = -FLT_MAX;
    for (int p = pl; p < HW; p += 16) {
        const float v = base[(long long)p * cs + c];
        m = v > m ? v : m;
    }
    s_m[pl][threadIdx.x & 63] = m;
    __syncthreads();
    const int cl = threadIdx.x & 63;
    m = s_m[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, s_m[k][cl]);
    for (int p = pl; p < HW; p += 16) base[(long long)p * cs + C + c] = m;
}

extern "C" int rdpn6d_global_max_concat_f32(float* buf, int B, int HW, int C, int cs, void* stream)
{
    RD_REQUIRE(buf && B > 0 && HW > 0 && C > 0 && C % 64 == 0 && 2 * C <= cs, "shape");
    hipLaunchKernelGGL(global_max_concat_kernel, dim3(C / 64, B), dim3(1024), 0, (hipStream_t)stream, buf, HW, C, cs);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(G groups of C/G = 4 channels) + ReLU, in place, one workgroup per sample.
// Two passes (mean, then centred variance) with wave-shuffle + LDS reductions.
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

#define GN_THREADS 1024  // one workgroup per sample: 1024 lanes keep a single crop (per-image inference) short
// H2 = false: in place on the fp32 tensor.  H2 = true: x stays as it is, the result goes out as an h2 tensor [B*HW][C/32][hi x 32 | lo x 32]
// fp16 holding 16 * value (conv_igemm_h2.hip) - the input format of the next ConvPnPNet layer on the fp16 matrix pipe.
// RES: the thread's pixels (at most RES of them) stay in REGISTERS between the three passes - one read of the tensor instead of three;
// the sums are formed in the same order as the re-reading form (RES = 0, any map size), so both give identical bits.
template <bool H2, int RES>
__global__ __launch_bounds__(GN_THREADS) void groupnorm4_relu_kernel(float* __restrict__ x, int HW, int C,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, _Float16* __restrict__ y_h2,
                                                                    int* __restrict__ overflow_flag)
{
    // The groups are independent: workgroup (crop blockIdx.x, part blockIdx.y of gridDim.y) owns G = (C/4) / gridDim.y of them, thread t
    // its group g = t % G and pixel lane pl = t / G.  (One workgroup per crop - gridDim.y = 1 - left 64 crops on a quarter of the CUs with
    // 32 pixels per thread and pass: 34 us for the 32x32 map; four parts: a full chip, 8 pixels per thread.)
    const int G = C / 4 / (int)gridDim.y;
    const int PL = GN_THREADS / G;
    __shared__ float s_part[GN_THREADS];
    __shared__ float s_mean[64], s_rstd[64];
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    float* base = x + (long long)blockIdx.x * HW * C + ((int)blockIdx.y * G + g) * 4;
    gamma += (int)blockIdx.y * G * 4;
    beta += (int)blockIdx.y * G * 4;
    f32x4 keep[RES > 0 ? RES : 1];
    if constexpr (RES > 0) {
#pragma unroll
        for (int j = 0; j < RES; ++j) {
            const int p = pl + j * PL;
            keep[j] = p < HW ? *reinterpret_cast<const f32x4*>(base + (long long)p * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float s = 0.f;
    if constexpr (RES > 0) {
#pragma unroll
        for (int j = 0; j < RES; ++j)
            if (pl + j * PL < HW) s += (keep[j][0] + keep[j][1]) + (keep[j][2] + keep[j][3]);
    } else {
        for (int p = pl; p < HW; p += PL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * C);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    // block sum per group in a FIXED order: lanes of a wave that share a group (G <= 32 divides 64: lane % G) by xor-shuffles, then the
    // sixteen wave partials by one thread per group (a serial walk over all PL partials was 128 dependent LDS reads - 4.5 us per phase
    // with the whole workgroup waiting)
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    for (int o = 32; o >= G; o >>= 1) s += __shfl_xor(s, o);
    if (ln < G) s_part[wv * G + ln] = s;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int i = 0; i < GN_THREADS / 64; ++i) t += s_part[i * G + threadIdx.x];
        s_mean[threadIdx.x] = t / (float)(HW * 4);
    }
    __syncthreads();
    const float mean = s_mean[g];
    float q = 0.f;
    if constexpr (RES > 0) {
#pragma unroll
        for (int j = 0; j < RES; ++j)
            if (pl + j * PL < HW) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = keep[j][e] - mean;
                    q += dlt * dlt;
                }
            }
    } else {
        for (int p = pl; p < HW; p += PL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = v[e] - mean;
                q += dlt * dlt;
            }
        }
    }
    for (int o = 32; o >= G; o >>= 1) q += __shfl_xor(q, o);
    __syncthreads();
    if (ln < G) s_part[wv * G + ln] = q;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int i = 0; i < GN_THREADS / 64; ++i) t += s_part[i * G + threadIdx.x];
        s_rstd[threadIdx.x] = 1.0f / sqrtf(t / (float)(HW * 4) + 1e-5f);
    }
    __syncthreads();
    const float rstd = s_rstd[g];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g * 4);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + g * 4);
    bool over = false;
    auto emit = [&](const int p, f32x4 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float o = (v[e] - mean) * rstd * ga[e] + be[e];
            v[e] = o > 0.f ? o : 0.f;
        }
        if constexpr (H2) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sc = v[e] * 16.f;
                over |= !(fabsf(sc) <= 65504.f);
                sc = fminf(fmaxf(sc, -65504.f), 65504.f);
                hi[e] = (_Float16)sc;
                lo[e] = (_Float16)(sc - (float)hi[e]);
            }
            const int c = ((int)blockIdx.y * G + g) * 4;
            _Float16* qq = y_h2 + ((long long)blockIdx.x * HW + p) * (2 * (long long)C) + (c >> 5) * 64 + (c & 31);
            *reinterpret_cast<f16x4*>(qq) = hi;
            *reinterpret_cast<f16x4*>(qq + 32) = lo;
        } else {
            *reinterpret_cast<f32x4*>(base + (long long)p * C) = v;
        }
    };
    if constexpr (RES > 0) {
#pragma unroll
        for (int j = 0; j < RES; ++j)
            if (pl + j * PL < HW) emit(pl + j * PL, keep[j]);
    } else {
        for (int p = pl; p < HW; p += PL) emit(p, *reinterpret_cast<const f32x4*>(base + (long long)p * C));
    }
    if (H2 && over && overflow_flag) *overflow_flag = 1;
}

template <bool H2>
static void gn_launch(float* x, int B, int HW, int C, int G, const float* gamma, const float* beta, _Float16* y_h2, int* flag, hipStream_t s)
{
    const int parts = (G % 4 == 0 && HW >= 256) ? 4 : 1;  // (small maps: one workgroup per crop is already short)
    const int per_thread = (HW + GN_THREADS / (G / parts) - 1) / (GN_THREADS / (G / parts));  // pixels a thread owns
    if (per_thread <= 2)
        hipLaunchKernelGGL((groupnorm4_relu_kernel<H2, 2>), dim3(B, parts), dim3(GN_THREADS), 0, s, x, HW, C, gamma, beta, y_h2, flag);
    else if (per_thread <= 8)
        hipLaunchKernelGGL((groupnorm4_relu_kernel<H2, 8>), dim3(B, parts), dim3(GN_THREADS), 0, s, x, HW, C, gamma, beta, y_h2, flag);
    else if (per_thread <= 16)
        hipLaunchKernelGGL((groupnorm4_relu_kernel<H2, 16>), dim3(B, parts), dim3(GN_THREADS), 0, s, x, HW, C, gamma, beta, y_h2, flag);
    else
        hipLaunchKernelGGL((groupnorm4_relu_kernel<H2, 0>), dim3(B, parts), dim3(GN_THREADS), 0, s, x, HW, C, gamma, beta, y_h2, flag);
}

extern "C" int rdpn6d_groupnorm_relu_f32(float* x, int B, int HW, int C, int G, const float* gamma, const float* beta,
                                         void* stream)
{
    RD_REQUIRE(x && gamma && beta && B > 0 && HW > 0, "null/shape");
    RD_REQUIRE(C == 4 * G && G <= 64 && 256 % G == 0, "only C/G == 4 with G | 256 is implemented (GroupNorm(32,128))");
    gn_launch<false>(x, B, HW, C, G, gamma, beta, nullptr, nullptr, (hipStream_t)stream);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_groupnorm_relu_h2(const float* x, int B, int HW, int C, int G, const float* gamma, const float* beta, void* y_h2,
                                        int* overflow_flag, void* stream)
{
    RD_REQUIRE(x && gamma && beta && y_h2 && B > 0 && HW > 0, "null/shape");
    RD_REQUIRE(C == 4 * G && G <= 64 && 256 % G == 0 && C % 32 == 0, "only C/G == 4 with G | 256 and C % 32 == 0 is implemented (GroupNorm(32,128))");
    gn_launch<true>(const_cast<float*>(x), B, HW, C, G, gamma, beta, (_Float16*)y_h2, overflow_flag, (hipStream_t)stream);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// ------------------------------------------------------------------------------------------------
// Dense-map glue.  Pass 1 (only for mask attention): per-sample min / max of the mask channel.
__global__ __launch_bounds__(256) void mask_minmax_kernel(const float* __restrict__ head, int head_cs, int HW,
                                                          float* __restrict__ minmax)
{
    __shared__ float s_mn[4], s_mx[4];
    const float* base = head + (long long)blockIdx.x * HW * head_cs;
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int p = threadIdx.x; p < HW; p += 256) {
        const float v = base[(long long)p * head_cs];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        minmax[blockIdx.x * 2 + 0] = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        minmax[blockIdx.x * 2 + 1] = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    }
}

// Pass 2: one thread per pixel.  Reads the head's NHWC row, writes (a) the NCHW maps the reference
// API returns and (b) the ConvPnPNet input row [xyz | coord2d | anchor | softmax(region[1:]) | pad].
// The arg-max is taken ON the softmax output with first-max tie-break, exactly like
// GDRN.py:206-209 (two different logits can round to the same probability).
// H2: the ConvPnPNet input row goes out as an h2 record (pnp_cs % 32 == 0 channels: [hi x 32 | lo x 32] fp16 per 32-channel group,
// 16 * value; conv_igemm_h2.hip) instead of fp32, and a value outside the format's range (|v| > 4094, inf, NaN - the inputs are
// caller data: depth xyz, anchors) raises the plan's range flag like every other h2 writer.
template <int KMAX, bool H2 = false, int MC = 1>
__global__ __launch_bounds__(256) void dense_glue_kernel(const float* __restrict__ head, int head_cs,
                                                         const float* __restrict__ coord2d,
                                                         const float* __restrict__ fps, int B, int HW, int K,
                                                         int mask_attention, const float* __restrict__ minmax,
                                                         float* __restrict__ out_nchw, float* __restrict__ pnp_in,
                                                         int pnp_cs, int* __restrict__ argmax_out, int* __restrict__ overflow_flag)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * HW) return;
    const int b = (int)(i / HW), p = (int)(i - (long long)b * HW);
    const float* h = head + i * head_cs;
    // MC = mask channels of the head row: 1 (MASK_LOSS_TYPE L1 | BCE) or 2 (CE: [mask0 mask1 | x y z | region bg + K])
    const int C = MC + 3 + K + 1;
    float v[MC + 3 + KMAX + 1];
    // head row -> registers (16-byte loads), and out to the NCHW API tensor (coalesced over pixels)
#pragma unroll
    for (int c4 = 0; c4 < (MC + 3 + KMAX + 1 + 3) / 4; ++c4) {
        if (c4 * 4 < C) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(h + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c4 * 4 + e < MC + 3 + KMAX + 1) v[c4 * 4 + e] = t[e];
        }
    }
    float* o = out_nchw + (long long)b * C * HW + p;
#pragma unroll
    for (int c = 0; c < MC + 3 + KMAX + 1; ++c)
        if (c < C) o[(long long)c * HW] = v[c];
    // softmax over region[1..K]
    float mx = -FLT_MAX;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) mx = fmaxf(mx, v[MC + 4 + k]);
    float sum = 0.f;
    float e[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            e[k] = expf(v[MC + 4 + k] - mx);
            sum += e[k];
        }
    int am = 0;
    float best = -1.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            e[k] = e[k] / sum;
            if (e[k] > best) { best = e[k]; am = k; }
        }
    if (argmax_out) argmax_out[i] = am;
    float att = 1.f;
    if (mask_attention == 1) {  // MASK_LOSS_TYPE L1: per-crop min-max
        const float mn = minmax[b * 2], mxm = minmax[b * 2 + 1];
        att = (v[0] - mn) / (mxm - mn);  // no epsilon, as model_utils.py:34
    } else if (mask_attention == 2) {  // MASK_LOSS_TYPE BCE: torch.sigmoid (model_utils.py:35-37)
        att = 1.f / (1.f + expf(-v[0]));
    }
    const float* cd = coord2d + (long long)b * 5 * HW + p;
    const float* an = fps + ((long long)b * K + am) * 3;
    float row[11];
    row[0] = v[MC]; row[1] = v[MC + 1]; row[2] = v[MC + 2];
#pragma unroll
    for (int c = 0; c < 5; ++c) row[3 + c] = cd[(long long)c * HW];
    row[8] = an[0]; row[9] = an[1]; row[10] = an[2];
    if constexpr (H2) {
        // The row (pnp_cs x 4 bytes per pixel) is STAGED through LDS and leaves the workgroup as one contiguous block of 256 rows in 16-byte
        // pieces, consecutive lanes on consecutive addresses: a thread writing its own 256-byte row put every store instruction of a wave
        // on 64 different rows (the kernel ran 71 us against 38 for the fp32 row).  LDS row stride = row bytes + 8: 8-byte accesses, two
        // lanes per bank.  (The host launches this variant only when B * HW is a multiple of 256: no partial workgroup.)
        extern __shared__ __attribute__((aligned(16))) unsigned char glue_smem[];
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const int rowb = pnp_cs * 4, rows = rowb + 8;
        unsigned char* mine = glue_smem + (size_t)threadIdx.x * rows;
        bool over = false;
#pragma unroll
        for (int c8 = 0; c8 < (11 + KMAX + 31) / 32 * 4; ++c8) {  // groups of 8 channels; whole 32-channel groups (zero padded)
            if (c8 * 8 < pnp_cs) {
                float sv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = c8 * 8 + u;
                    float val = 0.f;
                    if (ch < 11) val = row[ch < 11 ? ch : 0];
                    else if (ch - 11 < KMAX) val = (ch - 11 < K) ? e[ch - 11 < KMAX ? ch - 11 : 0] : 0.f;
                    sv[u] = val * att * 16.f;
                }
                rd_h8 hi, lo;
                over |= rd_h2_split8(sv, hi, lo);
                unsigned char* dst = mine + (c8 >> 2) * 128 + (c8 & 3) * 16;
                *reinterpret_cast<f16x4*>(dst) = f16x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<f16x4*>(dst + 8) = f16x4{hi[4], hi[5], hi[6], hi[7]};
                *reinterpret_cast<f16x4*>(dst + 64) = f16x4{lo[0], lo[1], lo[2], lo[3]};
                *reinterpret_cast<f16x4*>(dst + 72) = f16x4{lo[4], lo[5], lo[6], lo[7]};
            }
        }
        const f16x4 z = {};
        for (int c = (11 + KMAX + 31) / 32 * 32; c < pnp_cs; c += 4) {
            unsigned char* dst = mine + (c >> 5) * 128 + (c & 31) * 2;
            *reinterpret_cast<f16x4*>(dst) = z;
            *reinterpret_cast<f16x4*>(dst + 64) = z;
        }
        if (over && overflow_flag) *overflow_flag = 1;
        __syncthreads();
        unsigned char* gout = reinterpret_cast<unsigned char*>(pnp_in) + (size_t)blockIdx.x * 256 * rowb;
        const int ppr = rowb / 16, pieces = 256 * ppr;  // 16-byte pieces per row / per workgroup
        for (int q = threadIdx.x; q < pieces; q += 256) {
            const int px = q / ppr, j = q - px * ppr;
            const unsigned char* src = glue_smem + (size_t)px * rows + j * 16;
            const f16x4 a0 = *reinterpret_cast<const f16x4*>(src), a1 = *reinterpret_cast<const f16x4*>(src + 8);
            *reinterpret_cast<rd_h8*>(gout + (size_t)q * 16) = rd_h8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        }
        return;
    }
    float* q = pnp_in + i * pnp_cs;
    // 11 + K channels, then zero pad up to pnp_cs (all register indices are compile-time constants)
#pragma unroll
    for (int c4 = 0; c4 < (11 + KMAX + 3) / 4 + 1; ++c4) {
        if (c4 * 4 < pnp_cs) {
            f32x4 t;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = c4 * 4 + u;
                float val = 0.f;
                if (ch < 11) val = row[ch < 11 ? ch : 0];
                else if (ch - 11 < KMAX) val = (ch - 11 < K) ? e[ch - 11 < KMAX ? ch - 11 : 0] : 0.f;
                t[u] = val * att;
            }
            *reinterpret_cast<f32x4*>(q + c4 * 4) = t;
        }
    }
    for (int c = ((11 + KMAX + 3) / 4 + 1) * 4; c < pnp_cs; c += 4) *reinterpret_cast<f32x4*>(q + c) = f32x4{0.f, 0.f, 0.f, 0.f};
}

static int glue_mask_args(int mask_attention, int mask_type, int* att_mode, int* mc)
{
    // mask_type = ROT_HEAD.MASK_LOSS_TYPE as get_mask_prob reads it (models/model_utils.py:24-42): 0 L1 | 1 BCE | 2 CE
    RD_REQUIRE(mask_type >= 0 && mask_type <= 2, "mask_type: 0 L1 | 1 BCE | 2 CE");
    // CE: the reference's own branch cannot run (torch.softmax(pred_mask, dim=1, keepdim=True) is a TypeError, model_utils.py:39)
    RD_REQUIRE(!(mask_attention && mask_type == 2), "MASK_ATTENTION with MASK_LOSS_TYPE CE: the reference's get_mask_prob raises there");
    *att_mode = mask_attention ? (mask_type == 1 ? 2 : 1) : 0;
    *mc = mask_type == 2 ? 2 : 1;
    return RDPN6D_OK;
}

extern "C" int rdpn6d_dense_glue_mt_f32(const float* head, int head_cs, const float* coord2d, const float* fps, int B,
                                        int HW, int K, int mask_attention, int mask_type, float* minmax_scratch, float* out_nchw,
                                        float* pnp_in, int pnp_cs, int* argmax_out, void* stream)
{
    RD_REQUIRE(head && coord2d && fps && out_nchw && pnp_in, "null pointer");
    RD_REQUIRE(B > 0 && HW > 0 && K >= 2 && K <= 64, "K in 2..64");
    int att = 0, mc = 1;
    if (int rc = glue_mask_args(mask_attention, mask_type, &att, &mc)) return rc;
    RD_REQUIRE(head_cs % 4 == 0 && head_cs >= mc + 4 + K && pnp_cs % 4 == 0 && pnp_cs >= 11 + K, "channel strides");
    RD_REQUIRE(att != 1 || minmax_scratch, "mask attention needs a [B,2] scratch");
    hipStream_t s = (hipStream_t)stream;
    if (att == 1) {
        hipLaunchKernelGGL(mask_minmax_kernel, dim3(B), dim3(256), 0, s, head, head_cs, HW, minmax_scratch);
        RD_LAUNCH_CHECK();
    }
    const unsigned blocks = (unsigned)(((long long)B * HW + 255) / 256);
#define RD_GLUE(KM, MCV) hipLaunchKernelGGL((dense_glue_kernel<KM, false, MCV>), dim3(blocks), dim3(256), 0, s, head, head_cs, coord2d, fps, \
                                            B, HW, K, att, minmax_scratch, out_nchw, pnp_in, pnp_cs, argmax_out, (int*)nullptr)
    if (K <= 32) { if (mc == 1) RD_GLUE(32, 1); else RD_GLUE(32, 2); }
    else { if (mc == 1) RD_GLUE(64, 1); else RD_GLUE(64, 2); }
#undef RD_GLUE
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_dense_glue_f32(const float* head, int head_cs, const float* coord2d, const float* fps, int B,
                                     int HW, int K, int mask_attention, float* minmax_scratch, float* out_nchw,
                                     float* pnp_in, int pnp_cs, int* argmax_out, void* stream)
{
    return rdpn6d_dense_glue_mt_f32(head, head_cs, coord2d, fps, B, HW, K, mask_attention, 0, minmax_scratch, out_nchw, pnp_in, pnp_cs,
                                    argmax_out, stream);
}

template <int KM, int MCV>
static int glue_h2_launch(unsigned blocks, size_t smem, hipStream_t s, const float* head, int head_cs, const float* coord2d,
                           const float* fps, int B, int HW, int K, int att, float* minmax_scratch, float* out_nchw, void* pnp_in_h2,
                           int pnp_cs, int* argmax_out, int* overflow_flag)
{
    auto kern = dense_glue_kernel<KM, true, MCV>;
    RD_LDS_OPT_IN(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, s, head, head_cs, coord2d, fps, B, HW, K, att, minmax_scratch, out_nchw,
                       reinterpret_cast<float*>(pnp_in_h2), pnp_cs, argmax_out, overflow_flag);
    return RDPN6D_OK;
}

extern "C" int rdpn6d_dense_glue_mt_h2(const float* head, int head_cs, const float* coord2d, const float* fps, int B, int HW, int K,
                                       int mask_attention, int mask_type, float* minmax_scratch, float* out_nchw, void* pnp_in_h2,
                                       int pnp_cs, int* argmax_out, int* overflow_flag, void* stream)
{
    RD_REQUIRE(head && coord2d && fps && out_nchw && pnp_in_h2, "null pointer");
    RD_REQUIRE(B > 0 && HW > 0 && K >= 2 && K <= 64, "K in 2..64");
    int att = 0, mc = 1;
    if (int rc = glue_mask_args(mask_attention, mask_type, &att, &mc)) return rc;
    RD_REQUIRE(head_cs % 4 == 0 && head_cs >= mc + 4 + K && pnp_cs % 32 == 0 && pnp_cs >= 11 + K, "channel strides (h2 row: pnp_cs % 32 == 0)");
    RD_REQUIRE(att != 1 || minmax_scratch, "mask attention needs a [B,2] scratch");
    hipStream_t s = (hipStream_t)stream;
    if (att == 1) {
        hipLaunchKernelGGL(mask_minmax_kernel, dim3(B), dim3(256), 0, s, head, head_cs, HW, minmax_scratch);
        RD_LAUNCH_CHECK();
    }
    RD_REQUIRE(((long long)B * HW) % 256 == 0, "the h2 glue kernel stages whole workgroups of 256 pixels: B * HW % 256 == 0");
    const unsigned blocks = (unsigned)(((long long)B * HW) / 256);
    const size_t smem = (size_t)256 * (pnp_cs * 4 + 8);
    int rc;
#define RD_GLUE_H2(KM, MCV) rc = glue_h2_launch<KM, MCV>(blocks, smem, s, head, head_cs, coord2d, fps, B, HW, K, att, minmax_scratch, out_nchw, \
                                                         pnp_in_h2, pnp_cs, argmax_out, overflow_flag)
    if (K <= 32) { if (mc == 1) RD_GLUE_H2(32, 1); else RD_GLUE_H2(32, 2); }
    else { if (mc == 1) RD_GLUE_H2(64, 1); else RD_GLUE_H2(64, 2); }
#undef RD_GLUE_H2
    if (rc != RDPN6D_OK) return rc;
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_dense_glue_h2(const float* head, int head_cs, const float* coord2d, const float* fps, int B, int HW, int K,
                                    int mask_attention, float* minmax_scratch, float* out_nchw, void* pnp_in_h2, int pnp_cs,
                                    int* argmax_out, int* overflow_flag, void* stream)
{
    return rdpn6d_dense_glue_mt_h2(head, head_cs, coord2d, fps, B, HW, K, mask_attention, 0, minmax_scratch, out_nchw, pnp_in_h2, pnp_cs,
                                   argmax_out, overflow_flag, stream);
}

// ------------------------------------------------------------------------------------------------
// Pose decode: rot6d -> R_allo, SITE translation, allocentric -> egocentric.  One thread per crop.
__global__ void pose_decode_kernel(const float* __restrict__ rt, int rt_stride, const float* __restrict__ cams,
                                   const float* __restrict__ centers, const float* __restrict__ whs,
                                   const float* __restrict__ ratios, int B, int is_allo, int train_variant,
                                   float* __restrict__ rot, float* __restrict__ trans)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* p = rt + (long long)b * rt_stride;
    // ortho6d_to_mat_batch: x = norm(a1); z = norm(x x a2); y = z x x; columns [x y z]
    float x[3] = {p[0], p[1], p[2]}, a2[3] = {p[3], p[4], p[5]};
    float n = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    n = fmaxf(n, 1e-12f);
    x[0] /= n; x[1] /= n; x[2] /= n;
    float z[3] = {x[1] * a2[2] - x[2] * a2[1], x[2] * a2[0] - x[0] * a2[2], x[0] * a2[1] - x[1] * a2[0]};
    n = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    n = fmaxf(n, 1e-12f);
    z[0] /= n; z[1] /= n; z[2] /= n;
    const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    float Ra[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
    // SITE: c = delta * wh + centre ; z = z_rel * resize_ratio ; back-project
    const float* K = cams + b * 9;
    const float cx = p[6] * whs[b * 2 + 0] + centers[b * 2 + 0];
    const float cy = p[7] * whs[b * 2 + 1] + centers[b * 2 + 1];
    const float tz = p[8] * ratios[b];
    const float t[3] = {tz * (cx - K[2]) / K[0], tz * (cy - K[5]) / K[4], tz};
    trans[b * 3 + 0] = t[0]; trans[b * 3 + 1] = t[1]; trans[b * 3 + 2] = t[2];
    float* R = rot + b * 9;
    if (!is_allo) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Ra[i];
        return;
    }
    if (!train_variant) {
        // numpy path (utils.py:39-94): fp32 ray, fp64 axis-angle, fp64 product, fp32 result
        const float tn = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        const float ray[3] = {t[0] / tn, t[1] / tn, t[2] / tn};
        const double angle = acos((double)ray[2]);
        if (angle > 0.0) {
            double ax = -(double)ray[1], ay = (double)ray[0], az = 0.0;  // cross((0,0,1), ray)
            const double an = sqrt(ax * ax + ay * ay + az * az);
            ax /= an; ay /= an; az /= an;
            const double c = cos(angle), s = sin(angle), C = 1.0 - c;
            const double xs = ax * s, ys = ay * s, zs = az * s, xC = ax * C, yC = ay * C, zC = az * C;
            const double xyC = ax * yC, yzC = ay * zC, zxC = az * xC;
            const double M[9] = {ax * xC + c, xyC - zs, zxC + ys, xyC + zs, ay * yC + c, yzC - xs,
                                 zxC - ys, yzC + xs, az * zC + c};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    R[i * 3 + j] = (float)(M[i * 3 + 0] * (double)Ra[0 * 3 + j] + M[i * 3 + 1] * (double)Ra[1 * 3 + j] +
                                           M[i * 3 + 2] * (double)Ra[2 * 3 + j]);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = Ra[i];
        }
    } else {
        // torch path (utils.py:208-236): fp32 quaternion, eps = 1e-4 in both norms
        const float eps = 1e-4f;
        const float tn = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + eps;
        const float ray[3] = {t[0] / tn, t[1] / tn, t[2] / tn};
        const float angle = acosf(ray[2]);
        float ax = -ray[1], ay = ray[0], az = 0.f;
        const float an = sqrtf(ax * ax + ay * ay + az * az) + eps;
        ax /= an; ay /= an; az /= an;
        const float sh = sinf(angle * 0.5f);
        float qw = cosf(angle * 0.5f), qx = ax * sh, qy = ay * sh, qz = az * sh;
        const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        qw /= qn; qx /= qn; qy /= qn; qz /= qn;
        const float X = qx * 2.f, Y = qy * 2.f, Z = qz * 2.f;
        const float wX = qw * X, wY = qw * Y, wZ = qw * Z, xX = qx * X, xY = qx * Y, xZ = qx * Z, yY = qy * Y,
                    yZ = qy * Z, zZ = qz * Z;
        const float M[9] = {1.f - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.f - (xX + zZ), yZ - wX,
                            xZ - wY, yZ + wX, 1.f - (xX + yY)};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[i * 3 + j] = M[i * 3 + 0] * Ra[0 * 3 + j] + M[i * 3 + 1] * Ra[1 * 3 + j] + M[i * 3 + 2] * Ra[2 * 3 + j];
    }
}

extern "C" int rdpn6d_pose_decode_f32(const float* rt, int rt_stride, const float* roi_cams, const float* roi_centers,
                                      const float* roi_whs, const float* resize_ratios, int B, int is_allo,
                                      int train_variant, float* rot, float* trans, void* stream)
{
    RD_REQUIRE(rt && roi_cams && roi_centers && roi_whs && resize_ratios && rot && trans, "null pointer");
    RD_REQUIRE(B > 0 && rt_stride >= 9, "shape");
    hipLaunchKernelGGL(pose_decode_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, rt, rt_stride,
                       roi_cams, roi_centers, roi_whs, resize_ratios, B, is_allo, train_variant, rot, trans);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
