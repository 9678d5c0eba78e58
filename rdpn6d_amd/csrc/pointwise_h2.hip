// The kernels between the two-plane fp16 ("h2") convolutions of the point-wise fusion branch (resnet_backbone.py:303-340,
// md_pointnet :39-54), reading and writing h2 tensors ([pixels][C/32][2][32] fp16 holding 16*a = hi + lo, see
// conv_igemm_h2.hip) so that no fp32 copy of those activations is ever made:
//   upsample_bilinear_h2   UpsamplingBilinear2d(scale_factor=f), align_corners=True (:280), fp32 interpolation of the values
//   xyz_subsample_h2       nearest 1/step subsample of the crop's depth-xyz channels into one 32-channel group [x y z 0 ...]
//   global_max_concat_h2   max over the pixels of channels [0,C), broadcast into channels [C,2C) (:51-52)
#include "common.h"
#include "h2_format.h"
#include <float.h>
#include <cstdlib>

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
                                            int out_cs, int out_co, int relu, int* __restrict__ overflow_flag)
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
        float os[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = hy * (hx * v[0][e] + lx * v[1][e]) + ly * (hx * v[2][e] + lx * v[3][e]);  // as upsample_bilinear_kernel
            if (relu) o = fmaxf(o, 0.f);
            os[e] = o * H2_SCALE;
        }
        over |= rd_h2_split8(os, oh, ol);
        _Float16* dp = y + h2_off(((long long)b * Ho + oy) * Wo + ox, out_cs, out_co + c);
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

// The max alone, as an h2 record per crop and channel ([B][C/32][hi32 | lo32]): what the ConvTranspose of the head needs from the
// broadcast half of md_pointnet's concat (:51-52) - a spatially constant input contributes a per-crop constant (rdpn6d_convt3x3s2_
// const_bias_f32), so the [pixels][2C] tensor is never written.  grid = (C/64, B); block 256 = 32 pixel lanes x 8 groups of 8 channels.
__global__ __launch_bounds__(256) void global_max_h2_kernel(const _Float16* __restrict__ buf, int HW, int C, int cs, _Float16* __restrict__ out)
{
    __shared__ float s_m[32][64];
    __shared__ _Float16 s_h[32][64], s_l[32][64];
    const int b = blockIdx.y, q = threadIdx.x & 7, pl = threadIdx.x >> 3, c = blockIdx.x * 64 + q * 8;
    const long long p0 = (long long)b * HW;
    float m[8];
    f16x8 mh, ml;
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -FLT_MAX; mh[e] = (_Float16)0.f; ml[e] = (_Float16)0.f; }
    for (int p = pl; p < HW; p += 32) {
        const _Float16* sp = buf + h2_off(p0 + p, cs, c);
        const f16x8 h = *reinterpret_cast<const f16x8*>(sp), l = *reinterpret_cast<const f16x8*>(sp + 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (float)h[e] + (float)l[e];
            if (v > m[e]) { m[e] = v; mh[e] = h[e]; ml[e] = l[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s_m[pl][q * 8 + e] = m[e]; s_h[pl][q * 8 + e] = mh[e]; s_l[pl][q * 8 + e] = ml[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cl = threadIdx.x;
        float bm = s_m[0][cl];
        _Float16 bh = s_h[0][cl], bl = s_l[0][cl];
        for (int k = 1; k < 32; ++k)
            if (s_m[k][cl] > bm) { bm = s_m[k][cl]; bh = s_h[k][cl]; bl = s_l[k][cl]; }
        _Float16* dp = out + h2_off(b, C, blockIdx.x * 64 + cl);
        dp[0] = bh;
        dp[32] = bl;
    }
}

// ConvTranspose2d(k 3, stride 2, pad 1, output_padding 1) over a spatially CONSTANT input g: output pixel (oy, ox) receives
// sum over its valid taps of W[ky][kx] . g.  Even rows use ky = 1; odd rows ky = 2 (input row i) and ky = 0 (input row i + 1, missing
// for the last output row); the same for columns.  V[b][(ky*3+kx)*F + n] = W[ky][kx][n] . g_b comes from a one-pixel convolution;
// out[phase = py*2+px][b][variant = last_row*2 + last_col][n] = scale[n] * sum of the valid taps - the per-crop bias the four phase
// convolutions add in their epilogues (rdpn6d_conv2d_h2_cb).
__global__ void convt_const_bias_kernel(const float* __restrict__ V, const float* __restrict__ scale, int B, int F, float* __restrict__ out)
{
    const long long total = (long long)16 * B * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % F);
        long long r = i / F;
        const int var = (int)(r % 4);
        r /= 4;
        const int b = (int)(r % B);
        const int ph = (int)(r / B);
        const int py = ph >> 1, px = ph & 1, lr = var >> 1, lc = var & 1;
        const float* v = V + (long long)b * 9 * F + n;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool yok = py == 0 ? ky == 1 : (ky == 2 || (ky == 0 && !lr));
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool xok = px == 0 ? kx == 1 : (kx == 2 || (kx == 0 && !lc));
                if (yok && xok) acc += v[(ky * 3 + kx) * F];
            }
        }
        out[i] = (scale ? scale[n] : 1.f) * acc;
    }
}

}  // namespace

extern "C" int rdpn6d_global_max_h2(const void* x_h2, int B, int HW, int C, int cs, void* gmax_h2, void* stream)
{
    RD_REQUIRE(x_h2 && gmax_h2 && B > 0 && HW > 0 && C > 0 && C % 64 == 0 && C <= cs && cs % 32 == 0, "shape");
    hipLaunchKernelGGL(global_max_h2_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_h2, HW, C, cs, (_Float16*)gmax_h2);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_convt3x3s2_const_bias_f32(const float* V, const float* scale, int B, int F, float* out, void* stream)
{
    RD_REQUIRE(V && out && B > 0 && F > 0, "shape");
    const long long total = (long long)16 * B * F;
    hipLaunchKernelGGL(convt_const_bias_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, V, scale, B, F, out);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_upsample_bilinear_h2_ex(const void* x, int B, int H, int W, int C, int factor, void* y, int out_cs, int out_co, int relu,
                                              int* overflow_flag, void* stream);
extern "C" int rdpn6d_upsample_bilinear_h2(const void* x, int B, int H, int W, int C, int factor, void* y, int* overflow_flag, void* stream)
{
    return rdpn6d_upsample_bilinear_h2_ex(x, B, H, W, C, factor, y, C, 0, 0, overflow_flag, stream);
}

// the same into the channel slice [out_co, out_co + C) of an h2 tensor with out_cs channels per pixel, optionally followed by ReLU
// (a 1x1 convolution + BatchNorm commute with the interpolation - its weights sum to 1 - so "conv at the low resolution, then
// up-sample + ReLU" replaces "up-sample, then conv + ReLU" at 1/16 of the convolution's rows)
extern "C" int rdpn6d_upsample_bilinear_h2_ex(const void* x, int B, int H, int W, int C, int factor, void* y, int out_cs, int out_co, int relu,
                                              int* overflow_flag, void* stream)
{
    RD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && factor >= 1, "shape (C % 32)");
    RD_REQUIRE(out_cs % 32 == 0 && out_co % 32 == 0 && out_co + C <= out_cs, "output slice: whole 32-channel groups");
    const long long total = (long long)B * H * factor * W * factor * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    hipLaunchKernelGGL(upsample_bilinear_h2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, B, H, W, C, factor,
                       (_Float16*)y, out_cs, out_co, relu, overflow_flag);
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

// =====================================================================================================================
// Fused front of the network on the fp16 matrix pipe: conv1 7x7 stride 2 pad 3 (channels 0..2 of the NCHW crop) + folded
// BatchNorm + ReLU + MaxPool2d(3, 2, 1) (resnet_backbone.py:272-275, :321-324), writing the POOLED activation as an h2 tensor -
// the input of the h2 trunk.  Replaces three kernels (VALU stem 0.37 ms, max-pool 0.09 ms, fp32 -> h2 split 0.03 ms at B = 64) and
// the 268 MB fp32 stem activation they passed around.
//
// The stem as an implicit GEMM  [stem pixels] x [K] x [64 channels]  in the two-plane fp16 arithmetic of conv_igemm_h2.hip (hi / lo
// terms of 16 x value, three exact partial products per fp32 product, fp32 accumulation).  The reduction index is laid out as
// k = (c*7 + ky)*8 + kx (kx = 7: zero weight) = 168 -> 192, so that the 8 consecutive k-values an MFMA lane feeds are 8 CONSECUTIVE
// samples of one input row: a fragment is four 4-byte LDS reads of a pre-split patch, no per-element address or conversion work.
// One 256-thread workgroup per 8 x 16 tile of POOLED pixels:
//   1. the 39 x 71 x 3 input patch -> registers -> hi / lo fp16 planes in LDS (zero outside the image);
//   2. a wave owns stem m-tiles (32 consecutive pixels of the 17 x 33 stem tile = the pooled tile's 3x3 windows); per 32-k chunk it
//      fetches its weight fragments (L2-resident h2 tensor [64][6][hi32 | lo32]) and issues 6 MFMAs per m-tile and channel half;
//   3. per 16-channel slice: scale / shift (BatchNorm and the two power-of-two format scales folded) + ReLU -> LDS (overlaying the
//      dead patch); pixels outside the image become 0, which a ReLU output can never lose to (max-pool pads with -inf);
//      3x3 / stride-2 max per pooled pixel, hi / lo split, 16-byte stores of the h2 record.
#ifdef RDPN6D_PROBE
__device__ unsigned long long* g_stem_probe = nullptr;  // probe builds: tools/probe_stem.py
extern "C" int rdpn6d_debug_stem_probe(void* buf)
{
    RD_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stem_probe), &buf, sizeof(buf)));
    return RDPN6D_OK;
}
#define SPT(i) const unsigned long long spt##i = __builtin_readcyclecounter()
#else
#define SPT(i)
#endif

namespace {

constexpr int SP_PH = 8, SP_PW = 16;                        // pooled tile
constexpr int SP_SH = 2 * SP_PH + 1, SP_SW = 2 * SP_PW + 1; // stem tile 17 x 33
constexpr int SP_NPIX = SP_SH * SP_SW;                      // 561
constexpr int SP_MT = (SP_NPIX + 31) / 32;                  // 18 m-tiles
constexpr int SP_IH = 2 * SP_SH + 5, SP_IW = 2 * SP_SW + 5; // input patch 39 x 71
constexpr int SP_IWS = SP_IW + 3;                           // padded row (74 halfs: rows start 4-byte aligned, reads run 1 sample over)
constexpr int SP_PLANE = 3 * SP_IH * SP_IWS + 8;            // halfs per plane (+ slack for the 8-wide reads at the very end)
constexpr int SP_KC = 6;                                    // 32-k chunks (K = 192)
constexpr int SP_TS = 20;                                   // stem-tile row stride in floats: a 16-channel slice + 4
constexpr int SP_LDS_A = 2 * SP_PLANE * 2;
constexpr int SP_LDS_B = SP_MT * 32 * SP_TS * 4;            // one 16-channel slice of the stem tile at a time (all 576 m-tile rows): 45 KiB
constexpr int SP_LDS = SP_LDS_A > SP_LDS_B ? SP_LDS_A : SP_LDS_B;

typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));

// first patch element (relative to the pixel's top-left sample) of the 8-wide k-row rw = k / 8 = c*7 + ky; rows 21..23 are padding
__host__ __device__ constexpr int sp_rowoff(int rw) { return rw >= 21 ? 0 : ((rw / 7) * SP_IH + rw % 7) * SP_IWS; }

// the 39 x 71 x 3 input patch of a tile -> hi / lo fp16 planes in LDS (zero outside the image); the caller synchronises
__device__ __forceinline__ void sp_load_patch(const float* __restrict__ x, const int xc, const int R, const int b, const int iy0, const int ix0,
                                              const int tid, _Float16* s_hi, _Float16* s_lo)
{
    {   // The patch, as ALIGNED 16-byte loads: ix0 = 4*px0 - 5, so columns ix0 - 3 + 4j (j = 0 .. 18) are 16-byte aligned in the crop's
        // rows (R % 4 == 0) and cover patch columns -3 .. 72 - the 71 real ones, two of the three padding columns (read times zero
        // weights only: any finite value does) and three in front that are dropped.  9 loads per thread instead of 33 scalar ones; all
        // of them are issued before the first LDS store (a load-store loop serialises the round trips).
        typedef float sp_f4 __attribute__((ext_vector_type(4)));
        constexpr int F4R = 19, NF4 = 3 * SP_IH * F4R, NP4 = (NF4 + 255) / 256;
        static_assert(4 * F4R - 3 >= SP_IW && 4 * F4R - 3 <= SP_IWS, "19 aligned float4 cover a patch row and stay inside its padding");
        sp_f4 pv[NP4];
#pragma unroll
        for (int t = 0; t < NP4; ++t) {
            const int i = tid + 256 * t;
            const int row = i / F4R, j = i - row * F4R, c = row / SP_IH, py = row - c * SP_IH;
            const int iy = iy0 + py, col = ix0 - 3 + 4 * j;
            pv[t] = sp_f4{0.f, 0.f, 0.f, 0.f};
            if (i < NF4 && (unsigned)iy < (unsigned)R && (unsigned)col < (unsigned)R)
                pv[t] = *reinterpret_cast<const sp_f4*>(x + (((long long)b * xc + c) * R + iy) * R + col);
        }
        for (int i = tid; i < 3 * SP_IH * (SP_IWS - SP_IW) + 8; i += 256) {  // the row padding and the slack: zeros (read, times zero weights)
            const int row = i / (SP_IWS - SP_IW), q = i - row * (SP_IWS - SP_IW);
            const int idx = row < 3 * SP_IH ? row * SP_IWS + SP_IW + q : 3 * SP_IH * SP_IWS + (i - 3 * SP_IH * (SP_IWS - SP_IW));
            s_hi[idx] = (_Float16)0.f;
            s_lo[idx] = (_Float16)0.f;
        }
        __syncthreads();  // (the padding zeros first: the stores below put real samples into two of the three padding columns)
#pragma unroll
        for (int t = 0; t < NP4; ++t) {
            const int i = tid + 256 * t;
            const int row = i / F4R, j = i - row * F4R;
            if (i < NF4) {
                _Float16 h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = pv[t][e] * H2_SCALE;  // image / 255 in [0, 1]: no range issue
                    h[e] = (_Float16)v;
                    l[e] = (_Float16)(v - (float)h[e]);
                }
                const int o = row * SP_IWS + 4 * j - 3;  // patch column 4j - 3 + e; o is odd: (o + 1, o + 2) is a 4-byte aligned pair
                if (j > 0) {
                    s_hi[o] = h[0];
                    s_lo[o] = l[0];
                    typedef _Float16 sp_h2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<sp_h2*>(s_hi + o + 1) = sp_h2{h[1], h[2]};
                    *reinterpret_cast<sp_h2*>(s_lo + o + 1) = sp_h2{l[1], l[2]};
                }
                s_hi[o + 3] = h[3];
                s_lo[o + 3] = l[3];
            }
        }
    }
}

// OUT: 0 = the pooled activation as an h2 tensor; 1 / 2 = as a plain bf16 / fp16 NHWC tensor [B, R/4, R/4, 64] (the 16-bit inference
// mode, cfg.TEST.AMP_TEST: the arithmetic stays the fp32-accurate h2 one, only the stored result is rounded); 3 / 4 = NO pooling and
// no ReLU: scale * conv + shift of the 16 x 32 stem pixels the tile owns as a bf16 / fp16 NHWC tensor [B, R/2, R/2, 64] - the RAW stem
// convolution of the mixed-precision training forward (BatchNorm with batch statistics, ReLU and the max-pool are separate launches
// there; the tile's first row / column, computed for the pooling windows of the inference forms, is simply not stored)
template <int OUT>
__global__ __launch_bounds__(256, 2) void stem_pool_h2_kernel(const float* __restrict__ x, int xc, int R, const _Float16* __restrict__ w_h2,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              _Float16* __restrict__ y, int* __restrict__ overflow_flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    _Float16* s_hi = reinterpret_cast<_Float16*>(sp_smem);
    _Float16* s_lo = s_hi + SP_PLANE;
    float* s_t = reinterpret_cast<float*>(sp_smem);  // stem-tile slice, overlays the patch after the MFMA phase
    const int Rp = R / 4;                            // pooled resolution (stem: R / 2)
    const int b = blockIdx.z, py0 = blockIdx.y * SP_PH, px0 = blockIdx.x * SP_PW;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;  // first stem pixel of the tile
    const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;  // first input sample of the patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SPT(0);

    sp_load_patch(x, xc, R, b, iy0, ix0, tid, s_hi, s_lo);
    __syncthreads();
    SPT(1);

    // ---- MFMA phase.  Wave w owns m-tiles w, w+4, ... (at most 5); lane = (row r, k-half).
    const int r = lane & 31, half = lane >> 5;
    constexpr int MPW = (SP_MT + 3) / 4;
    f32x16 acc[MPW][2];
    int base[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;
        int p = (wave + 4 * m) * 32 + r;
        p = p < SP_NPIX ? p : 0;  // rows past the tile compute pixel 0 again and are never stored
        const int sy = p / SP_SW, sx = p - sy * SP_SW;
        base[m] = 2 * sy * SP_IWS + 2 * sx;  // even: 4-byte aligned in both planes
    }
    // weight fragments of chunk cc: channel row n = nt*32 + r, 16-byte slot 2j + half of its 128-byte record (global, L2-resident)
    const unsigned char* wrow[2] = {reinterpret_cast<const unsigned char*>(w_h2) + (size_t)(r * SP_KC) * 128 + half * 16,
                                    reinterpret_cast<const unsigned char*>(w_h2) + (size_t)((32 + r) * SP_KC) * 128 + half * 16};
    sp_u32x4 fb[2][4], fbn[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nt][j] = *reinterpret_cast<const sp_u32x4*>(wrow[nt] + j * 32);
#pragma unroll
    for (int cc = 0; cc < SP_KC; ++cc) {
        if (cc + 1 < SP_KC) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) fbn[nt][j] = *reinterpret_cast<const sp_u32x4*>(wrow[nt] + (cc + 1) * 128 + j * 32);
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
            if (wave + 4 * m >= SP_MT) continue;  // wave-uniform
            sp_u32x4 ah[2], al[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // k-row of this lane's 8 values: 4*cc + 2*s + half
                const int off = base[m] + (half ? sp_rowoff(4 * cc + 2 * s + 1) : sp_rowoff(4 * cc + 2 * s));
                const unsigned* ph = reinterpret_cast<const unsigned*>(s_hi + off);
                const unsigned* pl = reinterpret_cast<const unsigned*>(s_lo + off);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ah[s][q] = ph[q];
                    al[s][q] = pl[q];
                }
            }
            // lo*hi, hi*lo, hi*hi per k16 step (as conv_igemm_h2.hip), the two channel halves alternating
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[s]), __builtin_bit_cast(f16x8, fb[nt][s]), acc[m][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]), __builtin_bit_cast(f16x8, fb[nt][2 + s]), acc[m][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]), __builtin_bit_cast(f16x8, fb[nt][s]), acc[m][nt], 0, 0, 0);
            }
        }
        if (cc + 1 < SP_KC) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[nt][j] = fbn[nt][j];
        }
    }

    SPT(2);
    // ---- epilogue + pooling, one 16-channel slice of the stem tile at a time (the slice overlays the dead patch).
    // Accumulator element e of lane (r, half) = pixel row (e&3) + 8*(e>>2) + 4*half of the m-tile, channel nt*32 + r.
    bool over = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __syncthreads();  // q = 0: every wave is done with the patch; q > 0: the previous slice has been pooled
        if ((r >> 4) == (q & 1)) {
            // 3 instructions per value (fma, max, ds_write with an immediate offset): the row of accumulator element e is
            // (wave + 4m)*32 + 4*half + (e&3) + 8*(e>>2), so only the first term needs an address.  Pixels outside the image (the tile's
            // first row / column at the top / left edge of the crop) are left as they are and SKIPPED by the pooling below.
            const int nt = q >> 1, n = nt * 32 + r;
            const float sc = scale[n], sh = shift[n];
#pragma unroll
            for (int m = 0; m < MPW; ++m) {
                if (wave + 4 * m >= SP_MT) continue;
                float* tp = s_t + ((wave + 4 * m) * 32 + 4 * half) * SP_TS + (r & 15);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[m][nt][e] * sc + sh;
                    tp[((e & 3) + 8 * (e >> 2)) * SP_TS] = OUT >= 3 ? v : fmaxf(v, 0.f);
                }
            }
        }
        __syncthreads();
        if constexpr (OUT >= 3) {  // raw stem pixels (ty, tx) = (1 .. 16, 1 .. 32) of the tile: one work item = (pixel, 8 channels)
            const int Rs = R / 2;
#pragma unroll
            for (int u = 0; u < (2 * SP_PH) * (2 * SP_PW) * 2 / 256; ++u) {
                const int item = tid + 256 * u;
                const int c8 = (item & 1) * 8, pix = item >> 1, ty = 1 + pix / (2 * SP_PW), tx = 1 + pix % (2 * SP_PW);
                const int sy = sy0 + ty, sx = sx0 + tx;
                if (sy < Rs && sx < Rs) {
                    const float* tp = s_t + (ty * SP_SW + tx) * SP_TS + c8;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(tp), c = *reinterpret_cast<const f32x4*>(tp + 4);
                    const float m8[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
                    sp_u32x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        typedef float sp_f2 __attribute__((ext_vector_type(2)));
                        if constexpr (OUT == 3) {
                            typedef __bf16 sp_b2 __attribute__((ext_vector_type(2)));
                            pk[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f2{m8[2 * e], m8[2 * e + 1]}, sp_b2));
                        } else {
                            typedef _Float16 sp_h2v __attribute__((ext_vector_type(2)));
                            pk[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f2{m8[2 * e], m8[2 * e + 1]}, sp_h2v));
                        }
                    }
                    unsigned short* dp = reinterpret_cast<unsigned short*>(y) + (((long long)b * Rs + sy) * Rs + sx) * 64 + q * 16 + c8;
                    *reinterpret_cast<sp_u32x4*>(dp) = pk;
                }
            }
        } else {   // 3x3 / stride 2 max-pool + h2 record: one work item = (pooled pixel, 8 channels) per thread
            const int c8 = (tid & 1) * 8, pq = tid >> 1, qy = pq / SP_PW, qx = pq - qy * SP_PW;
            const int py = py0 + qy, px = px0 + qx;
            if (py < Rp && px < Rp) {
                float m8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) m8[e] = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        // stem pixel (sy0 + 2qy + dy, sx0 + 2qx + dx): only row / column -1 can lie outside (MaxPool pads with -inf;
                        // leaving the tap out = the 0 a ReLU output can never lose to)
                        if ((dy == 0 && sy0 + 2 * qy < 0) || (dx == 0 && sx0 + 2 * qx < 0)) continue;
                        const float* tp = s_t + ((2 * qy + dy) * SP_SW + 2 * qx + dx) * SP_TS + c8;
                        const f32x4 a = *reinterpret_cast<const f32x4*>(tp), c = *reinterpret_cast<const f32x4*>(tp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {  // (ReLU outputs: no NaN ordering question, one v_max_f32 each)
                            m8[e] = __builtin_fmaxf(m8[e], a[e]);
                            m8[4 + e] = __builtin_fmaxf(m8[4 + e], c[e]);
                        }
                    }
                if constexpr (OUT == 0) {
                    f16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) m8[e] *= H2_SCALE;
                    over |= rd_h2_split8(m8, hi, lo);
                    _Float16* dp = y + h2_off(((long long)b * Rp + py) * Rp + px, 64, q * 16 + c8);
                    *reinterpret_cast<f16x8*>(dp) = hi;
                    *reinterpret_cast<f16x8*>(dp + 32) = lo;
                } else {
                    sp_u32x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (OUT == 1) {
                            typedef __bf16 sp_b2 __attribute__((ext_vector_type(2)));
                            typedef float sp_f2 __attribute__((ext_vector_type(2)));
                            pk[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f2{m8[2 * e], m8[2 * e + 1]}, sp_b2));
                        } else {
                            typedef _Float16 sp_h2v __attribute__((ext_vector_type(2)));
                            typedef float sp_f2 __attribute__((ext_vector_type(2)));
                            pk[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f2{m8[2 * e], m8[2 * e + 1]}, sp_h2v));
                        }
                    }
                    unsigned short* dp = reinterpret_cast<unsigned short*>(y) + (((long long)b * Rp + py) * Rp + px) * 64 + q * 16 + c8;
                    *reinterpret_cast<sp_u32x4*>(dp) = pk;
                }
            }
        }
    }
    if (over && overflow_flag) *overflow_flag = 1;
#ifdef RDPN6D_PROBE
    {
        const unsigned long long spt3 = __builtin_readcyclecounter();
        if (g_stem_probe && lane == 0) {
            unsigned long long* o = g_stem_probe + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 * 4 + wave * 4;
            o[0] = spt1 - spt0;
            o[1] = spt2 - spt1;
            o[2] = spt3 - spt2;
            o[3] = 1;
        }
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the pooled forms (OUT 0 / 1 / 2) with the 3x3 / stride-2 max taken IN REGISTERS.  Probe of the kernel above: 7 600 cycles of
// patch load, 19 000 of MFMA phase, 21 300 of epilogue per workgroup - the epilogue wrote every accumulator to LDS (160 ds_write per
// lane, four 16-channel slices with two barriers each) and read each value back ~2.25 times for the pooling.  Here the m-tiles follow
// the pooling structure instead of the row-major pixel order:
//     m-tile t = 0 .. 16 : stem ROW t of the 17 x 33 tile, columns 0 .. 31  (MFMA row i = column i)
//     m-tile 17          : stem COLUMN 32, rows 0 .. 16                      (MFMA row i = row i; rows 17 .. 31 idle)
//   - still 18 m-tiles.  Wave w owns rows 4w .. 4w + 3 (wave 3 also row 16, wave 0 the column tile).  An accumulator register of lane
//   (r, half) holds channel nt*32 + r at columns 8g + 4*half + j (e = 4g + j): the horizontal window of pooled column q is
//       q even: columns 4(q/2) .. + 2        = three registers of ONE lane
//       q odd : two registers of one lane + the first register of the NEXT group of four, which the lane 32 away holds (one
//               cross-half exchange per group), or the column tile's value for q = 15 (through LDS: 17 x 64 floats)
//   and the vertical window of pooled row 2w + {0, 1} is rows 4w .. 4w + 2 / 4w + 2 .. 4w + 4 = the wave's own m-tiles, except row
//   4w + 4, whose horizontally pooled values the next wave publishes through LDS (3 x 16 x 64 floats).  The pooled 8 x 16 x 64 tile is
//   then staged in LDS as finished records and leaves in 16-byte pieces.  LDS traffic of the epilogue: 20 KB instead of ~480 KB.
// The conv values are those of the kernel above bit for bit (same k order per element; the all-padding second k16 step of the last
// chunk - zero weights - is skipped), max is exact: identical outputs (tests/test_gpu_h2.py).
constexpr int SP2_ROWB0 = 256 + 32, SP2_ROWB1 = 128 + 32;      // staged output row: h2 record / 16-bit row, + 32 bytes against bank conflicts
constexpr int SP2_OUT_BYTES = SP_PH * SP_PW * SP2_ROWB0;       // 36 864 >= the patch planes
constexpr int SP2_COL = SP2_OUT_BYTES > SP_LDS_A ? SP2_OUT_BYTES : SP_LDS_A;
constexpr int SP2_ROW = SP2_COL + SP_SH * 64 * 4;
constexpr int SP2_LDS = SP2_ROW + 3 * 16 * 64 * 4;

template <int OUT>
__global__ __launch_bounds__(256, 2) void stem_pool_h2_v2_kernel(const float* __restrict__ x, int xc, int R, const _Float16* __restrict__ w_h2,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 _Float16* __restrict__ y, int* __restrict__ overflow_flag)
{
    static_assert(OUT >= 0 && OUT <= 2, "pooled forms only");
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    _Float16* s_hi = reinterpret_cast<_Float16*>(sp_smem);
    _Float16* s_lo = s_hi + SP_PLANE;
    float* s_col = reinterpret_cast<float*>(sp_smem + SP2_COL);
    float* s_row = reinterpret_cast<float*>(sp_smem + SP2_ROW);
    const int Rp = R / 4;
    const int b = blockIdx.z, py0 = blockIdx.y * SP_PH, px0 = blockIdx.x * SP_PW;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    SPT(0);
    sp_load_patch(x, xc, R, b, iy0, ix0, tid, s_hi, s_lo);
    __syncthreads();
    SPT(1);

    // ---- MFMA phase
    const int r = lane & 31, half = lane >> 5;
    constexpr int MPW = 5;
    const bool has5 = wave == 0 || wave == 3;  // wave-uniform
    f32x16 acc[MPW][2];
    int base[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;
        int sy = 4 * wave + m, sx = r;
        if (m == 4) {
            sy = wave == 3 ? 16 : (r < SP_SH ? r : 0);
            sx = wave == 3 ? r : 32;
        }
        base[m] = 2 * sy * SP_IWS + 2 * sx;
    }
    const unsigned char* wrow[2] = {reinterpret_cast<const unsigned char*>(w_h2) + (size_t)(r * SP_KC) * 128 + half * 16,
                                    reinterpret_cast<const unsigned char*>(w_h2) + (size_t)((32 + r) * SP_KC) * 128 + half * 16};
    sp_u32x4 fb[2][4], fbn[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nt][j] = *reinterpret_cast<const sp_u32x4*>(wrow[nt] + j * 32);
#pragma unroll
    for (int cc = 0; cc < SP_KC; ++cc) {
        if (cc + 1 < SP_KC) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) fbn[nt][j] = *reinterpret_cast<const sp_u32x4*>(wrow[nt] + (cc + 1) * 128 + j * 32);
        }
        // k-rows 4*cc + 2*s + half; rows 21 .. 23 are padding (zero weights): the second k16 step of the last chunk is all padding
        constexpr int NS_LAST = 1;
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
            if (m == 4 && !has5) continue;
            sp_u32x4 ah[2], al[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (cc == SP_KC - 1 && s >= NS_LAST) continue;
                const int off = base[m] + (half ? sp_rowoff(4 * cc + 2 * s + 1) : sp_rowoff(4 * cc + 2 * s));
                const unsigned* ph = reinterpret_cast<const unsigned*>(s_hi + off);
                const unsigned* pl = reinterpret_cast<const unsigned*>(s_lo + off);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ah[s][q] = ph[q];
                    al[s][q] = pl[q];
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (cc == SP_KC - 1 && s >= NS_LAST) continue;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[s]), __builtin_bit_cast(f16x8, fb[nt][s]), acc[m][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]), __builtin_bit_cast(f16x8, fb[nt][2 + s]), acc[m][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s]), __builtin_bit_cast(f16x8, fb[nt][s]), acc[m][nt], 0, 0, 0);
            }
        }
        if (cc + 1 < SP_KC) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[nt][j] = fbn[nt][j];
        }
    }

    SPT(2);
    // ---- BatchNorm + ReLU in place.  Stem pixels outside the image (row / column -1 at the top / left edge of the crop) become 0, which a
    // ReLU output can never lose to (MaxPool2d pads with -inf).
    const bool top = sy0 < 0, left = sx0 < 0;  // block-uniform
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float sc = scale[nt * 32 + r], sh = shift[nt * 32 + r];
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
            if (m == 4 && !has5) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][nt][e] = fmaxf(acc[m][nt][e] * sc + sh, 0.f);
            const bool row_tile = m < 4 || wave == 3;
            if (row_tile && left && half == 0) acc[m][nt][0] = 0.f;               // column 0 of a row tile
            if (m == 0 && wave == 0 && top) {                                     // row 0
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][nt][e] = 0.f;
            }
            if (m == 4 && wave == 0 && top && half == 0) acc[m][nt][0] = 0.f;     // (row 0, column 32) of the column tile
        }
    }
    // the column tile's values for the q = 15 windows of every row
    if (wave == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ty = (e & 3) + 8 * (e >> 2) + 4 * half;
                if (ty < SP_SH) s_col[ty * 64 + nt * 32 + r] = acc[4][nt][e];
            }
    }
    __syncthreads();

    // ---- horizontal 3-max per row tile: hp[2g + u] = pooled column 4g + 2*half + u
    auto hpool = [&](const f32x16& v, const float xcol, float(&hp)[8]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            hp[2 * g] = fmaxf(fmaxf(v[4 * g], v[4 * g + 1]), v[4 * g + 2]);
            const float send = half ? v[4 * g] : v[4 * ((g + 1) & 3)];
            float recv = __shfl_xor(send, 32, 64);
            if (g == 3) recv = half ? xcol : recv;  // (lower half: slot 3 is unused)
            hp[2 * g + 1] = fmaxf(fmaxf(v[4 * g + 2], v[4 * g + 3]), recv);
        }
    };
    float hp[MPW][2][8];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
        if (m == 4 && wave != 3) continue;
        const int t = m == 4 ? 16 : 4 * wave + m;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) hpool(acc[m][nt], s_col[t * 64 + nt * 32 + r], hp[m][nt]);
    }
    if (wave >= 1) {  // row 4*wave is the last row of the previous wave's second pooled row
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 8; ++k) s_row[((wave - 1) * 16 + nt * 8 + k) * 64 + lane] = hp[0][nt][k];
    }
    __syncthreads();

    // ---- vertical 3-max, the finished records into LDS (over the dead patch), out in 16-byte pieces
    constexpr int ROWB = OUT == 0 ? SP2_ROWB0 : SP2_ROWB1;
    bool over = false;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float p0[8], p1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            p0[k] = fmaxf(fmaxf(hp[0][nt][k], hp[1][nt][k]), hp[2][nt][k]);
            const float xr = wave == 3 ? hp[4][nt][k] : s_row[(wave * 16 + nt * 8 + k) * 64 + lane];
            p1[k] = fmaxf(fmaxf(hp[2][nt][k], hp[3][nt][k]), xr);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float(&pv)[8] = q ? p1 : p0;
            unsigned char* dst = sp_smem + (size_t)((2 * wave + q) * SP_PW + 2 * half) * ROWB;  // pooled pixel (2*wave + q, 4g + 2*half + u)
            if constexpr (OUT == 0) {
                f16x8 hi, lo;
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] *= H2_SCALE;
                over |= rd_h2_split8(pv, hi, lo);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    _Float16* d = reinterpret_cast<_Float16*>(dst + (size_t)(4 * (k >> 1) + (k & 1)) * ROWB + nt * 128) + r;
                    d[0] = hi[k];
                    d[32] = lo[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    unsigned short* d = reinterpret_cast<unsigned short*>(dst + (size_t)(4 * (k >> 1) + (k & 1)) * ROWB) + nt * 32 + r;
                    typedef float sp_f2 __attribute__((ext_vector_type(2)));
                    if constexpr (OUT == 1) {
                        typedef __bf16 sp_b2 __attribute__((ext_vector_type(2)));
                        *d = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(sp_f2{pv[k], 0.f}, sp_b2)) & 0xffffu);
                    } else {
                        typedef _Float16 sp_h2v __attribute__((ext_vector_type(2)));
                        *d = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(sp_f2{pv[k], 0.f}, sp_h2v)) & 0xffffu);
                    }
                }
            }
        }
    }
    __syncthreads();
    constexpr int PPP = OUT == 0 ? 16 : 8;  // 16-byte pieces per pooled pixel
#pragma unroll
    for (int i = 0; i < SP_PH * SP_PW * PPP / 256; ++i) {
        const int piece = tid + 256 * i, pix = piece / PPP, j = piece - pix * PPP;
        const int py = py0 + pix / SP_PW, px = px0 + pix % SP_PW;
        if (py < Rp && px < Rp) {
            const sp_u32x4 v = *reinterpret_cast<const sp_u32x4*>(sp_smem + (size_t)pix * ROWB + j * 16);
            unsigned char* dp = reinterpret_cast<unsigned char*>(y) + ((((long long)b * Rp + py) * Rp + px) * PPP + j) * 16;
            *reinterpret_cast<sp_u32x4*>(dp) = v;
        }
    }
    if (over && overflow_flag) *overflow_flag = 1;
#ifdef RDPN6D_PROBE
    {
        const unsigned long long spt3 = __builtin_readcyclecounter();
        if (g_stem_probe && lane == 0) {
            unsigned long long* o = g_stem_probe + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 * 4 + wave * 4;
            o[0] = spt1 - spt0;
            o[1] = spt2 - spt1;
            o[2] = spt3 - spt2;
            o[3] = 1;
        }
    }
#endif
}

}  // namespace

// x [B, xc, R, R] fp32 NCHW (channels 0..2 used); w_h2: conv1 weights with the reduction index k = (c*7 + ky)*8 + kx padded to
// 192 (zeros at kx = 7 and k >= 168) as an h2 tensor [64][6][2][32] fp16 holding w * 2^sw(n) (gdrn.pack_stem_h2_weight);
// scale / shift [64]: folded BatchNorm times 2^-(sw(n)+4) / plain shift.  y: pooled activation [B, R/4, R/4, 64] as an h2 tensor.
extern "C" int rdpn6d_stem_pool_h2_ex(const float* x, int B, int xc, int R, const void* w_h2, const float* scale, const float* shift, void* y,
                                      int out_fmt, int* overflow_flag, void* stream)
{
    RD_REQUIRE(x && w_h2 && scale && shift && y, "null pointer");
    RD_REQUIRE(B > 0 && xc >= 3 && R > 0 && R % 4 == 0, "shape (R % 4)");
    const bool force_v1 = (out_fmt & 0x100) != 0;  // tests: + 0x100 = the round-3 kernel (pooling through LDS) for the pooled forms
    out_fmt &= 0xff;
    RD_REQUIRE(out_fmt >= 0 && out_fmt <= 4, "out_fmt: 0 = h2 tensor, 1 = bf16 NHWC, 2 = fp16 NHWC (pooled); 3 = bf16, 4 = fp16 raw stem output");
    RD_REQUIRE((reinterpret_cast<size_t>(x) & 15) == 0, "x must be 16-byte aligned (the patch is read with aligned 16-byte loads)");
    const int Rp = R / 4;
    dim3 grid(rd_cdiv(Rp, SP_PW), rd_cdiv(Rp, SP_PH), B);
    hipStream_t s = (hipStream_t)stream;
    static const bool v1 = getenv("RDPN6D_STEM_V1") != nullptr;  // A/B runs: the round-3 kernel (pooling through LDS)
    if (out_fmt <= 2 && !v1 && !force_v1) {  // pooled forms: the 3x3 / stride-2 max in registers (stem_pool_h2_v2_kernel)
        if (out_fmt == 0) {
            RD_LDS_OPT_IN(stem_pool_h2_v2_kernel<0>, SP2_LDS);
            hipLaunchKernelGGL(stem_pool_h2_v2_kernel<0>, grid, dim3(256), SP2_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
        } else if (out_fmt == 1) {
            RD_LDS_OPT_IN(stem_pool_h2_v2_kernel<1>, SP2_LDS);
            hipLaunchKernelGGL(stem_pool_h2_v2_kernel<1>, grid, dim3(256), SP2_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
        } else {
            RD_LDS_OPT_IN(stem_pool_h2_v2_kernel<2>, SP2_LDS);
            hipLaunchKernelGGL(stem_pool_h2_v2_kernel<2>, grid, dim3(256), SP2_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
        }
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    if (out_fmt == 0) {
        RD_LDS_OPT_IN(stem_pool_h2_kernel<0>, SP_LDS);
        hipLaunchKernelGGL(stem_pool_h2_kernel<0>, grid, dim3(256), SP_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
    } else if (out_fmt == 1) {
        RD_LDS_OPT_IN(stem_pool_h2_kernel<1>, SP_LDS);
        hipLaunchKernelGGL(stem_pool_h2_kernel<1>, grid, dim3(256), SP_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
    } else if (out_fmt == 2) {
        RD_LDS_OPT_IN(stem_pool_h2_kernel<2>, SP_LDS);
        hipLaunchKernelGGL(stem_pool_h2_kernel<2>, grid, dim3(256), SP_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
    } else if (out_fmt == 3) {
        RD_LDS_OPT_IN(stem_pool_h2_kernel<3>, SP_LDS);
        hipLaunchKernelGGL(stem_pool_h2_kernel<3>, grid, dim3(256), SP_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
    } else {
        RD_LDS_OPT_IN(stem_pool_h2_kernel<4>, SP_LDS);
        hipLaunchKernelGGL(stem_pool_h2_kernel<4>, grid, dim3(256), SP_LDS, s, x, xc, R, (const _Float16*)w_h2, scale, shift, (_Float16*)y, overflow_flag);
    }
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// conv1 weights OIHW [64][3][7][7] fp32 -> the h2 tensor of the stem kernels + the per-channel factor 2^-sw(n) / 16 (what
// gdrn.pack_stem_h2_weight computes with torch ops at plan-build time), as ONE launch: the training step re-packs after every
// optimizer step.  One wavefront per output channel.
__global__ __launch_bounds__(64) void stem_pack_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ out, float* __restrict__ inv)
{
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* wn = w + n * 147;
    float mx = 0.f;
    for (int i = lane; i < 147; i += 64) mx = fmaxf(mx, fabsf(wn[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    // 2^sw brings the largest weight into [2^13, 2^14): sw = floor(13 - log2(mx))
    int ex = 0;
    float sw = 1.f;
    if (mx > 0.f) {
        const float m = frexpf(mx, &ex);  // mx = m * 2^ex, m in [0.5, 1)
        sw = ldexpf(1.f, (m == 0.5f ? 14 : 13) - ex);
    }
    for (int k = lane; k < 192; k += 64) {
        const int row = k >> 3, kx = k & 7;  // row = c * 7 + ky
        const float v = (row < 21 && kx < 7) ? wn[row * 7 + kx] * sw : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        _Float16* o = out + ((size_t)(n * SP_KC + (k >> 5)) * 2) * 32 + (k & 31);
        o[0] = hi;
        o[32] = lo;
    }
    if (lane == 0) inv[n] = 1.f / (sw * H2_SCALE);
}

extern "C" int rdpn6d_stem_pack_h2(const float* w_oihw, void* w_h2, float* inv_scale, void* stream)
{
    RD_REQUIRE(w_oihw && w_h2 && inv_scale, "null pointer");
    hipLaunchKernelGGL(stem_pack_h2_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, w_oihw, (_Float16*)w_h2, inv_scale);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_stem_pool_h2(const float* x, int B, int xc, int R, const void* w_h2, const float* scale, const float* shift, void* y,
                                   int* overflow_flag, void* stream)
{
    return rdpn6d_stem_pool_h2_ex(x, B, xc, R, w_h2, scale, shift, y, 0, overflow_flag, stream);
}
