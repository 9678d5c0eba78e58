// GPU crop builder: from full frames + detections to the network inputs
// roi_img [B,6,R,R] = RGB/255 | depth back-projected with the crop-adjusted intrinsics, and
// roi_coord_2d [B,5,R/4,R/4] = depth_xyz[:, ::4, ::4] | warped 2D coordinate grid.
// Follows core/gdrn_modeling/data_loader.py:523-627 (test) / :700-836 (train) and core/utils/data_utils.py:81-152:
//   trans = get_affine_transform(center, scale, 0, R)  ==  u' = (R/scale)(u - c) + R/2
//   cv2.warpAffine(img, trans, (R,R), INTER_LINEAR, BORDER_CONSTANT 0)
// cv2's bilinear warp is restated from OpenCV 4.5.5 imgwarp.cpp (WarpAffineInvoker + remapBilinear): the affine map is
// inverted in double, source coordinates are evaluated in 1/1024-pixel fixed point (+ rounding delta 16) and truncated
// to 1/32 pixel; uint8 images use 15-bit integer weights with round-to-nearest, float images float weights.
// PARITY UNPINNED: cv2 (opencv-python 4.5.5.62) is third-party and not installed here; the executable specification
// is oracle/crop_oracle.py.  This removes the 15.7 GB/s host->device stream a 10 k crops/s feed would otherwise need.
#include "common.h"

#pragma clang fp contract(off)

struct CropAffine {  // inverse map (dst -> src) of one ROI, as cv2.warpAffine builds it
    double m[6];
};

__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }  // saturate_cast<int>(double) = lrint

// fixed-point source coordinate of destination pixel (x,y): returns (sx, sy) integer part and (ax, ay) in 1/32
__device__ __forceinline__ void warp_coord(const CropAffine& A, int x, int y, int& sx, int& sy, int& ax, int& ay)
{
    const int AB_SCALE = 1024, round_delta = 16;
    const int adelta = cv_round(A.m[0] * x * AB_SCALE), bdelta = cv_round(A.m[3] * x * AB_SCALE);
    const int X0 = cv_round((A.m[1] * y + A.m[2]) * AB_SCALE) + round_delta;
    const int Y0 = cv_round((A.m[4] * y + A.m[5]) * AB_SCALE) + round_delta;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    sx = X >> 5; sy = Y >> 5; ax = X & 31; ay = Y & 31;
}

// grid = (R*R/256, B)
__global__ __launch_bounds__(256) void crop_builder_kernel(const unsigned char* __restrict__ images /* [N,H,W,3] */,
                                                           const float* __restrict__ depths /* [N,H,W] */, int H, int W,
                                                           const int* __restrict__ img_idx, const CropAffine* __restrict__ inv_in,
                                                           const CropAffine* __restrict__ inv_out, const double* __restrict__ Knew /* [B,4] fx fy cx cy of A@K */,
                                                           const double* __restrict__ ratio /* [B] out_res/scale */, int R,
                                                           float* __restrict__ roi_img, float* __restrict__ roi_coord_2d)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= R * R) return;
    const int y = p / R, x = p - y * R;
    const int n = img_idx[b];
    const unsigned char* img = images + (long long)n * H * W * 3;
    const float* dep = depths + (long long)n * H * W;
    int sx, sy, ax, ay;
    warp_coord(inv_in[b], x, y, sx, sy, ax, ay);
    // 15-bit integer weights for uint8 (exact: (32-a)(32-b)*32), float weights for float images
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    const float fx = (float)ax * (1.f / 32.f), fy = (float)ay * (1.f / 32.f);
    const float f00 = (1.f - fy) * (1.f - fx), f01 = (1.f - fy) * fx, f10 = fy * (1.f - fx), f11 = fy * fx;
    const bool x0 = (unsigned)sx < (unsigned)W, x1 = (unsigned)(sx + 1) < (unsigned)W;
    const bool y0 = (unsigned)sy < (unsigned)H, y1 = (unsigned)(sy + 1) < (unsigned)H;
    float* oimg = roi_img + (long long)b * 6 * R * R;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v00 = (x0 && y0) ? img[((long long)sy * W + sx) * 3 + c] : 0, v01 = (x1 && y0) ? img[((long long)sy * W + sx + 1) * 3 + c] : 0;
        const int v10 = (x0 && y1) ? img[((long long)(sy + 1) * W + sx) * 3 + c] : 0, v11 = (x1 && y1) ? img[((long long)(sy + 1) * W + sx + 1) * 3 + c] : 0;
        const int u8 = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
        oimg[(long long)c * R * R + p] = (float)((double)u8 / 255.0);  // (image - 0) / 255.0 in float64, cast at the end
    }
    const float d00 = (x0 && y0) ? dep[(long long)sy * W + sx] : 0.f, d01 = (x1 && y0) ? dep[(long long)sy * W + sx + 1] : 0.f;
    const float d10 = (x0 && y1) ? dep[(long long)(sy + 1) * W + sx] : 0.f, d11 = (x1 && y1) ? dep[(long long)(sy + 1) * W + sx + 1] : 0.f;
    float dz = d00 * f00;
    dz = dz + d01 * f01;
    dz = dz + d10 * f10;
    dz = dz + d11 * f11;
    // depth / resize_ratio (float64), cast to float32 (pt2), back-projection in float64, final float32 cast
    const float pt2 = (float)((double)dz / ratio[b]);
    const double* K = Knew + b * 4;
    const float X3 = (float)(((double)(float)x - K[2]) * (double)pt2 / K[0]);
    const float Y3 = (float)(((double)(float)y - K[3]) * (double)pt2 / K[1]);
    oimg[(long long)3 * R * R + p] = X3;
    oimg[(long long)4 * R * R + p] = Y3;
    oimg[(long long)5 * R * R + p] = pt2;
    // output-resolution side input: every 4th pixel carries depth_xyz[::4, ::4] + the warped coordinate grid
    if ((x & 3) == 0 && (y & 3) == 0) {
        const int Ro = R / 4, xo = x >> 2, yo = y >> 2, po = yo * Ro + xo;
        float* oc = roi_coord_2d + (long long)b * 5 * Ro * Ro;
        oc[po] = X3; oc[Ro * Ro + po] = Y3; oc[2 * Ro * Ro + po] = pt2;
        int tx, ty, bx, by;
        warp_coord(inv_out[b], xo, yo, tx, ty, bx, by);
        const float gx = (float)bx * (1.f / 32.f), gy = (float)by * (1.f / 32.f);
        const float g00 = (1.f - gy) * (1.f - gx), g01 = (1.f - gy) * gx, g10 = gy * (1.f - gx), g11 = gy * gx;
        const bool a0 = (unsigned)tx < (unsigned)W, a1 = (unsigned)(tx + 1) < (unsigned)W;
        const bool c0 = (unsigned)ty < (unsigned)H, c1 = (unsigned)(ty + 1) < (unsigned)H;
        // coord_2d = meshgrid(linspace(0,1,W), linspace(0,1,H)) in float32
        const double stepx = 1.0 / (double)(W - 1), stepy = 1.0 / (double)(H - 1);
        const float cx0 = (float)(tx * stepx), cx1 = (float)((tx + 1) * stepx), cy0 = (float)(ty * stepy), cy1 = (float)((ty + 1) * stepy);
        float u = ((a0 && c0) ? cx0 : 0.f) * g00;
        u = u + ((a1 && c0) ? cx1 : 0.f) * g01;
        u = u + ((a0 && c1) ? cx0 : 0.f) * g10;
        u = u + ((a1 && c1) ? cx1 : 0.f) * g11;
        float v = ((a0 && c0) ? cy0 : 0.f) * g00;
        v = v + ((a1 && c0) ? cy0 : 0.f) * g01;
        v = v + ((a0 && c1) ? cy1 : 0.f) * g10;
        v = v + ((a1 && c1) ? cy1 : 0.f) * g11;
        oc[3 * Ro * Ro + po] = u;
        oc[4 * Ro * Ro + po] = v;
    }
}

// images [N,H,W,3] uint8 (cfg.INPUT.FORMAT order), depths [N,H,W] f32 (metres); per ROI: image index, inverse affine maps for
// the R and R/4 crops (6 doubles each, as cv2.warpAffine derives them), fx fy cx cy of (A @ K), resize_ratio.
// The host side (rdpn6d_amd/crop.py) derives those per-ROI scalars exactly like the loader does.
extern "C" int rdpn6d_crop_builder_f32(const unsigned char* images, const float* depths, int N, int H, int W, const int* img_idx,
                                       const double* inv_in, const double* inv_out, const double* Knew, const double* ratio,
                                       int B, int R, float* roi_img, float* roi_coord_2d, void* stream)
{
    RD_REQUIRE(images && depths && img_idx && inv_in && inv_out && Knew && ratio && roi_img && roi_coord_2d, "null pointer");
    RD_REQUIRE(N > 0 && H > 1 && W > 1 && B > 0 && R > 0 && R % 4 == 0, "shape");
    hipLaunchKernelGGL(crop_builder_kernel, dim3((R * R + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, images, depths, H, W, img_idx,
                       (const CropAffine*)inv_in, (const CropAffine*)inv_out, Knew, ratio, R, roi_img, roi_coord_2d);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
