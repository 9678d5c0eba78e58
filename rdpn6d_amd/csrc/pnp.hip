// 2D-3D correspondence selection (row A8) for the PnP solve (gfx950).
//
// select_correspondences_kernel restates, operation by operation in fp32, what the reference does on the host with numpy
// before it calls cv2.solvePnPRansac (paths relative to the reference root):
//   core/gdrn_modeling/engine_utils.py:118-136   get_out_mask, L1 type: (m - min) / (max - min) per crop, no epsilon
//   core/gdrn_modeling/gdrn_evaluator.py:105-121 xyz = (c - 0.5) * extent;  uv = coord2d * (W, H);
//                                                keep mask > thr  &  |xyz_c| > 1e-4 * extent_c  for all three axes;
//                                                boolean-mask gather = ROW-MAJOR pixel order
// One 256-thread workgroup per crop; the gather is an order-preserving compaction (wave ballot + prefix over the waves).
// Bit-exact against tests/golden/select_golden.npz (outputs of the reference's own functions); this file is compiled with
// floating-point contraction off.
#include "common.h"
#include <float.h>

#pragma clang fp contract(off)

#define SEL_THREADS 256
#define SEL_WAVES (SEL_THREADS / 64)

__global__ __launch_bounds__(SEL_THREADS) void select_correspondences_kernel(
    const float* __restrict__ out_nchw, int C, const float* __restrict__ coord2d, int C2, int u_ch, int v_ch,
    const float* __restrict__ extents, const int* __restrict__ im_hw, int im_H, int im_W, int HW, float mask_thr,
    float* __restrict__ image_points, float* __restrict__ model_points, int* __restrict__ counts,
    unsigned char* __restrict__ sel_mask, float* __restrict__ out_mask)
{
    __shared__ float s_mn[SEL_WAVES], s_mx[SEL_WAVES];
    __shared__ int s_cnt[SEL_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* m = out_nchw + (size_t)b * C * HW;
    const float* cu = coord2d + ((size_t)b * C2 + u_ch) * HW;
    const float* cv = coord2d + ((size_t)b * C2 + v_ch) * HW;
    const float ex = extents[b * 3 + 0], ey = extents[b * 3 + 1], ez = extents[b * 3 + 2];
    const float fw = (float)(im_hw ? im_hw[2 * b + 1] : im_W), fh = (float)(im_hw ? im_hw[2 * b] : im_H);
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int p = tid; p < HW; p += SEL_THREADS) { const float v = m[p]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    mn = s_mn[0]; mx = s_mx[0];
#pragma unroll
    for (int wv = 1; wv < SEL_WAVES; wv++) { mn = fminf(mn, s_mn[wv]); mx = fmaxf(mx, s_mx[wv]); }
    const float range = mx - mn;
    const float tx = 0.0001f * ex, ty = 0.0001f * ey, tz = 0.0001f * ez;
    float* ip = image_points + (size_t)b * HW * 2;
    float* mp = model_points + (size_t)b * HW * 3;
    int n = 0;
    for (int base = 0; base < HW; base += SEL_THREADS) {
        const int p = base + tid;
        bool sel = false;
        float x = 0.f, y = 0.f, z = 0.f;
        if (p < HW) {
            const float nm = (m[p] - mn) / range;  // 0/0 = NaN for a constant mask: every comparison below is then false
            if (out_mask) out_mask[(size_t)b * HW + p] = nm;
            x = (m[HW + p] - 0.5f) * ex;
            y = (m[2 * HW + p] - 0.5f) * ey;
            z = (m[3 * HW + p] - 0.5f) * ez;
            sel = (nm > mask_thr) && (fabsf(x) > tx) && (fabsf(y) > ty) && (fabsf(z) > tz);
            if (sel_mask) sel_mask[(size_t)b * HW + p] = sel ? 1 : 0;
        }
        const unsigned long long bal = __ballot(sel);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();  // (s_cnt of the previous pass has been read by everyone)
        if (lane == 0) s_cnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < SEL_WAVES; wv++) { const int c = s_cnt[wv]; woff += wv < wave ? c : 0; tot += c; }
        if (sel) {
            const int pos = n + woff + before;
            ip[2 * pos] = cu[p] * fw;
            ip[2 * pos + 1] = cv[p] * fh;
            mp[3 * pos] = x; mp[3 * pos + 1] = y; mp[3 * pos + 2] = z;
        }
        n += tot;
    }
    if (tid == 0) counts[b] = n;
}

extern "C" int rdpn6d_select_correspondences_f32(const float* out_nchw, int C, const float* coord2d, int C2, int u_ch, int v_ch,
                                                 const float* extents, const int* im_hw, int im_H, int im_W, int B, int HW,
                                                 float mask_thr, float* image_points, float* model_points, int* counts,
                                                 unsigned char* sel_mask, float* out_mask, void* stream)
{
    RD_REQUIRE(out_nchw && coord2d && extents && image_points && model_points && counts, "null pointer");
    RD_REQUIRE(B > 0 && HW > 0 && C >= 4, "B, HW > 0; the map tensor holds mask | coor_x | coor_y | coor_z | ...");
    RD_REQUIRE(C2 >= 2 && u_ch >= 0 && u_ch < C2 && v_ch >= 0 && v_ch < C2, "2D-coordinate channels");
    RD_REQUIRE(im_hw || (im_H > 0 && im_W > 0), "image size");
    hipLaunchKernelGGL(select_correspondences_kernel, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, out_nchw, C, coord2d,
                       C2, u_ch, v_ch, extents, im_hw, im_H, im_W, HW, mask_thr, image_points, model_points, counts, sel_mask,
                       out_mask);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
