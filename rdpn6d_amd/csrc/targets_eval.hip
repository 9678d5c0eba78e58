// The two small kernels either side of the training / inference path.
//
// 1. region / residual training targets (core/utils/data_utils.py:229-244 xyz_to_region +
//    core/gdrn_modeling/data_loader.py:881-903): per pixel, nearest region anchor (float64 Euclidean distance,
//    first arg-min like scipy cdist + np.argmin), residual delta = xyz - anchor, rotated by the GT pose and
//    normalised by the object extent.  The reference runs scipy cdist on the CPU per sample.
// 2. pose errors ADD / ADI / re / te (lib/pysixd/pose_error.py:297-337,400-436) in float64: ADI's nearest
//    neighbour search (scipy cKDTree in the reference) is an exact brute-force min over the LDS-resident
//    transformed model points.
#include "common.h"
#include <float.h>

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void region_targets_kernel(const float* __restrict__ xyz /* [B,HW,3] */,
                                                             const double* __restrict__ fps /* [B,K,3] */,
                                                             const float* __restrict__ rot /* [B,9] */,
                                                             const float* __restrict__ extent /* [B,3] */, int B, int HW, int K,
                                                             float* __restrict__ roi_xyz /* [B,3,HW] */,
                                                             long long* __restrict__ roi_region /* [B,HW] */)
{
    extern __shared__ double s_fps[];  // K*3
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < K * 3; i += 256) s_fps[i] = fps[(long long)b * K * 3 + i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* v = xyz + ((long long)b * HW + p) * 3;
    const double x = (double)v[0], y = (double)v[1], z = (double)v[2];
    int best = 0;
    double bd = DBL_MAX;
    for (int k = 0; k < K; ++k) {
        const double dx = x - s_fps[k * 3], dy = y - s_fps[k * 3 + 1], dz = z - s_fps[k * 3 + 2];
        double d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        if (d < bd) { bd = d; best = k; }
    }
    const bool fg = (v[0] != 0.f) || (v[1] != 0.f) || (v[2] != 0.f);
    roi_region[(long long)b * HW + p] = fg ? (long long)(best + 1) : 0;
    const double d0 = x - s_fps[best * 3], d1 = y - s_fps[best * 3 + 1], d2 = z - s_fps[best * 3 + 2];
    const float* R = rot + b * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double r = (double)R[c * 3] * d0;
        r = r + (double)R[c * 3 + 1] * d1;
        r = r + (double)R[c * 3 + 2] * d2;
        // the reference keeps delta in float64 through the normalisation and casts to float32 at the very end
        roi_xyz[((long long)b * 3 + c) * HW + p] = (float)(r / (double)extent[b * 3 + c] + 0.5);
    }
}

extern "C" int rdpn6d_region_targets_f32(const float* xyz_hwc, const double* fps, const float* rot, const float* extent, int B,
                                         int HW, int K, float* roi_xyz_chw, long long* roi_region, void* stream)
{
    RD_REQUIRE(xyz_hwc && fps && rot && extent && roi_xyz_chw && roi_region, "null pointer");
    RD_REQUIRE(B > 0 && HW > 0 && K > 0 && K <= 1024, "shape");
    hipLaunchKernelGGL(region_targets_kernel, dim3((HW + 255) / 256, B), dim3(256), K * 3 * sizeof(double), (hipStream_t)stream, xyz_hwc,
                       fps, rot, extent, B, HW, K, roi_xyz_chw, roi_region);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// one workgroup per pose; pts [n,3] shared by all poses (pts_stride = 0) or per pose (pts_stride = n*3)
__global__ __launch_bounds__(256) void pose_errors_kernel(const double* __restrict__ est /* [B,12] R|t */,
                                                          const double* __restrict__ gt /* [B,12] */,
                                                          const double* __restrict__ pts, long long pts_stride, int n,
                                                          double* __restrict__ scratch /* [B,n,3] est points when n is large */,
                                                          int use_lds, double* __restrict__ out /* [B,4] add adi re te */)
{
    extern __shared__ double s_est[];  // n*3 when use_lds
    __shared__ double s_red[4][2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* E = est + b * 12;
    const double* G = gt + b * 12;
    const double* P = pts + (long long)b * pts_stride;
    double* pe = use_lds ? s_est : scratch + (long long)b * n * 3;
    double add = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
        double e[3], g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            e[c] = ((E[c * 3] * x + E[c * 3 + 1] * y) + E[c * 3 + 2] * z) + E[9 + c];
            g[c] = ((G[c * 3] * x + G[c * 3 + 1] * y) + G[c * 3 + 2] * z) + G[9 + c];
            pe[i * 3 + c] = e[c];
        }
        const double d0 = e[0] - g[0], d1 = e[1] - g[1], d2 = e[2] - g[2];
        add += sqrt((d0 * d0 + d1 * d1) + d2 * d2);
    }
    __syncthreads();
    __threadfence_block();
    double adi = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
        double g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = ((G[c * 3] * x + G[c * 3 + 1] * y) + G[c * 3 + 2] * z) + G[9 + c];
        double best = DBL_MAX;
        for (int j = 0; j < n; ++j) {
            const double d0 = pe[j * 3] - g[0], d1 = pe[j * 3 + 1] - g[1], d2 = pe[j * 3 + 2] - g[2];
            const double d = (d0 * d0 + d1 * d1) + d2 * d2;
            best = d < best ? d : best;
        }
        adi += sqrt(best);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { add += __shfl_xor(add, o); adi += __shfl_xor(adi, o); }
    if ((tid & 63) == 0) { s_red[tid >> 6][0] = add; s_red[tid >> 6][1] = adi; }
    __syncthreads();
    if (tid == 0) {
        add = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
        adi = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
        out[b * 4 + 0] = add / (double)n;
        out[b * 4 + 1] = adi / (double)n;
        // re: trace(R_est R_gt^T)
        double tr = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) tr += E[i * 3 + k] * G[i * 3 + k];
        tr = tr <= 3.0 ? tr : 3.0;
        double c = 0.5 * (tr - 1.0);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        out[b * 4 + 2] = acos(c) * (180.0 / 3.14159265358979323846);
        const double t0 = G[9] - E[9], t1 = G[10] - E[10], t2 = G[11] - E[11];
        out[b * 4 + 3] = sqrt((t0 * t0 + t1 * t1) + t2 * t2);
    }
}

extern "C" int rdpn6d_pose_errors_f64(const double* est, const double* gt, const double* pts, int pts_per_pose, int n, int B,
                                      double* scratch, double* out, void* stream)
{
    RD_REQUIRE(est && gt && pts && out && B > 0 && n > 0, "null/shape");
    const size_t lds = (size_t)n * 3 * sizeof(double);
    const int use_lds = lds <= 150 * 1024;
    RD_REQUIRE(use_lds || scratch, "more than 6400 model points need a [B,n,3] double scratch");
    if (use_lds) {
        RD_LDS_OPT_IN(pose_errors_kernel, 150 * 1024);
    }
    hipLaunchKernelGGL(pose_errors_kernel, dim3(B), dim3(256), use_lds ? lds : 0, (hipStream_t)stream, est, gt, pts,
                       pts_per_pose ? (long long)n * 3 : 0LL, n, scratch, use_lds, out);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
