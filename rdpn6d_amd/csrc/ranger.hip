// Fused multi-tensor Ranger step (RAdam + Lookahead + gradient centralisation) over flat HBM buffers.
//
// Replaces the per-tensor Python loop of lib/torch_utils/solver/ranger.py:100-200 (164 tensors x ~12 ATen
// kernels per step) by two launches:
//   1. gc_row_mean_kernel   - gradient centralisation statistics: mean of the gradient over all dims but the
//                             first, for every output row of every conv / FC weight (ranger.py:146-148)
//   2. ranger_update_kernel - one pass over parameter / gradient / exp_avg / exp_avg_sq / slow buffers
//                             (5 reads + 4 writes of 4 bytes per element: HBM-bound, 16-byte accesses)
// The RAdam rectification scalars depend only on the step count and are computed on the host
// (ranger.py:159-180); lookahead (every k steps, ranger.py:191-198) is a uniform branch in the kernel.
#include "common.h"
#include <cstdint>

// work item: a contiguous run of elements [off, off+len) that shares one centralisation mean (row >= 0: mean[row], computed by
// gc_row_mean_kernel; row >= RANGER_ROW_HERE: the item is the whole row and takes its mean itself) or none (-1)
#define RANGER_ROW_HERE 0x40000000
struct RangerWork {
    long long off;
    int len;
    int row;
};

__global__ __launch_bounds__(256) void gc_row_mean_kernel(const float* __restrict__ grad, const long long* __restrict__ row_off,
                                                          const int* __restrict__ row_len, int nrows, float* __restrict__ mean,
                                                          float inv_scale)
{
    __shared__ double s[4];
    const int r = blockIdx.x;
    if (r >= nrows) return;
    const float* g = grad + row_off[r];
    const int n = row_len[r];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)(g[i] * inv_scale);  // (inv_scale == 1: the value itself)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) mean[r] = (float)(((s[0] + s[1]) + (s[2] + s[3])) / (double)n);
}

__global__ __launch_bounds__(256) void ranger_update_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                            float* __restrict__ m, float* __restrict__ v, float* __restrict__ slow,
                                                            const RangerWork* __restrict__ work, int nwork,
                                                            const float* __restrict__ mean, float beta1, float beta2, float eps,
                                                            float neg_step_lr, float wd_lr, int rectified, int lookahead,
                                                            float alpha, float inv_scale, const int* __restrict__ found_inf)
{
    const int w = blockIdx.x;
    if (w >= nwork) return;
    if (found_inf && *found_inf) return;  // GradScaler.step: a step whose gradients are not finite is skipped (every workgroup sees the same flag)
    const RangerWork it = work[w];
    float mu = 0.f;
    if (it.row >= RANGER_ROW_HERE) {
        // the item IS a whole centralisation row: its mean is taken here, with gc_row_mean_kernel's own summation (same bits), instead of
        // a separate pass over the gradients (51 us per step for this network)
        __shared__ double s_gc[4];
        double a = 0.0;
        for (int i = threadIdx.x; i < it.len; i += 256) a += (double)(grad[it.off + i] * inv_scale);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if ((threadIdx.x & 63) == 0) s_gc[threadIdx.x >> 6] = a;
        __syncthreads();
        mu = (float)(((s_gc[0] + s_gc[1]) + (s_gc[2] + s_gc[3])) / (double)it.len);
    } else if (it.row >= 0) {
        mu = mean[it.row];
    }
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    for (int i = threadIdx.x; i < it.len; i += 256) {
        const long long j = it.off + i;
        const float g = grad[j] * inv_scale - mu;
        float vv = v[j] * beta2 + omb2 * g * g;
        float mm = m[j] * beta1 + omb1 * g;
        float pp = p[j];
        if (wd_lr != 0.f) pp += pp * (-wd_lr);
        if (rectified) pp += neg_step_lr * (mm / (sqrtf(vv) + eps));
        else pp += neg_step_lr * mm;
        v[j] = vv;
        m[j] = mm;
        if (lookahead) {
            float sl = slow[j];
            sl += alpha * (pp - sl);
            slow[j] = sl;
            pp = sl;
        }
        p[j] = pp;
    }
}

// 1 -> *flag if any of g[0 .. n) is NaN / +-Inf (the flag is cleared by the launcher first): GradScaler's found_inf for the whole flat
// gradient in one pass (torch: abs, compare, two reductions and a host read of 36 M elements)
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long long n, int* __restrict__ flag)
{
    const long long n4 = n >> 2;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float f = v[q];  // (a bit_cast of the vector ELEMENT itself reads element 0 with this compiler)
            bad |= (__builtin_bit_cast(unsigned, f) & 0x7f800000u) == 0x7f800000u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float f = g[n4 * 4 + threadIdx.x];
        bad |= (__builtin_bit_cast(unsigned, f) & 0x7f800000u) == 0x7f800000u;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1;  // (every writer stores the same value)
}

static int ranger_step_impl(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* slow, const void* work, int nwork,
                            const long long* row_off, const int* row_len, int nrows, float* row_mean, float beta1, float beta2, float eps,
                            float neg_step_lr, float wd_lr, int rectified, int lookahead, float alpha, float inv_scale,
                            const int* found_inf, void* stream)
{
    RD_REQUIRE(param && grad && exp_avg && exp_avg_sq && slow && work && nwork > 0, "null pointer / empty work list");
    RD_REQUIRE(nrows == 0 || (row_off && row_len && row_mean), "row tables");
    hipStream_t s = (hipStream_t)stream;
    if (nrows > 0) {
        hipLaunchKernelGGL(gc_row_mean_kernel, dim3(nrows), dim3(256), 0, s, grad, row_off, row_len, nrows, row_mean, inv_scale);
        RD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ranger_update_kernel, dim3(nwork), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, slow,
                       (const RangerWork*)work, nwork, row_mean, beta1, beta2, eps, neg_step_lr, wd_lr, rectified, lookahead, alpha, inv_scale,
                       found_inf);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_ranger_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* slow,
                                      const void* work /* RangerWork[nwork] */, int nwork, const long long* row_off,
                                      const int* row_len, int nrows, float* row_mean, float beta1, float beta2, float eps,
                                      float neg_step_lr, float wd_lr, int rectified, int lookahead, float alpha, void* stream)
{
    return ranger_step_impl(param, grad, exp_avg, exp_avg_sq, slow, work, nwork, row_off, row_len, nrows, row_mean, beta1, beta2, eps, neg_step_lr,
                            wd_lr, rectified, lookahead, alpha, 1.0f, nullptr, stream);
}

// The step under a loss scale (the reference: GradScaler.unscale_ + GradScaler.step around Ranger, engine.py:302-309) without the
// separate passes over the gradients: every gradient is read as grad * inv_scale (the buffer keeps the scaled values), and with
// found_inf != NULL the whole step is skipped on the device when *found_inf != 0 (rdpn6d_grad_nonfinite_f32 sets it).
extern "C" int rdpn6d_ranger_step_scaled_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* slow, const void* work,
                                             int nwork, const long long* row_off, const int* row_len, int nrows, float* row_mean, float beta1,
                                             float beta2, float eps, float neg_step_lr, float wd_lr, int rectified, int lookahead, float alpha,
                                             float inv_scale, const int* found_inf, void* stream)
{
    return ranger_step_impl(param, grad, exp_avg, exp_avg_sq, slow, work, nwork, row_off, row_len, nrows, row_mean, beta1, beta2, eps, neg_step_lr,
                            wd_lr, rectified, lookahead, alpha, inv_scale, found_inf, stream);
}

extern "C" int rdpn6d_grad_nonfinite_f32(const float* grad, long long n, int* flag, void* stream)
{
    RD_REQUIRE(grad && flag && n > 0 && ((uintptr_t)grad & 15) == 0, "gradient buffer (16-byte aligned)");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess) return RDPN6D_EHIP;
    const long long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3((unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096)), dim3(256), 0, s, grad, n, flag);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
