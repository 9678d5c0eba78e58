// Training-mode normalisation kernels (forward with batch statistics, and backward), NHWC fp32.
//
// BatchNorm2d in train mode (torch defaults: eps 1e-5, momentum 0.1, biased variance for the
// normalisation, unbiased for the running estimate) as used by every BN of
// core/gdrn_modeling/models/{resnet_backbone.py, cdpn_rot_head_region.py} (NORM="BN" -> plain
// per-GPU BatchNorm, core/utils/layer_utils.py:30), and GroupNorm(32,128)+ReLU of
// models/conv_pnp_net.py:80-82.  All are HBM-bound: one 256-byte coalesced row segment (64
// channels) per wavefront, double-precision partial sums, two-stage reductions (partials in
// global memory + a tiny finalize kernel) so results do not depend on atomics ordering.
#include "common.h"
#include <float.h>
#include <cstdlib>
// BatchNorm backward-apply reads x and dy for the last time in the step: streaming (non-temporal) loads, -0.03 ms per bf16 step at B = 32
// (round 5; RDPN6D_BN_NT=0: plain loads)
static bool bn_nt_loads() { static const int v = getenv("RDPN6D_BN_NT") ? atoi(getenv("RDPN6D_BN_NT")) : 1; return v != 0; }

#define NS_MAX 512  // row splits of the partial reductions (the scratch contract: 512 * C * 2 doubles).  128 left a 131 072 x 256 tensor on 256
                    // workgroups of 4 waves, sixteen dependent load rounds each: latency-bound at a third of the HBM rate

// ---------------------------------------------------------------------------------------------
// Stage 1 of every per-channel reduction.  MODE 0: (sum x, sum x^2)         [BN statistics, bias grads]
//                                            MODE 1: (sum g, sum g*xhat)       [BN backward], g = dy*(y>0) if relu
// grid = (C/64, S); block = 256 = 16 row lanes x 16 channel quads (16-byte loads)
// V = channels per thread: 4 (16-byte fp32 loads, 64 channels per workgroup) or 8 (16-byte bf16 loads, 128 channels per
// workgroup - the bf16-stored activations of the mixed-precision step, so that a wavefront still reads 256-byte rows)
template <int MODE, typename T, int V, int CG>
__global__ __launch_bounds__(256) void chan_partial_kernel(const T* __restrict__ x, int xcs, int xco,
                                                           const T* __restrict__ dy, int dcs, int dco,
                                                           const T* __restrict__ y, int ycs, int yco,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           long long M, int C, int relu, double* __restrict__ partial)
{
    // block = RL row lanes x CG channel groups of V: every thread streams 16-byte loads, a wavefront covers 64 / CG rows x CG * 16 B
    // (CG = 16: 256-byte rows; CG = 8 when the tensor has only 8 groups - 64 channels of a 16-bit tensor - so no lane idles)
    constexpr int CB = CG * V;  // channels per workgroup
    constexpr int RL = 256 / CG;
    __shared__ double s_a[RL][CB], s_b[RL][CB];
    const int q = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int c = blockIdx.x * CB + q * V;
    const int S = gridDim.y;
    const long long rows_per = (M + S - 1) / S;
    const long long m_lo = (long long)blockIdx.y * rows_per, m_hi = m_lo + rows_per < M ? m_lo + rows_per : M;
    double a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { a[e] = 0.0; b[e] = 0.0; }
    if (c < C) {  // C % V == 0 is required by the callers
        float mu[V], is[V], ga[V], be[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            mu[e] = MODE == 1 ? mean[c + e] : 0.f;
            is[e] = MODE == 1 ? invstd[c + e] : 0.f;
            ga[e] = (MODE == 1 && relu == 2) ? gamma[c + e] : 0.f;
            be[e] = (MODE == 1 && relu == 2) ? beta[c + e] : 0.f;
        }
        // four rows per iteration: all loads of the group are issued before the first dependent use (memory-level
        // parallelism - the kernel is latency-bound otherwise); rows past the end are clamped and contribute zero
        for (long long m0 = m_lo + rl; m0 < m_hi; m0 += 4 * RL) {
            float xv[4][V], g[4][V], yv[4][V];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long mr = m0 + RL * u;
                ok[u] = mr < m_hi;
                const long long m = ok[u] ? mr : m0;
                rd_ldv<T, V>(x + m * xcs + xco + c, xv[u]);
                if (MODE == 1) {
                    rd_ldv<T, V>(dy + m * dcs + dco + c, g[u]);
                    if (relu == 1) rd_ldv<T, V>(y + m * ycs + yco + c, yv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const double xd = ok[u] ? (double)xv[u][e] : 0.0;
                        a[e] += xd;
                        b[e] += xd * xd;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float ge = ok[u] ? g[u][e] : 0.f;
                        if (relu == 1) ge = yv[u][e] > 0.f ? ge : 0.f;
                        else if (relu == 2) ge = bn_stored_positive<T>(bn_fwd_value(xv[u][e], mu[e], is[e], ga[e], be[e])) ? ge : 0.f;
                        const float xh = (xv[u][e] - mu[e]) * is[e];
                        a[e] += (double)ge;
                        b[e] += (double)ge * (double)xh;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { s_a[rl][q * V + e] = a[e]; s_b[rl][q * V + e] = b[e]; }
    __syncthreads();
    if (threadIdx.x < CB && blockIdx.x * CB + threadIdx.x < C) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) { ta += s_a[r][threadIdx.x]; tb += s_b[r][threadIdx.x]; }
        const int cc = blockIdx.x * CB + threadIdx.x;
        partial[((long long)blockIdx.y * C + cc) * 2 + 0] = ta;
        partial[((long long)blockIdx.y * C + cc) * 2 + 1] = tb;
    }
}

template <int MODE, typename T>
static void chan_partial_launch(const T* x, int xcs, int xco, const T* dy, int dcs, int dco, const T* y, int ycs, int yco,
                                const float* mean, const float* invstd, long long M, int C, int relu, double* scratch, int S,
                                hipStream_t s, const float* gamma = nullptr, const float* beta = nullptr)
{
    // 8 channels per thread when every operand slice allows 16-byte bf16 accesses
    const bool wide = sizeof(T) == 2 && C % 8 == 0 && xcs % 8 == 0 && xco % 8 == 0 &&
                      (MODE == 0 || (dcs % 8 == 0 && dco % 8 == 0 && (relu != 1 || (ycs % 8 == 0 && yco % 8 == 0))));
    if constexpr (sizeof(T) == 2) {
        if (wide && C <= 64) {
            hipLaunchKernelGGL((chan_partial_kernel<MODE, T, 8, 8>), dim3((C + 63) / 64, S), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y,
                               ycs, yco, mean, invstd, gamma, beta, M, C, relu, scratch);
            return;
        }
        if (wide) {
            hipLaunchKernelGGL((chan_partial_kernel<MODE, T, 8, 16>), dim3((C + 127) / 128, S), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y,
                               ycs, yco, mean, invstd, gamma, beta, M, C, relu, scratch);
            return;
        }
    }
    if (C <= 32) {
        hipLaunchKernelGGL((chan_partial_kernel<MODE, T, 4, 8>), dim3((C + 31) / 32, S), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y, ycs,
                           yco, mean, invstd, gamma, beta, M, C, relu, scratch);
        return;
    }
    hipLaunchKernelGGL((chan_partial_kernel<MODE, T, 4, 16>), dim3((C + 63) / 64, S), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y, ycs,
                       yco, mean, invstd, gamma, beta, M, C, relu, scratch);
}

// channels per workgroup of the variant chan_partial_launch picks (for the split count)
template <typename T> static int chan_cb(int C)
{
    if (sizeof(T) == 2 && C % 8 == 0) return C <= 64 ? 64 : 128;
    return C <= 32 ? 32 : 64;
}

// Stage 2 helpers: block = 256 threads = 16 split lanes x 16 channels (C / 16 workgroups: a 64-channel layer gets four, not one);
// every lane adds its S / 16 partials with the loads of four splits in flight, the lanes are combined through LDS in a fixed tree.
#define FIN_CH 16
__device__ __forceinline__ void combine_partials(const double* __restrict__ partial, int S, int C, int c, bool ok, double& a,
                                                 double& b)
{
    __shared__ double s_a[16][FIN_CH], s_b[16][FIN_CH];
    const int sl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    a = 0.0; b = 0.0;
    if (ok) {
        const double* p = partial + (long long)c * 2;
        const long long st = (long long)C * 2;
        int k = sl;
        for (; k + 48 < S; k += 64) {
            const double a0 = p[k * st], b0 = p[k * st + 1], a1 = p[(k + 16) * st], b1 = p[(k + 16) * st + 1];
            const double a2 = p[(k + 32) * st], b2 = p[(k + 32) * st + 1], a3 = p[(k + 48) * st], b3 = p[(k + 48) * st + 1];
            a += (a0 + a1) + (a2 + a3);
            b += (b0 + b1) + (b2 + b3);
        }
        for (; k < S; k += 16) { a += p[k * st]; b += p[k * st + 1]; }
    }
    s_a[sl][cl] = a; s_b[sl][cl] = b;
    __syncthreads();
    if (sl == 0) {
        double ta[4], tb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ta[g] = (s_a[4 * g][cl] + s_a[4 * g + 1][cl]) + (s_a[4 * g + 2][cl] + s_a[4 * g + 3][cl]);
            tb[g] = (s_b[4 * g][cl] + s_b[4 * g + 1][cl]) + (s_b[4 * g + 2][cl] + s_b[4 * g + 3][cl]);
        }
        a = (ta[0] + ta[1]) + (ta[2] + ta[3]);
        b = (tb[0] + tb[1]) + (tb[2] + tb[3]);
    }
}

// Stage 2 for BN statistics: mean, invstd, running-stat update.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ partial, int S, int C, long long M,
                                                                float eps, float momentum, float* __restrict__ mean,
                                                                float* __restrict__ invstd, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var)
{
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
    const bool ok = c < C;
    double a, b;
    combine_partials(partial, S, C, c, ok, a, b);
    if (!ok || threadIdx.x >= FIN_CH) return;
    const double mu = a / (double)M;
    double var = b / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// The same finalize for MANY partial rows (the convolution epilogues' rows: rdpn6d_conv2d_bf16_bnstats writes one per wave row of
// an M tile - 1 024 for a head layer at B = 32): 32 split lanes x 8 channels per workgroup, fixed-order tree.
__global__ __launch_bounds__(256) void bn_stats_finalize_rows_kernel(const double* __restrict__ partial, int S, int C, long long M,
                                                                     float eps, float momentum, float* __restrict__ mean,
                                                                     float* __restrict__ invstd, float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var)
{
    constexpr int SL = 32, CH = 8;
    __shared__ double s_a[SL][CH], s_b[SL][CH];
    const int cl = threadIdx.x % CH, sl = threadIdx.x / CH;
    const int c = blockIdx.x * CH + cl;
    const bool ok = c < C;
    double a = 0.0, b = 0.0;
    if (ok) {
        const double* p = partial + (long long)c * 2;
        const long long st = (long long)C * 2;
        int k = sl;
        for (; k + 3 * SL < S; k += 4 * SL) {
            const double a0 = p[k * st], b0 = p[k * st + 1], a1 = p[(k + SL) * st], b1 = p[(k + SL) * st + 1];
            const double a2 = p[(k + 2 * SL) * st], b2 = p[(k + 2 * SL) * st + 1], a3 = p[(k + 3 * SL) * st], b3 = p[(k + 3 * SL) * st + 1];
            a += (a0 + a1) + (a2 + a3);
            b += (b0 + b1) + (b2 + b3);
        }
        for (; k < S; k += SL) { a += p[k * st]; b += p[k * st + 1]; }
    }
    s_a[sl][cl] = a; s_b[sl][cl] = b;
    __syncthreads();
#pragma unroll
    for (int w = SL / 2; w >= 1; w >>= 1) {
        if (sl < w) { s_a[sl][cl] += s_a[sl + w][cl]; s_b[sl][cl] += s_b[sl + w][cl]; }
        __syncthreads();
    }
    if (sl != 0 || !ok) return;
    const double mu = s_a[0][cl] / (double)M;
    double var = s_b[0][cl] / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// plain sums of MANY partial rows (the rows of rdpn6d_conv2d_bf16_bnbwd): out_a = sum of column 0, out_b = sum of column 1
__global__ __launch_bounds__(256) void chan_sum_finalize_rows_kernel(const double* __restrict__ partial, int S, int C,
                                                                     float* __restrict__ out_a, float* __restrict__ out_b)
{
    constexpr int SL = 32, CH = 8;
    __shared__ double s_a[SL][CH], s_b[SL][CH];
    const int cl = threadIdx.x % CH, sl = threadIdx.x / CH;
    const int c = blockIdx.x * CH + cl;
    const bool ok = c < C;
    double a = 0.0, b = 0.0;
    if (ok) {
        const double* p = partial + (long long)c * 2;
        const long long st = (long long)C * 2;
        int k = sl;
        for (; k + 3 * SL < S; k += 4 * SL) {
            const double a0 = p[k * st], b0 = p[k * st + 1], a1 = p[(k + SL) * st], b1 = p[(k + SL) * st + 1];
            const double a2 = p[(k + 2 * SL) * st], b2 = p[(k + 2 * SL) * st + 1], a3 = p[(k + 3 * SL) * st], b3 = p[(k + 3 * SL) * st + 1];
            a += (a0 + a1) + (a2 + a3);
            b += (b0 + b1) + (b2 + b3);
        }
        for (; k < S; k += SL) { a += p[k * st]; b += p[k * st + 1]; }
    }
    s_a[sl][cl] = a; s_b[sl][cl] = b;
    __syncthreads();
#pragma unroll
    for (int w = SL / 2; w >= 1; w >>= 1) {
        if (sl < w) { s_a[sl][cl] += s_a[sl + w][cl]; s_b[sl][cl] += s_b[sl + w][cl]; }
        __syncthreads();
    }
    if (sl != 0 || !ok) return;
    out_a[c] = (float)s_a[0][cl];
    out_b[c] = (float)s_b[0][cl];
}

// Stage 2 for plain sums (bias gradients) and for BN backward (dgamma, dbeta).
__global__ __launch_bounds__(256) void chan_sum_finalize_kernel(const double* __restrict__ partial, int S, int C,
                                                                float* __restrict__ out_a, float* __restrict__ out_b,
                                                                int accumulate)
{
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
    const bool ok = c < C;
    double a, b;
    combine_partials(partial, S, C, c, ok, a, b);
    if (!ok || threadIdx.x >= FIN_CH) return;
    if (out_a) out_a[c] = (accumulate ? out_a[c] : 0.f) + (float)a;
    if (out_b) out_b[c] = (accumulate ? out_b[c] : 0.f) + (float)b;
}

static int pick_splits(long long M, int C, int cb = 64)
{
    // >= 4 rows per row lane = ONE round of loads per thread (was 16 = four dependent rounds: 18 us for a 4-MB layer3 tensor on 64
    // workgroups, where the data takes 3)
    static const int rows_lane = getenv("RDPN6D_BN_ROWS_PER_LANE") ? atoi(getenv("RDPN6D_BN_ROWS_PER_LANE")) : 4;  // profiling
    long long s = (M + 16 * rows_lane - 1) / (16 * rows_lane);
    const long long want = 2048 / ((C + cb - 1) / cb) + 1;  // enough workgroups to fill the chip (8 per CU)
    if (s > want) s = want;
    if (s > NS_MAX) s = NS_MAX;
    if (s < 1) s = 1;
    return (int)s;
}

template <typename T>
static int bn_train_stats_impl(const T* x, long long M, int C, int cs, int co, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, double* scratch, void* stream)
{
    RD_REQUIRE(x && mean && invstd && scratch, "null pointer");
    RD_REQUIRE(M > 0 && C > 0 && co + C <= cs && C % 4 == 0 && cs % 4 == 0 && co % 4 == 0, "shape / 16-byte alignment");
    const int S = pick_splits(M, C, chan_cb<T>(C));
    hipStream_t s = (hipStream_t)stream;
    chan_partial_launch<0, T>(x, cs, co, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, M, C, 0, scratch, S, s);
    RD_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, scratch, S, C, M, eps, momentum, mean,
                       invstd, running_mean, running_var);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_bn_train_stats_f32(const float* x, long long M, int C, int cs, int co, float eps, float momentum,
                                         float* mean, float* invstd, float* running_mean, float* running_var,
                                         double* scratch /* >= 512*C*2 doubles */, void* stream)
{
    return bn_train_stats_impl<float>(x, M, C, cs, co, eps, momentum, mean, invstd, running_mean, running_var, scratch, stream);
}
// bf16 activation (mixed-precision training with bf16-stored activations): same statistics, fp64 partial sums
extern "C" int rdpn6d_bn_train_stats_bf16(const void* x, long long M, int C, int cs, int co, float eps, float momentum,
                                          float* mean, float* invstd, float* running_mean, float* running_var, double* scratch,
                                          void* stream)
{
    return bn_train_stats_impl<rd_bf16_t>((const rd_bf16_t*)x, M, C, cs, co, eps, momentum, mean, invstd, running_mean,
                                          running_var, scratch, stream);
}

// Stage 2 alone: mean / invstd / running statistics from S rows of per-channel (sum, sum of squares) partials [S][C][2] that somebody
// else produced - the epilogue of the convolution in front of the BatchNorm (rdpn6d_conv2d_bf16_bnstats)
extern "C" int rdpn6d_bn_stats_finalize(const double* partial, int S, int C, long long M, float eps, float momentum, float* mean,
                                        float* invstd, float* running_mean, float* running_var, void* stream)
{
    RD_REQUIRE(partial && mean && invstd && S > 0 && C > 0 && M > 0, "arguments");
    hipStream_t s = (hipStream_t)stream;
    if (S > 64)
        hipLaunchKernelGGL(bn_stats_finalize_rows_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, S, C, M, eps, momentum, mean, invstd,
                           running_mean, running_var);
    else
        hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, partial, S, C, M, eps, momentum, mean,
                           invstd, running_mean, running_var);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// per-channel sum of rows (bias gradient): out[c] (+)= sum_m x[m, co + c]
template <typename T>
static int channel_sum_impl(const T* x, long long M, int C, int cs, int co, float* out, int accumulate, double* scratch, void* stream)
{
    RD_REQUIRE(x && out && scratch && M > 0 && C > 0 && co + C <= cs && C % 4 == 0 && cs % 4 == 0 && co % 4 == 0, "shape / alignment");
    const int S = pick_splits(M, C, chan_cb<T>(C));
    hipStream_t s = (hipStream_t)stream;
    chan_partial_launch<0, T>(x, cs, co, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, M, C, 0, scratch, S, s);
    RD_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, scratch, S, C, out, nullptr, accumulate);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
extern "C" int rdpn6d_channel_sum_f32(const float* x, long long M, int C, int cs, int co, float* out, int accumulate,
                                      double* scratch, void* stream)
{
    return channel_sum_impl<float>(x, M, C, cs, co, out, accumulate, scratch, stream);
}
extern "C" int rdpn6d_channel_sum_bf16(const void* x, long long M, int C, int cs, int co, float* out, int accumulate,
                                       double* scratch, void* stream)
{
    return channel_sum_impl<rd_bf16_t>((const rd_bf16_t*)x, M, C, cs, co, out, accumulate, scratch, stream);
}

// ---------------------------------------------------------------------------------------------
// y = act(((x - mean) * invstd) * gamma + beta (+ res)), 4 channels per thread
template <typename T, int V>
__global__ void bn_apply_kernel(const T* __restrict__ x, int xcs, int xco, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ res, int rcs, int rco,
                                T* __restrict__ y, int ycs, int yco, long long M, int C, int relu)
{
    const int CV = C / V;
    const long long total = M * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        const long long m = i / CV;
        float v[V], o[V];
        rd_ldv<T, V>(x + m * xcs + xco + c, v);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = bn_fwd_value(v[e], mean[c + e], invstd[c + e], gamma[c + e], beta[c + e]);
        if (res) {
            float r[V];
            rd_ldv<T, V>(res + m * rcs + rco + c, r);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] += r[e];
        }
        if (relu)
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
        rd_stv<T, V>(y + m * ycs + yco + c, o);
    }
}

template <typename T>
static int bn_apply_impl(const T* x, int xcs, int xco, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const T* res, int rcs, int rco, T* y, int ycs, int yco, long long M, int C, int relu,
                         void* stream)
{
    RD_REQUIRE(x && mean && invstd && gamma && beta && y, "null pointer");
    RD_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && xco % 4 == 0 && yco % 4 == 0 && xcs % 4 == 0 && ycs % 4 == 0, "shape/alignment");
    if constexpr (sizeof(T) == 2) {
        if (C % 8 == 0 && xcs % 8 == 0 && xco % 8 == 0 && ycs % 8 == 0 && yco % 8 == 0 && (!res || (rcs % 8 == 0 && rco % 8 == 0))) {
            const long long total8 = M * (C / 8);
            const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
            hipLaunchKernelGGL((bn_apply_kernel<T, 8>), dim3(blocks8), dim3(256), 0, (hipStream_t)stream, x, xcs, xco, mean, invstd, gamma,
                               beta, res, rcs, rco, y, ycs, yco, M, C, relu);
            RD_LAUNCH_CHECK();
            return RDPN6D_OK;
        }
    }
    const long long total = M * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((bn_apply_kernel<T, 4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xcs, xco, mean, invstd, gamma, beta,
                       res, rcs, rco, y, ycs, yco, M, C, relu);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
extern "C" int rdpn6d_bn_apply_f32(const float* x, int xcs, int xco, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, const float* res, int rcs, int rco, float* y,
                                   int ycs, int yco, long long M, int C, int relu, void* stream)
{
    return bn_apply_impl<float>(x, xcs, xco, mean, invstd, gamma, beta, res, rcs, rco, y, ycs, yco, M, C, relu, stream);
}
extern "C" int rdpn6d_bn_apply_bf16(const void* x, int xcs, int xco, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, const void* res, int rcs, int rco, void* y, int ycs, int yco, long long M,
                                    int C, int relu, void* stream)
{
    return bn_apply_impl<rd_bf16_t>((const rd_bf16_t*)x, xcs, xco, mean, invstd, gamma, beta, (const rd_bf16_t*)res, rcs, rco,
                                    (rd_bf16_t*)y, ycs, yco, M, C, relu, stream);
}

// ---------------------------------------------------------------------------------------------
// BN backward.  g = dy * (y > 0) when the BN was followed by ReLU: relu == 1 reads the stored activation y; relu == 2 (no residual
// between the BN and its ReLU) re-derives the mask from x - bn_fwd_value, rounded as it was stored - and never touches y: one
// tensor read less in the reduction pass and in the apply pass (the head's seven 67-MB activations and the stem's, at B = 32).
//   dgamma = sum g*xhat, dbeta = sum g
//   dx = gamma*invstd * (g - dbeta/M - xhat*dgamma/M);   dres (optional) = g  (identity branch of a residual block)
template <typename T, int V, bool NT = false>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ x, int xcs, int xco, const T* __restrict__ dy, int dcs, int dco,
                                    const T* __restrict__ y, int ycs, int yco, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, T* __restrict__ dx, int xgcs, int xgco,
                                    T* __restrict__ dres, int rcs, int rco, long long M, int C, int relu)
{
    const int CV = C / V;
    const long long total = M * CV;
    const float invM = 1.0f / (float)M;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        const long long m = i / CV;
        float xv[V], g[V], o[V];
        if constexpr (NT) {
            rd_ldv8_nt<T>(x + m * xcs + xco + c, xv);
            rd_ldv8_nt<T>(dy + m * dcs + dco + c, g);
        } else {
            rd_ldv<T, V>(x + m * xcs + xco + c, xv);
            rd_ldv<T, V>(dy + m * dcs + dco + c, g);
        }
        if (relu == 1) {
            float yv[V];
            rd_ldv<T, V>(y + m * ycs + yco + c, yv);
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
        } else if (relu == 2) {
#pragma unroll
            for (int e = 0; e < V; ++e)
                g[e] = bn_stored_positive<T>(bn_fwd_value(xv[e], mean[c + e], invstd[c + e], gamma[c + e], beta[c + e])) ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (xv[e] - mean[c + e]) * invstd[c + e];
            o[e] = gamma[c + e] * invstd[c + e] * (g[e] - dbeta[c + e] * invM - xh * dgamma[c + e] * invM);
        }
        rd_stv<T, V>(dx + m * xgcs + xgco + c, o);
        if (dres) rd_stv<T, V>(dres + m * rcs + rco + c, g);
    }
}

template <typename T>
static int bn_backward_impl(const T* x, int xcs, int xco, const T* dy, int dcs, int dco, const T* y, int ycs, int yco,
                            const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, T* dx, int xgcs,
                            int xgco, T* dres, int rcs, int rco, long long M, int C, int relu, double* scratch, void* stream,
                            const float* beta = nullptr)
{
    RD_REQUIRE(x && dy && mean && invstd && gamma && dgamma && dbeta && dx && scratch, "null pointer");
    RD_REQUIRE(relu >= 0 && relu <= 2, "relu: 0 none, 1 mask from y, 2 mask re-derived from x");
    RD_REQUIRE(relu != 1 || y, "ReLU mask needs the forward output");
    RD_REQUIRE(relu != 2 || beta, "re-deriving the ReLU mask needs beta");
    RD_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "shape");
    const int S = pick_splits(M, C, chan_cb<T>(C));
    hipStream_t s = (hipStream_t)stream;
    chan_partial_launch<1, T>(x, xcs, xco, dy, dcs, dco, y, ycs, yco, mean, invstd, M, C, relu, scratch, S, s, gamma, beta);
    RD_LAUNCH_CHECK();
    // partial = (sum g, sum g*xhat) -> dbeta, dgamma
    hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, scratch, S, C, dbeta, dgamma, 0);
    RD_LAUNCH_CHECK();
    if constexpr (sizeof(T) == 2) {
        if (C % 8 == 0 && xcs % 8 == 0 && xco % 8 == 0 && dcs % 8 == 0 && dco % 8 == 0 && xgcs % 8 == 0 && xgco % 8 == 0 &&
            (relu != 1 || (ycs % 8 == 0 && yco % 8 == 0)) && (!dres || (rcs % 8 == 0 && rco % 8 == 0))) {
            const long long total8 = M * (C / 8);
            const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
            if (bn_nt_loads())
                hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8, true>), dim3(blocks8), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y, ycs, yco, mean,
                                   invstd, gamma, beta, dgamma, dbeta, dx, xgcs, xgco, dres, rcs, rco, M, C, relu);
            else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8>), dim3(blocks8), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y, ycs, yco, mean,
                               invstd, gamma, beta, dgamma, dbeta, dx, xgcs, xgco, dres, rcs, rco, M, C, relu);
            RD_LAUNCH_CHECK();
            return RDPN6D_OK;
        }
    }
    const long long total = M * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, y, ycs, yco, mean, invstd,
                       gamma, beta, dgamma, dbeta, dx, xgcs, xgco, dres, rcs, rco, M, C, relu);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
extern "C" int rdpn6d_bn_backward_f32(const float* x, int xcs, int xco, const float* dy, int dcs, int dco, const float* y,
                                      int ycs, int yco, const float* mean, const float* invstd, const float* gamma,
                                      float* dgamma, float* dbeta, float* dx, int xgcs, int xgco, float* dres, int rcs,
                                      int rco, long long M, int C, int relu, double* scratch, void* stream)
{
    return bn_backward_impl<float>(x, xcs, xco, dy, dcs, dco, y, ycs, yco, mean, invstd, gamma, dgamma, dbeta, dx, xgcs, xgco, dres, rcs,
                                   rco, M, C, relu, scratch, stream);
}
extern "C" int rdpn6d_bn_backward_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const void* y, int ycs,
                                       int yco, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                                       float* dbeta, void* dx, int xgcs, int xgco, void* dres, int rcs, int rco, long long M, int C,
                                       int relu, double* scratch, void* stream)
{
    return bn_backward_impl<rd_bf16_t>((const rd_bf16_t*)x, xcs, xco, (const rd_bf16_t*)dy, dcs, dco, (const rd_bf16_t*)y, ycs, yco, mean,
                                       invstd, gamma, dgamma, dbeta, (rd_bf16_t*)dx, xgcs, xgco, (rd_bf16_t*)dres, rcs, rco, M, C, relu,
                                       scratch, stream);
}

// BN + ReLU backward without the stored activation (relu == 2 above): the mask is sign(bn_fwd_value(x)) as the forward stored it
extern "C" int rdpn6d_bn_relu_backward_f32(const float* x, int xcs, int xco, const float* dy, int dcs, int dco, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                           float* dx, int xgcs, int xgco, long long M, int C, double* scratch, void* stream)
{
    return bn_backward_impl<float>(x, xcs, xco, dy, dcs, dco, nullptr, 0, 0, mean, invstd, gamma, dgamma, dbeta, dx, xgcs, xgco, nullptr, 0,
                                   0, M, C, 2, scratch, stream, beta);
}
extern "C" int rdpn6d_bn_relu_backward_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const float* mean,
                                            const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                            void* dx, int xgcs, int xgco, long long M, int C, double* scratch, void* stream)
{
    return bn_backward_impl<rd_bf16_t>((const rd_bf16_t*)x, xcs, xco, (const rd_bf16_t*)dy, dcs, dco, nullptr, 0, 0, mean, invstd, gamma,
                                       dgamma, dbeta, (rd_bf16_t*)dx, xgcs, xgco, nullptr, 0, 0, M, C, 2, scratch, stream, beta);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(G groups of 4 channels) + ReLU, training form: out-of-place, statistics saved.
// The second half of rdpn6d_bn_relu_backward_*: dgamma / dbeta from S rows [S][C][2] of (sum g, sum g * xhat) partials that the
// input-gradient convolution in front wrote (rdpn6d_conv2d_bf16_bnbwd), then the dx pass (mask re-derived from x as there).
template <typename T>
static int bn_relu_backward_apply_impl(const T* x, int xcs, int xco, const T* dy, int dcs, int dco, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, float* dgamma, float* dbeta, T* dx, int xgcs, int xgco,
                                       long long M, int C, const double* partial, int S, void* stream)
{
    RD_REQUIRE(x && dy && mean && invstd && gamma && beta && dgamma && dbeta && dx && partial && S > 0, "null pointer");
    RD_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(chan_sum_finalize_rows_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, S, C, dbeta, dgamma);
    RD_LAUNCH_CHECK();
    if constexpr (sizeof(T) == 2) {
        if (C % 8 == 0 && xcs % 8 == 0 && xco % 8 == 0 && dcs % 8 == 0 && dco % 8 == 0 && xgcs % 8 == 0 && xgco % 8 == 0) {
            const long long total8 = M * (C / 8);
            const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
            if (bn_nt_loads())
                hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8, true>), dim3(blocks8), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, (const T*)nullptr, 0, 0,
                                   mean, invstd, gamma, beta, dgamma, dbeta, dx, xgcs, xgco, (T*)nullptr, 0, 0, M, C, 2);
            else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8>), dim3(blocks8), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, (const T*)nullptr, 0, 0,
                               mean, invstd, gamma, beta, dgamma, dbeta, dx, xgcs, xgco, (T*)nullptr, 0, 0, M, C, 2);
            RD_LAUNCH_CHECK();
            return RDPN6D_OK;
        }
    }
    const long long total = M * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, x, xcs, xco, dy, dcs, dco, (const T*)nullptr, 0, 0, mean,
                       invstd, gamma, beta, dgamma, dbeta, dx, xgcs, xgco, (T*)nullptr, 0, 0, M, C, 2);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
extern "C" int rdpn6d_bn_relu_backward_apply_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const float* mean,
                                                  const float* invstd, const float* gamma, const float* beta, float* dgamma,
                                                  float* dbeta, void* dx, int xgcs, int xgco, long long M, int C,
                                                  const double* partial, int S, void* stream)
{
    return bn_relu_backward_apply_impl<rd_bf16_t>((const rd_bf16_t*)x, xcs, xco, (const rd_bf16_t*)dy, dcs, dco, mean, invstd, gamma, beta,
                                                  dgamma, dbeta, (rd_bf16_t*)dx, xgcs, xgco, M, C, partial, S, stream);
}

// ... and of rdpn6d_bn_backward_bf16(relu = 1) - the last BatchNorm of a residual block: mask from the stored block output y, dres = the
// masked gradient (the identity branch's) - after rdpn6d_conv2d_bf16_bnbwd_y wrote the S rows of partial sums.
extern "C" int rdpn6d_bn_backward_apply_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const void* y, int ycs,
                                             int yco, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                                             float* dbeta, void* dx, int xgcs, int xgco, void* dres, int rcs, int rco, long long M, int C,
                                             const double* partial, int S, void* stream)
{
    typedef rd_bf16_t T;
    RD_REQUIRE(x && dy && y && mean && invstd && gamma && dgamma && dbeta && dx && partial && S > 0, "null pointer");
    RD_REQUIRE(M > 0 && C > 0 && C % 8 == 0, "shape");
    RD_REQUIRE(xcs % 8 == 0 && xco % 8 == 0 && dcs % 8 == 0 && dco % 8 == 0 && xgcs % 8 == 0 && xgco % 8 == 0 && ycs % 8 == 0 && yco % 8 == 0 &&
                   (!dres || (rcs % 8 == 0 && rco % 8 == 0)),
               "16-byte aligned channel slices");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(chan_sum_finalize_rows_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, S, C, dbeta, dgamma);
    RD_LAUNCH_CHECK();
    const long long total8 = M * (C / 8);
    const int blocks8 = (int)((total8 + 255) / 256 < 16384 ? (total8 + 255) / 256 : 16384);
    if (bn_nt_loads())
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8, true>), dim3(blocks8), dim3(256), 0, s, (const T*)x, xcs, xco, (const T*)dy, dcs, dco, (const T*)y,
                           ycs, yco, mean, invstd, gamma, (const float*)nullptr, dgamma, dbeta, (T*)dx, xgcs, xgco, (T*)dres, rcs, rco, M, C, 1);
    else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8>), dim3(blocks8), dim3(256), 0, s, (const T*)x, xcs, xco, (const T*)dy, dcs, dco, (const T*)y, ycs,
                       yco, mean, invstd, gamma, (const float*)nullptr, dgamma, dbeta, (T*)dx, xgcs, xgco, (T*)dres, rcs, rco, M, C, 1);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// (NT threads per crop: 1024 for maps of >= 256 pixels - one workgroup per crop is all the parallelism there is, B = 32 workgroups on
//  256 CUs, and with 256 threads each of them walked 128 pixels three times: 75 us forward / 110 us backward for ConvPnPNet's first map)
template <int NT>
__global__ __launch_bounds__(NT) void gn4_fwd_train_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ stats /* [B][G][2] mean, rstd */)
{
    // (the groups are independent: workgroup (crop blockIdx.x, part blockIdx.y) owns GT / gridDim.y of the GT = C / 4 groups)
    const int GT = C / 4, G = GT / (int)gridDim.y, g0 = (int)blockIdx.y * G, PL = NT / G;
    __shared__ float s_part[NT];
    __shared__ float s_mean[64], s_rstd[64];
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    const float* xb = x + (long long)blockIdx.x * HW * C + (g0 + g) * 4;
    float* yb = y + (long long)blockIdx.x * HW * C + (g0 + g) * 4;
    float s = 0.f;
    for (int p = pl; p < HW; p += PL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)p * C);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int i = 0; i < PL; ++i) t += s_part[i * G + threadIdx.x];
        s_mean[threadIdx.x] = t / (float)(HW * 4);
    }
    __syncthreads();
    const float mean = s_mean[g];
    float q = 0.f;
    for (int p = pl; p < HW; p += PL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
    }
    __syncthreads();
    s_part[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int i = 0; i < PL; ++i) t += s_part[i * G + threadIdx.x];
        const float r = 1.0f / sqrtf(t / (float)(HW * 4) + 1e-5f);
        s_rstd[threadIdx.x] = r;
        stats[((long long)blockIdx.x * GT + g0 + threadIdx.x) * 2 + 0] = s_mean[threadIdx.x];
        stats[((long long)blockIdx.x * GT + g0 + threadIdx.x) * 2 + 1] = r;
    }
    __syncthreads();
    const float rstd = s_rstd[g];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + (g0 + g) * 4), be = *reinterpret_cast<const f32x4*>(beta + (g0 + g) * 4);
    for (int p = pl; p < HW; p += PL) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float o = (v[e] - mean) * rstd * ga[e] + be[e]; v[e] = o > 0.f ? o : 0.f; }
        *reinterpret_cast<f32x4*>(yb + (long long)p * C) = v;
    }
}

// backward: g = dy*(y>0); per (b, group): s1 = sum g*gamma, s2 = sum g*gamma*xhat over the group's HW*4 elements;
// dx = rstd*(g*gamma - s1/n - xhat*s2/n); per-sample partial dgamma/dbeta -> [B][C][2] (reduced by channel_sum afterwards)
template <int NT>
__global__ __launch_bounds__(NT) void gn4_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ dy, const float* __restrict__ gamma,
                                                     const float* __restrict__ stats, float* __restrict__ dx,
                                                     float* __restrict__ dgb /* [B][2][C] */, int HW, int C)
{
    const int GT = C / 4, G = GT / (int)gridDim.y, g0 = (int)blockIdx.y * G, PL = NT / G;  // (parts of the groups as in the forward)
    __shared__ float s_p1[NT], s_p2[NT];
    __shared__ float s_s1[64], s_s2[64];
    __shared__ float s_dg[NT][4], s_db[NT][4];
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    const long long base = (long long)blockIdx.x * HW * C + (g0 + g) * 4;
    const float mean = stats[((long long)blockIdx.x * GT + g0 + g) * 2], rstd = stats[((long long)blockIdx.x * GT + g0 + g) * 2 + 1];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + (g0 + g) * 4);
    float s1 = 0.f, s2 = 0.f;
    float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    for (int p = pl; p < HW; p += PL) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + (long long)p * C);
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + base + (long long)p * C);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + base + (long long)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = yv[e] > 0.f ? gv[e] : 0.f;
            const float xh = (xv[e] - mean) * rstd;
            s1 += gg * ga[e];
            s2 += gg * ga[e] * xh;
            dg[e] += gg * xh;
            db[e] += gg;
        }
    }
    s_p1[threadIdx.x] = s1;
    s_p2[threadIdx.x] = s2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { s_dg[threadIdx.x][e] = dg[e]; s_db[threadIdx.x][e] = db[e]; }
    __syncthreads();
    if (threadIdx.x < G) {
        float t1 = 0.f, t2 = 0.f, tg[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
        for (int i = 0; i < PL; ++i) {
            t1 += s_p1[i * G + threadIdx.x];
            t2 += s_p2[i * G + threadIdx.x];
#pragma unroll
            for (int e = 0; e < 4; ++e) { tg[e] += s_dg[i * G + threadIdx.x][e]; tb[e] += s_db[i * G + threadIdx.x][e]; }
        }
        s_s1[threadIdx.x] = t1;
        s_s2[threadIdx.x] = t2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dgb[((long long)blockIdx.x * 2 + 0) * C + (g0 + threadIdx.x) * 4 + e] = tg[e];
            dgb[((long long)blockIdx.x * 2 + 1) * C + (g0 + threadIdx.x) * 4 + e] = tb[e];
        }
    }
    __syncthreads();
    const float n = (float)(HW * 4);
    const float m1 = s_s1[g] / n, m2 = s_s2[g] / n;
    for (int p = pl; p < HW; p += PL) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + (long long)p * C);
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + base + (long long)p * C);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + base + (long long)p * C);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = yv[e] > 0.f ? gv[e] : 0.f;
            const float xh = (xv[e] - mean) * rstd;
            o[e] = rstd * (gg * ga[e] - m1 - xh * m2);
        }
        *reinterpret_cast<f32x4*>(dx + base + (long long)p * C) = o;
    }
}

extern "C" int rdpn6d_groupnorm_relu_train_f32(const float* x, float* y, int B, int HW, int C, int G, const float* gamma,
                                               const float* beta, float* stats, void* stream)
{
    RD_REQUIRE(x && y && gamma && beta && stats && B > 0 && HW > 0, "null/shape");
    RD_REQUIRE(C == 4 * G && G <= 64 && 256 % G == 0, "only C/G == 4 with G | 256 is implemented (GroupNorm(32,128))");
    if (HW >= 256) hipLaunchKernelGGL(gn4_fwd_train_kernel<1024>, dim3(B, G % 4 == 0 ? 4 : 1), dim3(1024), 0, (hipStream_t)stream, x, y, HW, C, gamma, beta, stats);
    else hipLaunchKernelGGL(gn4_fwd_train_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, x, y, HW, C, gamma, beta, stats);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_groupnorm_relu_backward_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                                  const float* stats, float* dx, float* dgamma, float* dbeta,
                                                  float* dgb_scratch /* [B][2][C] */, double* scratch, int B, int HW, int C,
                                                  int G, void* stream)
{
    RD_REQUIRE(x && y && dy && gamma && stats && dx && dgamma && dbeta && dgb_scratch && scratch, "null pointer");
    RD_REQUIRE(C == 4 * G && G <= 64 && 256 % G == 0, "only C/G == 4 with G | 256 is implemented");
    if (HW >= 256) hipLaunchKernelGGL(gn4_bwd_kernel<1024>, dim3(B, G % 4 == 0 ? 4 : 1), dim3(1024), 0, (hipStream_t)stream, x, y, dy, gamma, stats, dx, dgb_scratch, HW, C);
    else hipLaunchKernelGGL(gn4_bwd_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, x, y, dy, gamma, stats, dx, dgb_scratch, HW, C);
    RD_LAUNCH_CHECK();
    // [B][2C] rows -> per-column sums: dgamma = cols [0,C), dbeta = cols [C,2C)
    int rc = rdpn6d_channel_sum_f32(dgb_scratch, B, C, 2 * C, 0, dgamma, 0, scratch, stream);
    if (rc) return rc;
    return rdpn6d_channel_sum_f32(dgb_scratch, B, C, 2 * C, C, dbeta, 0, scratch, stream);
}
