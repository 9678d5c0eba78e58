// Farthest point sampling on gfx950: one 1024-thread workgroup per point cloud.
//
// Replaces core/csrc/fps/src/farthest_point_sampling.cpp (update_min_dist :41-54,
// find_max_dist_idx :56-73, sample_farthest_points :77-105, ..._init_center :122-160) behind the
// same two extern "C" symbols (src/ext.h:1-14).
//
// Bit-exact index parity with the g++ -O2 (no FMA) reference is part of the contract:
//   * squared distance = ((dx*dx)+(dy*dy))+(dz*dz), one IEEE rounding per op (__fmul_rn/__fadd_rn
//     are never contracted into FMAs);
//   * arg-max with strict '>' from max_d = 0 / max_idx = 0 and lowest-index tie-break: each thread
//     scans its points in ascending index order, candidates are merged with (value desc, index asc);
//   * already selected points are skipped: they are tagged with min_dist = -1, which can neither
//     win the arg-max (needs > 0) nor be lowered by the update (needs dist < -1).
// Points (and their running min-distance) live in registers when the cloud has <= 16 points per
// thread (N <= 16384: every BOP model after the usual decimation), otherwise they stream from L2.
#include "common.h"
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <time.h>

// HIP's __fmul_rn/__fadd_rn are plain operators, so the compiler's default fp-contract=fast would
// still fuse them into FMAs: contraction is switched off for this whole translation unit (and the
// build passes -ffp-contract=off for this file as well).
#pragma clang fp contract(off)

#define FPS_THREADS 1024

__device__ __forceinline__ float fps_sqdist(float px, float py, float pz, float cx, float cy, float cz)
{
    const float dx = __fsub_rn(px, cx), dy = __fsub_rn(py, cy), dz = __fsub_rn(pz, cz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// (value desc, index asc) merge across the workgroup; returns the winner to every thread.
__device__ __forceinline__ int fps_block_argmax(float v, int idx, float* s_v, int* s_i)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // protect s_v/s_i reuse across iterations
    if (lane == 0) { s_v[wave] = v; s_i[wave] = idx; }
    __syncthreads();
    if (wave == 0) {
        v = lane < FPS_THREADS / 64 ? s_v[lane] : 0.f;
        idx = lane < FPS_THREADS / 64 ? s_i[lane] : INT_MAX;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o);
            const int oi = __shfl_xor(idx, o);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if (lane == 0) s_i[16] = (idx == INT_MAX) ? 0 : idx;
    }
    __syncthreads();
    return s_i[16];
}

// PPT > 0: points cached in registers (PPT per thread); PPT == 0: stream from global memory.
template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ pts_all,
                                                          const int* __restrict__ offsets, int sn, int start,
                                                          int* __restrict__ idxs_all, float* __restrict__ md_all)
{
    __shared__ float s_v[17];
    __shared__ int s_i[17];
    __shared__ float s_red[6][16];
    const int obj = blockIdx.x;
    const int p0 = offsets[obj], pn = offsets[obj + 1] - p0;
    const float* pts = pts_all + (long long)p0 * 3;
    float* md_g = md_all + p0;
    int* idxs = idxs_all + (long long)obj * sn;
    const int tid = threadIdx.x;
    if (pn <= 0) return;

    constexpr int NR = PPT > 0 ? PPT : 1;
    float rx[NR], ry[NR], rz[NR], rmd[NR];
    if (PPT > 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int i = tid + j * FPS_THREADS;
            const bool ok = i < pn;
            rx[j] = ok ? pts[i * 3 + 0] : 0.f;
            ry[j] = ok ? pts[i * 3 + 1] : 0.f;
            rz[j] = ok ? pts[i * 3 + 2] : 0.f;
            rmd[j] = ok ? FLT_MAX : -1.f;  // out-of-range slots behave like selected points
        }
    } else {
        for (int i = tid; i < pn; i += FPS_THREADS) md_g[i] = FLT_MAX;
    }

    int cur;
    if (start < 0) {
        // bounding-box centre (:131-138), then min_dist = ||p - c||^2 (:140-141)
        float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        if (PPT > 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j)
                if (tid + j * FPS_THREADS < pn) {
                    mx[0] = fmaxf(mx[0], rx[j]); mx[1] = fmaxf(mx[1], ry[j]); mx[2] = fmaxf(mx[2], rz[j]);
                    mn[0] = fminf(mn[0], rx[j]); mn[1] = fminf(mn[1], ry[j]); mn[2] = fminf(mn[2], rz[j]);
                }
        } else {
            for (int i = tid; i < pn; i += FPS_THREADS)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = pts[(long long)i * 3 + c];
                    mx[c] = fmaxf(mx[c], v);
                    mn[c] = fminf(mn[c], v);
                }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
                mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            }
        if ((tid & 63) == 0)
#pragma unroll
            for (int c = 0; c < 3; ++c) { s_red[c][tid >> 6] = mx[c]; s_red[3 + c][tid >> 6] = mn[c]; }
        __syncthreads();
        float ctr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = s_red[c][0], b = s_red[3 + c][0];
            for (int w = 1; w < FPS_THREADS / 64; ++w) { a = fmaxf(a, s_red[c][w]); b = fminf(b, s_red[3 + c][w]); }
            ctr[c] = __fmul_rn(__fadd_rn(a, b), 0.5f);
        }
        float bv = 0.f;
        int bi = INT_MAX;
        if (PPT > 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j)
                if (tid + j * FPS_THREADS < pn) {
                    const float dd = fps_sqdist(rx[j], ry[j], rz[j], ctr[0], ctr[1], ctr[2]);
                    rmd[j] = fminf(dd, FLT_MAX);
                    if (rmd[j] > bv) { bv = rmd[j]; bi = tid + j * FPS_THREADS; }
                }
        } else {
            for (int i = tid; i < pn; i += FPS_THREADS) {
                const float dd = fminf(fps_sqdist(pts[(long long)i * 3], pts[(long long)i * 3 + 1], pts[(long long)i * 3 + 2],
                                                  ctr[0], ctr[1], ctr[2]), FLT_MAX);
                md_g[i] = dd;
                if (dd > bv) { bv = dd; bi = i; }
            }
        }
        cur = fps_block_argmax(bv, bi, s_v, s_i);
    } else {
        cur = start % pn;
    }

    for (int it = 0; it < sn; ++it) {
        if (tid == 0) idxs[it] = cur;
        if (it == sn - 1) break;
        const float cx = pts[(long long)cur * 3 + 0], cy = pts[(long long)cur * 3 + 1], cz = pts[(long long)cur * 3 + 2];
        float bv = 0.f;
        int bi = INT_MAX;
        if (PPT > 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int i = tid + j * FPS_THREADS;
                if (i == cur) rmd[j] = -1.f;  // mask[cur] = true
                const float dd = fps_sqdist(rx[j], ry[j], rz[j], cx, cy, cz);
                if (dd < rmd[j]) rmd[j] = dd;
                if (rmd[j] > bv) { bv = rmd[j]; bi = i; }
            }
        } else {
            for (int i = tid; i < pn; i += FPS_THREADS) {
                float m = md_g[i];
                if (i == cur) m = -1.f;
                const float dd = fps_sqdist(pts[(long long)i * 3], pts[(long long)i * 3 + 1], pts[(long long)i * 3 + 2], cx, cy, cz);
                if (dd < m) m = dd;
                md_g[i] = m;
                if (m > bv) { bv = m; bi = i; }
            }
        }
        cur = fps_block_argmax(bv, bi, s_v, s_i);
    }
}

// ------------------------------------------------------------------------------------------------
// Clouds of more than 16 384 points: the single-workgroup kernel streams them from L2 every iteration (one CU's bandwidth).  This form
// gives a cloud G = ceil(pn / 16 384) <= 16 workgroups, each with ITS slice of the points and their running minimum distance in
// registers; an iteration is a local arg-max, one 64-bit record (value bits << 32 | index) per workgroup in a double-buffered slot
// array, a counter barrier across the cloud's workgroups (agent-scope atomics; a monotonic counter, never reset) and the same
// (value desc, index asc) merge of the G records by every workgroup - the indices are those of the single-workgroup kernel, bit for
// bit.  All G workgroups of every cloud must be resident at once (the launcher keeps G * clouds <= 128 on the 256 CUs); a barrier that
// does not complete within ~10 ms raises the error word in the workspace instead of hanging the device.
struct FpsMultiWs {
    int counter, error, pad[2];
    unsigned long long slot[2][16];
    int bbox[16][8];  // float bits: max xyz, min xyz
};

__device__ __forceinline__ void fps_grid_sync(FpsMultiWs* ws, const int target, bool& dead)
{
    __syncthreads();
    if (threadIdx.x == 0 && !dead) {
        __hip_atomic_fetch_add(&ws->counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&ws->counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 18) || __hip_atomic_load(&ws->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&ws->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    if (__hip_atomic_load(&ws->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
}

__global__ __launch_bounds__(FPS_THREADS) void fps_multi_kernel(const float* __restrict__ pts_all, const int* __restrict__ offsets, int sn,
                                                                int start, int* __restrict__ idxs_all, FpsMultiWs* __restrict__ ws_all)
{
    constexpr int PPT = 16;
    __shared__ float s_v[17];
    __shared__ int s_i[17];
    __shared__ float s_red[6][16];
    const int obj = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    const int p0 = offsets[obj], pn = offsets[obj + 1] - p0;
    const float* pts = pts_all + (long long)p0 * 3;
    int* idxs = idxs_all + (long long)obj * sn;
    FpsMultiWs* ws = ws_all + obj;
    const int tid = threadIdx.x;
    if (pn <= 0) return;  // (every workgroup of the cloud)
    const int base = g * PPT * FPS_THREADS;
    bool dead = false;
    int phase = 0;

    float rx[PPT], ry[PPT], rz[PPT], rmd[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = base + tid + j * FPS_THREADS;
        const bool ok = i < pn;
        rx[j] = ok ? pts[(long long)i * 3 + 0] : 0.f;
        ry[j] = ok ? pts[(long long)i * 3 + 1] : 0.f;
        rz[j] = ok ? pts[(long long)i * 3 + 2] : 0.f;
        rmd[j] = ok ? FLT_MAX : -1.f;
    }
    // the cloud-wide winner of this workgroup's (bv, bi): publish, barrier, merge
    auto merge = [&](float bv, int bi) -> int {
        const int w = fps_block_argmax(bv, bi, s_v, s_i);  // (returns the index; the value is s_v-reduced in wave 0: recompute below)
        // the winning VALUE of this workgroup: every thread knows the index; the owner publishes
        if (tid == 0) s_v[16] = 0.f;
        __syncthreads();
        if (bi == w && bv > 0.f) s_v[16] = bv;  // exactly one thread holds index w as its candidate when any value is > 0
        __syncthreads();
        const int buf = phase & 1;
        if (tid == 0) {
            const unsigned long long rec = ((unsigned long long)__float_as_uint(s_v[16]) << 32) | (unsigned)w;
            __hip_atomic_store(&ws->slot[buf][g], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++phase;
        fps_grid_sync(ws, G * phase, dead);
        if (tid == 0) {
            float v = 0.f;
            int idx = INT_MAX;
            for (int q = 0; q < G; ++q) {
                const unsigned long long r = __hip_atomic_load(&ws->slot[buf][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float ov = __uint_as_float((unsigned)(r >> 32));
                const int oi = (int)(unsigned)r;
                if (ov > v || (ov == v && ov > 0.f && oi < idx)) { v = ov; idx = oi; }
            }
            s_i[16] = (idx == INT_MAX || dead) ? 0 : idx;
        }
        __syncthreads();
        return s_i[16];
    };

    int cur;
    if (start < 0) {
        float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (base + tid + j * FPS_THREADS < pn) {
                mx[0] = fmaxf(mx[0], rx[j]); mx[1] = fmaxf(mx[1], ry[j]); mx[2] = fmaxf(mx[2], rz[j]);
                mn[0] = fminf(mn[0], rx[j]); mn[1] = fminf(mn[1], ry[j]); mn[2] = fminf(mn[2], rz[j]);
            }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
                mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            }
        if ((tid & 63) == 0)
#pragma unroll
            for (int c = 0; c < 3; ++c) { s_red[c][tid >> 6] = mx[c]; s_red[3 + c][tid >> 6] = mn[c]; }
        __syncthreads();
        if (tid < 6) {
            float a = s_red[tid][0];
            for (int w = 1; w < FPS_THREADS / 64; ++w) a = tid < 3 ? fmaxf(a, s_red[tid][w]) : fminf(a, s_red[tid][w]);
            __hip_atomic_store(&ws->bbox[g][tid], (int)__float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++phase;
        fps_grid_sync(ws, G * phase, dead);
        if (tid < 6) {
            float a = tid < 3 ? -FLT_MAX : FLT_MAX;
            for (int q = 0; q < G; ++q) {
                const float b = __uint_as_float((unsigned)__hip_atomic_load(&ws->bbox[q][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                a = tid < 3 ? fmaxf(a, b) : fminf(a, b);
            }
            s_red[tid][0] = a;
        }
        __syncthreads();
        float ctr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) ctr[c] = __fmul_rn(__fadd_rn(s_red[c][0], s_red[3 + c][0]), 0.5f);
        float bv = 0.f;
        int bi = INT_MAX;
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (base + tid + j * FPS_THREADS < pn) {
                const float dd = fps_sqdist(rx[j], ry[j], rz[j], ctr[0], ctr[1], ctr[2]);
                rmd[j] = fminf(dd, FLT_MAX);
                if (rmd[j] > bv) { bv = rmd[j]; bi = base + tid + j * FPS_THREADS; }
            }
        cur = merge(bv, bi);
    } else {
        cur = start % pn;
    }

    for (int it = 0; it < sn; ++it) {
        if (tid == 0 && g == 0) idxs[it] = dead ? -1 : cur;
        if (it == sn - 1) break;
        const float cx = pts[(long long)cur * 3 + 0], cy = pts[(long long)cur * 3 + 1], cz = pts[(long long)cur * 3 + 2];
        float bv = 0.f;
        int bi = INT_MAX;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = base + tid + j * FPS_THREADS;
            if (i == cur) rmd[j] = -1.f;
            const float dd = fps_sqdist(rx[j], ry[j], rz[j], cx, cy, cz);
            if (dd < rmd[j]) rmd[j] = dd;
            if (rmd[j] > bv) { bv = rmd[j]; bi = i; }
        }
        cur = merge(bv, bi);
    }
}

__global__ void fps_multi_init_kernel(FpsMultiWs* ws, int nobj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nobj) { ws[i].counter = 0; ws[i].error = 0; }
}

extern "C" long long rdpn6d_fps_workspace_bytes(int nobj) { return (long long)(nobj > 0 ? nobj : 0) * (long long)sizeof(FpsMultiWs); }

extern "C" int rdpn6d_fps_device(const float* d_pts, const int* d_offsets, int nobj, int max_pn, int sn, int start,
                                 int* d_idxs, float* d_mindist, void* stream)
{
    RD_REQUIRE(d_pts && d_offsets && d_idxs && d_mindist, "null pointer");
    RD_REQUIRE(nobj > 0 && max_pn > 0 && sn > 0, "shape");
    hipStream_t s = (hipStream_t)stream;
    if (max_pn <= 4 * FPS_THREADS)
        hipLaunchKernelGGL(fps_kernel<4>, dim3(nobj), dim3(FPS_THREADS), 0, s, d_pts, d_offsets, sn, start, d_idxs, d_mindist);
    else if (max_pn <= 16 * FPS_THREADS)
        hipLaunchKernelGGL(fps_kernel<16>, dim3(nobj), dim3(FPS_THREADS), 0, s, d_pts, d_offsets, sn, start, d_idxs, d_mindist);
    else
        hipLaunchKernelGGL(fps_kernel<0>, dim3(nobj), dim3(FPS_THREADS), 0, s, d_pts, d_offsets, sn, start, d_idxs, d_mindist);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// with a workspace of rdpn6d_fps_workspace_bytes(nobj) bytes: clouds of 16 385 .. 262 144 points run on ceil(max_pn / 16 384) workgroups
// each (fps_multi_kernel above) when all of them fit the chip at once; everything else as rdpn6d_fps_device.  The first int of the
// workspace entry [obj] is a barrier counter, the second an error word: non-zero after the launch = a barrier timed out (indices -1).
// CONTRACT of this asynchronous entry: the launch is an ordinary (non-cooperative) one, co-residency of a cloud's workgroups is only
// likely (G * nobj <= 128 of 256 CUs), so the caller MUST read the error words once the stream has drained and, if one is set, call
// rdpn6d_fps_device for that batch (rdpn6d_fps_host does exactly that).
extern "C" int rdpn6d_fps_device_ws(const float* d_pts, const int* d_offsets, int nobj, int max_pn, int sn, int start, int* d_idxs,
                                    float* d_mindist, void* workspace, long long workspace_bytes, void* stream)
{
    RD_REQUIRE(d_pts && d_offsets && d_idxs && d_mindist, "null pointer");
    RD_REQUIRE(nobj > 0 && max_pn > 0 && sn > 0, "shape");
    static const bool off = getenv("RDPN6D_FPS_NO_MULTI") != nullptr;  // profiling
    const int G = (max_pn + 16 * FPS_THREADS - 1) / (16 * FPS_THREADS);
    if (off || !workspace || G < 2 || G > 16 || (long long)G * nobj > 128 || workspace_bytes < rdpn6d_fps_workspace_bytes(nobj))
        return rdpn6d_fps_device(d_pts, d_offsets, nobj, max_pn, sn, start, d_idxs, d_mindist, stream);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fps_multi_init_kernel, dim3((nobj + 63) / 64), dim3(64), 0, s, (FpsMultiWs*)workspace, nobj);
    hipLaunchKernelGGL(fps_multi_kernel, dim3(G, nobj), dim3(FPS_THREADS), 0, s, d_pts, d_offsets, sn, start, d_idxs, (FpsMultiWs*)workspace);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_fps_host(const float* pts, int* idxs, int pn, int sn, int start)
{
    RD_REQUIRE(pts && idxs, "null pointer");
    RD_REQUIRE(pn > 0 && sn > 0, "pn and sn must be positive");
    float *d_pts = nullptr, *d_md = nullptr;
    int *d_off = nullptr, *d_idx = nullptr;
    void* d_ws = nullptr;
    const int off[2] = {0, pn};
    int rc = RDPN6D_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_pts, sizeof(float) * 3 * (size_t)pn)) != hipSuccess ||
        (e = hipMalloc(&d_md, sizeof(float) * (size_t)pn)) != hipSuccess || (e = hipMalloc(&d_ws, sizeof(FpsMultiWs))) != hipSuccess ||
        (e = hipMalloc(&d_off, sizeof(off))) != hipSuccess || (e = hipMalloc(&d_idx, sizeof(int) * (size_t)sn)) != hipSuccess ||
        (e = hipMemcpy(d_pts, pts, sizeof(float) * 3 * (size_t)pn, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_off, off, sizeof(off), hipMemcpyHostToDevice)) != hipSuccess) {
        rdpn6d_set_error("fps: device staging failed: %s", hipGetErrorString(e));
        rc = RDPN6D_EHIP;
    }
    if (rc == RDPN6D_OK && (e = hipMemset(d_ws, 0, sizeof(FpsMultiWs))) != hipSuccess) {
        rdpn6d_set_error("fps: workspace: %s", hipGetErrorString(e));
        rc = RDPN6D_EHIP;
    }
    if (rc == RDPN6D_OK) rc = rdpn6d_fps_device_ws(d_pts, d_off, 1, pn, sn, start, d_idx, d_md, d_ws, sizeof(FpsMultiWs), nullptr);
    int wsh[2] = {0, 0};
    if (rc == RDPN6D_OK && ((e = hipMemcpy(idxs, d_idx, sizeof(int) * (size_t)sn, hipMemcpyDeviceToHost)) != hipSuccess ||
                            (e = hipMemcpy(wsh, d_ws, sizeof(wsh), hipMemcpyDeviceToHost)) != hipSuccess)) {
        rdpn6d_set_error("fps: copy back failed: %s", hipGetErrorString(e));
        rc = RDPN6D_EHIP;
    }
    if (rc == RDPN6D_OK && getenv("RDPN6D_FPS_TEST_TIMEOUT")) {  // tests: pretend the barrier gave up (indices -1, error word set)
        wsh[1] = 1;
        for (int i = 0; i < sn; ++i) idxs[i] = -1;
    }
    if (rc == RDPN6D_OK && wsh[1] != 0) {
        // the cloud's workgroups were not resident together (another stream / process held the CUs): the spin barrier of the ordinary
        // launch gave up.  The one-workgroup-per-cloud kernel needs no co-residency and yields the same indices bit for bit: run it.
        rc = rdpn6d_fps_device(d_pts, d_off, 1, pn, sn, start, d_idx, d_md, nullptr);
        if (rc == RDPN6D_OK && (e = hipMemcpy(idxs, d_idx, sizeof(int) * (size_t)sn, hipMemcpyDeviceToHost)) != hipSuccess) {
            rdpn6d_set_error("fps: copy back failed: %s", hipGetErrorString(e));
            rc = RDPN6D_EHIP;
        }
    }
    (void)hipFree(d_pts); (void)hipFree(d_md); (void)hipFree(d_off); (void)hipFree(d_idx); (void)hipFree(d_ws);
    return rc;
}

static void fps_void_entry(float* pts, int* idxs, int pn, int sn, int start)
{
    if (rdpn6d_fps_host(pts, idxs, pn, sn, start) != RDPN6D_OK) {
        // the reference ABI has no error channel: fail loudly, never fall back to a CPU path
        fprintf(stderr, "[rdpn6d] farthest_point_sampling FAILED: %s\n", rdpn6d_last_error());
        if (idxs) for (int i = 0; i < sn; ++i) idxs[i] = -1;
    }
}

extern "C" void farthest_point_sampling_init_center(float* pts, int* idxs, int pn, int sn)
{
    fps_void_entry(pts, idxs, pn, sn, -1);
}

extern "C" void farthest_point_sampling(float* pts, int* idxs, int pn, int sn)
{
    srand((unsigned)time(0));  // same draw as the reference (:93-94)
    const int start = pn > 0 ? rand() % pn : 0;
    fps_void_entry(pts, idxs, pn, sn, start);
}
