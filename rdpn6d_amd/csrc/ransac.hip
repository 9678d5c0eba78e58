// Per-crop RANSAC + Kabsch pose solve on the RGB-D residual correspondences (gfx950).
//
// One 512-thread workgroup (8 wavefronts) per crop:
//   phase 0  min / max of the mask channel (normalisation of engine_utils.get_out_mask, L1 type)
//   phase 1  foreground selection + ORDER-PRESERVING compaction of the correspondences
//            (anchor index, q = P - delta) into LDS, 512 pixels per pass (ballot + prefix)
//   phase 2  ONE HYPOTHESIS PER WAVEFRONT: every lane redraws the same 3-point sample
//            (counter-based integer hash), builds the triad alignment, lanes < K hold the K
//            transformed anchors in registers, then the 64 lanes sweep the LDS-resident
//            correspondences (anchor fetched with a cross-lane shuffle) and the wave's inlier count
//            lands on the LDS scoreboard counts[h] next to its pose
//   phase 3  sequential scan of the scoreboard with the confidence-driven stop (IEEE double
//            multiplies only: bit-reproducible)
//   phase 4  inlier mask of the winner + Kabsch/Horn refit on its inliers (double accumulation,
//            fixed reduction tree) by closed-form quaternion eigen-solve (cyclic Jacobi, 4x4)
//
// The executable specification is oracle/ransac_oracle.c; inlier masks, counts and the winning
// hypothesis are bit-exact against it under a fixed seed (fp32 arithmetic is written operation by
// operation and this file is compiled with contraction off), the refit agrees to ~1e-7.
// No reference implementation exists for this solve: it takes the role that
// cv2.solvePnPRansac plays at lib/pysixd/misc.py:170-179 / gdrn_evaluator.py:316-435.
#include "common.h"
#include <float.h>
#include <cstdlib>

#pragma clang fp contract(off)

#define RS_THREADS 1024
#define RS_WAVES (RS_THREADS / 64)
#define RS_MAX_ITERS 256

#ifdef RDPN6D_PROBE
// probe build only (RDPN6D_PROBE=1 python -m rdpn6d_amd.build; tools/debug/ransac_phases.py): 100 MHz timestamps of workgroup 0 at the
// phase boundaries of ransac_kabsch_kernel
__device__ unsigned long long* g_rs_probe = nullptr;
extern "C" int rdpn6d_debug_ransac_probe(void* buf)
{
    RD_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rs_probe), &buf, sizeof(buf)));
    return RDPN6D_OK;
}
#define RS_STAMP(i)                                                                                    \
    do {                                                                                               \
        if (g_rs_probe && blockIdx.x == 0 && threadIdx.x == 0) g_rs_probe[i] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned rs_hash(unsigned seed, unsigned b, unsigned h, unsigned t, unsigned j)
{
    unsigned x = seed;
    x ^= b * 0x9E3779B1u;
    x ^= h * 0x85EBCA77u;
    x ^= t * 0xC2B2AE3Du;
    x ^= j * 0x27D4EB2Fu;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ bool rs_frame(const float* p0, const float* p1, const float* p2, float* u1, float* u2,
                                         float* u3)
{
    float e1[3], e2[3], w[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { e1[c] = p1[c] - p0[c]; e2[c] = p2[c] - p0[c]; }
    w[0] = e1[1] * e2[2] - e1[2] * e2[1];
    w[1] = e1[2] * e2[0] - e1[0] * e2[2];
    w[2] = e1[0] * e2[1] - e1[1] * e2[0];
    float n1 = e1[0] * e1[0]; n1 = n1 + e1[1] * e1[1]; n1 = n1 + e1[2] * e1[2];
    float n2 = e2[0] * e2[0]; n2 = n2 + e2[1] * e2[1]; n2 = n2 + e2[2] * e2[2];
    float nw = w[0] * w[0]; nw = nw + w[1] * w[1]; nw = nw + w[2] * w[2];
    const float lim = 1e-10f * (n1 * n2);
    if (!(nw > lim) || !(n1 > 0.f)) return false;
    const float s1 = sqrtf(n1), sw = sqrtf(nw);
#pragma unroll
    for (int c = 0; c < 3; c++) { u1[c] = e1[c] / s1; u3[c] = w[c] / sw; }
    u2[0] = u3[1] * u1[2] - u3[2] * u1[1];
    u2[1] = u3[2] * u1[0] - u3[0] * u1[2];
    u2[2] = u3[0] * u1[1] - u3[1] * u1[0];
    return true;
}

__device__ __forceinline__ bool rs_triad(const float* a0, const float* a1, const float* a2, const float* q0,
                                         const float* q1, const float* q2, float* pose)
{
    float u1[3], u2[3], u3[3], v1[3], v2[3], v3[3];
    if (!rs_frame(a0, a1, a2, u1, u2, u3)) return false;
    if (!rs_frame(q0, q1, q2, v1, v2, v3)) return false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float r = v1[i] * u1[j];
            r = r + v2[i] * u2[j];
            r = r + v3[i] * u3[j];
            pose[i * 3 + j] = r;
        }
    float ab[3], qb[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float s = a0[c] + a1[c]; s = s + a2[c]; ab[c] = s / 3.0f;
        float u = q0[c] + q1[c]; u = u + q2[c]; qb[c] = u / 3.0f;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float r = pose[i * 3 + 0] * ab[0];
        r = r + pose[i * 3 + 1] * ab[1];
        r = r + pose[i * 3 + 2] * ab[2];
        pose[9 + i] = qb[i] - r;
    }
    return true;
}

__device__ __forceinline__ void rs_apply(const float* pose, const float* a, float* out)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float r = pose[i * 3 + 0] * a[0];
        r = r + pose[i * 3 + 1] * a[1];
        r = r + pose[i * 3 + 2] * a[2];
        out[i] = r + pose[9 + i];
    }
}

__device__ void rs_horn(const double* S, double* R)
{
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    double A[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                      {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                      {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                      {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 16; sweep++) {
        // converged?  (cyclic Jacobi converges quadratically: 5 - 7 sweeps for a 4 x 4; the remaining ones of the 16 would rotate by
        // angles below 1e-15 - on ONE lane, in software fp64 sqrt / div, they were 45 us of this kernel's 140)
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[0][3] * A[0][3] + A[1][2] * A[1][2] + A[1][3] * A[1][3] + A[2][3] * A[2][3];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2] + A[3][3] * A[3][3];
        if (off <= 1e-32 * dg) break;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    // select the eigenvector of the largest eigenvalue with compile-time indices only
    double best = A[0][0];
    double w = V[0][0], x = V[1][0], y = V[2][0], z = V[3][0];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (A[i][i] > best) { best = A[i][i]; w = V[0][i]; x = V[1][i]; y = V[2][i]; z = V[3][i]; }
    const double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// deterministic block-wide sum of NV doubles (fixed butterfly + fixed wave order); result to all threads
template <int NV>
__device__ __forceinline__ void rs_block_sum(double* v, double* s_buf /* [RS_WAVES*NV + NV] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; k++) s_buf[wave * NV + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < RS_WAVES; wv++) t += s_buf[wv * NV + k];
        v[k] = t;
    }
}

__global__ __launch_bounds__(RS_THREADS) void ransac_kabsch_kernel(
    const float* __restrict__ out_nchw, const float* __restrict__ coord2d, const float* __restrict__ fps,
    const float* __restrict__ extents, const float* __restrict__ ratios, const int* __restrict__ region_argmax, int HW,
    int K, float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed, float* __restrict__ pose_out,
    int* __restrict__ n_inliers, unsigned char* __restrict__ inlier_mask, int* __restrict__ best_hyp,
    const float* __restrict__ net_pose, float max_t_diff, int stage, int* __restrict__ g_cnt, float* __restrict__ g_pose, int mask_type)
{
    // stage 0: the whole solve in one workgroup per crop.  stage 1 (grid B x parts): selection + this part's share of the hypotheses,
    // counts / poses to the GLOBAL scoreboard g_cnt [B][RS_MAX_ITERS] / g_pose [B][RS_MAX_ITERS][12]; stage 2 (grid B): selection again,
    // the scoreboard read back, scan + refit.  Same hypotheses, same counts, same scan order: bit-identical results - the split only
    // spreads the 100 wavefront-hypotheses of a crop over `parts` CUs (B = 64 crops left three quarters of the chip idle).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // carve (all offsets multiples of 16 bytes)
    double* s_dbl = reinterpret_cast<double*>(smem_raw);                       // RS_WAVES*15 + 16 doubles
    float* s_pose = reinterpret_cast<float*>(s_dbl + RS_WAVES * 15 + 16);       // 12 * RS_MAX_ITERS
    int* s_cnt = reinterpret_cast<int*>(s_pose + 12 * RS_MAX_ITERS);            // RS_MAX_ITERS
    float* s_anchor = reinterpret_cast<float*>(s_cnt + RS_MAX_ITERS);           // 3 * 64
    int* s_misc = reinterpret_cast<int*>(s_anchor + 3 * 64);                    // 64 ints
    float* s_q = reinterpret_cast<float*>(s_misc + 64);                         // 3 * HW
    unsigned short* s_pix = reinterpret_cast<unsigned short*>(s_q + 3 * (size_t)HW);  // HW
    unsigned char* s_ai = reinterpret_cast<unsigned char*>(s_pix + HW);         // HW

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // mask_type (ROT_HEAD.MASK_LOSS_TYPE as engine_utils.get_out_mask reads it): 0 L1 min-max | 1 BCE sigmoid | 2 CE arg-max of TWO
    // mask channels - the coordinates then start at channel 2
    const int MC = mask_type == 2 ? 2 : 1;
    const int C = MC + 3 + K + 1;
    const float* m = out_nchw + (size_t)b * C * HW;
    const float* cd = coord2d + (size_t)b * 5 * HW;
    const float* A = fps + (size_t)b * K * 3;
    const float ex = extents[b * 3 + 0], ey = extents[b * 3 + 1], ez = extents[b * 3 + 2];
    const float ratio = ratios[b];
    const int* am = region_argmax + (size_t)b * HW;

    RS_STAMP(0);
    // ---- phase 0: min / max of the mask
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int p = tid; p < HW; p += RS_THREADS) { const float v = m[p]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    float* s_f = reinterpret_cast<float*>(s_misc);
    if (lane == 0) { s_f[wave] = mn; s_f[RS_WAVES + wave] = mx; }
    for (int i = tid; i < 3 * K; i += RS_THREADS) s_anchor[i] = A[i];
    if (inlier_mask && stage != 1) for (int p = tid; p < HW; p += RS_THREADS) inlier_mask[(size_t)b * HW + p] = 0;
    __syncthreads();
    mn = s_f[0]; mx = s_f[RS_WAVES];
#pragma unroll
    for (int wv = 1; wv < RS_WAVES; wv++) { mn = fminf(mn, s_f[wv]); mx = fmaxf(mx, s_f[RS_WAVES + wv]); }
    const float range = mx - mn;
    __syncthreads();

    RS_STAMP(1);
    // ---- phase 1: ordered compaction, RS_THREADS pixels per pass
    int n = 0;
    for (int base = 0; base < HW; base += RS_THREADS) {
        const int p = base + tid;
        bool sel = false;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (p < HW) {
            float nm;
            if (mask_type == 0) nm = (m[p] - mn) / range;
            else if (mask_type == 1) nm = 1.f / (1.f + expf(-m[p]));
            else nm = m[HW + p] > m[p] ? 1.f : 0.f;
            const float dz = cd[2 * HW + p];
            sel = nm > mask_thr && dz > 0.f;
            if (sel) {
                const float Px = cd[p] * ratio, Py = cd[HW + p] * ratio, Pz = dz * ratio;
                const float dlx = (m[(size_t)MC * HW + p] - 0.5f) * ex, dly = (m[(size_t)(MC + 1) * HW + p] - 0.5f) * ey,
                            dlz = (m[(size_t)(MC + 2) * HW + p] - 0.5f) * ez;
                qx = Px - dlx; qy = Py - dly; qz = Pz - dlz;
            }
        }
        const unsigned long long bal = __ballot(sel);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_misc[2 * RS_WAVES + wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < RS_WAVES; wv++) { const int c = s_misc[2 * RS_WAVES + wv]; woff += wv < wave ? c : 0; tot += c; }
        if (sel) {
            const int pos = n + woff + before;
            s_q[3 * pos] = qx; s_q[3 * pos + 1] = qy; s_q[3 * pos + 2] = qz;
            s_ai[pos] = (unsigned char)am[p];
            s_pix[pos] = (unsigned short)p;
        }
        n += tot;
        __syncthreads();
    }

    RS_STAMP(2);
    const float thr2 = inlier_thr * inlier_thr;
    for (int h = tid; h < iters; h += RS_THREADS) s_cnt[h] = -1;
    __syncthreads();

    int best = -1, best_cnt = 0;
    const int hstep = stage == 1 ? RS_WAVES * (int)gridDim.y : RS_WAVES;
    if (stage == 1 && n < 3) {
        for (int h = (int)blockIdx.y * RS_THREADS + tid; h < iters; h += RS_THREADS * (int)gridDim.y) g_cnt[(size_t)b * RS_MAX_ITERS + h] = -1;
        return;
    }
    if (stage == 2 && n >= 3) {
        for (int h = tid; h < iters; h += RS_THREADS) s_cnt[h] = g_cnt[(size_t)b * RS_MAX_ITERS + h];
        __syncthreads();
    }
    if (n >= 3) {
        // ---- phase 2: one hypothesis per wavefront
        for (int h = (stage == 1 ? (int)blockIdx.y * RS_WAVES : 0) + wave; h < iters && stage != 2; h += hstep) {
            float pose[12];
            bool ok = false;
            if (net_pose && h == 0) {  // hypothesis 0 = the network's own pose (process_net_and_pnp: extrinsic guess)
#pragma unroll
                for (int i = 0; i < 12; i++) pose[i] = net_pose[b * 12 + i];
                ok = true;
            }
            for (int t = 0; t < 8 && !ok; t++) {
                const int i0 = (int)(rs_hash(seed, b, h, t, 0) % (unsigned)n);
                const int i1 = (int)(rs_hash(seed, b, h, t, 1) % (unsigned)n);
                const int i2 = (int)(rs_hash(seed, b, h, t, 2) % (unsigned)n);
                const int k0 = s_ai[i0], k1 = s_ai[i1], k2 = s_ai[i2];
                if (k0 == k1 || k0 == k2 || k1 == k2) continue;
                float a0[3], a1[3], a2[3], q0[3], q1[3], q2[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    a0[c] = s_anchor[3 * k0 + c]; a1[c] = s_anchor[3 * k1 + c]; a2[c] = s_anchor[3 * k2 + c];
                    q0[c] = s_q[3 * i0 + c]; q1[c] = s_q[3 * i1 + c]; q2[c] = s_q[3 * i2 + c];
                }
                ok = rs_triad(a0, a1, a2, q0, q1, q2, pose);
            }
            if (!ok) {  // wave-uniform
                if (stage == 1 && lane == 0) g_cnt[(size_t)b * RS_MAX_ITERS + h] = -1;
                continue;
            }
            // lanes < K hold the transformed anchor of region `lane`
            float ta[3] = {0.f, 0.f, 0.f};
            if (lane < K) {
                const float ak[3] = {s_anchor[3 * lane], s_anchor[3 * lane + 1], s_anchor[3 * lane + 2]};
                rs_apply(pose, ak, ta);
            }
            int cnt = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const bool in = i < n;
                const int k = in ? s_ai[i] : 0;
                const float tx = __shfl(ta[0], k), ty = __shfl(ta[1], k), tz = __shfl(ta[2], k);
                if (in) {
                    const float r0 = tx - s_q[3 * i], r1 = ty - s_q[3 * i + 1], r2 = tz - s_q[3 * i + 2];
                    float d2 = r0 * r0; d2 = d2 + r1 * r1; d2 = d2 + r2 * r2;
                    cnt += d2 < thr2 ? 1 : 0;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0) {
                if (stage == 1) {  // the scoreboard of the split solve lives in global memory (read back by stage 2)
                    g_cnt[(size_t)b * RS_MAX_ITERS + h] = cnt;
#pragma unroll
                    for (int i = 0; i < 12; i++) g_pose[((size_t)b * RS_MAX_ITERS + h) * 12 + i] = pose[i];
                } else {
                    s_cnt[h] = cnt;  // the LDS inlier scoreboard
#pragma unroll
                    for (int i = 0; i < 12; i++) s_pose[12 * h + i] = pose[i];
                }
            }
        }
        if (stage == 1) return;
        __syncthreads();
        RS_STAMP(3);
        // ---- phase 3: scoreboard scan with the confidence-driven stop
        if (tid == 0) {
            int niters = iters;
            for (int h = 0; h < iters && h < niters; h++) {
                const int cnt = s_cnt[h];
                if (cnt > best_cnt && cnt >= 3) {
                    best = h;
                    best_cnt = cnt;
                    const double w = (double)cnt / (double)n;
                    const double miss = 1.0 - w * w * w, target = 1.0 - (double)confidence;
                    double prod = 1.0;
                    int k = 0;
                    while (prod > target && k < iters) { prod *= miss; k++; }
                    if (k < niters) niters = k;
                }
            }
            s_misc[0] = best;
            s_misc[1] = best_cnt;
        }
        __syncthreads();
        best = s_misc[0];
        best_cnt = s_misc[1];
    }
    if (tid == 0) {
        n_inliers[b] = best_cnt;
        if (best_hyp) best_hyp[b] = best;
    }
    float* po = pose_out + b * 12;
    if (best < 0) {
        // too few correspondences: the plain solve reports the -100 sentinel (gdrn_evaluator.py:408-411), the
        // network-initialised one keeps the network pose (gdrn_evaluator.py:297-300)
        if (tid < 12) po[tid] = net_pose ? net_pose[b * 12 + tid] : -100.f;
        return;
    }

    RS_STAMP(4);
    // ---- phase 4: inliers of the winner + Kabsch / Horn refit (double, fixed reduction tree)
    float pose[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pose[i] = stage == 2 ? g_pose[((size_t)b * RS_MAX_ITERS + best) * 12 + i] : s_pose[12 * best + i];
    // ONE sweep over the correspondences: first and (un-centred) second moments of the winner's inliers in double, then
    // S = sum a q^T - n abar qbar^T (coordinates of order 0.1 .. 1 m: the subtraction loses two of sixteen digits; the oracle's
    // two-pass centred form agrees to 1e-12) - half the LDS traffic and one block reduction instead of two
    double acc[15];
#pragma unroll
    for (int c = 0; c < 15; c++) acc[c] = 0.0;
    for (int i = tid; i < n; i += RS_THREADS) {
        const int k = s_ai[i];
        const float ak[3] = {s_anchor[3 * k], s_anchor[3 * k + 1], s_anchor[3 * k + 2]};
        float ta[3];
        rs_apply(pose, ak, ta);
        const float r0 = ta[0] - s_q[3 * i], r1 = ta[1] - s_q[3 * i + 1], r2 = ta[2] - s_q[3 * i + 2];
        float d2 = r0 * r0; d2 = d2 + r1 * r1; d2 = d2 + r2 * r2;
        if (d2 < thr2) {
            if (inlier_mask) inlier_mask[(size_t)b * HW + s_pix[i]] = 1;
#pragma unroll
            for (int c = 0; c < 3; c++) { acc[c] += (double)ak[c]; acc[3 + c] += (double)s_q[3 * i + c]; }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) acc[6 + r * 3 + c] += (double)ak[r] * (double)s_q[3 * i + c];
        }
    }
    rs_block_sum<15>(acc, s_dbl);
    double abar[3], qbar[3], S[9];
#pragma unroll
    for (int c = 0; c < 3; c++) { abar[c] = acc[c] / (double)best_cnt; qbar[c] = acc[3 + c] / (double)best_cnt; }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) S[r * 3 + c] = acc[6 + r * 3 + c] - (double)best_cnt * abar[r] * qbar[c];
    RS_STAMP(5);
    if (tid == 0) {
        double R[9];
        rs_horn(S, R);
        RS_STAMP(6);
#pragma unroll
        for (int i = 0; i < 9; i++) po[i] = (float)R[i];
#pragma unroll
        for (int i = 0; i < 3; i++)
            po[9 + i] = (float)(qbar[i] - (R[i * 3] * abar[0] + R[i * 3 + 1] * abar[1] + R[i * 3 + 2] * abar[2]));
        if (net_pose) {  // "translation error too large" guard (gdrn_evaluator.py:293-296): keep the network's t
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double dt = (double)po[9 + i] - (double)net_pose[b * 12 + 9 + i];
                d2 += dt * dt;
            }
            if (sqrt(d2) > (double)max_t_diff)
#pragma unroll
                for (int i = 0; i < 3; i++) po[9 + i] = net_pose[b * 12 + 9 + i];
        }
    }
}

static size_t rs_smem_bytes(int HW)
{
    size_t s = sizeof(double) * (RS_WAVES * 15 + 16) + sizeof(float) * 12 * RS_MAX_ITERS + sizeof(int) * RS_MAX_ITERS +
               sizeof(float) * 3 * 64 + sizeof(int) * 64;
    s += sizeof(float) * 3 * (size_t)HW + sizeof(unsigned short) * (size_t)HW + (size_t)HW;
    return (s + 15) & ~(size_t)15;
}

// Workspace of the split solve (global scoreboard): counts [B][RS_MAX_ITERS] int32 + poses [B][RS_MAX_ITERS][12] fp32
extern "C" long long rdpn6d_ransac_workspace_bytes(int B) { return B > 0 ? (long long)B * RS_MAX_ITERS * (4 + 48) : 0; }

static int rs_launch(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                     const float* resize_ratios, const int* region_argmax, int B, int HW, int K, float mask_thr,
                     float inlier_thr, int iters, float confidence, unsigned seed, float* pose_out, int* n_inliers,
                     unsigned char* inlier_mask, int* best_hyp, const float* net_pose, float max_t_diff, void* workspace,
                     long long workspace_bytes, void* stream, int mask_type = 0)
{
    RD_REQUIRE(out_nchw && coord2d && fps && extents && resize_ratios && region_argmax && pose_out && n_inliers, "null pointer");
    RD_REQUIRE(mask_type >= 0 && mask_type <= 2, "mask_type: 0 L1 (min-max) | 1 BCE (sigmoid) | 2 CE (arg-max of two channels)");
    RD_REQUIRE(B > 0 && HW > 0 && HW <= 16384, "HW must be in 1..16384 (LDS-resident correspondences)");
    RD_REQUIRE(K >= 3 && K <= 64, "K in 3..64");
    RD_REQUIRE(iters >= 1 && iters <= RS_MAX_ITERS, "iters in 1..256");
    RD_REQUIRE(inlier_thr > 0.f && confidence > 0.f && confidence < 1.f, "thresholds");
    RD_REQUIRE(pose_out != net_pose, "pose_out must not alias net_pose");
    const size_t smem = rs_smem_bytes(HW);
    RD_REQUIRE(smem <= 160 * 1024, "correspondences do not fit the 160 KiB LDS");
    RD_LDS_OPT_IN(ransac_kabsch_kernel, 160 * 1024);
    // Split solve: with fewer crops than CUs the hypotheses of a crop are spread over `parts` workgroups (stage 1), a second launch scans
    // the global scoreboard and refits (stage 2).  Bit-identical to the one-workgroup form (same hypotheses, counts, scan order).
    int parts = 256 / B;
    parts = parts > 4 ? 4 : parts;
    const int rounds = (iters + RS_WAVES - 1) / RS_WAVES;
    parts = parts > rounds ? rounds : parts;
    static const int no_split = getenv("RDPN6D_RANSAC_NO_SPLIT") ? 1 : 0;  // profiling
    if (parts >= 2 && !no_split && workspace && workspace_bytes >= rdpn6d_ransac_workspace_bytes(B)) {
        int* g_cnt = reinterpret_cast<int*>(workspace);
        float* g_pose = reinterpret_cast<float*>(g_cnt + (size_t)B * RS_MAX_ITERS);
        hipLaunchKernelGGL(ransac_kabsch_kernel, dim3(B, parts), dim3(RS_THREADS), smem, (hipStream_t)stream, out_nchw, coord2d, fps,
                           extents, resize_ratios, region_argmax, HW, K, mask_thr, inlier_thr, iters, confidence, seed,
                           pose_out, n_inliers, inlier_mask, best_hyp, net_pose, max_t_diff, 1, g_cnt, g_pose, mask_type);
        RD_LAUNCH_CHECK();
        hipLaunchKernelGGL(ransac_kabsch_kernel, dim3(B), dim3(RS_THREADS), smem, (hipStream_t)stream, out_nchw, coord2d, fps,
                           extents, resize_ratios, region_argmax, HW, K, mask_thr, inlier_thr, iters, confidence, seed,
                           pose_out, n_inliers, inlier_mask, best_hyp, net_pose, max_t_diff, 2, g_cnt, g_pose, mask_type);
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    hipLaunchKernelGGL(ransac_kabsch_kernel, dim3(B), dim3(RS_THREADS), smem, (hipStream_t)stream, out_nchw, coord2d, fps,
                       extents, resize_ratios, region_argmax, HW, K, mask_thr, inlier_thr, iters, confidence, seed,
                       pose_out, n_inliers, inlier_mask, best_hyp, net_pose, max_t_diff, 0, (int*)nullptr, (float*)nullptr, mask_type);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_ransac_kabsch_ex(const float* out_nchw, const float* coord2d, const float* fps,
                                       const float* extents, const float* resize_ratios, const int* region_argmax, int B,
                                       int HW, int K, float mask_thr, float inlier_thr, int iters, float confidence,
                                       unsigned seed, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                                       int* best_hyp, void* stream)
{
    return rs_launch(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax, B, HW, K, mask_thr, inlier_thr, iters,
                     confidence, seed, pose_out, n_inliers, inlier_mask, best_hyp, nullptr, 0.f, nullptr, 0, stream);
}

// The same solves with a caller-provided workspace (rdpn6d_ransac_workspace_bytes(B) bytes): enables the split form above.
// net_pose == NULL: the plain solve (mode ignored); else mode 1 / 2 as rdpn6d_ransac_kabsch_net_f32.
extern "C" int rdpn6d_ransac_kabsch_ws(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                                       const float* resize_ratios, const int* region_argmax, const float* net_pose, int B, int HW, int K,
                                       float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed, int mode,
                                       float max_t_diff, float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp,
                                       void* workspace, long long workspace_bytes, void* stream)
{
    return rdpn6d_ransac_kabsch_ws_mt(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax, net_pose, B, HW, K, mask_thr, 0,
                                      inlier_thr, iters, confidence, seed, mode, max_t_diff, pose_out, n_inliers, inlier_mask, best_hyp,
                                      workspace, workspace_bytes, stream);
}

// ... and with the mask read as ROT_HEAD.MASK_LOSS_TYPE prescribes (engine_utils.get_out_mask): mask_type 0 L1 = per-crop min-max (every
// other entry point), 1 BCE = sigmoid, 2 CE = arg-max over two mask channels (out_nchw is then [B, 2 + 3 + K + 1, HW])
extern "C" int rdpn6d_ransac_kabsch_ws_mt(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                                          const float* resize_ratios, const int* region_argmax, const float* net_pose, int B, int HW,
                                          int K, float mask_thr, int mask_type, float inlier_thr, int iters, float confidence,
                                          unsigned seed, int mode, float max_t_diff, float* pose_out, int* n_inliers,
                                          unsigned char* inlier_mask, int* best_hyp, void* workspace, long long workspace_bytes,
                                          void* stream)
{
    if (net_pose) {
        RD_REQUIRE(mode == 1 || mode == 2, "mode: 1 = net + RANSAC, 2 = net + least-squares over all points");
        RD_REQUIRE(max_t_diff > 0.f, "max_t_diff");
        if (mode == 2) {
            iters = 1;
            inlier_thr = __builtin_huge_valf();
        }
    }
    return rs_launch(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax, B, HW, K, mask_thr, inlier_thr, iters, confidence,
                     seed, pose_out, n_inliers, inlier_mask, best_hyp, net_pose, net_pose ? max_t_diff : 0.f, workspace, workspace_bytes,
                     stream, mask_type);
}

// Network-initialised solve = process_net_and_pnp (gdrn_evaluator.py:187-314).  net_pose [B,12] (R row-major | t) is the
// learned pose.  mode 1 ("ransac"): it enters as hypothesis 0 next to iters-1 sampled ones, then the inlier refit;
// mode 2 ("iter", the reference's solvePnP(ITERATIVE) = a least-squares fit over ALL selected points): one closed-form
// Kabsch fit over all correspondences.  Both keep the network pose when fewer than 3 correspondences survive, and
// the network translation when the solved one moved by more than max_t_diff (1 m in the reference).
extern "C" int rdpn6d_ransac_kabsch_net_f32(const float* out_nchw, const float* coord2d, const float* fps,
                                            const float* extents, const float* resize_ratios, const int* region_argmax,
                                            const float* net_pose, int B, int HW, int K, float mask_thr, float inlier_thr,
                                            int iters, float confidence, unsigned seed, int mode, float max_t_diff,
                                            float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp,
                                            void* stream)
{
    RD_REQUIRE(net_pose, "null pointer");
    RD_REQUIRE(mode == 1 || mode == 2, "mode: 1 = net + RANSAC, 2 = net + least-squares over all points");
    RD_REQUIRE(max_t_diff > 0.f, "max_t_diff");
    if (mode == 2) {
        iters = 1;
        inlier_thr = __builtin_huge_valf();  // every selected correspondence counts
    }
    return rs_launch(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax, B, HW, K, mask_thr, inlier_thr, iters,
                     confidence, seed, pose_out, n_inliers, inlier_mask, best_hyp, net_pose, max_t_diff, nullptr, 0, stream);
}

extern "C" int rdpn6d_ransac_kabsch_f32(const float* out_nchw, const float* coord2d, const float* fps,
                                        const float* extents, const float* resize_ratios, const int* region_argmax, int B,
                                        int HW, int K, float mask_thr, float inlier_thr, int iters, float confidence,
                                        unsigned seed, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                                        void* stream)
{
    return rdpn6d_ransac_kabsch_ex(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax, B, HW, K, mask_thr,
                                   inlier_thr, iters, confidence, seed, pose_out, n_inliers, inlier_mask, nullptr, stream);
}
