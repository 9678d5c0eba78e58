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
    unsigned char* __restrict__ sel_mask, float* __restrict__ out_mask, int mask_type)
{
    // mask_type = ROT_HEAD.MASK_LOSS_TYPE as get_out_mask reads it (engine_utils.py:118-136): 0 "L1" per-crop min-max, 1 "BCE" sigmoid,
    // 2 "CE" arg-max over the TWO mask channels (the map tensor is then [mask0 mask1 | x y z | ...]: the coordinates start at channel 2)
    const int M = mask_type == 2 ? 2 : 1;
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
            float nm;
            if (mask_type == 0) nm = (m[p] - mn) / range;  // 0/0 = NaN for a constant mask: every comparison below is then false
            else if (mask_type == 1) nm = 1.f / (1.f + expf(-m[p]));  // torch.sigmoid
            else nm = m[HW + p] > m[p] ? 1.f : 0.f;  // torch.argmax over (mask0, mask1): the first maximum wins a tie
            if (out_mask) out_mask[(size_t)b * HW + p] = nm;
            x = (m[(size_t)M * HW + p] - 0.5f) * ex;
            y = (m[(size_t)(M + 1) * HW + p] - 0.5f) * ey;
            z = (m[(size_t)(M + 2) * HW + p] - 0.5f) * ez;
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

extern "C" int rdpn6d_select_correspondences_mt_f32(const float* out_nchw, int C, const float* coord2d, int C2, int u_ch, int v_ch,
                                                    const float* extents, const int* im_hw, int im_H, int im_W, int B, int HW,
                                                    float mask_thr, int mask_type, float* image_points, float* model_points,
                                                    int* counts, unsigned char* sel_mask, float* out_mask, void* stream)
{
    RD_REQUIRE(out_nchw && coord2d && extents && image_points && model_points && counts, "null pointer");
    RD_REQUIRE(mask_type >= 0 && mask_type <= 2, "mask_type: 0 L1 (min-max) | 1 BCE (sigmoid) | 2 CE (arg-max of two channels)");
    RD_REQUIRE(B > 0 && HW > 0 && C >= (mask_type == 2 ? 5 : 4), "B, HW > 0; the map tensor holds mask | coor_x | coor_y | coor_z | ...");
    RD_REQUIRE(C2 >= 2 && u_ch >= 0 && u_ch < C2 && v_ch >= 0 && v_ch < C2, "2D-coordinate channels");
    RD_REQUIRE(im_hw || (im_H > 0 && im_W > 0), "image size");
    hipLaunchKernelGGL(select_correspondences_kernel, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, out_nchw, C, coord2d,
                       C2, u_ch, v_ch, extents, im_hw, im_H, im_W, HW, mask_thr, image_points, model_points, counts, sel_mask,
                       out_mask, mask_type);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_select_correspondences_f32(const float* out_nchw, int C, const float* coord2d, int C2, int u_ch, int v_ch,
                                                 const float* extents, const int* im_hw, int im_H, int im_W, int B, int HW,
                                                 float mask_thr, float* image_points, float* model_points, int* counts,
                                                 unsigned char* sel_mask, float* out_mask, void* stream)
{
    return rdpn6d_select_correspondences_mt_f32(out_nchw, C, coord2d, C2, u_ch, v_ch, extents, im_hw, im_H, im_W, B, HW, mask_thr, 0,
                                                image_points, model_points, counts, sel_mask, out_mask, stream);
}

// =====================================================================================================================
// 2D-3D RANSAC-PnP (rows A9 / A10): the on-device counterpart of lib/pysixd/misc.py:145-194 pnp_v2 -> cv2.solvePnPRansac as
// called from gdrn_evaluator.py:316-435 (process_pnp_ransac: reprojection threshold 3 px, 100 iterations) and :187-314
// (process_net_and_pnp: the network pose as extrinsic guess with 20 iterations, or SOLVEPNP_ITERATIVE from it).
//
// One 512-thread workgroup (8 wavefronts) per crop; the crop's correspondences (uv, xyz: 20 bytes each, n <= HW) live in LDS.
//   phase 1  correspondences HBM -> LDS
//   phase 2  ONE HYPOTHESIS PER WAVEFRONT: every lane draws the same 4 distinct correspondences (counter-based integer hash),
//            solves the Lambda-Twist P3P on three of them in fp64 and keeps the solution that reprojects the fourth best; then the
//            64 lanes sweep the LDS-resident correspondences, score them by REPROJECTION error < thr and the wave's inlier count
//            lands on the LDS scoreboard next to its pose
//   phase 3  sequential scan of the scoreboard with the confidence-driven stop (minimal set of 4)
//   phase 4  inlier mask of the winner, then PNP_REFIT_ITERS Gauss-Newton steps on the reprojection error of its inliers
//            (6 parameters, normal equations reduced over the workgroup in a fixed tree, Cayley-transform rotation update)
// The executable specification is oracle/pnp_oracle.c: fp64 arithmetic of + - * / sqrt only, written operation by operation in
// the same order (this file is compiled with contraction off), so masks, counts and the winning hypothesis are bit-exact under
// a fixed seed; the refit differs by summation order (~1e-9).  Parity with cv2 itself is UNPINNED (cv2 is not installed).
#define PNP_THREADS 512
#define PNP_WAVES (PNP_THREADS / 64)
#define PNP_MAX_ITERS 256
#define PNP_REFIT_ITERS 10
#define PNP_HUGE (__builtin_huge_val())

namespace {
__device__ static unsigned pnp_hash(unsigned seed, unsigned b, unsigned h, unsigned t, unsigned j)
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

__device__ static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* one real root of x^3 + b x^2 + c x + d by Newton-Raphson from a start on the correct side of the stationary points */
__device__ static double cubic_one_root(double b, double c, double d)
{
    double r0;
    if (b * b >= 3.0 * c) {
        const double v = sqrt(b * b - 3.0 * c);
        const double t1 = (-b - v) / 3.0;
        double k = ((t1 + b) * t1 + c) * t1 + d;
        if (k > 0.0) {
            r0 = t1 - sqrt(-k / (3.0 * t1 + b));
        } else {
            const double t2 = (-b + v) / 3.0;
            k = ((t2 + b) * t2 + c) * t2 + d;
            r0 = t2 + sqrt(-k / (3.0 * t2 + b));
        }
    } else {
        r0 = -b / 3.0;
        if (fabs((3.0 * r0 + 2.0 * b) * r0 + c) < 1e-4) r0 += 1.0;
    }
    for (int it = 0; it < 50; it++) {
        const double fx = ((r0 + b) * r0 + c) * r0 + d;
        if (it >= 7 && fabs(fx) < 1e-13) break;
        const double fpx = (3.0 * r0 + 2.0 * b) * r0 + c;
        if (fpx == 0.0) break;
        r0 -= fx / fpx;
    }
    return r0;
}

/* real roots of x^2 + b x + c; returns their number (0 or 2) */
__device__ static int quad_roots(double b, double c, double* r1, double* r2)
{
    const double disc = b * b - 4.0 * c;
    if (!(disc >= 0.0)) return 0;
    const double y = sqrt(disc);
    const double q = b < 0.0 ? 0.5 * (-b + y) : 0.5 * (-b - y);
    *r1 = q;
    *r2 = q != 0.0 ? c / q : 0.0;
    return 2;
}

__device__ static double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
/* trace(adj(A) B) for 3x3 */
__device__ static double tr_adj_mul(const double* A, const double* B)
{
    double adj[9];
    adj[0] = A[4] * A[8] - A[5] * A[7];
    adj[1] = A[2] * A[7] - A[1] * A[8];
    adj[2] = A[1] * A[5] - A[2] * A[4];
    adj[3] = A[5] * A[6] - A[3] * A[8];
    adj[4] = A[0] * A[8] - A[2] * A[6];
    adj[5] = A[2] * A[3] - A[0] * A[5];
    adj[6] = A[3] * A[7] - A[4] * A[6];
    adj[7] = A[1] * A[6] - A[0] * A[7];
    adj[8] = A[0] * A[4] - A[1] * A[3];
    double t = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t += adj[i * 3 + j] * B[j * 3 + i];
    return t;
}

/* unit eigenvector of the symmetric 3x3 matrix A for the eigenvalue e: the largest cross product of two rows of A - e I */
__device__ static int eigvec_sym3(const double* A, double e, double* v)
{
    double M[9];
    for (int i = 0; i < 9; i++) M[i] = A[i];
    M[0] -= e; M[4] -= e; M[8] -= e;
    double c01[3], c02[3], c12[3];
    cross3(M, M + 3, c01);
    cross3(M, M + 6, c02);
    cross3(M + 3, M + 6, c12);
    const double n01 = dot3(c01, c01), n02 = dot3(c02, c02), n12 = dot3(c12, c12);
    const double* best = c01;
    double nb = n01;
    if (n02 > nb) { best = c02; nb = n02; }
    if (n12 > nb) { best = c12; nb = n12; }
    if (!(nb > 0.0)) return 0;
    const double s = 1.0 / sqrt(nb);
    v[0] = best[0] * s; v[1] = best[1] * s; v[2] = best[2] * s;
    return 1;
}

/* P3P: bearings y[3][3] (unit), model points x[3][3] -> up to 4 poses (R row-major, t); returns their number */
__device__ static int p3p_lambdatwist(const double y[3][3], const double x[3][3], double R[4][9], double t[4][3])
{
    const double b12 = -2.0 * dot3(y[0], y[1]), b13 = -2.0 * dot3(y[0], y[2]), b23 = -2.0 * dot3(y[1], y[2]);
    double d12[3], d13[3], d23[3], d12xd13[3];
    for (int c = 0; c < 3; c++) { d12[c] = x[0][c] - x[1][c]; d13[c] = x[0][c] - x[2][c]; d23[c] = x[1][c] - x[2][c]; }
    cross3(d12, d13, d12xd13);
    const double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
    if (!(a12 > 0.0 && a13 > 0.0 && a23 > 0.0) || !(dot3(d12xd13, d12xd13) > 1e-24 * a12 * a13)) return 0;
    /* D1 = a23 M12 - a12 M23,  D2 = a23 M13 - a13 M23  (symmetric, row-major) */
    const double D1[9] = {a23, 0.5 * a23 * b12, 0.0, 0.5 * a23 * b12, a23 - a12, -0.5 * a12 * b23, 0.0, -0.5 * a12 * b23, -a12};
    const double D2[9] = {a23, 0.0, 0.5 * a23 * b13, 0.0, -a13, -0.5 * a13 * b23, 0.5 * a23 * b13, -0.5 * a13 * b23, a23 - a13};
    /* det(D1 + g D2) = c0 + c1 g + c2 g^2 + c3 g^3 */
    const double c3 = det3(D2), c0 = det3(D1), c1 = tr_adj_mul(D1, D2), c2 = tr_adj_mul(D2, D1);
    if (c3 == 0.0) return 0;
    const double pb = c2 / c3, pc = c1 / c3, pd = c0 / c3;
    double roots[3];
    int nroots = 1;
    roots[0] = cubic_one_root(pb, pc, pd);
    {   /* deflate: x^2 + (pb + r) x + (pc + (pb + r) r) */
        const double qb = pb + roots[0], qc = pc + qb * roots[0];
        double r1, r2;
        if (quad_roots(qb, qc, &r1, &r2)) { roots[1] = r1; roots[2] = r2; nroots = 3; }
    }
    double Ls[4][3];
    int valid = 0;
    for (int ri = 0; ri < nroots && valid == 0; ri++) {
        const double g = roots[ri];
        double A[9];
        for (int i = 0; i < 9; i++) A[i] = D1[i] + g * D2[i];
        /* the two non-zero eigenvalues: roots of e^2 - tr e + (sum of principal 2x2 minors) */
        const double tr = A[0] + A[4] + A[8];
        const double mn = (A[0] * A[4] - A[1] * A[3]) + (A[0] * A[8] - A[2] * A[6]) + (A[4] * A[8] - A[5] * A[7]);
        double e1, e2;
        if (!quad_roots(-tr, mn, &e1, &e2)) continue;
        if (fabs(e1) < fabs(e2)) { const double tmp = e1; e1 = e2; e2 = tmp; }
        if (!(e1 * e2 < 0.0)) continue;  /* not a pair of real planes for this root */
        double v1[3], v2[3];
        if (!eigvec_sym3(A, e1, v1) || !eigvec_sym3(A, e2, v2)) continue;
        const double v = sqrt(-e2 / e1);
        for (int sgn = 0; sgn < 2 && valid < 4; sgn++) {
            const double s = sgn == 0 ? v : -v;
            /* plane n . L = 0 with n = v1 + s v2  ->  l1 = w0 l2 + w1 l3 */
            const double n0 = v1[0] + s * v2[0], n1 = v1[1] + s * v2[1], n2 = v1[2] + s * v2[2];
            if (n0 == 0.0) continue;
            const double w0 = -n1 / n0, w1 = -n2 / n0;
            const double qa = (a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12;
            if (qa == 0.0) continue;
            const double qb = (a13 * b12 * w1 - a12 * b13 * w0 - 2.0 * w0 * w1 * (a12 - a13)) / qa;
            const double qc = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) / qa;
            double taus[2];
            if (!quad_roots(qb, qc, &taus[0], &taus[1])) continue;
            for (int ti = 0; ti < 2 && valid < 4; ti++) {
                const double tau = taus[ti];
                if (!(tau > 0.0)) continue;
                const double den = tau * (b23 + tau) + 1.0;
                if (!(den > 0.0)) continue;
                const double l2 = sqrt(a23 / den), l3 = tau * l2, l1 = w0 * l2 + w1 * l3;
                if (!(l1 >= 0.0)) continue;
                Ls[valid][0] = l1; Ls[valid][1] = l2; Ls[valid][2] = l3;
                valid++;
            }
        }
    }
    int nsol = 0;
    for (int k = 0; k < valid; k++) {
        double l1 = Ls[k][0], l2 = Ls[k][1], l3 = Ls[k][2];
        for (int it = 0; it < 5; it++) {  /* Newton on the three quadrics */
            const double r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
            const double r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
            const double r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
            if (fabs(r1) + fabs(r2) + fabs(r3) < 1e-10 * (a12 + a13 + a23)) break;
            const double j11 = 2.0 * l1 + b12 * l2, j12 = 2.0 * l2 + b12 * l1;
            const double j21 = 2.0 * l1 + b13 * l3, j23 = 2.0 * l3 + b13 * l1;
            const double j32 = 2.0 * l2 + b23 * l3, j33 = 2.0 * l3 + b23 * l2;
            const double det = -j11 * j23 * j32 - j12 * j21 * j33;
            if (det == 0.0) break;
            const double id = 1.0 / det;
            const double dl1 = id * (-j23 * j32 * r1 - j12 * j33 * r2 + j12 * j23 * r3);
            const double dl2 = id * (-j21 * j33 * r1 + j11 * j33 * r2 - j11 * j23 * r3);
            const double dl3 = id * (j21 * j32 * r1 - j11 * j32 * r2 - j12 * j21 * r3);
            l1 -= dl1; l2 -= dl2; l3 -= dl3;
        }
        if (!(l1 > 0.0 && l2 > 0.0 && l3 > 0.0)) continue;
        /* R maps (d12, d13, d12 x d13) onto (l1 y1 - l2 y2, l1 y1 - l3 y3, their cross product); t = l1 y1 - R x1 */
        double yd1[3], yd2[3], yx[3], ry1[3];
        for (int c = 0; c < 3; c++) { ry1[c] = l1 * y[0][c]; yd1[c] = ry1[c] - l2 * y[1][c]; yd2[c] = ry1[c] - l3 * y[2][c]; }
        cross3(yd1, yd2, yx);
        const double X[9] = {d12[0], d13[0], d12xd13[0], d12[1], d13[1], d12xd13[1], d12[2], d13[2], d12xd13[2]};
        const double dX = det3(X);
        if (dX == 0.0) continue;
        const double iX = 1.0 / dX;
        double Xi[9];
        Xi[0] = (X[4] * X[8] - X[5] * X[7]) * iX; Xi[1] = (X[2] * X[7] - X[1] * X[8]) * iX; Xi[2] = (X[1] * X[5] - X[2] * X[4]) * iX;
        Xi[3] = (X[5] * X[6] - X[3] * X[8]) * iX; Xi[4] = (X[0] * X[8] - X[2] * X[6]) * iX; Xi[5] = (X[2] * X[3] - X[0] * X[5]) * iX;
        Xi[6] = (X[3] * X[7] - X[4] * X[6]) * iX; Xi[7] = (X[1] * X[6] - X[0] * X[7]) * iX; Xi[8] = (X[0] * X[4] - X[1] * X[3]) * iX;
        const double Y[9] = {yd1[0], yd2[0], yx[0], yd1[1], yd2[1], yx[1], yd1[2], yd2[2], yx[2]};
        double* Rk = R[nsol];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rk[i * 3 + j] = Y[i * 3 + 0] * Xi[0 * 3 + j] + Y[i * 3 + 1] * Xi[1 * 3 + j] + Y[i * 3 + 2] * Xi[2 * 3 + j];
        for (int i = 0; i < 3; i++) t[nsol][i] = ry1[i] - (Rk[i * 3] * x[0][0] + Rk[i * 3 + 1] * x[0][1] + Rk[i * 3 + 2] * x[0][2]);
        nsol++;
    }
    return nsol;
}

/* squared reprojection error (pixels) of model point p under pose (R, t) and intrinsics; +inf behind the camera */
__device__ static double reproj_err2(const double* R, const double* t, const double* K4 /* fx fy cx cy */, const float* p, const float* uv)
{
    const double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    const double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    const double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    if (!(Z > 0.0)) return PNP_HUGE;
    const double du = K4[0] * X / Z + K4[2] - (double)uv[0];
    const double dv = K4[1] * Y / Z + K4[3] - (double)uv[1];
    return du * du + dv * dv;
}

/* hypothesis h of crop b: sample 4 distinct correspondences, P3P on the first three, keep the solution with the smallest
 * reprojection error of the fourth.  Returns 0 if no pose came out. */
__device__ static int pnp_hypothesis(unsigned seed, unsigned b, unsigned h, int n, const float* ip, const float* mp, const double* K4, double* R, double* t)
{
    for (unsigned tr = 0; tr < 8; tr++) {
        int idx[4];
        for (int j = 0; j < 4; j++) idx[j] = (int)(pnp_hash(seed, b, h, tr, (unsigned)j) % (unsigned)n);
        if (idx[0] == idx[1] || idx[0] == idx[2] || idx[0] == idx[3] || idx[1] == idx[2] || idx[1] == idx[3] || idx[2] == idx[3]) continue;
        double y[3][3], x[3][3];
        for (int j = 0; j < 3; j++) {
            const double bx = ((double)ip[2 * idx[j]] - K4[2]) / K4[0], by = ((double)ip[2 * idx[j] + 1] - K4[3]) / K4[1];
            const double inv = 1.0 / sqrt(bx * bx + by * by + 1.0);
            y[j][0] = bx * inv; y[j][1] = by * inv; y[j][2] = inv;
            for (int c = 0; c < 3; c++) x[j][c] = (double)mp[3 * idx[j] + c];
        }
        double Rs[4][9], ts[4][3];
        const int ns = p3p_lambdatwist(y, x, Rs, ts);
        int bestk = -1;
        double beste = PNP_HUGE;
        for (int k = 0; k < ns; k++) {
            const double e = reproj_err2(Rs[k], ts[k], K4, mp + 3 * idx[3], ip + 2 * idx[3]);
            if (e < beste) { beste = e; bestk = k; }
        }
        if (bestk < 0) continue;
        for (int i = 0; i < 9; i++) R[i] = Rs[bestk][i];
        for (int i = 0; i < 3; i++) t[i] = ts[bestk][i];
        return 1;
    }
    return 0;
}

/* solve the symmetric positive definite 6x6 system H d = g in place (Gaussian elimination with partial pivoting); 0 if singular */
__device__ static int solve6(double H[6][6], double g[6])
{
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(H[r][c]) > fabs(H[p][c])) p = r;
        if (H[p][c] == 0.0) return 0;
        if (p != c) {
            for (int k = 0; k < 6; k++) { const double tmp = H[c][k]; H[c][k] = H[p][k]; H[p][k] = tmp; }
            const double tg = g[c]; g[c] = g[p]; g[p] = tg;
        }
        for (int r = c + 1; r < 6; r++) {
            const double f = H[r][c] / H[c][c];
            for (int k = c; k < 6; k++) H[r][k] -= f * H[c][k];
            g[r] -= f * g[c];
        }
    }
    for (int c = 5; c >= 0; c--) {
        double s = g[c];
        for (int k = c + 1; k < 6; k++) s -= H[c][k] * g[k];
        g[c] = s / H[c][c];
    }
    return 1;
}


// ---------------------------------------------------------------------------------------------------------------------
// EPnP minimal solver (cfg.TEST.PNP_MINIMAL = "epnp"; rdpn6d_ransac_pnp_ex minimal = 1): what the reference's call names -
// cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_EPNP), lib/pysixd/misc.py:170-179: minimal sets of FIVE, EPnP refit on the inliers.
// The executable specification is oracle/pnp_oracle.c (ep_* functions, restated from the EPnP paper); every function here is that
// code operation by operation.  One hypothesis per wavefront as before: the small linear algebra (3x3 / 4x4 Jacobi, 6x{3,4,5} least
// squares, Gauss-Newton on the betas, Horn) runs redundantly on all 64 lanes (uniform data); the 12x12 part does not fit registers
// and goes through a per-wavefront LDS scratch: M^T M one entry per lane, the round-robin Jacobi with the six rotation parameters of
// a step on six lanes and row / column k of a rotation on lane k (k < 12) - the oracle's per-k expressions, so eigenvalues and
// eigenvectors come out bit for bit; the three beta candidates (refinement, Horn, reprojection error) run one per lane.
#define EP_SWEEPS 30

template <int N>
__device__ static void ep_jacobi_small(double* A, double* V)
{
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < EP_SWEEPS; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < N; p++) {
            dg += A[p * N + p] * A[p * N + p];
            for (int q = p + 1; q < N; q++) off += A[p * N + q] * A[p * N + q];
        }
        if (!(off > 1e-30 * dg)) break;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (apq == 0.0) continue;
                const double app = A[p * N + p], aqq = A[q * N + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < N; k++) {
                    if (k != p && k != q) {
                        const double akp = A[k * N + p], akq = A[k * N + q];
                        const double nkp = c * akp - sn * akq, nkq = sn * akp + c * akq;
                        A[k * N + p] = nkp; A[p * N + k] = nkp;
                        A[k * N + q] = nkq; A[q * N + k] = nkq;
                    }
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - sn * vkq;
                    V[k * N + q] = sn * vkp + c * vkq;
                }
                A[p * N + p] = app - t * apq;
                A[q * N + q] = aqq + t * apq;
                A[p * N + q] = 0.0;
                A[q * N + p] = 0.0;
            }
    }
}

// the 12 x 12 form on one wavefront: A, V [144] in LDS.  ROUND-ROBIN pair order (oracle ep_jacobi, n == 12): a step has six disjoint
// pairs, whose rotation parameters - one fp64 divide / square-root chain of ~1 000 cycles each, the cost of a rotation - are computed
// by lanes 0..5 AT ONCE from the matrix at the start of the step (no rotation of the step touches another pair's three entries: the
// same numbers as computing each when its turn comes); the six rotations are then applied in the oracle's order, row / column k of a
// rotation on lane k (k < 12) with exactly its per-k expressions.  LDS operations of one wavefront execute in program order; the wave
// barriers keep the compiler from moving an access across a step another lane depends on.
__device__ static void ep_jacobi12_wave(double* A, double* V, int lane)
{
    for (int e = lane; e < 144; e += 64) V[e] = (e / 12) == (e % 12) ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int sweep = 0; sweep < EP_SWEEPS; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < 12; p++) {
            dg += A[p * 12 + p] * A[p * 12 + p];
            for (int q = p + 1; q < 12; q++) off += A[p * 12 + q] * A[p * 12 + q];
        }
        if (!(off > 1e-30 * dg)) break;
        for (int st = 0; st < 11; st++) {
            // this lane's pair (lanes >= 6 repeat pair lane % 6: harmless, their results are not read)
            const int i6 = lane % 6;
            const int ra = i6 == 0 ? 11 : (st + i6) % 11, rb = i6 == 0 ? st : (st + 11 - i6) % 11;
            const int mp = ra < rb ? ra : rb, mq = ra < rb ? rb : ra;
            const double mapq = A[mp * 12 + mq], mapp = A[mp * 12 + mp], maqq = A[mq * 12 + mq];
            double mt = 0.0, mc = 1.0, msn = 0.0;
            if (mapq != 0.0) {
                const double theta = (maqq - mapp) / (2.0 * mapq);
                mt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                mc = 1.0 / sqrt(mt * mt + 1.0);
                msn = mt * mc;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int p = __shfl(mp, j), q = __shfl(mq, j);
                const double apq = __shfl(mapq, j);
                if (apq == 0.0) continue;  // uniform
                const double app = __shfl(mapp, j), aqq = __shfl(maqq, j), t = __shfl(mt, j), c = __shfl(mc, j), sn = __shfl(msn, j);
                const int k = lane;
                if (k < 12) {
                    if (k != p && k != q) {
                        const double akp = A[k * 12 + p], akq = A[k * 12 + q];
                        const double nkp = c * akp - sn * akq, nkq = sn * akp + c * akq;
                        A[k * 12 + p] = nkp; A[p * 12 + k] = nkp;
                        A[k * 12 + q] = nkq; A[q * 12 + k] = nkq;
                    }
                    const double vkp = V[k * 12 + p], vkq = V[k * 12 + q];
                    V[k * 12 + p] = c * vkp - sn * vkq;
                    V[k * 12 + q] = sn * vkp + c * vkq;
                }
                if (lane == 0) {
                    A[p * 12 + p] = app - t * apq;
                    A[q * 12 + q] = aqq + t * apq;
                    A[p * 12 + q] = 0.0;
                    A[q * 12 + p] = 0.0;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

__device__ static int ep_lsq(int m, int n, const double* A, const double* b, double* x)
{
    double N[5][6];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int r = 0; r < m; r++) s += A[r * n + i] * A[r * n + j];
            N[i][j] = s;
        }
        double s = 0.0;
        for (int r = 0; r < m; r++) s += A[r * n + i] * b[r];
        N[i][n] = s;
    }
    double tr = 0.0;
    for (int i = 0; i < n; i++) tr += N[i][i];
    for (int i = 0; i < n; i++) N[i][i] += 1e-13 * tr;
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(N[r][c]) > fabs(N[p][c])) p = r;
        if (N[p][c] == 0.0) return 0;
        if (p != c)
            for (int k = 0; k <= n; k++) { const double tmp = N[c][k]; N[c][k] = N[p][k]; N[p][k] = tmp; }
        for (int r = c + 1; r < n; r++) {
            const double f = N[r][c] / N[c][c];
            for (int k = c; k <= n; k++) N[r][k] -= f * N[c][k];
        }
    }
    for (int c = n - 1; c >= 0; c--) {
        double s = N[c][n];
        for (int k = c + 1; k < n; k++) s -= N[c][k] * x[k];
        x[c] = s / N[c][c];
    }
    return 1;
}

__device__ static void ep_horn(const double* S, double* R)
{
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    double A[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                    Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                    Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy,
                    Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
    double V[16];
    ep_jacobi_small<4>(A, V);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (A[i * 4 + i] > A[best * 4 + best]) best = i;
    double w = V[0 * 4 + best], x = V[1 * 4 + best], y = V[2 * 4 + best], z = V[3 * 4 + best];
    const double nq = sqrt(w * w + x * x + y * y + z * z);
    w = w / nq; x = x / nq; y = y / nq; z = z / nq;
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}

struct ep_frame {
    double c0[3], U[9], ik[3], cws[4][3];
};

__device__ static void ep_alphas(const ep_frame& f, const float* p, double* a)
{
    const double d0 = (double)p[0] - f.c0[0], d1 = (double)p[1] - f.c0[1], d2 = (double)p[2] - f.c0[2];
    a[1] = (f.U[0] * d0 + f.U[3] * d1 + f.U[6] * d2) * f.ik[0];
    a[2] = (f.U[1] * d0 + f.U[4] * d1 + f.U[7] * d2) * f.ik[1];
    a[3] = (f.U[2] * d0 + f.U[5] * d1 + f.U[8] * d2) * f.ik[2];
    a[0] = 1.0 - a[1] - a[2] - a[3];
}

__device__ static int ep_frame_from_moments(int cnt, const double* sp, const double* spp, ep_frame& f)
{
    const double inv = 1.0 / (double)cnt;
    for (int c = 0; c < 3; c++) f.c0[c] = sp[c] * inv;
    double C[9];
    C[0] = spp[0] - sp[0] * f.c0[0]; C[1] = spp[1] - sp[0] * f.c0[1]; C[2] = spp[2] - sp[0] * f.c0[2];
    C[4] = spp[3] - sp[1] * f.c0[1]; C[5] = spp[4] - sp[1] * f.c0[2]; C[8] = spp[5] - sp[2] * f.c0[2];
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    ep_jacobi_small<3>(C, f.U);
    double lmax = C[0] > C[4] ? C[0] : C[4];
    if (C[8] > lmax) lmax = C[8];
    if (!(lmax > 0.0)) return 0;
    for (int i = 0; i < 3; i++) {
        const double lam = C[i * 3 + i];
        if (!(lam > 1e-12 * lmax)) return 0;
        const double k = sqrt(lam * inv);
        f.ik[i] = 1.0 / k;
        for (int c = 0; c < 3; c++) f.cws[i + 1][c] = f.c0[c] + k * f.U[c * 3 + i];
    }
    for (int c = 0; c < 3; c++) f.cws[0][c] = f.c0[c];
    return 1;
}

// element `col` of the two rows of M of one correspondence (oracle ep_accumulate_mtm's r1 / r2)
__device__ __forceinline__ void ep_row_elem(const double* a, double u, double v, const double* K4, int col, double& r1, double& r2)
{
    const int j = col / 3, c = col - 3 * j;
    const double aj = a[j];
    r1 = c == 0 ? aj * K4[0] : (c == 1 ? 0.0 : aj * (K4[2] - u));
    r2 = c == 0 ? 0.0 : (c == 1 ? aj * K4[1] : aj * (K4[3] - v));
}

// from the null-space vectors vv[4][12] (LDS or registers) to the three beta candidates (oracle ep_betas after the Jacobi step)
__device__ static void ep_betas_from_null(const double* vv /* [4][12] */, const ep_frame& f, double betas[3][4])
{
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double L[6][10], rho[6];
    for (int j = 0; j < 6; j++) {
        double dv[4][3];
        for (int k = 0; k < 4; k++)
            for (int c = 0; c < 3; c++) dv[k][c] = vv[k * 12 + 3 * pa[j] + c] - vv[k * 12 + 3 * pb[j] + c];
        L[j][0] = dot3(dv[0], dv[0]);
        L[j][1] = 2.0 * dot3(dv[0], dv[1]);
        L[j][2] = dot3(dv[1], dv[1]);
        L[j][3] = 2.0 * dot3(dv[0], dv[2]);
        L[j][4] = 2.0 * dot3(dv[1], dv[2]);
        L[j][5] = dot3(dv[2], dv[2]);
        L[j][6] = 2.0 * dot3(dv[0], dv[3]);
        L[j][7] = 2.0 * dot3(dv[1], dv[3]);
        L[j][8] = 2.0 * dot3(dv[2], dv[3]);
        L[j][9] = dot3(dv[3], dv[3]);
        double d[3];
        for (int c = 0; c < 3; c++) d[c] = f.cws[pa[j]][c] - f.cws[pb[j]][c];
        rho[j] = dot3(d, d);
    }
    double Lsub[30], b5[5];
    const int c4[4] = {0, 1, 3, 6};
    for (int j = 0; j < 6; j++)
        for (int k = 0; k < 4; k++) Lsub[j * 4 + k] = L[j][c4[k]];
    for (int k = 0; k < 4; k++) betas[0][k] = 0.0;
    if (ep_lsq(6, 4, Lsub, rho, b5)) {
        if (b5[0] < 0.0) { betas[0][0] = sqrt(-b5[0]); betas[0][1] = -b5[1] / betas[0][0]; betas[0][2] = -b5[2] / betas[0][0]; betas[0][3] = -b5[3] / betas[0][0]; }
        else { betas[0][0] = sqrt(b5[0]); betas[0][1] = b5[1] / betas[0][0]; betas[0][2] = b5[2] / betas[0][0]; betas[0][3] = b5[3] / betas[0][0]; }
    }
    for (int j = 0; j < 6; j++)
        for (int k = 0; k < 3; k++) Lsub[j * 3 + k] = L[j][k];
    for (int k = 0; k < 4; k++) betas[1][k] = 0.0;
    if (ep_lsq(6, 3, Lsub, rho, b5)) {
        if (b5[0] < 0.0) { betas[1][0] = sqrt(-b5[0]); betas[1][1] = b5[2] < 0.0 ? sqrt(-b5[2]) : 0.0; }
        else { betas[1][0] = sqrt(b5[0]); betas[1][1] = b5[2] > 0.0 ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0.0) betas[1][0] = -betas[1][0];
    }
    for (int j = 0; j < 6; j++)
        for (int k = 0; k < 5; k++) Lsub[j * 5 + k] = L[j][k];
    for (int k = 0; k < 4; k++) betas[2][k] = 0.0;
    if (ep_lsq(6, 5, Lsub, rho, b5)) {
        if (b5[0] < 0.0) { betas[2][0] = sqrt(-b5[0]); betas[2][1] = b5[2] < 0.0 ? sqrt(-b5[2]) : 0.0; }
        else { betas[2][0] = sqrt(b5[0]); betas[2][1] = b5[2] > 0.0 ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0.0) betas[2][0] = -betas[2][0];
        betas[2][2] = betas[2][0] != 0.0 ? b5[3] / betas[2][0] : 0.0;
    }
    // five Gauss-Newton steps per candidate: the three candidates are independent and run the same code - lane l refines candidate l % 3
    // (the oracle's loop body on that candidate's numbers), then every lane collects all three from lanes 0, 1, 2
    {
        const int cnd = (int)(threadIdx.x & 63) % 3;
        double b[4];
        for (int k = 0; k < 4; k++) b[k] = cnd == 0 ? betas[0][k] : (cnd == 1 ? betas[1][k] : betas[2][k]);
        for (int it = 0; it < 5; it++) {
            double J[24], r[6], dx[4];
            for (int j = 0; j < 6; j++) {
                const double* l = L[j];
                J[j * 4 + 0] = 2.0 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                J[j * 4 + 1] = l[1] * b[0] + 2.0 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                J[j * 4 + 2] = l[3] * b[0] + l[4] * b[1] + 2.0 * l[5] * b[2] + l[8] * b[3];
                J[j * 4 + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2.0 * l[9] * b[3];
                r[j] = rho[j] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2]
                                 + l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            if (!ep_lsq(6, 4, J, r, dx)) break;
            for (int k = 0; k < 4; k++) b[k] += dx[k];
        }
        for (int c3 = 0; c3 < 3; c3++)
            for (int k = 0; k < 4; k++) betas[c3][k] = __shfl(b[k], c3);
    }
}

// after the Jacobi step on the wavefront's scratch: the four smallest eigenvalues' vectors go to rows 0..3 of A (= vv[4][12])
__device__ static void ep_pick_null_vectors(double* A, const double* V, int lane)
{
    int sel[4];
    bool used[12];
    for (int i = 0; i < 12; i++) used[i] = false;
    for (int k = 0; k < 4; k++) {
        int best = -1;
        for (int i = 0; i < 12; i++)
            if (!used[i] && (best < 0 || A[i * 12 + i] < A[best * 12 + best])) best = i;
        used[best] = true;
        sel[k] = best;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 48) {
        const int k = lane / 12, r = lane - 12 * k;
        A[k * 12 + r] = V[r * 12 + sel[k]];
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ static void ep_ccs(const double* vv, const double* b, double ccs[4][3])
{
    for (int j = 0; j < 4; j++)
        for (int c = 0; c < 3; c++)
            ccs[j][c] = b[0] * vv[3 * j + c] + b[1] * vv[12 + 3 * j + c] + b[2] * vv[24 + 3 * j + c] + b[3] * vv[36 + 3 * j + c];
}

/* EPnP over the cnt (= 5) correspondences idx[]: one wavefront, uniform data, scratch = this wavefront's A | V [288] in LDS */
__device__ static int epnp_solve_minimal(int cnt, const int* idx, const float* ip, const float* mp, const double* K4, double* scratch, int lane,
                                         double* R, double* t)
{
    double sp[3] = {0, 0, 0}, spp[6] = {0, 0, 0, 0, 0, 0};
    for (int q = 0; q < cnt; q++) {
        const int i = idx[q];
        const double x = mp[3 * i], y = mp[3 * i + 1], z = mp[3 * i + 2];
        sp[0] += x; sp[1] += y; sp[2] += z;
        spp[0] += x * x; spp[1] += x * y; spp[2] += x * z; spp[3] += y * y; spp[4] += y * z; spp[5] += z * z;
    }
    ep_frame f;
    if (!ep_frame_from_moments(cnt, sp, spp, f)) return 0;  // uniform
    double* A = scratch;
    double* V = scratch + 144;
    // M^T M: upper-triangle entry e on lane e (and e + 64), the correspondences in order
    for (int e = lane; e < 78; e += 64) {
        int i = 0, rem = e;
        while (rem >= 12 - i) { rem -= 12 - i; i++; }
        const int j = i + rem;
        double acc = 0.0;
        for (int q = 0; q < cnt; q++) {
            double a[4], r1i, r2i, r1j, r2j;
            ep_alphas(f, mp + 3 * idx[q], a);
            const double u = (double)ip[2 * idx[q]], v = (double)ip[2 * idx[q] + 1];
            ep_row_elem(a, u, v, K4, i, r1i, r2i);
            ep_row_elem(a, u, v, K4, j, r1j, r2j);
            acc += r1i * r1j + r2i * r2j;
        }
        A[i * 12 + j] = acc;
        A[j * 12 + i] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    ep_jacobi12_wave(A, V, lane);
    ep_pick_null_vectors(A, V, lane);
    double betas[3][4];
    ep_betas_from_null(A, f, betas);
    double a1[4];
    ep_alphas(f, mp + 3 * idx[0], a1);
    // the three candidates are independent and run the same code: lane l evaluates candidate l % 3 (the oracle's loop body on that
    // candidate's numbers), then the oracle's strict "<" comparison runs over lanes 0, 1, 2 in order and the winner's pose is fetched
    double err, Rc[9], tc[3];
    {
        const int cnd = lane % 3;
        double bsel[4];
        for (int k = 0; k < 4; k++) bsel[k] = cnd == 0 ? betas[0][k] : (cnd == 1 ? betas[1][k] : betas[2][k]);
        double ccs[4][3];
        ep_ccs(A, bsel, ccs);
        const double z1 = a1[0] * ccs[0][2] + a1[1] * ccs[1][2] + a1[2] * ccs[2][2] + a1[3] * ccs[3][2];
        if (z1 < 0.0)
            for (int j = 0; j < 4; j++)
                for (int c = 0; c < 3; c++) ccs[j][c] = -ccs[j][c];
        double sc[3] = {0, 0, 0}, swc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < cnt; q++) {
            const int i = idx[q];
            double a[4], pc[3];
            ep_alphas(f, mp + 3 * i, a);
            for (int c = 0; c < 3; c++) pc[c] = a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c];
            for (int c = 0; c < 3; c++) sc[c] += pc[c];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) swc[r * 3 + c] += (double)mp[3 * i + r] * pc[c];
        }
        const double inv = 1.0 / (double)cnt;
        double S[9], cbar[3];
        for (int c = 0; c < 3; c++) cbar[c] = sc[c] * inv;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) S[r * 3 + c] = swc[r * 3 + c] - sp[r] * cbar[c];
        ep_horn(S, Rc);
        for (int r = 0; r < 3; r++) tc[r] = cbar[r] - (Rc[r * 3] * f.c0[0] + Rc[r * 3 + 1] * f.c0[1] + Rc[r * 3 + 2] * f.c0[2]);
        err = 0.0;
        for (int q = 0; q < cnt; q++) {
            const double e2 = reproj_err2(Rc, tc, K4, mp + 3 * idx[q], ip + 2 * idx[q]);
            err += e2 < PNP_HUGE ? sqrt(e2) : 1e12;
        }
    }
    int bestc = -1;
    double beste = PNP_HUGE;
    for (int c3 = 0; c3 < 3; c3++) {
        const double e = __shfl(err, c3);
        if (e < beste) { beste = e; bestc = c3; }
    }
    const int src = bestc < 0 ? 0 : bestc;
    for (int i = 0; i < 9; i++) R[i] = __shfl(Rc[i], src);
    for (int i = 0; i < 3; i++) t[i] = __shfl(tc[i], src);
    return bestc >= 0 && beste < PNP_HUGE;
}

/* hypothesis h of crop b, EPnP minimal solver: five distinct correspondences (n >= 5) */
__device__ static int epnp_hypothesis(unsigned seed, unsigned b, unsigned h, int n, const float* ip, const float* mp, const double* K4, double* scratch,
                                      int lane, double* R, double* t)
{
    for (unsigned tr = 0; tr < 8; tr++) {
        int idx[5], dup = 0;
        for (int j = 0; j < 5; j++) idx[j] = (int)(pnp_hash(seed, b, h, tr, (unsigned)j) % (unsigned)n);
        for (int i = 0; i < 5; i++)
            for (int j = i + 1; j < 5; j++) dup |= idx[i] == idx[j];
        if (dup) continue;
        if (epnp_solve_minimal(5, idx, ip, mp, K4, scratch, lane, R, t)) return 1;
    }
    return 0;
}

// deterministic block-wide sum of NV doubles (fixed butterfly + fixed wave order); result to all threads
template <int NV>
__device__ __forceinline__ void pnp_block_sum(double* v, double* s_buf /* [PNP_WAVES * NV] */)
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
        for (int wv = 0; wv < PNP_WAVES; wv++) t += s_buf[wv * NV + k];
        v[k] = t;
    }
}

// solve H d = g (6x6, Gaussian elimination with partial pivoting) - as oracle solve6
__device__ static int pnp_solve6(double (&H)[6][6], double (&g)[6])
{
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(H[r][c]) > fabs(H[p][c])) p = r;
        if (H[p][c] == 0.0) return 0;
        if (p != c) {
            for (int k = 0; k < 6; k++) { const double tmp = H[c][k]; H[c][k] = H[p][k]; H[p][k] = tmp; }
            const double tg = g[c]; g[c] = g[p]; g[p] = tg;
        }
        for (int r = c + 1; r < 6; r++) {
            const double f = H[r][c] / H[c][c];
            for (int k = c; k < 6; k++) H[r][k] -= f * H[c][k];
            g[r] -= f * g[c];
        }
    }
    for (int c = 5; c >= 0; c--) {
        double s = g[c];
        for (int k = c + 1; k < 6; k++) s -= H[c][k] * g[k];
        g[c] = s / H[c][c];
    }
    return 1;
}

// EPnP over the winner's inliers (this thread's points i = tid + k * PNP_THREADS flagged in use_bits), the whole workgroup: moments,
// M^T M and the per-candidate sums are block reductions (fixed tree: deterministic, but not the oracle's serial order - the refit
// agrees with it to ~1e-9 like the Gauss-Newton one), the 12 x 12 Jacobi runs on wavefront 0's scratch, everything else redundantly
// on every thread.  Writes s_cur (R | t) and returns 1 (uniform) when the inlier set was solvable.
template <int CH>
__device__ __forceinline__ void ep_mtm_chunk(const double* r1, const double* r2, double* acc /* [26] */)
{
    int e = 0;
#pragma unroll
    for (int i = 0; i < 12; i++)
#pragma unroll
        for (int j = i; j < 12; j++, e++)
            if (e >= CH * 26 && e < CH * 26 + 26) acc[e - CH * 26] += r1[i] * r1[j] + r2[i] * r2[j];
}
template <int CH>
__device__ static void ep_mtm_pass(int n, const unsigned* use_bits, const ep_frame& f, const float* s_ip, const float* s_mp, const double* K4,
                                   double* s_red, double* A)
{
    double acc[26];
#pragma unroll
    for (int q = 0; q < 26; q++) acc[q] = 0.0;
    int k = 0;
    for (int i = threadIdx.x; i < n; i += PNP_THREADS, k++) {
        if (!((use_bits[k >> 5] >> (k & 31)) & 1u)) continue;
        double a[4], r1[12], r2[12];
        ep_alphas(f, s_mp + 3 * i, a);
        const double u = (double)s_ip[2 * i], v = (double)s_ip[2 * i + 1];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            r1[3 * j] = a[j] * K4[0]; r1[3 * j + 1] = 0.0;          r1[3 * j + 2] = a[j] * (K4[2] - u);
            r2[3 * j] = 0.0;          r2[3 * j + 1] = a[j] * K4[1]; r2[3 * j + 2] = a[j] * (K4[3] - v);
        }
        ep_mtm_chunk<CH>(r1, r2, acc);
    }
    pnp_block_sum<26>(acc, s_red);
    if (threadIdx.x == 0) {
        int e = 0;
        for (int i = 0; i < 12; i++)
            for (int j = i; j < 12; j++, e++)
                if (e >= CH * 26 && e < CH * 26 + 26) { A[i * 12 + j] = acc[e - CH * 26]; A[j * 12 + i] = acc[e - CH * 26]; }
    }
}

__device__ static int epnp_refit_block(int n, const unsigned* use_bits, const float* s_ip, const float* s_mp, const double* K4, double* s_red,
                                       double* scratch, double* s_cur)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double m[10];
#pragma unroll
    for (int q = 0; q < 10; q++) m[q] = 0.0;
    int first = 0x7fffffff, k = 0;
    for (int i = tid; i < n; i += PNP_THREADS, k++) {
        if (!((use_bits[k >> 5] >> (k & 31)) & 1u)) continue;
        const double x = s_mp[3 * i], y = s_mp[3 * i + 1], z = s_mp[3 * i + 2];
        m[0] += x; m[1] += y; m[2] += z;
        m[3] += x * x; m[4] += x * y; m[5] += x * z; m[6] += y * y; m[7] += y * z; m[8] += z * z;
        m[9] += 1.0;
        if (i < first) first = i;
    }
    pnp_block_sum<10>(m, s_red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(first, o); first = other < first ? other : first; }
    __syncthreads();
    if (lane == 0) s_red[wave] = (double)first;
    __syncthreads();
    for (int wv = 0; wv < PNP_WAVES; wv++) { const int fw = (int)s_red[wv]; first = fw < first ? fw : first; }
    __syncthreads();
    const int cnt = (int)m[9];
    if (cnt < 5) return 0;
    ep_frame f;
    if (!ep_frame_from_moments(cnt, m, m + 3, f)) return 0;
    double* A = scratch;
    double* V = scratch + 144;
    ep_mtm_pass<0>(n, use_bits, f, s_ip, s_mp, K4, s_red, A);
    ep_mtm_pass<1>(n, use_bits, f, s_ip, s_mp, K4, s_red, A);
    ep_mtm_pass<2>(n, use_bits, f, s_ip, s_mp, K4, s_red, A);
    __syncthreads();
    if (wave == 0) {
        ep_jacobi12_wave(A, V, lane);
        ep_pick_null_vectors(A, V, lane);
    }
    __syncthreads();
    double betas[3][4];
    ep_betas_from_null(A, f, betas);
    double a1[4];
    ep_alphas(f, s_mp + 3 * first, a1);
    int bestc = -1;
    double beste = PNP_HUGE, Rb[9], tb[3];
    for (int cnd = 0; cnd < 3; cnd++) {
        double ccs[4][3];
        ep_ccs(A, betas[cnd], ccs);
        const double z1 = a1[0] * ccs[0][2] + a1[1] * ccs[1][2] + a1[2] * ccs[2][2] + a1[3] * ccs[3][2];
        if (z1 < 0.0)
            for (int j = 0; j < 4; j++)
                for (int c = 0; c < 3; c++) ccs[j][c] = -ccs[j][c];
        double acc[12];
#pragma unroll
        for (int q = 0; q < 12; q++) acc[q] = 0.0;
        k = 0;
        for (int i = tid; i < n; i += PNP_THREADS, k++) {
            if (!((use_bits[k >> 5] >> (k & 31)) & 1u)) continue;
            double a[4], pc[3];
            ep_alphas(f, s_mp + 3 * i, a);
            for (int c = 0; c < 3; c++) pc[c] = a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c];
            for (int c = 0; c < 3; c++) acc[c] += pc[c];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) acc[3 + r * 3 + c] += (double)s_mp[3 * i + r] * pc[c];
        }
        pnp_block_sum<12>(acc, s_red);
        const double inv = 1.0 / (double)cnt;
        double S[9], Rc[9], tc[3], cbar[3];
        for (int c = 0; c < 3; c++) cbar[c] = acc[c] * inv;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) S[r * 3 + c] = acc[3 + r * 3 + c] - m[r] * cbar[c];
        ep_horn(S, Rc);
        for (int r = 0; r < 3; r++) tc[r] = cbar[r] - (Rc[r * 3] * f.c0[0] + Rc[r * 3 + 1] * f.c0[1] + Rc[r * 3 + 2] * f.c0[2]);
        double err[1] = {0.0};
        k = 0;
        for (int i = tid; i < n; i += PNP_THREADS, k++) {
            if (!((use_bits[k >> 5] >> (k & 31)) & 1u)) continue;
            const double e2 = reproj_err2(Rc, tc, K4, s_mp + 3 * i, s_ip + 2 * i);
            err[0] += e2 < PNP_HUGE ? sqrt(e2) : 1e12;
        }
        pnp_block_sum<1>(err, s_red);
        if (err[0] < beste) {
            beste = err[0];
            bestc = cnd;
            for (int i = 0; i < 9; i++) Rb[i] = Rc[i];
            for (int i = 0; i < 3; i++) tb[i] = tc[i];
        }
    }
    const int ok = bestc >= 0 && beste < PNP_HUGE;
    __syncthreads();
    if (ok && tid == 0) {
        for (int i = 0; i < 9; i++) s_cur[i] = Rb[i];
        for (int i = 0; i < 3; i++) s_cur[9 + i] = tb[i];
    }
    __syncthreads();
    return ok;
}

template <bool EPNP>
__global__ __launch_bounds__(PNP_THREADS) void ransac_pnp_kernel(
    const float* __restrict__ image_points, const float* __restrict__ model_points, const int* __restrict__ counts,
    const float* __restrict__ cams, const float* __restrict__ net_pose, int HW, float reproj_thr, int iters, float confidence,
    unsigned seed, int mode, float max_t_diff, float* __restrict__ pose_out, int* __restrict__ n_inliers,
    unsigned char* __restrict__ inlier_mask, int* __restrict__ best_hyp, int part, int* __restrict__ g_cnt, double* __restrict__ g_pose)
{
    // part 0: the whole solve in one workgroup per crop.  With fewer crops than compute units a crop's hypotheses are spread over
    // gridDim.y workgroups (part 1: phases 1 - 2 only, counts and poses to the GLOBAL scoreboard g_cnt / g_pose [B][PNP_MAX_ITERS]) and a
    // second launch (part 2: one workgroup per crop reads the scoreboard, phases 3 - 4) finishes - the same per-hypothesis arithmetic and the
    // same scan order, hence bit-identical results (tests/test_gpu_pnp.py); at B = 64 the EPnP form runs 13 rounds of hypotheses on 64
    // of 256 compute units otherwise (6.5 ms), a per-image batch on a handful
    extern __shared__ __attribute__((aligned(16))) unsigned char pnp_smem[];
    double* s_red = reinterpret_cast<double*>(pnp_smem);               // PNP_WAVES * 27
    double* s_pose = s_red + PNP_WAVES * 27;                           // 12 * PNP_MAX_ITERS (R | t per hypothesis)
    double* s_cur = s_pose + 12 * PNP_MAX_ITERS;                       // 12: the pose being refined
    double* s_ep = s_cur + 12;                                         // EPNP: PNP_WAVES * 288 (a wavefront's 12 x 12 A | V)
    int* s_cnt = reinterpret_cast<int*>(s_ep + (EPNP ? PNP_WAVES * 288 : 0));  // PNP_MAX_ITERS
    int* s_misc = s_cnt + PNP_MAX_ITERS;                               // 4
    float* s_ip = reinterpret_cast<float*>(s_misc + 4);                // 2 * HW
    float* s_mp = s_ip + 2 * (size_t)HW;                               // 3 * HW

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts[b];
    // (exactly four correspondences: EPnP's null space is four-dimensional there - such a crop takes the P3P + 1 path in either mode)
    const bool epnp = EPNP && n >= 5;
    const int msize = epnp ? 5 : 4;
    const double K4[4] = {(double)cams[b * 9 + 0], (double)cams[b * 9 + 4], (double)cams[b * 9 + 2], (double)cams[b * 9 + 5]};
    const double thr2 = (double)reproj_thr * (double)reproj_thr;
    float* po = pose_out + b * 12;
    unsigned char* msk = inlier_mask + (size_t)b * HW;
    if (part != 1)
        for (int p = tid; p < HW; p += PNP_THREADS) msk[p] = 0;
    if (n < 4) {  // gdrn_evaluator.py:391-392 (sentinel) / :297-300 (keep the network pose)
        if (part != 1) {
            if (tid < 12) po[tid] = net_pose ? net_pose[b * 12 + tid] : -100.f;
            if (tid == 0) { n_inliers[b] = 0; best_hyp[b] = -1; }
        }
        return;
    }
    // ---- phase 1
    {
        const float* gi = image_points + (size_t)b * HW * 2;
        const float* gm = model_points + (size_t)b * HW * 3;
        for (int i = tid; i < 2 * n; i += PNP_THREADS) s_ip[i] = gi[i];
        for (int i = tid; i < 3 * n; i += PNP_THREADS) s_mp[i] = gm[i];
    }
    for (int h = tid; h < iters; h += PNP_THREADS) s_cnt[h] = part == 2 ? g_cnt[(size_t)b * PNP_MAX_ITERS + h] : -1;
    if (part == 2)
        for (int i = tid; i < 12 * iters; i += PNP_THREADS) s_pose[i] = g_pose[(size_t)b * 12 * PNP_MAX_ITERS + i];
    __syncthreads();

    int best = -1, best_cnt = 0;
    if (mode != 2) {
        // ---- phase 2: one hypothesis per wavefront (part 1: this workgroup's share, results to the global scoreboard)
        const int h_step = PNP_WAVES * (part == 1 ? (int)gridDim.y : 1);
        for (int h = wave + PNP_WAVES * (part == 1 ? (int)blockIdx.y : 0); part != 2 && h < iters; h += h_step) {
            double R[9], t[3];
            int ok = 0;
            if (mode == 1 && h == 0) {
#pragma unroll
                for (int i = 0; i < 9; i++) R[i] = (double)net_pose[b * 12 + i];
#pragma unroll
                for (int i = 0; i < 3; i++) t[i] = (double)net_pose[b * 12 + 9 + i];
                ok = 1;
            } else {
                ok = epnp ? epnp_hypothesis(seed, (unsigned)b, (unsigned)h, n, s_ip, s_mp, K4, s_ep + wave * 288, lane, R, t)
                          : pnp_hypothesis(seed, (unsigned)b, (unsigned)h, n, s_ip, s_mp, K4, R, t);
            }
            if (!ok) {  // wave-uniform
                if (part == 1 && lane == 0) g_cnt[(size_t)b * PNP_MAX_ITERS + h] = -1;
                continue;
            }
            int cnt = 0;
            for (int i = lane; i < n; i += 64) cnt += reproj_err2(R, t, K4, s_mp + 3 * i, s_ip + 2 * i) < thr2 ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0) {
                int* cdst = part == 1 ? g_cnt + (size_t)b * PNP_MAX_ITERS : s_cnt;       // the inlier scoreboard: LDS, or global in the
                double* pdst = part == 1 ? g_pose + (size_t)b * 12 * PNP_MAX_ITERS : s_pose;  // split form
                cdst[h] = cnt;
#pragma unroll
                for (int i = 0; i < 9; i++) pdst[12 * h + i] = R[i];
#pragma unroll
                for (int i = 0; i < 3; i++) pdst[12 * h + 9 + i] = t[i];
            }
        }
        if (part == 1) return;
        __syncthreads();
        // ---- phase 3: scoreboard scan with the confidence-driven stop
        if (tid == 0) {
            int niters = iters;
            for (int h = 0; h < iters && h < niters; h++) {
                const int cnt = s_cnt[h];
                if (cnt > best_cnt && cnt >= msize) {
                    best = h;
                    best_cnt = cnt;
                    const double w = (double)cnt / (double)n;
                    const double miss = 1.0 - (msize == 5 ? w * w * w * w * w : w * w * w * w), target = 1.0 - (double)confidence;
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
        if (best < 0) {
            if (tid < 12) po[tid] = net_pose ? net_pose[b * 12 + tid] : -100.f;
            if (tid == 0) { n_inliers[b] = 0; best_hyp[b] = -1; }
            return;
        }
        if (tid < 12) s_cur[tid] = s_pose[12 * best + tid];
    } else {
        if (tid < 12) s_cur[tid] = (double)net_pose[b * 12 + tid];
    }
    __syncthreads();

    // ---- phase 4: inlier mask of the winner (mode 2: every correspondence takes part), Gauss-Newton refit
    unsigned use_bits[(16384 / PNP_THREADS + 31) / 32 + 1];  // this thread's points (i = tid + k * PNP_THREADS): inlier flags
#pragma unroll
    for (int q = 0; q < (int)(sizeof(use_bits) / 4); q++) use_bits[q] = 0u;
    {
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = s_cur[i];
#pragma unroll
        for (int i = 0; i < 3; i++) t[i] = s_cur[9 + i];
        int k = 0;
        for (int i = tid; i < n; i += PNP_THREADS, k++) {
            const bool in = mode == 2 ? true : reproj_err2(R, t, K4, s_mp + 3 * i, s_ip + 2 * i) < thr2;
            if (in) use_bits[k >> 5] |= 1u << (k & 31);
            if (mode != 2 && in) msk[i] = 1;
        }
    }
    if (epnp && mode != 2) {
        // solvePnP(inliers, SOLVEPNP_EPNP): the final model of the reference's call; a degenerate inlier set keeps the minimal model
        epnp_refit_block(n, use_bits, s_ip, s_mp, K4, s_red, s_ep, s_cur);
    }
    for (int it = 0; it < ((epnp && mode != 2) ? 0 : PNP_REFIT_ITERS); it++) {
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = s_cur[i];
#pragma unroll
        for (int i = 0; i < 3; i++) t[i] = s_cur[9 + i];
        double acc[27];  // 21 upper entries of H (row-major over r <= c) + 6 of g
#pragma unroll
        for (int q = 0; q < 27; q++) acc[q] = 0.0;
        int k = 0;
        for (int i = tid; i < n; i += PNP_THREADS, k++) {
            if (!((use_bits[k >> 5] >> (k & 31)) & 1u)) continue;
            const float* p = s_mp + 3 * i;
            const double rp[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2], R[3] * p[0] + R[4] * p[1] + R[5] * p[2], R[6] * p[0] + R[7] * p[1] + R[8] * p[2]};
            const double X = rp[0] + t[0], Y = rp[1] + t[1], Z = rp[2] + t[2];
            if (!(Z > 0.0)) continue;
            const double iz = 1.0 / Z;
            const double ru = K4[0] * X * iz + K4[2] - (double)s_ip[2 * i], rv = K4[1] * Y * iz + K4[3] - (double)s_ip[2 * i + 1];
            const double a0 = K4[0] * iz, a2 = -K4[0] * X * iz * iz, b1 = K4[1] * iz, b2 = -K4[1] * Y * iz * iz;
            double Ju[6], Jv[6];
            Ju[0] = a2 * rp[1];               Ju[1] = a0 * rp[2] - a2 * rp[0];  Ju[2] = -a0 * rp[1];
            Jv[0] = -b1 * rp[2] + b2 * rp[1]; Jv[1] = -b2 * rp[0];              Jv[2] = b1 * rp[0];
            Ju[3] = a0; Ju[4] = 0.0; Ju[5] = a2;
            Jv[3] = 0.0; Jv[4] = b1; Jv[5] = b2;
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
#pragma unroll
                for (int c = r; c < 6; c++) acc[q++] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
                acc[21 + r] += Ju[r] * ru + Jv[r] * rv;
            }
        }
        pnp_block_sum<27>(acc, s_red);
        if (tid == 0) {
            double H[6][6], g[6];
            int q = 0;
            for (int r = 0; r < 6; r++)
                for (int c = r; c < 6; c++) H[r][c] = acc[q++];
            double trc = 0.0;
            for (int r = 0; r < 6; r++) { trc += H[r][r]; g[r] = acc[21 + r]; }
            for (int r = 0; r < 6; r++) {
                H[r][r] += 1e-12 * trc;
                for (int c = 0; c < r; c++) H[r][c] = H[c][r];
            }
            if (pnp_solve6(H, g)) {
                const double a[3] = {-0.5 * g[0], -0.5 * g[1], -0.5 * g[2]};
                const double aa = dot3(a, a), inv = 1.0 / (1.0 + aa);
                const double C[9] = {(1.0 - aa + 2.0 * a[0] * a[0]) * inv, (2.0 * a[0] * a[1] - 2.0 * a[2]) * inv, (2.0 * a[0] * a[2] + 2.0 * a[1]) * inv,
                                     (2.0 * a[1] * a[0] + 2.0 * a[2]) * inv, (1.0 - aa + 2.0 * a[1] * a[1]) * inv, (2.0 * a[1] * a[2] - 2.0 * a[0]) * inv,
                                     (2.0 * a[2] * a[0] - 2.0 * a[1]) * inv, (2.0 * a[2] * a[1] + 2.0 * a[0]) * inv, (1.0 - aa + 2.0 * a[2] * a[2]) * inv};
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) s_cur[i * 3 + j] = C[i * 3] * R[j] + C[i * 3 + 1] * R[3 + j] + C[i * 3 + 2] * R[6 + j];
                s_cur[9] = t[0] - g[3]; s_cur[10] = t[1] - g[4]; s_cur[11] = t[2] - g[5];
                s_misc[2] = 1;
            } else {
                s_misc[2] = 0;
            }
        }
        __syncthreads();
        if (!s_misc[2]) break;  // uniform
    }
    if (mode == 2) {  // inliers of the refined pose
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = s_cur[i];
#pragma unroll
        for (int i = 0; i < 3; i++) t[i] = s_cur[9 + i];
        double c[1] = {0.0};
        for (int i = tid; i < n; i += PNP_THREADS)
            if (reproj_err2(R, t, K4, s_mp + 3 * i, s_ip + 2 * i) < thr2) { msk[i] = 1; c[0] += 1.0; }
        pnp_block_sum<1>(c, s_red);
        best_cnt = (int)c[0];
        best = 0;
    }
    if (tid == 0) {
        n_inliers[b] = best_cnt;
        best_hyp[b] = best;
        for (int i = 0; i < 12; i++) po[i] = (float)s_cur[i];
        if (net_pose) {  // "translation error too large" guard (gdrn_evaluator.py:293-296): keep the network's t
            double d2 = 0.0;
            for (int i = 0; i < 3; i++) { const double dt = (double)po[9 + i] - (double)net_pose[b * 12 + 9 + i]; d2 += dt * dt; }
            if (sqrt(d2) > (double)max_t_diff)
                for (int i = 0; i < 3; i++) po[9 + i] = net_pose[b * 12 + 9 + i];
        }
    }
}

size_t pnp_smem_bytes(int HW, bool epnp)
{
    size_t s = sizeof(double) * (PNP_WAVES * 27 + 12 * PNP_MAX_ITERS + 12 + (epnp ? PNP_WAVES * 288 : 0)) + sizeof(int) * (PNP_MAX_ITERS + 4)
               + sizeof(float) * 5 * (size_t)HW;
    return (s + 15) & ~(size_t)15;
}

}  // namespace

// image_points [B,HW,2] / model_points [B,HW,3] / counts [B]: the output of rdpn6d_select_correspondences_f32; cams [B,9];
// net_pose [B,12] or NULL.  mode 0: RANSAC (TEST.PNP_TYPE = "ransac_pnp"); 1: the network pose is hypothesis 0
// ("net_ransac_pnp", 20 iterations in the reference); 2: Gauss-Newton from the network pose over all correspondences
// ("net_iter_pnp").  inlier_mask [B,HW] is indexed like the correspondence lists.
// workspace of the split form: the global scoreboard [B][PNP_MAX_ITERS] of (count, pose)
extern "C" long long rdpn6d_ransac_pnp_workspace_bytes(int B) { return B > 0 ? (long long)B * PNP_MAX_ITERS * (4 + 12 * 8) : 0; }

template <bool EPNP>
static int pnp_launch(const float* image_points, const float* model_points, const int* counts, const float* cams, const float* net_pose, int B,
                      int HW, float reproj_thr, int iters, float confidence, unsigned seed, int mode, float max_t_diff, float* pose_out,
                      int* n_inliers, unsigned char* inlier_mask, int* best_hyp, void* workspace, long long workspace_bytes, hipStream_t s)
{
    const size_t smem = pnp_smem_bytes(HW, EPNP);
    RD_REQUIRE(smem <= 160 * 1024, "correspondences do not fit the 160 KiB LDS");
    RD_LDS_OPT_IN(ransac_pnp_kernel<EPNP>, 160 * 1024);
    // fewer crops than compute units: spread a crop's hypotheses over `parts` workgroups (at most one round of hypotheses each)
    int parts = 1;
    if (mode != 2 && workspace && workspace_bytes >= rdpn6d_ransac_pnp_workspace_bytes(B)) {
        static const int cus = [] { int d = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) n = pr.multiProcessorCount; return n; }();
        const int rounds = (iters + PNP_WAVES - 1) / PNP_WAVES;
        parts = cus / B;
        if (parts > rounds) parts = rounds;
        if (const char* e = getenv("RDPN6D_PNP_PARTS")) parts = atoi(e);  // (A/B runs)
        if (parts < 1) parts = 1;
    }
    if (parts >= 2) {
        int* g_cnt = reinterpret_cast<int*>(workspace);
        double* g_pose = reinterpret_cast<double*>(g_cnt + (size_t)B * PNP_MAX_ITERS);
        hipLaunchKernelGGL(ransac_pnp_kernel<EPNP>, dim3(B, parts), dim3(PNP_THREADS), smem, s, image_points, model_points, counts, cams, net_pose,
                           HW, reproj_thr, iters, confidence, seed, mode, max_t_diff, pose_out, n_inliers, inlier_mask, best_hyp, 1, g_cnt, g_pose);
        RD_LAUNCH_CHECK();
        hipLaunchKernelGGL(ransac_pnp_kernel<EPNP>, dim3(B), dim3(PNP_THREADS), smem, s, image_points, model_points, counts, cams, net_pose,
                           HW, reproj_thr, iters, confidence, seed, mode, max_t_diff, pose_out, n_inliers, inlier_mask, best_hyp, 2, g_cnt, g_pose);
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    hipLaunchKernelGGL(ransac_pnp_kernel<EPNP>, dim3(B), dim3(PNP_THREADS), smem, s, image_points, model_points, counts, cams, net_pose, HW,
                       reproj_thr, iters, confidence, seed, mode, max_t_diff, pose_out, n_inliers, inlier_mask, best_hyp, 0, nullptr, nullptr);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// minimal: 0 = P3P + 1 (sets of 4, Gauss-Newton refit), 1 = EPnP (sets of 5, EPnP refit on the inliers: cv2.SOLVEPNP_EPNP's structure,
// lib/pysixd/misc.py:170-179; cfg.TEST.PNP_MINIMAL = "epnp").  workspace (rdpn6d_ransac_pnp_workspace_bytes(B) bytes, or NULL): lets a
// batch of fewer crops than compute units spread each crop's hypotheses over several workgroups (two launches; same results bit for bit)
extern "C" int rdpn6d_ransac_pnp_ws(const float* image_points, const float* model_points, const int* counts, const float* cams,
                                    const float* net_pose, int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed,
                                    int mode, float max_t_diff, int minimal, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                                    int* best_hyp, void* workspace, long long workspace_bytes, void* stream)
{
    RD_REQUIRE(image_points && model_points && counts && cams && pose_out && n_inliers && inlier_mask && best_hyp, "null pointer");
    RD_REQUIRE(B > 0 && HW > 0 && HW <= 6400, "HW must be in 1..6400 (LDS-resident correspondences)");
    RD_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || net_pose), "mode 0 | 1 | 2 (1 and 2 need the network pose)");
    RD_REQUIRE(minimal == 0 || minimal == 1, "minimal solver: 0 P3P + 1 | 1 EPnP");
    RD_REQUIRE(iters >= 1 && iters <= PNP_MAX_ITERS && reproj_thr > 0.f && confidence > 0.f && confidence < 1.f, "iterations / thresholds");
    RD_REQUIRE(mode == 0 || max_t_diff > 0.f, "max_t_diff");
    RD_REQUIRE(pose_out != net_pose, "pose_out must not alias net_pose");
    if (minimal == 1)
        return pnp_launch<true>(image_points, model_points, counts, cams, net_pose, B, HW, reproj_thr, iters, confidence, seed, mode, max_t_diff,
                                pose_out, n_inliers, inlier_mask, best_hyp, workspace, workspace_bytes, (hipStream_t)stream);
    return pnp_launch<false>(image_points, model_points, counts, cams, net_pose, B, HW, reproj_thr, iters, confidence, seed, mode, max_t_diff,
                             pose_out, n_inliers, inlier_mask, best_hyp, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int rdpn6d_ransac_pnp_ex(const float* image_points, const float* model_points, const int* counts, const float* cams,
                                    const float* net_pose, int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed,
                                    int mode, float max_t_diff, int minimal, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                                    int* best_hyp, void* stream)
{
    return rdpn6d_ransac_pnp_ws(image_points, model_points, counts, cams, net_pose, B, HW, reproj_thr, iters, confidence, seed, mode, max_t_diff,
                                minimal, pose_out, n_inliers, inlier_mask, best_hyp, nullptr, 0, stream);
}

extern "C" int rdpn6d_ransac_pnp_f32(const float* image_points, const float* model_points, const int* counts, const float* cams,
                                     const float* net_pose, int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed,
                                     int mode, float max_t_diff, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                                     int* best_hyp, void* stream)
{
    return rdpn6d_ransac_pnp_ex(image_points, model_points, counts, cams, net_pose, B, HW, reproj_thr, iters, confidence, seed, mode, max_t_diff, 0,
                                pose_out, n_inliers, inlier_mask, best_hyp, stream);
}
