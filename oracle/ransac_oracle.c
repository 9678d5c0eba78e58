/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this object.
 *
 * PARITY UNPINNED with respect to the reference: the per-crop RANSAC + Kabsch pose solve on the
 * RGB-D residual correspondences is a capability the north star asks for that the reference does
 * not implement.  The reference's classical solver is
 * cv2.solvePnPRansac (opencv-python 4.5.5.62, third-party, absent from /root/reference and not
 * installed here; call sites lib/pysixd/misc.py:170-179, gdrn_evaluator.py:382-389).  What is
 * taken from those call sites: mask > MASK_THR_TEST on the min-max normalised mask
 * (engine_utils.py:118-136, gdrn_evaluator.py:110-115), a fixed iteration budget (100), a
 * confidence-driven adaptive stop (0.99), the best model = most inliers, a final re-solve on the
 * inliers of the best model, and the sentinel pose -100 when there are too few points
 * (gdrn_evaluator.py:391-392).  Geometry: for a foreground pixel with camera point
 * P (depth) and predicted residual delta, P - delta = R*anchor[region] + t.
 *
 * This file is the executable specification the HIP kernel (rdpn6d_amd/csrc/ransac.hip) is held
 * to: inlier masks, inlier counts and the chosen hypothesis must be BIT-EXACT under a fixed seed
 * (all fp32 arithmetic below is written operation by operation, compiled with
 * -ffp-contract=off; the random draws and the adaptive stop use integers / IEEE double
 * multiplications only).  The refit accumulates in double; its result is compared to 1e-5.
 * Analytic tests (known pose + noise + outliers) validate both against ground truth.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static unsigned rs_hash(unsigned seed, unsigned b, unsigned h, unsigned t, unsigned j)
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

/* orthonormal frame of a triangle; returns 0 if degenerate */
static int rs_frame(const float* p0, const float* p1, const float* p2, float u1[3], float u2[3], float u3[3])
{
    float e1[3], e2[3], w[3];
    for (int c = 0; c < 3; c++) { e1[c] = p1[c] - p0[c]; e2[c] = p2[c] - p0[c]; }
    w[0] = e1[1] * e2[2] - e1[2] * e2[1];
    w[1] = e1[2] * e2[0] - e1[0] * e2[2];
    w[2] = e1[0] * e2[1] - e1[1] * e2[0];
    float n1 = e1[0] * e1[0]; n1 = n1 + e1[1] * e1[1]; n1 = n1 + e1[2] * e1[2];
    float n2 = e2[0] * e2[0]; n2 = n2 + e2[1] * e2[1]; n2 = n2 + e2[2] * e2[2];
    float nw = w[0] * w[0]; nw = nw + w[1] * w[1]; nw = nw + w[2] * w[2];
    float lim = 1e-10f * (n1 * n2);
    if (!(nw > lim) || !(n1 > 0.f)) return 0;
    float s1 = sqrtf(n1), sw = sqrtf(nw);
    for (int c = 0; c < 3; c++) { u1[c] = e1[c] / s1; u3[c] = w[c] / sw; }
    u2[0] = u3[1] * u1[2] - u3[2] * u1[1];
    u2[1] = u3[2] * u1[0] - u3[0] * u1[2];
    u2[2] = u3[0] * u1[1] - u3[1] * u1[0];
    return 1;
}

/* minimal 3-point rigid alignment a -> q (triad); pose = R row-major (9) | t (3) */
static int rs_triad(const float* a0, const float* a1, const float* a2, const float* q0, const float* q1,
                    const float* q2, float pose[12])
{
    float u1[3], u2[3], u3[3], v1[3], v2[3], v3[3];
    if (!rs_frame(a0, a1, a2, u1, u2, u3)) return 0;
    if (!rs_frame(q0, q1, q2, v1, v2, v3)) return 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float r = v1[i] * u1[j];
            r = r + v2[i] * u2[j];
            r = r + v3[i] * u3[j];
            pose[i * 3 + j] = r;
        }
    float ab[3], qb[3];
    for (int c = 0; c < 3; c++) {
        float s = a0[c] + a1[c]; s = s + a2[c]; ab[c] = s / 3.0f;
        float u = q0[c] + q1[c]; u = u + q2[c]; qb[c] = u / 3.0f;
    }
    for (int i = 0; i < 3; i++) {
        float r = pose[i * 3 + 0] * ab[0];
        r = r + pose[i * 3 + 1] * ab[1];
        r = r + pose[i * 3 + 2] * ab[2];
        pose[9 + i] = qb[i] - r;
    }
    return 1;
}

static void rs_apply(const float pose[12], const float* a, float out[3])
{
    for (int i = 0; i < 3; i++) {
        float r = pose[i * 3 + 0] * a[0];
        r = r + pose[i * 3 + 1] * a[1];
        r = r + pose[i * 3 + 2] * a[2];
        out[i] = r + pose[9 + i];
    }
}

/* Horn's closed-form absolute orientation: largest eigenvector of the 4x4 N matrix by cyclic Jacobi */
static void rs_horn(const double S[9], double R[9])
{
    double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    double A[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                      {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                      {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                      {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 16; sweep++)
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double apq = A[p][q];
                if (apq == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (A[i][i] > A[best][best]) best = i;
    double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/*
 * out_nchw [B,4+K+1,HW]  (ch0 mask, ch1..3 residual xyz), coord2d [B,5,HW] (ch0..2 depth xyz / ratio),
 * fps [B,K,3], extents [B,3], ratios [B], region_argmax [B,HW] (0..K-1)
 * -> pose_out [B,12], n_inliers [B], inlier_mask [B,HW] (may be NULL), best_hyp [B] (may be NULL)
 */
/* net_pose != NULL: the network-initialised solve (role of process_net_and_pnp, gdrn_evaluator.py:187-314): the
 * learned pose is hypothesis 0, is kept when fewer than 3 correspondences survive (:297-300), and its translation is
 * kept when the solved one moved by more than max_t_diff (:293-296). */
static void ransac_impl(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                        const float* ratios, const int* region_argmax, int B, int HW, int K, float mask_thr,
                        float inlier_thr, int iters, float confidence, unsigned seed, float* pose_out,
                        int* n_inliers, unsigned char* inlier_mask, int* best_hyp, const float* net_pose, float max_t_diff,
                        int mask_type)
{
    /* mask_type = ROT_HEAD.MASK_LOSS_TYPE as get_out_mask reads it (core/gdrn_modeling/engine_utils.py:118-136): 0 L1 per-crop
     * min-max, 1 BCE sigmoid, 2 CE arg-max over TWO mask channels (the residual xyz then start at channel 2) */
    const int MC = mask_type == 2 ? 2 : 1;
    const int C = MC + 3 + K + 1;
    float* q = (float*)malloc(sizeof(float) * 3 * (size_t)HW);
    int* ai = (int*)malloc(sizeof(int) * (size_t)HW);
    int* pix = (int*)malloc(sizeof(int) * (size_t)HW);
    int* counts = (int*)malloc(sizeof(int) * (size_t)(iters > 0 ? iters : 1));
    float* poses = (float*)malloc(sizeof(float) * 12 * (size_t)(iters > 0 ? iters : 1));
    float* ta = (float*)malloc(sizeof(float) * 3 * (size_t)K);
    const float thr2 = inlier_thr * inlier_thr;
    for (int b = 0; b < B; b++) {
        const float* m = out_nchw + (size_t)b * C * HW;
        const float* cd = coord2d + (size_t)b * 5 * HW;
        const float* A = fps + (size_t)b * K * 3;
        const float* e = extents + b * 3;
        const float ratio = ratios[b];
        float mn = m[0], mx = m[0];
        for (int p = 1; p < HW; p++) { mn = m[p] < mn ? m[p] : mn; mx = m[p] > mx ? m[p] : mx; }
        int n = 0;
        for (int p = 0; p < HW; p++) {
            float nm;
            if (mask_type == 0) nm = (m[p] - mn) / (mx - mn);
            else if (mask_type == 1) nm = 1.f / (1.f + expf(-m[p]));
            else nm = m[HW + p] > m[p] ? 1.f : 0.f;
            float dz = cd[2 * HW + p];
            if (nm > mask_thr && dz > 0.f) {
                for (int c = 0; c < 3; c++) {
                    float P = cd[c * HW + p] * ratio;
                    float dl = (m[(MC + c) * HW + p] - 0.5f) * e[c];
                    q[3 * n + c] = P - dl;
                }
                ai[n] = region_argmax[(size_t)b * HW + p];
                pix[n] = p;
                n++;
            }
        }
        if (inlier_mask) memset(inlier_mask + (size_t)b * HW, 0, (size_t)HW);
        float* po = pose_out + b * 12;
        int best = -1, best_cnt = 0;
        if (n >= 3) {
            for (int h = 0; h < iters; h++) {
                counts[h] = -1;
                int ok = 0;
                float* ps = poses + 12 * h;
                if (net_pose && h == 0) {
                    for (int i = 0; i < 12; i++) ps[i] = net_pose[b * 12 + i];
                    ok = 1;
                }
                for (int t = 0; t < 8 && !ok; t++) {
                    int i0 = (int)(rs_hash(seed, b, h, t, 0) % (unsigned)n);
                    int i1 = (int)(rs_hash(seed, b, h, t, 1) % (unsigned)n);
                    int i2 = (int)(rs_hash(seed, b, h, t, 2) % (unsigned)n);
                    if (ai[i0] == ai[i1] || ai[i0] == ai[i2] || ai[i1] == ai[i2]) continue;
                    ok = rs_triad(A + 3 * ai[i0], A + 3 * ai[i1], A + 3 * ai[i2], q + 3 * i0, q + 3 * i1, q + 3 * i2, ps);
                }
                if (!ok) continue;
                for (int k = 0; k < K; k++) rs_apply(ps, A + 3 * k, ta + 3 * k);
                int cnt = 0;
                for (int i = 0; i < n; i++) {
                    const float* tk = ta + 3 * ai[i];
                    float r0 = tk[0] - q[3 * i], r1 = tk[1] - q[3 * i + 1], r2 = tk[2] - q[3 * i + 2];
                    float d2 = r0 * r0; d2 = d2 + r1 * r1; d2 = d2 + r2 * r2;
                    cnt += d2 < thr2;
                }
                counts[h] = cnt;
            }
            /* sequential scan with the confidence-driven stop (deterministic: IEEE double multiplies only) */
            int niters = iters;
            for (int h = 0; h < iters && h < niters; h++) {
                int cnt = counts[h];
                if (cnt > best_cnt && cnt >= 3) {
                    best = h;
                    best_cnt = cnt;
                    double w = (double)cnt / (double)n;
                    double miss = 1.0 - w * w * w, prod = 1.0, target = 1.0 - (double)confidence;
                    int k = 0;
                    while (prod > target && k < iters) { prod *= miss; k++; }
                    if (k < niters) niters = k;
                }
            }
        }
        if (best_hyp) best_hyp[b] = best;
        n_inliers[b] = best_cnt;
        if (best < 0) {
            for (int i = 0; i < 12; i++) po[i] = net_pose ? net_pose[b * 12 + i] : -100.f;
            continue;
        }
        /* inliers of the best hypothesis, then Kabsch/Horn refit on them (double accumulation) */
        const float* ps = poses + 12 * best;
        for (int k = 0; k < K; k++) rs_apply(ps, A + 3 * k, ta + 3 * k);
        double sa[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
        int cnt = 0;
        for (int i = 0; i < n; i++) {
            const float* tk = ta + 3 * ai[i];
            float r0 = tk[0] - q[3 * i], r1 = tk[1] - q[3 * i + 1], r2 = tk[2] - q[3 * i + 2];
            float d2 = r0 * r0; d2 = d2 + r1 * r1; d2 = d2 + r2 * r2;
            if (d2 < thr2) {
                if (inlier_mask) inlier_mask[(size_t)b * HW + pix[i]] = 1;
                for (int c = 0; c < 3; c++) { sa[c] += A[3 * ai[i] + c]; sq[c] += q[3 * i + c]; }
                cnt++;
            }
        }
        double abar[3], qbar[3], S[9] = {0};
        for (int c = 0; c < 3; c++) { abar[c] = sa[c] / cnt; qbar[c] = sq[c] / cnt; }
        for (int i = 0; i < n; i++) {
            const float* tk = ta + 3 * ai[i];
            float r0 = tk[0] - q[3 * i], r1 = tk[1] - q[3 * i + 1], r2 = tk[2] - q[3 * i + 2];
            float d2 = r0 * r0; d2 = d2 + r1 * r1; d2 = d2 + r2 * r2;
            if (d2 < thr2)
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) S[r * 3 + c] += (A[3 * ai[i] + r] - abar[r]) * (q[3 * i + c] - qbar[c]);
        }
        double R[9];
        rs_horn(S, R);
        for (int i = 0; i < 9; i++) po[i] = (float)R[i];
        for (int i = 0; i < 3; i++)
            po[9 + i] = (float)(qbar[i] - (R[i * 3] * abar[0] + R[i * 3 + 1] * abar[1] + R[i * 3 + 2] * abar[2]));
        if (net_pose) {
            double d2 = 0.0;
            for (int i = 0; i < 3; i++) {
                double dt = (double)po[9 + i] - (double)net_pose[b * 12 + 9 + i];
                d2 += dt * dt;
            }
            if (sqrt(d2) > (double)max_t_diff)
                for (int i = 0; i < 3; i++) po[9 + i] = net_pose[b * 12 + 9 + i];
        }
    }
    free(q); free(ai); free(pix); free(counts); free(poses); free(ta);
}

void oracle_ransac_kabsch(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                          const float* ratios, const int* region_argmax, int B, int HW, int K, float mask_thr,
                          float inlier_thr, int iters, float confidence, unsigned seed, float* pose_out,
                          int* n_inliers, unsigned char* inlier_mask, int* best_hyp)
{
    ransac_impl(out_nchw, coord2d, fps, extents, ratios, region_argmax, B, HW, K, mask_thr, inlier_thr, iters, confidence,
                seed, pose_out, n_inliers, inlier_mask, best_hyp, NULL, 0.f, 0);
}

/* the plain solve with the mask read as MASK_LOSS_TYPE prescribes (mask_type 0 L1 | 1 BCE | 2 CE, see ransac_impl) */
void oracle_ransac_kabsch_mt(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                             const float* ratios, const int* region_argmax, int B, int HW, int K, float mask_thr, int mask_type,
                             float inlier_thr, int iters, float confidence, unsigned seed, float* pose_out,
                             int* n_inliers, unsigned char* inlier_mask, int* best_hyp)
{
    ransac_impl(out_nchw, coord2d, fps, extents, ratios, region_argmax, B, HW, K, mask_thr, inlier_thr, iters, confidence,
                seed, pose_out, n_inliers, inlier_mask, best_hyp, NULL, 0.f, mask_type);
}

/* mode 1: net pose + iters-1 sampled hypotheses + inlier refit; mode 2: one least-squares fit over all selected points */
void oracle_ransac_kabsch_net(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                              const float* ratios, const int* region_argmax, const float* net_pose, int B, int HW, int K,
                              float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed, int mode,
                              float max_t_diff, float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp)
{
    if (mode == 2) {
        iters = 1;
        inlier_thr = HUGE_VALF;
    }
    ransac_impl(out_nchw, coord2d, fps, extents, ratios, region_argmax, B, HW, K, mask_thr, inlier_thr, iters, confidence,
                seed, pose_out, n_inliers, inlier_mask, best_hyp, net_pose, max_t_diff, 0);
}
