/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this object.
 *
 * PARITY UNPINNED with respect to the reference: the reference's 2D-3D pose solve is
 * cv2.solvePnPRansac(EPnP, reprojectionError = 3 px, iterationsCount = 100 | 20 with an extrinsic guess, confidence 0.99)
 * of opencv-python 4.5.5.62 - third-party, absent from /root/reference and not installed here - called at
 * lib/pysixd/misc.py:145-194 (pnp_v2) from core/gdrn_modeling/gdrn_evaluator.py:316-435 (process_pnp_ransac) and
 * :187-314 (process_net_and_pnp).  What this restatement takes from those call sites and from the published structure of
 * solvePnPRansac: minimal sets -> closed-form minimal solver -> inliers by REPROJECTION error < 3 px -> iteration count
 * shortened from the confidence -> final least-squares solve on the inliers of the best model; fewer than 4 correspondences
 * = the -100 sentinel pose; with a network pose: it is hypothesis 0 (useExtrinsicGuess) or the start of a plain iterative
 * least-squares solve (SOLVEPNP_ITERATIVE), the network pose is kept below 4 correspondences and its translation is kept
 * when the solved one moved more than 1 m (gdrn_evaluator.py:293-296).
 *
 * Minimal solver, two of them (oracle_ransac_pnp_ex's `minimal`; cfg.TEST.PNP_MINIMAL): 1 = EPnP on sets of FIVE with an EPnP refit on
 * the inliers - what the reference's call names (flags=cv2.SOLVEPNP_EPNP), restated from the EPnP paper further down; 0 = P3P on
 * three correspondences, disambiguated by the reprojection error of a fourth (what OpenCV's P3P / AP3P RANSAC models do), with a
 * Gauss-Newton refit - the default, and what a crop with exactly four correspondences takes in either mode.  The P3P is the Lambda-Twist formulation (Persson & Nordberg, ECCV 2018), written from its
 * derivation: depths L = (l1, l2, l3) along the unit bearings y_i satisfy  l_i^2 + l_j^2 + b_ij l_i l_j = a_ij
 * (a_ij = |x_i - x_j|^2, b_ij = -2 y_i.y_j), i.e. three quadrics L^T M_ij L = a_ij.  D1 = a23 M12 - a12 M23 and
 * D2 = a23 M13 - a13 M23 are homogeneous; for a root g of det(D1 + g D2) = 0 the form D0 = D1 + g D2 is a pair of planes
 * (eigenvalues e1, e2, 0 with e1 e2 < 0), each plane gives l1 = w0 l2 + w1 l3, and with l3 = tau l2 the first two quadrics
 * leave a quadratic in tau; l2 follows from the third.  Roots are refined by Newton steps on the three quadrics.
 * Refit: Gauss-Newton on the reprojection error over the inliers, 6 parameters, rotation updated by a Cayley transform.
 *
 * Everything that decides an inlier mask is IEEE double arithmetic made of + - * / sqrt only, written operation by operation
 * and compiled with -ffp-contract=off, so that the HIP kernel (rdpn6d_amd/csrc/pnp.hip) reproduces masks, counts and the
 * winning hypothesis BIT FOR BIT under a fixed seed; the refit sums in a different order there and agrees to ~1e-9.
 * Both are validated against analytic ground truth (tests/test_pnp_oracle.py, tests/test_gpu_pnp.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PNP_MAX_ITERS 256
#define PNP_REFIT_ITERS 10

static unsigned pnp_hash(unsigned seed, unsigned b, unsigned h, unsigned t, unsigned j)
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

static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* one real root of x^3 + b x^2 + c x + d by Newton-Raphson from a start on the correct side of the stationary points */
static double cubic_one_root(double b, double c, double d)
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
static int quad_roots(double b, double c, double* r1, double* r2)
{
    const double disc = b * b - 4.0 * c;
    if (!(disc >= 0.0)) return 0;
    const double y = sqrt(disc);
    const double q = b < 0.0 ? 0.5 * (-b + y) : 0.5 * (-b - y);
    *r1 = q;
    *r2 = q != 0.0 ? c / q : 0.0;
    return 2;
}

static double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
/* trace(adj(A) B) for 3x3 */
static double tr_adj_mul(const double* A, const double* B)
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
static int eigvec_sym3(const double* A, double e, double* v)
{
    double M[9];
    memcpy(M, A, sizeof(M));
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
static int p3p_lambdatwist(const double y[3][3], const double x[3][3], double R[4][9], double t[4][3])
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
static double reproj_err2(const double* R, const double* t, const double* K4 /* fx fy cx cy */, const float* p, const float* uv)
{
    const double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    const double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    const double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    if (!(Z > 0.0)) return HUGE_VAL;
    const double du = K4[0] * X / Z + K4[2] - (double)uv[0];
    const double dv = K4[1] * Y / Z + K4[3] - (double)uv[1];
    return du * du + dv * dv;
}

/* hypothesis h of crop b: sample 4 distinct correspondences, P3P on the first three, keep the solution with the smallest
 * reprojection error of the fourth.  Returns 0 if no pose came out. */
static int pnp_hypothesis(unsigned seed, unsigned b, unsigned h, int n, const float* ip, const float* mp, const double* K4, double* R, double* t)
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
        double beste = HUGE_VAL;
        for (int k = 0; k < ns; k++) {
            const double e = reproj_err2(Rs[k], ts[k], K4, mp + 3 * idx[3], ip + 2 * idx[3]);
            if (e < beste) { beste = e; bestk = k; }
        }
        if (bestk < 0) continue;
        memcpy(R, Rs[bestk], 9 * sizeof(double));
        memcpy(t, ts[bestk], 3 * sizeof(double));
        return 1;
    }
    return 0;
}

/* solve the symmetric positive definite 6x6 system H d = g in place (Gaussian elimination with partial pivoting); 0 if singular */
static int solve6(double H[6][6], double g[6])
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

/* one Gauss-Newton step's normal equations over the points with use[i] != 0: H (21 upper entries as 6x6), g, and apply it */
static void pnp_refit(int n, const float* ip, const float* mp, const unsigned char* use, const double* K4, double* R, double* t)
{
    for (int it = 0; it < PNP_REFIT_ITERS; it++) {
        double H[6][6], g[6];
        memset(H, 0, sizeof(H));
        memset(g, 0, sizeof(g));
        for (int i = 0; i < n; i++) {
            if (use && !use[i]) continue;
            const float* p = mp + 3 * i;
            const double rp[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2], R[3] * p[0] + R[4] * p[1] + R[5] * p[2], R[6] * p[0] + R[7] * p[1] + R[8] * p[2]};
            const double X = rp[0] + t[0], Y = rp[1] + t[1], Z = rp[2] + t[2];
            if (!(Z > 0.0)) continue;
            const double iz = 1.0 / Z;
            const double ru = K4[0] * X * iz + K4[2] - (double)ip[2 * i], rv = K4[1] * Y * iz + K4[3] - (double)ip[2 * i + 1];
            /* d(u,v)/d(Xc) and Xc = (I + [w]x) R p + t + u:  dXc/dw = -[R p]x,  dXc/du = I */
            const double a0 = K4[0] * iz, a2 = -K4[0] * X * iz * iz, b1 = K4[1] * iz, b2 = -K4[1] * Y * iz * iz;
            double Ju[6], Jv[6];
            Ju[0] = a2 * rp[1];               Ju[1] = a0 * rp[2] - a2 * rp[0];  Ju[2] = -a0 * rp[1];
            Jv[0] = -b1 * rp[2] + b2 * rp[1]; Jv[1] = -b2 * rp[0];              Jv[2] = b1 * rp[0];
            Ju[3] = a0; Ju[4] = 0.0; Ju[5] = a2;
            Jv[3] = 0.0; Jv[4] = b1; Jv[5] = b2;
            for (int r = 0; r < 6; r++) {
                for (int c = r; c < 6; c++) H[r][c] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
                g[r] += Ju[r] * ru + Jv[r] * rv;
            }
        }
        double trc = 0.0;
        for (int r = 0; r < 6; r++) trc += H[r][r];
        for (int r = 0; r < 6; r++) {
            H[r][r] += 1e-12 * trc;
            for (int c = 0; c < r; c++) H[r][c] = H[c][r];
        }
        if (!solve6(H, g)) return;
        /* R <- Cayley(-d_w) R with a = -d_w / 2:  C = ((1 - a.a) I + 2 a a^T + 2 [a]x) / (1 + a.a);  t <- t - d_u */
        const double a[3] = {-0.5 * g[0], -0.5 * g[1], -0.5 * g[2]};
        const double aa = dot3(a, a), inv = 1.0 / (1.0 + aa);
        const double C[9] = {(1.0 - aa + 2.0 * a[0] * a[0]) * inv, (2.0 * a[0] * a[1] - 2.0 * a[2]) * inv, (2.0 * a[0] * a[2] + 2.0 * a[1]) * inv,
                             (2.0 * a[1] * a[0] + 2.0 * a[2]) * inv, (1.0 - aa + 2.0 * a[1] * a[1]) * inv, (2.0 * a[1] * a[2] - 2.0 * a[0]) * inv,
                             (2.0 * a[2] * a[0] - 2.0 * a[1]) * inv, (2.0 * a[2] * a[1] + 2.0 * a[0]) * inv, (1.0 - aa + 2.0 * a[2] * a[2]) * inv};
        double Rn[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rn[i * 3 + j] = C[i * 3] * R[j] + C[i * 3 + 1] * R[3 + j] + C[i * 3 + 2] * R[6 + j];
        memcpy(R, Rn, sizeof(Rn));
        t[0] -= g[3]; t[1] -= g[4]; t[2] -= g[5];
    }
}


/* =====================================================================================================================
 * EPnP (Lepetit, Moreno-Noguer, Fua: "EPnP: An Accurate O(n) Solution to the PnP Problem", IJCV 2009) - the solver the reference's
 * call names: cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_EPNP) (lib/pysixd/misc.py:170-179): RANSAC over MINIMAL SETS OF FIVE
 * correspondences, each solved by EPnP, and a final EPnP over the inliers of the best model (no iterative refinement).  Restated
 * from the paper (OpenCV's epnp.cpp follows it step for step; its source is third-party and absent here: PARITY UNPINNED):
 *   1. four control points: the centroid c0 and c0 + sqrt(lambda_i / n) u_i along the principal directions u_i of the point set;
 *   2. every model point as an affine combination sum_j alpha_j c_j (barycentric coordinates; with the control points on the
 *      principal axes the inverse of [c1-c0 c2-c0 c3-c0] is diag(1/k) U^T);
 *   3. two rows of M (2n x 12) per correspondence: sum_j alpha_j (fu X_j + (uc - u) Z_j) = 0, sum_j alpha_j (fv Y_j + (vc - v) Z_j) = 0
 *      in the unknown camera-frame control points (X_j Y_j Z_j);
 *   4. the four eigenvectors v_0..v_3 of M^T M (12 x 12, cyclic Jacobi) with the smallest eigenvalues span the solution:
 *      x = sum_k beta_k v_k; the betas follow from the six control-point distances (L b = rho, the paper's linearisations for
 *      N = 4, 2, 3 unknown betas) and five Gauss-Newton steps on the distance constraints;
 *   5. each of the three beta candidates gives camera-frame control points, hence camera-frame model points, hence (R, t) by
 *      Horn's absolute orientation; the candidate with the smallest mean reprojection error wins.
 * Arithmetic: IEEE double, + - * / sqrt only, operation order fixed (the HIP kernel mirrors it: csrc/pnp.hip).
 */
#define EP_SWEEPS 30

/* one Jacobi rotation (p, q) of the symmetric n x n matrix A (both triangles kept) and the eigenvector matrix V (columns): a rotation
 * touches row / column k of A and row k of V independently for every k - the HIP kernel gives those to lanes 0..n-1 */
static void ep_rotate(int n, double* A, double* V, int p, int q)
{
    const double apq = A[p * n + q];
    if (apq == 0.0) return;
    const double app = A[p * n + p], aqq = A[q * n + q];
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
    for (int k = 0; k < n; k++) {
        if (k != p && k != q) {
            const double akp = A[k * n + p], akq = A[k * n + q];
            const double nkp = c * akp - sn * akq, nkq = sn * akp + c * akq;
            A[k * n + p] = nkp; A[p * n + k] = nkp;
            A[k * n + q] = nkq; A[q * n + k] = nkq;
        }
        const double vkp = V[k * n + p], vkq = V[k * n + q];
        V[k * n + p] = c * vkp - sn * vkq;
        V[k * n + q] = sn * vkp + c * vkq;
    }
    A[p * n + p] = app - t * apq;
    A[q * n + q] = aqq + t * apq;
    A[p * n + q] = 0.0;
    A[q * n + p] = 0.0;
}

/* Jacobi eigen-decomposition of the symmetric n x n matrix A (row-major): on return diag(A) holds the eigenvalues and the COLUMNS of V
 * the eigenvectors.  Sweeps until the off-diagonal mass is below 1e-30 of the diagonal's.  Pair order: cyclic by rows for the small
 * matrices (n = 3, 4); for n = 12 the ROUND-ROBIN (tournament) order - 11 steps of 6 DISJOINT pairs: (11, s) and (s + i, s - i) mod 11,
 * i = 1..5.  The rotations of a step do not touch each other's three defining entries, so computing a pair's angle when its turn comes
 * (here) or all six angles from the matrix at the start of the step (the GPU: one pair per lane - the fp64 divide / square-root chain
 * is the cost of a rotation) gives the same numbers bit for bit; the rotations are applied in the order below in both. */
static void ep_jacobi(int n, double* A, double* V)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < EP_SWEEPS; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < n; p++) {
            dg += A[p * n + p] * A[p * n + p];
            for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        }
        if (!(off > 1e-30 * dg)) break;
        if (n == 12) {
            for (int s = 0; s < 11; s++)
                for (int i = 0; i < 6; i++) {
                    const int a = i == 0 ? 11 : (s + i) % 11, b = i == 0 ? s : (s + 11 - i) % 11;
                    ep_rotate(n, A, V, a < b ? a : b, a < b ? b : a);
                }
        } else {
            for (int p = 0; p < n - 1; p++)
                for (int q = p + 1; q < n; q++) ep_rotate(n, A, V, p, q);
        }
    }
}

/* least squares min |A x - b| for an m x n system (n <= 5) through the normal equations (a 1e-13 trace ridge keeps a rank-deficient
 * linearisation solvable; Gaussian elimination with partial pivoting); 0 if a pivot vanishes */
static int ep_lsq(int m, int n, const double* A, const double* b, double* x)
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

/* Horn's absolute orientation, S[r][c] = sum (w_r - wbar_r)(c_c - cbar_c): the rotation taking model to camera coordinates */
static void ep_horn(const double S[9], double R[9])
{
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    double A[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                    Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                    Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy,
                    Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
    double V[16];
    ep_jacobi(4, A, V);
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

typedef struct {
    double c0[3];     /* centroid = control point 0 */
    double U[9];      /* principal directions in the columns */
    double ik[3];     /* 1 / k_i, k_i = sqrt(lambda_i / n): control point i = c0 + k_i U[:, i-1] */
    double cws[4][3];
} ep_frame;

/* barycentric coordinates of model point p */
static void ep_alphas(const ep_frame* f, const float* p, double a[4])
{
    const double d0 = (double)p[0] - f->c0[0], d1 = (double)p[1] - f->c0[1], d2 = (double)p[2] - f->c0[2];
    a[1] = (f->U[0] * d0 + f->U[3] * d1 + f->U[6] * d2) * f->ik[0];
    a[2] = (f->U[1] * d0 + f->U[4] * d1 + f->U[7] * d2) * f->ik[1];
    a[3] = (f->U[2] * d0 + f->U[5] * d1 + f->U[8] * d2) * f->ik[2];
    a[0] = 1.0 - a[1] - a[2] - a[3];
}

/* control points from the first and second moments of the point set (sum p, sum p p^T over cnt points); 0 for a (near-)planar set */
static int ep_frame_from_moments(int cnt, const double sp[3], const double spp[6] /* xx xy xz yy yz zz */, ep_frame* f)
{
    const double inv = 1.0 / (double)cnt;
    for (int c = 0; c < 3; c++) f->c0[c] = sp[c] * inv;
    double C[9];
    C[0] = spp[0] - sp[0] * f->c0[0]; C[1] = spp[1] - sp[0] * f->c0[1]; C[2] = spp[2] - sp[0] * f->c0[2];
    C[4] = spp[3] - sp[1] * f->c0[1]; C[5] = spp[4] - sp[1] * f->c0[2]; C[8] = spp[5] - sp[2] * f->c0[2];
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    ep_jacobi(3, C, f->U);
    double lmax = C[0] > C[4] ? C[0] : C[4];
    if (C[8] > lmax) lmax = C[8];
    if (!(lmax > 0.0)) return 0;
    for (int i = 0; i < 3; i++) {
        const double lam = C[i * 3 + i];
        if (!(lam > 1e-12 * lmax)) return 0;  /* planar / collinear set: the barycentric frame is singular */
        const double k = sqrt(lam * inv);
        f->ik[i] = 1.0 / k;
        for (int c = 0; c < 3; c++) f->cws[i + 1][c] = f->c0[c] + k * f->U[c * 3 + i];
    }
    for (int c = 0; c < 3; c++) f->cws[0][c] = f->c0[c];
    return 1;
}

/* the two rows of M for one correspondence, accumulated into the upper triangle of M^T M (78 entries, row-major over i <= j) */
static void ep_accumulate_mtm(const double a[4], double u, double v, const double* K4, double* mtm /* [78] */)
{
    double r1[12], r2[12];
    for (int j = 0; j < 4; j++) {
        r1[3 * j] = a[j] * K4[0]; r1[3 * j + 1] = 0.0;          r1[3 * j + 2] = a[j] * (K4[2] - u);
        r2[3 * j] = 0.0;          r2[3 * j + 1] = a[j] * K4[1]; r2[3 * j + 2] = a[j] * (K4[3] - v);
    }
    int e = 0;
    for (int i = 0; i < 12; i++)
        for (int j = i; j < 12; j++, e++) mtm[e] += r1[i] * r1[j] + r2[i] * r2[j];
}

/* steps 4 - 5 without the pose: from M^T M (upper triangle) to the three beta candidates and the null-space vectors */
static int ep_betas(const double* mtm, const ep_frame* f, double v[4][12], double betas[3][4])
{
    double A[144], V[144];
    int e = 0;
    for (int i = 0; i < 12; i++)
        for (int j = i; j < 12; j++, e++) { A[i * 12 + j] = mtm[e]; A[j * 12 + i] = mtm[e]; }
    ep_jacobi(12, A, V);
    /* the four smallest eigenvalues, ascending (ties: lower index first) */
    int used[12] = {0};
    for (int k = 0; k < 4; k++) {
        int best = -1;
        for (int i = 0; i < 12; i++)
            if (!used[i] && (best < 0 || A[i * 12 + i] < A[best * 12 + best])) best = i;
        used[best] = 1;
        for (int r = 0; r < 12; r++) v[k][r] = V[r * 12 + best];
    }
    /* differences of the control-point triples of every null vector over the six pairs, L (6 x 10), rho */
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double dv[4][6][3], L[6][10], rho[6];
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 6; j++)
            for (int c = 0; c < 3; c++) dv[k][j][c] = v[k][3 * pa[j] + c] - v[k][3 * pb[j] + c];
    for (int j = 0; j < 6; j++) {
        L[j][0] = dot3(dv[0][j], dv[0][j]);
        L[j][1] = 2.0 * dot3(dv[0][j], dv[1][j]);
        L[j][2] = dot3(dv[1][j], dv[1][j]);
        L[j][3] = 2.0 * dot3(dv[0][j], dv[2][j]);
        L[j][4] = 2.0 * dot3(dv[1][j], dv[2][j]);
        L[j][5] = dot3(dv[2][j], dv[2][j]);
        L[j][6] = 2.0 * dot3(dv[0][j], dv[3][j]);
        L[j][7] = 2.0 * dot3(dv[1][j], dv[3][j]);
        L[j][8] = 2.0 * dot3(dv[2][j], dv[3][j]);
        L[j][9] = dot3(dv[3][j], dv[3][j]);
        double d[3];
        for (int c = 0; c < 3; c++) d[c] = f->cws[pa[j]][c] - f->cws[pb[j]][c];
        rho[j] = dot3(d, d);
    }
    double Lsub[30], b5[5];
    /* N = 4: unknowns b11 b12 b13 b14 (columns 0 1 3 6) */
    static const int c4[4] = {0, 1, 3, 6};
    for (int j = 0; j < 6; j++)
        for (int k = 0; k < 4; k++) Lsub[j * 4 + k] = L[j][c4[k]];
    for (int k = 0; k < 4; k++) betas[0][k] = 0.0;
    if (ep_lsq(6, 4, Lsub, rho, b5)) {
        if (b5[0] < 0.0) { betas[0][0] = sqrt(-b5[0]); betas[0][1] = -b5[1] / betas[0][0]; betas[0][2] = -b5[2] / betas[0][0]; betas[0][3] = -b5[3] / betas[0][0]; }
        else { betas[0][0] = sqrt(b5[0]); betas[0][1] = b5[1] / betas[0][0]; betas[0][2] = b5[2] / betas[0][0]; betas[0][3] = b5[3] / betas[0][0]; }
    }
    /* N = 2: b11 b12 b22 (columns 0 1 2) */
    for (int j = 0; j < 6; j++)
        for (int k = 0; k < 3; k++) Lsub[j * 3 + k] = L[j][k];
    for (int k = 0; k < 4; k++) betas[1][k] = 0.0;
    if (ep_lsq(6, 3, Lsub, rho, b5)) {
        if (b5[0] < 0.0) { betas[1][0] = sqrt(-b5[0]); betas[1][1] = b5[2] < 0.0 ? sqrt(-b5[2]) : 0.0; }
        else { betas[1][0] = sqrt(b5[0]); betas[1][1] = b5[2] > 0.0 ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0.0) betas[1][0] = -betas[1][0];
    }
    /* N = 3: b11 b12 b22 b13 b23 (columns 0 1 2 3 4) */
    for (int j = 0; j < 6; j++)
        for (int k = 0; k < 5; k++) Lsub[j * 5 + k] = L[j][k];
    for (int k = 0; k < 4; k++) betas[2][k] = 0.0;
    if (ep_lsq(6, 5, Lsub, rho, b5)) {
        if (b5[0] < 0.0) { betas[2][0] = sqrt(-b5[0]); betas[2][1] = b5[2] < 0.0 ? sqrt(-b5[2]) : 0.0; }
        else { betas[2][0] = sqrt(b5[0]); betas[2][1] = b5[2] > 0.0 ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0.0) betas[2][0] = -betas[2][0];
        betas[2][2] = betas[2][0] != 0.0 ? b5[3] / betas[2][0] : 0.0;
    }
    /* five Gauss-Newton steps on  sum_j (rho_j - beta^T L_j beta)^2  for each candidate */
    for (int cnd = 0; cnd < 3; cnd++) {
        double* b = betas[cnd];
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
    }
    return 1;
}

/* camera-frame control points of a beta candidate */
static void ep_ccs(const double v[4][12], const double* b, double ccs[4][3])
{
    for (int j = 0; j < 4; j++)
        for (int c = 0; c < 3; c++)
            ccs[j][c] = b[0] * v[0][3 * j + c] + b[1] * v[1][3 * j + c] + b[2] * v[2][3 * j + c] + b[3] * v[3][3 * j + c];
}

/* EPnP over the correspondences listed in idx[0..cnt) (idx == NULL: all i < cnt with use[i] != 0, or all when use == NULL).
 * Returns 0 when the set is degenerate. */
static int epnp_solve(int cnt_or_n, const int* idx, const unsigned char* use, const float* ip, const float* mp, const double* K4, double* R, double* t)
{
    /* pass 1: moments */
    double sp[3] = {0, 0, 0}, spp[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0, first = -1;
    for (int q = 0; q < cnt_or_n; q++) {
        const int i = idx ? idx[q] : q;
        if (!idx && use && !use[i]) continue;
        const double x = mp[3 * i], y = mp[3 * i + 1], z = mp[3 * i + 2];
        sp[0] += x; sp[1] += y; sp[2] += z;
        spp[0] += x * x; spp[1] += x * y; spp[2] += x * z; spp[3] += y * y; spp[4] += y * z; spp[5] += z * z;
        if (first < 0) first = i;
        cnt++;
    }
    if (cnt < 4) return 0;
    ep_frame f;
    if (!ep_frame_from_moments(cnt, sp, spp, &f)) return 0;
    /* pass 2: M^T M */
    double mtm[78];
    for (int e = 0; e < 78; e++) mtm[e] = 0.0;
    for (int q = 0; q < cnt_or_n; q++) {
        const int i = idx ? idx[q] : q;
        if (!idx && use && !use[i]) continue;
        double a[4];
        ep_alphas(&f, mp + 3 * i, a);
        ep_accumulate_mtm(a, (double)ip[2 * i], (double)ip[2 * i + 1], K4, mtm);
    }
    double v[4][12], betas[3][4];
    if (!ep_betas(mtm, &f, v, betas)) return 0;
    /* pass 3: per candidate the camera-frame points' moments against the model points', Horn, mean reprojection error */
    double a1[4];
    ep_alphas(&f, mp + 3 * first, a1);
    int bestc = -1;
    double beste = HUGE_VAL;
    for (int cnd = 0; cnd < 3; cnd++) {
        double ccs[4][3];
        ep_ccs(v, betas[cnd], ccs);
        /* the eigenvectors' sign is arbitrary: the first point must lie in front of the camera */
        const double z1 = a1[0] * ccs[0][2] + a1[1] * ccs[1][2] + a1[2] * ccs[2][2] + a1[3] * ccs[3][2];
        if (z1 < 0.0)
            for (int j = 0; j < 4; j++)
                for (int c = 0; c < 3; c++) ccs[j][c] = -ccs[j][c];
        double sc[3] = {0, 0, 0}, swc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < cnt_or_n; q++) {
            const int i = idx ? idx[q] : q;
            if (!idx && use && !use[i]) continue;
            double a[4], pc[3];
            ep_alphas(&f, mp + 3 * i, a);
            for (int c = 0; c < 3; c++) pc[c] = a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c];
            for (int c = 0; c < 3; c++) sc[c] += pc[c];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) swc[r * 3 + c] += (double)mp[3 * i + r] * pc[c];
        }
        const double inv = 1.0 / (double)cnt;
        double S[9], Rc[9], tc[3], cbar[3];
        for (int c = 0; c < 3; c++) cbar[c] = sc[c] * inv;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) S[r * 3 + c] = swc[r * 3 + c] - sp[r] * cbar[c];  /* sum (w - wbar)(c - cbar)^T */
        ep_horn(S, Rc);
        for (int r = 0; r < 3; r++) tc[r] = cbar[r] - (Rc[r * 3] * f.c0[0] + Rc[r * 3 + 1] * f.c0[1] + Rc[r * 3 + 2] * f.c0[2]);
        double err = 0.0;
        for (int q = 0; q < cnt_or_n; q++) {
            const int i = idx ? idx[q] : q;
            if (!idx && use && !use[i]) continue;
            const double e2 = reproj_err2(Rc, tc, K4, mp + 3 * i, ip + 2 * i);
            err += e2 < HUGE_VAL ? sqrt(e2) : 1e12;
        }
        if (err < beste) {
            beste = err;
            bestc = cnd;
            memcpy(R, Rc, sizeof(Rc));
            memcpy(t, tc, sizeof(tc));
        }
    }
    return bestc >= 0 && beste < HUGE_VAL;
}

/* hypothesis h of crop b with the EPnP minimal solver: five distinct correspondences (n >= 5) */
static int epnp_hypothesis(unsigned seed, unsigned b, unsigned h, int n, const float* ip, const float* mp, const double* K4, double* R, double* t)
{
    for (unsigned tr = 0; tr < 8; tr++) {
        int idx[5], dup = 0;
        for (int j = 0; j < 5; j++) idx[j] = (int)(pnp_hash(seed, b, h, tr, (unsigned)j) % (unsigned)n);
        for (int i = 0; i < 5; i++)
            for (int j = i + 1; j < 5; j++) dup |= idx[i] == idx[j];
        if (dup) continue;
        if (epnp_solve(5, idx, NULL, ip, mp, K4, R, t)) return 1;
    }
    return 0;
}

/*
 * image_points [B][HW][2], model_points [B][HW][3] (first counts[b] rows valid: the output of the correspondence selection),
 * cams [B][9] (K row-major), net_pose [B][12] or NULL.  mode 0: plain RANSAC; 1: network pose = hypothesis 0; 2: no RANSAC,
 * Gauss-Newton from the network pose over all correspondences.
 * pose_out [B][12] (R row-major | t), n_inliers [B], inlier_mask [B][HW] (indexed like the correspondence lists), best_hyp [B].
 */
/* minimal: 0 = P3P + 1 (sets of 4, Gauss-Newton refit: the round-2 solver), 1 = EPnP (sets of 5, EPnP refit on the inliers: the
 * solver the reference's call names, cfg.TEST.PNP_MINIMAL = "epnp") */
void oracle_ransac_pnp_ex(const float* image_points, const float* model_points, const int* counts, const float* cams, const float* net_pose,
                          int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed, int mode, float max_t_diff, int minimal,
                          float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp)
{
    /* (exactly four correspondences: EPnP's null space is then four-dimensional and its linearisations are not a solver - measured 21
     *  degrees off on exact data; such a crop takes the P3P + 1 path, which is exact on four points) */
    if (iters > PNP_MAX_ITERS) iters = PNP_MAX_ITERS;
    const double thr2 = (double)reproj_thr * (double)reproj_thr;
    for (int b = 0; b < B; b++) {
        const float* ip = image_points + (size_t)b * HW * 2;
        const float* mp = model_points + (size_t)b * HW * 3;
        unsigned char* msk = inlier_mask + (size_t)b * HW;
        float* po = pose_out + b * 12;
        const int n = counts[b];
        const int epnp = minimal == 1 && n >= 5, msize = epnp ? 5 : 4;
        const double K4[4] = {cams[b * 9 + 0], cams[b * 9 + 4], cams[b * 9 + 2], cams[b * 9 + 5]};
        memset(msk, 0, (size_t)HW);
        n_inliers[b] = 0;
        best_hyp[b] = -1;
        double R[9], t[3];
        if (n < 4) {  /* gdrn_evaluator.py:391-392 / :297-300 */
            for (int i = 0; i < 12; i++) po[i] = net_pose ? net_pose[b * 12 + i] : -100.f;
            continue;
        }
        if (mode == 2) {
            for (int i = 0; i < 9; i++) R[i] = net_pose[b * 12 + i];
            for (int i = 0; i < 3; i++) t[i] = net_pose[b * 12 + 9 + i];
            pnp_refit(n, ip, mp, NULL, K4, R, t);
            int cnt = 0;
            for (int i = 0; i < n; i++)
                if (reproj_err2(R, t, K4, mp + 3 * i, ip + 2 * i) < thr2) { msk[i] = 1; cnt++; }
            n_inliers[b] = cnt;
            best_hyp[b] = 0;
        } else {
            static double Rh[PNP_MAX_ITERS][9], th[PNP_MAX_ITERS][3];
            int cnts[PNP_MAX_ITERS];
            for (int h = 0; h < iters; h++) {
                cnts[h] = -1;
                int ok = 0;
                if (mode == 1 && h == 0) {
                    for (int i = 0; i < 9; i++) Rh[0][i] = net_pose[b * 12 + i];
                    for (int i = 0; i < 3; i++) th[0][i] = net_pose[b * 12 + 9 + i];
                    ok = 1;
                } else {
                    ok = epnp ? epnp_hypothesis(seed, (unsigned)b, (unsigned)h, n, ip, mp, K4, Rh[h], th[h])
                                      : pnp_hypothesis(seed, (unsigned)b, (unsigned)h, n, ip, mp, K4, Rh[h], th[h]);
                }
                if (!ok) continue;
                int cnt = 0;
                for (int i = 0; i < n; i++) cnt += reproj_err2(Rh[h], th[h], K4, mp + 3 * i, ip + 2 * i) < thr2 ? 1 : 0;
                cnts[h] = cnt;
            }
            int best = -1, best_cnt = 0, niters = iters;
            for (int h = 0; h < iters && h < niters; h++) {
                const int cnt = cnts[h];
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
            best_hyp[b] = best;
            n_inliers[b] = best_cnt;
            if (best < 0) {
                for (int i = 0; i < 12; i++) po[i] = net_pose ? net_pose[b * 12 + i] : -100.f;
                continue;
            }
            memcpy(R, Rh[best], sizeof(R));
            memcpy(t, th[best], sizeof(t));
            for (int i = 0; i < n; i++) msk[i] = reproj_err2(R, t, K4, mp + 3 * i, ip + 2 * i) < thr2 ? 1 : 0;
            if (epnp) {  /* solvePnP(inliers, SOLVEPNP_EPNP); a degenerate inlier set keeps the minimal model */
                double Rr[9], tr3[3];
                if (epnp_solve(n, NULL, msk, ip, mp, K4, Rr, tr3)) { memcpy(R, Rr, sizeof(R)); memcpy(t, tr3, sizeof(t)); }
            } else {
                pnp_refit(n, ip, mp, msk, K4, R, t);
            }
        }
        for (int i = 0; i < 9; i++) po[i] = (float)R[i];
        for (int i = 0; i < 3; i++) po[9 + i] = (float)t[i];
        if (net_pose) {
            double d2 = 0.0;
            for (int i = 0; i < 3; i++) { const double dt = (double)po[9 + i] - (double)net_pose[b * 12 + 9 + i]; d2 += dt * dt; }
            if (sqrt(d2) > (double)max_t_diff)
                for (int i = 0; i < 3; i++) po[9 + i] = net_pose[b * 12 + 9 + i];
        }
    }
}

void oracle_ransac_pnp(const float* image_points, const float* model_points, const int* counts, const float* cams, const float* net_pose,
                       int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed, int mode, float max_t_diff,
                       float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp)
{
    oracle_ransac_pnp_ex(image_points, model_points, counts, cams, net_pose, B, HW, reproj_thr, iters, confidence, seed, mode, max_t_diff, 0,
                         pose_out, n_inliers, inlier_mask, best_hyp);
}

/* EPnP on its own over the first n correspondences (tests: noise-free data must give back the pose) */
int oracle_epnp(const float* image_points, const float* model_points, int n, const float* cam9, double* R, double* t)
{
    const double K4[4] = {cam9[0], cam9[4], cam9[2], cam9[5]};
    return epnp_solve(n, NULL, NULL, image_points, model_points, K4, R, t);
}
