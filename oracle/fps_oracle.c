/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 *
 * Plain-C restatement of the reference's farthest point sampling:
 *   /root/reference/core/csrc/fps/src/farthest_point_sampling.cpp
 *     update_min_dist                      :41-54
 *     find_max_dist_idx                    :56-73
 *     sample_farthest_points (random start):77-105
 *     sample_farthest_points_init_center   :122-160
 *     extern "C" entry points              :166-204
 *
 * Parity is PINNED: tests/test_fps_oracle.py checks this restatement against
 * (a) committed golden index vectors produced by the reference's own .cpp
 * compiled in place (oracle/Makefile -> oracle/_ref/libfps_ref.so) and
 * (b) that library directly whenever it is present.
 *
 * Numerics that matter for bit-exact indices: fp32 only,
 * squared distance evaluated as ((dx*dx)+(dy*dy))+(dz*dz) with one rounding
 * per operation (the reference is g++ -O2 on baseline x86-64: no FMA), strict
 * '>' in the arg-max starting from max_d = 0 / max_idx = 0, ties -> lowest
 * index, already-selected points skipped in both loops.
 * Build with -ffp-contract=off (see oracle/Makefile).
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

static inline float sqdist(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

/* reference :41-54 */
static void update_min_dist(const float *pts, const unsigned char *mask, float *min_dist, int pn, int cur)
{
    for (int i = 0; i < pn; i++) {
        if (mask[i]) continue;
        float d = sqdist(pts + 3 * i, pts + 3 * cur);
        if (d < min_dist[i]) min_dist[i] = d;
    }
}

/* reference :56-73 */
static int find_max_dist_idx(const unsigned char *mask, const float *min_dist, int pn)
{
    int max_idx = 0;
    float max_d = 0.f;
    for (int i = 0; i < pn; i++) {
        if (mask[i]) continue;
        if (min_dist[i] > max_d) { max_idx = i; max_d = min_dist[i]; }
    }
    return max_idx;
}

/* shared tail of reference :95-104 and :150-159 */
static void fps_loop(const float *pts, unsigned char *mask, float *min_dist, int *idxs, int pn, int sn, int cur)
{
    for (int i = 0; i < sn; i++) {
        mask[cur] = 1;
        idxs[i] = cur;
        if (i < sn - 1) {
            update_min_dist(pts, mask, min_dist, pn, cur);
            cur = find_max_dist_idx(mask, min_dist, pn);
        }
    }
}

/* reference :186-204 -> :122-160 */
void oracle_fps_init_center(const float *pts, int *idxs, int pn, int sn)
{
    if (pn <= 0 || sn <= 0) return;
    unsigned char *mask = (unsigned char *)calloc((size_t)pn, 1);
    float *min_dist = (float *)malloc(sizeof(float) * (size_t)pn);
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int i = 0; i < pn; i++)
        for (int c = 0; c < 3; c++) {
            float v = pts[3 * i + c];
            /* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a  (:107-120) */
            mx[c] = (mx[c] < v) ? v : mx[c];
            mn[c] = (v < mn[c]) ? v : mn[c];
        }
    /* (max+min)/2.f == (max+min)*(1.f/2.f)  (:22, :138) */
    float center[3];
    for (int c = 0; c < 3; c++) center[c] = (mx[c] + mn[c]) * (1.f / 2.f);
    for (int i = 0; i < pn; i++) {
        float d = sqdist(pts + 3 * i, center);
        min_dist[i] = (FLT_MAX < d) ? FLT_MAX : d; /* min(d, FLT_MAX) (:141) */
    }
    int cur = find_max_dist_idx(mask, min_dist, pn);
    fps_loop(pts, mask, min_dist, idxs, pn, sn, cur);
    free(mask);
    free(min_dist);
}

/* reference :166-184 -> :77-105 with the random start index pinned to `start`
 * (the reference draws it as srand(time(0)); rand()%pn at :93-94). */
void oracle_fps_from_start(const float *pts, int *idxs, int pn, int sn, int start)
{
    if (pn <= 0 || sn <= 0) return;
    unsigned char *mask = (unsigned char *)calloc((size_t)pn, 1);
    float *min_dist = (float *)malloc(sizeof(float) * (size_t)pn);
    for (int i = 0; i < pn; i++) min_dist[i] = FLT_MAX;
    fps_loop(pts, mask, min_dist, idxs, pn, sn, start % pn);
    free(mask);
    free(min_dist);
}
