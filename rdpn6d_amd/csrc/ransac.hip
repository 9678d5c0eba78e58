// placeholder: per-crop RANSAC + Kabsch kernel lands here next
#include "common.h"
extern "C" int rdpn6d_ransac_kabsch_f32(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                                        const float* resize_ratios, const int* region_argmax, int B, int HW, int K,
                                        float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed,
                                        float* pose_out, int* n_inliers, unsigned char* inlier_mask, void* stream)
{
    rdpn6d_set_error("rdpn6d_ransac_kabsch_f32: not built yet");
    return RDPN6D_EINVAL;
}
