/*
 * rdpn6d.h - C ABI of librdpn6d_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * RDPN6D hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * Conventions
 *   - every `d_*` / activation / weight pointer is a DEVICE pointer (HBM), fp32 unless noted;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - functions returning int return 0 on success, a negative RDPN6D_E* code otherwise and
 *     leave a message retrievable with rdpn6d_last_error();
 *   - internal activation layout is NHWC ("pixels x channels", channel counts padded to a
 *     multiple of 16 with zeros); tensors that cross the reference's Python API are NCHW.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef RDPN6D_H
#define RDPN6D_H

#ifdef __cplusplus
extern "C" {
#endif

#define RDPN6D_OK 0
#define RDPN6D_EINVAL -1  /* bad argument                      */
#define RDPN6D_EHIP -2    /* HIP runtime error / no GPU        */
#define RDPN6D_ENOMEM -3

const char* rdpn6d_last_error(void);
int rdpn6d_version(void);
/* number of visible HIP devices (0 on a CPU-only box); never throws */
int rdpn6d_device_count(void);

/* ------------------------------------------------------------------ farthest point sampling
 * Replaces core/csrc/fps/src/ext.h:1-14 (same two symbols, same signature, HOST pointers,
 * synchronous).  pts [pn,3] f32 C-contiguous, idxs [sn] i32.  The plain variant draws its start
 * index like the reference does (srand(time(0)); rand()%pn, farthest_point_sampling.cpp:93-94).
 * On failure (no GPU) they write -1 into idxs and print to stderr: there is no CPU fallback. */
void farthest_point_sampling(float* pts, int* idxs, int pn, int sn);
void farthest_point_sampling_init_center(float* pts, int* idxs, int pn, int sn);
/* Extra entry points (status code; pinned start; batched device-resident form).
 * start < 0 : bbox-centre initialisation (== farthest_point_sampling_init_center)
 * start >= 0: random-start variant with the start index pinned to start % pn            */
int rdpn6d_fps_host(const float* pts, int* idxs, int pn, int sn, int start);
/* nobj clouds concatenated in d_pts; cloud o = points [d_offsets[o], d_offsets[o+1]);
 * d_idxs [nobj, sn]; d_mindist = scratch of d_offsets[nobj] floats.                      */
int rdpn6d_fps_device(const float* d_pts, const int* d_offsets, int nobj, int max_pn, int sn, int start,
                      int* d_idxs, float* d_mindist, void* stream);
/* The same with a workspace of rdpn6d_fps_workspace_bytes(nobj) bytes: a cloud of 16 385 .. 262 144 points is sampled by
 * ceil(max_pn / 16 384) workgroups (points and minimum distances in registers, one cross-workgroup barrier per sample) instead of one
 * workgroup streaming it from L2; same indices bit for bit.  workspace[obj] = {int barrier counter, int error, ...}: error != 0 after
 * the launch = a barrier timed out (indices -1).  The launch is an ordinary one (co-residency of a cloud's workgroups is likely, not
 * guaranteed): a caller of this asynchronous entry MUST read the error word once the stream has drained and re-run the batch through
 * rdpn6d_fps_device when it is set; rdpn6d_fps_host (and with it both reference-named symbols) does so itself. */
long long rdpn6d_fps_workspace_bytes(int nobj);
int rdpn6d_fps_device_ws(const float* d_pts, const int* d_offsets, int nobj, int max_pn, int sn, int start, int* d_idxs,
                         float* d_mindist, void* workspace, long long workspace_bytes, void* stream);

/* ------------------------------------------------------------------ implicit-GEMM convolution
 * One kernel family serves every conv / transposed-conv phase / FC layer of
 * core/gdrn_modeling/models/{resnet_backbone.py:264-340, cdpn_rot_head_region.py:83-138,
 * conv_pnp_net.py:75-95} (cuDNN / cuBLAS calls in the reference).
 *   y[pix_out(b,oy,ox), co + n] = act( scale[n] * sum_{t,c} x[b, oy*stride+dy[t], ox*stride+dx[t], ci + c]
 *                                                         * w[n, t, c] + shift[n] (+ res[...]) )
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32).                      */
typedef struct {
    const float* x;      /* input, NHWC [B,H,W,in_cs]                                      */
    const float* w;      /* packed weights [Npad][ntaps][Cin]                             */
    const float* scale;  /* [Npad] per-output-channel multiplier (folded BN) or NULL = 1   */
    const float* shift;  /* [Npad] per-output-channel offset (folded BN / bias) or NULL = 0 */
    const float* res;    /* optional residual, NHWC [B,OH,OW,res_cs], or NULL              */
    float* y;            /* output, NHWC [B,OH,OW,out_cs]                                  */
    int B, H, W;         /* input batch / spatial                                          */
    int Cin;             /* reduction channels (multiple of 16)                            */
    int in_cs, in_co;    /* input channel stride / first channel (multiples of 4)          */
    int Ho, Wo;          /* iteration space (outputs per phase)                            */
    int stride;          /* input step per output step                                     */
    int ntaps;           /* 1..9                                                           */
    int dy[9], dx[9];    /* input offset of tap t (pad already folded in; may be negative) */
    int N, Npad;         /* real / padded output channels                                  */
    int OH, OW;          /* full output spatial size                                       */
    int osy, osx, ooy, oox; /* output pixel = (oy*osy+ooy, ox*osx+oox)                    */
    int out_cs, out_co;  /* output channel stride / first channel                          */
    int res_cs, res_co;
    int act;             /* 0 none, 1 ReLU, 2 LeakyReLU(slope)                             */
    float slope;
} rdpn6d_conv_desc;
int rdpn6d_conv2d_f32(const rdpn6d_conv_desc* d, void* stream);
/* Split-K form for skinny problems (the FC layers of ConvPnPNet, conv_pnp_net.py:101-104,158-162: M = batch rows,
 * K up to 8192): ksplit K-slices are computed by separate workgroups into `workspace`
 * (rdpn6d_conv_splitk_ws_floats(d, ksplit) floats) and reduced in a fixed order by a second kernel that applies the
 * epilogue.  Needs a linear output geometry (osy=osx=1, no phase offsets). */
long long rdpn6d_conv_splitk_ws_floats(const rdpn6d_conv_desc* d, int ksplit);
int rdpn6d_conv2d_splitk_f32(const rdpn6d_conv_desc* d, int ksplit, float* workspace, void* stream);
/* tile configuration (BM x BN) the launcher picks for this descriptor - used to attribute rocprof
 * kernel names / roofline figures to layers; rdpn6d_conv_force_tile(0,0) restores the heuristic */
int rdpn6d_conv_tile_for(const rdpn6d_conv_desc* d, int* bm, int* bn);
void rdpn6d_conv_force_tile(int bm, int bn);
/* K order of the implicit GEMM: 1 (default) = taps innermost inside a 16-channel chunk, 0 = tap-major (A/B switch
 * kept for profiling; results are identical up to fp32 summation order) */
void rdpn6d_conv_set_tap_inner(int v);

/* Reduced-precision mode of the same operator (the reference's autocast path: engine.py:279, gdrn_evaluator.py:625):
 * x and w are bf16 (16-bit) arrays; y and res are bf16 or - when out_f32 != 0 - both fp32; scale/shift are fp32; all
 * channel strides/offsets are in elements.  Cin % 32 == 0, in_cs % 8 == 0, in_co % 8 == 0.  fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, one rounding (RNE) on the store. */
int rdpn6d_conv2d_bf16(const rdpn6d_conv_desc* d, int out_f32, void* stream);
/* split-K form, as rdpn6d_conv2d_splitk_f32 (same workspace size, linear output geometry) */
int rdpn6d_conv2d_splitk_bf16(const rdpn6d_conv_desc* d, int out_f32, int ksplit, float* workspace, void* stream);
/* Training forward of a conv + BatchNorm pair: the convolution (16-bit output, no split-K) whose epilogue also writes the BatchNorm
 * statistics' partial sums - per channel (sum, sum of squares) of the values AS STORED, rows stats_row0 .. stats_row0 + *stats_rows - 1
 * of a [rows][N][2] double array.  *stats_rows = 0: the geometry does not allow it (ragged tiles, unaligned slices), the convolution ran
 * normally and the caller uses rdpn6d_bn_train_stats_bf16.  rdpn6d_bn_stats_finalize (below) finishes the statistics. */
int rdpn6d_conv2d_bf16_bnstats(const rdpn6d_conv_desc* d, double* stats, int stats_row0, int* stats_rows, void* stream);
/* Training backward of a (BatchNorm + ReLU) -> conv pair: the input-gradient convolution of the LATER layer, whose output dy is the gradient
 * w.r.t. the BatchNorm's activation; its epilogue also writes the BatchNorm's backward partial sums - per channel (sum g, sum g * xhat), the
 * ReLU mask re-derived from the BatchNorm input bn_x as in rdpn6d_bn_relu_backward_* - rows [*rows][N][2] doubles (*rows = 0: geometry
 * not eligible, plain convolution done).  rdpn6d_bn_relu_backward_apply_bf16 turns the rows into dgamma / dbeta and runs the dx pass. */
int rdpn6d_conv2d_bf16_bnbwd(const rdpn6d_conv_desc* d, const void* bn_x, int bn_cs, int bn_co, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, double* partial, int* rows, void* stream);
int rdpn6d_bn_relu_backward_apply_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, void* dx,
                                       int xgcs, int xgco, long long M, int C, const double* partial, int S, void* stream);
/* The same pair for the LAST BatchNorm of a residual block, y = relu(bn(x) + identity) (torchvision BasicBlock / Bottleneck: out += identity;
 * relu): the input-gradient convolution that writes the gradient w.r.t. the block output (its own residual input added in the epilogue)
 * also writes that BatchNorm's backward sums with the mask y > 0 read from the STORED block output bn_y, and the apply step finishes as
 * rdpn6d_bn_backward_bf16(relu = 1) would (dres = the masked gradient).  *rows == 0: the launch took a kernel without this epilogue. */
int rdpn6d_conv2d_bf16_bnbwd_y(const rdpn6d_conv_desc* d, const void* bn_x, int bn_cs, int bn_co, const void* bn_y, int y_cs, int y_co,
                               const float* mean, const float* invstd, double* partial, int* rows, void* stream);
int rdpn6d_bn_backward_apply_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const void* y, int ycs, int yco,
                                  const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dx,
                                  int xgcs, int xgco, void* dres, int rcs, int rco, long long M, int C, const double* partial, int S,
                                  void* stream);
void rdpn6d_conv_bf16_force_tile(int bm, int bn);
/* profiling: LDS stages of the 4-wave tiles (0 = the heuristic: 3 for 64x64 tiles with >= 64 K-chunks, else 2) */
void rdpn6d_conv_bf16_force_stages(int nst);
int rdpn6d_conv_bf16_tile_for(const rdpn6d_conv_desc* d, int* bm, int* bn);
/* 1 when rdpn6d_conv2d_bf16 runs this problem on the eight-wave ping-pong kernel (csrc/conv_igemm_bf16_pp.hip: N % 128 == 0, Cin % 64 == 0,
 * >= 224 tiles of 128x128 / 256x128, the layer not taken by the 256x256 eight-phase kernel) */
int rdpn6d_conv_bf16_uses_pingpong(const rdpn6d_conv_desc* d, int out_f32);
/* fp32-ACCURATE convolution on the bf16 matrix pipe ("bf16x3", csrc/conv_igemm_bf16x3.hip): every fp32 operand is held as
 * three bf16 planes (a = a1 + a2 + a3, 24 significand bits) and a product is the six partial products a_i*b_j, i+j <= 4,
 * accumulated in fp32 - the dropped terms are <= 2^-26 relative, below one fp32 rounding.  Same operator and descriptor as
 * rdpn6d_conv2d_f32 (cuDNN fp32 convolution in the reference), with
 *   d->x = plane 0 of the activation planes  [3][x_plane_elems] bf16 (each plane NHWC [B,H,W,in_cs]),
 *   d->w = plane 0 of the packed weight planes [3][w_plane_elems] bf16 (each [Npad][ntaps][Cin]),
 *   d->y = fp32 output (may be NULL), d->res = fp32 residual, y_planes = optional [3][y_plane_elems] bf16 planes of the
 *   result (the next layer's input).  Requirements: rdpn6d_conv_bf16x3_eligible(d) != 0.
 * rdpn6d_split_bf16x3 converts n fp32 values to the three planes (plane_elems >= n, multiple of 8). */
int rdpn6d_split_bf16x3(const float* x, long long n, void* planes, long long plane_elems, void* stream);
int rdpn6d_conv_bf16x3_eligible(const rdpn6d_conv_desc* d);
/* which kernel would run: 2 = 256x256 8-phase (N % 256 == 0, >= 160 tiles), 1 = 128x128..64x64 tile kernel, 0 = none */
int rdpn6d_conv_bf16x3_kernel_for(const rdpn6d_conv_desc* d);
int rdpn6d_conv2d_bf16x3(const rdpn6d_conv_desc* d, long long x_plane_elems, long long w_plane_elems, void* y_planes,
                         long long y_plane_elems, void* stream);
/* ... with the residual given as three bf16 planes [3][res_plane_elems] (geometry d->res_cs / d->res_co, d->res == NULL) */
int rdpn6d_conv2d_bf16x3_ex(const rdpn6d_conv_desc* d, long long x_plane_elems, long long w_plane_elems, void* y_planes,
                            long long y_plane_elems, const void* res_planes, long long res_plane_elems, void* stream); /* 256x128 | 128x128 | 128x64 | 64x128 | 64x64 */
/* fp32-ACCURATE convolution on the fp16 matrix pipe with TWO planes per operand ("h2", csrc/conv_igemm_h2.hip): a * 2^s = hi + lo
 * in fp16 (22 significand bits), a product = lo*hi + hi*lo + hi*hi (three exact partial products, fp32 accumulation): half the
 * MFMA work and half the accumulator roundings of bf16x3 - measured error vs fp64 no larger than the fp32-MFMA kernel's.
 * Same operator and descriptor as rdpn6d_conv2d_f32 with
 *   d->x = activation as an h2 tensor [pixels][in_cs/32][2][32] fp16 holding a*16 (|a| < 4094),
 *   d->w = packed weights as an h2 tensor [Npad][ntaps][Cin/32][2][32] fp16 holding w * 2^sw(n) (the caller folds
 *          2^-(sw(n)+4) into d->scale), d->y = fp32 output (may be NULL), d->res = fp32 residual,
 *   y_h2 / res_h2 = optional h2 tensors of the result (the next layer's input) / of the residual,
 *   overflow_flag = device int set to 1 if an output had to be clamped to the fp16 range (never an inf).
 * Requirements: Cin, in_cs, in_co % 32 == 0; N % 8 == 0; Npad % 64 == 0; 16-byte aligned output slices; tensors < 4 GiB.
 * rdpn6d_split_h2 converts an fp32 NHWC channel slice to an h2 tensor. */
int rdpn6d_split_h2(const float* x, int src_cs, int src_co, int C, void* dst, long long npix, int* overflow_flag, void* stream);
int rdpn6d_conv_h2_kernel_for(const rdpn6d_conv_desc* d); /* 2 = 256x256 eight-phase, 1 = 128x128..64x64 tile kernel, 0 = not eligible */
int rdpn6d_conv2d_h2(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, void* stream);
/* the same with a per-crop bias [B][4][Npad] fp32 added after scale / shift (before residual and activation): row (b, v) with
 * v = (output row == OH-1) * 2 + (output column == OW-1); NULL = rdpn6d_conv2d_h2.  Used by the ConvTranspose phases of the head for
 * the spatially constant half of their input (below). */
int rdpn6d_conv2d_h2_cb(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                        void* stream);
/* ... and a workspace: a launch too small to fill the chip (per-image batches: one crop's layer4 is 64 rows x 4608 reductions) cuts
 * K into slices that write fp32 partial tiles; a second launch adds them in slice order (deterministic) and runs the epilogue.
 * rdpn6d_conv_h2_workspace_bytes = what the layer wants (0: it does not split); launches of one stream can share a buffer; without
 * a (large enough) workspace the call is rdpn6d_conv2d_h2_cb. */
long long rdpn6d_conv_h2_workspace_bytes(const rdpn6d_conv_desc* d);
/* Weights once more, FRAGMENT-MAJOR, for the kernels that load their weight fragments straight from L2 instead of staging the weight
 * tile through LDS (the LDS port is what bounds the trunk's K loops; layer3 at B = 64): rdpn6d_h2_weight_frag re-orders an h2 weight
 * tensor [Npad][ntaps][cchunks][hi x 32 | lo x 32] into [Npad/32][ntaps][cchunks][slot 0..7][row 0..31][8 halfs] (same bytes);
 * rdpn6d_conv_h2_wfrag_wanted says whether a layer's kernel has that form; rdpn6d_conv2d_h2_wf = rdpn6d_conv2d_h2 + the re-ordered
 * weights (null or an unwanted w_frag: the ordinary kernel).  Results are bit-identical with and without. */
/* measurement only: while `buf` (device memory, 4 x uint64) is set, workgroup 0 of every 256x256 eight-phase h2 launch leaves
 * {s_memtime, s_memrealtime} at its start and end there: ticks of the shader clock against the constant 100 MHz counter = the clock the
 * power-limited dominant kernel ran at (bench.py roofline.clock_ghz); NULL switches it off */
void rdpn6d_conv_h2_set_clock_probe(unsigned long long* buf);
void rdpn6d_conv_h2_set_wfrag(int mode); /* 0 (default): no layer wants them - measured slower, kept for the record; 1: 128x128; 2: + 256x128 */
/* Column-max form: for a layer whose output only a per-group channel max reads (resnet_backbone.py:51-52: the point-wise branch's
 * adaptive max over a crop's pixels, when nothing else needs the layer's output).  keys [groups][Npad] uint64 (zero before the call)
 * receive the h2 record of max over the group's rows of scale * conv + shift; rdpn6d_h2_colmax_decode turns them into the record
 * rdpn6d_global_max_h2 would have produced from the written activation and zeroes them again.  The activation is never written. */
int rdpn6d_conv_h2_colmax_ok(const rdpn6d_conv_desc* d, int rows_per_group);
int rdpn6d_conv2d_h2_colmax(const rdpn6d_conv_desc* d, unsigned long long* keys, int rows_per_group, int* overflow_flag, void* stream);
int rdpn6d_h2_colmax_decode(unsigned long long* keys, int groups, int N, int Npad, void* out_h2, void* stream);
int rdpn6d_conv_h2_wfrag_wanted(const rdpn6d_conv_desc* d);
int rdpn6d_h2_weight_frag(const void* w_h2, int Npad, int ntaps, int cchunks, void* w_frag, void* stream);
int rdpn6d_conv2d_h2_wf(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const void* w_frag, void* stream);
int rdpn6d_conv2d_h2_ws(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                        void* workspace, long long workspace_bytes, void* stream);
/* The same convolution with a 1x1 OUTPUT convolution fused into its epilogue - the dense head's last 3x3 layer + features.21
 * (core/gdrn_modeling/models/cdpn_rot_head_region.py:130-138): out[pixel][n] = scale1[n] * sum_c act(conv)[pixel][c] * w1[n][c] +
 * bias1[n], n < n_out; the 256-channel activation is never written.  w1_h2: h2 records [64][N/32][hi|lo] of the [64][1][N] fp32 matrix
 * (rows >= n_out zero); scale1 / bias1: 64 floats; out fp32 [pixels][out_cs], out_cs % 8 == 0, n_out <= out_cs <= 64; desc.y must be
 * NULL.  Needs the 256x256 kernel with one tile across N: rdpn6d_conv_h2_fuse1x1_ok(desc) != 0. */
int rdpn6d_conv_h2_fuse1x1_ok(const rdpn6d_conv_desc* d);
int rdpn6d_conv2d_h2_fuse1x1(const rdpn6d_conv_desc* d, const void* res_h2, int* overflow_flag, const float* crop_bias,
                             const void* w1_h2, const float* scale1, const float* bias1, float* out, int out_cs, int n_out,
                             void* stream);
/* h2 forms of the kernels between the h2 convolutions of the point-wise fusion branch (same argument meaning as the _f32 entry
 * points; activations are h2 tensors, C / out_cs / out_co multiples of 32; the xyz subsample fills one whole 32-channel group
 * [x y z 0 ...]; csrc/pointwise_h2.hip) */
int rdpn6d_upsample_bilinear_h2(const void* x, int B, int H, int W, int C, int factor, void* y, int* overflow_flag, void* stream);
/* ... into the channel slice [out_co, out_co + C) of an h2 tensor with out_cs channels per pixel, ReLU optional: the second half of
 * "1x1 convolution + BatchNorm at the low resolution, then up-sample + ReLU" (= up-sample, convolution, BatchNorm, ReLU of
 * resnet_backbone.py:280 / :46: the interpolation weights sum to 1, so the affine map commutes with it) */
int rdpn6d_upsample_bilinear_h2_ex(const void* x, int B, int H, int W, int C, int factor, void* y, int out_cs, int out_co, int relu,
                                   int* overflow_flag, void* stream);
int rdpn6d_xyz_subsample_h2(const float* x, int B, int xc, int R, int step, void* y, int out_cs, int out_co, int* overflow_flag,
                            void* stream);
int rdpn6d_global_max_concat_h2(void* buf, int B, int HW, int C, int cs, void* stream);
/* md_pointnet's concat [l3 | broadcast(global max l3)] (resnet_backbone.py:51-52) feeds a ConvTranspose2d(3, 2, 1, output_padding 1)
 * (cdpn_rot_head_region.py): a spatially constant input contributes a per-crop constant per output parity and border position, so
 * the broadcast half need not exist.  rdpn6d_global_max_h2: the max over the pixels of channels [0, C) of an h2 tensor as an h2
 * record per crop, gmax_h2 [B][C/32][2][32].  A one-pixel h2 convolution of it with the weights of the constant half gives
 * V [B][9*F] fp32, V[b][(ky*3+kx)*F + n] = sum_c W[C+c][n][ky][kx] * g[b][c]; rdpn6d_convt3x3s2_const_bias_f32 adds the valid taps:
 * out[py*2+px][b][v][n] = scale[n] * sum (v as in rdpn6d_conv2d_h2_cb: the last output row / column lack the ky / kx = 0 tap). */
int rdpn6d_global_max_h2(const void* x_h2, int B, int HW, int C, int cs, void* gmax_h2, void* stream);
int rdpn6d_convt3x3s2_const_bias_f32(const float* V, const float* scale, int B, int F, float* out /* [4][B][4][F] */, void* stream);
/* fused front of the network for the h2 path: conv1 7x7/2 + folded BN + ReLU + MaxPool2d(3,2,1) (resnet_backbone.py:272-275)
 * as an implicit GEMM on the fp16 matrix pipe (two-plane arithmetic, fp32-accurate), writing the pooled activation
 * [B, R/4, R/4, 64] as an h2 tensor.  x [B, xc, R, R] fp32 NCHW (channels 0..2 used), 16-byte aligned, R % 4 == 0 (the input
 * patch is read with aligned 16-byte loads); w_h2 [64][6][2][32] fp16 = the conv1 weights with the reduction index
 * k = (c*7+ky)*8+kx padded to 192 (zeros at kx = 7 and k >= 168; gdrn.pack_stem_h2_weight); scale = BN scale * 2^-(sw(n)+4),
 * shift = BN shift. */
int rdpn6d_stem_pool_h2(const float* x, int B, int xc, int R, const void* w_h2, const float* scale, const float* shift, void* y,
                        int* overflow_flag, void* stream);
/* the same kernel with the pooled activation stored as out_fmt: 0 = h2 tensor (above), 1 = bf16, 2 = fp16 NHWC [B, R/4, R/4, 64] - the
 * front of the 16-bit inference mode (cfg.TEST.AMP_TEST; gdrn_evaluator.py:625): fp32-accurate arithmetic, one rounding on the store */
int rdpn6d_stem_pool_h2_ex(const float* x, int B, int xc, int R, const void* w_h2, const float* scale, const float* shift, void* y,
                           int out_fmt, int* overflow_flag, void* stream);
/* conv1 weights OIHW [64][3][7][7] fp32 -> the h2 weight tensor of rdpn6d_stem_pool_h2* ([64][6][2][32] fp16, reduction index
 * k = (c*7 + ky)*8 + kx padded to 192) and the per-channel factor 2^-sw(n) / 16 to use as its `scale` (times a folded BatchNorm scale, if
 * any): one launch - the training step re-packs after every optimizer step.  rdpn6d_stem_pool_h2_ex(out_fmt = 3 | 4) is the RAW stem
 * convolution (no ReLU, no pooling) as a bf16 / fp16 NHWC tensor [B, R/2, R/2, 64]: the mixed-precision training forward. */
int rdpn6d_stem_pack_h2(const float* w_oihw, void* w_h2, float* inv_scale, void* stream);
void rdpn6d_conv_bf16_force_chunk(int row_bytes); /* profiling: 0 = auto, 64 forces 32-channel K-chunks (measured slower) */
/* bf16 forms of the kernels between the bf16 convolutions (same argument meaning as the _f32 entry points; activations
 * bf16 NHWC with C % 8 == 0; the stem and the xyz subsample read the fp32 NCHW crop and write bf16) */
int rdpn6d_stem_conv7x7_bf16(const float* x, int B, int xc, int R, const float* w, const float* scale,
                             const float* shift, void* y, void* stream);
int rdpn6d_maxpool3x3s2_bf16(const void* x, int B, int H, int W, int C, void* y, void* stream);
int rdpn6d_upsample_bilinear_bf16(const void* x, int B, int H, int W, int C, int factor, void* y, void* stream);
int rdpn6d_xyz_subsample_bf16(const float* x, int B, int xc, int R, int step, void* y, int out_cs, int out_co,
                              void* stream);
int rdpn6d_global_max_concat_bf16(void* buf, int B, int HW, int C, int cs, void* stream);
/* fp32 NHWC channel slice -> compact bf16 NHWC (RNE): dst[p][c] = src[p*src_cs + src_co + c] for c < C, 0 for
 * C <= c < dst_cs (dst_cs % 8 == 0).  Feeds the bf16 convolutions of the mixed-precision training step. */
int rdpn6d_cast_f32_bf16(const float* src, int src_cs, int src_co, int C, void* dst, int dst_cs, long long npix,
                         void* stream);

/* ------------------------------------------------------------------ stem & point-wise kernels
 * conv1 7x7/2 + BN + ReLU on channels 0..2 of the NCHW 6-channel crop (resnet_backbone.py:272,
 * :304,:321-323).  x NCHW [B,xc,R,R]; w [64][7][7][3]; y NHWC [B,R/2,R/2,64].             */
int rdpn6d_stem_conv7x7_f32(const float* x, int B, int xc, int R, const float* w, const float* scale,
                            const float* shift, float* y, void* stream);
/* MaxPool2d(3,2,1) (resnet_backbone.py:275), NHWC, C multiple of 4 */
int rdpn6d_maxpool3x3s2_f32(const float* x, int B, int H, int W, int C, float* y, void* stream);
/* UpsamplingBilinear2d(scale_factor=4) == align_corners=True (resnet_backbone.py:280,:332) */
int rdpn6d_upsample_bilinear_f32(const float* x, int B, int H, int W, int C, int factor, float* y, void* stream);
/* F.interpolate(xyz,(R/8,R/8),'nearest') == x[:,3:6,::8,::8] (resnet_backbone.py:304-306),
 * written into channels [out_co, out_co+3) of an NHWC buffer                              */
int rdpn6d_xyz_subsample_f32(const float* x, int B, int xc, int R, int step, float* y, int out_cs, int out_co,
                             void* stream);
/* adaptive_max_pool2d->(1,1) + broadcast + concat (resnet_backbone.py:51-54): reads channels
 * [0,C) of buf NHWC [B,HW,cs], writes max over HW into channels [C,2C)                    */
int rdpn6d_global_max_concat_f32(float* buf, int B, int HW, int C, int cs, void* stream);
/* GroupNorm(G, C) + ReLU (conv_pnp_net.py:80-82), NHWC in place; eps 1e-5                 */
int rdpn6d_groupnorm_relu_f32(float* x, int B, int HW, int C, int G, const float* gamma, const float* beta,
                              void* stream);
/* GroupNorm(G, C) + ReLU reading the fp32 tensor (left unchanged) and writing the result as an h2 tensor [B*HW][C/32][hi|lo] */
int rdpn6d_groupnorm_relu_h2(const float* x, int B, int HW, int C, int G, const float* gamma, const float* beta, void* y_h2,
                             int* overflow_flag, void* stream);

/* ------------------------------------------------------------------ dense-map glue
 * GDRN.forward:196-233 + get_mask_prob (model_utils.py:24-42) + ConvPnPNet input assembly
 * (conv_pnp_net.py:129-137) in one pass over the head output.
 *   head NHWC [B,HW,head_cs] with channels [mask | x y z | region bg+K];
 *   coord2d NCHW [B,5,HW]; fps [B,K,3]
 *   out_nchw [B,4+K+1,HW]  (the reference's mask/coor_x/coor_y/coor_z/region views)
 *   pnp_in NHWC [B,HW,pnp_cs] = [x y z | coord2d(5) | anchor(3) | softmax(K) | 0-pad] (* mask attention)
 *   argmax_out [B,HW] int32 (region index 0..K-1), may be NULL
 * mask_attention: 0 none, 1 mul (min-max normalised mask, no epsilon as in the reference).  */
int rdpn6d_dense_glue_f32(const float* head, int head_cs, const float* coord2d, const float* fps, int B, int HW,
                          int K, int mask_attention, float* minmax_scratch, float* out_nchw, float* pnp_in,
                          int pnp_cs, int* argmax_out, void* stream);
/* The same pass with the ConvPnPNet input as an h2 tensor [B*HW][pnp_cs/32][hi x 32 | lo x 32] fp16 (16 * value; pnp_cs % 32 == 0,
 * zero padded) for ConvPnPNet on the fp16 matrix pipe (conv_pnp_net.py:129-137 -> rdpn6d_conv2d_h2); a value outside the format's
 * range (|v| > 4094, inf, NaN) raises *overflow_flag like every other h2 writer. */
int rdpn6d_dense_glue_h2(const float* head, int head_cs, const float* coord2d, const float* fps, int B, int HW, int K,
                         int mask_attention, float* minmax_scratch, float* out_nchw, void* pnp_in_h2, int pnp_cs,
                         int* argmax_out, int* overflow_flag, void* stream);
/* Both with the mask read as ROT_HEAD.MASK_LOSS_TYPE prescribes (get_mask_prob, models/model_utils.py:24-42; the entry points above
 * are mask_type 0): mask_type 0 "L1" = per-crop min-max, 1 "BCE" = sigmoid, 2 "CE" = TWO mask channels - head / out_nchw rows are then
 * [mask0 mask1 | x y z | region bg+K] (get_xyz_mask_region_out_dim, GDRN.py:637-659) and mask_attention must be 0: the reference's own
 * CE branch raises (torch.softmax(..., keepdim=True), model_utils.py:39). */
int rdpn6d_dense_glue_mt_f32(const float* head, int head_cs, const float* coord2d, const float* fps, int B, int HW, int K,
                             int mask_attention, int mask_type, float* minmax_scratch, float* out_nchw, float* pnp_in, int pnp_cs,
                             int* argmax_out, void* stream);
int rdpn6d_dense_glue_mt_h2(const float* head, int head_cs, const float* coord2d, const float* fps, int B, int HW, int K,
                            int mask_attention, int mask_type, float* minmax_scratch, float* out_nchw, void* pnp_in_h2, int pnp_cs,
                            int* argmax_out, int* overflow_flag, void* stream);

/* ------------------------------------------------------------------ pose decode
 * ortho6d_to_mat_batch (core/utils/rot_reps.py:34-49) + pose_from_predictions_test
 * (models/pose_from_pred_centroid_z.py:52-141) + allocentric_to_egocentric
 * (core/utils/utils.py:39-94, fp64 axis-angle like the numpy original).
 *   rt [B,rt_stride] = [rot6d(6) | centroid dx dy | z_rel]; out rot [B,9], trans [B,3]
 * train_variant != 0 selects pose_from_predictions_train / allo_to_ego_mat_torch (eps 1e-4). */
int rdpn6d_pose_decode_f32(const float* rt, int rt_stride, const float* roi_cams, const float* roi_centers,
                           const float* roi_whs, const float* resize_ratios, int B, int is_allo, int train_variant,
                           float* rot, float* trans, void* stream);

/* ------------------------------------------------------------------ per-crop RANSAC + Kabsch
 * New capability named by the north star (no reference implementation; replaces the role of
 * cv2.solvePnPRansac at lib/pysixd/misc.py:170-179 for the RGB-D residual formulation):
 * correspondences (anchor[region_i], P_i - delta_i) per foreground pixel, one hypothesis per
 * wavefront, inlier counts on an LDS scoreboard, adaptive stop, Kabsch refit on the inliers.
 *   out_nchw [B,4+K+1,HW] dense maps; coord2d [B,5,HW] (ch 0..2 = depth xyz / resize_ratio)
 *   pose_out [B,12] = R row-major | t;   n_inliers [B];   inlier_mask [B,HW] uint8 (may be NULL)
 * A crop with fewer than 3 usable correspondences gets the sentinel pose -100 (as the reference
 * does for n<4, gdrn_evaluator.py:391-392). */
int rdpn6d_ransac_kabsch_f32(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                             const float* resize_ratios, const int* region_argmax, int B, int HW, int K,
                             float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed,
                             float* pose_out, int* n_inliers, unsigned char* inlier_mask, void* stream);
/* same, additionally reporting the index of the winning hypothesis per crop (-1 = none) */
int rdpn6d_ransac_kabsch_ex(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                            const float* resize_ratios, const int* region_argmax, int B, int HW, int K,
                            float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed,
                            float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp, void* stream);
/* Network-initialised solve = process_net_and_pnp (gdrn_evaluator.py:187-314; cfg.TEST.PNP_TYPE "net_ransac_pnp" /
 * "net_iter_pnp").  net_pose [B,12] = the learned pose (R row-major | t), device memory, must not alias pose_out.
 * mode 1: net_pose is hypothesis 0 next to iters-1 sampled ones (the reference passes iterationsCount=20), then the
 * inlier refit; mode 2: one closed-form least-squares fit over ALL selected correspondences (the role of
 * solvePnP(ITERATIVE)).  Fewer than 3 correspondences -> net_pose (:297-300); |t - t_net| > max_t_diff -> t_net
 * (:293-296, 1.0 there). */
int rdpn6d_ransac_kabsch_net_f32(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                                 const float* resize_ratios, const int* region_argmax, const float* net_pose, int B,
                                 int HW, int K, float mask_thr, float inlier_thr, int iters, float confidence,
                                 unsigned seed, int mode, float max_t_diff, float* pose_out, int* n_inliers,
                                 unsigned char* inlier_mask, int* best_hyp, void* stream);
/* The same solves with a caller-provided workspace of rdpn6d_ransac_workspace_bytes(B) bytes: with fewer crops than CUs the hypotheses
 * of a crop are spread over up to 4 workgroups (global scoreboard) and a second launch scans + refits - bit-identical results.
 * net_pose NULL = the plain solve; else mode 1 / 2 as above. */
long long rdpn6d_ransac_workspace_bytes(int B);
int rdpn6d_ransac_kabsch_ws(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                            const float* resize_ratios, const int* region_argmax, const float* net_pose, int B, int HW, int K,
                            float mask_thr, float inlier_thr, int iters, float confidence, unsigned seed, int mode,
                            float max_t_diff, float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp,
                            void* workspace, long long workspace_bytes, void* stream);
/* ... with the mask read as ROT_HEAD.MASK_LOSS_TYPE prescribes (engine_utils.get_out_mask): mask_type 0 L1 min-max (every other
 * RANSAC entry point), 1 BCE sigmoid, 2 CE arg-max over two mask channels (out_nchw [B, 2 + 3 + K + 1, HW]) */
int rdpn6d_ransac_kabsch_ws_mt(const float* out_nchw, const float* coord2d, const float* fps, const float* extents,
                               const float* resize_ratios, const int* region_argmax, const float* net_pose, int B, int HW, int K,
                               float mask_thr, int mask_type, float inlier_thr, int iters, float confidence, unsigned seed, int mode,
                               float max_t_diff, float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp,
                               void* workspace, long long workspace_bytes, void* stream);


/* ================================================================== training step (forward with batch statistics,
 * losses, backward).  Rows L / O of SURVEY.md section 8a: GDRN.gdrn_loss (models/GDRN.py:373-633), the autograd graph
 * PyTorch builds for GDRN.forward(do_loss=True), torch BatchNorm2d/GroupNorm train-mode semantics.
 * dgrad of every conv reuses rdpn6d_conv2d_f32 with re-packed (flipped / transposed / phase-split) weights. */

/* conv1 7x7/2 raw output (no folded BN, no ReLU) for training */
int rdpn6d_stem_conv7x7_raw_f32(const float* x, int B, int xc, int R, const float* w, float* y, void* stream);
/* BatchNorm2d train mode, statistics over the M rows of x[M, cs] channels [co, co+C): mean, 1/sqrt(var+eps) (biased var),
 * running stats updated with `momentum` (unbiased var) when given.  scratch >= 512*C*2 doubles. */
int rdpn6d_bn_train_stats_f32(const float* x, long long M, int C, int cs, int co, float eps, float momentum, float* mean,
                              float* invstd, float* running_mean, float* running_var, double* scratch, void* stream);
/* the second half of rdpn6d_bn_train_stats_*: mean / invstd / running statistics from S rows [S][C][2] of per-channel (sum, sum of
 * squares) partials over M values per channel */
int rdpn6d_bn_stats_finalize(const double* partial, int S, int C, long long M, float eps, float momentum, float* mean, float* invstd,
                             float* running_mean, float* running_var, void* stream);
/* y = act((x-mean)*invstd*gamma + beta (+ res)) */
int rdpn6d_bn_apply_f32(const float* x, int xcs, int xco, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const float* res, int rcs, int rco, float* y, int ycs, int yco, long long M, int C,
                        int relu, void* stream);
/* BN backward (g = dy*(y>0) if relu): dgamma, dbeta, dx; dres (optional) = g for the identity branch */
int rdpn6d_bn_backward_f32(const float* x, int xcs, int xco, const float* dy, int dcs, int dco, const float* y, int ycs,
                           int yco, const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                           float* dx, int xgcs, int xgco, float* dres, int rcs, int rco, long long M, int C, int relu,
                           double* scratch, void* stream);
/* BN + ReLU backward (no residual between them) WITHOUT the stored activation: the mask y > 0 is re-derived from x with the forward's
 * own expression and rounding - one tensor read less in both passes */
int rdpn6d_bn_relu_backward_f32(const float* x, int xcs, int xco, const float* dy, int dcs, int dco, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dx,
                                int xgcs, int xgco, long long M, int C, double* scratch, void* stream);
/* out[c] (+)= sum over rows of x[m, co+c]  (bias gradients) */
int rdpn6d_channel_sum_f32(const float* x, long long M, int C, int cs, int co, float* out, int accumulate, double* scratch,
                           void* stream);
/* GroupNorm(G, C=4G)+ReLU, out of place, statistics [B][G][2] saved; and its backward */
int rdpn6d_groupnorm_relu_train_f32(const float* x, float* y, int B, int HW, int C, int G, const float* gamma,
                                    const float* beta, float* stats, void* stream);
int rdpn6d_groupnorm_relu_backward_f32(const float* x, const float* y, const float* dy, const float* gamma, const float* stats,
                                       float* dx, float* dgamma, float* dbeta, float* dgb_scratch, double* scratch, int B,
                                       int HW, int C, int G, void* stream);
/* weight gradient as an implicit GEMM reducing over pixels: out[a][t][b] = sum_m A[m][a] * Bg[gather(m,t)][b] */
long long rdpn6d_wgrad_scratch_floats(int Bn, int Ha, int Wa, int Ca, int Cb, int ntaps);
int rdpn6d_wgrad_f32(const float* A, int a_cs, int a_co, int Ca, const float* Bg, int b_cs, int b_co, int Cb, int Bn, int Ha,
                     int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx, float* out, float* partial,
                     void* stream);
/* bf16 form for the mixed-precision training step: A and Bg are compact bf16 NHWC copies (channel strides / offsets in
 * elements, multiples of 8); Ca_ld / Cb_ld = readable channels of the slices (multiples of 8, >= Ca / Cb, zero beyond the
 * real count); fp32 accumulation, fp32 out / partial exactly as rdpn6d_wgrad_f32 (same scratch size). */
int rdpn6d_wgrad_bf16(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co, int Cb,
                      int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx,
                      float* out, float* partial, void* stream);
/* the same two kernels with the result scattered straight into a caller-defined layout (e.g. the parameter's own OIHW
 * gradient): element (a, tap, b) -> out[a*sa + tap*st + b*sb] for a < Ca_out <= Ca, b < Cb_out <= Cb */
int rdpn6d_wgrad_f32_strided(const float* A, int a_cs, int a_co, int Ca, const float* Bg, int b_cs, int b_co, int Cb, int Bn,
                             int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx, float* out,
                             long long sa, long long st, long long sb, int Ca_out, int Cb_out, float* partial, void* stream);
int rdpn6d_wgrad_bf16_strided(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co, int Cb,
                              int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                              const int* dx, float* out, long long sa, long long st, long long sb, int Ca_out, int Cb_out,
                              float* partial, void* stream);
/* G (1..16) weight gradients of ONE geometry - the same-shaped k x k convolutions of a ResNet stage - in one launch and one reduce:
 * the chip is filled by the problems' tiles instead of by K-splits of each.  A_list / B_list / out_list: HOST arrays of G device
 * pointers (gradient w.r.t. the output, input activation, OIHW gradient = the strided target of rdpn6d_wgrad_bf16_strided with
 * st = 1, sb = ntaps); partial: at least rdpn6d_wgrad_group_scratch_floats(G, ...) floats. */
long long rdpn6d_wgrad_group_scratch_floats(int G, int Bn, int Ha, int Wa, int Ca, int Cb, int ntaps);
int rdpn6d_wgrad_bf16_group(int G, const void* const* A_list, int a_cs, int a_co, int Ca, int Ca_ld, const void* const* B_list, int b_cs,
                            int b_co, int Cb, int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                            const int* dx, float* const* out_list, long long sa, long long st, long long sb, int Ca_out, int Cb_out,
                            float* partial, long long partial_floats, void* stream);
/* bf16x3 form (fp32-accurate weight gradient on the bf16 matrix pipe, see rdpn6d_conv2d_bf16x3): A / Bg = plane 0 of the three
 * bf16 planes [3][a_plane_elems] / [3][b_plane_elems] of the NHWC gradient / activation (rdpn6d_split_bf16x3); Ca, Cb > 64 */
int rdpn6d_wgrad_bf16x3_strided(const void* A, long long a_plane_elems, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg,
                                long long b_plane_elems, int b_cs, int b_co, int Cb, int Cb_ld, int Bn, int Ha, int Wa, int Hb,
                                int Wb, int stride, int ntaps, const int* dy, const int* dx, float* out, long long sa,
                                long long st, long long sb, int Ca_out, int Cb_out, float* partial, void* stream);
int rdpn6d_maxpool3x3s2_backward_f32(const float* x, const float* dy, int B, int H, int W, int C, float* dx, void* stream);
int rdpn6d_upsample_bilinear_backward_f32(const float* dy, int B, int H, int W, int C, int factor, float* dx, void* stream);
int rdpn6d_global_max_concat_backward_f32(const float* feat, const float* dfeat, int B, int HW, int C, int cs, float* dl3,
                                          void* stream);
/* dense losses (loss_coor_x/y/z, loss_mask, loss_region, loss_region_my) and d(loss)/d(head) in one pass */
int rdpn6d_dense_losses_f32(const float* head, int head_cs, const float* gt_xyz, const float* mask_visib,
                            const float* mask_trunc, const long long* gt_region, int B, int HW, int K, float xyz_lw,
                            float mask_lw, float region_lw, float* dhead, float* losses, double* scratch, void* stream);
int rdpn6d_dense_glue_backward_f32(const float* head, int head_cs, const float* coord2d, const float* fps, const int* argmax,
                                   const float* dpnp, int pnp_cs, int B, int HW, int K, int mask_attention,
                                   const float* minmax, float* dhead, float* datt_scratch, void* stream);
/* ... with ROT_HEAD.MASK_LOSS_TYPE (GDRN.py:450-463, models/model_utils.py:24-42): mask_type 0 "L1" (the two entries above), 1 "BCE"
 * (nn.BCEWithLogitsLoss mean; sigmoid mask attention), 2 "CE" (nn.CrossEntropyLoss over TWO mask channels, mean; head rows are then
 * [mask0 mask1 | x y z | region bg+K] and mask_attention must be 0 - the reference's get_mask_prob raises there) */
int rdpn6d_dense_losses_mt_f32(const float* head, int head_cs, const float* gt_xyz, const float* mask_visib, const float* mask_trunc,
                               const long long* gt_region, int B, int HW, int K, float xyz_lw, float mask_lw, float region_lw,
                               int mask_type, float* dhead, float* losses, double* scratch, void* stream);
int rdpn6d_dense_glue_backward_mt_f32(const float* head, int head_cs, const float* coord2d, const float* fps, const int* argmax,
                                      const float* dpnp, int pnp_cs, int B, int HW, int K, int mask_attention, int mask_type,
                                      const float* minmax, float* dhead, float* datt_scratch, void* stream);
/* pose decode (train variant) + loss_PM_R, loss_centroid, loss_z and their gradient w.r.t. the 9 head outputs */
int rdpn6d_pose_train_f32(const float* rt, int rt_stride, const float* roi_cams, const float* roi_centers,
                          const float* roi_whs, const float* resize_ratios, const float* roi_extents, const float* gt_rot,
                          const float* gt_trans_ratio, const float* points, int npts, int B, int is_allo, float pm_lw,
                          int pm_norm_by_extent, float centroid_lw, float z_lw, float* rot, float* trans, float* d_rt,
                          float* losses, float* scratch, void* stream);
/* Same with PNP_NET.PM_LOSS_SYM (losses/pm_loss.py:97-99 -> core/utils/pose_utils.py:430-482, which the reference runs on
 * the host with a device->host copy of every predicted rotation): sym_rots [B][ksym][9] holds each crop's model-to-model
 * symmetry rotations, sym_counts[b] <= ksym how many are valid (0 = not symmetric); per crop the target becomes the
 * candidate among {Rgt, Rgt*S_k} closest (rotation error, first best wins) to the predicted rotation.  gt_rot_used
 * [B][9] (optional) receives the chosen targets. */
int rdpn6d_pose_train_sym_f32(const float* rt, int rt_stride, const float* roi_cams, const float* roi_centers,
                              const float* roi_whs, const float* resize_ratios, const float* roi_extents,
                              const float* gt_rot, const float* gt_trans_ratio, const float* points, int npts, int B,
                              int is_allo, float pm_lw, int pm_norm_by_extent, float centroid_lw, float z_lw,
                              const float* sym_rots, const int* sym_counts, int ksym, float* gt_rot_used, float* rot,
                              float* trans, float* d_rt, float* losses, float* scratch, void* stream);
/* The per-step training scalars the reference pushes to detectron2's EventStorage (core/gdrn_modeling/models/GDRN.py:306-328:
 * compute_mean_re_te of models/model_utils.py:45-57 = mean re [deg] / te over the batch, lib/pysixd/pose_error.py:400-440, and
 * sixteen .item() reads of crop 0), on the device: out17 = {error_R, error_t [cm], error_tx/ty/tz [cm], t_pred xyz, pred_t_ xyz
 * (rt columns 6..8), t_gt xyz, gt_trans_ratio xyz}.  No host synchronisation. */
int rdpn6d_train_vis_scalars_f32(const float* rot, const float* trans, const float* gt_rot, const float* gt_trans, const float* rt,
                                 int rt_stride, const float* gt_trans_ratio, int B, float* out17, void* stream);
/* Fused multi-tensor Ranger step over flat buffers (replaces lib/torch_utils/solver/ranger.py:100-200).
 * work = array of {int64 off; int32 len; int32 row} runs (row = index of the centralisation mean, -1 = none; row >= 0x40000000: the run IS
 * a whole centralisation row and the update kernel takes its mean itself - no entry in row_off / row_len for it);
 * row_off/row_len describe the rows whose gradient mean is subtracted (gradient centralisation);
 * neg_step_lr = -step_size*lr and rectified = (N_sma > threshold) are computed on the host from the step count. */
int rdpn6d_ranger_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* slow, const void* work,
                           int nwork, const long long* row_off, const int* row_len, int nrows, float* row_mean, float beta1,
                           float beta2, float eps, float neg_step_lr, float wd_lr, int rectified, int lookahead, float alpha,
                           void* stream);
/* The same step under a loss scale (the reference: GradScaler.unscale_ / GradScaler.step around the optimizer, engine.py:302-309)
 * without separate passes over the gradients: every gradient is read as grad * inv_scale (the buffer keeps the scaled values); with
 * found_inf != NULL the whole step is skipped on the device when *found_inf != 0.  rdpn6d_grad_nonfinite_f32 computes that flag:
 * *flag = 1 if any of grad[0 .. n) is NaN or +-Inf, else 0 (grad 16-byte aligned). */
int rdpn6d_ranger_step_scaled_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* slow, const void* work,
                                  int nwork, const long long* row_off, const int* row_len, int nrows, float* row_mean, float beta1,
                                  float beta2, float eps, float neg_step_lr, float wd_lr, int rectified, int lookahead, float alpha,
                                  float inv_scale, const int* found_inf, void* stream);
int rdpn6d_grad_nonfinite_f32(const float* grad, long long n, int* flag, void* stream);
int rdpn6d_act_backward_f32(float* dy, const float* y, long long n, float slope, void* stream);
/* y[b][c][r] = x[b][r][c] (B matrices of R x C floats): the last ConvPnPNet map in the reference's NCHW-flatten order for fc1
 * (conv_pnp_net.py:151: x.view(-1, featdim * 8 * 8) of an NCHW tensor), and its gradient back - instead of permuting fc1's 8.4 M weights every step */
int rdpn6d_transpose_rc_f32(const float* x, int B, int R, int C, float* y, void* stream);
int rdpn6d_rgb_to_nhwc4_f32(const float* x, int B, int xc, int R, float* y, void* stream);
/* patch matrix of the stem for its weight gradient: out [B*(R/2)^2][160], column (ky*7+kx)*3+c = x[b][c][2oy-3+ky][2ox-3+kx]
 * (0 outside the image; columns 147..159 zero) - dW(conv1) is then one rdpn6d_wgrad_f32 call with a single tap */
int rdpn6d_stem_im2col_f32(const float* x, int B, int xc, int R, float* out, void* stream);
/* row-patch matrix of the stem (what the training step uses): out [B*(R/2)^2][64], column r*32 + kx*3 + c = x[b][c][2j+r][2ox-3+kx]
 * for pixel (j, ox) (0 outside the image; columns 21..31 of each half zero) - dW(conv1) is then one stride-1 rdpn6d_wgrad_* call
 * with the four taps dy = -2..1, dx = 0 over the (R/2) x (R/2) map: dW[n][c][ky][kx] = out4[n][t][r*32 + kx*3 + c], r = (ky+1)&1,
 * t = (ky+1-r)/2 (resnet_backbone.py:272 backward) */
int rdpn6d_stem_rowpatch_f32(const float* x, int B, int xc, int R, float* out, void* stream);

/* Weight re-packing of the training step in one launch.  Entry: dst[(o*dT + t)*dIpad + i] = src[operm(o)*so +
 * iperm(i)*si + toff[t]] for o < O, t < T, i < I (operm / iperm may be NULL = identity; dst and/or dst_bf16 are written;
 * padding entries of dst are never touched).  Workgroup b handles 1 024 consecutive (o, i) pairs (2 048 when T == 1) - all T taps of
 * each - of entry blk_desc[b], starting at pair blk_off[b] (pair = o*I + i); with bit 30 of blk_desc[b] set (operm == iperm == NULL
 * only) it handles the tile of 64 i x 16 o (64 o when T <= 2) whose first pair is blk_off[b], moved through LDS - the form for
 * entries whose consecutive i are far apart in src (si > so).  Table and maps live in device memory (`start` is unused by the
 * kernel, kept for bookkeeping). */
typedef struct {
    const float* src;
    float* dst;
    void* dst_bf16;
    const int* operm;
    const int* iperm;
    long long so, si, start;
    int O, T, I, dT, dIpad;
    int toff[9];
} rdpn6d_repack_desc;
int rdpn6d_repack_f32(const rdpn6d_repack_desc* table_dev, const int* blk_desc_dev, const long long* blk_off_dev, int nblocks,
                      void* stream);

/* bf16-stored training activations (mixed-precision step, cfg.SOLVER.AMP.ENABLED): the same kernels as their _f32
 * namesakes with every activation / gradient tensor (x, y, dy, dx, res, dres, feat ...) in bf16 - strides and offsets in
 * elements, multiples of 4 - and fp32 statistics, parameters, parameter gradients and arithmetic. */
int rdpn6d_bn_train_stats_bf16(const void* x, long long M, int C, int cs, int co, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, double* scratch, void* stream);
int rdpn6d_bn_apply_bf16(const void* x, int xcs, int xco, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const void* res, int rcs, int rco, void* y, int ycs, int yco, long long M, int C,
                         int relu, void* stream);
int rdpn6d_bn_backward_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const void* y, int ycs, int yco,
                            const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dx,
                            int xgcs, int xgco, void* dres, int rcs, int rco, long long M, int C, int relu, double* scratch,
                            void* stream);
int rdpn6d_bn_relu_backward_bf16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, void* dx,
                                 int xgcs, int xgco, long long M, int C, double* scratch, void* stream);
int rdpn6d_channel_sum_bf16(const void* x, long long M, int C, int cs, int co, float* out, int accumulate, double* scratch,
                            void* stream);
int rdpn6d_maxpool3x3s2_backward_bf16(const void* x, const void* dy, int B, int H, int W, int C, void* dx, void* stream);
int rdpn6d_upsample_bilinear_backward_bf16(const void* dy, int B, int H, int W, int C, int factor, void* dx, void* stream);
int rdpn6d_global_max_concat_backward_bf16(const void* feat, const void* dfeat, int B, int HW, int C, int cs, void* dl3,
                                           void* stream);
int rdpn6d_stem_im2col_bf16(const float* x, int B, int xc, int R, void* out, void* stream);
int rdpn6d_stem_rowpatch_bf16(const float* x, int B, int xc, int R, void* out, void* stream);
int rdpn6d_stem_conv7x7_raw_bf16(const float* x, int B, int xc, int R, const float* w, void* y, void* stream);

/* ================================================================== "next" rows of SURVEY.md section 8f
 * rank 3: region / residual training targets (core/utils/data_utils.py:229-244, data_loader.py:881-903).
 *   xyz_hwc [B,HW,3] f32 (model-space crop, 0 = background), fps [B,K,3] f64 (the loader's float64 anchors),
 *   rot [B,9] f32 (GT pose), extent [B,3] -> roi_xyz [B,3,HW] f32, roi_region [B,HW] int64 (0 = bg, 1..K) */
int rdpn6d_region_targets_f32(const float* xyz_hwc, const double* fps, const float* rot, const float* extent, int B, int HW,
                              int K, float* roi_xyz_chw, long long* roi_region, void* stream);
/* rank 4: ADD / ADI / re (deg) / te per pose in float64 (lib/pysixd/pose_error.py:297-337,400-436).
 *   est, gt [B,12] = R row-major | t; pts [n,3] shared (pts_per_pose = 0) or [B,n,3]; out [B,4] = add, adi, re, te */
int rdpn6d_pose_errors_f64(const double* est, const double* gt, const double* pts, int pts_per_pose, int n, int B,
                           double* scratch, double* out, void* stream);
/* row A8: 2D-3D correspondence selection in front of the PnP solve, on device (core/gdrn_modeling/engine_utils.py:102-136
 * get_out_coor / get_out_mask, core/gdrn_modeling/gdrn_evaluator.py:89-126 get_img_model_points_with_coords2d).
 *   out_nchw [B,C,HW] f32: channel 0 = mask, 1..3 = coor_x/y/z (what GDRN.forward returns); coord2d [B,C2,HW] f32 with the
 *   normalised u / v coordinates in channels u_ch / v_ch (RDPN's roi_coord_2d: C2 = 5, u_ch = 3, v_ch = 4; the reference call
 *   site passes channels 0 / 1 "as given"); extents [B,3]; im_hw [B,2] int (H, W) per crop or NULL -> im_H, im_W for all.
 *   -> image_points [B,HW,2] px, model_points [B,HW,3] m (first counts[b] rows valid, ROW-MAJOR pixel order = numpy's boolean
 *   gather), counts [B]; optional sel_mask [B,HW] u8 and out_mask [B,HW] (the min-max normalised mask; NaN for a constant
 *   mask like the reference).  Bit-exact vs the reference functions (tests/golden/select_golden.npz). */
int rdpn6d_select_correspondences_f32(const float* out_nchw, int C, const float* coord2d, int C2, int u_ch, int v_ch,
                                      const float* extents, const int* im_hw, int im_H, int im_W, int B, int HW, float mask_thr,
                                      float* image_points, float* model_points, int* counts, unsigned char* sel_mask,
                                      float* out_mask, void* stream);
/* ... with get_out_mask's other branches (engine_utils.py:118-136): mask_type 0 "L1" min-max (= the entry above), 1 "BCE" sigmoid,
 * 2 "CE" arg-max over TWO mask channels (channels 0, 1 = mask, 2..4 = coor_x/y/z; out_mask = 0.0 / 1.0).  Pinned by
 * tests/golden/mask_types_golden.npz (the reference's own functions). */
int rdpn6d_select_correspondences_mt_f32(const float* out_nchw, int C, const float* coord2d, int C2, int u_ch, int v_ch,
                                         const float* extents, const int* im_hw, int im_H, int im_W, int B, int HW, float mask_thr,
                                         int mask_type, float* image_points, float* model_points, int* counts,
                                         unsigned char* sel_mask, float* out_mask, void* stream);
/* rows A9 / A10: 2D-3D RANSAC-PnP on device, the role of lib/pysixd/misc.py:145-194 pnp_v2 -> cv2.solvePnPRansac(EPnP, 3 px, 100 it.)
 * at gdrn_evaluator.py:316-435 and of process_net_and_pnp (:187-314).  image_points [B,HW,2] px / model_points [B,HW,3] m / counts [B]
 * = the output of rdpn6d_select_correspondences_f32; cams [B,9] K row-major; net_pose [B,12] (R row-major | t) or NULL.
 *   mode 0  RANSAC: one hypothesis per wavefront (Lambda-Twist P3P on 3 + 1 correspondences, fp64), inliers by reprojection
 *           error < reproj_thr from LDS, confidence-driven stop, Gauss-Newton refit on the winner's inliers; < 4
 *           correspondences or no model = the -100 sentinel pose;
 *   mode 1  net_pose is hypothesis 0 (useExtrinsicGuess; the reference runs 20 iterations);  mode 2  Gauss-Newton from net_pose
 *           over all correspondences (SOLVEPNP_ITERATIVE); both keep net_pose below 4 correspondences and its translation
 *           when the solved one moved by more than max_t_diff.
 * -> pose_out [B,12], n_inliers [B], inlier_mask [B,HW] (indexed like the lists), best_hyp [B].  Masks / counts / winner are
 * bit-exact vs oracle/pnp_oracle.c under a fixed seed; parity with cv2 is unpinned (cv2 absent). */
int rdpn6d_ransac_pnp_f32(const float* image_points, const float* model_points, const int* counts, const float* cams,
                          const float* net_pose, int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed, int mode,
                          float max_t_diff, float* pose_out, int* n_inliers, unsigned char* inlier_mask, int* best_hyp, void* stream);
/* ... with the minimal solver as an argument (cfg.TEST.PNP_MINIMAL): minimal 0 = the entry above (P3P + 1 on sets of 4, Gauss-Newton
 * refit); minimal 1 = EPnP - what the reference's call names: cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_EPNP), lib/pysixd/misc.py:170-179
 * (called from gdrn_evaluator.py:261-292 / :386-389): minimal sets of FIVE, each solved by EPnP (control points, barycentric
 * coordinates, null space of M^T M by a 12 x 12 Jacobi on the wavefront's LDS scratch, betas + Gauss-Newton, Horn), confidence stop
 * on w^5, and a final EPnP over the inliers of the best model (no iterative refinement).  A crop with exactly four correspondences
 * takes the P3P + 1 path in either mode (EPnP's null space is four-dimensional there).  Masks / counts / winner bit-exact vs
 * oracle_ransac_pnp_ex (oracle/pnp_oracle.c, restated from the EPnP paper; cv2 absent: PARITY UNPINNED), the refit pose to ~1e-9. */
int rdpn6d_ransac_pnp_ex(const float* image_points, const float* model_points, const int* counts, const float* cams,
                         const float* net_pose, int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed,
                         int mode, float max_t_diff, int minimal, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                         int* best_hyp, void* stream);
/* ... and with a caller-provided workspace (rdpn6d_ransac_pnp_workspace_bytes(B) bytes of device memory, or NULL = the entry above):
 * with fewer crops than compute units each crop's hypotheses are spread over up to ceil(iters / 8) workgroups (a global scoreboard, a
 * second launch for scan + refit) - the reference's test loop feeds the 1 - 15 crops of one image per call; results bit-identical to
 * the one-launch form. */
long long rdpn6d_ransac_pnp_workspace_bytes(int B);
int rdpn6d_ransac_pnp_ws(const float* image_points, const float* model_points, const int* counts, const float* cams,
                         const float* net_pose, int B, int HW, float reproj_thr, int iters, float confidence, unsigned seed,
                         int mode, float max_t_diff, int minimal, float* pose_out, int* n_inliers, unsigned char* inlier_mask,
                         int* best_hyp, void* workspace, long long workspace_bytes, void* stream);
/* rank 1: GPU crop builder (core/gdrn_modeling/data_loader.py:523-627, core/utils/data_utils.py:81-152; cv2.warpAffine
 * bilinear arithmetic restated, parity with cv2 unpinned).  images [N,H,W,3] u8, depths [N,H,W] f32; per ROI: image index,
 * inverse affine maps for the R and R/4 crops (6 doubles each), fx fy cx cy of (A @ K), resize_ratio ->
 * roi_img [B,6,R,R], roi_coord_2d [B,5,R/4,R/4] */
int rdpn6d_crop_builder_f32(const unsigned char* images, const float* depths, int N, int H, int W, const int* img_idx,
                            const double* inv_in, const double* inv_out, const double* Knew, const double* ratio, int B, int R,
                            float* roi_img, float* roi_coord_2d, void* stream);


/* ================================================================== IEEE fp16 twins of the 16-bit entry points
 * The reference's mixed precision is torch.cuda.amp.autocast + GradScaler in FLOAT16 (core/gdrn_modeling/engine.py:279-309,
 * main_gdrn.py:143 precision=16; gdrn_evaluator.py:625 for AMP_TEST).  Every rdpn6d_*_bf16 entry point above exists a second
 * time as rdpn6d_*_fp16 with identical arguments: the same source compiled with the 16-bit storage format = IEEE half and
 * v_mfma_f32_32x32x16_f16 (csrc/common.h, rdpn6d_amd/build.py).  rdpn6d_repack_fp16 = rdpn6d_repack_f32 writing fp16 mirrors.
 * Selected by cfg.SOLVER.AMP.DTYPE / cfg.TEST.AMP_DTYPE = "fp16" (default "bf16"). */
int rdpn6d_conv2d_fp16(const rdpn6d_conv_desc* d, int out_f32, void* stream);
int rdpn6d_conv2d_splitk_fp16(const rdpn6d_conv_desc* d, int out_f32, int ksplit, float* workspace, void* stream);
int rdpn6d_conv2d_fp16_bnbwd(const rdpn6d_conv_desc* d, const void* bn_x, int bn_cs, int bn_co, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, double* partial, int* rows, void* stream);
int rdpn6d_bn_relu_backward_apply_fp16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, void* dx,
                                       int xgcs, int xgco, long long M, int C, const double* partial, int S, void* stream);
int rdpn6d_conv2d_fp16_bnbwd_y(const rdpn6d_conv_desc* d, const void* bn_x, int bn_cs, int bn_co, const void* bn_y, int y_cs, int y_co,
                               const float* mean, const float* invstd, double* partial, int* rows, void* stream);
int rdpn6d_bn_backward_apply_fp16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const void* y, int ycs, int yco,
                                  const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dx,
                                  int xgcs, int xgco, void* dres, int rcs, int rco, long long M, int C, const double* partial, int S,
                                  void* stream);
int rdpn6d_conv2d_fp16_bnstats(const rdpn6d_conv_desc* d, double* stats, int stats_row0, int* stats_rows, void* stream);
void rdpn6d_conv_fp16_force_tile(int bm, int bn);
void rdpn6d_conv_fp16_force_stages(int nst);
int rdpn6d_conv_fp16_tile_for(const rdpn6d_conv_desc* d, int* bm, int* bn);
int rdpn6d_conv_fp16_uses_pingpong(const rdpn6d_conv_desc* d, int out_f32);
void rdpn6d_conv_fp16_force_chunk(int row_bytes);
int rdpn6d_stem_conv7x7_fp16(const float* x, int B, int xc, int R, const float* w, const float* scale,
                             const float* shift, void* y, void* stream);
int rdpn6d_maxpool3x3s2_fp16(const void* x, int B, int H, int W, int C, void* y, void* stream);
int rdpn6d_upsample_bilinear_fp16(const void* x, int B, int H, int W, int C, int factor, void* y, void* stream);
int rdpn6d_xyz_subsample_fp16(const float* x, int B, int xc, int R, int step, void* y, int out_cs, int out_co,
                              void* stream);
int rdpn6d_global_max_concat_fp16(void* buf, int B, int HW, int C, int cs, void* stream);
int rdpn6d_cast_f32_fp16(const float* src, int src_cs, int src_co, int C, void* dst, int dst_cs, long long npix,
                         void* stream);
int rdpn6d_wgrad_fp16(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co, int Cb,
                      int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx,
                      float* out, float* partial, void* stream);
int rdpn6d_wgrad_fp16_strided(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co, int Cb,
                              int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                              const int* dx, float* out, long long sa, long long st, long long sb, int Ca_out, int Cb_out,
                              float* partial, void* stream);
int rdpn6d_wgrad_fp16_group(int G, const void* const* A_list, int a_cs, int a_co, int Ca, int Ca_ld, const void* const* B_list, int b_cs,
                            int b_co, int Cb, int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                            const int* dx, float* const* out_list, long long sa, long long st, long long sb, int Ca_out, int Cb_out,
                            float* partial, long long partial_floats, void* stream);
int rdpn6d_bn_train_stats_fp16(const void* x, long long M, int C, int cs, int co, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, double* scratch, void* stream);
int rdpn6d_bn_apply_fp16(const void* x, int xcs, int xco, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const void* res, int rcs, int rco, void* y, int ycs, int yco, long long M, int C,
                         int relu, void* stream);
int rdpn6d_bn_relu_backward_fp16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, void* dx,
                                 int xgcs, int xgco, long long M, int C, double* scratch, void* stream);
int rdpn6d_bn_backward_fp16(const void* x, int xcs, int xco, const void* dy, int dcs, int dco, const void* y, int ycs, int yco,
                            const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dx,
                            int xgcs, int xgco, void* dres, int rcs, int rco, long long M, int C, int relu, double* scratch,
                            void* stream);
int rdpn6d_channel_sum_fp16(const void* x, long long M, int C, int cs, int co, float* out, int accumulate, double* scratch,
                            void* stream);
int rdpn6d_maxpool3x3s2_backward_fp16(const void* x, const void* dy, int B, int H, int W, int C, void* dx, void* stream);
int rdpn6d_upsample_bilinear_backward_fp16(const void* dy, int B, int H, int W, int C, int factor, void* dx, void* stream);
int rdpn6d_global_max_concat_backward_fp16(const void* feat, const void* dfeat, int B, int HW, int C, int cs, void* dl3,
                                           void* stream);
int rdpn6d_stem_im2col_fp16(const float* x, int B, int xc, int R, void* out, void* stream);
int rdpn6d_stem_rowpatch_fp16(const float* x, int B, int xc, int R, void* out, void* stream);
int rdpn6d_stem_conv7x7_raw_fp16(const float* x, int B, int xc, int R, const float* w, void* y, void* stream);
int rdpn6d_repack_fp16(const rdpn6d_repack_desc* table_dev, const int* blk_desc_dev, const long long* blk_off_dev, int nblocks,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
