// Shared pieces of the bf16 convolution kernels (conv_igemm_bf16.hip, conv3x3_halo_bf16.hip): launch arguments and the
// fused epilogue (scale/shift, residual, activation; coalesced 16-byte stores through an LDS transpose for full tiles).
#pragma once
#include "common.h"

struct ConvBArgs {
    rdpn6d_conv_desc d;
    long long M;
    int HoWo;
    int cchunks;  // Cin*2 / RB
    int nk;       // ntaps * cchunks
    int Ktot;     // ntaps * Cin
    int mtiles, ntiles;
    int linear_out;
    int out_f32;
    unsigned x_bytes, w_bytes;
    unsigned long long dy_pack, dx_pack;
    int kper;        // split-K: K-chunks per split (blockIdx.y = split); nk when not split
    float* partial;  // split-K: raw fp32 accumulators [split][M][Npad]; nullptr = fused epilogue
    int vec_out;  // 16-byte aligned output / residual channel slices: coalesced epilogue through LDS
    // Training forward (rdpn6d_conv2d_bf16_bnstats): per-channel (sum, sum of squares) of the STORED 16-bit results, one row of
    // [N][2] doubles per wave row of an M tile (row = stats_row0 + mtile * WM + wm) - the BatchNorm statistics' partial sums
    // without a second pass over the tensor.  Only the coalesced epilogue writes them (the launcher checks the geometry).
    double* stats = nullptr;
    int stats_row0 = 0;
    // Training backward (rdpn6d_conv2d_bf16_bnbwd): this launch is the input-gradient convolution whose output dy IS the gradient
    // w.r.t. the activation of a BatchNorm + ReLU (no residual in between).  With bnb_x set, the rows written to `stats` are that
    // BatchNorm's backward sums instead: per channel (sum g, sum g * xhat), g = dy as stored where the re-derived activation is
    // positive (bn_fwd_value / bn_stored_positive, csrc/common.h), xhat from the BatchNorm's input bnb_x - chan_partial_kernel<1>'s pass
    // over dy and x, done while the dy tile is in registers.
    const void* bnb_x = nullptr;
    int bnb_cs = 0, bnb_co = 0;
    const float *bnb_mean = nullptr, *bnb_invstd = nullptr, *bnb_gamma = nullptr, *bnb_beta = nullptr;
    // ... or of a residual block's last BatchNorm (rdpn6d_conv2d_bf16_bnbwd_y): dy is the gradient w.r.t. the block OUTPUT
    // y = relu(bn(x) + identity), so the mask is the stored y > 0 (the STATS == 2 instantiation reads bnb_y next to bnb_x; beta unused)
    const void* bnb_y = nullptr;
    int bnb_ycs = 0, bnb_yco = 0;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef rd_bf16_t bf16_t;
#define f2bf rd_f2bf
#define bf2f rd_bf2f

// activation of N values behind ONE wave-uniform branch (a per-element `if (act == ..)` inside the unrolled loops made
// the compiler shuffle the whole value array through a branch ladder: ~75 instructions per output)
template <int N>
__device__ __forceinline__ void conv_bf16_act(float (&v)[N], const int act, const float slope)
{
    if (act == 1) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
    } else if (act == 2) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * slope;
    }
}

// Coalesced form of the fused epilogue (full column tiles, 16-byte aligned channel slices).  MASK_ROWS: rows >= M of a
// ragged last tile are skipped (their accumulators are zeros from out-of-range input rows).
// STATS: the instantiation that also writes the BatchNorm partial sums (ConvBArgs::stats / bnb_x) - a template flag, not a run-time
// branch, because its registers (parameters and pre-loaded rows of 8 channels) would otherwise raise the VGPR count of EVERY kernel that
// shares this epilogue: the 64x128 inference tile went from 124 to 134 VGPRs = from four to three wavefronts per SIMD (-2.4 % crops/s).
template <int BM, int BN, int WM, int WN, int TM, int TN, bool MASK_ROWS, int STATS = 0>
__device__ __forceinline__ void conv_bf16_epilogue_vec(const ConvBArgs& a, f32x16 (&acc)[TM][TN], unsigned char* smem,
                                                       const long long m0, const int n0, const int wave, const int lane,
                                                       const int wm, const int wn)
{
    const rdpn6d_conv_desc& d = a.d;
    const int frow = lane & 31;
    const int hi = lane >> 5;
    const bf16_t* resb = reinterpret_cast<const bf16_t*>(d.res);
    bf16_t* yb = reinterpret_cast<bf16_t*>(d.y);
    // Coalesced form for full tiles: the accumulator layout has one channel per lane and pixels across registers, i.e. a
    // direct store moves 2 bytes per lane.  Each wave instead transposes its tile through its own slice of the (now
    // idle) staging LDS - fp32, scale/shift already applied - and walks it back row-wise: 8 channels = 16 bytes per
    // lane for the residual load and the store.  Same fp32 operations in the same order as the scalar path below.
    {
        constexpr int WC = BN / WN;       // channels per wave tile
        constexpr int CS = WC + 8;        // LDS row stride in floats (+32 B: the two half-waves hit disjoint banks)
        __syncthreads();                  // every wave is done with the staging buffers (and no DMA is still landing)
        float* cst = reinterpret_cast<float*>(smem) + wave * (32 * CS);
        const int nb = n0 + wn * WC;
        auto pixel_of = [&](const long long m) -> long long {
            if (a.linear_out) return m;
            const int mm = (int)m;
            const int b = mm / a.HoWo;
            const int rem = mm - b * a.HoWo;
            const int oy = rem / d.Wo;
            const int ox = rem - oy * d.Wo;
            return ((long long)b * d.OH + (oy * d.osy + d.ooy)) * d.OW + (ox * d.osx + d.oox);
        };
        float scj[TN], shj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nb + j * 32 + frow;
            scj[j] = d.scale ? d.scale[n] : 1.f;
            shj[j] = d.shift ? d.shift[n] : 0.f;
        }
        float st1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, st2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // a.stats
        float bmu[8], bis[8], bga[8], bbe[8];  // a.bnb_x: the BatchNorm parameters of this lane's 8 channels
        constexpr int NRB = 32 / (64 / (WC / 8));  // row groups per 32-row accumulator tile in the 16-bit walk
        rd_u32x4 bxq[TM][NRB];                     // ... and the BatchNorm-input rows of this lane's outputs, ALL requested up front (one
                                                   // dependent load per row group cost the 256x256 kernel 34 us per head layer)
        rd_u32x4 byq[STATS == 2 ? TM : 1][STATS == 2 ? NRB : 1];  // STATS == 2: the stored block-output rows (the ReLU mask) as well
        if (STATS && a.stats && a.bnb_x && !a.out_f32) {
            const int c = nb + (lane % (WC / 8)) * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                bmu[q] = a.bnb_mean[c + q];
                bis[q] = a.bnb_invstd[c + q];
                if constexpr (STATS != 2) {
                    bga[q] = a.bnb_gamma[c + q];
                    bbe[q] = a.bnb_beta[c + q];
                }
            }
            constexpr int LPRB = WC / 8, RPIB = 64 / LPRB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rr = 0; rr < NRB; ++rr) {
                    const long long mrow = m0 + wm * (BM / WM) + i * 32 + rr * RPIB + lane / LPRB;
                    const long long mr = mrow < a.M ? mrow : 0;  // (linear geometry: pixel = row; rows past M are skipped below)
                    bxq[i][rr] = *reinterpret_cast<const rd_u32x4*>(reinterpret_cast<const bf16_t*>(a.bnb_x) + mr * a.bnb_cs + a.bnb_co + c);
                    if constexpr (STATS == 2)
                        byq[i][rr] = *reinterpret_cast<const rd_u32x4*>(reinterpret_cast<const bf16_t*>(a.bnb_y) + mr * a.bnb_ycs + a.bnb_yco + c);
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e] * scj[j] + shj[j];
            }
            if (a.out_f32) {  // fp32 output (training: raw conv results / input gradients): 4 channels = 16 bytes per lane
                constexpr int LPR = WC / 4, RPI = 64 / LPR;
                const int rrow = lane / LPR, c4 = (lane % LPR) * 4;
#pragma unroll
                for (int rr = 0; rr < 32 / RPI; ++rr) {
                    const int row = rr * RPI + rrow;
                    const long long mrow = m0 + wm * (BM / WM) + i * 32 + row;
                    if (MASK_ROWS && mrow >= a.M) continue;
                    const long long pix = pixel_of(mrow);
                    const f32x4 cv = *reinterpret_cast<const f32x4*>(cst + row * CS + c4);
                    float v[4] = {cv[0], cv[1], cv[2], cv[3]};
                    if (d.res) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(d.res + pix * d.res_cs + d.res_co + nb + c4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += rv[q];
                    }
                    conv_bf16_act(v, d.act, d.slope);
                    const f32x4 ov = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(d.y + pix * d.out_cs + d.out_co + nb + c4) = ov;
                }
            } else {
                constexpr int LPR = WC / 8, RPI = 64 / LPR;
                const int rrow = lane / LPR, c8 = (lane % LPR) * 8;
#pragma unroll
                for (int rr = 0; rr < 32 / RPI; ++rr) {
                    const int row = rr * RPI + rrow;
                    const long long mrow = m0 + wm * (BM / WM) + i * 32 + row;
                    if (MASK_ROWS && mrow >= a.M) continue;
                    const long long pix = pixel_of(mrow);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    if (resb) {
                        float rv[8];
                        rd_unpack8(*reinterpret_cast<const rd_u32x4*>(resb + pix * d.res_cs + d.res_co + nb + c8), rv);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rv[q];
                    }
                    conv_bf16_act(v, d.act, d.slope);
                    const rd_u32x4 pk = rd_pack8(v);
                    *reinterpret_cast<rd_u32x4*>(yb + pix * d.out_cs + d.out_co + nb + c8) = pk;
                    if (STATS && a.stats) {  // of the values as stored (what a BatchNorm reading the tensor would see)
                        float r[8];
                        rd_unpack8(pk, r);
                        if (a.bnb_x) {  // backward sums of the BatchNorm + ReLU this gradient flows into (linear geometry: pix = row)
                            float xr[8];
                            rd_unpack8(bxq[i][rr], xr);
                            if constexpr (STATS == 2) {  // residual block: the mask is the stored block output (chan_partial_kernel's relu == 1)
                                float yr[8];
                                rd_unpack8(byq[i][rr], yr);
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    const float g = yr[q] > 0.f ? r[q] : 0.f;
                                    st1[q] += g;
                                    st2[q] += g * ((xr[q] - bmu[q]) * bis[q]);
                                }
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    const float g = bn_stored_positive<bf16_t>(bn_fwd_value(xr[q], bmu[q], bis[q], bga[q], bbe[q])) ? r[q] : 0.f;
                                    st1[q] += g;
                                    st2[q] += g * ((xr[q] - bmu[q]) * bis[q]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                st1[q] += r[q];
                                st2[q] += r[q] * r[q];
                            }
                        }
                    }
                }
            }
        }
        if (STATS && a.stats && !a.out_f32) {
            // lanes with the same (lane % LPR) hold the same 8 channels of different rows: a fixed butterfly over the row-lane bits,
            // then lanes 0 .. LPR-1 write the wave's row of partial sums (<= (BM / WM) values per sum in fp32, fp64 from here on)
            constexpr int LPR = WC / 8;
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    st1[q] += __shfl_xor(st1[q], off, 64);
                    st2[q] += __shfl_xor(st2[q], off, 64);
                }
            if (lane < LPR) {
                const long long row = a.stats_row0 + (m0 / BM) * WM + wm;
                double* p = a.stats + (row * d.N + nb + lane * 8) * 2;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    p[2 * q] = (double)st1[q];
                    p[2 * q + 1] = (double)st2[q];
                }
            }
        }
    }
}

// acc: the wave's (BM/WM) x (BN/WN) tile as TM x TN 32x32 MFMA accumulators; smem: the workgroup's dynamic LDS (idle
// once the K loop is done); wm / wn: the wave's position in the WM x WN grid.
template <int BM, int BN, int WM, int WN, int TM, int TN, int STATS = 0>
__device__ __forceinline__ void conv_bf16_epilogue(const ConvBArgs& a, f32x16 (&acc)[TM][TN], unsigned char* smem, const long long m0,
                                                   const int n0, const int wave, const int lane, const int wm, const int wn)
{
    const rdpn6d_conv_desc& d = a.d;
    const int frow = lane & 31;
    if (a.partial) {  // split-K: raw partial sums, reduced (with the epilogue) by conv_bf16_splitk_epilogue_kernel
        float* part = a.partial + (long long)blockIdx.y * a.M * d.Npad;
        const int hi2 = lane >> 5;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + frow;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long long m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi2;
                    if (m < a.M) part[m * d.Npad + n] = acc[i][j][e];
                }
        }
        return;
    }
    // ---- fused epilogue (fp32 math, one rounding on the store)
    const int hi = lane >> 5;
    const bf16_t* resb = reinterpret_cast<const bf16_t*>(d.res);
    bf16_t* yb = reinterpret_cast<bf16_t*>(d.y);
    if (a.vec_out && n0 + BN <= d.N && m0 + BM <= a.M) {
        conv_bf16_epilogue_vec<BM, BN, WM, WN, TM, TN, false, STATS>(a, acc, smem, m0, n0, wave, lane, wm, wn);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + frow;
        const float sc = d.scale ? d.scale[n] : 1.f;
        const float sh = d.shift ? d.shift[n] : 0.f;
        const bool n_ok = n < d.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (m < a.M && n_ok) {
                    long long pix;
                    if (a.linear_out) {
                        pix = m;
                    } else {
                        const int mm = (int)m;
                        const int b = mm / a.HoWo;
                        const int rem = mm - b * a.HoWo;
                        const int oy = rem / d.Wo;
                        const int ox = rem - oy * d.Wo;
                        pix = ((long long)b * d.OH + (oy * d.osy + d.ooy)) * d.OW + (ox * d.osx + d.oox);
                    }
                    float v = acc[i][j][e] * sc + sh;
                    if (d.res) v += a.out_f32 ? d.res[pix * d.res_cs + d.res_co + n] : bf2f(resb[pix * d.res_cs + d.res_co + n]);
                    if (d.act == 1) v = v > 0.f ? v : 0.f;
                    else if (d.act == 2) v = v > 0.f ? v : v * d.slope;
                    if (a.out_f32) d.y[pix * d.out_cs + d.out_co + n] = v;
                    else yb[pix * d.out_cs + d.out_co + n] = f2bf(v);
                }
            }
        }
    }
}
