// Implicit-GEMM convolution on the gfx950 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = ntaps*Cin.
//   A[m][k] is gathered on the fly from the NHWC activation (one contiguous 64-byte run of
//   16 channels per (pixel, tap)), B[n][k] is the packed weight [Npad][ntaps][Cin].
//   Workgroup = 256 threads = 4 wavefronts (2x2); each wavefront owns a (BM/2)x(BN/2) sub-tile
//   made of 32x32 MFMA accumulators.  K advances in chunks of 16: global -> VGPR prefetch of
//   chunk k+1 is issued before the 8 MFMA k-steps of chunk k, then written to the other LDS
//   buffer (one barrier per chunk).  LDS rows are 16 data + 4 pad floats (80 B): both the
//   ds_write_b128 staging stores and the ds_read_b128 fragment loads are bank-conflict free.
//   Lane l feeds the MFMA with k = s (l < 32) and k = 8 + s (l >= 32) at step s, so each
//   lane reads its eight k-values as two 16-byte LDS loads.
//   Epilogue (fused): per-channel scale/shift (folded BatchNorm or bias), optional residual
//   add, ReLU / LeakyReLU, NHWC store with channel stride/offset (concat for free) and output
//   pixel stride/offset (the four sub-pixel phases of the stride-2 transposed conv).
//   blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous range of tiles so
//   that the N-tiles of one M-tile and neighbouring M-tiles (shared halo rows) hit one L2.
//
// Replaces the cuDNN convolutions behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear in
// core/gdrn_modeling/models/{resnet_backbone.py, cdpn_rot_head_region.py, conv_pnp_net.py}.
#include "common.h"

struct ConvKArgs {
    rdpn6d_conv_desc d;
    long long M;
    int HoWo;
    int cchunks;  // Cin / 16
    int nk;       // ntaps * cchunks
    int Ktot;     // ntaps * Cin
    int mtiles, ntiles;
    int linear_out;
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvKArgs a)
{
    constexpr int LDS = 20;           // floats per LDS row (16 + 4 pad)
    constexpr int TM = BM / 64;       // 32x32 tiles per wave along M
    constexpr int TN = BN / 64;       // ... along N
    constexpr int AR = BM / 64;       // A rows staged per thread
    constexpr int BR = BN / 64;       // B rows staged per thread
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS;

    const rdpn6d_conv_desc& d = a.d;
    // ---- XCD-aware tile mapping (bijective for any grid size)
    const int nblk = a.mtiles * a.ntiles;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
    const int nt = logical % a.ntiles;
    const int mt = logical / a.ntiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int kq = tid & 3;    // which float4 of the 16-float k-chunk this thread stages
    const int r0 = tid >> 2;   // staging row (0..63)

    // ---- per-thread A-row geometry (fixed for the whole K loop)
    int a_pix[AR], a_iy[AR], a_ix[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const long long m = m0 + r0 + 64 * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? (int)m : 0;
        const int b = mm / a.HoWo;
        const int rem = mm - b * a.HoWo;
        const int oy = rem / d.Wo;
        const int ox = rem - oy * d.Wo;
        a_pix[i] = b * d.H * d.W;
        a_iy[i] = oy * d.stride;
        a_ix[i] = ox * d.stride;
    }
    const float* wrow[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) wrow[i] = d.w + (long long)(n0 + r0 + 64 * i) * a.Ktot + kq * 4;

    f32x4 ra[AR], rb[BR];
    auto load_global = [&](int tap, int c0, int kc) {
        const int dy = d.dy[tap], dx = d.dx[tap];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            if (ok) {
                const float* p = d.x + ((long long)(a_pix[i] + iy * d.W + ix) * d.in_cs + d.in_co + c0 + kq * 4);
                ra[i] = *reinterpret_cast<const f32x4*>(p);
            } else {
                ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + kc * 16);
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + r0 + 64 * i) * LDS + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + r0 + 64 * i) * LDS + kq * 4]) = rb[i];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31;          // fragment row (A: pixel, B: channel) inside a 32-tile
    const int koff = (lane >> 5) * 8;    // this lane's 8 k-values start here

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int tap = 0, cc = 0;  // chunk kc = tap*cchunks + cc
    load_global(0, 0, 0);
    store_lds(0);
    __syncthreads();

    for (int kc = 0; kc < a.nk; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < a.nk;
        if (more) {
            if (++cc == a.cchunks) { cc = 0; ++tap; }
            load_global(tap, cc * 16, kc + 1);
        }
        f32x4 fa[TM][2], fb[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float* p = &As[(buf * BM + wm * (BM / 2) + i * 32 + frow) * LDS + koff];
            fa[i][0] = *reinterpret_cast<const f32x4*>(p);
            fa[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* p = &Bs[(buf * BN + wn * (BN / 2) + j * 32 + frow) * LDS + koff];
            fb[j][0] = *reinterpret_cast<const f32x4*>(p);
            fb[j][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s >> 2][s & 3], fb[j][s >> 2][s & 3],
                                                                     acc[i][j], 0, 0, 0);
        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- fused epilogue
    const int hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + frow;
        const float sc = d.scale ? d.scale[n] : 1.f;
        const float sh = d.shift ? d.shift[n] : 0.f;
        const bool n_ok = n < d.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
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
                    if (d.res) v += d.res[pix * d.res_cs + d.res_co + n];
                    if (d.act == 1) v = v > 0.f ? v : 0.f;
                    else if (d.act == 2) v = v > 0.f ? v : v * d.slope;
                    d.y[pix * d.out_cs + d.out_co + n] = v;
                }
            }
        }
    }
}

static int g_force_bm = 0, g_force_bn = 0;
extern "C" void rdpn6d_conv_force_tile(int bm, int bn) { g_force_bm = bm; g_force_bn = bn; }

static void conv_pick_tile(const rdpn6d_conv_desc* d, long long M, int* pbm, int* pbn)
{
    int bn = (d->Npad % 128 == 0) ? 128 : 64;
    int bm = 128;
    // small problems: prefer more, smaller tiles so that all 256 CUs get work
    if ((long long)rd_cdiv(M, 128) * (d->Npad / bn) < 512) bm = 64;
    if (bm == 64 && bn == 128 && (long long)rd_cdiv(M, 64) * (d->Npad / 128) < 512) bn = 64;
    if (g_force_bm) bm = g_force_bm;
    if (g_force_bn && d->Npad % g_force_bn == 0) bn = g_force_bn;
    *pbm = bm;
    *pbn = bn;
}

// which tile configuration rdpn6d_conv2d_f32 will use for this descriptor (for profiling / roofline)
extern "C" int rdpn6d_conv_tile_for(const rdpn6d_conv_desc* d, int* bm, int* bn)
{
    RD_REQUIRE(d && bm && bn, "null pointer");
    conv_pick_tile(d, (long long)d->B * d->Ho * d->Wo, bm, bn);
    return RDPN6D_OK;
}

extern "C" int rdpn6d_conv2d_f32(const rdpn6d_conv_desc* d, void* stream)
{
    RD_REQUIRE(d && d->x && d->w && d->y, "null pointer");
    RD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "empty tensor");
    RD_REQUIRE(d->Cin > 0 && d->Cin % 16 == 0, "Cin must be a positive multiple of 16");
    RD_REQUIRE(d->in_cs % 4 == 0 && d->in_co % 4 == 0 && d->in_co + d->Cin <= d->in_cs, "input channel slice");
    RD_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9, "ntaps in 1..9");
    RD_REQUIRE(d->N > 0 && d->Npad >= d->N && d->Npad % 64 == 0, "Npad must be a multiple of 64 >= N");
    RD_REQUIRE(d->out_co + d->N <= d->out_cs, "output channel slice");
    RD_REQUIRE(d->stride >= 1 && d->osy >= 1 && d->osx >= 1, "strides");
    RD_REQUIRE((d->Ho - 1) * d->osy + d->ooy < d->OH && (d->Wo - 1) * d->osx + d->oox < d->OW, "output geometry");
    RD_REQUIRE(!d->res || d->res_co + d->N <= d->res_cs, "residual channel slice");
    ConvKArgs a;
    a.d = *d;
    a.M = (long long)d->B * d->Ho * d->Wo;
    RD_REQUIRE(a.M < (1LL << 31), "B*Ho*Wo must fit 31 bits");
    a.HoWo = d->Ho * d->Wo;
    a.cchunks = d->Cin / 16;
    a.nk = d->ntaps * a.cchunks;
    a.Ktot = d->ntaps * d->Cin;
    a.linear_out = (d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo);
    int bm, bn;
    conv_pick_tile(d, a.M, &bm, &bn);
    a.mtiles = rd_cdiv(a.M, bm);
    a.ntiles = d->Npad / bn;
    dim3 grid((unsigned)(a.mtiles * a.ntiles)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_igemm_f32_kernel<128, 128>), grid, block, 0, s, a);
    else if (bm == 128 && bn == 64) hipLaunchKernelGGL((conv_igemm_f32_kernel<128, 64>), grid, block, 0, s, a);
    else if (bm == 64 && bn == 128) hipLaunchKernelGGL((conv_igemm_f32_kernel<64, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_f32_kernel<64, 64>), grid, block, 0, s, a);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
