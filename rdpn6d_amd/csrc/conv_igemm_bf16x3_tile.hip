// bf16x3 (fp32-accurate, see conv_igemm_bf16x3.hip) convolution for the layers the 256x256 kernel cannot take: narrow
// outputs (64 / 128 channels), few rows.  Structure of conv_igemm_bf16.hip's 2-stage form (2x2 wavefronts of 32x32
// accumulators, LDS-DMA staging with hardware bounds checking, one barrier per K-chunk, fragment double buffer, two
// workgroups per CU), with every operand as three bf16 planes: a K-chunk is RB bytes per row and plane (RB = 32: one k16
// step, for 128x128 tiles; RB = 64: two steps, for the tiles with a 64-wide side), a k16 step of a 32x32 tile pair is six
// MFMAs (plane pairs p + q <= 2) - 1.5 MFMAs per fragment read instead of 1, which is what lifts this two-barrier
// structure above its plain-bf16 efficiency.
#include "conv_x3_args.h"

#include <cstdio>
#include <cstdlib>

namespace {

template <int BM, int BN, int RB>
__global__ __launch_bounds__(256, 2) void conv_x3_tile_kernel(const ConvX3Args ax)
{
    constexpr int NW = 4, NST = 2;
    constexpr int SL = RB / 16;      // 16-byte slots per LDS row
    constexpr int RPP = 1024 / RB;   // rows per 1-KiB DMA piece
    constexpr int RPB = 256 / RB;    // rows per 256-byte bank period
    constexpr int NJ = RB / 32;      // k16 MFMA steps per chunk
    constexpr int TM = BM / 64, TN = BN / 64;              // 32x32 accumulator tiles per wave (wave tile BM/2 x BN/2)
    constexpr int AG = BM / RPP / NW, BG = BN / RPP / NW;  // DMA pieces per wave and plane
    static_assert(TM >= 1 && TN >= 1 && AG >= 1 && BG >= 1, "tile / wave layout");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // A(stage, plane) = smem + ((stage*3 + plane) * BM) * RB;  B likewise after all A stages
    unsigned char* As = smem;
    unsigned char* Bs = smem + NST * 3 * BM * RB;

    const ConvBArgs& a = ax.b;
    const rdpn6d_conv_desc& d = a.d;
    const int nblk = a.mtiles * a.ntiles;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk;
    const int nt = logical % a.ntiles;
    const int mt = logical / a.ntiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int prow = lane / SL;   // row inside the DMA piece
    const int pslot = lane % SL;  // physical slot written by this lane
    unsigned a_base[AG], a_mask[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int row = (wave + NW * i) * RPP + prow;
        const long long m = m0 + row;
        const bool ok = m < a.M;
        const int mm = ok ? (int)m : 0;
        const int b = mm / a.HoWo;
        const int rem = mm - b * a.HoWo;
        const int oy = rem / d.Wo;
        const int ox = rem - oy * d.Wo;
        const int iy = oy * d.stride, ix = ox * d.stride;
        const int lslot = pslot ^ ((row / RPB) & (SL - 1));
        a_base[i] = ((unsigned)((b * d.H + iy) * d.W + ix) * (unsigned)d.in_cs + (unsigned)d.in_co) * 2u + (unsigned)lslot * 16u;
        unsigned mask = 0;
        for (int t = 0; t < d.ntaps; ++t) {
            const int dy = (int)((a.dy_pack >> (4 * t)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * t)) & 15ull) - 8;
            mask |= (ok && (unsigned)(iy + dy) < (unsigned)d.H && (unsigned)(ix + dx) < (unsigned)d.W) ? (1u << t) : 0u;
        }
        a_mask[i] = mask;
    }
    unsigned w_off[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave + NW * i) * RPP + prow;
        const int lslot = pslot ^ ((row / RPB) & (SL - 1));
        w_off[i] = (unsigned)(n0 + row) * (unsigned)a.Ktot * 2u + (unsigned)lslot * 16u;
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);
    const int px_bytes = d.in_cs * 2;

    auto stage_chunk = [&](const int tap, const int cc, const int st) {
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned toff = (unsigned)((dy * d.W + dx) * px_bytes + cc * RB);  // wave-uniform
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const unsigned kill = ((a_mask[i] >> tap) & 1u) - 1u;  // all ones outside the image / past M: reads zeros
            const unsigned off = (a_base[i] + toff) | kill;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                unsigned char* dst = As + (((st * 3 + p) * BM) + (wave + NW * i) * RPP) * RB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)dst, 16, (int)off, (int)(p * ax.x_plane_bytes), 0, 0);
            }
        }
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 2u + (unsigned)cc * (unsigned)RB;
#pragma unroll
        for (int i = 0; i < BG; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                unsigned char* dst = Bs + (((st * 3 + p) * BN) + (wave + NW * i) * RPP) * RB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr_t)dst, 16, (int)(w_off[i] + wk), (int)(p * ax.w_plane_bytes), 0, 0);
            }
    };
    // K order: channel-chunk major, taps innermost
    const int nk = a.nk;
    int ld_cc = 0, ld_tap = 0, ld_left = nk - 1;
    auto next_chunk = [](int& tap, int& cc, int& left, const int ntaps) {
        const int go = left > 0 ? 1 : 0;
        left -= go;
        tap += go;
        const int wrap = tap == ntaps ? 1 : 0;
        tap = wrap ? 0 : tap;
        cc += wrap;
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31;
    const int half = lane >> 5;

    auto read_frags = [&](int st, u32x4 (&fa)[TM][3][NJ], u32x4 (&fb)[TN][3][NJ]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = wm * (BM / 2) + i * 32 + frow;
            const int sw = (R / RPB) & (SL - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned char* q = As + (((st * 3 + p) * BM) + R) * RB;
#pragma unroll
                for (int j = 0; j < NJ; ++j) fa[i][p][j] = *reinterpret_cast<const u32x4*>(q + (((2 * j + half) ^ sw) << 4));
            }
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int R = wn * (BN / 2) + jn * 32 + frow;
            const int sw = (R / RPB) & (SL - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned char* q = Bs + (((st * 3 + p) * BN) + R) * RB;
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[jn][p][j] = *reinterpret_cast<const u32x4*>(q + (((2 * j + half) ^ sw) << 4));
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma = [&](const u32x4 (&fa)[TM][3][NJ], const u32x4 (&fb)[TN][3][NJ]) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // plane pairs, smallest terms first
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][PA[pr]][j]),
                                                                             __builtin_bit_cast(bf16x8, fb[jn][PB[pr]][j]),
                                                                             acc[i][jn], 0, 0, 0);
    };

    u32x4 fa0[TM][3][NJ], fb0[TN][3][NJ], fa1[TM][3][NJ], fb1[TN][3][NJ];
    stage_chunk(ld_tap, ld_cc, 0);
    next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
    stage_chunk(ld_tap, ld_cc, 1);
    next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
    __syncthreads();
    read_frags(0, fa0, fb0);
    __syncthreads();  // stage 0 is re-filled by the first loop step: every wave must have its fragments first
    const int npairs = nk >> 1;
    for (int pr = 0; pr < npairs; ++pr) {
        stage_chunk(ld_tap, ld_cc, 0);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
        read_frags(1, fa1, fb1);
        mma(fa0, fb0);
        __syncthreads();

        stage_chunk(ld_tap, ld_cc, 1);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
        read_frags(0, fa0, fb0);
        mma(fa1, fb1);
        __syncthreads();
    }
    if (nk & 1) mma(fa0, fb0);

    // ---- epilogue: coalesced through an LDS transpose of one 32 x WC slice per wave (8 channels per lane)
    {
        const int hi = lane >> 5;
        constexpr int WC = BN / 2, CS = WC + 8, LPR = WC / 8, RPI = 64 / LPR;
        __syncthreads();  // the staging buffers are idle and nothing is still landing (the barriers above drain the DMA queue)
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
        const int rrow = lane / LPR, c8 = (lane % LPR) * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e] * scj[j] + shj[j];
#pragma unroll
            for (int rr = 0; rr < 32 / RPI; ++rr) {
                const int row = rr * RPI + rrow;
                const long long mrow = m0 + wm * (BM / 2) + i * 32 + row;
                if (mrow < a.M && nb + c8 < d.N) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    x3_finish_row8(ax, v, pixel_of(mrow), nb + c8);
                }
            }
        }
    }
}

template <int BM, int BN, int RB>
int launch_tile(const ConvX3Args& ax, hipStream_t s)
{
    constexpr int lds_stage = 2 * 3 * (BM + BN) * RB;
    constexpr int lds_epi = 4 * 32 * (BN / 2 + 8) * 4;
    constexpr int lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    auto kern = conv_x3_tile_kernel<BM, BN, RB>;
    if (lds > 64 * 1024) {
        RD_LDS_OPT_IN(kern, lds);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ax.b.mtiles * ax.b.ntiles)), dim3(256), lds, s, ax);
    return RDPN6D_OK;
}

}  // namespace

// tile for the small-tile bf16x3 kernel (0 = not eligible): N % 8 == 0 and Npad % 64 == 0; 128x128 needs Cin % 16 == 0, the
// tiles with a 64-wide side Cin % 32 == 0
void conv_x3_pick_tile(const rdpn6d_conv_desc* d, long long M, int* pbm, int* pbn)
{
    *pbm = *pbn = 0;
    if (d->Npad % 64 || d->N % 8 || d->Cin % 16) return;
    int bn = (d->Npad % 128 == 0) ? 128 : 64;
    int bm = 128;
    if ((long long)rd_cdiv(M, 128) * (d->Npad / bn) < 512) bm = 64;
    if (bm == 64 && bn == 128 && (long long)rd_cdiv(M, 64) * (d->Npad / 128) < 512) bn = 64;
    // 32-channel chunks (two k16 steps, 48 MFMAs per barrier) beat the 128x128 tile's 16-channel chunks: 159 vs 138 TFLOP/s on
    // the 128-channel stage of the trunk - the 128x128 tile is only for reduction widths that are not multiples of 32
    if (bm == 128 && bn == 128 && d->Cin % 32 == 0) bm = 64;
    if ((bm != 128 || bn != 128) && d->Cin % 32) {
        if (d->Npad % 128) return;
        bm = bn = 128;
    }
    if (const char* f = getenv("RDPN6D_X3_TILE")) {  // profiling: "bm,bn"
        int fbm = 0, fbn = 0;
        if (sscanf(f, "%d,%d", &fbm, &fbn) == 2 && d->Npad % fbn == 0 && (d->Cin % 32 == 0 || (fbm == 128 && fbn == 128))) {
            bm = fbm;
            bn = fbn;
        }
    }
    *pbm = bm;
    *pbn = bn;
}

int conv_x3_launch_tile(ConvX3Args& ax, int bm, int bn, hipStream_t s)
{
    ConvBArgs& a = ax.b;
    const int rb = (bm == 128 && bn == 128) ? 32 : 64;
    a.cchunks = a.d.Cin * 2 / rb;
    a.nk = a.d.ntaps * a.cchunks;
    a.mtiles = rd_cdiv(a.M, bm);
    a.ntiles = a.d.Npad / bn;
    if (bm == 128 && bn == 128) return launch_tile<128, 128, 32>(ax, s);
    if (bm == 128 && bn == 64) return launch_tile<128, 64, 64>(ax, s);
    if (bm == 64 && bn == 128) return launch_tile<64, 128, 64>(ax, s);
    return launch_tile<64, 64, 64>(ax, s);
}
