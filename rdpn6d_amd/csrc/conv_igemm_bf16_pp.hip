// 16-bit (bf16 | fp16, csrc/common.h) form of the eight-wave ping-pong kernel of conv_igemm_h2_pp.hip for the trunk layers of the
// reduced-precision mode and of the mixed-precision training step (N % 128 == 0, >= 224 tiles, Cin % 64 == 0): the two-stage 2x2-wave
// tile kernels of conv_igemm_bf16.hip run these layers at 22-30 % of the 16-bit MFMA peak, bound by the L2 -> LDS round trip of their
// drained two-stage ring.  Same structure as the h2 kernel - two groups of four wavefronts one barrier apart, an NST-stage ring fed D
// chunks ahead, counted vmcnt waits, the next step's DMA addresses computed under the MFMAs - on 128-byte rows of 64 channels: four
// k16 steps with ONE product each per chunk (the h2 kernel: two steps x three products), and the shared coalesced epilogue
// (conv_bf16_common.h: residual, activation, 16-bit or fp32 store, the BatchNorm partial sums of the training forward).
#include "conv_bf16_common.h"

#include <cstdlib>
#include <type_traits>

template <int V>
using ic = std::integral_constant<int, V>;

namespace {

template <int BM, int BN, int WM, int WN, int NST, int PM, int STATS = 0>
__global__ __launch_bounds__(512) void conv_lp_pp_kernel(const ConvBArgs a)
{
    constexpr int RB = 128, RPP = 8, NW = 8;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int AG = BM / RPP / NW, BG = BN / RPP / NW;  // LDS-DMA pieces per wave and chunk
    constexpr int P = AG + BG, PL = P - PM, D = NST - 1;
    static_assert(WM * WN == NW && WM == 2 && TM >= 1 && TN >= 1 && AG >= 1 && BG >= 1, "wave grid / tile");
    static_assert(PM >= 0 && PM <= 2 && PL >= 1 && D >= 2, "pieces in the MFMA part; at least three stages");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + NST * BM * RB;

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
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = grp, wn = wq;

    // ---- DMA addressing: this wave moves pieces wave, wave + 8, ... (8 rows x 128 B each) of the A rows and of the B rows of a chunk
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned px_bytes = (unsigned)d.in_cs * 2u;
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
        const int lslot = pslot ^ ((row >> 1) & 7);
        a_base[i] = (unsigned)((b * d.H + iy) * d.W + ix) * px_bytes + (unsigned)d.in_co * 2u + (unsigned)lslot * 16u;
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
        const int lslot = pslot ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)(n0 + row) * (unsigned)a.Ktot * 2u + (unsigned)lslot * 16u;
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);

    unsigned dma_off[P];
    auto stage_addr = [&](const int tap, const int cc, const bool valid) {
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned toff = (unsigned)((dy * d.W + dx) * (int)px_bytes + cc * RB);  // wave-uniform
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;                                 // past the last chunk: out of range, zeros land
#pragma unroll
        for (int i = 0; i < AG; ++i) dma_off[i] = (a_base[i] + toff) | (((a_mask[i] >> tap) & 1u) - 1u) | sel;
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 2u + (unsigned)cc * (unsigned)RB;
#pragma unroll
        for (int i = 0; i < BG; ++i) dma_off[AG + i] = (w_off[i] + wk) | sel;
    };
    auto stage_piece = [&](auto ic_, const int st) {
        constexpr int i = decltype(ic_)::value;
        if constexpr (i < AG) {
            unsigned char* dst = As + ((st * BM) + (wave + NW * i) * RPP) * RB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)dst, 16, (int)dma_off[i], 0, 0, 0);
        } else {
            unsigned char* dst = Bs + ((st * BN) + (wave + NW * (i - AG)) * RPP) * RB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr_t)dst, 16, (int)dma_off[i], 0, 0, 0);
        }
    };
    auto stage_pieces = [&](auto firstc, auto lastc, const int st) {  // pieces [first, last)
        constexpr int first = decltype(firstc)::value, last = decltype(lastc)::value;
        if constexpr (first + 0 < last) stage_piece(ic<first + 0>{}, st);
        if constexpr (first + 1 < last) stage_piece(ic<first + 1>{}, st);
        if constexpr (first + 2 < last) stage_piece(ic<first + 2>{}, st);
        if constexpr (first + 3 < last) stage_piece(ic<first + 3>{}, st);
        if constexpr (first + 4 < last) stage_piece(ic<first + 4>{}, st);
        if constexpr (first + 5 < last) stage_piece(ic<first + 5>{}, st);
        if constexpr (first + 6 < last) stage_piece(ic<first + 6>{}, st);
        if constexpr (first + 7 < last) stage_piece(ic<first + 7>{}, st);
        static_assert(last - first <= 8, "pieces per wave and chunk");
    };

    const int frow = lane & 31;

    const int nk = a.nk;                       // chunk order: channel-chunk major, taps innermost
    int ld_cc = 0, ld_tap = 0, ld_idx = 0;  // the chunk the DMA stream is at
    auto next_chunk = [&]() {
        ++ld_idx;
        ++ld_tap;
        const int wrap = ld_tap == d.ntaps ? 1 : 0;
        ld_tap = wrap ? 0 : ld_tap;
        ld_cc += wrap;
    };

    // ---- fragment addressing
    const int half = lane >> 5;
    u32x4 fa[TM][4], fb[TN][4];
    auto read_frags = [&](const int st) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = wm * WTM + i * 32 + frow;
            const int sw = (R >> 1) & 7;
            const unsigned char* q = As + ((st * BM) + R) * RB;
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[i][j] = *reinterpret_cast<const u32x4*>(q + (((2 * j + half) ^ sw) << 4));
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int R = wn * WTN + jn * 32 + frow;
            const int sw = (R >> 1) & 7;
            const unsigned char* q = Bs + ((st * BN) + R) * RB;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[jn][j] = *reinterpret_cast<const u32x4*>(q + (((2 * j + half) ^ sw) << 4));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma_group = [&](auto jc) {  // k16 step j of the chunk: one product per accumulator tile
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = RD_LP_MFMA_32x32x16(fa[i][j], fb[jn][j], acc[i][jn]);
    };

    // ---- prologue: chunks 0 .. D-1 into stages 0 .. D-1
#pragma unroll
    for (int c = 0; c < D; ++c) {
        stage_addr(ld_tap, ld_cc, ld_idx < nk);
        stage_pieces(ic<0>{}, ic<P>{}, c);
        next_chunk();
    }
    stage_addr(ld_tap, ld_cc, ld_idx < nk);                              // addresses of chunk D, issued in step 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * P) : "memory");  // chunk 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                        // ... and everybody's
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on

    int st_rd = 0, st_wr = D;  // stage of chunk k, stage of chunk k + D
    for (int k = 0; k < nk; ++k) {
        // ---- L(k): fragments of chunk k; this wave's first PL pieces of chunk k + D
        __builtin_amdgcn_sched_barrier(0);
        read_frags(st_rd);
        __builtin_amdgcn_sched_barrier(0);
        stage_pieces(ic<0>{}, ic<PL>{}, st_wr);  // (addresses of chunk k + D: computed between the MFMAs of the previous step)
        __builtin_amdgcn_sched_barrier(0);
        // outstanding and newer than chunk k + 1: chunks k + 2 .. k + D - 1 (P each) + the PL pieces just issued -> chunk k + 1 has
        // landed; the fragment reads are back (nobody may still be reading a stage the other group is about to re-fill)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 2) * P + PL) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- M(k): the MFMAs of chunk k, the remaining PM pieces between them
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma_group(ic<0>{});
        mma_group(ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PM >= 1) stage_piece(ic<PL>{}, st_wr);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(ic<2>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PM >= 2) stage_piece(ic<PL + 1>{}, st_wr);
        __builtin_amdgcn_sched_barrier(0);
        // the DMA addresses of the NEXT step's chunk (k + 1 + D) under the last MFMAs
        next_chunk();
        stage_addr(ld_tap, ld_cc, ld_idx < nk);
        mma_group(ic<3>{});
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        st_rd = st_rd == NST - 1 ? 0 : st_rd + 1;
        st_wr = st_wr == NST - 1 ? 0 : st_wr + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * P) : "memory");  // newer than chunk k + 1: chunks k + 2 .. k + D
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the out-of-range pieces of the last D steps: nothing may still be landing in LDS)
    if (grp == 0) __builtin_amdgcn_s_barrier();       // re-align the two groups

    // ---- epilogue: the shared coalesced one (scale / shift, residual, activation, 16-bit or fp32 store, BatchNorm partial sums);
    // every wave is past its last fragment read and no DMA is in flight - its __syncthreads() re-purposes the staging LDS
    conv_bf16_epilogue_vec<BM, BN, WM, WN, TM, TN, true, STATS>(a, acc, smem, m0, n0, wave, lane, wm, wn);
}

template <int BM, int BN, int WM, int WN, int NST, int PM>
static int launch_lp_pp(const ConvBArgs& a, hipStream_t s)
{
    constexpr int lds_stage = NST * (BM + BN) * 128;
    constexpr int lds_epi = 8 * 32 * (BN / WN + 8) * 4;
    constexpr int lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS");
    if (a.stats && a.bnb_y) {  // ... of a residual block's last BatchNorm (mask from the stored block output)
        auto kern = conv_lp_pp_kernel<BM, BN, WM, WN, NST, PM, 2>;
        RD_LDS_OPT_IN(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), lds, s, a);
        return RDPN6D_OK;
    }
    if (a.stats) {  // training: the instantiation whose epilogue writes BatchNorm partial sums
        auto kern = conv_lp_pp_kernel<BM, BN, WM, WN, NST, PM, 1>;
        RD_LDS_OPT_IN(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), lds, s, a);
        return RDPN6D_OK;
    }
    auto kern = conv_lp_pp_kernel<BM, BN, WM, WN, NST, PM>;
    RD_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), lds, s, a);
    return RDPN6D_OK;
}

}  // namespace

// The tile shape the ping-pong kernel would take this problem with (0 = 128x128, four stages; 2 = 256x128, three stages), or -1: it needs
// 128-byte K rows (Cin % 64 == 0), full 128-wide column tiles, the coalesced epilogue, >= 8 chunks and >= 224 tiles (one per CU).
// RDPN6D_LP_PP = 0 switches it off (profiling).
int conv_lp_pp_plan(const ConvBArgs& a, int rb, int* pbm, int* pbn)
{
    static const int on = getenv("RDPN6D_LP_PP") ? atoi(getenv("RDPN6D_LP_PP")) : 1;
    static const int force = getenv("RDPN6D_LP_PP_SHAPE") ? atoi(getenv("RDPN6D_LP_PP_SHAPE")) : -1;  // profiling: 0 | 2
    const rdpn6d_conv_desc& d = a.d;
    if (!on || rb != 128 || d.Npad % 128 != 0 || d.N != d.Npad || !a.vec_out || a.nk < 8) return -1;
    const int nt = d.Npad / 128;
    int shape = 0;
    if ((long long)rd_cdiv(a.M, 256) * nt >= 224 && (long long)rd_cdiv(a.M, 128) * nt > 288) shape = 2;
    if (force == 0 || force == 2) shape = force;
    const int bm = shape == 0 ? 128 : 256;
    if ((long long)rd_cdiv(a.M, bm) * nt < 224) return -1;
    *pbm = bm;
    *pbn = 128;
    return shape;
}

int conv_lp_launch_pp(const ConvBArgs& a, int shape, hipStream_t s)
{
    if (shape == 0) return launch_lp_pp<128, 128, 2, 4, 4, 2>(a, s);
    if (shape == 2) return launch_lp_pp<256, 128, 2, 4, 3, 2>(a, s);
    rdpn6d_set_error("conv_lp_launch_pp: unknown tile shape %d", shape);
    return RDPN6D_EINVAL;
}
