// 3x3 / stride 1 / pad 1 convolution on the bf16 matrix pipe with the input HALO resident in LDS.
//
// The implicit-GEMM kernel (conv_igemm_bf16.hip) re-fetches the A operand once per tap: nine shifted copies of the same
// input tile travel L2 -> LDS for every 64-channel slab, and at 128-wide tiles that path (64 B/clk/CU) needs its full
// rate at MFMA peak.  Here a workgroup owns TR full image rows (TR x W = 256 output pixels) x 128
// output channels:
//   * per 64-channel slab the (TR+2) x (W+2) input halo is DMA'd ONCE (zero border by the hardware bounds check) into
//     one of two LDS halo buffers, while the previous slab is being consumed (its pieces are spread over the 9 tap steps);
//   * the nine taps read their A fragments from that halo at shifted pixel addresses (ds_read_b128, XOR-swizzled by the
//     halo pixel index); only the 128 x 64 weight slab of the tap (16 KiB) is streamed per step, three LDS stages, DMA
//     three steps ahead of its MFMAs, counted vmcnt, one raw barrier per step (as the NST = 3 form of conv_igemm_bf16);
//   * L2 -> LDS traffic per FLOP drops 3x (5.2 vs 15.6 bytes per kFLOP for the head layers).
// K order (channel slab major, taps inner) and the epilogue are those of conv_igemm_bf16_kernel: results are bit-identical.
// 8 wavefronts as 4 (pixels) x 2 (channels), wave tile 64 x 64, 149 KiB of LDS, one workgroup per CU.
#include "conv_bf16_common.h"

template <int W>  // image width: 64 | 32 | 16  (TR = 256 / W rows per workgroup)
__global__ __launch_bounds__(512, 1) void conv3x3_halo_bf16_kernel(const ConvBArgs a)
{
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, NW = 8, TM = 2, TN = 2, NJ = 4;
    constexpr int TR = BM / W, HW2 = W + 2, HP = (TR + 2) * HW2;  // halo pixels
    constexpr int NP = (HP + 7) / 8;                               // 1-KiB halo pieces (8 pixels x 128 B)
    constexpr int HALO_BYTES = NP * 1024;
    constexpr int BST = BN * 128;                                  // one weight stage: 128 rows x 128 B
    constexpr int NDMA = 3;                                        // per wave per step: 2 weight pieces + 1 halo piece (or a dummy)
    static_assert(NP <= 7 * 8, "halo pieces must fit the first seven tap steps");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* halo = smem;                       // [2][HALO_BYTES]
    unsigned char* Bs = smem + 2 * HALO_BYTES;        // [3][BST]
    unsigned char* dummy = Bs + 3 * BST;              // 1 KiB sink for the padding DMA

    const rdpn6d_conv_desc& d = a.d;
    const int nblk = a.mtiles * a.ntiles;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
    const int nt = logical % a.ntiles;
    const int mt = logical / a.ntiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;
    // the tile = rows [y0, y0 + TR) of image b
    const int tiles_per_img = d.H / TR;
    const int img = mt / tiles_per_img;
    const int y0 = (mt - img * tiles_per_img) * TR;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);
    const unsigned oob = a.x_bytes;
    const unsigned px_bytes = (unsigned)d.in_cs * 2u;
    const unsigned img_base = ((unsigned)(img * d.H * d.W) * (unsigned)d.in_cs + (unsigned)d.in_co) * 2u;

    // ---- DMA geometry.  One piece = 8 rows x 128 B; lane L lands at (row L/8, physical 16-byte slot L%8) and fetches the
    // logical slot (L%8) ^ swz(row index), swz(i) = (i >> 1) & 7 (see conv_igemm_bf16.hip).
    const int prow = lane >> 3, pslot = lane & 7;
    auto halo_piece = [&](const int piece, const int cc, const int buf) {
        // piece >= NP (or no next slab): a dummy load keeps the per-step DMA count uniform for the counted vmcnt
        const int hp = piece * 8 + prow;
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = piece < NP && hp < HP && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
        const unsigned off = ok ? img_base + (unsigned)(iy * d.W + ix) * px_bytes + (unsigned)cc * 128u + (unsigned)((pslot ^ ((hp >> 1) & 7)) * 16)
                                : oob;
        unsigned char* dst = piece < NP ? halo + buf * HALO_BYTES + piece * 1024 : dummy;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };
    unsigned w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + NW * i) * 8 + prow;
        w_off[i] = (unsigned)(n0 + row) * (unsigned)a.Ktot * 2u + (unsigned)((pslot ^ ((row >> 1) & 7)) * 16);
    }
    auto weight_stage = [&](const int tap, const int cc, const int st) {
        const unsigned wk = ((unsigned)tap * (unsigned)d.Cin + (unsigned)cc * 64u) * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr_t)(Bs + st * BST + (wave + NW * i) * 1024), 16, (int)(w_off[i] + wk), 0, 0, 0);
    };

    // ---- fragment geometry
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, half = lane >> 5;
    int hbase[TM];  // halo pixel index of this lane's output pixel (tile i), centre tap
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = wm * 64 + i * 32 + frow;
        const int ty = p / W, tx = p - ty * W;
        hbase[i] = (ty + 1) * HW2 + tx + 1;
    }
    auto read_frags = [&](const int tap, const int buf, const int st, u32x4 (&fa)[TM][NJ], u32x4 (&fb)[TN][NJ]) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int shift = dy * HW2 + dx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int hp = hbase[i] + shift;
            const int sw = (hp >> 1) & 7;
            const unsigned char* p = halo + buf * HALO_BYTES + hp * 128;
#pragma unroll
            for (int j = 0; j < NJ; ++j) fa[i][j] = *reinterpret_cast<const u32x4*>(p + (((2 * j + half) ^ sw) << 4));
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int R = wn * 64 + jn * 32 + frow;
            const int sw = (R >> 1) & 7;
            const unsigned char* p = Bs + st * BST + R * 128;
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[jn][j] = *reinterpret_cast<const u32x4*>(p + (((2 * j + half) ^ sw) << 4));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mma = [&](const u32x4 (&fa)[TM][NJ], const u32x4 (&fb)[TN][NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][j]),
                                                                         __builtin_bit_cast(bf16x8, fb[jn][j]), acc[i][jn], 0, 0, 0);
    };

    // ---- step bookkeeping: step s = (slab cc, tap); "ld" = the weight slab being fetched (three steps ahead)
    const int cch = a.cchunks;
    const int nsteps = 9 * cch;
    int ld_tap = 0, ld_cc = 0, ld_left = nsteps - 1;
    auto ld_next = [&]() {
        const int go = ld_left > 0 ? 1 : 0;
        ld_left -= go;
        ld_tap += go;
        const int wrap = ld_tap == 9 ? 1 : 0;
        ld_tap = wrap ? 0 : ld_tap;
        ld_cc += wrap;
    };

    // ---- prologue: halo of slab 0 (7 pieces per wave), weights of steps 0..2
#pragma unroll
    for (int t = 0; t < 7; ++t) halo_piece(t * NW + wave, 0, 0);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        weight_stage(ld_tap, ld_cc, s);
        ld_next();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 fa0[TM][NJ], fb0[TN][NJ], fa1[TM][NJ], fb1[TN][NJ];
    read_frags(0, 0, 0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // weight stage 0 is re-filled by the first step
    asm volatile("" ::: "memory");

    // Invariant at the top of step s = (cc, tap): fragments of step s in registers; weights of s+1 landed and published
    // in stage (s+1)%3, of s+2 in flight into stage (s+2)%3, stage s%3 free; halo of slab cc complete in buffer cc&1;
    // pieces of slab cc+1 issued during taps 0..6 of slab cc land (counted vmcnt) before tap 8 reads the next slab.
    int tap = 0, cc = 0, st_free = 0, st_next = 1;
    auto advance = [&]() {
        ++tap;
        if (tap == 9) { tap = 0; ++cc; }
        st_free = st_next;
        st_next = st_next == 2 ? 0 : st_next + 1;
    };
    auto step_issue = [&]() {
        weight_stage(ld_tap, ld_cc, st_free);
        ld_next();
        halo_piece((tap < 7 && cc + 1 < cch) ? tap * NW + wave : NP, cc + 1, (cc + 1) & 1);
    };
    // (A ping-pong schedule - waves w and w + 4 of a SIMD half a step apart, two barriers per step - and a forced
    //  MFMA / ds_read / DMA issue interleave were both measured slower: 427 and 381 us vs 369 us on the head layer.)
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int npairs = nsteps >> 1;  // nsteps = 9 * cch; odd when cch is odd
    for (int pr = 0; pr < npairs; ++pr) {
        {
            step_issue();
            const int ntap = tap == 8 ? 0 : tap + 1, ncc = tap == 8 ? cc + 1 : cc;
            read_frags(ntap, ncc & 1, st_next, fa1, fb1);
            mma(fa0, fb0);
            publish();
            advance();
        }
        {
            step_issue();
            const int ntap = tap == 8 ? 0 : tap + 1, ncc = tap == 8 ? cc + 1 : cc;
            read_frags(ntap, ncc & 1, st_next, fa0, fb0);
            mma(fa1, fb1);
            publish();
            advance();
        }
    }
    if (nsteps & 1) mma(fa0, fb0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    conv_bf16_epilogue<BM, BN, WM, WN, TM, TN>(a, acc, smem, m0, n0, wave, lane, wm, wn);
}

static size_t halo_lds_bytes(int W)
{
    const int TR = 256 / W, HP = (TR + 2) * (W + 2), NP = (HP + 7) / 8;
    const size_t stage = 2 * (size_t)NP * 1024 + 3 * 128 * 128 + 1024;
    const size_t epi = 8 * 32 * (64 + 8) * 4;
    return stage > epi ? stage : epi;
}

// eligibility: 3x3 / stride 1 / pad 1 in the canonical tap order, linear output, W in {64, 32, 16}, whole tiles
bool conv3x3_halo_bf16_eligible(const rdpn6d_conv_desc* d)
{
    if (d->ntaps != 9 || d->stride != 1 || d->Ho != d->H || d->Wo != d->W) return false;
    for (int t = 0; t < 9; ++t)
        if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1) return false;
    if (!(d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo)) return false;
    if (d->W != 64 && d->W != 32 && d->W != 16) return false;
    if (d->H % (256 / d->W) != 0) return false;
    if (d->Cin % 64 != 0 || d->N % 128 != 0 || d->Npad != d->N) return false;
    return true;
}

int conv3x3_halo_bf16_launch(ConvBArgs& a, hipStream_t s)
{
    const rdpn6d_conv_desc* d = &a.d;
    a.mtiles = (int)(a.M / 256);
    a.ntiles = d->N / 128;
    const size_t lds = halo_lds_bytes(d->W);
    static bool configured[3] = {false, false, false};
    const int wi = d->W == 64 ? 0 : (d->W == 32 ? 1 : 2);
    const void* kern = wi == 0 ? reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<64>)
                               : (wi == 1 ? reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<32>)
                                          : reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<16>));
    if (!configured[wi]) {
        RD_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[wi] = true;
    }
    dim3 grid((unsigned)(a.mtiles * a.ntiles)), block(512);
    if (wi == 0) hipLaunchKernelGGL(conv3x3_halo_bf16_kernel<64>, grid, block, lds, s, a);
    else if (wi == 1) hipLaunchKernelGGL(conv3x3_halo_bf16_kernel<32>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(conv3x3_halo_bf16_kernel<16>, grid, block, lds, s, a);
    return RDPN6D_OK;
}
