// Implicit-GEMM convolution on the gfx950 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = ntaps*Cin.
//   A[m][k] is gathered on the fly from the NHWC activation (one contiguous 64-byte run of
//   16 channels per (pixel, tap)), B[n][k] is the packed weight [Npad][ntaps][Cin].
//   Workgroup = 256 threads = 4 wavefronts (2x2); each wavefront owns a (BM/2)x(BN/2) sub-tile
//   made of 32x32 MFMA accumulators.  K advances in chunks of 16 through a 3-stage pipeline: while
//   the 8 MFMA k-steps of chunk k issue from one fragment register set, the fragments of chunk k+1
//   are read from LDS into the other set and chunk k+2 is DMA'd HBM/L2 -> LDS directly
//   (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write; one barrier per chunk, no LDS
//   latency exposed after it).  LDS rows are 64 B, XOR-swizzled on the DMA source address and on
//   the ds_read_b128 fragment loads, which are bank-conflict free.
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
    int tap_inner;
    unsigned x_bytes;  // size of the input tensor in bytes (buffer descriptor bound)
    unsigned w_bytes;  // size of the packed weight tensor in bytes
    unsigned long long dy_pack, dx_pack;  // tap offsets, 4 bits each, biased by +8
    int kper;        // split-K: chunks per split (blockIdx.y = split); nk when not split
    float* partial;  // split-K: raw accumulators [split][M][Npad]; nullptr = fused epilogue
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BM, int BN>
__global__ __launch_bounds__(256, 3) void conv_igemm_f32_kernel(const ConvKArgs a)
{
    constexpr int ROWF = 16;          // floats per LDS row (one 64-byte k-chunk, no padding: LDS-DMA writes lane-linear)
    constexpr int NST = 3;            // LDS stages (chunk k in MFMA, k+1 in fragment registers, k+2 landing by DMA)
    constexpr int TM = BM / 64;       // 32x32 tiles per wave along M
    constexpr int TN = BN / 64;       // ... along N
    constexpr int AG = BM / 64;       // 16-row groups (1 KiB DMA pieces) per wave for the A tile
    constexpr int BG = BN / 64;       // ... for the B tile
    __shared__ __attribute__((aligned(1024))) float smem[NST * (BM + BN) * ROWF];
    float* As = smem;
    float* Bs = smem + NST * BM * ROWF;

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
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- staging geometry.  One buffer_load_dwordx4 ... lds per wave moves 16 rows x 64 B straight into LDS:
    // lane L lands at byte L*16 of the piece = (row L/4, physical 16-byte slot L%4).  The slot is XOR-swizzled with
    // bits 2..3 of the row so that the ds_read_b128 fragment loads (16 lanes = 16 different rows, same logical
    // slot) hit 16 distinct 16-byte bank groups; the swizzle is applied to the SOURCE address here and to the read.
    const int prow = lane >> 2;                       // row inside the 16-row piece
    const int pslot = lane & 3;                       // physical slot written by this lane
    int a_pix[AG], a_iy[AG], a_ix[AG], a_ch[AG];
    bool a_ok[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int row = (wave + 4 * i) * 16 + prow;
        const long long m = m0 + row;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? (int)m : 0;
        const int b = mm / a.HoWo;
        const int rem = mm - b * a.HoWo;
        const int oy = rem / d.Wo;
        const int ox = rem - oy * d.Wo;
        a_pix[i] = b * d.H * d.W;
        a_iy[i] = oy * d.stride;
        a_ix[i] = ox * d.stride;
        a_ch[i] = (pslot ^ ((row >> 2) & 3)) * 4;    // logical channel offset (floats) inside the 16-channel chunk
    }
    unsigned w_off[BG];  // byte offset of (weight row, logical slot) inside the packed weight tensor
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave + 4 * i) * 16 + prow;
        w_off[i] = ((unsigned)(n0 + row) * (unsigned)a.Ktot + (unsigned)((pslot ^ ((row >> 2) & 3)) * 4)) * 4u;
    }

    // buffer descriptors: bounds-checked in hardware, so a tap that falls outside the image (or a row past M) gets an
    // out-of-range offset and DMA-writes zeros - no branch, no select, nothing to wait for inside a chunk.
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);
    const unsigned oob = a.x_bytes;  // any offset >= num_records returns 0
    unsigned a_off[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) a_off[i] = ((unsigned)a_pix[i] * (unsigned)d.in_cs + (unsigned)(d.in_co + a_ch[i])) * 4u;
    const unsigned px_bytes = (unsigned)d.in_cs * 4u;

    // K order: tap_inner = 1 walks channel-chunk-major with the taps innermost (the 9 shifted reads of one
    // 16-channel slab of the block's input footprint follow each other: L1/L2 locality); 0 = tap-major.
    auto stage_chunk = [&](const int tap, const int cc, const int st) {
        // tap offsets are packed 4 bits each (+8 bias) in two kernel arguments: pure ALU, no memory access
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned c0b = (unsigned)cc * 64u;
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            const unsigned off = ok ? a_off[i] + (unsigned)(iy * d.W + ix) * px_bytes + c0b : oob;
            float* dst = As + (st * BM + (wave + 4 * i) * 16) * ROWF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
        }
        const unsigned wk = ((unsigned)tap * (unsigned)d.Cin) * 4u + c0b;
#pragma unroll
        for (int i = 0; i < BG; ++i) {
            float* dst = Bs + (st * BN + (wave + 4 * i) * 16) * ROWF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr_t)dst, 16, (int)(w_off[i] + wk), 0, 0, 0);
        }
    };
    // chunk counter -> (tap, cc), advanced branch-free; clamped at the last chunk so that the loop body can
    // stage unconditionally (the two redundant DMAs past the end re-read the last chunk into a stage nobody reads)
    const int k_begin = (int)blockIdx.y * a.kper;
    const int nk = min(a.kper, a.nk - k_begin);
    int ld_tap, ld_cc, ld_left = nk - 1;
    if (a.tap_inner) {
        ld_cc = k_begin / d.ntaps;
        ld_tap = k_begin - ld_cc * d.ntaps;
    } else {
        ld_tap = k_begin / a.cchunks;
        ld_cc = k_begin - ld_tap * a.cchunks;
    }
    auto next_chunk = [](int& tap, int& cc, int& left, const int ntaps, const int cchunks, const int tap_inner) {
        const int go = left > 0 ? 1 : 0;
        left -= go;
        if (tap_inner) {
            tap += go;
            const int wrap = tap == ntaps ? 1 : 0;
            tap = wrap ? 0 : tap;
            cc += wrap;
        } else {
            cc += go;
            const int wrap = cc == cchunks ? 1 : 0;
            cc = wrap ? 0 : cc;
            tap += wrap;
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31;          // fragment row (A: pixel, B: channel) inside a 32-tile
    const int half = lane >> 5;          // this lane's 8 k-values are logical slots 2*half, 2*half+1

    auto read_frags = [&](int st, f32x4 (&fa)[TM][2], f32x4 (&fb)[TN][2]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = wm * (BM / 2) + i * 32 + frow;
            const int sw = (R >> 2) & 3;
            const float* p = &As[(st * BM + R) * ROWF];
            fa[i][0] = *reinterpret_cast<const f32x4*>(p + (((2 * half) ^ sw) << 2));
            fa[i][1] = *reinterpret_cast<const f32x4*>(p + (((2 * half + 1) ^ sw) << 2));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int R = wn * (BN / 2) + j * 32 + frow;
            const int sw = (R >> 2) & 3;
            const float* p = &Bs[(st * BN + R) * ROWF];
            fb[j][0] = *reinterpret_cast<const f32x4*>(p + (((2 * half) ^ sw) << 2));
            fb[j][1] = *reinterpret_cast<const f32x4*>(p + (((2 * half + 1) ^ sw) << 2));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma = [&](const f32x4 (&fa)[TM][2], const f32x4 (&fb)[TN][2]) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s >> 2][s & 3], fb[j][s >> 2][s & 3],
                                                                     acc[i][j], 0, 0, 0);
    };

    // ---- prologue: chunks 0 and 1 into LDS stages 0 and 1, fragments of chunk 0 into registers
    stage_chunk(ld_tap, ld_cc, 0);
    next_chunk(ld_tap, ld_cc, ld_left, d.ntaps, a.cchunks, a.tap_inner);
    stage_chunk(ld_tap, ld_cc, 1);
    next_chunk(ld_tap, ld_cc, ld_left, d.ntaps, a.cchunks, a.tap_inner);
    // The DMA of the chunk staged in a step stays in flight across that step's barrier: a step waits with a counted
    // s_waitcnt vmcnt(<DMAs of one chunk>) - the chunk staged one step earlier has landed for this wave - and a raw s_barrier
    // publishes it (a __syncthreads would drain the queue and expose the L2 -> LDS round trip of the newest chunk every step).
    constexpr int NDMA = AG + BG;
    auto publish = [&]() {
        // lgkmcnt(0): this wave's fragment reads of the previous step have returned, so the stage they came from may be
        // re-filled by whoever passes the barrier
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    publish();  // chunk 0 has landed (chunk 1 may still be in flight)
    f32x4 fa0[TM][2], fb0[TN][2], fa1[TM][2], fb1[TN][2];
    read_frags(0, fa0, fb0);

    // ---- main loop over chunk PAIRS (fragment double buffer statically indexed, body branch-free).
    // step for chunk kc: (1) start the DMA of chunk kc+2 into its LDS stage (its last readers finished two steps ago),
    // (2) counted wait + barrier: chunk kc+1 is in LDS for every wave, (3) LDS->register fragments of chunk kc+1,
    // (4) 8 MFMA k-steps of chunk kc.
    int st_next = 1, st_stage = 2;  // LDS stage holding chunk kc+1 / receiving chunk kc+2
    const int npairs = nk >> 1;
    for (int pr = 0; pr < npairs; ++pr) {
        stage_chunk(ld_tap, ld_cc, st_stage);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps, a.cchunks, a.tap_inner);
        publish();
        read_frags(st_next, fa1, fb1);
        mma(fa0, fb0);
        st_next = st_next == NST - 1 ? 0 : st_next + 1;
        st_stage = st_stage == NST - 1 ? 0 : st_stage + 1;

        stage_chunk(ld_tap, ld_cc, st_stage);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps, a.cchunks, a.tap_inner);
        publish();
        read_frags(st_next, fa0, fb0);
        mma(fa1, fb1);
        st_next = st_next == NST - 1 ? 0 : st_next + 1;
        st_stage = st_stage == NST - 1 ? 0 : st_stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail DMAs must not land after the workgroup has retired
    if (nk & 1) mma(fa0, fb0);  // odd chunk count: the last chunk's fragments are already in registers

    const int hi = lane >> 5;
    // ---- split-K: raw partial sums, reduced (with the epilogue) by splitk_epilogue_kernel
    if (a.partial) {
        float* part = a.partial + (long long)blockIdx.y * a.M * d.Npad;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + frow;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long long m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (m < a.M) part[m * d.Npad + n] = acc[i][j][e];
                }
        }
        return;
    }
    // ---- fused epilogue
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

// split-K second pass: sum the partial accumulators over the splits (fixed order: deterministic) and apply the epilogue
__global__ void splitk_epilogue_kernel(const float* __restrict__ partial, int nsplit, long long M, rdpn6d_conv_desc d)
{
    const long long total = M * d.N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / d.N;
        const int n = (int)(i - m * d.N);
        // the slices are added in slice order (same bits as one load per iteration), eight loads in flight: fc1's 47 slices were 47
        // dependent round trips for a 64 x 1024 output (15 us)
        const float* p = partial + m * d.Npad + n;
        const long long st = M * d.Npad;
        float v = 0.f;
        int s = 0;
        for (; s + 8 <= nsplit; s += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(long long)(s + u) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; s < nsplit; ++s) v += p[(long long)s * st];
        v = v * (d.scale ? d.scale[n] : 1.f) + (d.shift ? d.shift[n] : 0.f);
        if (d.res) v += d.res[m * d.res_cs + d.res_co + n];
        if (d.act == 1) v = v > 0.f ? v : 0.f;
        else if (d.act == 2) v = v > 0.f ? v : v * d.slope;
        d.y[m * d.out_cs + d.out_co + n] = v;
    }
}

static int g_force_bm = 0, g_force_bn = 0, g_tap_inner = 1;
extern "C" void rdpn6d_conv_set_tap_inner(int v) { g_tap_inner = v; }
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

static int conv2d_f32_impl(const rdpn6d_conv_desc* d, int ksplit, float* workspace, void* stream);

extern "C" int rdpn6d_conv2d_f32(const rdpn6d_conv_desc* d, void* stream) { return conv2d_f32_impl(d, 1, nullptr, stream); }

// Split-K form for skinny problems (the FC layers: M = batch rows, K up to 8192): the K range is cut into ksplit
// slices computed by separate workgroups (grid.y), partial sums land in `workspace` and a second kernel reduces
// them in a fixed order and applies the epilogue.  Output geometry must be linear (no phase offsets).
extern "C" long long rdpn6d_conv_splitk_ws_floats(const rdpn6d_conv_desc* d, int ksplit)
{
    if (!d || ksplit < 1) return 0;
    return (long long)ksplit * d->B * d->Ho * d->Wo * d->Npad;
}

extern "C" int rdpn6d_conv2d_splitk_f32(const rdpn6d_conv_desc* d, int ksplit, float* workspace, void* stream)
{
    RD_REQUIRE(ksplit >= 1 && (ksplit == 1 || workspace), "split-K needs a workspace");
    return conv2d_f32_impl(d, ksplit, workspace, stream);
}

static int conv2d_f32_impl(const rdpn6d_conv_desc* d, int ksplit, float* workspace, void* stream)
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
    const long long xb = (long long)d->B * d->H * d->W * d->in_cs * 4;
    RD_REQUIRE(xb < (1LL << 32) - 64, "input tensor must be smaller than 4 GiB (32-bit buffer offsets)");
    a.x_bytes = (unsigned)xb;
    a.tap_inner = g_tap_inner;
    const long long wb = (long long)d->Npad * d->ntaps * d->Cin * 4;
    RD_REQUIRE(wb < (1LL << 32) - 64, "packed weights must be smaller than 4 GiB");
    a.w_bytes = (unsigned)wb;
    a.dy_pack = a.dx_pack = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        RD_REQUIRE(d->dy[t] >= -8 && d->dy[t] <= 7 && d->dx[t] >= -8 && d->dx[t] <= 7, "tap offsets must be in -8..7");
        a.dy_pack |= (unsigned long long)(d->dy[t] + 8) << (4 * t);
        a.dx_pack |= (unsigned long long)(d->dx[t] + 8) << (4 * t);
    }
    int bm, bn;
    conv_pick_tile(d, a.M, &bm, &bn);
    a.mtiles = rd_cdiv(a.M, bm);
    a.ntiles = d->Npad / bn;
    a.kper = a.nk;
    a.partial = nullptr;
    int nsplit = 1;
    if (ksplit > 1) {
        RD_REQUIRE(a.linear_out, "split-K needs a linear output geometry");
        a.kper = rd_cdiv(a.nk, ksplit);
        nsplit = rd_cdiv(a.nk, a.kper);
        if (nsplit > 1) a.partial = workspace;
        else a.kper = a.nk;
    }
    dim3 grid((unsigned)(a.mtiles * a.ntiles), (unsigned)nsplit), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_igemm_f32_kernel<128, 128>), grid, block, 0, s, a);
    else if (bm == 128 && bn == 64) hipLaunchKernelGGL((conv_igemm_f32_kernel<128, 64>), grid, block, 0, s, a);
    else if (bm == 64 && bn == 128) hipLaunchKernelGGL((conv_igemm_f32_kernel<64, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_f32_kernel<64, 64>), grid, block, 0, s, a);
    RD_LAUNCH_CHECK();
    if (nsplit > 1) {
        const long long total = a.M * d->N;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, workspace, nsplit, a.M, *d);
        RD_LAUNCH_CHECK();
    }
    return RDPN6D_OK;
}
