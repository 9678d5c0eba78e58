// Implicit-GEMM convolution on the gfx950 bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), fp32 accumulation.
//
// Reduced-precision mode of the same operator as conv_igemm.hip (the reference's counterpart is its autocast
// path: core/gdrn_modeling/engine.py:279 `autocast(enabled=AMP_ON)` / gdrn_evaluator.py:625 `AMP_TEST`).
//   * activations and packed weights are bf16 in HBM (NHWC, [Npad][ntaps][Cin]); accumulators, the folded
//     BatchNorm scale/shift and the residual add are fp32; the result is rounded once (RNE) on the store, or
//     written as fp32 for the layers whose consumers are fp32 kernels (head output, FC layers).
//   * GEMM view, tiling (2x2 wavefronts of 32x32 accumulators), LDS-DMA staging with hardware bounds checking,
//     the XOR swizzle and the XCD-aware tile mapping follow conv_igemm.hip.  A K-chunk is RB bytes per row:
//     RB = 128 (64 channels, four k16 MFMA steps) when Cin % 64 == 0, else RB = 64 (32 channels).
//   * lane l feeds MFMA step j with the 16-byte slot 2j + (l >= 32): any bijection of the chunk's k-values onto
//     (step, half, element) is valid as long as A and B use the same one.
//   * NST = 2 LDS stages (4-wave tiles up to 128x128, two workgroups per CU): chunk k is in fragment registers (MFMA),
//     chunk k+1 is being read LDS -> registers from one stage while chunk k+2 lands by DMA in the other (whose
//     fragments were read one step earlier); one __syncthreads per chunk (which drains the DMA queue).
//   * NST = 3 stages (the 256x128 tile, 8 wavefronts as 4x2, 144 KiB of LDS, one workgroup per CU): the DMA runs TWO
//     chunks ahead and is never drained inside the loop - each step ends with s_waitcnt vmcnt(<DMAs of one chunk>)
//     + a raw s_barrier, so the loads of chunk k+3 stay in flight across the barrier that publishes chunk k+2.  The
//     L2 -> LDS round trip (~1.5 us under load) is what bounds the 2-stage form; the wider tile also moves 25 % fewer
//     bytes per FLOP.
#include "conv_bf16_common.h"

#include <cstdlib>

template <int BM, int BN, int RB, int WM, int WN, int NST, int STATS = 0>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void conv_igemm_bf16_kernel(const ConvBArgs a)
{
    constexpr int NW = WM * WN;      // wavefronts per workgroup
    constexpr int SL = RB / 16;      // 16-byte slots per LDS row
    constexpr int RPP = 1024 / RB;   // rows per 1-KiB DMA piece
    constexpr int RPB = 256 / RB;    // rows per 256-byte bank period
    constexpr int NJ = RB / 32;      // k16 MFMA steps per chunk
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;    // 32x32 accumulator tiles per wave
    constexpr int AG = BM / RPP / NW, BG = BN / RPP / NW;  // DMA pieces per wave
    constexpr int NDMA = AG + BG;                          // LDS-DMA instructions per wave per chunk
    static_assert(TM >= 1 && TN >= 1 && AG >= 1 && BG >= 1, "tile / wave layout");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // NST * (BM + BN) * RB bytes
    unsigned char* As = smem;
    unsigned char* Bs = smem + NST * BM * RB;

    const rdpn6d_conv_desc& d = a.d;
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

    const int prow = lane / SL;   // row inside the DMA piece
    const int pslot = lane % SL;  // physical slot written by this lane
    int a_iy[AG], a_ix[AG];
    unsigned a_off[AG];
    bool a_ok[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int row = (wave + NW * i) * RPP + prow;
        const long long m = m0 + row;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? (int)m : 0;
        const int b = mm / a.HoWo;
        const int rem = mm - b * a.HoWo;
        const int oy = rem / d.Wo;
        const int ox = rem - oy * d.Wo;
        a_iy[i] = oy * d.stride;
        a_ix[i] = ox * d.stride;
        const int lslot = pslot ^ ((row / RPB) & (SL - 1));
        a_off[i] = ((unsigned)(b * d.H * d.W) * (unsigned)d.in_cs + (unsigned)d.in_co) * 2u + (unsigned)lslot * 16u;
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
    const unsigned oob = a.x_bytes;
    const unsigned px_bytes = (unsigned)d.in_cs * 2u;

    auto stage_chunk = [&](const int tap, const int cc, const int st) {
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned c0b = (unsigned)cc * (unsigned)RB;
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            const unsigned off = ok ? a_off[i] + (unsigned)(iy * d.W + ix) * px_bytes + c0b : oob;
            unsigned char* dst = As + (st * BM + (wave + NW * i) * RPP) * RB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
        }
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 2u + c0b;
#pragma unroll
        for (int i = 0; i < BG; ++i) {
            unsigned char* dst = Bs + (st * BN + (wave + NW * i) * RPP) * RB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr_t)dst, 16, (int)(w_off[i] + wk), 0, 0, 0);
        }
    };
    // K order: channel-chunk major, taps innermost (see conv_igemm.hip); split-K: this workgroup's slice of the chunks
    const int k_begin = (int)blockIdx.y * a.kper;
    const int nk = min(a.kper, a.nk - k_begin);
    int ld_cc = k_begin / d.ntaps;
    int ld_tap = k_begin - ld_cc * d.ntaps, ld_left = nk - 1;
    auto next_chunk = [](int& tap, int& cc, int& left, const int ntaps) {
        const int go = left > 0 ? 1 : 0;
        left -= go;
        tap += go;
        const int wrap = tap == ntaps ? 1 : 0;
        tap = wrap ? 0 : tap;
        cc += wrap;
    };

    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31;
    const int half = lane >> 5;

    auto read_frags = [&](int st, u32x4 (&fa)[TM][NJ], u32x4 (&fb)[TN][NJ]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = wm * (BM / WM) + i * 32 + frow;
            const int sw = (R / RPB) & (SL - 1);
            const unsigned char* p = As + (st * BM + R) * RB;
#pragma unroll
            for (int j = 0; j < NJ; ++j) fa[i][j] = *reinterpret_cast<const u32x4*>(p + (((2 * j + half) ^ sw) << 4));
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int R = wn * (BN / WN) + jn * 32 + frow;
            const int sw = (R / RPB) & (SL - 1);
            const unsigned char* p = Bs + (st * BN + R) * RB;
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
                    acc[i][jn] = RD_LP_MFMA_32x32x16(fa[i][j], fb[jn][j], acc[i][jn]);
    };

    u32x4 fa0[TM][NJ], fb0[TN][NJ], fa1[TM][NJ], fb1[TN][NJ];
    if constexpr (NST == 2) {
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
    } else {
        // Invariant at the top of step kc: fragments of chunk kc in registers; stage (kc+1)%3 holds chunk kc+1, landed
        // and published; chunk kc+2 is in flight into stage (kc+2)%3; stage kc%3 is free (its fragments were read one
        // step ago and a barrier has passed).  Step: DMA chunk kc+3 -> stage kc%3; LDS -> registers of chunk kc+1; MFMA
        // of chunk kc; wait until only this step's NDMA loads are outstanding (chunk kc+2 has landed for this wave)
        // and the fragment reads have returned; barrier (publishes chunk kc+2, frees stage (kc+1)%3).
        auto publish = [&]() {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        stage_chunk(ld_tap, ld_cc, 0);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
        stage_chunk(ld_tap, ld_cc, 1);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
        stage_chunk(ld_tap, ld_cc, 2);
        next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(0, fa0, fb0);
        publish();
        int st_free = 0, st_next = 1;
        const int npairs = nk >> 1;
        for (int pr = 0; pr < npairs; ++pr) {
            stage_chunk(ld_tap, ld_cc, st_free);
            next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
            read_frags(st_next, fa1, fb1);
            mma(fa0, fb0);
            publish();
            st_free = st_next;
            st_next = st_next == 2 ? 0 : st_next + 1;

            stage_chunk(ld_tap, ld_cc, st_free);
            next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
            read_frags(st_next, fa0, fb0);
            mma(fa1, fb1);
            publish();
            st_free = st_next;
            st_next = st_next == 2 ? 0 : st_next + 1;
        }
        if (nk & 1) mma(fa0, fb0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing in LDS when the block retires
    }

    conv_bf16_epilogue<BM, BN, WM, WN, TM, TN, STATS>(a, acc, smem, m0, n0, wave, lane, wm, wn);
}

// split-K second pass: fixed-order sum of the partial accumulators + the epilogue (output / residual bf16, or fp32)
__global__ void conv_bf16_splitk_epilogue_kernel(const float* __restrict__ partial, int nsplit, long long M, rdpn6d_conv_desc d,
                                                 int out_f32)
{
    const long long total = M * d.N;
    const bf16_t* resb = reinterpret_cast<const bf16_t*>(d.res);
    bf16_t* yb = reinterpret_cast<bf16_t*>(d.y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / d.N;
        const int n = (int)(i - m * d.N);
        const float* p = partial + m * d.Npad + n;  // (slice order kept; eight loads in flight, see conv_igemm.hip)
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
        if (d.res) v += out_f32 ? d.res[m * d.res_cs + d.res_co + n] : bf2f(resb[m * d.res_cs + d.res_co + n]);
        if (d.act == 1) v = v > 0.f ? v : 0.f;
        else if (d.act == 2) v = v > 0.f ? v : v * d.slope;
        if (out_f32) d.y[m * d.out_cs + d.out_co + n] = v;
        else yb[m * d.out_cs + d.out_co + n] = f2bf(v);
    }
}

// conv_igemm_bf16_pp.hip: the eight-wave ping-pong form for the trunk layers (N % 128 == 0, >= 224 tiles)
int conv_lp_pp_plan(const ConvBArgs& a, int rb, int* pbm, int* pbn);
int conv_lp_launch_pp(const ConvBArgs& a, int shape, hipStream_t s);
// conv_igemm_bf16_8ph.hip: the 256x256 8-phase form for the large GEMM-like layers
bool conv_bf16_8ph_eligible(const ConvBArgs& a, int rb);
int conv_bf16_launch_8ph(const ConvBArgs& a, hipStream_t s);

static int g_bforce_bm = 0, g_bforce_bn = 0, g_bforce_rb = 0, g_bforce_nst = 0;
extern "C" void rdpn6d_conv_bf16_force_stages(int nst) { g_bforce_nst = nst; }  // 0 = auto, 2 | 3 (profiling: LDS stages of the 4-wave tiles)
extern "C" void rdpn6d_conv_bf16_force_chunk(int row_bytes) { g_bforce_rb = row_bytes; }  // 0 = auto, 64 | 128 (profiling)
extern "C" void rdpn6d_conv_bf16_force_tile(int bm, int bn) { g_bforce_bm = bm; g_bforce_bn = bn; }

static void conv_bf16_pick_tile(const rdpn6d_conv_desc* d, long long M, int rb, int* pbm, int* pbn)
{
    int bn = (d->Npad % 128 == 0) ? 128 : 64;
    int bm = 128;
    // (the 256x128 / 8-wave / 3-stage form measures 0.96x of 128x128 on the head layers - 791 vs 821 TFLOP/s - and is
    //  therefore only reachable through rdpn6d_conv_bf16_force_tile(256, 128))
    if (bm == 128 && (long long)rd_cdiv(M, 128) * (d->Npad / bn) < 512) bm = 64;
    if (bm == 64 && bn == 128 && (long long)rd_cdiv(M, 64) * (d->Npad / 128) < 512) bn = 64;
    // 256x256 8-phase kernel (conv_igemm_bf16_8ph.hip): GEMM-like layers with at least one full round of 256-row tiles
    const int nk = d->ntaps * (d->Cin / 64);
    const bool can8 = rb == 128 && d->Npad % 256 == 0 && d->N == d->Npad && nk >= 2 && (nk & 1) == 0;
    if (can8 && !g_bforce_bm && (long long)rd_cdiv(M, 256) * (d->Npad / 256) >= 256) bm = bn = 256;
    if (g_bforce_bm && (g_bforce_bm != 256 || (rb == 128 && d->Npad % 128 == 0))) bm = g_bforce_bm;
    if (g_bforce_bn && d->Npad % g_bforce_bn == 0) bn = g_bforce_bn;
    if (bm == 256 && !(bn == 256 && can8)) bn = 128;
    *pbm = bm;
    *pbn = bn;
}

extern "C" int rdpn6d_conv_bf16_tile_for(const rdpn6d_conv_desc* d, int* bm, int* bn)
{
    RD_REQUIRE(d && bm && bn, "null pointer");
    conv_bf16_pick_tile(d, (long long)d->B * d->Ho * d->Wo, d->Cin % 64 == 0 ? 128 : 64, bm, bn);
    return RDPN6D_OK;
}

// 1 when rdpn6d_conv2d_bf16 (fused output, no split-K, no forced tile) runs this problem on the eight-wave ping-pong kernel
extern "C" int rdpn6d_conv_bf16_uses_pingpong(const rdpn6d_conv_desc* d, int out_f32)
{
    if (!d || d->Cin <= 0 || d->Cin % 64 != 0 || g_bforce_bm || g_bforce_rb == 64) return 0;
    ConvBArgs a;
    a.d = *d;
    a.M = (long long)d->B * d->Ho * d->Wo;
    a.nk = d->ntaps * (d->Cin / 64);
    const int al = out_f32 ? 4 : 8;
    a.vec_out = (d->out_cs % al == 0 && d->out_co % al == 0 && (!d->res || (d->res_cs % al == 0 && d->res_co % al == 0))) ? 1 : 0;
    int bm, bn;
    conv_bf16_pick_tile(d, a.M, 128, &bm, &bn);
    if (bm == 256 && bn == 256 && a.vec_out) return 0;  // the 256x256 eight-phase kernel takes it
    return conv_lp_pp_plan(a, 128, &bm, &bn) >= 0 ? 1 : 0;
}

template <int BM, int BN, int RB, int WM, int WN, int NST>
static int conv_bf16_launch_one(const ConvBArgs& a, int nsplit, hipStream_t s)
{
    constexpr int lds_stage = NST * (BM + BN) * RB;
    constexpr int lds_epi = WM * WN * 32 * (BN / WN + 8) * 4;  // per-wave transpose slices of the coalesced epilogue
    constexpr int lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    if (a.stats) {  // training: the instantiation whose epilogue writes BatchNorm partial sums (RB = 128 only: the trunk / head layers)
        if constexpr (RB == 128 && WM * WN == 4) {
            if (a.bnb_y) {  // ... of a residual block's last BatchNorm: mask from the stored block output
                auto kern = conv_igemm_bf16_kernel<BM, BN, RB, WM, WN, NST, 2>;
                if (lds > 64 * 1024) {
                    RD_LDS_OPT_IN(kern, lds);
                }
                hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles), (unsigned)nsplit), dim3(64 * WM * WN), lds, s, a);
                return RDPN6D_OK;
            }
            auto kern = conv_igemm_bf16_kernel<BM, BN, RB, WM, WN, NST, 1>;
            if (lds > 64 * 1024) {
                RD_LDS_OPT_IN(kern, lds);
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles), (unsigned)nsplit), dim3(64 * WM * WN), lds, s, a);
            return RDPN6D_OK;
        }
    }
    auto kern = conv_igemm_bf16_kernel<BM, BN, RB, WM, WN, NST>;
    if (lds > 64 * 1024) {
        RD_LDS_OPT_IN(kern, lds);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles), (unsigned)nsplit), dim3(64 * WM * WN), lds, s, a);
    return RDPN6D_OK;
}

template <int RB>
static int conv_bf16_launch(const ConvBArgs& a, int bm, int bn, int nsplit, hipStream_t s)
{
    if constexpr (RB == 128) {
        if (bm == 256) return conv_bf16_launch_one<256, 128, RB, 4, 2, 3>(a, nsplit, s);
    }
    // Three LDS stages (the DMA two chunks ahead, never drained inside the loop) for the 4-wave tiles: pays where the K loop is
    // long and the tile small - 64x64 tiles of the 512-channel layers (layer4: 72 chunks; 38 -> 25 us at B = 32, 40 -> 32 at B = 64) -
    // and loses occupancy (96 KiB for a 128x128 tile: one workgroup per CU instead of two) everywhere else (measured, B = 32 / 64)
    // Round 4: those measurements re-launched one problem on hot caches.  Inside a step every operand is cold (the working set of a
    // training step is GBs; weights are touched once per step): behind a 512-MB fill the two-stage loop, whose __syncthreads drains the
    // DMA queue every chunk, exposes the HBM miss of EVERY chunk - layer3 at B = 32 (36 chunks): 17.5 us hot, 30.8 us cold = what the
    // step's kernel trace shows - and three stages hide most of it (18.1 / 23.0 us); 128x128 40.3 -> 29.1, 64x128 35.6 -> 22.7.  Short
    // loops (layer1 / layer2: 9 / 18 chunks) stay on two stages (cold: 26.0 vs 28.7 us, 22.4 vs 26.2).  RDPN6D_CONV_LP_NST3_K: threshold.
    static const int k3 = getenv("RDPN6D_CONV_LP_NST3_K") ? atoi(getenv("RDPN6D_CONV_LP_NST3_K")) : 24;  // profiling
    const int nst = g_bforce_nst ? g_bforce_nst : ((a.kper >= 64 && bm == 64 && bn == 64) || a.kper >= k3 ? 3 : 2);
    if (nst == 3) {
        if (bm == 128 && bn == 128) return conv_bf16_launch_one<128, 128, RB, 2, 2, 3>(a, nsplit, s);
        if (bm == 128 && bn == 64) return conv_bf16_launch_one<128, 64, RB, 2, 2, 3>(a, nsplit, s);
        if (bm == 64 && bn == 128) return conv_bf16_launch_one<64, 128, RB, 2, 2, 3>(a, nsplit, s);
        return conv_bf16_launch_one<64, 64, RB, 2, 2, 3>(a, nsplit, s);
    }
    if (bm == 128 && bn == 128) return conv_bf16_launch_one<128, 128, RB, 2, 2, 2>(a, nsplit, s);
    if (bm == 128 && bn == 64) return conv_bf16_launch_one<128, 64, RB, 2, 2, 2>(a, nsplit, s);
    if (bm == 64 && bn == 128) return conv_bf16_launch_one<64, 128, RB, 2, 2, 2>(a, nsplit, s);
    return conv_bf16_launch_one<64, 64, RB, 2, 2, 2>(a, nsplit, s);
}

struct ConvBnBwd {  // see ConvBArgs::bnb_x / bnb_y
    const void* x;
    int cs, co;
    const float *mean, *invstd, *gamma, *beta;
    const void* y = nullptr;
    int ycs = 0, yco = 0;
};
static int conv2d_bf16_impl(const rdpn6d_conv_desc* d, int out_f32, int ksplit, float* workspace, void* stream, double* stats = nullptr,
                            int stats_row0 = 0, int* stats_rows = nullptr, const ConvBnBwd* bnb = nullptr);

extern "C" int rdpn6d_conv2d_bf16(const rdpn6d_conv_desc* d, int out_f32, void* stream)
{
    return conv2d_bf16_impl(d, out_f32, 1, nullptr, stream);
}

// split-K form (see rdpn6d_conv2d_splitk_f32): workspace of rdpn6d_conv_splitk_ws_floats(d, ksplit) floats, linear output
extern "C" int rdpn6d_conv2d_splitk_bf16(const rdpn6d_conv_desc* d, int out_f32, int ksplit, float* workspace, void* stream)
{
    RD_REQUIRE(ksplit >= 1 && (ksplit == 1 || workspace), "split-K needs a workspace");
    return conv2d_bf16_impl(d, out_f32, ksplit, workspace, stream);
}

// Training forward: the convolution + the per-channel partial sums of the BatchNorm that follows it, written by the epilogue
// (see ConvBArgs::stats).  stats: [stats_row0 + rows][N][2] doubles; *stats_rows = the rows this launch wrote, or 0 when the geometry
// does not allow it (ragged tiles, unaligned slices, fp32 output) - the convolution ran normally and the caller falls back to
// rdpn6d_bn_train_stats_bf16.  rdpn6d_bn_stats_finalize turns the rows into mean / invstd / running statistics.
extern "C" int rdpn6d_conv2d_bf16_bnstats(const rdpn6d_conv_desc* d, double* stats, int stats_row0, int* stats_rows, void* stream)
{
    RD_REQUIRE(stats && stats_rows && stats_row0 >= 0, "stats buffer");
    return conv2d_bf16_impl(d, 0, 1, nullptr, stream, stats, stats_row0, stats_rows);
}

// Training backward: the input-gradient convolution of the layer AFTER a BatchNorm + ReLU, whose epilogue also writes that BatchNorm's
// backward partial sums - per channel (sum g, sum g * xhat) with the ReLU mask re-derived from the BatchNorm input bn_x - instead of a
// separate reduction pass over dy and x (rdpn6d_bn_relu_backward_*'s first kernel).  Rows / fall-back as rdpn6d_conv2d_bf16_bnstats;
// rdpn6d_bn_relu_backward_apply_bf16 finishes.  Needs a linear output geometry and bn_x laid out like the output (same pixels).
extern "C" int rdpn6d_conv2d_bf16_bnbwd(const rdpn6d_conv_desc* d, const void* bn_x, int bn_cs, int bn_co, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, double* partial, int* rows,
                                        void* stream)
{
    RD_REQUIRE(bn_x && mean && invstd && gamma && beta && partial && rows, "null pointer");
    RD_REQUIRE(bn_cs % 8 == 0 && bn_co % 8 == 0 && d && bn_co + d->N <= bn_cs, "BatchNorm input slice: 16-byte aligned, N channels");
    RD_REQUIRE(d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo, "linear output geometry");
    const ConvBnBwd b = {bn_x, bn_cs, bn_co, mean, invstd, gamma, beta};
    return conv2d_bf16_impl(d, 0, 1, nullptr, stream, partial, 0, rows, &b);
}

// The same for the LAST BatchNorm of a residual block, y = relu(bn(x) + identity): this launch is the input-gradient convolution that
// writes the gradient w.r.t. the block output y (its own residual input - the next block's identity gradient - added in the epilogue),
// and the ReLU mask is the STORED y > 0 (chan_partial_kernel's relu == 1).  rdpn6d_bn_backward_apply_bf16 finishes.  *rows == 0 (and a
// normal convolution) when the launch falls to a kernel without this epilogue (the 256x256 one) or the geometry does not allow it.
extern "C" int rdpn6d_conv2d_bf16_bnbwd_y(const rdpn6d_conv_desc* d, const void* bn_x, int bn_cs, int bn_co, const void* bn_y, int y_cs,
                                          int y_co, const float* mean, const float* invstd, double* partial, int* rows, void* stream)
{
    RD_REQUIRE(bn_x && bn_y && mean && invstd && partial && rows, "null pointer");
    RD_REQUIRE(bn_cs % 8 == 0 && bn_co % 8 == 0 && d && bn_co + d->N <= bn_cs, "BatchNorm input slice: 16-byte aligned, N channels");
    RD_REQUIRE(y_cs % 8 == 0 && y_co % 8 == 0 && y_co + d->N <= y_cs, "block output slice: 16-byte aligned, N channels");
    if (!(d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo)) {
        // strided / phased output (an input-gradient phase of a stride-2 convolution): the epilogue's row bookkeeping does not apply -
        // as promised above, the plain convolution runs and *rows == 0 tells the caller to take the sums in a separate pass
        *rows = 0;
        return conv2d_bf16_impl(d, 0, 1, nullptr, stream, nullptr, 0, nullptr, nullptr);
    }
    ConvBnBwd b = {bn_x, bn_cs, bn_co, mean, invstd, nullptr, nullptr};
    b.y = bn_y;
    b.ycs = y_cs;
    b.yco = y_co;
    return conv2d_bf16_impl(d, 0, 1, nullptr, stream, partial, 0, rows, &b);
}

static int conv2d_bf16_impl(const rdpn6d_conv_desc* d, int out_f32, int ksplit, float* workspace, void* stream, double* stats,
                            int stats_row0, int* stats_rows, const ConvBnBwd* bnb)
{
    RD_REQUIRE(d && d->x && d->w && d->y, "null pointer");
    RD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "empty tensor");
    RD_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, "Cin must be a positive multiple of 32");
    RD_REQUIRE(d->in_cs % 8 == 0 && d->in_co % 8 == 0 && d->in_co + d->Cin <= d->in_cs, "input channel slice");
    RD_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9, "ntaps in 1..9");
    RD_REQUIRE(d->N > 0 && d->Npad >= d->N && d->Npad % 64 == 0, "Npad must be a multiple of 64 >= N");
    RD_REQUIRE(d->out_co + d->N <= d->out_cs, "output channel slice");
    RD_REQUIRE(d->stride >= 1 && d->osy >= 1 && d->osx >= 1, "strides");
    RD_REQUIRE((d->Ho - 1) * d->osy + d->ooy < d->OH && (d->Wo - 1) * d->osx + d->oox < d->OW, "output geometry");
    RD_REQUIRE(!d->res || d->res_co + d->N <= d->res_cs, "residual channel slice");
    ConvBArgs a;
    a.d = *d;
    a.M = (long long)d->B * d->Ho * d->Wo;
    RD_REQUIRE(a.M < (1LL << 31), "B*Ho*Wo must fit 31 bits");
    a.HoWo = d->Ho * d->Wo;
    const int rb = (d->Cin % 64 == 0 && g_bforce_rb != 64) ? 128 : 64;
    a.cchunks = d->Cin * 2 / rb;
    a.nk = d->ntaps * a.cchunks;
    a.Ktot = d->ntaps * d->Cin;
    a.linear_out = (d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo);
    a.out_f32 = out_f32 ? 1 : 0;
    const int al = out_f32 ? 4 : 8;  // elements per 16 bytes
    a.vec_out = (d->out_cs % al == 0 && d->out_co % al == 0 && (!d->res || (d->res_cs % al == 0 && d->res_co % al == 0))) ? 1 : 0;
    const long long xb = (long long)d->B * d->H * d->W * d->in_cs * 2;
    RD_REQUIRE(xb < (1LL << 32) - 64, "input tensor must be smaller than 4 GiB (32-bit buffer offsets)");
    a.x_bytes = (unsigned)xb;
    const long long wb = (long long)d->Npad * d->ntaps * d->Cin * 2;
    RD_REQUIRE(wb < (1LL << 32) - 64, "packed weights must be smaller than 4 GiB");
    a.w_bytes = (unsigned)wb;
    a.dy_pack = a.dx_pack = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        RD_REQUIRE(d->dy[t] >= -8 && d->dy[t] <= 7 && d->dx[t] >= -8 && d->dx[t] <= 7, "tap offsets must be in -8..7");
        a.dy_pack |= (unsigned long long)(d->dy[t] + 8) << (4 * t);
        a.dx_pack |= (unsigned long long)(d->dx[t] + 8) << (4 * t);
    }
    hipStream_t s = (hipStream_t)stream;
    int bm, bn;
    conv_bf16_pick_tile(d, a.M, rb, &bm, &bn);
    a.mtiles = rd_cdiv(a.M, bm);
    a.ntiles = d->Npad / bn;
    a.kper = a.nk;
    a.partial = nullptr;
    int nsplit = 1;
    if (bm == 256 && bn == 256 && (ksplit > 1 || !a.vec_out)) {  // no split-K / unaligned-slice form of the 8-phase kernel
        bm = bn = 128;
        a.mtiles = rd_cdiv(a.M, bm);
        a.ntiles = d->Npad / bn;
    }
    if (stats_rows) *stats_rows = 0;
    if (stats && !out_f32 && a.vec_out && d->N == d->Npad && ksplit <= 1 && rb == 128 && !(bm == 256 && bn == 128) &&
        ((bm == 256 && bn == 256 && !(bnb && bnb->y)) || (a.M % bm == 0 && !(bm == 256 && bn == 256)))) {
        // every tile takes the coalesced epilogue (full column tiles: bn divides Npad = N; full row tiles, or the 8-phase kernel's masked rows)
        a.stats = stats;
        a.stats_row0 = stats_row0;
        if (bnb) {
            a.bnb_x = bnb->x; a.bnb_cs = bnb->cs; a.bnb_co = bnb->co;
            a.bnb_mean = bnb->mean; a.bnb_invstd = bnb->invstd; a.bnb_gamma = bnb->gamma; a.bnb_beta = bnb->beta;
            a.bnb_y = bnb->y; a.bnb_ycs = bnb->ycs; a.bnb_yco = bnb->yco;
        }
        *stats_rows = a.mtiles * 2;  // wave rows per tile: WM of the launch below
    }
    if (bm == 256 && bn == 256) {
        RD_REQUIRE(conv_bf16_8ph_eligible(a, rb), "256x256 tile needs Cin % 64 == 0, Npad % 256 == 0 and an even K-tile count");
        const int rc8 = conv_bf16_launch_8ph(a, s);
        if (rc8 != RDPN6D_OK) return rc8;
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    if (ksplit <= 1 && !g_bforce_bm) {
        int pbm, pbn;
        const int shape = conv_lp_pp_plan(a, rb, &pbm, &pbn);
        if (shape >= 0) {
            a.mtiles = rd_cdiv(a.M, pbm);
            a.ntiles = d->Npad / pbn;
            if (a.stats) *stats_rows = a.mtiles * 2;  // WM = 2 (the kernel masks the rows of a ragged last tile)
            const int rcp = conv_lp_launch_pp(a, shape, s);
            if (rcp != RDPN6D_OK) return rcp;
            RD_LAUNCH_CHECK();
            return RDPN6D_OK;
        }
    }
    if (ksplit > 1 && bm != 256) {
        RD_REQUIRE(a.linear_out, "split-K needs a linear output geometry");
        a.kper = rd_cdiv(a.nk, ksplit);
        nsplit = rd_cdiv(a.nk, a.kper);
        if (nsplit > 1) a.partial = workspace;
        else a.kper = a.nk;
    }
    const int rc = rb == 128 ? conv_bf16_launch<128>(a, bm, bn, nsplit, s) : conv_bf16_launch<64>(a, bm, bn, nsplit, s);
    if (rc != RDPN6D_OK) return rc;
    RD_LAUNCH_CHECK();
    if (nsplit > 1) {
        const long long total = a.M * d->N;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(conv_bf16_splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, workspace, nsplit, a.M, *d, a.out_f32);
        RD_LAUNCH_CHECK();
    }
    return RDPN6D_OK;
}
