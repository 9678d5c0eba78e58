// fp32-accurate (h2: two fp16 planes per operand, three partial products - see conv_igemm_h2.hip) implicit-GEMM convolution for the
// layers that are too narrow / too short for the 256x256 eight-phase kernel: the ResNet trunk (N = 64 .. 512, 4 096 .. 262 144 rows at
// B = 64).  Round 3: the 2x2-wave tile kernel of conv_igemm_h2.hip runs these layers at MfmaUtil 37-41 %; its probe build
// (profiles/r3_probe_tile_kernel.md) shows why: a K-chunk step is a SEQUENCE inside each wavefront - fragment reads (~250 cycles until
// they are back), the chunk's 1-KiB LDS-DMA pieces (~65 cycles of issue EACH, not covered by the wave's own MFMAs: 6 pieces = 410
// cycles), the 12 MFMAs (400), s_waitcnt + s_barrier (180) - and the second workgroup on the CU overlaps it only by chance.
//
// This kernel makes the overlap structural, the way the eight-phase kernel does: ONE workgroup of EIGHT wavefronts per CU in two
// groups of four (waves i and i + 4 share a SIMD) that run ONE BARRIER APART.  A wavefront alternates
//     L(k): ds_read the fragments of chunk k (single-buffered) + issue its share of the LDS-DMA pieces of chunk k + D   | s_barrier
//     M(k): the MFMAs of chunk k (the last PM pieces between them)                                                      | s_barrier
// and while group 0 is in M(k), group 1 is in L(k) (it executes one extra barrier before the loop, group 0 one after it): every
// SIMD always has one wave feeding the matrix pipe and one wave issuing reads / DMA.  The DMA runs D = NST - 1 chunks ahead through
// a ring of NST stages and is never drained inside the loop: counted s_waitcnt vmcnt at the end of both parts (below) guarantee
// that chunk k + 1 has landed - for BOTH groups' pieces - before the barrier that lets the other group read it.
//
// Tile BM x BN on a WM x WN grid of wavefronts (group = upper half of the row blocks), wave tile (BM / WM) x (BN / WN) of 32x32 MFMA
// tiles; 128-byte h2 rows, XOR-swizzled 16-byte slots and DMA addressing exactly as in the tile kernel.  The epilogue issues every
// residual load of the wave's tile BEFORE the LDS transposes (the tile kernel's one-dependent-load-per-row-group epilogue cost
// 9 000 cycles per workgroup, 8-34 % of its life).
#include "conv_h2_common.h"

#include <cstdio>
#include <cstdlib>

#ifdef RDPN6D_PROBE
// probe build only (tools/probe_h2_tile.py): per-wave cycle sums of the parts of a ping-pong step
__device__ unsigned long long* g_h2pp_probe = nullptr;
extern "C" int rdpn6d_debug_h2pp_probe(void* buf)
{
    RD_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_h2pp_probe), &buf, sizeof(buf)));
    return RDPN6D_OK;
}
#define PP_T(i) const unsigned long long pt##i = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define PP_ACC(j, a, b) pp_sum[j] += pt##b - pt##a
#else
#define PP_T(i)
#define PP_ACC(j, a, b)
#endif

namespace {

// BFG (round 5, "B from global"): the weight fragments do not pass through LDS.  The probe of the plain form says the K loop is bound by
// the CU's LDS port - per 32-channel chunk a 128x128 tile moves 32 KiB of DMA into LDS and its eight waves read 96 KiB of fragments out
// of it: 1 024 cycles at 128 B/clk against 768 cycles of MFMA.  A third of that traffic is the weight tile (16 KiB in, 32 KiB out), and
// weights are static: packed FRAGMENT-MAJOR at plan time ([n/32][tap][chunk][slot 0..7][row 0..31][16 B]: rdpn6d_h2_weight_frag), the
// four 16-byte fragments a lane feeds to the MFMAs of a chunk are four fully coalesced 1-KiB wave loads (lane * 16 bytes apart) straight
// into registers - two chunks ahead through a three-slot register ring, so the loop is unrolled three times.  LDS then carries the
// activation tile only (16 KiB in, 64 KiB out per chunk: 640 cycles), the vector-memory path the activation DMA plus the weight
// fragments (48 KiB per chunk at 64 B/clk: 768 cycles) and the matrix pipe its 768 cycles - three balanced paths instead of one
// overloaded.  Same k order per accumulator: results bit-identical to the plain form.
template <int BM, int BN, int WM, int WN, int NST, int PM, bool BFG = false>
__global__ __launch_bounds__(512) void conv_h2_pp_kernel(const ConvH2Args ax)
{
    constexpr int RB = 128, RPP = 8, NW = 8;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int AG = BM / RPP / NW, BG = BFG ? 0 : BN / RPP / NW;  // LDS-DMA pieces per wave and chunk
    constexpr int NB = BFG ? 4 * TN : 0;                              // BFG: 16-byte weight-fragment loads per wave and chunk
    constexpr int P = AG + BG, PL = P - PM, D = NST - 1;
    constexpr int Q = P + NB;                                         // entries of the wave's vmcnt queue per chunk
    static_assert(WM * WN == NW && WM == 2 && TM >= 1 && TN >= 1 && AG >= 1 && (BFG || BG >= 1), "wave grid / tile");
    static_assert(PM >= 0 && PM <= 2 && PL >= 1 && D >= 2, "pieces in the MFMA part; at least three stages");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + NST * BM * RB;
#ifdef RDPN6D_PROBE
    const unsigned long long pp_start = __builtin_readcyclecounter();
    unsigned long long pp_e[6] = {0, 0, 0, 0, 0, 0};
#endif

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
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = grp, wn = wq;

    // ---- DMA addressing: this wave moves pieces wave, wave + 8, ... (8 rows x 128 B each) of the A rows and of the B rows of a chunk
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned px_bytes = (unsigned)d.in_cs * 4u;  // an h2 pixel is 2 x in_cs halfs
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
        a_base[i] = (unsigned)((b * d.H + iy) * d.W + ix) * px_bytes + (unsigned)d.in_co * 4u + (unsigned)lslot * 16u;
        unsigned mask = 0;
        for (int t = 0; t < d.ntaps; ++t) {
            const int dy = (int)((a.dy_pack >> (4 * t)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * t)) & 15ull) - 8;
            mask |= (ok && (unsigned)(iy + dy) < (unsigned)d.H && (unsigned)(ix + dx) < (unsigned)d.W) ? (1u << t) : 0u;
        }
        a_mask[i] = mask;
    }
    unsigned w_off[BG > 0 ? BG : 1];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave + NW * i) * RPP + prow;
        const int lslot = pslot ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)(n0 + row) * (unsigned)a.Ktot * 4u + (unsigned)lslot * 16u;
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BFG ? ax.w_frag : (const void*)d.w), 0, a.w_bytes, 0x00020000);

    unsigned dma_off[P];
    auto stage_addr = [&](const int tap, const int cc, const bool valid) {
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned toff = (unsigned)((dy * d.W + dx) * (int)px_bytes + cc * RB);  // wave-uniform
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;                                 // past the last chunk: out of range, zeros land
#pragma unroll
        for (int i = 0; i < AG; ++i) dma_off[i] = (a_base[i] + toff) | (((a_mask[i] >> tap) & 1u) - 1u) | sel;
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 4u + (unsigned)cc * (unsigned)RB;
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

    // ---- epilogue operands, requested FIRST (they are the oldest entries of the vmcnt queue, so the prologue's wait for chunk 0 covers
    // them): the h2 residual records of this lane's output rows (TM * NRR independent 2 x 16-byte loads) and the folded BatchNorm
    // scale / shift of its channels.  Requested in the epilogue they cost the workgroup 4 000 - 9 000 cycles of exposed HBM latency
    // (probe build, tools/probe_h2_pp.py: every workgroup of the launch reaches its epilogue at the same time).
    constexpr int CS = WTN + 8, LPR = WTN / 8, RPI = 64 / LPR, NRR = 32 / RPI;
    const int frow = lane & 31;
    const int nb = n0 + wn * WTN;
    const int rrow = lane / LPR, c8 = (lane % LPR) * 8;
    const long long mbase = m0 + wm * WTM;
    f16x8 rh[TM][NRR], rl[TM][NRR];
    long long pixs[TM][NRR];
    const bool res_pre = ax.res_h2 != nullptr && d.res == nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int rr = 0; rr < NRR; ++rr) {
            const long long mrow = mbase + i * 32 + rr * RPI + rrow;
            const bool ok = mrow < a.M && nb + c8 < d.N;
            pixs[i][rr] = ok ? h2_pixel_of(a, mrow) : -1;
            if (res_pre && ok) {
                const int c = d.res_co + nb + c8;
                const _Float16* rp = reinterpret_cast<const _Float16*>(ax.res_h2) + pixs[i][rr] * (2 * d.res_cs) + (c >> 5) * 64 + (c & 31);
                rh[i][rr] = *reinterpret_cast<const f16x8*>(rp);
                rl[i][rr] = *reinterpret_cast<const f16x8*>(rp + 32);
            } else {
                rh[i][rr] = f16x8{};
                rl[i][rr] = f16x8{};
            }
        }
    float scj[TN], shj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nb + j * 32 + frow;
        scj[j] = d.scale ? d.scale[n] : 1.f;
        shj[j] = d.shift ? d.shift[n] : 0.f;
    }

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
    u32x4 fa[TM][4], fb[BFG ? 3 * TN : TN][4];  // BFG: ring of three chunks (slot s holds chunk k with k % 3 == s)
    // BFG: the fragments of 32-column block (n0 + wn * WTN) / 32 + jn, chunk (tap, cc): 4 KiB at ((n32 * ntaps + tap) * cchunks + cc) * 4096,
    // fragment j of lane l at + j * 1024 + l * 16
    const unsigned wf_lane = (unsigned)lane * 16u;
    const unsigned wf_n32 = (unsigned)(n0 + wn * WTN) / 32u;
    auto load_bfrags = [&](auto slotc, const int tap, const int cc, const bool valid) {
        constexpr int slot = decltype(slotc)::value;
        if constexpr (BFG) {
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const unsigned base = (((wf_n32 + (unsigned)jn) * (unsigned)d.ntaps + (unsigned)tap) * (unsigned)a.cchunks + (unsigned)cc) * 4096u;
                const unsigned so = valid ? base : 0xFFFFF000u;  // past the last chunk: out of range, zeros (never used)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fb[slot * TN + jn][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, (int)wf_lane, (int)(so + j * 1024u), 0));
            }
        }
    };
    auto read_frags = [&](const int st) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = wm * WTM + i * 32 + frow;
            const int sw = (R >> 1) & 7;
            const unsigned char* q = As + ((st * BM) + R) * RB;
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[i][j] = *reinterpret_cast<const u32x4*>(q + (((2 * j + half) ^ sw) << 4));
        }
        if constexpr (!BFG) {
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const int R = wn * WTN + jn * 32 + frow;
                const int sw = (R >> 1) & 7;
                const unsigned char* q = Bs + ((st * BN) + R) * RB;
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[jn][j] = *reinterpret_cast<const u32x4*>(q + (((2 * j + half) ^ sw) << 4));
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

    auto mma_group = [&](auto prc, auto slotc) {
        constexpr int pr = decltype(prc)::value, slot = decltype(slotc)::value;
        H2_PAIRS;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = h2_mfma(fa[i][H2_PA[pr]], fb[slot * TN + jn][H2_PB[pr]], acc[i][jn]);
    };

    // BFG: position of the weight-fragment stream (two chunks ahead of the MFMAs)
    int lb_cc = 0, lb_tap = 0, lb_idx = 0;
    auto next_b = [&]() {
        ++lb_idx;
        ++lb_tap;
        const int wrap = lb_tap == d.ntaps ? 1 : 0;
        lb_tap = wrap ? 0 : lb_tap;
        lb_cc += wrap;
    };
    static_assert(!BFG || D <= 3, "BFG: the counted waits below assume the weight stream (2 chunks ahead) is not behind the DMA ring");

    // ---- prologue: chunks 0 .. D-1 into stages 0 .. D-1, issued as the steps -D .. -1 of the loop would (step s: first PL pieces of chunk
    // s + D, [BFG] the weight fragments of chunk s + 2, the remaining PM pieces) so that the counted waits hold from step 0 on
    auto pro_step = [&](auto cc_) {
        constexpr int c = decltype(cc_)::value;
        if constexpr (c < D) {
            stage_addr(ld_tap, ld_cc, ld_idx < nk);
            stage_pieces(ic<0>{}, ic<PL>{}, c);
            if constexpr (BFG && c - D + 2 >= 0) {
                load_bfrags(ic<c - D + 2>{}, lb_tap, lb_cc, lb_idx < nk);
                next_b();
            }
            stage_pieces(ic<PL>{}, ic<P>{}, c);
            next_chunk();
        }
    };
    pro_step(ic<0>{});
    pro_step(ic<1>{});
    pro_step(ic<2>{});
    pro_step(ic<3>{});
    static_assert(D <= 4, "prologue steps");
    stage_addr(ld_tap, ld_cc, ld_idx < nk);                              // addresses of chunk D, issued in step 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * Q) : "memory");  // chunk 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                        // ... and everybody's
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on

    int st_rd = 0, st_wr = D;  // stage of chunk k, stage of chunk k + D
#ifdef RDPN6D_PROBE
    unsigned long long pp_sum[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long pp_t0 = __builtin_readcyclecounter();
#endif
    // vmcnt queue of a wave, per step s: [PL pieces of chunk s + D] [NB weight fragments of chunk s + 2] [PM pieces of chunk s + D]
    //   end of L(k): chunk k + 1's pieces (steps k + 1 - D) are older than steps k + 2 - D .. k - 1 (Q each) and this step's PL + NB;
    //                BFG: the fragments of chunk k (step k - 2) are older than 2 Q entries
    //   end of M(k): chunk k + 1's pieces are older than steps k + 2 - D .. k
    constexpr int N1a = (D - 2) * Q + PL + NB, N1 = (BFG && N1a > 2 * Q) ? 2 * Q : N1a, N3 = (D - 1) * Q;
    static_assert(N1 <= 63 && N3 <= 63, "vmcnt is a 6-bit counter");
    auto step = [&](auto slotc) {
        constexpr int slot = decltype(slotc)::value;  // BFG: ring slot of this chunk's weight fragments
        // ---- L(k): fragments of chunk k; this wave's first PL pieces of chunk k + D
        __builtin_amdgcn_sched_barrier(0);
        PP_T(0);
        read_frags(st_rd);
        __builtin_amdgcn_sched_barrier(0);
        stage_pieces(ic<0>{}, ic<PL>{}, st_wr);  // (addresses of chunk k + D: computed between the MFMAs of the previous step)
        if constexpr (BFG) {
            load_bfrags(ic<(slot + 2) % 3>{}, lb_tap, lb_cc, lb_idx < nk);
            next_b();
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_T(1);
        // chunk k + 1 has landed (and, BFG, this chunk's weight fragments); the fragment reads are back (nobody may still be reading a
        // stage the other group is about to re-fill)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N1) : "memory");
        PP_T(2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- M(k): the MFMAs of chunk k, the remaining PM pieces between them
        __builtin_amdgcn_sched_barrier(0);
        PP_T(3);
        __builtin_amdgcn_s_setprio(1);
        mma_group(ic<0>{}, slotc);
        mma_group(ic<1>{}, slotc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PM >= 1) stage_piece(ic<PL>{}, st_wr);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(ic<2>{}, slotc);
        mma_group(ic<3>{}, slotc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PM >= 2) stage_piece(ic<PL + 1>{}, st_wr);
        __builtin_amdgcn_sched_barrier(0);
        // the DMA addresses of the NEXT step's chunk (k + 1 + D), here, where their ~50 scalar / vector instructions run under the
        // MFMAs instead of lengthening the L part (probe: per chunk 1 105 -> 987 cycles on layer3)
        next_chunk();
        stage_addr(ld_tap, ld_cc, ld_idx < nk);
        mma_group(ic<4>{}, slotc);
        mma_group(ic<5>{}, slotc);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        PP_T(4);
        st_rd = st_rd == NST - 1 ? 0 : st_rd + 1;
        st_wr = st_wr == NST - 1 ? 0 : st_wr + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N3) : "memory");  // chunk k + 1 has landed
        PP_T(5);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PP_T(6);
        PP_ACC(0, 0, 1);  // reads + addresses + PL pieces issued
        PP_ACC(1, 1, 2);  // s_waitcnt vmcnt + lgkmcnt
        PP_ACC(2, 2, 3);  // barrier after L
        PP_ACC(3, 3, 4);  // MFMAs + PM pieces
        PP_ACC(4, 4, 5);  // s_waitcnt vmcnt
        PP_ACC(5, 5, 6);  // barrier after M
    };
    if constexpr (BFG) {
        int k = 0;
        for (; k + 3 <= nk; k += 3) {
            step(ic<0>{});
            step(ic<1>{});
            step(ic<2>{});
        }
        if (k < nk) step(ic<0>{});
        if (k + 1 < nk) step(ic<1>{});
    } else {
        for (int k = 0; k < nk; ++k) step(ic<0>{});
    }
#ifdef RDPN6D_PROBE
    const unsigned long long pp_loop_end = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the out-of-range pieces of the last D steps: nothing may still be landing in LDS)
    if (grp == 0) __builtin_amdgcn_s_barrier();       // re-align the two groups
#ifdef RDPN6D_PROBE
    pp_e[0] = __builtin_readcyclecounter();
#endif

    // ---- epilogue
    {
        const int hi = lane >> 5;
        // every wave is past its last fragment read (lgkmcnt(0) in its last L part) and no DMA is in flight (vmcnt(0) above): a bare
        // barrier, not __syncthreads() - whose fence would also wait for anything a kernel variant still has in flight to global memory
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef RDPN6D_PROBE
        pp_e[1] = __builtin_readcyclecounter();
#endif
        float* cst = reinterpret_cast<float*>(smem) + wave * (32 * CS);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e] * scj[j] + shj[j];
#pragma unroll
            for (int rr = 0; rr < NRR; ++rr) {
                const int row = rr * RPI + rrow;
                if (pixs[i][rr] >= 0) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    if (res_pre) h2_finish_row8_t<true>(ax, v, pixs[i][rr], nb + c8, rh[i][rr], rl[i][rr]);
                    else h2_finish_row8_t<false>(ax, v, pixs[i][rr], nb + c8, rh[i][rr], rl[i][rr]);
                }
            }
        }
    }
#ifdef RDPN6D_PROBE
    pp_e[2] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp_e[3] = __builtin_readcyclecounter();
    if (g_h2pp_probe && lane == 0) {
        unsigned long long* o = g_h2pp_probe + ((size_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 6; ++i) o[i] = pp_sum[i];
        o[6] = pp_loop_end - pp_t0;
        o[7] = (unsigned long long)nk;
        o[8] = pp_t0 - pp_start;        // set-up + prologue (first chunks landed, first barrier)
        o[9] = pp_e[0] - pp_loop_end;   // drain + re-align
        o[10] = pp_e[1] - pp_e[0];      // residual loads issued + __syncthreads
        o[11] = pp_e[2] - pp_e[1];      // scale / shift, transposes, finish, stores issued
        o[12] = pp_e[3] - pp_e[2];      // stores done
    }
#endif
}

template <int BM, int BN, int WM, int WN, int NST, int PM, bool BFG = false>
int launch_pp(const ConvH2Args& ax, hipStream_t s)
{
    constexpr int lds_stage = NST * (BM + (BFG ? 0 : BN)) * 128;
    constexpr int lds_epi = 8 * 32 * (BN / WN + 8) * 4;
    constexpr int lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv_h2_pp_kernel<BM, BN, WM, WN, NST, PM, BFG>;
    RD_LDS_OPT_IN(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(ax.b.mtiles * ax.b.ntiles)), dim3(512), lds, s, ax);
    return RDPN6D_OK;
}

}  // namespace

// shape: 0 = 128x128 tile (2 x 4 waves of 64x32, four LDS stages), 2 = 256x128 (2 x 4 waves of 128x32, three stages)
int conv_h2_launch_pp(ConvH2Args& ax, int shape, hipStream_t s)
{
    static const int pm = getenv("RDPN6D_H2_PP_PM") ? atoi(getenv("RDPN6D_H2_PP_PM")) : 2;  // profiling: DMA pieces inside the MFMA part
    if (ax.w_frag) {  // weight fragments straight from L2 (fragment-major weights: rdpn6d_h2_weight_frag); A ring one stage deeper
        static const int bpm = getenv("RDPN6D_H2_BFG_PM") ? atoi(getenv("RDPN6D_H2_BFG_PM")) : 1;
        if (shape == 0) return bpm == 0 ? launch_pp<128, 128, 2, 4, 4, 0, true>(ax, s) : launch_pp<128, 128, 2, 4, 4, 1, true>(ax, s);
        if (shape == 2) return bpm == 0 ? launch_pp<256, 128, 2, 4, 4, 0, true>(ax, s) : bpm == 1 ? launch_pp<256, 128, 2, 4, 4, 1, true>(ax, s) : launch_pp<256, 128, 2, 4, 4, 2, true>(ax, s);
    }
    if (shape == 0) return pm == 0 ? launch_pp<128, 128, 2, 4, 4, 0>(ax, s) : pm == 1 ? launch_pp<128, 128, 2, 4, 4, 1>(ax, s) : launch_pp<128, 128, 2, 4, 4, 2>(ax, s);
    if (shape == 2) return pm == 0 ? launch_pp<256, 128, 2, 4, 3, 0>(ax, s) : pm == 1 ? launch_pp<256, 128, 2, 4, 3, 1>(ax, s) : launch_pp<256, 128, 2, 4, 3, 2>(ax, s);
    rdpn6d_set_error("conv_h2_launch_pp: unknown tile shape %d", shape);
    return RDPN6D_EINVAL;
}
