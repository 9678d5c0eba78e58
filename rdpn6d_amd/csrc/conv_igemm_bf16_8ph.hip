// 256x256 implicit-GEMM convolution on the bf16 matrix pipe with an 8-phase ping-pong schedule (two K-tiles per loop
// iteration), for the layers that look like a large plain GEMM: Cin % 64 == 0, Npad % 256 == 0, an even number of
// 64-channel K-tiles and enough 256-row tiles to fill the chip (the dense head: 75 % of the network's FLOPs).
//
// Why a second kernel: the 128x128 / two-barriers-per-K-step form (conv_igemm_bf16.hip) tops out at ~36 % of the bf16
// peak on gfx950 whatever is done to its pipeline (cdna_hip_programming.md "the step-3 structure's
// ceiling"): every wave alternates between fragment reads and MFMAs, so the matrix pipe idles while LDS data returns.
// Here eight waves (2 x 4, each owning 128 x 64 outputs) form two groups of four - one wave of each group per SIMD -
// that run ONE BARRIER APART: while a group issues its 8 MFMAs of a phase (one 64x32 quadrant x K=64), the other
// group issues its LDS fragment reads and its share of the LDS-DMA, and the roles swap at the next barrier.
//
//   phase j of an EVEN K-tile t           fragment reads (ds_read_b128)        MFMA quadrant      DMA issued (2 x 1 KiB per wave)
//     0                                  A-h0 (8)                             C00 += A0 B0       A-h1 of tile t+1
//     1                                  B-h1 (4)                             C01 += A0 B1       B-h0 of tile t+2
//     2                                  A-h1 (8)                             C11 += A1 B1       A-h0 of tile t+2
//     3                                  B-h1 of tile t+1 (4)                 C10 += A1 B0       B-h1 of tile t+2
//   odd K-tiles mirror this in N (C01, C00, C10, C11; B-h0 <-> B-h1), so the B fragments of a K-tile's first quadrant
//   are always read one phase early into the B register set the running quadrant does not use.
//
// LDS: 2 buffers x {A-h0, A-h1, B-h0, B-h1} x 16 KiB = 128 KiB.  A "half" gathers what ONE phase reads: A-h{q} = rows
// {wr*128 + q*64 .. +64} of both wave rows, B-h{q} = columns {wc*64 + q*32 .. +32} of the four wave columns, so a slot
// has a single reading phase R (4t-1 for the first B half, 4t, 4t+1, 4t+2) and may be re-staged from the MFMA part of phase R+1 on (by then both groups have
// waited for their reads).  A phase is: fragment reads | barrier | lgkmcnt(0), two MFMAs, the phase's two DMA pieces
// (address = row base + wave-uniform tap offset, border handling by a per-row tap bit mask), six MFMAs,
// s_waitcnt vmcnt(8) | barrier.  The DMA stream runs SEVEN half-tiles ahead of the phase counter (half-tile g is issued
// in phase g-7); after the wait of phase P the four newest half-tiles (P+4..P+7) may still be in flight, P+3 has landed
// for this wave.  Because the groups are one barrier apart, group 1's wait of phase P overlaps group 0's reads of phase
// P+1: data retired in phase P is first read in phase P+2 (reads of phase R need half-tiles <= R+1, retired by phase
// R-2: issued <= R+5, in flight R+2..R+5).  Past the last K-tile the stream issues out-of-range pieces (zeros, no
// memory traffic) into dead slots so that the counts stay uniform.  Row layout inside a slot, XOR swizzle and fragment
// mapping are those of conv_igemm_bf16.hip; every output accumulates in the same order, so results are bit-identical.
#include "conv_bf16_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HT_BYTES = 16384;               // one half-tile slot: 128 rows x 128 B
constexpr int LDS_8PH = 2 * 4 * HT_BYTES;     // 128 KiB

template <int V>
using ic = std::integral_constant<int, V>;

// ABL: timing-only ablation mask of the probe build (-DRDPN6D_PROBE; results are wrong when != 0):
// 1 = no DMA inside the loop, 2 = no fragment reads, 4 = no MFMAs, 8 = no counted vmcnt wait
template <int ABL, bool STATS = false>
__global__ __launch_bounds__(512) void conv_igemm_bf16_8ph_kernel(const ConvBArgs a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // slot(A, q, buf) = smem + (buf*2 + q) * 16 KiB;  slot(B, q, buf) = smem + 64 KiB + (buf*2 + q) * 16 KiB
    const rdpn6d_conv_desc& d = a.d;
    const int nblk = a.mtiles * a.ntiles;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk;
    const int nt = logical % a.ntiles;
    const int mt = logical / a.ntiles;
    const long long m0 = (long long)mt * 256;
    const int n0 = nt * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- DMA addressing: this wave moves pieces `wave` and `wave + 8` (8 rows x 128 B each) of every half-tile
    const int prow = lane >> 3;                    // row inside the piece
    const int lr_lo = wave * 8 + prow;             // local row of piece 0 (piece 1: + 64)
    const unsigned lslot16 = (unsigned)((lane & 7) ^ ((lr_lo >> 1) & 7)) * 16u;  // (lr + 64) >> 1 has the same low 3 bits
    // per staged row: byte offset of its centre pixel (+ the lane's swizzled 16-byte slot) and one validity bit per tap
    // (image border / rows past M), so that staging a piece costs an add, a bit test and an OR inside the loop
    unsigned a_base[2][2], a_mask[2][2];
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = m0 + i * 128 + qm * 64 + lr_lo;
            const bool ok = m < a.M;
            const int mm = ok ? (int)m : 0;
            const int b = mm / a.HoWo;
            const int rem = mm - b * a.HoWo;
            const int oy = rem / d.Wo;
            const int ox = rem - oy * d.Wo;
            const int iy = oy * d.stride, ix = ox * d.stride;
            a_base[qm][i] = ((unsigned)((b * d.H + iy) * d.W + ix) * (unsigned)d.in_cs + (unsigned)d.in_co) * 2u + lslot16;
            unsigned mask = 0;
            for (int t = 0; t < d.ntaps; ++t) {
                const int dy = (int)((a.dy_pack >> (4 * t)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * t)) & 15ull) - 8;
                mask |= (ok && (unsigned)(iy + dy) < (unsigned)d.H && (unsigned)(ix + dx) < (unsigned)d.W) ? (1u << t) : 0u;
            }
            a_mask[qm][i] = mask;
        }
    unsigned w_off[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = lr_lo + 64 * i;  // local row = (wave column) * 32 + column inside the 32-wide quadrant half
            const int col = (lr >> 5) * 64 + qn * 32 + (lr & 31);
            w_off[qn][i] = (unsigned)(n0 + col) * (unsigned)a.Ktot * 2u + lslot16;
        }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);
    const int px_bytes = d.in_cs * 2;

    // Staging a half-tile is split in two: the per-lane offsets of the wave's two pieces (computed in the reading part of
    // a phase, where the wave has issue slots to spare) and the two LDS-DMA instructions (placed between the MFMAs).
    // valid = false (K-tiles past the end): every lane's offset becomes 0xFFFFFFFF, beyond num_records of either
    // descriptor - the hardware then writes zeros without touching memory
    unsigned st_off[2];
    auto addr_A = [&](auto qmc, const int tap, const int cc, const bool valid) {
        constexpr int qm = decltype(qmc)::value;
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned toff = (unsigned)((dy * d.W + dx) * px_bytes + cc * 128);  // wave-uniform
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned kill = ((a_mask[qm][i] >> tap) & 1u) - 1u;  // 0 when the tap is inside the image, else all ones
            st_off[i] = (a_base[qm][i] + toff) | kill | sel;
        }
    };
    auto addr_B = [&](auto qnc, const int tap, const int cc, const bool valid) {
        constexpr int qn = decltype(qnc)::value;
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 2u + (unsigned)cc * 128u;
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 2; ++i) st_off[i] = (w_off[qn][i] + wk) | sel;
    };
    // piece i of half-tile (isB, q) of buffer buf
    auto issue = [&](auto isBc, auto qc, const int buf, auto ic_) {
        constexpr int isB = decltype(isBc)::value, q = decltype(qc)::value, i = decltype(ic_)::value;
        unsigned char* dst = smem + isB * 4 * HT_BYTES + (buf * 2 + q) * HT_BYTES + (wave + 8 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? wsrc : xsrc, (lds_ptr_t)dst, 16, (int)st_off[i], 0, 0, 0);
    };
    auto stage_A = [&](auto qmc, const int buf, const int tap, const int cc, const bool valid) {
        addr_A(qmc, tap, cc, valid);
        issue(ic<0>{}, qmc, buf, ic<0>{});
        issue(ic<0>{}, qmc, buf, ic<1>{});
    };
    auto stage_B = [&](auto qnc, const int buf, const int tap, const int cc, const bool valid) {
        addr_B(qnc, tap, cc, valid);
        issue(ic<1>{}, qnc, buf, ic<0>{});
        issue(ic<1>{}, qnc, buf, ic<1>{});
    };

    // ---- fragment addressing
    const int frow = lane & 31;
    const int half = lane >> 5;
    unsigned fa_base[2], fa_sw[2];  // byte offset of the lane's row inside an A slot, its swizzle key
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = wr * 64 + i * 32 + frow;
        fa_base[i] = (unsigned)lr * 128u;
        fa_sw[i] = (unsigned)((lr >> 1) & 7);
    }
    const int lrb = wc * 32 + frow;
    const unsigned fb_base = (unsigned)lrb * 128u, fb_sw = (unsigned)((lrb >> 1) & 7);

    u32x4 fa[2][4], fb0[4], fb1[4];
    auto read_A = [&](const int qm, const int buf) {
        const unsigned char* slot = smem + (buf * 2 + qm) * HT_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fa[i][j] = *reinterpret_cast<const u32x4*>(slot + fa_base[i] + ((((unsigned)(2 * j + half)) ^ fa_sw[i]) << 4));
    };
    auto read_B = [&](const int qn, const int buf, u32x4 (&fb)[4]) {
        const unsigned char* slot = smem + 4 * HT_BYTES + (buf * 2 + qn) * HT_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fb[j] = *reinterpret_cast<const u32x4*>(slot + fb_base + ((((unsigned)(2 * j + half)) ^ fb_sw) << 4));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // k16 steps [j0, j1) of one quadrant: acc[qm*2 + i][qn] += A(i, j) * B(j)
    auto mma_part = [&](auto qmc, auto qnc, const u32x4 (&fb)[4], auto j0c, auto j1c) {
        constexpr int qm = decltype(qmc)::value, qn = decltype(qnc)::value;
#pragma unroll
        for (int j = decltype(j0c)::value; j < decltype(j1c)::value; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[qm * 2 + i][qn] = RD_LP_MFMA_32x32x16(fa[i][j], fb[j], acc[qm * 2 + i][qn]);
    };

    // K order: channel-chunk major, taps innermost (as conv_igemm_bf16.hip)
    const int nk = a.nk;
    int t1_tap = 0, t1_cc = 0, t2_tap = 0, t2_cc = 0;  // K-tiles t+1 and t+2 of the DMA stream
    auto advance = [&](int& tap, int& cc) {
        ++tap;
        const int wrap = tap == d.ntaps ? 1 : 0;
        tap = wrap ? 0 : tap;
        cc += wrap;
    };

    // ---- prologue: half-tiles 0..6 = all of K-tile 0 and {A-h0, B-h0, B-h1} of K-tile 1
    stage_B(ic<0>{}, 0, 0, 0, true);  // even K-tiles: B-h0, A-h0, B-h1, A-h1
    stage_A(ic<0>{}, 0, 0, 0, true);
    stage_B(ic<1>{}, 0, 0, 0, true);
    stage_A(ic<1>{}, 0, 0, 0, true);
    advance(t1_tap, t1_cc);  // K-tile 1 (odd K-tiles: B-h1, A-h0, B-h0, A-h1)
    stage_B(ic<1>{}, 1, t1_tap, t1_cc, nk > 1);
    stage_A(ic<0>{}, 1, t1_tap, t1_cc, nk > 1);
    stage_B(ic<0>{}, 1, t1_tap, t1_cc, nk > 1);
    t2_tap = t1_tap;
    t2_cc = t1_cc;
    advance(t2_tap, t2_cc);  // K-tile 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // B-h0, A-h0, B-h1 of K-tile 0 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!(ABL & 2)) read_B(0, 0, fb0);  // "phase -1": the first quadrant's B fragments
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on

    // One phase.  L part: fragment reads + the offsets of this phase's two DMA pieces; barrier; M part: 8 MFMAs at raised
    // priority with one DMA instruction after the 2nd and one after the 4th, then the counted wait; barrier.
    auto phase = [&](auto jc, auto bufc, const int t) {
        constexpr int j = decltype(jc)::value, buf = decltype(bufc)::value;
        constexpr bool RD = !(ABL & 2), DMA = !(ABL & 1), MMA = !(ABL & 4);
        // buf = parity of the K-tile.  Odd K-tiles walk the quadrants mirrored in N (C01, C00, C10, C11) so that the B
        // fragments of a K-tile's first quadrant can be read one phase early (phase 3 of the previous K-tile, whose MFMAs
        // use the other B register set): 8 / 4 / 8 / 4 fragment reads per phase instead of 12 / 4 / 8 / 0.
        // half-tile staged by this phase: j = 0: A-h1 of K-tile t+1 (other buffer); j = 1, 2, 3: first B half, A-h0, second
        // B half of K-tile t+2 (first B half = B-h0 for even, B-h1 for odd K-tiles)
        constexpr int sB = (j == 1 || j == 3) ? 1 : 0;
        constexpr int sQ = j == 0 ? 1 : (j == 2 ? 0 : (j == 1 ? buf : (buf ^ 1)));
        constexpr int sBuf = j == 0 ? (buf ^ 1) : buf;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RD) {
            if constexpr (j == 0) {
                read_A(0, buf);
            } else if constexpr (j == 1) {
                if constexpr (buf == 0) read_B(1, buf, fb1);
                else read_B(0, buf, fb0);
            } else if constexpr (j == 2) {
                read_A(1, buf);
            } else {  // first B half of the next K-tile
                if constexpr (buf == 0) read_B(1, buf ^ 1, fb1);
                else read_B(0, buf ^ 1, fb0);
            }
        }
        if constexpr (DMA) {
            if constexpr (j == 0) addr_A(ic<1>{}, t1_tap, t1_cc, t + 1 < nk);
            else if constexpr (j == 2) addr_A(ic<0>{}, t2_tap, t2_cc, t + 2 < nk);
            else addr_B(ic<sQ>{}, t2_tap, t2_cc, t + 2 < nk);
        }
        if constexpr (j == 3) {
            t1_tap = t2_tap;
            t1_cc = t2_cc;
            advance(t2_tap, t2_cc);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        constexpr int qm = (j >> 1), qn = ((j == 1 || j == 2) ? 1 : 0) ^ buf;
        const u32x4 (&fb)[4] = qn ? fb1 : fb0;
        if constexpr (MMA) mma_part(ic<qm>{}, ic<qn>{}, fb, ic<0>{}, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DMA) issue(ic<sB>{}, ic<sQ>{}, sBuf, ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MMA) mma_part(ic<qm>{}, ic<qn>{}, fb, ic<1>{}, ic<2>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DMA) issue(ic<sB>{}, ic<sQ>{}, sBuf, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MMA) mma_part(ic<qm>{}, ic<qn>{}, fb, ic<2>{}, ic<4>{});
        __builtin_amdgcn_s_setprio(0);
        if constexpr (!(ABL & 8) && DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    for (int t = 0; t < nk; t += 2) {
        phase(ic<0>{}, ic<0>{}, t);
        phase(ic<1>{}, ic<0>{}, t);
        phase(ic<2>{}, ic<0>{}, t);
        phase(ic<3>{}, ic<0>{}, t);
        phase(ic<0>{}, ic<1>{}, t + 1);
        phase(ic<1>{}, ic<1>{}, t + 1);
        phase(ic<2>{}, ic<1>{}, t + 1);
        phase(ic<3>{}, ic<1>{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing in LDS (the tail pieces are zeros)
    if (wr == 0) __builtin_amdgcn_s_barrier();        // re-align the two groups

    if constexpr (ABL & 16) {  // probe: no epilogue (one dependent store keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 123.456f) a.d.y[0] = sum;
    } else {
        // full column tiles and aligned channel slices are part of the eligibility: only the coalesced epilogue is needed
        conv_bf16_epilogue_vec<256, 256, 2, 4, 4, 2, true, STATS>(a, acc, smem, m0, n0, wave, lane, wr, wc);
    }
}

}  // namespace

// true when conv_igemm_bf16_8ph_kernel can run this problem (the caller decides whether it should)
bool conv_bf16_8ph_eligible(const ConvBArgs& a, int rb)
{
    return rb == 128 && a.d.Npad % 256 == 0 && a.d.N == a.d.Npad && a.vec_out && a.nk >= 2 && (a.nk & 1) == 0;
}

template <int ABL>
static int launch_8ph(const ConvBArgs& a, hipStream_t s)
{
    if constexpr (ABL == 0) {
        if (a.stats) {  // training: the instantiation whose epilogue writes BatchNorm partial sums
            auto kern = conv_igemm_bf16_8ph_kernel<0, true>;
            RD_LDS_OPT_IN(kern, LDS_8PH);
            hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), LDS_8PH, s, a);
            return RDPN6D_OK;
        }
    }
    auto kern = conv_igemm_bf16_8ph_kernel<ABL>;
    RD_LDS_OPT_IN(kern, LDS_8PH);
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), LDS_8PH, s, a);
    return RDPN6D_OK;
}

int conv_bf16_launch_8ph(const ConvBArgs& a, hipStream_t s)
{
#ifdef RDPN6D_PROBE
    static const int abl = getenv("RDPN6D_ABL") ? atoi(getenv("RDPN6D_ABL")) : 0;
    switch (abl) {
    case 1: return launch_8ph<1>(a, s);
    case 2: return launch_8ph<2>(a, s);
    case 3: return launch_8ph<3>(a, s);
    case 4: return launch_8ph<4>(a, s);
    case 5: return launch_8ph<5>(a, s);
    case 6: return launch_8ph<6>(a, s);
    case 7: return launch_8ph<7>(a, s);
    case 8: return launch_8ph<8>(a, s);
    case 16: return launch_8ph<16>(a, s);
    case 23: return launch_8ph<23>(a, s);
    default: break;
    }
#endif
    return launch_8ph<0>(a, s);
}
