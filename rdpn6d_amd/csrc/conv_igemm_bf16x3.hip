// fp32-accurate implicit-GEMM convolution on the bf16 matrix pipe ("bf16x3"): every fp32 operand is held as three bf16
// terms  a = a1 + a2 + a3  (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2); 3 x 8 significand bits with
// round-to-nearest steps represent a 24-bit fp32 significand to 2^-27) and a product is evaluated as the six partial
// products a_i * b_j with i + j <= 4, each exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped
// terms (a2 b3, a3 b2, a3 b3) are <= 2^-26 relative - below one fp32 rounding - so the result carries fp32 accuracy
// (tests: error against fp64 no larger than the fp32-MFMA kernel's) while the MFMA work runs at the bf16 rate:
// 6 bf16 MFMA flops per algorithmic flop = 2.67x the fp32 matrix pipe's peak (2500 / 6 = 416.7 vs 157.3 TFLOP/s).
//
// Schedule = the 256x256 8-phase ping-pong of conv_igemm_bf16_8ph.hip (two wave groups one barrier apart, counted vmcnt,
// LDS-DMA stream seven half-tiles ahead), re-dimensioned: a K-tile is 16 channels (one k16 MFMA step) x 3 planes, so a
// half-tile slot is 3 planes x 128 rows x 32 B = 12 KiB and 2 buffers x 4 half-tiles = 96 KiB of LDS; a phase multiplies
// one 64x32 quadrant with 2 m-tiles x 6 plane pairs = 12 MFMAs (384 cycles) from 6 + 3 fragment reads - half the LDS and
// DMA bytes per MFMA of the plain bf16 kernel.  Operands live in HBM as three bf16 planes ([plane][B,H,W,C] activations
// written by rdpn6d_split_bf16x3 or by this kernel's own epilogue, [plane][Npad][ntaps][Cin] weights).
// A half-tile is 12 one-KiB DMA pieces (3 planes x 4 blocks of 32 rows): wave w moves piece (plane w>>2, block w&3) and,
// for w < 4, piece (plane 2, block w) - group 0 issues two pieces per phase and waits with vmcnt(8), group 1 one and
// vmcnt(4); both keep the four newest half-tiles in flight.
#include "conv_x3_args.h"

#include <type_traits>

namespace {

constexpr int X3_HT_BYTES = 3 * 128 * 32;           // half-tile slot: 3 planes x 128 rows x 32 B
constexpr int X3_STAGE_BYTES = 2 * 4 * X3_HT_BYTES; // 96 KiB
constexpr int X3_EPI_BYTES = 8 * 32 * (64 + 8) * 4; // the coalesced epilogue's transpose slices (re-uses the staging LDS)
constexpr int X3_LDS = X3_STAGE_BYTES > X3_EPI_BYTES ? X3_STAGE_BYTES : X3_EPI_BYTES;

template <int V>
using ic = std::integral_constant<int, V>;

__global__ __launch_bounds__(512) void conv_igemm_bf16x3_kernel(const ConvX3Args ax)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // slot(A, q, buf) = smem + (buf*2 + q) * 12 KiB;  slot(B, q, buf) = smem + 48 KiB + (buf*2 + q) * 12 KiB
    const ConvBArgs& a = ax.b;
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

    // ---- DMA addressing.  A piece is 32 rows x 32 B; lane -> row lane>>1, 16-byte slot lane&1 (XOR-swizzled by bit 3 of
    // the row).  Both pieces of a wave cover the SAME 32 rows (block w&3) of different planes: one row address per lane.
    const int blk = wave & 3;
    const int lr = blk * 32 + (lane >> 1);  // local row inside a half-tile
    const unsigned lslot16 = (unsigned)((lane & 1) ^ ((lr >> 3) & 1)) * 16u;
    const unsigned pl0 = (unsigned)(wave >> 2);  // plane of the first piece (0 or 1); second piece (waves 0-3): plane 2
    unsigned a_base[2], a_mask[2], w_off[2];
#pragma unroll
    for (int qm = 0; qm < 2; ++qm) {
        const long long m = m0 + (lr >> 6) * 128 + qm * 64 + (lr & 63);
        const bool ok = m < a.M;
        const int mm = ok ? (int)m : 0;
        const int b = mm / a.HoWo;
        const int rem = mm - b * a.HoWo;
        const int oy = rem / d.Wo;
        const int ox = rem - oy * d.Wo;
        const int iy = oy * d.stride, ix = ox * d.stride;
        a_base[qm] = ((unsigned)((b * d.H + iy) * d.W + ix) * (unsigned)d.in_cs + (unsigned)d.in_co) * 2u + lslot16;
        unsigned mask = 0;
        for (int t = 0; t < d.ntaps; ++t) {
            const int dy = (int)((a.dy_pack >> (4 * t)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * t)) & 15ull) - 8;
            mask |= (ok && (unsigned)(iy + dy) < (unsigned)d.H && (unsigned)(ix + dx) < (unsigned)d.W) ? (1u << t) : 0u;
        }
        a_mask[qm] = mask;
    }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        const int col = (lr >> 5) * 64 + qn * 32 + (lr & 31);
        w_off[qn] = (unsigned)(n0 + col) * (unsigned)a.Ktot * 2u + lslot16;
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);
    const int px_bytes = d.in_cs * 2;
    const unsigned xp0 = pl0 * ax.x_plane_bytes, xp2 = 2u * ax.x_plane_bytes;
    const unsigned wp0 = pl0 * ax.w_plane_bytes, wp2 = 2u * ax.w_plane_bytes;

    // offsets of this phase's piece(s): st_off = plane-0-relative offset of the lane's 16 bytes (all ones = out of range)
    unsigned st_off;
    auto addr_A = [&](auto qmc, const int tap, const int cc, const bool valid) {
        constexpr int qm = decltype(qmc)::value;
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned toff = (unsigned)((dy * d.W + dx) * px_bytes + cc * 32);  // wave-uniform
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;
        const unsigned kill = ((a_mask[qm] >> tap) & 1u) - 1u;
        st_off = (a_base[qm] + toff) | kill | sel;
    };
    auto addr_B = [&](auto qnc, const int tap, const int cc, const bool valid) {
        constexpr int qn = decltype(qnc)::value;
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 2u + (unsigned)cc * 32u;
        st_off = (w_off[qn] + wk) | (valid ? 0u : 0xFFFFFFFFu);
    };
    // piece `second` (0: plane pl0, 1: plane 2 - waves 0..3 only) of half-tile (isB, q) of buffer buf.  The plane offset is
    // added as the scalar offset of the buffer instruction, so an all-ones (out-of-range) lane offset stays out of range.
    auto issue = [&](auto isBc, auto qc, const int buf, auto secondc) {
        constexpr int isB = decltype(isBc)::value, q = decltype(qc)::value, second = decltype(secondc)::value;
        const unsigned plane = second ? 2u : pl0;
        unsigned char* dst = smem + isB * 4 * X3_HT_BYTES + (buf * 2 + q) * X3_HT_BYTES + plane * 4096u + blk * 1024;
        const unsigned soff = isB ? (second ? wp2 : wp0) : (second ? xp2 : xp0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? wsrc : xsrc, (lds_ptr_t)dst, 16, (int)st_off, (int)soff, 0, 0);
    };
    auto stage = [&](auto isBc, auto qc, const int buf, const int tap, const int cc, const bool valid) {
        if constexpr (decltype(isBc)::value) addr_B(qc, tap, cc, valid);
        else addr_A(qc, tap, cc, valid);
        issue(isBc, qc, buf, ic<0>{});
        if (wr == 0) issue(isBc, qc, buf, ic<1>{});
    };

    // ---- fragment addressing: lane (row frow, k-half) reads 16 bytes = 8 channels of one plane
    const int frow = lane & 31;
    const int half = lane >> 5;
    unsigned fa_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wr * 64 + i * 32 + frow;
        fa_off[i] = (unsigned)r * 32u + (unsigned)((half ^ ((r >> 3) & 1)) << 4);
    }
    const int rb = wc * 32 + frow;
    const unsigned fb_off = (unsigned)rb * 32u + (unsigned)((half ^ ((rb >> 3) & 1)) << 4);

    u32x4 fa[2][3], fb0[3], fb1[3];  // [m-tile][plane], [plane]
    auto read_A = [&](const int qm, const int buf) {
        const unsigned char* slot = smem + (buf * 2 + qm) * X3_HT_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const u32x4*>(slot + p * 4096 + fa_off[i]);
    };
    auto read_B = [&](const int qn, const int buf, u32x4 (&fb)[3]) {
        const unsigned char* slot = smem + 4 * X3_HT_BYTES + (buf * 2 + qn) * X3_HT_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[p] = *reinterpret_cast<const u32x4*>(slot + p * 4096 + fb_off);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // plane pairs (a_p, b_q), p + q <= 2 (0-based), smallest terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0);
    // pairs [p0, p1) of one quadrant, both m-tiles
    auto mma_part = [&](auto qmc, auto qnc, const u32x4 (&fb)[3], auto p0c, auto p1c) {
        constexpr int qm = decltype(qmc)::value, qn = decltype(qnc)::value;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int pr = decltype(p0c)::value; pr < decltype(p1c)::value; ++pr)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[qm * 2 + i][qn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][PA[pr]]),
                                                                              __builtin_bit_cast(bf16x8, fb[PB[pr]]),
                                                                              acc[qm * 2 + i][qn], 0, 0, 0);
    };

    // K order: channel-chunk major, taps innermost
    const int nk = a.nk;
    int t1_tap = 0, t1_cc = 0, t2_tap = 0, t2_cc = 0;  // K-tiles t+1 and t+2 of the DMA stream
    auto advance = [&](int& tap, int& cc) {
        ++tap;
        const int wrap = tap == d.ntaps ? 1 : 0;
        tap = wrap ? 0 : tap;
        cc += wrap;
    };

    // ---- prologue: half-tiles 0..6 (A-h0, B-h0, B-h1, A-h1 of K-tile 0; A-h0, B-h0, B-h1 of K-tile 1)
    stage(ic<0>{}, ic<0>{}, 0, 0, 0, true);
    stage(ic<1>{}, ic<0>{}, 0, 0, 0, true);
    stage(ic<1>{}, ic<1>{}, 0, 0, 0, true);
    stage(ic<0>{}, ic<1>{}, 0, 0, 0, true);
    advance(t1_tap, t1_cc);
    stage(ic<0>{}, ic<0>{}, 1, t1_tap, t1_cc, nk > 1);
    stage(ic<1>{}, ic<0>{}, 1, t1_tap, t1_cc, nk > 1);
    stage(ic<1>{}, ic<1>{}, 1, t1_tap, t1_cc, nk > 1);
    t2_tap = t1_tap;
    t2_cc = t1_cc;
    advance(t2_tap, t2_cc);
    if (wr == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the first three half-tiles have landed (own pieces)
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on

    // One phase (see conv_igemm_bf16_8ph.hip): reads + offsets | barrier | 12 MFMAs with the DMA piece(s) in their shadow,
    // counted wait | barrier.  Half-tile staged by phase j: 0: A-h1 of tile t+1 (other buffer); 1: A-h0, 2: B-h0, 3: B-h1
    // of tile t+2.
    auto phase = [&](auto jc, auto bufc, const int t) {
        constexpr int j = decltype(jc)::value, buf = decltype(bufc)::value;
        constexpr int sB = j >= 2 ? 1 : 0, sQ = (j == 0 || j == 3) ? 1 : 0, sBuf = j == 0 ? (buf ^ 1) : buf;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j == 0) {
            read_B(0, buf, fb0);
            read_A(0, buf);
        } else if constexpr (j == 1) {
            read_B(1, buf, fb1);
        } else if constexpr (j == 2) {
            read_A(1, buf);
        }
        if constexpr (j == 0) addr_A(ic<1>{}, t1_tap, t1_cc, t + 1 < nk);
        else if constexpr (j == 1) addr_A(ic<0>{}, t2_tap, t2_cc, t + 2 < nk);
        else if constexpr (j == 2) addr_B(ic<0>{}, t2_tap, t2_cc, t + 2 < nk);
        else addr_B(ic<1>{}, t2_tap, t2_cc, t + 2 < nk);
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
        constexpr int qm = (j >> 1), qn = (j == 1 || j == 2) ? 1 : 0;
        const u32x4 (&fb)[3] = qn ? fb1 : fb0;
        mma_part(ic<qm>{}, ic<qn>{}, fb, ic<0>{}, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        issue(ic<sB>{}, ic<sQ>{}, sBuf, ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_part(ic<qm>{}, ic<qn>{}, fb, ic<1>{}, ic<2>{});
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 0) issue(ic<sB>{}, ic<sQ>{}, sBuf, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_part(ic<qm>{}, ic<qn>{}, fb, ic<2>{}, ic<6>{});
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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

    // ---- epilogue: scale/shift (+ fp32 residual), activation; fp32 store and / or the three bf16 planes of the result
    {
        const int hi = lane >> 5;
        constexpr int CS = 64 + 8;
        __syncthreads();
        float* cst = reinterpret_cast<float*>(smem) + wave * (32 * CS);
        const int nb = n0 + wc * 64;
        auto pixel_of = [&](const long long m) -> long long {
            if (a.linear_out) return m;
            const int mm = (int)m;
            const int b = mm / a.HoWo;
            const int rem = mm - b * a.HoWo;
            const int oy = rem / d.Wo;
            const int ox = rem - oy * d.Wo;
            return ((long long)b * d.OH + (oy * d.osy + d.ooy)) * d.OW + (ox * d.osx + d.oox);
        };
        float scj[2], shj[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb + j * 32 + frow;
            scj[j] = d.scale ? d.scale[n] : 1.f;
            shj[j] = d.shift ? d.shift[n] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e] * scj[j] + shj[j];
            // 8 channels per lane, 8 lanes per row, 8 rows per pass (two 16-byte fp32 stores / one 16-byte store per plane)
            const int rrow = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = rr * 8 + rrow;
                const long long mrow = m0 + wr * 128 + i * 32 + row;
                if (mrow >= a.M) continue;
                const long long pix = pixel_of(mrow);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                x3_finish_row8(ax, v, pix, nb + c8);
            }
        }
    }
}

// x [n] fp32 -> planes [3][plane_elems] bf16 (plane_elems >= n)
__global__ void split_bf16x3_kernel(const float* __restrict__ x, long long n, bf16_t* __restrict__ planes, long long plane_elems)
{
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        bf16_t t[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rd_split3(v[q], t[0][q], t[1][q], t[2][q]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint2 u;
            u.x = (unsigned)t[p][0] | ((unsigned)t[p][1] << 16);
            u.y = (unsigned)t[p][2] | ((unsigned)t[p][3] << 16);
            *reinterpret_cast<uint2*>(planes + p * plane_elems + 4 * i) = u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 * 4; i < n; ++i) rd_split3(x[i], planes[i], planes[plane_elems + i], planes[2 * plane_elems + i]);
}

}  // namespace

extern "C" int rdpn6d_split_bf16x3(const float* x, long long n, void* planes, long long plane_elems, void* stream)
{
    RD_REQUIRE(x && planes && n > 0 && plane_elems >= n && plane_elems % 8 == 0, "null pointer / plane size");
    const long long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks))), dim3(256), 0,
                       (hipStream_t)stream, x, n, (bf16_t*)planes, plane_elems);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// conv_igemm_bf16x3_tile.hip: 128x128 .. 64x64 tiles for the narrow / short layers
void conv_x3_pick_tile(const rdpn6d_conv_desc* d, long long M, int* pbm, int* pbn);
int conv_x3_launch_tile(ConvX3Args& ax, int bm, int bn, hipStream_t s);

static bool x3_common_ok(const rdpn6d_conv_desc* d)
{
    return d->Cin % 16 == 0 && d->in_cs % 8 == 0 && d->in_co % 8 == 0 && d->out_cs % 8 == 0 && d->out_co % 8 == 0 &&
           (!d->res || (d->res_cs % 4 == 0 && d->res_co % 4 == 0));
}
// the 256x256 8-phase kernel can run the layer
static bool x3_big_ok(const rdpn6d_conv_desc* d)
{
    const int nk = d->ntaps * (d->Cin / 16);
    return x3_common_ok(d) && d->Npad % 256 == 0 && d->N == d->Npad && nk >= 2 && (nk & 1) == 0;
}
// ... and should: at least 160 tiles and a last round of tiles (one per CU) that is not mostly empty
static bool x3_big_pays(const rdpn6d_conv_desc* d, long long M)
{
    const long long tiles = (long long)rd_cdiv(M, 256) * (d->Npad / 256);
    const long long rounds = (tiles + 255) / 256;
    return tiles >= 160 && (double)tiles >= 0.62 * 256.0 * (double)rounds;
}

// which bf16x3 kernel rdpn6d_conv2d_bf16x3[_ex] would use: 2 = 256x256 8-phase, 1 = 128x128..64x64 tile kernel, 0 = none
extern "C" int rdpn6d_conv_bf16x3_kernel_for(const rdpn6d_conv_desc* d)
{
    if (!d || !x3_common_ok(d)) return 0;
    const long long M = (long long)d->B * d->Ho * d->Wo;
    if (x3_big_ok(d) && x3_big_pays(d, M)) return 2;
    int bm, bn;
    conv_x3_pick_tile(d, M, &bm, &bn);
    if (bm) return 1;
    return x3_big_ok(d) ? 2 : 0;
}
extern "C" int rdpn6d_conv_bf16x3_eligible(const rdpn6d_conv_desc* d) { return rdpn6d_conv_bf16x3_kernel_for(d) != 0; }

extern "C" int rdpn6d_conv2d_bf16x3_ex(const rdpn6d_conv_desc* d, long long x_plane_elems, long long w_plane_elems, void* y_planes,
                                       long long y_plane_elems, const void* res_planes, long long res_plane_elems, void* stream)
{
    RD_REQUIRE(d && d->x && d->w && (d->y || y_planes), "null pointer");
    const int which = rdpn6d_conv_bf16x3_kernel_for(d);
    RD_REQUIRE(which != 0, "bf16x3 needs Cin % 16 == 0 (% 32 for 64-wide tiles), Npad % 64 == 0, N % 8 == 0, 16-byte aligned slices");
    RD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->ntaps >= 1 && d->ntaps <= 9, "shape");
    RD_REQUIRE(d->in_co + d->Cin <= d->in_cs && d->out_co + d->N <= d->out_cs, "channel slices");
    RD_REQUIRE((d->Ho - 1) * d->osy + d->ooy < d->OH && (d->Wo - 1) * d->osx + d->oox < d->OW, "output geometry");
    RD_REQUIRE(!(d->res && res_planes), "residual either as an fp32 tensor or as planes");
    RD_REQUIRE(!res_planes || (d->res_cs % 8 == 0 && d->res_co % 8 == 0 && d->res_co + d->N <= d->res_cs), "residual planes slice");
    ConvX3Args ax;
    ConvBArgs& a = ax.b;
    a.d = *d;
    a.M = (long long)d->B * d->Ho * d->Wo;
    RD_REQUIRE(a.M < (1LL << 31), "B*Ho*Wo must fit 31 bits");
    a.HoWo = d->Ho * d->Wo;
    a.cchunks = d->Cin / 16;
    a.nk = d->ntaps * a.cchunks;
    a.Ktot = d->ntaps * d->Cin;
    a.linear_out = (d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo);
    a.out_f32 = 1;
    a.vec_out = 1;
    const long long in_elems = (long long)d->B * d->H * d->W * d->in_cs;
    RD_REQUIRE(x_plane_elems >= in_elems && (2 * x_plane_elems + in_elems) * 2 < (1LL << 32) - 64, "activation planes (32-bit offsets)");
    const long long w_elems = (long long)d->Npad * d->ntaps * d->Cin;
    RD_REQUIRE(w_plane_elems >= w_elems && (2 * w_plane_elems + w_elems) * 2 < (1LL << 32) - 64, "weight planes (32-bit offsets)");
    RD_REQUIRE(!y_planes || y_plane_elems >= (long long)d->B * d->OH * d->OW * d->out_cs, "output planes");
    RD_REQUIRE(!res_planes || res_plane_elems >= (long long)d->B * d->OH * d->OW * d->res_cs, "residual planes");
    a.x_bytes = (unsigned)((2 * x_plane_elems + in_elems) * 2);
    a.w_bytes = (unsigned)((2 * w_plane_elems + w_elems) * 2);
    ax.x_plane_bytes = (unsigned)(x_plane_elems * 2);
    ax.w_plane_bytes = (unsigned)(w_plane_elems * 2);
    ax.y_planes = y_planes;
    ax.y_plane_elems = y_plane_elems;
    ax.res_planes = res_planes;
    ax.res_plane_elems = res_plane_elems;
    a.dy_pack = a.dx_pack = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        RD_REQUIRE(d->dy[t] >= -8 && d->dy[t] <= 7 && d->dx[t] >= -8 && d->dx[t] <= 7, "tap offsets must be in -8..7");
        a.dy_pack |= (unsigned long long)(d->dy[t] + 8) << (4 * t);
        a.dx_pack |= (unsigned long long)(d->dx[t] + 8) << (4 * t);
    }
    a.kper = 0;
    a.partial = nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (which == 1) {
        int bm, bn;
        conv_x3_pick_tile(d, a.M, &bm, &bn);
        const int rc = conv_x3_launch_tile(ax, bm, bn, s);
        if (rc != RDPN6D_OK) return rc;
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    a.mtiles = rd_cdiv(a.M, 256);
    a.ntiles = d->Npad / 256;
    a.kper = a.nk;
    RD_LDS_OPT_IN(conv_igemm_bf16x3_kernel, X3_LDS);
    hipLaunchKernelGGL(conv_igemm_bf16x3_kernel, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), X3_LDS, s, ax);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_conv2d_bf16x3(const rdpn6d_conv_desc* d, long long x_plane_elems, long long w_plane_elems, void* y_planes,
                                    long long y_plane_elems, void* stream)
{
    return rdpn6d_conv2d_bf16x3_ex(d, x_plane_elems, w_plane_elems, y_planes, y_plane_elems, nullptr, 0, stream);
}
