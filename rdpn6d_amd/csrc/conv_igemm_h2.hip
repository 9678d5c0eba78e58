// fp32-ACCURATE implicit-GEMM convolution on the fp16 matrix pipe, TWO planes per operand ("h2").
//
// Every fp32 operand is held as two fp16 terms of its value scaled by a power of two:
//     a * 2^s = hi + lo,   hi = fp16(a * 2^s),   lo = fp16(a * 2^s - hi)            (both round-to-nearest-even)
// |a 2^s - hi - lo| <= 2^-22 |a 2^s| (11 + 11 significand bits; 2^-25 absolute once lo is subnormal - fp16 MFMA inputs keep
// their subnormals), and a product is evaluated as  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  - three exact fp32 products (11 x 11
// bits) accumulated in fp32 by v_mfma_f32_32x32x16_f16.  The dropped lo*lo term is <= 2^-22 relative.  Per product this is
// 2-4x an fp32 rounding, but a K-long reduction rounds its fp32 ACCUMULATOR at every step, and with HALF the accumulation
// steps of the three-plane bf16 scheme (conv_igemm_bf16x3.hip: six partial products) the sum ends up CLOSER to the exact
// result than both that scheme and the fp32-MFMA kernel (tests: error vs an fp64 convolution <= the fp32-MFMA kernel's on
// every shape, including operands spread over 2^-10..2^10 with >= 100x cancellation).  3 fp16 MFMA flops per algorithmic
// flop => a ceiling of 2500 / 3 = 833 TFLOP/s for fp32-accurate work (bf16x3: 416.7, fp32 MFMA pipe: 157.3).
//
// Range.  fp16 ends at 65504: activations are stored as a * 16 (|a| < 4094; lo stays a normal number down to |a| = 2^-7 and
// the representation error is <= max(2^-22 |a|, 1.9e-9) below), weights per output channel as w * 2^sw with max |w 2^sw| in
// [2^13, 2^14); the host folds 2^-(sw+4) into the epilogue's per-channel scale (exact).  An activation beyond the range is
// CLAMPED and reported through `overflow_flag` (the host then raises / falls back to bf16x3) - never an inf.
//
// Layout ("h2 tensor"): per pixel and per group of 32 channels one 128-byte record [hi x 32 | lo x 32] fp16, i.e.
// [pixels][C / 32][2][32] - the byte size of the fp32 tensor.  A 128-byte LDS row is therefore exactly the 64-"channel" row
// of the plain bf16 kernels (conv_igemm_bf16.hip / conv_igemm_bf16_8ph.hip): staging, swizzle, DMA stream and both schedules
// (2x2-wave tile kernel; 256x256 eight-phase ping-pong with the two wave groups one barrier apart) are theirs unchanged.
// What differs is the inner product: of a row's four 16-byte slot pairs j (j = 0,1: hi of channels 0-15 / 16-31; j = 2,3:
// lo) a k16 step s takes  A[2+s]*B[s] + A[s]*B[2+s] + A[s]*B[s]  - 6 MFMAs per 32x32 tile pair and row instead of 4, from
// the same fragment reads - and the epilogue, which writes fp32 and / or the h2 record of the result.
#include "conv_h2_common.h"

#include <cstdio>
#include <cstdlib>

#ifdef RDPN6D_PROBE
// probe build only (RDPN6D_PROBE=1 python -m rdpn6d_amd.build; tools/probe_h2_tile.py): per-wave cycle sums of the phases of a
// tile-kernel step, written by the SCHED = 9 variant (RDPN6D_H2_SCHED=9)
__device__ unsigned long long* g_h2_probe = nullptr;
extern "C" int rdpn6d_debug_h2_probe(void* buf)
{
    RD_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_h2_probe), &buf, sizeof(buf)));
    return RDPN6D_OK;
}
#endif

namespace {

// ============================================================================================ 256x256 eight-phase kernel
constexpr int HT_BYTES = 16384;            // one half-tile slot: 128 rows x 128 B
constexpr int LDS_8PH = 2 * 4 * HT_BYTES;  // 128 KiB
constexpr int LDS_8PH_FUSED = 8 * (32 * 72 * 4 + 32 * 272);  // 140 KiB: the fused-1x1 epilogue's per-wave transposition + A-fragment slices
static_assert(LDS_8PH_FUSED >= LDS_8PH && LDS_8PH_FUSED <= 160 * 1024, "LDS");

// Schedule, LDS map, DMA stream and counted waits: see conv_igemm_bf16_8ph.hip (this is that kernel with the h2 inner
// product - 12 MFMAs per phase instead of 8 - and the h2 epilogue).  A K-tile = one 128-byte row = 32 channels (hi | lo).
// CMAX (round 5): the COLUMN-MAX form for a layer whose output only a per-crop channel max reads (the point-wise branch's last
// convolution + BatchNorm under the exact rewrites of gdrn.py: its 134-MB h2 output was written by this kernel and read back by
// global_max_h2_kernel - 0.05 ms of a 6.7-ms step for nothing).  The epilogue takes scale * acc + shift exactly as the plain one, makes
// the h2 record of every value, and keeps per column the record of the largest reconstructed value of the workgroup's 256 rows (all
// of one crop: rows_per_group % 256 == 0) as a 64-bit key [ordered float bits of hi + lo | hi bits | lo bits]; one atomic max per
// (wave, column) merges the keys of a crop's workgroups - max is order-independent, the result is deterministic.  The activation is
// never written.  ax.fuse_out = the key table [groups][Npad] (zero = empty), ax.fuse_cs = rows per group.
template <bool FUSE, bool CMAX>
__device__ __forceinline__ void conv_h2_8ph_body(const ConvH2Args& ax)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
#ifdef RDPN6D_PROBE
    const unsigned long long pq_start = __builtin_readcyclecounter();
    const unsigned long long pq_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
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

    // ---- DMA addressing: this wave moves pieces `wave` and `wave + 8` (8 rows x 128 B each) of every half-tile
    const int prow = lane >> 3;
    const int lr_lo = wave * 8 + prow;
    const unsigned lslot16 = (unsigned)((lane & 7) ^ ((lr_lo >> 1) & 7)) * 16u;
    const unsigned px_bytes = (unsigned)d.in_cs * 4u;  // an h2 pixel is 2 x in_cs halfs
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
            a_base[qm][i] = (unsigned)((b * d.H + iy) * d.W + ix) * px_bytes + (unsigned)d.in_co * 4u + lslot16;
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
            const int lr = lr_lo + 64 * i;
            const int col = (lr >> 5) * 64 + qn * 32 + (lr & 31);
            w_off[qn][i] = (unsigned)(n0 + col) * (unsigned)a.Ktot * 4u + lslot16;
        }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);

    unsigned st_off[2];
    auto addr_A = [&](auto qmc, const int tap, const int cc, const bool valid) {
        constexpr int qm = decltype(qmc)::value;
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
#ifdef RDPN6D_PROBE
        // nsplit = 102: every tap but the first re-reads the tile's first chunk (centre rows, channels 0..31): real, non-zero data from
        // the CU's own L1 instead of L2 - separates the L2 traffic from the zero-operand effect of mode 101 on the matrix pipe's power
        const unsigned toff = (ax.nsplit == 102 && tap != 0) ? 0u : (unsigned)((dy * d.W + dx) * (int)px_bytes + cc * 128);
#else
        const unsigned toff = (unsigned)((dy * d.W + dx) * (int)px_bytes + cc * 128);  // wave-uniform
#endif
#ifdef RDPN6D_PROBE
        // timing-only ablation (wrong results): nsplit = 101 -> the activation half-tiles of every tap but the first read out of range
        // (the DMA instruction is still issued, no L2 traffic): what a halo-resident patch (A staged once per 32 channels) could save
        const unsigned sel = (valid && !(ax.nsplit == 101 && tap != 0)) ? 0u : 0xFFFFFFFFu;
#else
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned kill = ((a_mask[qm][i] >> tap) & 1u) - 1u;
            st_off[i] = (a_base[qm][i] + toff) | kill | sel;
        }
    };
    auto addr_B = [&](auto qnc, const int tap, const int cc, const bool valid) {
        constexpr int qn = decltype(qnc)::value;
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 4u + (unsigned)cc * 128u;
        const unsigned sel = valid ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 2; ++i) st_off[i] = (w_off[qn][i] + wk) | sel;
    };
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
    unsigned fa_base[2], fa_sw[2];
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

    // folded BatchNorm scale / shift of this lane's two channels, requested before the K loop (in the epilogue their L2 round trip was exposed)
    const int nb = n0 + wc * 64;
    float scj[2], shj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nb + j * 32 + frow;
        scj[j] = d.scale ? d.scale[n] : 1.f;
        shj[j] = d.shift ? d.shift[n] : 0.f;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // partial products [p0, p1) of one quadrant, both m-tiles
    auto mma_part = [&](auto qmc, auto qnc, const u32x4 (&fb)[4], auto p0c, auto p1c) {
        constexpr int qm = decltype(qmc)::value, qn = decltype(qnc)::value;
        H2_PAIRS;
#pragma unroll
        for (int pr = decltype(p0c)::value; pr < decltype(p1c)::value; ++pr)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[qm * 2 + i][qn] = h2_mfma(fa[i][H2_PA[pr]], fb[H2_PB[pr]], acc[qm * 2 + i][qn]);
    };

    const int nk = a.nk;
    int t1_tap = 0, t1_cc = 0, t2_tap = 0, t2_cc = 0;
    auto advance = [&](int& tap, int& cc) {
        ++tap;
        const int wrap = tap == d.ntaps ? 1 : 0;
        tap = wrap ? 0 : tap;
        cc += wrap;
    };

    // ---- prologue: half-tiles 0..6 = all of K-tile 0 and {B-h1, A-h0, B-h0} of K-tile 1
    stage_B(ic<0>{}, 0, 0, 0, true);
    stage_A(ic<0>{}, 0, 0, 0, true);
    stage_B(ic<1>{}, 0, 0, 0, true);
    stage_A(ic<1>{}, 0, 0, 0, true);
    advance(t1_tap, t1_cc);
    stage_B(ic<1>{}, 1, t1_tap, t1_cc, nk > 1);
    stage_A(ic<0>{}, 1, t1_tap, t1_cc, nk > 1);
    stage_B(ic<0>{}, 1, t1_tap, t1_cc, nk > 1);
    t2_tap = t1_tap;
    t2_cc = t1_cc;
    advance(t2_tap, t2_cc);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_B(0, 0, fb0);
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on

    auto phase = [&](auto jc, auto bufc, const int t) {
        constexpr int j = decltype(jc)::value, buf = decltype(bufc)::value;
        constexpr int sB = (j == 1 || j == 3) ? 1 : 0;
        constexpr int sQ = j == 0 ? 1 : (j == 2 ? 0 : (j == 1 ? buf : (buf ^ 1)));
        constexpr int sBuf = j == 0 ? (buf ^ 1) : buf;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j == 0) {
            read_A(0, buf);
        } else if constexpr (j == 1) {
            if constexpr (buf == 0) read_B(1, buf, fb1);
            else read_B(0, buf, fb0);
        } else if constexpr (j == 2) {
            read_A(1, buf);
        } else {
            if constexpr (buf == 0) read_B(1, buf ^ 1, fb1);
            else read_B(0, buf ^ 1, fb0);
        }
        if constexpr (j == 0) addr_A(ic<1>{}, t1_tap, t1_cc, t + 1 < nk);
        else if constexpr (j == 2) addr_A(ic<0>{}, t2_tap, t2_cc, t + 2 < nk);
        else addr_B(ic<sQ>{}, t2_tap, t2_cc, t + 2 < nk);
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
        mma_part(ic<qm>{}, ic<qn>{}, fb, ic<0>{}, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        issue(ic<sB>{}, ic<sQ>{}, sBuf, ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_part(ic<qm>{}, ic<qn>{}, fb, ic<1>{}, ic<2>{});
        __builtin_amdgcn_sched_barrier(0);
        issue(ic<sB>{}, ic<sQ>{}, sBuf, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_part(ic<qm>{}, ic<qn>{}, fb, ic<2>{}, ic<6>{});
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#ifdef RDPN6D_PROBE
    const unsigned long long pq_loop0 = __builtin_readcyclecounter();
#endif
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
#ifdef RDPN6D_PROBE
    const unsigned long long pq_loop1 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups

    if constexpr (FUSE) {
        // ---- fused epilogue: activation tile x 1x1 output weights.  Per wave and 32-row block: (1) transpose the block through the
        // wave's LDS slice, folded BatchNorm + ReLU, h2 split; (2) the h2 records go into a second slice in MFMA A-fragment order
        // (row stride 272 B: conflict-free 16-byte reads); (3) 24 MFMAs against the wave's 64-channel slice of the 1x1 weights (held in
        // registers for the whole epilogue) give the partial [32 x 64]; (4) the four waves that share the rows add their partials in
        // wave order (fixed: bit-reproducible whatever the batch slot) and write [32 x fuse_n] fp32.
        constexpr int CS = 64 + 8, AST = 272;
        constexpr int SLICE = 32 * CS * 4 + 32 * AST;  // 9216 + 8704 bytes per wave
        const int hi = lane >> 5;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float* cst = reinterpret_cast<float*>(smem + wave * SLICE);
        unsigned char* ast = smem + wave * SLICE + 32 * CS * 4;
        const int rrow = lane >> 3, c8 = (lane & 7) * 8;
        // this wave's slice of the 1x1 weights: n-block jn, channel chunk cch (channels nb + 32 cch ..), slot pair j
        u32x4 fw[2][2][4];
        {
            const unsigned char* wbase = reinterpret_cast<const unsigned char*>(ax.fuse_w);
            const int nchunks = d.Npad / 32;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int cch = 0; cch < 2; ++cch)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        fw[jn][cch][j] = *reinterpret_cast<const u32x4*>(wbase + ((size_t)(jn * 32 + frow) * nchunks + (nb >> 5) + cch) * 128 + ((2 * j + half) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e] * scj[j] + shj[j];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = rr * 8 + rrow;
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                conv_bf16_act(v, d.act, d.slope);
                float sv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) sv[q] = v[q] * H2_SCALE;
                f16x8 vh, vl;
                const bool over = h2_split8(sv, vh, vl);
                if (over && ax.overflow_flag && m0 + wr * 128 + i * 32 + row < a.M) *ax.overflow_flag = 1;
                unsigned char* ap = ast + row * AST + (c8 >> 5) * 128 + (c8 & 31) * 2;
                *reinterpret_cast<f16x8*>(ap) = vh;
                *reinterpret_cast<f16x8*>(ap + 64) = vl;
            }
            // (the wave's own LDS writes are ordered before its reads; data dependence through lgkmcnt)
            f32x16 oacc[2];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[jn][e] = 0.f;
            {
                H2_PAIRS;
#pragma unroll
                for (int cch = 0; cch < 2; ++cch) {
                    u32x4 af[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const u32x4*>(ast + frow * AST + cch * 128 + ((2 * j + half) << 4));
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) oacc[jn] = h2_mfma(af[H2_PA[pr]], fw[jn][cch][H2_PB[pr]], oacc[jn]);
                }
            }
            // partial [32 rows x 64 n] of this wave -> its cst slice (free again), then the four waves of the row group add in wave order
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + jn * 32 + frow] = oacc[jn][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                const int row = lane >> 1, n8 = wc * 16 + (lane & 1) * 8;  // this wave finishes columns [16 wc, 16 wc + 16) of the 32 rows
                float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const float* ps = reinterpret_cast<const float*>(smem + (wr * 4 + w4) * SLICE) + row * CS + n8;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(ps), p1 = *reinterpret_cast<const f32x4*>(ps + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        r[q] += p0[q];
                        r[4 + q] += p1[q];
                    }
                }
                const long long mrow = m0 + wr * 128 + i * 32 + row;
                if (mrow < a.M && n8 < ax.fuse_cs) {
                    float* op = ax.fuse_out + h2_pixel_of(a, mrow) * ax.fuse_cs + n8;
                    f32x4 o0, o1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o0[q] = r[q] * ax.fuse_scale[n8 + q] + ax.fuse_bias[n8 + q];
                        o1[q] = r[4 + q] * ax.fuse_scale[n8 + 4 + q] + ax.fuse_bias[n8 + 4 + q];
                    }
                    *reinterpret_cast<f32x4*>(op) = o0;
                    *reinterpret_cast<f32x4*>(op + 4) = o1;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave has read the partials before the next block overwrites them
            asm volatile("" ::: "memory");
        }
        return;
    }
    if constexpr (CMAX) {
        // lane (frow, half) holds column nb + j*32 + frow at rows i*32 + (e&3) + 8*(e>>2) + 4*half of the wave's 128 rows.  The h2 record of a
        // value is a monotonic function of the value, so the record of the column's largest value is the record of max(scale * acc + shift):
        // one v_max per element, ONE split per lane and column (the element-wise form - split every value, compare 64-bit keys - ran
        // 64 us against the plain launch's 57); the range check of the format (|16 v| > 65504, inf, NaN) is one unsigned max of the |bit
        // patterns| per element, as in h2_format.h
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(ax.fuse_out);
        const int hi = lane >> 5;
        const long long rbase = m0 + wr * 128 + 4 * hi;
        const bool full = m0 + 256 <= a.M;  // block-uniform: no row of the tile lies past M
        unsigned mbits = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float vmax = -__builtin_huge_valf();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][j][e] * scj[j] + shj[j];
                    const bool valid = full || rbase + i * 32 + (e & 3) + 8 * (e >> 2) < a.M;
                    const unsigned ub = __float_as_uint(v) & 0x7fffffffu;
                    mbits = (valid && ub > mbits) ? ub : mbits;
                    vmax = valid ? __builtin_fmaxf(vmax, v) : vmax;
                }
            const float vo = __shfl_xor(vmax, 32, 64);
            vmax = __builtin_fmaxf(vmax, vo);
            const int n = nb + j * 32 + frow;
            if (hi == 0 && n < d.N && vmax > -__builtin_huge_valf()) {
                _Float16 hq, lq;
                h2_split(vmax * H2_SCALE, hq, lq);
                const float r = (float)hq + (float)lq;
                const unsigned u = __float_as_uint(r);
                const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                const unsigned pay = ((unsigned)__builtin_bit_cast(unsigned short, hq) << 16) | (unsigned)__builtin_bit_cast(unsigned short, lq);
                atomicMax(keys + (m0 / ax.fuse_cs) * d.Npad + n, ((unsigned long long)ord << 32) | pay);
            }
        }
        // |16 v| > 65504  <=>  |v| > 4094 (0x457fe000), inf and NaN order above it too
        if (mbits > 0x457fe000u && ax.overflow_flag) *ax.overflow_flag = 1;
        return;
    }
    // ---- epilogue: each wave transposes its 128 x 64 tile through its own LDS slice, 8 channels per lane on the way out
    {
        const int hi = lane >> 5;
        constexpr int CS = 64 + 8;
        // (every wave is past its last fragment read, no DMA is in flight: a bare barrier; __syncthreads() would add a fence)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float* cst = reinterpret_cast<float*>(smem) + wave * (32 * CS);
        const int rrow = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e] * scj[j] + shj[j];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = rr * 8 + rrow;
                const long long mrow = m0 + wr * 128 + i * 32 + row;
                if (mrow >= a.M) continue;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                h2_finish_row8(ax, v, h2_pixel_of(a, mrow), nb + c8);
            }
        }
    }
#ifdef RDPN6D_PROBE
    {
        const unsigned long long t_issued = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_done = __builtin_readcyclecounter();
        if (g_h2_probe && lane == 0) {
            unsigned long long* o = g_h2_probe + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = pq_loop0 - pq_start;   // set-up + prologue
            o[1] = pq_loop1 - pq_loop0;   // K loop
            o[2] = t_issued - pq_loop1;   // epilogue until the last store is issued
            o[3] = t_done - t_issued;     // ... until the stores are done
            o[4] = pq_rt0;                // start on the 100 MHz clock
            o[5] = __builtin_amdgcn_s_memrealtime();
            o[6] = (unsigned long long)nk;
        }
    }
#endif
}

template <bool FUSE>
__global__ __launch_bounds__(512) void conv_h2_8ph_kernel_t(const ConvH2Args ax)
{
    const bool probe = ax.clk_probe != nullptr && blockIdx.x == 0 && blockIdx.y == 0;  // uniform
    unsigned long long t0 = 0, r0 = 0;
    if (probe) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    conv_h2_8ph_body<FUSE, false>(ax);
    if (probe && threadIdx.x == 0) {
        ax.clk_probe[0] = t0; ax.clk_probe[1] = r0;
        ax.clk_probe[2] = __builtin_readcyclecounter(); ax.clk_probe[3] = __builtin_amdgcn_s_memrealtime();
    }
}
__global__ __launch_bounds__(512) void conv_h2_8ph_colmax_kernel(const ConvH2Args ax) { conv_h2_8ph_body<false, true>(ax); }

constexpr auto conv_h2_8ph_kernel = conv_h2_8ph_kernel_t<false>;

// ============================================================================================ 128x128 .. 64x64 tile kernel
// conv_igemm_bf16.hip's two-stage form (2x2 wavefronts, LDS-DMA staging, one barrier per K-chunk, fragment double buffer,
// two workgroups per CU) on 128-byte h2 rows: 6 MFMAs per 32x32 tile pair and chunk from 4 + 4 fragment reads.
// NST = 3 (tiles up to 72 KiB of LDS, still two workgroups per CU): the DMA runs TWO chunks ahead and is never drained inside
// the loop - a step ends with s_waitcnt vmcnt(<DMAs of one chunk>) + a raw s_barrier, so the loads of chunk k+3 stay in flight
// across the barrier that publishes chunk k+2 (the L2 -> LDS round trip, ~1.5 us under load, is what bounds the 2-stage form:
// a step's 12 - 24 MFMAs take 0.2 - 0.4 us).
template <int BM, int BN, int NST, int SCHED = 0>
__global__ __launch_bounds__(256, 2) void conv_h2_tile_kernel(const ConvH2Args ax)
{
    constexpr int NW = 4, RB = 128;
    constexpr int RPP = 1024 / RB;  // 8 rows per 1-KiB DMA piece
    constexpr int WGN = 2, WTM = BM / 2, WTN = BN / 2;  // 2 x 2 wavefronts (a 4 x 1 grid with 256x64 tiles for N = 64 was slower: layer1 90 vs 85 us)
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int AG = BM / RPP / NW, BG = BN / RPP / NW;
    constexpr int NDMA = AG + BG;  // LDS-DMA instructions per wave and chunk
    static_assert(TM >= 1 && TN >= 1 && AG >= 1 && BG >= 1, "tile / wave layout");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + NST * BM * RB;

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

    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned px_bytes = (unsigned)d.in_cs * 4u;
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
    unsigned w_off[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave + NW * i) * RPP + prow;
        const int lslot = pslot ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)(n0 + row) * (unsigned)a.Ktot * 4u + (unsigned)lslot * 16u;
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, a.w_bytes, 0x00020000);

    auto stage_chunk = [&](const int tap, const int cc, const int st) {
        const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned toff = (unsigned)((dy * d.W + dx) * (int)px_bytes + cc * RB);  // wave-uniform
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const unsigned kill = ((a_mask[i] >> tap) & 1u) - 1u;  // all ones outside the image / past M: reads zeros
            unsigned char* dst = As + ((st * BM) + (wave + NW * i) * RPP) * RB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)dst, 16, (int)((a_base[i] + toff) | kill), 0, 0, 0);
        }
        const unsigned wk = (unsigned)tap * (unsigned)d.Cin * 4u + (unsigned)cc * (unsigned)RB;
#pragma unroll
        for (int i = 0; i < BG; ++i) {
            unsigned char* dst = Bs + ((st * BN) + (wave + NW * i) * RPP) * RB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr_t)dst, 16, (int)(w_off[i] + wk), 0, 0, 0);
        }
    };
    const int nk = a.kper;  // all chunks, or K-slice blockIdx.y (chunk order: channel-chunk major, taps innermost)
    const int kt0 = (int)blockIdx.y * a.kper;
    int ld_cc = kt0 / d.ntaps, ld_tap = kt0 - (kt0 / d.ntaps) * d.ntaps, ld_left = nk - 1;
    auto next_chunk = [](int& tap, int& cc, int& left, const int ntaps) {
        const int go = left > 0 ? 1 : 0;
        left -= go;
        tap += go;
        const int wrap = tap == ntaps ? 1 : 0;
        tap = wrap ? 0 : tap;
        cc += wrap;
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 31;
    const int half = lane >> 5;

    auto read_frags = [&](int st, u32x4 (&fa)[TM][4], u32x4 (&fb)[TN][4]) {
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

    // ---- epilogue operands, requested FIRST (the oldest entries of the vmcnt queue: the first wait of the K loop covers them): the h2
    // residual records of this lane's output rows and the folded BatchNorm scale / shift of its channels.  Requested in the epilogue -
    // which every workgroup of a one-round launch reaches at the same time - they were 4 000 - 9 000 cycles of exposed HBM latency per
    // workgroup (probe builds: tools/probe_h2_tile.py, tools/probe_h2_pp.py)
    constexpr int WC = WTN, CS = WC + 8, LPR = WC / 8, RPI = 64 / LPR, NRR = 32 / RPI;
    const int nb = n0 + wn * WC;
    const int rrow = lane / LPR, c8 = (lane % LPR) * 8;
    f16x8 rh[TM][NRR], rl[TM][NRR];
    long long pixs[TM][NRR];
    const bool res_pre = ax.res_h2 != nullptr && d.res == nullptr && ax.partial == nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int rr = 0; rr < NRR; ++rr) {
            const long long mrow = m0 + wm * WTM + i * 32 + rr * RPI + rrow;
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
        scj[j] = (d.scale && !ax.partial) ? d.scale[n] : 1.f;
        shj[j] = (d.shift && !ax.partial) ? d.shift[n] : 0.f;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma = [&](const u32x4 (&fa)[TM][4], const u32x4 (&fb)[TN][4]) {
        H2_PAIRS;
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = h2_mfma(fa[i][H2_PA[pr]], fb[jn][H2_PB[pr]], acc[i][jn]);
    };

    u32x4 fa0[TM][4], fb0[TN][4], fa1[TM][4], fb1[TN][4];
    [[maybe_unused]] unsigned long long pr_sum[5] = {0, 0, 0, 0, 0};
    [[maybe_unused]] const unsigned long long pr_start = SCHED == 9 ? __builtin_readcyclecounter() : 0ull;
    [[maybe_unused]] const unsigned long long pr_rt0 = SCHED == 9 ? (__builtin_amdgcn_s_memrealtime() & 0xffffffffull) : 0ull;
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
        // Invariant at the top of step kc: fragments of chunk kc in registers; stage (kc+1)%3 holds chunk kc+1, landed and
        // published; chunk kc+2 is in flight into stage (kc+2)%3; stage kc%3 is free.  Step: DMA chunk kc+3 -> stage kc%3;
        // LDS -> registers of chunk kc+1; MFMAs of chunk kc; wait until only this step's NDMA loads are outstanding and the
        // fragment reads have returned; barrier (publishes chunk kc+2, frees stage (kc+1)%3).
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
        auto step = [&](const int stf, const int stn, u32x4 (&fcur_a)[TM][4], u32x4 (&fcur_b)[TN][4], u32x4 (&fnxt_a)[TM][4], u32x4 (&fnxt_b)[TN][4]) {
            if constexpr (SCHED == 0) {
                stage_chunk(ld_tap, ld_cc, stf);
                next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
                read_frags(stn, fnxt_a, fnxt_b);
                mma(fcur_a, fcur_b);
                publish();
            } else if constexpr (SCHED == 9) {
#ifdef RDPN6D_PROBE
                const unsigned long long t0 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
                read_frags(stn, fnxt_a, fnxt_b);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t1 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
                stage_chunk(ld_tap, ld_cc, stf);
                next_chunk(ld_tap, ld_cc, ld_left, d.ntaps);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t2 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
                mma(fcur_a, fcur_b);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t3 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
                const unsigned long long t4 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const unsigned long long t5 = __builtin_readcyclecounter();
                pr_sum[0] += t1 - t0;
                pr_sum[1] += t2 - t1;
                pr_sum[2] += t3 - t2;
                pr_sum[3] += t4 - t3;
                pr_sum[4] += t5 - t4;
#endif
            }
        };
        for (int pr = 0; pr < npairs; ++pr) {
            step(st_free, st_next, fa0, fb0, fa1, fb1);
            st_free = st_next;
            st_next = st_next == 2 ? 0 : st_next + 1;
            step(st_free, st_next, fa1, fb1, fa0, fb0);
            st_free = st_next;
            st_next = st_next == 2 ? 0 : st_next + 1;
        }
        if (nk & 1) mma(fa0, fb0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing in LDS when the epilogue re-uses it
    }
#ifdef RDPN6D_PROBE
    [[maybe_unused]] const unsigned long long pr_loop_end = SCHED == 9 ? __builtin_readcyclecounter() : 0ull;
#endif

    {
        const int hi = lane >> 5;
        // every wave is past its last fragment read; nothing may still be landing in LDS - then a bare barrier (not __syncthreads(), whose
        // fence would wait for every outstanding global access as well)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float* cst = reinterpret_cast<float*>(smem) + wave * (32 * CS);
        if (ax.partial) {  // split-K: the raw partial tile of this K-slice (rows past M included: the workspace is tile-padded)
            float* part = ax.partial + ((size_t)blockIdx.y * ax.mpad + (size_t)m0 + wm * WTM) * d.Npad + nb + c8;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) cst[((e & 3) + 8 * (e >> 2) + 4 * hi) * CS + j * 32 + frow] = acc[i][j][e];
#pragma unroll
                for (int rr = 0; rr < 32 / RPI; ++rr) {
                    const int row = rr * RPI + rrow;
                    float* pp = part + (size_t)(i * 32 + row) * d.Npad;
                    *reinterpret_cast<f32x4*>(pp) = *reinterpret_cast<const f32x4*>(cst + row * CS + c8);
                    *reinterpret_cast<f32x4*>(pp + 4) = *reinterpret_cast<const f32x4*>(cst + row * CS + c8 + 4);
                }
            }
            return;
        }
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
    if constexpr (SCHED == 9) {
        if (g_h2_probe && lane == 0) {
            const unsigned long long t_end = __builtin_readcyclecounter();
            unsigned long long* o = g_h2_probe + ((size_t)blockIdx.x * 4 + wave) * 8;
            for (int i = 0; i < 5; ++i) o[i] = pr_sum[i];
            o[5] = pr_loop_end - pr_start;  // prologue + loop
            o[6] = t_end - pr_loop_end;     // epilogue
            o[7] = pr_rt0 | (__builtin_amdgcn_s_memrealtime() << 32);  // start | end on the constant 100 MHz clock (low 32 bits each)
        }
    }
#endif
}

// Second pass of the split-K form: one thread = 8 consecutive channels of one output row; the K-slices are added in slice order
// (three loads in flight per step), then scale / shift and the common tail (residual, activation, fp32 and / or h2 stores).
__global__ __launch_bounds__(256) void h2_splitk_reduce_kernel(const ConvH2Args ax)
{
    const ConvBArgs& a = ax.b;
    const rdpn6d_conv_desc& d = a.d;
    const int n8 = d.Npad >> 3;
    const long long total = a.M * n8;
    const size_t sstride = (size_t)ax.mpad * d.Npad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / n8;
        const int ch = (int)(i - m * n8) * 8;
        if (ch >= d.N) continue;
        const float* pp = ax.partial + (size_t)m * d.Npad + ch;
        f32x4 s0 = *reinterpret_cast<const f32x4*>(pp), s1 = *reinterpret_cast<const f32x4*>(pp + 4);
        int sl = 1;
        for (; sl + 3 <= ax.nsplit; sl += 3) {
            const float* q = pp + sl * sstride;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(q), t1 = *reinterpret_cast<const f32x4*>(q + 4);
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(q + sstride), u1 = *reinterpret_cast<const f32x4*>(q + sstride + 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(q + 2 * sstride), w1 = *reinterpret_cast<const f32x4*>(q + 2 * sstride + 4);
            s0 = ((s0 + t0) + u0) + w0;
            s1 = ((s1 + t1) + u1) + w1;
        }
        for (; sl < ax.nsplit; ++sl) {
            const float* q = pp + sl * sstride;
            s0 += *reinterpret_cast<const f32x4*>(q);
            s1 += *reinterpret_cast<const f32x4*>(q + 4);
        }
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = s0[q] * (d.scale ? d.scale[ch + q] : 1.f) + (d.shift ? d.shift[ch + q] : 0.f);
            v[4 + q] = s1[q] * (d.scale ? d.scale[ch + 4 + q] : 1.f) + (d.shift ? d.shift[ch + 4 + q] : 0.f);
        }
        h2_finish_row8(ax, v, h2_pixel_of(a, m), ch);
    }
}

template <int BM, int BN, int NST, int SCHED = 0>
int launch_h2_tile_s(const ConvH2Args& ax, hipStream_t s);
template <int BM, int BN, int NST>
int launch_h2_tile(const ConvH2Args& ax, hipStream_t s)
{
#ifdef RDPN6D_PROBE
    static const int sched = getenv("RDPN6D_H2_SCHED") ? atoi(getenv("RDPN6D_H2_SCHED")) : 0;  // 9 = the step cut into timed parts (tools/probe_h2_tile.py)
    if (sched == 9) return launch_h2_tile_s<BM, BN, NST, 9>(ax, s);
#endif
    return launch_h2_tile_s<BM, BN, NST, 0>(ax, s);
}
template <int BM, int BN, int NST, int SCHED>
int launch_h2_tile_s(const ConvH2Args& ax, hipStream_t s)
{
    constexpr int lds_stage = NST * (BM + BN) * 128;
    constexpr int lds_epi = 4 * 32 * (BN / 2 + 8) * 4;
    constexpr int lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    static_assert(lds <= 80 * 1024 || (BM == 128 && BN == 128 && NST == 3), "two workgroups per CU (128x128 with three stages: one)");
    auto kern = conv_h2_tile_kernel<BM, BN, NST, SCHED>;
    if (lds > 64 * 1024) {
        RD_LDS_OPT_IN(kern, lds);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ax.b.mtiles * ax.b.ntiles), (unsigned)ax.nsplit), dim3(256), lds, s, ax);
    return RDPN6D_OK;
}

// ============================================================================================ fp32 -> h2
// x: fp32 NHWC pixels with channel stride src_cs, slice [src_co, src_co + C), C % 32 == 0 -> dst h2 tensor [npix][C/32][2][32]
__global__ void split_h2_kernel(const float* __restrict__ x, int src_cs, int src_co, int C, _Float16* __restrict__ dst, long long npix,
                                int* __restrict__ overflow_flag)
{
    const int c8n = C >> 3;
    const long long total = npix * c8n;
    bool over = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / c8n;
        const int c = (int)(i - p * c8n) * 8;
        const float* sp = x + p * src_cs + src_co + c;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        f16x8 hi, lo;
        float sv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sv[q] = v[q] * H2_SCALE;
        over |= h2_split8(sv, hi, lo);
        _Float16* pp = dst + p * (2 * (long long)C) + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<f16x8*>(pp) = hi;
        *reinterpret_cast<f16x8*>(pp + 32) = lo;
    }
    if (over && overflow_flag) *overflow_flag = 1;
}

bool h2_common_ok(const rdpn6d_conv_desc* d)
{
    return d->Cin % 32 == 0 && d->in_cs % 32 == 0 && d->in_co % 32 == 0 && d->out_cs % 8 == 0 && d->out_co % 8 == 0 && d->N % 8 == 0 &&
           d->Npad % 64 == 0 && (!d->res || (d->res_cs % 4 == 0 && d->res_co % 4 == 0));
}
bool h2_big_ok(const rdpn6d_conv_desc* d)
{
    const int nk = d->ntaps * (d->Cin / 32);
    return h2_common_ok(d) && d->Npad % 256 == 0 && d->N == d->Npad && nk >= 2 && (nk & 1) == 0;
}
bool h2_big_pays(const rdpn6d_conv_desc* d, long long M)
{
    const long long tiles = (long long)rd_cdiv(M, 256) * (d->Npad / 256);
    const long long rounds = (tiles + 255) / 256;
    return tiles >= 160 && (double)tiles >= 0.62 * 256.0 * (double)rounds;
}
void h2_pick_tile(const rdpn6d_conv_desc* d, long long M, int* pbm, int* pbn)
{
    int bn = (d->Npad % 128 == 0) ? 128 : 64;
    int bm = 128;
    if ((long long)rd_cdiv(M, 128) * (d->Npad / bn) < 512) bm = 64;
    if (bm == 64 && bn == 128 && (long long)rd_cdiv(M, 64) * (d->Npad / 128) < 512) bn = 64;
    if (const char* f = getenv("RDPN6D_H2_TILE")) {  // profiling: "bm,bn"
        int fbm = 0, fbn = 0;
        if (sscanf(f, "%d,%d", &fbm, &fbn) == 2 && (fbm == 64 || fbm == 128) && (fbn == 64 || fbn == 128) && d->Npad % fbn == 0) {
            bm = fbm;
            bn = fbn;
        }
    }
    *pbm = bm;
    *pbn = bn;
}

}  // namespace

extern "C" int rdpn6d_split_h2(const float* x, int src_cs, int src_co, int C, void* dst, long long npix, int* overflow_flag, void* stream)
{
    RD_REQUIRE(x && dst && npix > 0 && C > 0 && C % 32 == 0 && src_cs % 4 == 0 && src_co % 4 == 0 && src_co + C <= src_cs, "h2 split: C % 32, aligned slice");
    const long long blocks = (npix * (C / 8) + 255) / 256;
    hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, x, src_cs,
                       src_co, C, (_Float16*)dst, npix, overflow_flag);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

int conv_h2_launch_pp(ConvH2Args& ax, int shape, hipStream_t s);  // conv_igemm_h2_pp.hip

// Does the 8-wave ping-pong kernel (conv_igemm_h2_pp.hip) take this launch, and with which tile (0 = 128x128, 2 = 256x128)?  One
// workgroup per CU: it wants >= 224 tiles and a K loop long enough to amortise its prologue.  Measured at B = 64 (profiles/r3_logs/r3_h_conv.log):
// layer3 65.1 -> 58.3 us (128x128), layer2 65.2 -> 61.8 us (256x128: one round instead of two of 128x128, 67.7 us).  Two forms that
// were built and dropped: a 256x64 tile for N = 64 (layer1: 95-100 us against the tile kernel's 83) and K cut into 2 / 4 slices for
// launches with 56 .. 223 tiles (layer4: 70-71 us against 72).  RDPN6D_H2_PP = 0 switches the kernel off (profiling).
static int h2_pp_plan(const rdpn6d_conv_desc* d, long long M, int* pbm, int* pbn)
{
    static const int on = getenv("RDPN6D_H2_PP") ? atoi(getenv("RDPN6D_H2_PP")) : 1;
    static const int force = getenv("RDPN6D_H2_PP_SHAPE") ? atoi(getenv("RDPN6D_H2_PP_SHAPE")) : -1;  // profiling: 0 | 2
    if (!on || d->Npad % 128 != 0 || d->ntaps * (d->Cin / 32) < 8) return -1;
    int shape = 0;
    if ((long long)rd_cdiv(M, 256) * (d->Npad / 128) >= 224 && (long long)rd_cdiv(M, 128) * (d->Npad / 128) > 288) shape = 2;
    if (force == 0 || force == 2) shape = force;
    const int bm = shape == 0 ? 128 : 256;
    if ((long long)rd_cdiv(M, bm) * (d->Npad / 128) < 224) return -1;
    *pbm = bm;
    *pbn = 128;
    return shape;
}

// K-slices of the tile kernel for a launch too small to fill the chip (0 = do not split): the largest power of two that keeps
// tiles * slices <= 512 workgroups with >= 4 whole chunks per slice
static int h2_tile_ksplit(const rdpn6d_conv_desc* d, long long M, int bm, int bn)
{
    static const int off = getenv("RDPN6D_H2_NO_SPLITK") ? 1 : 0;  // profiling
    const long long tiles = (long long)rd_cdiv(M, bm) * (d->Npad / bn);
    static const int tmax = getenv("RDPN6D_H2_SPLIT_TILES") ? atoi(getenv("RDPN6D_H2_SPLIT_TILES")) : 128;  // profiling
    if (off || tiles >= tmax) return 0;
    const int nk = d->ntaps * (d->Cin / 32);
    int best = 0;
    for (int sl = 2; sl <= 32; sl *= 2)
        if (tiles * sl <= 512 && nk % sl == 0 && nk / sl >= 4) best = sl;
    return best;
}

// shapes of the ping-pong kernel that have a weights-from-L2 form (conv_igemm_h2_pp.hip, BFG).  MEASURED AND PARKED (round 5,
// profiles/r5_experiments.md): bit-identical, but layer3 runs 55.2 us against 52.5 with the weight tile in LDS and layer2 59 against 54 -
// the vector-memory path is no faster than the LDS port it relieves.  Off unless asked for: mode 1 = the 128x128 tile, 2 = + 256x128
// (rdpn6d_conv_h2_set_wfrag, or RDPN6D_H2_BFG in the environment for profiling runs)
static int g_h2_wfrag_mode = getenv("RDPN6D_H2_BFG") ? atoi(getenv("RDPN6D_H2_BFG")) : 0;
extern "C" void rdpn6d_conv_h2_set_wfrag(int mode) { g_h2_wfrag_mode = mode; }
// measurement: while set, every launch of the 256x256 eight-phase kernel (plain and fused-output instances) has its workgroup 0 write
// {s_memtime, s_memrealtime} x {start, end} to buf[0..3] (device memory, 32 bytes); null switches it off again
static unsigned long long* g_h2_clk_probe = nullptr;
extern "C" void rdpn6d_conv_h2_set_clock_probe(unsigned long long* buf) { g_h2_clk_probe = buf; }
static bool h2_wfrag_shape_ok(int shape)
{
    const int mode = g_h2_wfrag_mode;
    return mode >= 1 && (shape == 0 || (mode >= 2 && shape == 2));
}

// which h2 kernel rdpn6d_conv2d_h2 would use: 2 = 256x256 eight-phase, 1 = 128x128..64x64 tile kernel, 0 = not eligible
extern "C" int rdpn6d_conv_h2_kernel_for(const rdpn6d_conv_desc* d)
{
    if (!d || !h2_common_ok(d)) return 0;
    const long long M = (long long)d->B * d->Ho * d->Wo;
    return (h2_big_ok(d) && h2_big_pays(d, M)) ? 2 : 1;
}

extern "C" int rdpn6d_conv2d_h2_cb(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                                   void* stream);
extern "C" int rdpn6d_conv2d_h2(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, void* stream)
{
    return rdpn6d_conv2d_h2_cb(d, y_h2, res_h2, overflow_flag, nullptr, stream);
}

// bytes of workspace rdpn6d_conv2d_h2_ws wants for this layer (0: the launch does not split K)
extern "C" long long rdpn6d_conv_h2_workspace_bytes(const rdpn6d_conv_desc* d)
{
    if (!d || rdpn6d_conv_h2_kernel_for(d) != 1) return 0;
    const long long M = (long long)d->B * d->Ho * d->Wo;
    int bm, bn;
    if (h2_pp_plan(d, M, &bm, &bn) >= 0) return 0;
    h2_pick_tile(d, M, &bm, &bn);
    return (long long)h2_tile_ksplit(d, M, bm, bn) * rd_cdiv(M, bm) * bm * d->Npad * 4;
}

extern "C" int rdpn6d_conv2d_h2_ws(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                                   void* workspace, long long workspace_bytes, void* stream);
extern "C" int rdpn6d_conv2d_h2_cb(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                                   void* stream)
{
    return rdpn6d_conv2d_h2_ws(d, y_h2, res_h2, overflow_flag, crop_bias, nullptr, 0, stream);
}

struct H2Fuse {
    const void* w;  // null: the column-max form (out = the 64-bit key table, cs = rows per group)
    const float *scale, *bias;
    float* out;
    int cs, n;
};
static int conv2d_h2_impl(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                          void* workspace, long long workspace_bytes, const H2Fuse* fuse, void* stream, const void* w_frag = nullptr);

extern "C" int rdpn6d_conv2d_h2_ws(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                                   void* workspace, long long workspace_bytes, void* stream)
{
    return conv2d_h2_impl(d, y_h2, res_h2, overflow_flag, crop_bias, workspace, workspace_bytes, nullptr, stream);
}

// Does this layer's kernel take its weights FRAGMENT-MAJOR (rdpn6d_h2_weight_frag) next to the row-major h2 tensor?  1: pass them to
// rdpn6d_conv2d_h2_wf; the call works without them (the kernel then stages the weight tile through LDS like every other).
extern "C" int rdpn6d_conv_h2_wfrag_wanted(const rdpn6d_conv_desc* d)
{
    if (!d || rdpn6d_conv_h2_kernel_for(d) != 1) return 0;
    int bm, bn;
    const int shape = h2_pp_plan(d, (long long)d->B * d->Ho * d->Wo, &bm, &bn);
    return shape >= 0 && h2_wfrag_shape_ok(shape) ? 1 : 0;
}

extern "C" int rdpn6d_conv2d_h2_wf(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const void* w_frag,
                                   void* stream)
{
    return conv2d_h2_impl(d, y_h2, res_h2, overflow_flag, nullptr, nullptr, 0, nullptr, stream, w_frag);
}

// h2 weight tensor [Npad][ntaps][cchunks][hi x 32 | lo x 32] fp16 (gdrn.pack_h2_weight; a 128-byte record = eight 16-byte slots) ->
// fragment-major [Npad / 32][ntaps][cchunks][slot 0..7][row 0..31][8 halfs]: the 16-byte fragment slot s of the 32 rows of a block
// contiguous, so that a wave's fragment load (lane l: row l & 31, slot 2 j + (l >> 5)) is ONE 1-KiB run.  Npad % 32 == 0.
__global__ void h2_weight_frag_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, int ntaps, int cchunks, long long nslots)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // destination slot index
    if (i >= nslots) return;
    const int r = (int)(i & 31), sl = (int)((i >> 5) & 7);
    const long long blk = i >> 8;  // (n32 * ntaps + tap) * cchunks + cc
    const int cc = (int)(blk % cchunks);
    const long long t2 = blk / cchunks;
    const int tap = (int)(t2 % ntaps);
    const long long n32 = t2 / ntaps;
    out[i] = w[(((n32 * 32 + r) * ntaps + tap) * cchunks + cc) * 8 + sl];
}

extern "C" int rdpn6d_h2_weight_frag(const void* w_h2, int Npad, int ntaps, int cchunks, void* w_frag, void* stream)
{
    RD_REQUIRE(w_h2 && w_frag && w_h2 != w_frag, "null pointer / in place");
    RD_REQUIRE(Npad > 0 && Npad % 32 == 0 && ntaps >= 1 && cchunks >= 1, "shape (Npad % 32)");
    const long long nslots = (long long)Npad * ntaps * cchunks * 8;
    hipLaunchKernelGGL(h2_weight_frag_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)w_h2, (uint4*)w_frag, ntaps, cchunks, nslots);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// The convolution with a 1x1 output convolution fused into its epilogue (the dense head's last 3x3 layer + features.21,
// cdpn_rot_head_region.py:130-138): out[pixel][n] = scale1[n] * sum_c act(conv)[pixel][c] * w1[n][c] + bias1[n], n < n_out.  The
// activation is never written.  w1: h2 records [64][N/32][hi|lo] (gdrn.pack_h2_weight of the [64][1][N] fp32 matrix, rows >= n_out
// zero); scale1 / bias1: 64 floats; out: fp32 [pixels][out_cs], out_cs % 8 == 0, n_out <= out_cs <= 64.  Needs the 256x256 kernel with
// ONE tile across N (N == Npad == 256): rdpn6d_conv_h2_fuse1x1_ok.
extern "C" int rdpn6d_conv_h2_fuse1x1_ok(const rdpn6d_conv_desc* d)
{
    return d && rdpn6d_conv_h2_kernel_for(d) == 2 && d->Npad == 256 && d->N == 256;
}
extern "C" int rdpn6d_conv2d_h2_fuse1x1(const rdpn6d_conv_desc* d, const void* res_h2, int* overflow_flag, const float* crop_bias,
                                        const void* w1_h2, const float* scale1, const float* bias1, float* out, int out_cs, int n_out,
                                        void* stream)
{
    RD_REQUIRE(d && w1_h2 && scale1 && bias1 && out, "null pointer");
    RD_REQUIRE(rdpn6d_conv_h2_fuse1x1_ok(d), "fused 1x1 output convolution: needs the 256x256 kernel with N == Npad == 256");
    RD_REQUIRE(n_out >= 1 && n_out <= out_cs && out_cs <= 64 && out_cs % 8 == 0, "n_out <= out_cs <= 64, out_cs % 8 == 0");
    RD_REQUIRE(d->y == nullptr, "the fused form does not write the activation (desc.y must be null)");
    // the fused epilogue applies scale, shift and activation only: a residual or a per-crop bias would be dropped silently
    RD_REQUIRE(!res_h2 && !crop_bias && !d->res, "the fused 1x1 form takes no residual and no per-crop bias");
    const H2Fuse f = {w1_h2, scale1, bias1, out, out_cs, n_out};
    return conv2d_h2_impl(d, nullptr, res_h2, overflow_flag, crop_bias, nullptr, 0, &f, stream);
}

// The convolution whose output only a per-group (per-crop) channel max reads: keys [groups][Npad] uint64, ZERO before the call (the
// decode kernel below leaves them zero), get per (group, channel) the h2 record of the largest value scale * conv + shift over the
// group's rows_per_group consecutive output rows - see conv_h2_8ph_body<.., CMAX>.  Needs the 256x256 kernel, rows_per_group % 256 == 0,
// no activation, no residual; the activation itself is never written (desc.y must be null).
extern "C" int rdpn6d_conv_h2_colmax_ok(const rdpn6d_conv_desc* d, int rows_per_group)
{
    return d && rdpn6d_conv_h2_kernel_for(d) == 2 && rows_per_group > 0 && rows_per_group % 256 == 0 && d->act == 0 && !d->res &&
           ((long long)d->B * d->Ho * d->Wo) % rows_per_group == 0 && d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo;
}
extern "C" int rdpn6d_conv2d_h2_colmax(const rdpn6d_conv_desc* d, unsigned long long* keys, int rows_per_group, int* overflow_flag,
                                       void* stream)
{
    RD_REQUIRE(d && keys, "null pointer");
    RD_REQUIRE(rdpn6d_conv_h2_colmax_ok(d, rows_per_group), "column-max form: 256x256 kernel, rows_per_group % 256 == 0, linear output, no activation / residual");
    RD_REQUIRE(d->y == nullptr, "the column-max form does not write the activation (desc.y must be null)");
    // atomicMax never decreases: the table must be zero when the launch starts.  The decode kernel re-zeroes it, but a forward that
    // stopped between the two launches (an error, a partially replayed launch list) would leave stale maxima for the next one - so the
    // table is cleared HERE, on the launch stream, whatever happened before (B * Npad * 8 bytes: 256 KiB at B = 64)
    const long long groups = (long long)d->B * d->Ho * d->Wo / rows_per_group;
    RD_CHECK_HIP(hipMemsetAsync(keys, 0, (size_t)groups * d->Npad * sizeof(unsigned long long), (hipStream_t)stream));
    const H2Fuse f = {nullptr, nullptr, nullptr, reinterpret_cast<float*>(keys), rows_per_group, 0};
    return conv2d_h2_impl(d, nullptr, nullptr, overflow_flag, nullptr, nullptr, 0, &f, stream);
}

// keys [groups][Npad] -> the h2 record [groups][N/32][hi x 32 | lo x 32] of the maxima (what rdpn6d_global_max_h2 writes), and the
// keys back to zero for the next forward
__global__ void h2_colmax_decode_kernel(unsigned long long* __restrict__ keys, int groups, int N, int Npad, _Float16* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * N) return;
    const int g = i / N, n = i - g * N;
    const unsigned long long k = keys[(size_t)g * Npad + n];
    keys[(size_t)g * Npad + n] = 0ull;
    const unsigned pay = (unsigned)(k & 0xffffffffull);
    _Float16* o = out + ((size_t)g * (N / 32) + (n >> 5)) * 64 + (n & 31);
    o[0] = __builtin_bit_cast(_Float16, (unsigned short)(pay >> 16));
    o[32] = __builtin_bit_cast(_Float16, (unsigned short)(pay & 0xffffu));
}
extern "C" int rdpn6d_h2_colmax_decode(unsigned long long* keys, int groups, int N, int Npad, void* out_h2, void* stream)
{
    RD_REQUIRE(keys && out_h2 && groups > 0 && N > 0 && N % 32 == 0 && Npad >= N, "shape (N % 32)");
    hipLaunchKernelGGL(h2_colmax_decode_kernel, dim3((unsigned)((groups * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys, groups, N,
                       Npad, (_Float16*)out_h2);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

static int conv2d_h2_impl(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, const float* crop_bias,
                          void* workspace, long long workspace_bytes, const H2Fuse* fuse, void* stream, const void* w_frag)
{
    RD_REQUIRE(d && d->x && d->w && (d->y || y_h2 || fuse), "null pointer");
    const int which = rdpn6d_conv_h2_kernel_for(d);
    RD_REQUIRE(which != 0, "h2 needs Cin, in_cs, in_co % 32 == 0, N % 8 == 0, Npad % 64 == 0, 16-byte aligned output slices");
    RD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->ntaps >= 1 && d->ntaps <= 9, "shape");
    RD_REQUIRE(d->in_co + d->Cin <= d->in_cs && d->out_co + d->N <= d->out_cs, "channel slices");
    RD_REQUIRE((d->Ho - 1) * d->osy + d->ooy < d->OH && (d->Wo - 1) * d->osx + d->oox < d->OW, "output geometry");
    RD_REQUIRE(!(d->res && res_h2), "residual either as an fp32 tensor or as an h2 tensor");
    RD_REQUIRE(!res_h2 || (d->res_cs % 32 == 0 && d->res_co % 8 == 0 && d->res_co + d->N <= d->res_cs), "h2 residual slice");
    RD_REQUIRE(!y_h2 || (d->out_cs % 32 == 0), "h2 output: channel stride % 32");
    ConvH2Args ax;
    ConvBArgs& a = ax.b;
    a.d = *d;
    a.M = (long long)d->B * d->Ho * d->Wo;
    RD_REQUIRE(a.M < (1LL << 31), "B*Ho*Wo must fit 31 bits");
    a.HoWo = d->Ho * d->Wo;
    a.cchunks = d->Cin / 32;
    a.nk = d->ntaps * a.cchunks;
    a.Ktot = d->ntaps * d->Cin;
    a.linear_out = (d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo);
    a.out_f32 = 1;
    a.vec_out = 1;
    const long long x_bytes = (long long)d->B * d->H * d->W * d->in_cs * 4, w_bytes = (long long)d->Npad * d->ntaps * d->Cin * 4;
    RD_REQUIRE(x_bytes < (1LL << 32) - 64 && w_bytes < (1LL << 32) - 64, "h2 tensors are addressed with 32-bit offsets (< 4 GiB each)");
    a.x_bytes = (unsigned)x_bytes;
    a.w_bytes = (unsigned)w_bytes;
    ax.y_h2 = y_h2;
    ax.res_h2 = res_h2;
    ax.overflow_flag = overflow_flag;
    ax.crop_bias = crop_bias;
    ax.partial = nullptr;
    ax.fuse_w = fuse ? fuse->w : nullptr;
    ax.fuse_scale = fuse ? fuse->scale : nullptr;
    ax.fuse_bias = fuse ? fuse->bias : nullptr;
    ax.fuse_out = fuse ? fuse->out : nullptr;
    ax.fuse_cs = fuse ? fuse->cs : 0;
    ax.fuse_n = fuse ? fuse->n : 0;
    ax.w_frag = nullptr;
    ax.clk_probe = g_h2_clk_probe;
    ax.nsplit = 1;
    ax.mpad = 0;
    a.dy_pack = a.dx_pack = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        RD_REQUIRE(d->dy[t] >= -8 && d->dy[t] <= 7 && d->dx[t] >= -8 && d->dx[t] <= 7, "tap offsets must be in -8..7");
        a.dy_pack |= (unsigned long long)(d->dy[t] + 8) << (4 * t);
        a.dx_pack |= (unsigned long long)(d->dx[t] + 8) << (4 * t);
    }
    a.kper = a.nk;
    a.partial = nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (which == 1) {
        int bm, bn;
        const int shape = h2_pp_plan(d, a.M, &bm, &bn);
        if (shape >= 0) {
            a.mtiles = rd_cdiv(a.M, bm);
            a.ntiles = d->Npad / bn;
            ax.w_frag = h2_wfrag_shape_ok(shape) ? w_frag : nullptr;  // fragment-major weights: the kernel loads its weight fragments from L2
            const int rc = conv_h2_launch_pp(ax, shape, s);
            if (rc != RDPN6D_OK) return rc;
            RD_LAUNCH_CHECK();
            return RDPN6D_OK;
        }
        h2_pick_tile(d, a.M, &bm, &bn);
        a.mtiles = rd_cdiv(a.M, bm);
        a.ntiles = d->Npad / bn;
        const int ks = h2_tile_ksplit(d, a.M, bm, bn);
        if (ks && workspace && workspace_bytes >= (long long)ks * a.mtiles * bm * d->Npad * 4) {
            ax.nsplit = ks;
            ax.mpad = a.mtiles * bm;
            ax.partial = reinterpret_cast<float*>(workspace);
            a.kper = a.nk / ks;
        }
        int rc;
        static const int nst_env = getenv("RDPN6D_H2_NST") ? atoi(getenv("RDPN6D_H2_NST")) : 0;  // profiling: force 2 | 3 stages
        // three stages (where two workgroups per CU still fit, <= 80 KiB) pay on long K loops only: layer4's 144 chunks 97 -> 70 us, but
        // layer1's 18 chunks run 92 us against 84 with two stages (a third workgroup per CU fits and the longer prologue is not amortised)
        const bool three = nst_env ? nst_env == 3 : a.kper >= 32;
        if (bm == 128 && bn == 128) rc = (nst_env == 3) ? launch_h2_tile<128, 128, 3>(ax, s) : launch_h2_tile<128, 128, 2>(ax, s);  // (96 KiB with three stages: one workgroup per CU)
        else if (bm == 128) rc = three ? launch_h2_tile<128, 64, 3>(ax, s) : launch_h2_tile<128, 64, 2>(ax, s);
        else if (bn == 128) rc = three ? launch_h2_tile<64, 128, 3>(ax, s) : launch_h2_tile<64, 128, 2>(ax, s);
        else rc = three ? launch_h2_tile<64, 64, 3>(ax, s) : launch_h2_tile<64, 64, 2>(ax, s);
        if (rc != RDPN6D_OK) return rc;
        RD_LAUNCH_CHECK();
        if (ax.partial) {
            const long long items = a.M * (d->Npad / 8);
            hipLaunchKernelGGL(h2_splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096)), dim3(256), 0, s, ax);
            RD_LAUNCH_CHECK();
        }
        return RDPN6D_OK;
    }
    a.mtiles = rd_cdiv(a.M, 256);
    a.ntiles = d->Npad / 256;
    if (fuse && !fuse->w) {  // column max (rdpn6d_conv2d_h2_colmax): nothing written but the key table
        RD_LDS_OPT_IN(conv_h2_8ph_colmax_kernel, LDS_8PH);
        hipLaunchKernelGGL(conv_h2_8ph_colmax_kernel, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), LDS_8PH, s, ax);
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    if (fuse) {
        RD_LDS_OPT_IN(conv_h2_8ph_kernel_t<true>, LDS_8PH_FUSED);
        hipLaunchKernelGGL(conv_h2_8ph_kernel_t<true>, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), LDS_8PH_FUSED, s, ax);
        RD_LAUNCH_CHECK();
        return RDPN6D_OK;
    }
    RD_LDS_OPT_IN(conv_h2_8ph_kernel, LDS_8PH);
#ifdef RDPN6D_PROBE
    if (const char* e = getenv("RDPN6D_H2_ABL_A")) ax.nsplit = 100 + atoi(e);
#endif
    hipLaunchKernelGGL(conv_h2_8ph_kernel, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), LDS_8PH, s, ax);  // (never split: ax.partial stays null)
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
