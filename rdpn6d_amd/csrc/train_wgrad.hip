// Weight-gradient implicit GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   out[a][t][b] = sum over pixels m of  A[m][a] * Bg[gather(m, t)][b]
//     conv wgrad     : A = dY (pixels x Cout),        Bg = X gathered at (oy*s+dy_t, ox*s+dx_t)  -> dW packed [Cout][taps][Cin]
//     convT wgrad    : A = X  (input pixels x Cin),   Bg = dY gathered at (2iy-1+ky, 2ix-1+kx)   -> [Cin][taps][Cout]
//     linear wgrad   : A = dY (B x Nout), Bg = X (B x K), taps = 1
//   The reduction runs over PIXELS (the NHWC row index), so both operands are staged pixel-major
//   [16 pixels][128 channels] exactly as they lie in HBM (512-byte coalesced rows, no transpose) by LDS-DMA
//   (buffer_load_dwordx4 ... lds, no staging VGPRs / ds_write), 3 stages; the MFMA fragments are read column-wise
//   with conflict-free ds_read_b32 (32 consecutive channels per half-wave) into a double-buffered register set.
//   Split-K over pixel ranges: grid.y splits write fp32 partial tiles that a second kernel sums in a fixed
//   order (deterministic, no atomics).
// Backward of nn.Conv2d / nn.ConvTranspose2d / nn.Linear weights in core/gdrn_modeling/models/*.py.
#include "common.h"
#include <cstdlib>

struct WgradKArgs {
    const float* A;
    const float* Bg;
    float* partial;  // [S][Ca][ntaps][Cb]
    long long M;     // pixels of A = Bn*Ha*Wa
    int Ha, Wa, HaWa;
    int Hb, Wb;
    int stride;
    int ntaps;
    unsigned long long dy_pack, dx_pack;
    int Ca, Cb;
    int a_cs, a_co, b_cs, b_co;
    int atiles, btiles;  // tiles over Ca and over Cb (per tap)
    int nsplit;
    long long rows_per_split;
    unsigned a_bytes, b_bytes;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BA, int BB>
__global__ __launch_bounds__(256, 3) void wgrad_f32_kernel(const WgradKArgs a)
{
    constexpr int TA = BA / 64, TB = BB / 64;  // 32x32 tiles per wave (2x2 waves)
    constexpr int NST = 3;                      // LDS stages (same 3-stage pipeline as conv_igemm_f32_kernel)
    __shared__ __attribute__((aligned(1024))) float As[NST][16][BA];
    __shared__ __attribute__((aligned(1024))) float Bs[NST][16][BB];

    // XCD-aware mapping: workgroup b runs on XCD b % 8; give every XCD whole splits (all tiles of a split stream through
    // the same pixel range in lockstep, so each dY / X row is fetched into one L2 only, once).
    const int ntile = a.atiles * a.btiles * a.ntaps;
    const int nblk = ntile * a.nsplit;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk;
    const int tile = logical % ntile;
    const int split = logical / ntile;
    const int at = tile % a.atiles;
    const int rest = tile / a.atiles;
    const int bt = rest % a.btiles;
    const int tap = rest / a.btiles;
    const int a0 = at * BA, b0 = bt * BB;
    const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
    const long long m_lo = (long long)split * a.rows_per_split;
    const long long m_hi = m_lo + a.rows_per_split < a.M ? m_lo + a.rows_per_split : a.M;
    const int nchunks = m_hi > m_lo ? (int)((m_hi - m_lo + 15) / 16) : 0;

    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Bg), 0, a.b_bytes, 0x00020000);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // staging by LDS-DMA (buffer_load_dwordx4 ... lds): one wave instruction moves 1 KiB = (1024 / (BA*4)) pixel rows of
    // BA channels straight into the pixel-major LDS tile (lane-linear = exactly the tile's layout, no swizzle needed:
    // the fragment reads are ds_read_b32 over 32 consecutive channels).
    constexpr int A4 = BA / 4, B4 = BB / 4;              // float4 (= lanes) per pixel row
    constexpr int AROWS = 64 / A4, BROWS = 64 / B4;      // pixel rows per DMA piece (BA=128: 2, BA=64: 4)
    constexpr int APIECES = 16 / AROWS / 4, BPIECES = 16 / BROWS / 4;  // pieces per wave (BA=128: 2, BA=64: 1)
    const int ar = lane / A4, ac4 = lane % A4, br = lane / B4, bc4 = lane % B4;
    const bool a_cok = a0 + ac4 * 4 < a.Ca, b_cok = b0 + bc4 * 4 < a.Cb;
    const unsigned a_col = (unsigned)(a.a_co + a0 + ac4 * 4) * 4u, b_col = (unsigned)(a.b_co + b0 + bc4 * 4) * 4u;
    const unsigned a_row_bytes = (unsigned)a.a_cs * 4u, b_px_bytes = (unsigned)a.b_cs * 4u;

    // pixel coordinates of this lane's B rows, advanced incrementally (16 pixels per chunk): no divisions in the loop
    int b_ox[BPIECES], b_oy[BPIECES], b_bi[BPIECES];
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
        const long long m = m_lo + (wave + 4 * p) * BROWS + br;
        const int mm = m < a.M ? (int)m : 0;
        b_bi[p] = mm / a.HaWa;
        const int rem = mm - b_bi[p] * a.HaWa;
        b_oy[p] = rem / a.Wa;
        b_ox[p] = rem - b_oy[p] * a.Wa;
    }
    long long ld_m = m_lo;  // first pixel of the next chunk to stage
    auto stage_chunk = [&](const int st) {
#pragma unroll
        for (int p = 0; p < APIECES; ++p) {
            const int row0 = (wave + 4 * p) * AROWS;
            const long long m = ld_m + row0 + ar;
            const bool ok = a_cok && m < m_hi;
            const unsigned off = ok ? (unsigned)m * a_row_bytes + a_col : a.a_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(asrc, (lds_ptr_t)&As[st][row0][0], 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < BPIECES; ++p) {
            const int row0 = (wave + 4 * p) * BROWS;
            const long long m = ld_m + row0 + br;
            const int iy = b_oy[p] * a.stride + dy, ix = b_ox[p] * a.stride + dx;
            const bool ok = b_cok && m < m_hi && (unsigned)iy < (unsigned)a.Hb && (unsigned)ix < (unsigned)a.Wb;
            const unsigned off = ok ? (unsigned)((b_bi[p] * a.Hb + iy) * a.Wb + ix) * b_px_bytes + b_col : a.b_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bsrc, (lds_ptr_t)&Bs[st][row0][0], 16, (int)off, 0, 0, 0);
            // advance this row by 16 pixels
            b_ox[p] += 16;
            while (b_ox[p] >= a.Wa) { b_ox[p] -= a.Wa; if (++b_oy[p] == a.Ha) { b_oy[p] = 0; ++b_bi[p]; } }
        }
        ld_m += 16;
    };

    const int wa = wave >> 1, wb = wave & 1;
    const int frow = lane & 31, koff = (lane >> 5) * 8;

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto read_frags = [&](int st, float (&fa)[TA][8], float (&fb)[TB][8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int i = 0; i < TA; ++i) fa[i][s] = As[st][koff + s][wa * (BA / 2) + i * 32 + frow];
#pragma unroll
            for (int j = 0; j < TB; ++j) fb[j][s] = Bs[st][koff + s][wb * (BB / 2) + j * 32 + frow];
        }
    };
    auto mma = [&](const float (&fa)[TA][8], const float (&fb)[TB][8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    };

    if (nchunks > 0) {
        // rows past the split's end are bounds-checked to zero (m >= m_hi), so the loop body needs no branches
        stage_chunk(0);
        stage_chunk(1);
        __syncthreads();  // waits for the DMA (vmcnt(0)) + barrier
        float fa0[TA][8], fb0[TB][8], fa1[TA][8], fb1[TB][8];
        read_frags(0, fa0, fb0);
        int st_next = 1, st_stage = 2;
        const int npairs = nchunks >> 1;
        for (int pr = 0; pr < npairs; ++pr) {
            // fragment reads are issued BEFORE the DMA in program order: the compiler cannot prove that the DMA target
            // stage differs from the stage being read and would otherwise drain vmcnt(0) in the middle of the MFMAs
            read_frags(st_next, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            stage_chunk(st_stage);
            mma(fa0, fb0);
            __syncthreads();
            st_next = st_next == NST - 1 ? 0 : st_next + 1;
            st_stage = st_stage == NST - 1 ? 0 : st_stage + 1;

            read_frags(st_next, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            stage_chunk(st_stage);
            mma(fa1, fb1);
            __syncthreads();
            st_next = st_next == NST - 1 ? 0 : st_next + 1;
            st_stage = st_stage == NST - 1 ? 0 : st_stage + 1;
        }
        if (nchunks & 1) mma(fa0, fb0);
    }

    // partial[split][a][tap][b]: lanes hold consecutive b (32 x 4 B = 128-byte runs)
    const int hi = lane >> 5;
    float* po = a.partial + (long long)split * a.Ca * a.ntaps * a.Cb;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int bch = b0 + wb * (BB / 2) + j * 32 + frow;
        if (bch >= a.Cb) continue;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ach = a0 + wa * (BA / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (ach < a.Ca) po[((long long)ach * a.ntaps + tap) * a.Cb + bch] = acc[i][j][e];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 form (mixed-precision training, cfg.SOLVER.AMP.ENABLED): operands are the compact bf16 copies of dY and X
// that the bf16 forward / dgrad convolutions read anyway; fp32 accumulation on v_mfma_f32_32x32x16_bf16; fp32
// partial tiles, same split-K and reduce.  The reduction index (pixels) is the SLOW index of both NHWC operands,
// while an MFMA lane wants 8 consecutive k-values of one channel: the tiles are staged pixel-major [32 pixels][BA
// channels] by LDS-DMA exactly as they lie in HBM and the fragments are read with the gfx950 transpose read
// ds_read_b64_tr_b16 (each 16-lane group fetches a [4 pixels][16 channels] block, lane L receives channel L of the
// four pixels): two reads per 8-k fragment, no ds_write, no register shuffles.  16-byte chunks of a row are
// XOR-swizzled with the pixel index (on the DMA source address and on the read) so that the 8 row segments one
// read cycle touches cover all 64 banks.
#define RD_WGRAD_MAX_GROUP 16
struct WgradBArgs {
    const unsigned short* A;
    const unsigned short* Bg;
    float* partial;
    long long M;
    int Ha, Wa, HaWa;
    int Hb, Wb;
    int stride;
    int ntaps;
    unsigned long long dy_pack, dx_pack;
    int Ca, Cb;          // output extents (real channel counts)
    int Ca_ld, Cb_ld;    // readable channels of the operand slices (multiples of 8; zero beyond the real count)
    int a_cs, a_co, b_cs, b_co;
    int atiles, btiles;
    int nsplit;
    long long rows_per_split;
    unsigned a_bytes, b_bytes;
    unsigned a_plane, b_plane;  // bf16x3 form: byte distance between the operand planes (0 otherwise)
    // grouped form (rdpn6d_wgrad_bf16_group): ngroup problems of ONE geometry in one launch - the same-shaped convolutions of a
    // ResNet stage.  Workgroup -> (tile, split, problem); partial = [problem][split][Ca][ntaps][Cb]; A / Bg = Ag[0] / Bgg[0]
    int ngroup;
    const unsigned short* Ag[RD_WGRAD_MAX_GROUP];
    const unsigned short* Bgg[RD_WGRAD_MAX_GROUP];
};

typedef short rd_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 rd_bf16x8;

__device__ __forceinline__ rd_s16x4 lds_tr_b64(const void* p)
{
    rd_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p));
    return v;
}

// PL = 3: bf16x3 form (fp32-accurate, see conv_igemm_bf16x3.hip): both operands as three bf16 planes (plane strides
// a_plane / b_plane bytes), six MFMAs per k16 step and tile pair; one k16 step (16 pixels) per chunk so that the three
// stages x three planes of a 128x128 tile fit 72 KiB (two workgroups per CU).
template <int BA, int BB, int PL, int KS = (PL == 1 ? 2 : 1)>  // KS: k16 MFMA steps per chunk
__global__ __launch_bounds__(256, (PL == 1 && BA <= 128) ? 3 : 2) void wgrad_bf16_kernel(const WgradBArgs a)
{
    constexpr int KP = 16 * KS;                  // pixels per chunk
    constexpr int TA = BA / 64, TB = BB / 64;    // 32x32 tiles per wave (2x2 waves)
    constexpr int NST = 3;
    constexpr int ARB = BA * 2, BRB = BB * 2;    // row bytes
    extern __shared__ __attribute__((aligned(1024))) unsigned char wg_smem[];
    // As(stage, plane) = wg_smem + (stage*PL + plane) * KP*ARB;  Bs after all A slots
    auto As = [&](int st, int pl) { return wg_smem + (st * PL + pl) * (KP * ARB); };
    auto Bs = [&](int st, int pl) { return wg_smem + NST * PL * (KP * ARB) + (st * PL + pl) * (KP * BRB); };

    const int ntile = a.atiles * a.btiles * a.ntaps;
    const int nblk = ntile * a.nsplit * a.ngroup;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk;
    const int tile = logical % ntile;
    const int sg = logical / ntile;
    const int split = sg % a.nsplit;
    const int grp = sg / a.nsplit;
    const int at = tile % a.atiles;
    const int rest = tile / a.atiles;
    const int bt = rest % a.btiles;
    const int tap = rest / a.btiles;
    const int a0 = at * BA, b0 = bt * BB;
    const int dy = (int)((a.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((a.dx_pack >> (4 * tap)) & 15ull) - 8;
    const long long m_lo = (long long)split * a.rows_per_split;
    const long long m_hi = m_lo + a.rows_per_split < a.M ? m_lo + a.rows_per_split : a.M;
    const int nchunks = m_hi > m_lo ? (int)((m_hi - m_lo + KP - 1) / KP) : 0;

    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.Ag[grp]), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.Bgg[grp]), 0, a.b_bytes, 0x00020000);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // DMA geometry: one wave instruction = 1 KiB = AROWS pixel rows; lane -> (row, physical 16-byte chunk).  The
    // chunk a lane FETCHES is the physical chunk XOR swz(row): swz = 4 * ((row / rows-per-256-bytes) mod (chunks/4)).
    constexpr int ACH = ARB / 16, BCH = BRB / 16;                 // chunks per row (16 | 8)
    constexpr int AROWS = 64 / ACH, BROWS = 64 / BCH;             // rows per piece (4 | 8)
    constexpr int APIECES = KP / AROWS / 4, BPIECES = KP / BROWS / 4;
    static_assert(APIECES >= 1 && BPIECES >= 1, "chunk / tile layout");
    auto swz = [](int row, int ch) { return ch >= 16 ? 4 * (row & 3) : 4 * ((row >> 1) & 1); };  // (512-byte rows alias like 256-byte ones)
    const int ar = lane / ACH, aq = lane % ACH, br = lane / BCH, bq = lane % BCH;

    int b_ox[BPIECES], b_oy[BPIECES], b_bi[BPIECES];
#pragma unroll
    for (int p = 0; p < BPIECES; ++p) {
        const long long m = m_lo + (wave + 4 * p) * BROWS + br;
        const int mm = m < a.M ? (int)m : 0;
        b_bi[p] = mm / a.HaWa;
        const int rem = mm - b_bi[p] * a.HaWa;
        b_oy[p] = rem / a.Wa;
        b_ox[p] = rem - b_oy[p] * a.Wa;
    }
    const unsigned a_row_bytes = (unsigned)a.a_cs * 2u, b_px_bytes = (unsigned)a.b_cs * 2u;
    long long ld_m = m_lo;
    auto stage_chunk = [&](const int st) {
#pragma unroll
        for (int p = 0; p < APIECES; ++p) {
            const int row0 = (wave + 4 * p) * AROWS;
            const int row = row0 + ar;
            const int lq = aq ^ swz(row, ACH);                      // logical chunk fetched into physical slot aq
            const long long m = ld_m + row;
            const bool ok = m < m_hi && a0 + lq * 8 < a.Ca_ld;
            const unsigned off = ok ? (unsigned)m * a_row_bytes + (unsigned)(a.a_co + a0 + lq * 8) * 2u : 0xFFFFFF00u;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)  // the plane offset rides on the scalar offset: an out-of-range lane stays out of range
                __builtin_amdgcn_raw_ptr_buffer_load_lds(asrc, (lds_ptr_t)(As(st, pl) + row0 * ARB), 16, (int)off, (int)(pl * a.a_plane), 0, 0);
        }
#pragma unroll
        for (int p = 0; p < BPIECES; ++p) {
            const int row0 = (wave + 4 * p) * BROWS;
            const int row = row0 + br;
            const int lq = bq ^ swz(row, BCH);
            const long long m = ld_m + row;
            const int iy = b_oy[p] * a.stride + dy, ix = b_ox[p] * a.stride + dx;
            const bool ok = m < m_hi && b0 + lq * 8 < a.Cb_ld && (unsigned)iy < (unsigned)a.Hb && (unsigned)ix < (unsigned)a.Wb;
            const unsigned off = ok ? (unsigned)((b_bi[p] * a.Hb + iy) * a.Wb + ix) * b_px_bytes + (unsigned)(a.b_co + b0 + lq * 8) * 2u
                                    : 0xFFFFFF00u;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(bsrc, (lds_ptr_t)(Bs(st, pl) + row0 * BRB), 16, (int)off, (int)(pl * a.b_plane), 0, 0);
            b_ox[p] += KP;
            while (b_ox[p] >= a.Wa) { b_ox[p] -= a.Wa; if (++b_oy[p] == a.Ha) { b_oy[p] = 0; ++b_bi[p]; } }
        }
        ld_m += KP;
    };

    const int wa = wave >> 1, wb = wave & 1;
    // transpose-read geometry: 16-lane group g: channel block (g & 1) * 16 of the 32-wide MFMA tile, k-block kb = g >> 1;
    // lane L of the group addresses pixel row kb*8 + L/4 (+4 for the second read), channels (L%4)*4.. of its block
    const int g = lane >> 4, L = lane & 15;
    const int trow = (g >> 1) * 8 + (L >> 2);
    const int tcol = (g & 1) * 16 + (L & 3) * 4;  // channel offset inside the 32-channel tile
    auto frag_addr = [&](const unsigned char* base, int rowbytes, int nch, int row, int ch) {
        const int chunk = (ch >> 3) ^ swz(row, nch);
        return base + row * rowbytes + chunk * 16 + (ch & 7) * 2;
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragments of one chunk, flat: plane pl, A piece (tile i, k16 step s, half h) at pl*NFP + (i*KS+s)*2+h, B pieces after
    // the plane's A pieces
    constexpr int NFP = (TA + TB) * 2 * KS;  // per plane
    constexpr int NF = NFP * PL;
    auto read_frags = [&](int st, rd_s16x4 (&f)[NF]) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = s * 16 + trow + h * 4;
#pragma unroll
                    for (int i = 0; i < TA; ++i)
                        f[pl * NFP + (i * KS + s) * 2 + h] = lds_tr_b64(frag_addr(As(st, pl), ARB, ACH, row, wa * (BA / 2) + i * 32 + tcol));
#pragma unroll
                    for (int j = 0; j < TB; ++j)
                        f[pl * NFP + TA * 2 * KS + (j * KS + s) * 2 + h] =
                            lds_tr_b64(frag_addr(Bs(st, pl), BRB, BCH, row, wb * (BB / 2) + j * 32 + tcol));
                }
    };
    // The transpose reads are inline asm, invisible to the compiler's wait-count insertion: one s_waitcnt that formally
    // (re)defines every fragment register, so that no consumer can be scheduled ahead of it.
    auto settle = [&](rd_s16x4 (&f)[NF]) {
        if constexpr (NF == 24)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]),
                           "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]), "+v"(f[16]),
                           "+v"(f[17]), "+v"(f[18]), "+v"(f[19]), "+v"(f[20]), "+v"(f[21]), "+v"(f[22]), "+v"(f[23]));
        else if constexpr (NF == 16)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]),
                           "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
        else if constexpr (NF == 12)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]),
                           "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    };
    auto mma = [&](const rd_s16x4 (&f)[NF]) {
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        constexpr int NPR = PL == 1 ? 1 : 6;
        constexpr int PA[6] = {PL == 1 ? 0 : 2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // plane pairs, smallest terms first
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
                for (int i = 0; i < TA; ++i) {
                    const int fa = PA[pr] * NFP + (i * KS + s) * 2;
                    const s16x8 av = __builtin_shufflevector(f[fa], f[fa + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        const int fb = PB[pr] * NFP + TA * 2 * KS + (j * KS + s) * 2;
                        const s16x8 bv = __builtin_shufflevector(f[fb], f[fb + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[i][j] = RD_LP_MFMA_32x32x16(av, bv, acc[i][j]);
                    }
                }
    };

    if (nchunks > 0) {
        // The DMA of the chunk staged in a step stays in flight across that step's barrier: the step ends with a counted
        // s_waitcnt vmcnt(<DMAs of one chunk>) - the chunk staged one step earlier has landed - and a raw s_barrier
        // (a __syncthreads would drain the queue and expose the L2 -> LDS round trip in every step).
        constexpr int NDMA = (APIECES + BPIECES) * PL;
        auto publish = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        stage_chunk(0);
        stage_chunk(1);
        publish();  // chunk 0 has landed; chunk 1 may still be in flight
        rd_s16x4 f0[NF], f1[NF];
        read_frags(0, f0);
        settle(f0);
        int st_next = 1, st_stage = 2;
        const int npairs = nchunks >> 1;
        for (int pr = 0; pr < npairs; ++pr) {
            // step: DMA of the chunk after next, wait for the next chunk (staged one step ago) + barrier, fragment reads of
            // the next chunk (async), MFMAs of the current chunk, settle the reads
            stage_chunk(st_stage);
            publish();
            read_frags(st_next, f1);
            mma(f0);
            settle(f1);
            st_next = st_next == NST - 1 ? 0 : st_next + 1;
            st_stage = st_stage == NST - 1 ? 0 : st_stage + 1;

            stage_chunk(st_stage);
            publish();
            read_frags(st_next, f0);
            mma(f1);
            settle(f0);
            st_next = st_next == NST - 1 ? 0 : st_next + 1;
            st_stage = st_stage == NST - 1 ? 0 : st_stage + 1;
        }
        if (nchunks & 1) mma(f0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    const int frow = lane & 31, hi = lane >> 5;
    float* po = a.partial + ((long long)grp * a.nsplit + split) * a.Ca * a.ntaps * a.Cb;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int bch = b0 + wb * (BB / 2) + j * 32 + frow;
        if (bch >= a.Cb) continue;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ach = a0 + wa * (BA / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (ach < a.Ca) po[((long long)ach * a.ntaps + tap) * a.Cb + bch] = acc[i][j][e];
            }
    }
}

// 64 elements x 4 split lanes per workgroup: lane j adds the splits j, j + 4, ... (four loads in flight), the four lanes are combined
// in a fixed order.  (One thread per element walking all S splits was a chain of S dependent-address loads on a handful of
// workgroups: 61 us for ConvPnPNet's first layer - 10 240 outputs - where the partials are 10 MB.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int S, long long n, float* __restrict__ out,
                                                            int nsl)
{
    __shared__ float sm[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 64; base < n; base += (long long)gridDim.x * 64) {
        const long long i = base + e;
        float s = 0.f;
        if (i < n && sl < nsl) {
            const float* p = partial + i;
            int k = sl;
            for (; k + 3 * nsl < S; k += 4 * nsl) {
                const float a0 = p[(long long)k * n], a1 = p[(long long)(k + nsl) * n], a2 = p[(long long)(k + 2 * nsl) * n],
                            a3 = p[(long long)(k + 3 * nsl) * n];
                s = (((s + a0) + a1) + a2) + a3;
            }
            for (; k < S; k += nsl) s += p[(long long)k * n];
        }
        sm[sl][e] = s;
        __syncthreads();
        if (sl == 0 && i < n) out[i] = nsl == 1 ? sm[0][e] : (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
        __syncthreads();
    }
}

// the same fixed-order sum, scattered straight into the parameter's own gradient layout:
// out[a*sa + t*st + b*sb] for a < Ca_out, b < Cb_out (e.g. OIHW: sa = Cin*k*k, sb = k*k, st = 1)
struct WgradOut {
    float* out;
    long long sa, st, sb;
    int Ca_out, Cb_out;
};
// four consecutive b per thread (Cb % 4 == 0: 16-byte loads of every split's partial), 64 such quads x 4 split lanes per workgroup
// (see splitk_reduce_kernel); fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_strided_kernel(const float* __restrict__ partial, int S, int Ca, int ntaps, int Cb,
                                                                    WgradOut o, int nsl)
{
    __shared__ f32x4 sm[4][64];
    const long long n = (long long)Ca * ntaps * Cb;
    const long long n4 = n >> 2;
    const int cb4 = Cb >> 2;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 64; base < n4; base += (long long)gridDim.x * 64) {
        const long long i4 = base + e;
        int a = 0, b = 0, t = 0;
        bool ok = i4 < n4;
        if (ok) {
            b = (int)(i4 % cb4) * 4;
            const long long r = i4 / cb4;
            t = (int)(r % ntaps);
            a = (int)(r / ntaps);
            ok = a < o.Ca_out && b < o.Cb_out;
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (ok && sl < nsl) {
            const float* p = partial + i4 * 4;
            int k = sl;
            for (; k + 3 * nsl < S; k += 4 * nsl) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (long long)k * n);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (long long)(k + nsl) * n);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 2 * nsl) * n);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 3 * nsl) * n);
                s = (((s + v0) + v1) + v2) + v3;
            }
            for (; k < S; k += nsl) s = s + *reinterpret_cast<const f32x4*>(p + (long long)k * n);
        }
        sm[sl][e] = s;
        __syncthreads();
        if (sl == 0 && ok) {
            const f32x4 r = nsl == 1 ? sm[0][e] : (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
            float* q = o.out + a * o.sa + t * o.st + b * o.sb;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (b + c < o.Cb_out) q[c * o.sb] = r[c];
        }
        __syncthreads();
    }
}

// The OIHW case of the scatter (st == 1, sb == ntaps > 1: row a of the gradient is one contiguous run of Cb_out * ntaps floats, element
// (t, b) at b * ntaps + t): a workgroup sums a [ntaps][BC] panel of row a with the same 16-byte loads and the same fixed order, turns
// it through LDS and writes the run with consecutive lanes on consecutive addresses (the strided form writes 4-byte pieces 36 bytes apart).
struct WgradGroupOut {
    float* outs[RD_WGRAD_MAX_GROUP];  // problem blockIdx.z of a grouped launch writes outs[blockIdx.z]; outs[0] == nullptr: o.out
};
template <int BC>
__global__ __launch_bounds__(256) void splitk_reduce_oihw_kernel(const float* __restrict__ partial, int S, int Ca, int ntaps, int Cb, WgradOut o,
                                                                 const WgradGroupOut go)
{
    __shared__ float s_v[9][BC + 1];
    const int a = blockIdx.y, b0 = blockIdx.x * BC;
    const long long n = (long long)Ca * ntaps * Cb;
    if (go.outs[0]) {
        partial += (long long)blockIdx.z * S * n;
        o.out = go.outs[blockIdx.z];
    }
    const int nb = Cb - b0 < BC ? Cb - b0 : BC;  // multiple of 4
    const int nb4 = nb >> 2;
    for (int task = threadIdx.x; task < ntaps * nb4; task += 256) {
        const int t = task / nb4, b = (task - t * nb4) * 4;
        const float* p = partial + ((long long)a * ntaps + t) * Cb + b0 + b;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 4 <= S; k += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (long long)k * n);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 1) * n);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 2) * n);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 3) * n);
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; k < S; ++k) s = s + *reinterpret_cast<const f32x4*>(p + (long long)k * n);
#pragma unroll
        for (int e = 0; e < 4; ++e) s_v[t][b + e] = s[e];
    }
    __syncthreads();
    const int nout = (o.Cb_out - b0 < nb ? o.Cb_out - b0 : nb) * ntaps;  // real input channels of this panel
    float* q = o.out + a * o.sa + (long long)b0 * ntaps;
    for (int j = threadIdx.x; j < nout; j += 256) {
        const int b = j / ntaps, t = j - b * ntaps;
        q[j] = s_v[t][b];
    }
}

static bool wgrad_reduce_is_oihw(const WgradOut* o, int Ca, int ntaps, int Cb)
{
    return o && o->st == 1 && o->sb == ntaps && ntaps > 1 && ntaps <= 9 && Cb % 4 == 0 && o->Ca_out <= Ca;
}
static void wgrad_reduce(const float* partial, int S, int Ca, int ntaps, int Cb, float* out, const WgradOut* o, hipStream_t s,
                         const WgradGroupOut* go = nullptr, int G = 1)
{
    if (wgrad_reduce_is_oihw(o, Ca, ntaps, Cb)) {
        WgradGroupOut g;
        if (go) g = *go;
        else g.outs[0] = nullptr;
        if (Cb >= 128)
            hipLaunchKernelGGL(splitk_reduce_oihw_kernel<128>, dim3((Cb + 127) / 128, o->Ca_out, G), dim3(256), 0, s, partial, S, Ca, ntaps, Cb, *o, g);
        else
            hipLaunchKernelGGL(splitk_reduce_oihw_kernel<64>, dim3((Cb + 63) / 64, o->Ca_out, G), dim3(256), 0, s, partial, S, Ca, ntaps, Cb, *o, g);
        return;
    }
    const long long n = (long long)Ca * ntaps * Cb;
    const int rblocks = (int)((n + 63) / 64 < 8192 ? (n + 63) / 64 : 8192);  // 64 elements (quads) x 4 split lanes per workgroup
    const int rblocks4 = (int)((n / 4 + 63) / 64 < 8192 ? (n / 4 + 63) / 64 : 8192);
    static const int nsl = getenv("RDPN6D_REDUCE_LANES") ? atoi(getenv("RDPN6D_REDUCE_LANES")) : 4;  // 1 | 2 | 4 (profiling)
    if (o) hipLaunchKernelGGL(splitk_reduce_strided_kernel, dim3(rblocks4), dim3(256), 0, s, partial, S, Ca, ntaps, Cb, *o, nsl);
    else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rblocks), dim3(256), 0, s, partial, S, n, out, nsl);
}

// A [Bn*Ha*Wa rows, a_cs] ; Bg NHWC [Bn,Hb,Wb,b_cs]; out [Ca][ntaps][Cb] fp32;
// partial = scratch of at least rdpn6d_wgrad_scratch_floats(...) floats
extern "C" long long rdpn6d_wgrad_scratch_floats(int Bn, int Ha, int Wa, int Ca, int Cb, int ntaps);

// Split-K factor: fill an integral number of "rounds" of resident workgroups (256 CUs x blocks/CU that the
// tile variant's registers allow) - a grid of 1044 workgroups on 512 slots would run 3 rounds at 68 % utilisation.
static int wgrad_pick_splits(long long M, int tiles, int ba, int bb)
{
    const int occ = (ba == 128 && bb == 128) ? 3 : ((ba == 64 && bb == 64) ? 5 : 4);
    const long long slots = 256LL * occ;
    const long long maxs = (M + 255) / 256 > 0 ? (M + 255) / 256 : 1;  // >= 16 chunks of 16 pixels per split
    long long best = 1;
    double best_util = 0.0;
    for (int r = 1; r <= 4; ++r) {
        long long s = slots * r / tiles;
        if (s < 1) continue;
        if (s > maxs) s = maxs;
        if (s > 256) s = 256;
        const long long blocks = s * tiles;
        const long long rounds = (blocks + slots - 1) / slots;
        const double util = (double)blocks / (double)(rounds * slots);
        if (util > best_util + 1e-9) { best_util = util; best = s; }
    }
    return (int)best;
}

extern "C" long long rdpn6d_wgrad_scratch_floats(int Bn, int Ha, int Wa, int Ca, int Cb, int ntaps)
{
    const int ba = Ca > 64 ? 128 : 64, bb = Cb > 64 ? 128 : 64;
    const int tiles = ((Ca + ba - 1) / ba) * ((Cb + bb - 1) / bb) * ntaps;
    return (long long)wgrad_pick_splits((long long)Bn * Ha * Wa, tiles, ba, bb) * Ca * ntaps * Cb;
}

static int wgrad_f32_impl(const float* A, int a_cs, int a_co, int Ca, const float* Bg, int b_cs, int b_co, int Cb, int Bn,
                          int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx, float* out,
                          const WgradOut* so, float* partial, void* stream)
{
    RD_REQUIRE(A && Bg && out && partial && dy && dx, "null pointer");
    RD_REQUIRE(Bn > 0 && Ha > 0 && Wa > 0 && Hb > 0 && Wb > 0 && stride >= 1, "shape");
    RD_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "channel counts must be multiples of 4");
    RD_REQUIRE(a_cs % 4 == 0 && a_co % 4 == 0 && b_cs % 4 == 0 && b_co % 4 == 0, "channel slices must be 16-byte aligned");
    RD_REQUIRE(a_co + Ca <= a_cs && b_co + Cb <= b_cs, "channel slices");
    RD_REQUIRE(ntaps >= 1 && ntaps <= 9, "ntaps in 1..9");
    WgradKArgs a;
    a.A = A; a.Bg = Bg; a.partial = partial;
    a.M = (long long)Bn * Ha * Wa;
    RD_REQUIRE(a.M < (1LL << 31), "pixel count must fit 31 bits");
    a.Ha = Ha; a.Wa = Wa; a.HaWa = Ha * Wa; a.Hb = Hb; a.Wb = Wb; a.stride = stride; a.ntaps = ntaps;
    a.dy_pack = a.dx_pack = 0;
    for (int t = 0; t < ntaps; ++t) {
        RD_REQUIRE(dy[t] >= -8 && dy[t] <= 7 && dx[t] >= -8 && dx[t] <= 7, "tap offsets must be in -8..7");
        a.dy_pack |= (unsigned long long)(dy[t] + 8) << (4 * t);
        a.dx_pack |= (unsigned long long)(dx[t] + 8) << (4 * t);
    }
    a.Ca = Ca; a.Cb = Cb; a.a_cs = a_cs; a.a_co = a_co; a.b_cs = b_cs; a.b_co = b_co;
    const long long ab = a.M * a_cs * 4, bb_ = (long long)Bn * Hb * Wb * b_cs * 4;
    RD_REQUIRE(ab < (1LL << 32) - 64 && bb_ < (1LL << 32) - 64, "operands must be smaller than 4 GiB (32-bit buffer offsets)");
    a.a_bytes = (unsigned)ab; a.b_bytes = (unsigned)bb_;
    const int ba = Ca > 64 ? 128 : 64, bb = Cb > 64 ? 128 : 64;
    a.atiles = (Ca + ba - 1) / ba;
    a.btiles = (Cb + bb - 1) / bb;
    const int tiles = a.atiles * a.btiles * ntaps;
    const int S = wgrad_pick_splits(a.M, tiles, ba, bb);
    a.rows_per_split = ((a.M + S - 1) / S + 15) / 16 * 16;
    hipStream_t s = (hipStream_t)stream;
    a.nsplit = S;
    dim3 grid(tiles * S), block(256);
    if (ba == 128 && bb == 128) hipLaunchKernelGGL((wgrad_f32_kernel<128, 128>), grid, block, 0, s, a);
    else if (ba == 128) hipLaunchKernelGGL((wgrad_f32_kernel<128, 64>), grid, block, 0, s, a);
    else if (bb == 128) hipLaunchKernelGGL((wgrad_f32_kernel<64, 128>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wgrad_f32_kernel<64, 64>), grid, block, 0, s, a);
    RD_LAUNCH_CHECK();
    wgrad_reduce(partial, S, Ca, ntaps, Cb, out, so, s);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

extern "C" int rdpn6d_wgrad_f32(const float* A, int a_cs, int a_co, int Ca, const float* Bg, int b_cs, int b_co, int Cb,
                                int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                                const int* dx, float* out, float* partial, void* stream)
{
    return wgrad_f32_impl(A, a_cs, a_co, Ca, Bg, b_cs, b_co, Cb, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx, out, nullptr, partial,
                          stream);
}

// output scattered into a caller-defined layout: element (a, tap, b) -> out[a*sa + tap*st + b*sb], a < Ca_out, b < Cb_out
extern "C" int rdpn6d_wgrad_f32_strided(const float* A, int a_cs, int a_co, int Ca, const float* Bg, int b_cs, int b_co, int Cb,
                                        int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                                        const int* dx, float* out, long long sa, long long st, long long sb, int Ca_out,
                                        int Cb_out, float* partial, void* stream)
{
    RD_REQUIRE(Ca_out > 0 && Ca_out <= Ca && Cb_out > 0 && Cb_out <= Cb, "output extents");
    const WgradOut so = {out, sa, st, sb, Ca_out, Cb_out};
    return wgrad_f32_impl(A, a_cs, a_co, Ca, Bg, b_cs, b_co, Cb, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx, out, &so, partial, stream);
}

// bf16 operands (compact NHWC copies, channel strides/offsets in elements, multiples of 8); Ca_ld / Cb_ld = readable
// channels of the slices (>= Ca / Cb, zero beyond the real count); out / partial as rdpn6d_wgrad_f32
template <int BA, int BB, int PL, int KS = (PL == 1 ? 2 : 1)>
static int wgrad_bf16_launch(const WgradBArgs& a, dim3 grid, hipStream_t s)
{
    constexpr int lds = 3 * PL * (16 * KS) * (BA + BB) * 2;
    auto kern = wgrad_bf16_kernel<BA, BB, PL, KS>;
    if (lds > 64 * 1024) {
        RD_LDS_OPT_IN(kern, lds);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return RDPN6D_OK;
}

// a_plane_elems / b_plane_elems > 0: bf16x3 form - A and Bg are plane 0 of three bf16 planes (128x128 tiles only)
static int wgrad_bf16_impl(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co, int Cb,
                           int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx,
                           float* out, const WgradOut* so, float* partial, void* stream, long long a_plane_elems = 0,
                           long long b_plane_elems = 0, int G = 1, const void* const* A_list = nullptr, const void* const* B_list = nullptr,
                           float* const* out_list = nullptr, long long partial_floats = 0)
{
    RD_REQUIRE(A && Bg && out && partial && dy && dx, "null pointer");
    RD_REQUIRE(Bn > 0 && Ha > 0 && Wa > 0 && Hb > 0 && Wb > 0 && stride >= 1, "shape");
    RD_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "channel counts must be multiples of 4");
    RD_REQUIRE(Ca_ld >= Ca && Cb_ld >= Cb && Ca_ld % 8 == 0 && Cb_ld % 8 == 0, "readable channel counts: multiples of 8 >= the real ones");
    RD_REQUIRE(a_cs % 8 == 0 && a_co % 8 == 0 && b_cs % 8 == 0 && b_co % 8 == 0, "channel slices must be 16-byte aligned");
    RD_REQUIRE(a_co + Ca_ld <= a_cs && b_co + Cb_ld <= b_cs, "channel slices");
    RD_REQUIRE(ntaps >= 1 && ntaps <= 9, "ntaps in 1..9");
    WgradBArgs a;
    a.A = (const unsigned short*)A; a.Bg = (const unsigned short*)Bg; a.partial = partial;
    a.ngroup = G;
    WgradGroupOut gout;
    gout.outs[0] = nullptr;
    for (int g = 0; g < RD_WGRAD_MAX_GROUP; ++g) {
        a.Ag[g] = (const unsigned short*)(G > 1 && g < G ? A_list[g] : A);
        a.Bgg[g] = (const unsigned short*)(G > 1 && g < G ? B_list[g] : Bg);
        if (G > 1) gout.outs[g] = g < G ? out_list[g] : out_list[0];
    }
    a.M = (long long)Bn * Ha * Wa;
    RD_REQUIRE(a.M < (1LL << 31), "pixel count must fit 31 bits");
    a.Ha = Ha; a.Wa = Wa; a.HaWa = Ha * Wa; a.Hb = Hb; a.Wb = Wb; a.stride = stride; a.ntaps = ntaps;
    a.dy_pack = a.dx_pack = 0;
    for (int t = 0; t < ntaps; ++t) {
        RD_REQUIRE(dy[t] >= -8 && dy[t] <= 7 && dx[t] >= -8 && dx[t] <= 7, "tap offsets must be in -8..7");
        a.dy_pack |= (unsigned long long)(dy[t] + 8) << (4 * t);
        a.dx_pack |= (unsigned long long)(dx[t] + 8) << (4 * t);
    }
    a.Ca = Ca; a.Cb = Cb; a.Ca_ld = Ca_ld; a.Cb_ld = Cb_ld; a.a_cs = a_cs; a.a_co = a_co; a.b_cs = b_cs; a.b_co = b_co;
    const bool x3 = a_plane_elems > 0 || b_plane_elems > 0;
    long long ab = a.M * a_cs * 2, bb_ = (long long)Bn * Hb * Wb * b_cs * 2;
    if (x3) {
        RD_REQUIRE(a_plane_elems * 2 >= ab && b_plane_elems * 2 >= bb_ && a_plane_elems % 8 == 0 && b_plane_elems % 8 == 0, "plane sizes");
        RD_REQUIRE(Ca > 64 && Cb > 64, "the bf16x3 weight gradient has 128x128 tiles only");
        ab += 4 * a_plane_elems;
        bb_ += 4 * b_plane_elems;
    }
    RD_REQUIRE(ab < (1LL << 32) - 256 && bb_ < (1LL << 32) - 256, "operands must be smaller than 4 GiB (32-bit buffer offsets)");
    a.a_bytes = (unsigned)ab; a.b_bytes = (unsigned)bb_;
    a.a_plane = (unsigned)(a_plane_elems * 2); a.b_plane = (unsigned)(b_plane_elems * 2);
    int ba = Ca > 64 ? 128 : 64;
    const int bb = Cb > 64 ? 128 : 64;
    a.atiles = (Ca + ba - 1) / ba;
    a.btiles = (Cb + bb - 1) / bb;
    int tiles = a.atiles * a.btiles * ntaps;
    // (grouped: the split count that fills the chip with ALL the problems' tiles - a fraction of what each problem alone would take)
    int S = wgrad_pick_splits(a.M, tiles * G, ba, bb);  // same split count (and scratch size) as the fp32 form
    // 256 x 128 tile (2x2 waves of 128 x 64, one k16 step per chunk) for the wide layers with many pixels - the head's 256 -> 256 and
    // the ConvTranspose: a quarter fewer bytes through LDS-DMA per FLOP (the launch moves ~2.4 GB L2 -> LDS, > 10 TB/s), 12 transpose
    // reads per 8 MFMAs instead of 8 per 4, half the partial tiles: head layer at B = 32 247 -> 228 us, 1x1 512 -> 256 at 32^2 46 -> 42;
    // with few pixels (layer3 / layer4: 8 192 / 2 048) the halved workgroup count costs more (43 -> 47 us).  Never more splits than
    // the scratch was sized for.
    static const bool wide_off = getenv("RDPN6D_WGRAD_WIDE") && atoi(getenv("RDPN6D_WGRAD_WIDE")) == 0;  // profiling
    if (!x3 && !wide_off && G == 1 && Ca % 256 == 0 && bb == 128 && a.M >= 32768) {
        const int t2 = (Ca / 256) * a.btiles * ntaps;
        // the FEWEST splits that fill whole rounds of the 512 resident workgroups to >= 97 % (every split costs one partial tile
        // written and read again by the reduce: 2.4 MB for a head layer), else the best fill
        int best = 1;
        double best_util = 0.0;
        for (int sp = 1; sp <= S; ++sp) {
            if ((a.M + sp - 1) / sp < 512) break;  // >= 32 chunks of 16 pixels per split
            const long long blocks = (long long)sp * t2, rounds = (blocks + 511) / 512;
            const double util = (double)blocks / (double)(rounds * 512);
            if (util > best_util + 1e-9) { best_util = util; best = sp; }
            if (util >= 0.97) break;
        }
        ba = 256;
        a.atiles = Ca / 256;
        tiles = t2;
        S = best;
    }
    a.rows_per_split = ((a.M + S - 1) / S + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    a.nsplit = S;
    if (G > 1) {
        RD_REQUIRE(wgrad_reduce_is_oihw(so, Ca, ntaps, Cb), "grouped weight gradient: OIHW targets of a k x k convolution (k > 1)");
        RD_REQUIRE((long long)G * S * Ca * ntaps * Cb <= partial_floats, "grouped weight gradient: scratch too small (rdpn6d_wgrad_group_scratch_floats)");
    }
    dim3 grid(tiles * S * G);
    int rc;
    if (x3) rc = wgrad_bf16_launch<128, 128, 3>(a, grid, s);
    else if (ba == 256) rc = wgrad_bf16_launch<256, 128, 1, 1>(a, grid, s);  // (two k16 steps per chunk spill: 269 vs 228 us)
    else if (ba == 128 && bb == 128) rc = wgrad_bf16_launch<128, 128, 1>(a, grid, s);
    else if (ba == 128) rc = wgrad_bf16_launch<128, 64, 1>(a, grid, s);
    else if (bb == 128) rc = wgrad_bf16_launch<64, 128, 1>(a, grid, s);
    else rc = wgrad_bf16_launch<64, 64, 1>(a, grid, s);
    if (rc != RDPN6D_OK) return rc;
    RD_LAUNCH_CHECK();
    wgrad_reduce(partial, S, Ca, ntaps, Cb, out, so, s, G > 1 ? &gout : nullptr, G);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}

// G weight gradients of ONE geometry (the same-shaped 3x3 convolutions of a ResNet stage) in one launch + one reduce: the chip is
// filled by the problems' tiles instead of by K-splits of each (layer3 at B = 32: 36 tiles x 14 splits per convolution -> 396 tiles
// x 7 splits for eleven), i.e. longer K loops, a fraction of the partial-sum traffic and two launches instead of 2 G.
// A_list / B_list / out_list: HOST arrays of G device pointers (gradient w.r.t. the output, input activation, OIHW gradient).
extern "C" long long rdpn6d_wgrad_group_scratch_floats(int G, int Bn, int Ha, int Wa, int Ca, int Cb, int ntaps)
{
    const int ba = Ca > 64 ? 128 : 64, bb = Cb > 64 ? 128 : 64;
    const int tiles = ((Ca + ba - 1) / ba) * ((Cb + bb - 1) / bb) * ntaps;
    return (long long)G * wgrad_pick_splits((long long)Bn * Ha * Wa, tiles * G, ba, bb) * Ca * ntaps * Cb;
}

extern "C" int rdpn6d_wgrad_bf16_group(int G, const void* const* A_list, int a_cs, int a_co, int Ca, int Ca_ld, const void* const* B_list,
                                       int b_cs, int b_co, int Cb, int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps,
                                       const int* dy, const int* dx, float* const* out_list, long long sa, long long st, long long sb,
                                       int Ca_out, int Cb_out, float* partial, long long partial_floats, void* stream)
{
    RD_REQUIRE(G >= 1 && G <= RD_WGRAD_MAX_GROUP && A_list && B_list && out_list, "grouped weight gradient: 1..16 problems");
    for (int g = 0; g < G; ++g) RD_REQUIRE(A_list[g] && B_list[g] && out_list[g], "grouped weight gradient: null pointer");
    RD_REQUIRE(Ca_out > 0 && Ca_out <= Ca && Cb_out > 0 && Cb_out <= Cb, "output extents");
    const WgradOut so = {out_list[0], sa, st, sb, Ca_out, Cb_out};
    if (G == 1)
        return wgrad_bf16_impl(A_list[0], a_cs, a_co, Ca, Ca_ld, B_list[0], b_cs, b_co, Cb, Cb_ld, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx,
                               out_list[0], &so, partial, stream);
    return wgrad_bf16_impl(A_list[0], a_cs, a_co, Ca, Ca_ld, B_list[0], b_cs, b_co, Cb, Cb_ld, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx,
                           out_list[0], &so, partial, stream, 0, 0, G, A_list, B_list, out_list, partial_floats);
}

extern "C" int rdpn6d_wgrad_bf16(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co, int Cb,
                                 int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy,
                                 const int* dx, float* out, float* partial, void* stream)
{
    return wgrad_bf16_impl(A, a_cs, a_co, Ca, Ca_ld, Bg, b_cs, b_co, Cb, Cb_ld, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx, out,
                           nullptr, partial, stream);
}

extern "C" int rdpn6d_wgrad_bf16_strided(const void* A, int a_cs, int a_co, int Ca, int Ca_ld, const void* Bg, int b_cs, int b_co,
                                         int Cb, int Cb_ld, int Bn, int Ha, int Wa, int Hb, int Wb, int stride, int ntaps,
                                         const int* dy, const int* dx, float* out, long long sa, long long st, long long sb,
                                         int Ca_out, int Cb_out, float* partial, void* stream)
{
    RD_REQUIRE(Ca_out > 0 && Ca_out <= Ca && Cb_out > 0 && Cb_out <= Cb, "output extents");
    const WgradOut so = {out, sa, st, sb, Ca_out, Cb_out};
    return wgrad_bf16_impl(A, a_cs, a_co, Ca, Ca_ld, Bg, b_cs, b_co, Cb, Cb_ld, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx, out, &so,
                           partial, stream);
}

// bf16x3 form (fp32-accurate weight gradient on the bf16 matrix pipe, see conv_igemm_bf16x3.hip): A / Bg = plane 0 of three
// bf16 planes [3][a_plane_elems] / [3][b_plane_elems] of the NHWC gradient / activation (rdpn6d_split_bf16x3); Ca, Cb > 64
extern "C" int rdpn6d_wgrad_bf16x3_strided(const void* A, long long a_plane_elems, int a_cs, int a_co, int Ca, int Ca_ld,
                                           const void* Bg, long long b_plane_elems, int b_cs, int b_co, int Cb, int Cb_ld, int Bn,
                                           int Ha, int Wa, int Hb, int Wb, int stride, int ntaps, const int* dy, const int* dx,
                                           float* out, long long sa, long long st, long long sb, int Ca_out, int Cb_out,
                                           float* partial, void* stream)
{
    RD_REQUIRE(Ca_out > 0 && Ca_out <= Ca && Cb_out > 0 && Cb_out <= Cb, "output extents");
    RD_REQUIRE(a_plane_elems > 0 && b_plane_elems > 0, "plane sizes");
    const WgradOut so = {out, sa, st, sb, Ca_out, Cb_out};
    return wgrad_bf16_impl(A, a_cs, a_co, Ca, Ca_ld, Bg, b_cs, b_co, Cb, Cb_ld, Bn, Ha, Wa, Hb, Wb, stride, ntaps, dy, dx, out, &so,
                           partial, stream, a_plane_elems, b_plane_elems);
}
