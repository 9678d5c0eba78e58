// 3x3 / stride 1 / pad 1 convolution over 64 -> 64 channels on 64-pixel-wide maps in the h2 arithmetic (conv_igemm_h2.hip): ResNet
// layer1 (resnet_backbone.py:288-301: six BasicBlock convolutions on the [B, 64, 64, 64] map of a 256x256 crop), the one trunk stage
// whose K loop (18 chunks of 32 channels) is too short for the tile kernels: there every workgroup spends a third of its life in
// prologue + epilogue and a K-chunk costs 72 KiB of LDS traffic for 48 MFMAs - 150 % of the LDS port against 100 % of the matrix pipe
// (DESIGN.md section 4).  This kernel removes both:
//
//   * WEIGHTS IN REGISTERS.  64 x 576 x (hi, lo) = 144 KiB: four wavefronts (one per SIMD: the whole 512-entry register file each),
//     wave (wm, wn) keeps the weights of output channels [32 wn, 32 wn + 32) for ALL eighteen (channel group, tap) chunks - 72 x 16
//     bytes = 288 registers, loaded once per workgroup lifetime.  No weight ever goes through LDS.
//   * ACTIVATION PATCH RESIDENT IN LDS.  A tile = two output rows of one crop (128 pixels x 64 channels); its input patch - four rows
//     x 66 pixels (zero halo by the buffer rule) x 256 bytes - is staged ONCE by LDS-DMA (66 KiB) and the nine taps are nine offsets
//     into it: 8 x less LDS write traffic than re-staging every (tap, chunk).  Per chunk a wave reads 8 KiB of fragments (conflict-free:
//     16-slot XOR swizzle over the pixel column) for 12 MFMAs (384 matrix-pipe cycles): the LDS port is a third busy.
//   * PERSISTENT, DOUBLE-BUFFERED.  One workgroup per CU walks a contiguous range of tiles; the patch of tile t + 1 lands in the
//     second buffer while tile t is computed - no HBM burst at the start of every workgroup, one barrier per tile.
//   * SAME BITS AS THE TILE KERNEL.  Every output is ONE accumulation chain over the chunks in the tile kernel's order (channel group
//     major, taps minor, six partial products per chunk in H2_PAIRS order), so the result is bit-identical to conv_h2_tile_kernel's
//     (tests/test_gpu_h2.py) and every parity figure of the network is unchanged.
#include "conv_h2_common.h"

#include <cstdlib>

#ifdef RDPN6D_PROBE
// probe build only: shader-cycle stamps of workgroup 0, per wave and tile: [top, DMA issued, MFMA loop done, vmcnt wait done, epilogue done]
__device__ unsigned long long* g_c64_probe = nullptr;
extern "C" int rdpn6d_debug_c64_probe(void* buf)
{
    RD_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_c64_probe), &buf, sizeof(buf)));
    return RDPN6D_OK;
}
#define C64_STAMP(i)                                                                                                        \
    do {                                                                                                                    \
        if (g_c64_probe && blockIdx.x == 0 && lane == 0) g_c64_probe[((t - t_begin) * 4 + wave) * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define C64_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int C64_W = 64;                       // map width this kernel is specialised for
constexpr int C64_PW = C64_W + 2;               // patch width incl. the zero halo columns
constexpr int C64_PROWS = 4;                    // patch rows: two output rows + one above + one below
constexpr int C64_PATCH = C64_PROWS * C64_PW * 256;   // 67 584 bytes
constexpr int C64_SCR_ROW = 36;                 // fp32 row stride of a [32][32] transposition / partial-sum block
constexpr int C64_BLK = 32 * C64_SCR_ROW * 4;   // 4 608 bytes
constexpr int C64_LDS = 2 * C64_PATCH + 4 * C64_BLK;   // 153 600 bytes: two patches + one private transposition block per wave

static_assert(C64_PATCH == 67584 && C64_LDS <= 160 * 1024, "LDS map");

struct C64Args {
    ConvH2Args ax;   // d.x / d.w h2 tensors, epilogue operands as for the tile kernels (y_h2, res_h2, scale, shift, act, overflow flag)
    int H;           // map height (rows per crop)
    int ntiles;      // B * H / 2
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_c64_h2_kernel(const C64Args args)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const ConvH2Args& ax = args.ax;
    const ConvBArgs& a = ax.b;
    const rdpn6d_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, half = lane >> 5;
    const int H = args.H;
    const int tiles_per_crop = H >> 1;

    // ---- this wave's weights: output channel n = 32 wn + frow, chunk q = cg * 9 + tap (the tile kernel's K order: channel group major,
    // taps minor); packed weight row [n][tap][cg] = 128 bytes [hi x 32 | lo x 32]
    u32x4 wreg[18][4];
    {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(d.w);
        const int n = wn * 32 + frow;
#pragma unroll
        for (int q = 0; q < 18; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wreg[q][j] = *reinterpret_cast<const u32x4*>(wb + (((size_t)n * 9 + (q % 9)) * 2 + (q / 9)) * 128 + ((2 * j + half) << 4));
    }
    const int nb = wn * 32;
    const float sc = d.scale ? d.scale[nb + frow] : 1.f, sh = d.shift ? d.shift[nb + frow] : 0.f;

    // ---- patch DMA: 264 pixel records of 256 bytes = 66 pieces of 1 KiB; wave w moves pieces w, w + 4, ... (16 or 17 each)
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.x_bytes, 0x00020000);
    auto stage_patch = [&](const int tile, const int buf) {
        const int b = tile / tiles_per_crop, y0 = (tile - b * tiles_per_crop) * 2;
        unsigned char* pbase = smem + buf * C64_PATCH;
        for (int k = wave; k < 66; k += 4) {
            // a piece = 4 patch pixels x 256 bytes; inside a pixel's record the sixteen 16-byte slots (channel group cg = slot >> 3) sit at
            // position slot ^ (c & 15): ds_read_b128 is serviced in groups of 16 lanes over a 256-byte bank row (MI355X_MICROARCH.md,
            // LDS) - 16 consecutive pixels then hit 16 different slots
            const int pc = k * 4 + (lane >> 4);            // patch pixel r * 66 + c
            const int r = pc / C64_PW, c = pc - r * C64_PW;
            const int y = y0 - 1 + r, x = c - 1;
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)C64_W;
            const unsigned slot = (unsigned)((lane & 15) ^ (c & 15));   // logical slot this LDS position holds: cg * 8 + s
            const unsigned off = ok ? ((unsigned)((b * H + y) * C64_W + x) * 256u + slot * 16u) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)(pbase + k * 1024), 16, (int)off, 0, 0, 0);
        }
    };

    // the same pieces, described ONCE per lane so that inside the tile loop a piece costs a handful of VALU operations and can be issued
    // between the MFMAs of a chunk (issued together at the top of a tile the 17 pieces were 3 700 cycles of every tile's 19 000):
    // pdesc[i] = (offset relative to the tile's first pixel) << 3 | column ok << 2 | patch row
    int pdesc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const int k = wave + 4 * i;
        const int pc = (k < 66 ? k : 0) * 4 + (lane >> 4);
        const int r = pc / C64_PW, c = pc - r * C64_PW;
        const int slot = (lane & 15) ^ (c & 15);
        const int rel = ((r - 1) * C64_W + (c - 1)) * 256 + slot * 16;
        pdesc[i] = (rel << 3) | (((unsigned)(c - 1) < (unsigned)C64_W) ? 4 : 0) | r;
    }
    auto issue_piece = [&](const int i, const int tile, const int buf) {   // piece i of this wave for `tile` -> patch buffer `buf`
        const int k = wave + 4 * i;
        if (k >= 66) return;   // (wave-uniform: waves 2, 3 have 16 pieces)
        const int b = tile / tiles_per_crop, y0 = (tile - b * tiles_per_crop) * 2;
        const int pd = pdesc[i];
        const int y = y0 - 1 + (pd & 3);
        const bool ok = (pd & 4) && (unsigned)y < (unsigned)H;
        const unsigned off = ok ? (unsigned)((b * H + y0) * C64_W * 256 + (pd >> 3)) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr_t)(smem + buf * C64_PATCH + k * 1024), 16, (int)off, 0, 0, 0);
    };

    const int G = gridDim.x;
    const int t_begin = (int)(((long long)blockIdx.x * args.ntiles) / G), t_end = (int)(((long long)(blockIdx.x + 1) * args.ntiles) / G);
    if (t_begin >= t_end) return;
    stage_patch(t_begin, 0);
    float* scr = reinterpret_cast<float*>(smem + 2 * C64_PATCH) + wave * (32 * C64_SCR_ROW);   // this wave's private [32][36] block
    unsigned mh[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) mh[kx] = (unsigned)((half ^ ((frow + kx) & 15)) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the first patch
    const bool res_pre = ax.res_h2 != nullptr && d.res == nullptr;
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        __builtin_amdgcn_s_barrier();                      // patch t complete (every wave waited for its pieces); the other buffer is free
        asm volatile("" ::: "memory");
        C64_STAMP(0);
        const bool more = t + 1 < t_end;
        C64_STAMP(1);
        // the residual records of this wave's 64 x 32 outputs, requested now: they land under the MFMA loop (in the epilogue each was a
        // dependent HBM round trip per 16 rows)
        const int tb = t / tiles_per_crop, ty = (t - tb * tiles_per_crop) * 2 + wm;
        f16x8 rh[2][2], rl[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (res_pre) {
                    const long long pix = ((long long)tb * H + ty) * C64_W + mb * 32 + rr * 16 + (lane >> 2);
                    const int c = d.res_co + wn * 32 + (lane & 3) * 8;
                    const _Float16* rp = reinterpret_cast<const _Float16*>(ax.res_h2) + pix * (2 * d.res_cs) + (c >> 5) * 64 + (c & 31);
                    rh[mb][rr] = *reinterpret_cast<const f16x8*>(rp);
                    rl[mb][rr] = *reinterpret_cast<const f16x8*>(rp + 32);
                } else {
                    rh[mb][rr] = f16x8{};
                    rl[mb][rr] = f16x8{};
                }
            }

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        // this wave's output row = patch row wm + 1; fragment of m-block mb, chunk q = (cg, tap (ky, kx)): pixel column c = 32 mb + frow + kx
        u32x4 fa[2][2][4];
        // LDS address of fragment (m-block mb, slot pair j) of chunk q = (cg, ky, kx): record of patch pixel (wm + ky, 32 mb + frow + kx),
        // slot (cg * 8 + 2 j + half) ^ (column & 15) = (cg * 8 + 2 j) ^ (half ^ ((frow + kx) & 15)) - per lane only three values mh[kx];
        // everything else is an immediate.  The values are re-materialised per chunk (one v_xad_u32 per read; the empty asm keeps the
        // compiler from hoisting all 144 addresses out of the tile loop into registers the weights need).
        const unsigned lanebase = (unsigned)(buf * C64_PATCH) + (unsigned)((wm * C64_PW + frow) * 256);
        auto read_frag = [&](const int q, u32x4 (&f)[2][4]) {
            const int cg = q / 9, tap = q - cg * 9, ky = tap / 3, kx = tap - ky * 3;   // (q is a compile-time constant at every call)
            unsigned mhk = mh[kx];
            asm volatile("" : "+v"(mhk));
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned addr = (mhk ^ (unsigned)((cg * 8 + 2 * j) << 4)) + lanebase + (unsigned)((ky * C64_PW + mb * 32 + kx) * 256);
                    f[mb][j] = *reinterpret_cast<const u32x4*>(smem + addr);
                }
        };
        read_frag(0, fa[0]);
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            // one wave per SIMD: the next chunk's eight fragment reads are ISSUED before this chunk's twelve MFMAs and fly under them
            // (the scheduling barriers pin that order - left alone the compiler sinks every read to just before its first use and the
            // matrix pipe waits out an LDS round trip per MFMA)
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 18) read_frag(q + 1, fa[(q + 1) & 1]);
            if (q < 17 && more) issue_piece(q, t + 1, buf ^ 1);   // the next tile's patch: one LDS-DMA piece per chunk, under the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            H2_PAIRS;
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb] = h2_mfma(fa[q & 1][mb][H2_PA[pr]], wreg[q][H2_PB[pr]], acc[mb]);
        }
        __builtin_amdgcn_sched_barrier(0);

        C64_STAMP(2);
        // every VMEM operation issued so far - the next patch's pieces, the residual records - has had the whole MFMA loop to land; the
        // wait sits HERE, in front of the stores, so that no later wait ever has to drain them (stores count in vmcnt too)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C64_STAMP(3);
        // ---- epilogue, one 32 x 32 block at a time through the wave's private scratch (LDS operations of one wave are ordered): folded
        // BatchNorm, transposition, shared h2 tail (residual record, ReLU, hi / lo split, 16-byte stores)
        const int b = tb, y = ty;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * half) * C64_SCR_ROW + frow] = acc[mb][e] * sc + sh;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = rr * 16 + (lane >> 2), c8 = (lane & 3) * 8;
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(scr + row * C64_SCR_ROW + c8);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(scr + row * C64_SCR_ROW + c8 + 4);
                float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                const long long pix = ((long long)b * H + y) * C64_W + mb * 32 + row;
                if (res_pre) h2_finish_row8_t<true>(ax, v, pix, nb + c8, rh[mb][rr], rl[mb][rr]);
                else h2_finish_row8(ax, v, pix, nb + c8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the block is read before the next one overwrites it
        }
        C64_STAMP(4);
    }
}

}  // namespace

// is this layer one the specialised kernel takes?  3x3 / stride 1 / pad 1 taps in row-major order, 64 -> 64 channels, dense NHWC h2
// tensors of a 64-pixel-wide map, linear output
extern "C" int rdpn6d_conv3x3_c64_h2_ok(const rdpn6d_conv_desc* d)
{
    if (!d || d->Cin != 64 || d->in_cs != 64 || d->in_co != 0 || d->N != 64 || d->Npad != 64 || d->out_cs != 64 || d->out_co != 0) return 0;
    if (d->ntaps != 9 || d->stride != 1 || d->W != C64_W || d->Wo != C64_W || d->Ho != d->H || d->OH != d->H || d->OW != C64_W) return 0;
    if (d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0 || (d->H & 1) || d->res) return 0;
    for (int t = 0; t < 9; ++t)
        if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1) return 0;
    static const int off = getenv("RDPN6D_NO_C64") ? 1 : 0;  // profiling
    return off ? 0 : 1;
}

extern "C" int rdpn6d_conv3x3_c64_h2(const rdpn6d_conv_desc* d, void* y_h2, const void* res_h2, int* overflow_flag, void* stream)
{
    RD_REQUIRE(d && d->x && d->w && y_h2, "null pointer");
    RD_REQUIRE(rdpn6d_conv3x3_c64_h2_ok(d), "not a 3x3 / 64 -> 64 / 64-wide layer");
    RD_REQUIRE(!res_h2 || (d->res_cs == 64 && d->res_co == 0), "h2 residual: dense 64-channel tensor");
    C64Args k;
    ConvH2Args& ax = k.ax;
    ConvBArgs& a = ax.b;
    a.d = *d;
    a.d.y = nullptr;
    a.M = (long long)d->B * d->H * C64_W;
    RD_REQUIRE(a.M < (1LL << 31) && a.M * 256 < (1LL << 32) - 64, "tensor must stay below 4 GiB");
    a.HoWo = d->H * C64_W;
    a.cchunks = 2;
    a.nk = 18;
    a.Ktot = 9 * 64;
    a.linear_out = 1;
    a.out_f32 = 0;
    a.vec_out = 1;
    a.x_bytes = (unsigned)(a.M * 256);
    a.w_bytes = 64u * 9u * 256u;
    a.mtiles = a.ntiles = 1;
    a.kper = 18;
    a.partial = nullptr;
    a.dy_pack = a.dx_pack = 0;
    ax.y_h2 = y_h2;
    ax.res_h2 = res_h2;
    ax.overflow_flag = overflow_flag;
    ax.crop_bias = nullptr;
    ax.partial = nullptr;
    ax.nsplit = 1;
    ax.mpad = 0;
    ax.fuse_w = nullptr;
    ax.fuse_scale = ax.fuse_bias = nullptr;
    ax.fuse_out = nullptr;
    ax.fuse_cs = ax.fuse_n = 0;
    k.H = d->H;
    k.ntiles = d->B * (d->H / 2);
    int dev = 0, cus = 256;
    RD_CHECK_HIP(hipGetDevice(&dev));
    RD_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = k.ntiles < cus ? k.ntiles : cus;
    RD_LDS_OPT_IN(conv3x3_c64_h2_kernel, C64_LDS);
    hipLaunchKernelGGL(conv3x3_c64_h2_kernel, dim3(grid), dim3(256), C64_LDS, (hipStream_t)stream, k);
    RD_LAUNCH_CHECK();
    return RDPN6D_OK;
}
