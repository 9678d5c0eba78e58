// Calibration probe: the K-loop structure of conv_igemm_bf16_kernel<128,128,128,2,2,2> on a PLAIN GEMM
// C[M][N] = A[M][K] * B[N][K]^T (bf16, both K-major), to separate the GEMM core from the convolution's gather / short K.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_bf16_probe.hip -o gemm_probe && ./gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm128(const unsigned short* A, const unsigned short* B, float* C, int M, int N, int K)
{
    constexpr int RB = 128, SL = 8, RPP = 8, NJ = 4, TM = BM / 64, TN = BN / 64, AG = BM / RPP / 4, BG = BN / RPP / 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * BM * RB;
    const int ntn = N / BN;
    const int nblk = (M / BM) * ntn, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, kk = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
    const int nt = logical % ntn, mt = logical / ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int prow = lane >> 3, pslot = lane & 7;
    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (unsigned)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(B), 0, (unsigned)((size_t)N * K * 2), 0x00020000);
    unsigned a_off[AG], b_off[BG];
    for (int i = 0; i < AG; ++i) { const int row = (wave + 4 * i) * RPP + prow; a_off[i] = (unsigned)(m0 + row) * K * 2u + (unsigned)((pslot ^ ((row >> 1) & 7)) * 16); }
    for (int i = 0; i < BG; ++i) { const int row = (wave + 4 * i) * RPP + prow; b_off[i] = (unsigned)(n0 + row) * K * 2u + (unsigned)((pslot ^ ((row >> 1) & 7)) * 16); }
    auto stage = [&](int kc, int st) {
        for (int i = 0; i < AG; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(asrc, (lds_ptr_t)(As + (st * BM + (wave + 4 * i) * RPP) * RB), 16, (int)(a_off[i] + kc * RB), 0, 0, 0);
        for (int i = 0; i < BG; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(bsrc, (lds_ptr_t)(Bs + (st * BN + (wave + 4 * i) * RPP) * RB), 16, (int)(b_off[i] + kc * RB), 0, 0, 0);
    };
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, half = lane >> 5;
    auto frags = [&](int st, u32x4 (&fa)[TM][NJ], u32x4 (&fb)[TN][NJ]) {
        for (int i = 0; i < TM; ++i) { const int R = wm * (BM / 2) + i * 32 + frow; const int sw = (R >> 1) & 7; const unsigned char* p = As + (st * BM + R) * RB;
            for (int j = 0; j < NJ; ++j) fa[i][j] = *(const u32x4*)(p + (((2 * j + half) ^ sw) << 4)); }
        for (int i = 0; i < TN; ++i) { const int R = wn * (BN / 2) + i * 32 + frow; const int sw = (R >> 1) & 7; const unsigned char* p = Bs + (st * BN + R) * RB;
            for (int j = 0; j < NJ; ++j) fb[i][j] = *(const u32x4*)(p + (((2 * j + half) ^ sw) << 4)); }
    };
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mma = [&](const u32x4 (&fa)[TM][NJ], const u32x4 (&fb)[TN][NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][j]), __builtin_bit_cast(bf16x8, fb[jn][j]), acc[i][jn], 0, 0, 0);
    };
    const int nk = K / 64;
    stage(0, 0); stage(1, 1);
    __syncthreads();
    u32x4 fa0[TM][NJ], fb0[TN][NJ], fa1[TM][NJ], fb1[TN][NJ];
    frags(0, fa0, fb0);
    __syncthreads();
    int kc = 2;
    for (int pr = 0; pr < nk / 2; ++pr) {
        stage(kc < nk ? kc : nk - 1, 0); ++kc;
        frags(1, fa1, fb1);
        mma(fa0, fb0);
        __syncthreads();
        stage(kc < nk ? kc : nk - 1, 1); ++kc;
        frags(0, fa0, fb0);
        mma(fa1, fb1);
        __syncthreads();
    }
    const int hi = lane >> 5;
    for (int j = 0; j < TN; ++j) for (int i = 0; i < TM; ++i) for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi, n = n0 + wn * (BN / 2) + j * 32 + frow;
        C[(size_t)m * N + n] = acc[i][j][e];
    }
}

int main(int argc, char** argv)
{
    const int sizes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {262144, 256, 2304}, {262144, 256, 256}};
    for (auto& sz : sizes) {
        const int M = sz[0], N = sz[1], K = sz[2];
        std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
        for (auto& v : ha) v = 0x3c00 + (rand() & 0xff);
        for (auto& v : hb) v = 0x3c00 + (rand() & 0xff);
        unsigned short *A, *B; float* C;
        hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        const int lds = 2 * 256 * 128;
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        dim3 grid((M / 128) * (N / 128));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm128<128, 128>), grid, dim3(256), lds, 0, A, B, C, M, N, K);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm128<128, 128>), grid, dim3(256), lds, 0, A, B, C, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("gemm128 M=%d N=%d K=%d: %.1f us  %.1f TF/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
