// shader clock under different loads: spin for N s_memtime ticks per wave, wall time by events -> ticks per second.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/probe/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void spin(unsigned long long ticks, int mfma, float* out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    f32x16 acc = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    while (__builtin_readcyclecounter() - t0 < ticks) {
        if (mfma)
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 123.f) out[0] = acc[1];
}
int main()
{
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mfma = 0; mfma < 2; ++mfma)
        for (int blocks : {1, 256, 1024}) {
            const unsigned long long ticks = 20000000ull;
            spin<<<blocks, 256>>>(1000, mfma, out);
            hipDeviceSynchronize();
            hipEventRecord(e0); spin<<<blocks, 256>>>(ticks, mfma, out); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mfma=%d blocks=%4d: %llu ticks in %.3f ms -> %.3f GHz\n", mfma, blocks, ticks, ms, ticks / (ms * 1e6));
        }
    return 0;
}
