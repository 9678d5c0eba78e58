// Probe of ds_read_b64_tr_b16 semantics on gfx950: each lane supplies an 8-byte aligned LDS address; which 16-bit
// elements does each lane receive?  LDS element value = its own index (so the source of every element is visible).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode)
{
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int idx;
    if (mode == 0) idx = l * 4;                          // lane l -> elements 4l..4l+3 (contiguous 8 B per lane)
    else idx = (l % 16 / 4) * 64 + (l % 4) * 4 + (l / 16) * 256;   // lane -> row (L/4) of a 64-wide matrix, cols (L%4)*4.., group g -> +4 rows
    unsigned addr = (unsigned)(size_t)(&lds[idx]);
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main()
{
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
