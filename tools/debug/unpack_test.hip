#include "../../rdpn6d_amd/csrc/common.h"
#include <cstdio>
#include <vector>
extern "C" void rdpn6d_set_error(const char*, ...) {}
__global__ void k(const rd_u32x4* x, float* y) { float v[8]; rd_unpack8(x[threadIdx.x], v); for (int i=0;i<8;i++) y[threadIdx.x*8+i]=v[i]; }
__global__ void k2(const rd_bf16_t* x, float* y) { y[threadIdx.x] = rd_bf2f(x[threadIdx.x]); }
int main() {
    std::vector<_Float16> h(64); for (int i=0;i<64;i++) h[i]=(_Float16)(i*0.25f-5.f);
    void *dx; float* dy; hipMalloc(&dx, 128); hipMalloc(&dy, 64*4); hipMemcpy(dx, h.data(), 128, hipMemcpyHostToDevice);
    k<<<1,8>>>((const rd_u32x4*)dx, dy); std::vector<float> o(64); hipMemcpy(o.data(), dy, 256, hipMemcpyDeviceToHost);
    int bad=0; for (int i=0;i<64;i++) if (o[i]!=(float)h[i]) { if (bad<8) printf("unpack8 mismatch %d: %f vs %f\n", i, o[i], (float)h[i]); bad++; }
    printf("unpack8 bad %d\n", bad);
    k2<<<1,64>>>((const rd_bf16_t*)dx, dy); hipMemcpy(o.data(), dy, 256, hipMemcpyDeviceToHost);
    bad=0; for (int i=0;i<64;i++) if (o[i]!=(float)h[i]) bad++; printf("bf2f bad %d\n", bad);
    return 0;
}
