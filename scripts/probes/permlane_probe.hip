// dev probe: lane mapping of v_permlane16_swap / v_permlane32_swap on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 1024); k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"p16 r[0]", "p16 r[1]", "p32 r[0]", "p32 r[1]"};
    for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[v * 64 + i]); printf("\n"); }
    return 0;
}
