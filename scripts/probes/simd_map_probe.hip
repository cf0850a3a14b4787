// Probe (dev tool): which SIMD each wave of a 512-thread, 160 KiB-LDS workgroup lands on (HW_REG_HW_ID bits 5:4), gfx950.
// hipcc --offload-arch=gfx950 -O3 simd_map_probe.hip -o /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ char smem[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<<<256, 512, 160 * 1024>>>(d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0, hist[8][4] = {};
    for (int b = 0; b < 256; ++b) {
        bool ok = true;
        for (int w = 0; w < 8; ++w) { int s = (h[b * 8 + w] >> 4) & 3; hist[w][s]++; if (w >= 4 && s != (int)((h[b * 8 + w - 4] >> 4) & 3)) ok = false; }
        same += ok;
    }
    printf("workgroups whose waves w and w+4 share a SIMD: %d / 256\n", same);
    for (int w = 0; w < 8; ++w) printf("wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 4; ++b) { printf("wg %d simds:", b); for (int w = 0; w < 8; ++w) printf(" %d", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
    return 0;
}
