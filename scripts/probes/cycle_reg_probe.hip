// dev probe: does s_getreg_b32 hwreg(29) (SHADER_CYCLES on later targets) count shader cycles on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned long long* out) {
    unsigned int a, b, c;
    unsigned long long t0, t1;
    asm volatile("s_getreg_b32 %0, hwreg(29)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(t0));
    for (int i = 0; i < 64; ++i) asm volatile("s_nop 15");
    asm volatile("s_getreg_b32 %0, hwreg(29)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(b), "=s"(t1));
    for (int i = 0; i < 640; ++i) asm volatile("s_nop 15");
    asm volatile("s_getreg_b32 %0, hwreg(29)" : "=s"(c));
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; out[3] = t1 - t0; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    probe<<<1, 64>>>(d);
    unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("hwreg29: a=%llu b=%llu c=%llu  b-a=%lld c-b=%lld  memtime delta=%llu\n", h[0], h[1], h[2], (long long)(h[1]-h[0]), (long long)(h[2]-h[1]), h[3]);
    return 0;
}
