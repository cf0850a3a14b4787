// Probe (dev tool, not part of the library): operand pairing, scale semantics and issue rate of the block-scaled fp8 MFMA
// v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950.   hipcc --offload-arch=gfx950 -O3 mx_fp8_probe.hip -o /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void one(const v8i* a, const v8i* b, f32x4* c, int scale) {
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, scale, 0, scale);
    c[threadIdx.x] = acc;
}
template <int SCALED>
__global__ void rate(const v8i* a, const v8i* b, f32x4* c, int iters) {
    v8i av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
    f32x4 acc[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SCALED)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else {
                long a0 = ((long*)&av)[j & 3], b0 = ((long*)&bv)[j & 3];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc[j], 0, 0, 0);
            }
        }
    }
    f32x4 s = {0, 0, 0, 0};
    for (int j = 0; j < 8; ++j) s += acc[j];
    c[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float e4m3(uint8_t c) {
    int e = (c >> 3) & 15, m = c & 7;
    float v = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1 + m / 8.0f, e - 7);
    return (c & 0x80) ? -v : v;
}
int main() {
    uint8_t ha[64 * 32], hb[64 * 32];
    srand(1);
    const uint8_t codes[] = {0x00, 0x38, 0x40, 0x44, 0xB8, 0xC0, 0x30, 0xB0};  // 0, 1, 2, 3, -1, -2, 0.5, -0.5
    for (int i = 0; i < 64 * 32; ++i) { ha[i] = codes[rand() % 8]; hb[i] = codes[rand() % 8]; }
    v8i *da, *db; f32x4* dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * 16);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    for (int scale : {(int)0x7F7F7F7F, (int)0x80808080, (int)0x7F7F7F80}) {
        one<<<1, 64>>>(da, db, dc, scale);
        float hc[64 * 4];
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        double maxerr = 0, ratio = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                int col = l & 15, row = (l >> 4) * 4 + r;  // D[row][col]: A row = `row`, B col = `col`
                double ref = 0;
                for (int g = 0; g < 4; ++g)
                    for (int j = 0; j < 32; ++j) ref += e4m3(ha[(g * 16 + row) * 32 + j]) * e4m3(hb[(g * 16 + col) * 32 + j]);
                maxerr = fmax(maxerr, fabs(hc[l * 4 + r] - ref));
                if (ref != 0) ratio = hc[l * 4 + r] / ref;
            }
        printf("scale %08x: max |D - ref| = %g   (last D/ref = %g)\n", scale, maxerr, ratio);
    }
    v8i* ra; hipMalloc(&ra, sizeof(ha));
    f32x4* rc; hipMalloc(&rc, 1024 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int scaled = 0; scaled < 2; ++scaled) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (scaled) rate<1><<<1024, 256>>>(da, db, rc, iters); else rate<0><<<1024, 256>>>(da, db, rc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 1024.0 * 4 * iters * 8 * 2 * 16 * 16 * (scaled ? 128 : 32);
        printf("%s: %.3f ms, %.1f TFLOP/s\n", scaled ? "scaled 16x16x128" : "plain 16x16x32", ms, flop / ms / 1e9);
    }
    return 0;
}
