// Dev probe: the read-only stream ceiling of this box for the scan kernels' own access pattern.
// A persistent grid (one or two workgroups per CU) streams a 7.68 GB (int8 sketch of 10M x 768) or 15.36 GB (fp16 rows) buffer with
// the SAME instructions score_topk.hip / scan_regq.hip use -- `global_load_lds_dwordx4` into an LDS ring under counted
// `s_waitcnt vmcnt(N)` -- and nothing else: no query operand, no fragment reads, no MFMA, no epilogue.  Beside it: the plain
// `global_load_dwordx4` (to registers) form.  Chunks of 192 KiB (one 256-row tile of 768-byte rows) are dealt to workgroups
// round-robin, as the scan deals its tiles.  Prints TB/s per variant (median of 7 launches behind 3 warm-ups).
// Build: hipcc --offload-arch=gfx950 -O3 stream_ceiling.hip -o stream_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef __attribute__((address_space(3))) char* lds_char_ptr;
constexpr size_t kChunk = 192 * 1024;  // one tile of the int8 sketch at d = 768

template <int PIECES, bool NT>
__device__ __forceinline__ void glds_pieces(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(keep) : "s"(lds_dst) : "memory");
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        if constexpr (NT) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2 nt" ::"v"(voff), "s"(sbase), "n"(i * 1024) : "memory");
        else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(sbase), "n"(i * 1024) : "memory");
    }
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA ring: WAVES loader waves, each PIECES x 1 KiB per step (step = WAVES * PIECES KiB), S stages, P steps in flight,
// a workgroup barrier every G steps (0: free-running waves)
template <int WAVES, int PIECES, bool NT, int S, int P, int G>
__global__ __launch_bounds__(WAVES * 64) void dma_stream(const char* __restrict__ X, long nchunks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kStep = WAVES * PIECES * 1024;
    constexpr int kStepsPerChunk = (int)(kChunk / kStep);
    static_assert(kChunk % kStep == 0 && P < S && 2 * 0 + PIECES * P <= 60, "ring");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_char_ptr)smem;
    const uint32_t voff = (uint32_t)(wave * PIECES * 1024 + lane * 16);
    const long mine = blockIdx.x < nchunks ? (nchunks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long total = mine * kStepsPerChunk;
    const char* src = X + (size_t)blockIdx.x * kChunk;
    const size_t jump = ((size_t)gridDim.x - 1) * kChunk;
    int in_chunk = 0, stage = 0;
    long issued = 0;
    auto issue = [&]() {
        glds_pieces<PIECES, NT>(src, voff, smem_lds + stage * kStep);
        ++issued;
        if (issued < total) {
            src += kStep;
            if (++in_chunk == kStepsPerChunk) {
                in_chunk = 0;
                src += jump;
            }
        }
        if (++stage == S) stage = 0;
    };
    if (total == 0) return;
    for (int i = 0; i < P; ++i) issue();
    for (long s = 0; s < total; ++s) {
        issue();                        // step s + P
        wait_vmcnt<PIECES * P>();       // this wave's pieces of step s landed
        if (G > 0 && (s % (G > 0 ? G : 1)) == G - 1) __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0 && smem[17] == 123 && smem[4097] == 77) atomicAdd(sink, 1u);
}

// plain loads to registers: WAVES waves, each keeps R wave-instructions (1 KiB each) in flight
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int WAVES, bool NT, int R>
__global__ __launch_bounds__(WAVES * 64) void reg_stream(const char* __restrict__ X, long nchunks, unsigned* sink) {
    constexpr int kStep = WAVES * 1024;
    constexpr int kStepsPerChunk = (int)(kChunk / kStep);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t voff = (uint32_t)(wave * 1024 + lane * 16);
    const long mine = blockIdx.x < nchunks ? (nchunks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long total = mine * kStepsPerChunk;
    const char* src = X + (size_t)blockIdx.x * kChunk;
    const size_t jump = ((size_t)gridDim.x - 1) * kChunk;
    u32x4 r[R];
    u32x4 acc = {0, 0, 0, 0};
    int in_chunk = 0;
    long issued = 0;
    auto advance = [&]() {
        ++issued;
        if (issued < total) {
            src += kStep;
            if (++in_chunk == kStepsPerChunk) {
                in_chunk = 0;
                src += jump;
            }
        }
    };
    if (total == 0) return;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(r[i]) : "v"(voff), "s"(src) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[i]) : "v"(voff), "s"(src) : "memory");
        advance();
    }
    for (long s = 0; s < total; s += R) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            wait_vmcnt<R - 1>();
            asm volatile("" : "+v"(r[i]));
            acc ^= r[i];
            if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(r[i]) : "v"(voff), "s"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r[i]) : "v"(voff), "s"(src) : "memory");
            advance();
        }
    }
    wait_vmcnt<0>();
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) atomicAdd(sink, 1u);
}

struct Variant {
    const char* name;
    const void* fn;
    int threads, lds;
};

template <typename K>
static double time_kernel(K kern, int grid, int threads, int lds, const char* X, long nchunks, unsigned* sink, hipStream_t st) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 10; ++r) {
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, X, nchunks, sink);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 3) ms.push_back(t);
    }
    CHECK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

#define RUN(NAME, KERN, GRID, THREADS, LDS)                                                                                  \
    do {                                                                                                                     \
        for (int sz = 0; sz < 2; ++sz) {                                                                                     \
            const long nch = sz ? nchunks_all : nchunks_all / 2;                                                             \
            const double ms = time_kernel(KERN, GRID, THREADS, LDS, X, nch, sink, st);                                       \
            printf("%-58s grid %4d  %6.2f GB  %7.3f ms  %6.3f TB/s\n", NAME, GRID, nch * (double)kChunk / 1e9, ms,           \
                   nch * (double)kChunk / (ms * 1e-3) / 1e12);                                                               \
            fflush(stdout);                                                                                                  \
        }                                                                                                                    \
    } while (0)

int main() {
    const long nchunks_all = 78126;  // 78 126 x 192 KiB = 15.36 GB (20M sketch tiles' worth); half of it = the 10M-row int8 sketch
    char* X;
    unsigned* sink;
    CHECK(hipMalloc(&X, nchunks_all * kChunk));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(sink, 0, 4));
    CHECK(hipMemset(X, 0x5a, nchunks_all * kChunk));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CHECK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs; buffer 15.36 GB; rates = bytes / median launch time (7 of 10 launches)\n", prop.name, prop.multiProcessorCount);
    const int cus = prop.multiProcessorCount;
    // --- the X stream of score_topk.hip's slot loop alone: 4 loader waves x 4 pieces (16 KiB steps), 5 of 6 stages in flight
    RUN("LDS-DMA nt, 4 waves x 4 KiB, ring 6 / 5 ahead, free-running", (dma_stream<4, 4, true, 6, 5, 0>), cus, 256, 6 * 16384);
    RUN("LDS-DMA    , 4 waves x 4 KiB, ring 6 / 5 ahead, free-running", (dma_stream<4, 4, false, 6, 5, 0>), cus, 256, 6 * 16384);
    RUN("LDS-DMA nt, 4 waves x 4 KiB, ring 9 / 8 ahead, free-running", (dma_stream<4, 4, true, 9, 8, 0>), cus, 256, 9 * 16384);
    RUN("LDS-DMA nt, 4 waves x 4 KiB, ring 6 / 5 ahead, barrier per step", (dma_stream<4, 4, true, 6, 5, 1>), cus, 256, 6 * 16384);
    // --- the register-resident-query scan's stream: 4 waves x 2 pieces (8 KiB steps), deep ring
    RUN("LDS-DMA nt, 4 waves x 2 KiB, ring 16 / 14 ahead, free-running", (dma_stream<4, 2, true, 16, 14, 0>), cus, 256, 16 * 8192);
    RUN("LDS-DMA nt, 4 waves x 2 KiB, ring 16 / 13 ahead, barrier per 4", (dma_stream<4, 2, true, 16, 13, 4>), cus, 256, 16 * 8192);
    RUN("LDS-DMA nt, 4 waves x 2 KiB, ring 16 / 8 ahead, barrier per 4", (dma_stream<4, 2, true, 16, 8, 4>), cus, 256, 16 * 8192);
    RUN("LDS-DMA nt, 4 waves x 2 KiB, ring 16 / 14 ahead, barrier per step", (dma_stream<4, 2, true, 16, 14, 1>), cus, 256, 16 * 8192);
    RUN("LDS-DMA    , 4 waves x 2 KiB, ring 16 / 14 ahead, free-running", (dma_stream<4, 2, false, 16, 14, 0>), cus, 256, 16 * 8192);
    // --- more loader waves / two workgroups per CU
    RUN("LDS-DMA nt, 8 waves x 2 KiB, ring 9 / 8 ahead, free-running", (dma_stream<8, 2, true, 9, 8, 0>), cus, 512, 9 * 16384);
    RUN("LDS-DMA nt, 8 waves x 1 KiB, ring 16 / 15 ahead, free-running", (dma_stream<8, 1, true, 16, 15, 0>), cus, 512, 16 * 8192);
    RUN("LDS-DMA nt, 4 waves x 2 KiB, ring 9 / 8, 2 workgroups per CU", (dma_stream<4, 2, true, 9, 8, 0>), 2 * cus, 256, 9 * 8192);
    RUN("LDS-DMA nt, 4 waves x 4 KiB, ring 4 / 3, 2 workgroups per CU", (dma_stream<4, 4, true, 4, 3, 0>), 2 * cus, 256, 4 * 16384);
    // --- plain loads to registers
    RUN("global_load_dwordx4 nt, 4 waves, 16 in flight per wave", (reg_stream<4, true, 16>), cus, 256, 0);
    RUN("global_load_dwordx4 nt, 8 waves, 8 in flight per wave", (reg_stream<8, true, 8>), cus, 512, 0);
    RUN("global_load_dwordx4 nt, 8 waves, 16 in flight per wave", (reg_stream<8, true, 16>), cus, 512, 0);
    RUN("global_load_dwordx4   , 8 waves, 16 in flight per wave", (reg_stream<8, false, 16>), cus, 512, 0);
    RUN("global_load_dwordx4 nt, 16 waves, 8 in flight per wave", (reg_stream<16, true, 8>), cus, 1024, 0);
    RUN("global_load_dwordx4 nt, 8 waves, 16 in flight, 2 WG per CU", (reg_stream<8, true, 16>), 2 * cus, 512, 0);
    RUN("global_load_dwordx4 nt, 4 waves, 16 in flight, 4 WG per CU", (reg_stream<4, true, 16>), 4 * cus, 256, 0);
    unsigned h = 0;
    CHECK(hipMemcpy(&h, sink, 4, hipMemcpyDeviceToHost));
    printf("# sink %u\n", h);
    return 0;
}
