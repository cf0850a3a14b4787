// Measurement helper of libvqa_retrieval (not on any search path): the read-only stream rate of this device for the scan kernels' own
// access pattern -- `global_load_lds_dwordx4 ... nt` into an LDS ring under counted `s_waitcnt vmcnt(N)`, one persistent workgroup per CU,
// 192 KiB chunks dealt round-robin, nothing else (no query operand, no fragment reads, no MFMA, no epilogue).  This is the ceiling the
// scans' `roofline.frac` can be read against on THIS box (scripts/probes/stream_ceiling.hip is the stand-alone form with more variants:
// profiles/r06_stream_ceiling.txt, 6.8-7.0 TB/s); bench.py reports it as roofline.hbm_read_stream_measured_gbs.
#include "vqa_common.h"

namespace {

typedef __attribute__((address_space(3))) char* lds_char_ptr;
constexpr size_t kChunk = 192 * 1024;
constexpr int kStages = 6, kAhead = 5, kStep = 16 * 1024;

__global__ __launch_bounds__(256) void read_stream_kernel(const char* __restrict__ X, long long nchunks, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kStepsPerChunk = (int)(kChunk / kStep);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_char_ptr)smem;
    const uint32_t voff = (uint32_t)(wave * 4096 + lane * 16);
    const long long mine = blockIdx.x < nchunks ? (nchunks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long long total = mine * kStepsPerChunk;
    const char* src = X + (size_t)blockIdx.x * kChunk;
    const size_t jump = ((size_t)gridDim.x - 1) * kChunk;
    int in_chunk = 0, stage = 0;
    long long issued = 0;
    auto issue = [&]() {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(src), "s"(smem_lds + stage * kStep)
            : "memory");
        ++issued;
        if (issued < total) {
            src += kStep;
            if (++in_chunk == kStepsPerChunk) {
                in_chunk = 0;
                src += jump;
            }
        }
        if (++stage == kStages) stage = 0;
    };
    if (total == 0) return;
    for (int i = 0; i < kAhead; ++i) issue();
    for (long long s = 0; s < total; ++s) {
        issue();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * kAhead) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[17] == 123 && smem[4097] == 77 && sink) atomicAdd(sink, 1u);
}

}  // namespace

extern "C" int vqa_measure_read_stream(const void* device_buffer, int64_t bytes, int32_t reps, double* out_gbs, void* hip_stream) {
    VQA_REQUIRE(device_buffer && out_gbs, "vqa_measure_read_stream: null pointer");
    VQA_REQUIRE(bytes >= (int64_t)kChunk && reps >= 1 && reps <= 1000, "vqa_measure_read_stream: bytes=%lld reps=%d", (long long)bytes, reps);
    hipStream_t s = (hipStream_t)hip_stream;
    int dev = 0;
    hipDeviceProp_t prop;
    VQA_HIP_CHECK(hipGetDevice(&dev));
    VQA_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    static VqaPerDeviceOnce once;
    int rc = once.run([&](int) -> int {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStep));
        return VQA_OK;
    });
    if (rc != VQA_OK) return rc;
    const long long nchunks = bytes / (long long)kChunk;
    hipEvent_t e0, e1;
    VQA_HIP_CHECK(hipEventCreate(&e0));
    VQA_HIP_CHECK(hipEventCreate(&e1));
    float best = 0.f;
    for (int r = 0; r < reps + 2; ++r) {  // two warm-up launches, then the fastest of reps (a stream has no reason to be slower than its best)
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(read_stream_kernel, dim3(prop.multiProcessorCount), dim3(256), kStages * kStep, s, reinterpret_cast<const char*>(device_buffer), nchunks,
                           (unsigned*)nullptr);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            vqa_set_error("vqa_measure_read_stream: the stream kernel failed");
            return VQA_EHIP;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2 && (best == 0.f || ms < best)) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *out_gbs = (double)nchunks * (double)kChunk / ((double)best * 1e-3) / 1e9;
    return VQA_OK;
}
