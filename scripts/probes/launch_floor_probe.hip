// Dev probe: what does ONE dependent kernel of a launch chain cost on this box?  Chains of 64 launches on one stream, eager and as a
// replayed hipGraph: (a) an empty kernel, 1 workgroup; (b) empty, 192 workgroups of 512 threads; (c) one dependent global load + store
// per thread (192 x 512); (d) two dependent loads.  Build: hipcc --offload-arch=gfx950 -O3 launch_floor_probe.hip -o launch_floor_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty() {}
__global__ void k_load1(const float* a, float* b) { const int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[i] + 1.f; }
__global__ void k_load2(const float* a, const int* idx, float* b) { const int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[idx[i]] + 1.f; }
template <typename F> static double chain_us(F launch, hipStream_t s, bool graph) {
    const int N = 64, R = 50;
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < N; ++i) launch(s);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
    }
    auto run = [&]() { if (graph) hipGraphLaunch(exec, s); else for (int i = 0; i < N; ++i) launch(s); };
    for (int r = 0; r < 5; ++r) run();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) run();
    hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (exec) hipGraphExecDestroy(exec);
    return us / (R * N);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int n = 192 * 512;
    float *a, *b; int* idx;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&idx, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(idx, 0, n * 4);
    for (int graph = 0; graph < 2; ++graph) {
        printf("%s: empty x1 %.2f us | empty 192x512 %.2f us | one load+store 192x512 %.2f us | two dependent loads %.2f us   (per launch of a 64-launch chain)\n",
               graph ? "hipGraph replay" : "eager         ",
               chain_us([&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, s, graph),
               chain_us([&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(192), dim3(512), 0, st); }, s, graph),
               chain_us([&](hipStream_t st) { hipLaunchKernelGGL(k_load1, dim3(192), dim3(512), 0, st, a, b); }, s, graph),
               chain_us([&](hipStream_t st) { hipLaunchKernelGGL(k_load2, dim3(192), dim3(512), 0, st, a, idx, b); }, s, graph));
    }
    return 0;
}
