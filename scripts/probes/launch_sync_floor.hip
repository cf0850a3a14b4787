// Dev probe: the floor of "launch ONE kernel and wait for it from the host" on this box -- what vqa_index_search_host's one-launch form
// pays around its kernel.  (a) empty kernel, 4-byte argument; (b) empty kernel with a 3.7 KB by-value argument (the question inside the
// launch packet); (c) kernel that writes 16 bytes to mapped pinned memory.  Waits: hipStreamQuery spin vs hipStreamSynchronize.
// Build: hipcc --offload-arch=gfx950 -O3 launch_sync_floor.hip -o launch_sync_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <vector>
struct Big { uint4 v[224]; };
__global__ void k_small(int x) { (void)x; }
__global__ void k_big(Big b, int* out) { if (out && b.v[0].x == 0x7fffffffu) *out = 1; }
__global__ void k_write(float* out) { if (threadIdx.x == 0) { out[0] = 1.f; out[1] = 2.f; out[2] = 3.f; out[3] = 4.f; } }
template <typename F> static void run(const char* name, F launch, hipStream_t s, bool spin) {
    std::vector<double> t;
    for (int r = 0; r < 300; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        launch();
        if (spin) { while (hipStreamQuery(s) == hipErrorNotReady) {} } else hipStreamSynchronize(s);
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin() + 50, t.end());
    printf("%-44s %-11s median %.1f us  p10 %.1f\n", name, spin ? "query spin" : "synchronize", t[50 + 125], t[50 + 25]);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float* pinned; hipHostMalloc(&pinned, 4096, hipHostMallocMapped); float* pd; hipHostGetDevicePointer((void**)&pd, pinned, 0);
    Big b{}; 
    for (int spin = 1; spin >= 0; --spin) {
        run("empty kernel, 4-byte argument", [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, 1); }, s, spin);
        run("empty kernel, 3.7 KB argument", [&] { hipLaunchKernelGGL(k_big, dim3(16), dim3(128), 0, s, b, (int*)nullptr); }, s, spin);
        run("16 bytes to mapped pinned memory", [&] { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, s, pd); }, s, spin);
        run("same on the null stream", [&] { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, nullptr, pd); }, nullptr, spin);
    }
    return 0;
}
