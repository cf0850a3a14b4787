// Dev probe: what would ONE phase of a persistent single-question forward cost?  G workgroups of 512 threads, all resident, run P phases;
// a phase = (optionally) a device-coherent load of a small activation buffer another workgroup wrote in the phase before, a store of
// its own share, and a grid barrier.  Barrier forms: (0) fence-free -- stores waited for with s_waitcnt, one device-scope atomic add
// per workgroup, polling a device-scope load; (1) the same with __threadfence() on both sides (L2 write-back + invalidate per workgroup).
// Also: W bytes of "weights" per phase (a different slice every phase, read with plain loads) requested BEFORE the barrier wait or
// after it.  Build: hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load_coherent(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_coherent(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// FENCE: 0 fence-free, one counter; 1 fenced, one counter; 2 fence-free TREE: leaves of kLeaf workgroups (a 128-byte line each), the last
// arriver of a leaf adds to the root, everybody polls the root
constexpr int kLeaf = 12;
template <int FENCE>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned phase) {
    const unsigned G = gridDim.x;
    if (FENCE == 1) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE == 2) {
            const unsigned leaf = blockIdx.x / kLeaf, leaves = (G + kLeaf - 1) / kLeaf;
            const unsigned members = leaf + 1 < leaves ? kLeaf : G - leaf * kLeaf;
            const unsigned old = __hip_atomic_fetch_add(counter + 32 * (1 + leaf), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (phase + 1) * members) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spin = 0; spin < (1u << 22) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * leaves; ++spin) __builtin_amdgcn_s_sleep(1);
        } else {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spin = 0; spin < (1u << 22) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * G; ++spin) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (FENCE == 1) __threadfence();
}

template <int FENCE, int DATA, int PREFETCH>
__global__ __launch_bounds__(512) void phases(unsigned* counter, u32x4* x, u32x4* y, int units, const u32x4* w, int w_units_per_phase, int phases_n, unsigned* sink) {
    const int G = gridDim.x, tid = threadIdx.x, gid = blockIdx.x * 512 + tid;
    unsigned acc = 0;
    u32x4* src = x;
    u32x4* dst = y;
    for (int p = 0; p < phases_n; ++p) {
        u32x4 wv[4] = {};
        const u32x4* wp = w + (size_t)p % 16 * w_units_per_phase;  // a different 1/16 of the weight buffer every phase
        auto load_w = [&] {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = gid + i * G * 512;
                if (u < w_units_per_phase) wv[i] = wp[u];
            }
        };
        if (w_units_per_phase && !PREFETCH) load_w();
        if (DATA) {
            const int u = (gid * 7 + p * 131) % units;  // somebody else's unit of the previous phase
            const u32x4 v = DATA == 1 ? load_coherent(src + u) : src[u];
            acc += v.x;
            if (gid < units) {
                if (DATA == 1) store_coherent(dst + gid, u32x4{v.x + 1, v.y, v.z, (unsigned)p});
                else dst[gid] = u32x4{v.x + 1, v.y, v.z, (unsigned)p};
            }
        }
        acc += wv[0].x + wv[1].y + wv[2].z + wv[3].w;
        if (w_units_per_phase && PREFETCH) {  // the NEXT phase's weights travel while this workgroup waits at the barrier
            wp = w + (size_t)(p + 1) % 16 * w_units_per_phase;
            load_w();
        }
        grid_barrier<FENCE>(counter, (unsigned)p);
        if (PREFETCH) acc += wv[0].x + wv[1].y + wv[2].z + wv[3].w;
        u32x4* t = src; src = dst; dst = t;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int FENCE, int DATA, int PREFETCH>
static void run(const char* name, int G, int wbytes) {
    unsigned *counter, *sink; u32x4 *x, *y, *w;
    const int units = 4096;  // 64 KB of activations
    hipMalloc(&counter, 8192); hipMalloc(&sink, 64);
    if (DATA == 2) { hipExtMallocWithFlags((void**)&x, units * 16, hipDeviceMallocFinegrained); hipExtMallocWithFlags((void**)&y, units * 16, hipDeviceMallocFinegrained); }
    else if (DATA == 3) { hipExtMallocWithFlags((void**)&x, units * 16, hipDeviceMallocUncached); hipExtMallocWithFlags((void**)&y, units * 16, hipDeviceMallocUncached); }
    else { hipMalloc(&x, units * 16); hipMalloc(&y, units * 16); } hipMalloc(&w, (size_t)16 * (wbytes ? wbytes : 16));
    hipMemset(x, 0, units * 16); hipMemset(y, 0, units * 16); hipMemset(w, 1, (size_t)16 * (wbytes ? wbytes : 16));
    const int P = 500;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipMemset(counter, 0, 8192);
        hipEventRecord(e0);
        hipLaunchKernelGGL((phases<FENCE, DATA, PREFETCH>), dim3(G), dim3(512), 0, nullptr, counter, x, y, units, w, wbytes / 16, P, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    int wrong = 0;
    if (DATA) {  // every unit's counter went up by one per phase whoever's unit it was copied from: any stale read shows
        std::vector<unsigned> h(units * 4);
        hipMemcpy(h.data(), (P & 1) ? y : x, units * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < units; ++i) wrong += h[4 * i] != (unsigned)P * 4;  // (4 runs on the same buffers)
    }
    printf("G=%3d  %-52s weights %4d KB/phase: %.2f us per phase%s\n", G, name, wbytes >> 10, best * 1e3f / P, wrong ? "   STALE READS" : "");
    hipFree(counter); hipFree(sink); hipFree(x); hipFree(y); hipFree(w);
}

int main() {
    for (int G : {48, 144, 192, 256}) {
        run<0, 0, 0>("barrier only, fence-free", G, 0);
        run<1, 0, 0>("barrier only, __threadfence both sides", G, 0);
        run<0, 1, 0>("coherent load + store + fence-free barrier", G, 0);
        run<1, 1, 0>("coherent load + store + fenced barrier", G, 0);
        run<0, 1, 0>("... + weights read inside the phase", G, 4 << 20);
        run<0, 1, 1>("... + weights requested before the barrier wait", G, 4 << 20);
        run<0, 3, 0>("UNCACHED buffers, plain load + store + fence-free barrier", G, 0);
        run<0, 3, 1>("UNCACHED buffers, plain accesses, weights prefetched", G, 4 << 20);
        run<2, 0, 0>("barrier only, fence-free tree", G, 0);
        run<2, 1, 0>("coherent load + store + tree barrier", G, 0);
        run<2, 1, 0>("... + weights read inside the phase", G, 4 << 20);
    }
    return 0;
}
