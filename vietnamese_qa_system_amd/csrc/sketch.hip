// Exact re-scoring of the int8 sketch scan's candidates (large fp16 shards), gfx950 only.
//
// The sketch scan (score_topk.hip MODE 2) prunes with a rigorous upper bound on every (query, row) score and leaves, per
// workgroup, a region of candidate pairs (query << 32 | row position) -- a few thousand of the 2.3e9 pairs of a 9M-row scan.
// Here every pair gets its exact score s = sum_j q_j x_j over the STORED fp16 values (products exact in fp32, fixed summation
// order: a lane's 8 elements of each of its 16-byte units in turn, then a butterfly over the 32 lanes of the pair -- so equal
// rows give equal bits whatever the launch geometry), and the (score, position) key joins the query's candidate list; merge_partials_kernel then
// selects the k best.  The k rows the exact first stage found are re-scored the same way (one arithmetic for every score a
// search returns: exact duplicates on both sides of the stage boundary must tie bit for bit).
//
// Reference interface: the scoring inside `embeddings.search` (inference_pipeline/db_utils/heavy_ranker.py:98-101; txtai ->
// faiss IndexFlatIP: exact inner product with every row); the result is the same, the sketch only decides which rows need it.
#include "vqa_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ size_t unit_of(long long row, int u, int KT) {  // 16-byte unit u of a row in the TILED layout (convert.hip)
    const long long t = row >> 8;
    const int r = (int)(row & 255);
    return ((size_t)t * KT + (u >> 2)) * 1024 + (size_t)(r * 4 + ((u & 3) ^ (((r >> 3) & 1) * 3)));
}

// Half a wave (32 lanes) per candidate pair, two pairs per half in flight: every load of a pair is issued before the first is
// consumed (the rows are 64-byte pieces at 16 KiB strides of the tiled index: latency, not bandwidth, is what a one-pair-at-a-time
// loop pays).  grid (x, regions + 1): y < regions walks that workgroup's region of the sketch scan, y == regions the rows of
// the exact first stage (stage_pos [nq][k] positions, -1 = none).
constexpr int kRescoreUnitsMax = 8;  // 16-byte units per lane and pair: rows of up to 32 * 8 * 8 = 2048 elements take the fast path

__global__ __launch_bounds__(256) void rescore_kernel(const unsigned long long* __restrict__ regions, const unsigned* __restrict__ counts,
                                                      int cap, int nregions, const long long* __restrict__ stage_pos, int nq, int k,
                                                      const _Float16* __restrict__ X, const _Float16* __restrict__ Q, int KT,
                                                      vqa_key* __restrict__ cand_keys, unsigned* __restrict__ cand_cnt, int capq,
                                                      int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int y = blockIdx.y;
    const int total = y < nregions ? (int)counts[y] : nq * k;
    const int units = KT * 4;
    auto pair_of = [&](int i, int& q, long long& pos) {
        q = 0;
        pos = -1;
        if (i >= total) return;
        if (y < nregions) {
            const unsigned long long pr = regions[(size_t)y * cap + i];
            q = (int)(pr >> 32);
            pos = (long long)(pr & 0xFFFFFFFFull);
        } else {
            q = i / k;
            pos = stage_pos[i];
        }
    };
    for (int i0 = wave * 4; i0 < total; i0 += nwaves * 4) {
        int q[2];
        long long pos[2];
        pair_of(i0 + half, q[0], pos[0]);
        pair_of(i0 + 2 + half, q[1], pos[1]);
        float acc[2] = {0.f, 0.f};
        if (units <= 32 * kRescoreUnitsMax) {
            half8 xv[2][kRescoreUnitsMax], qv[2][kRescoreUnitsMax];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < kRescoreUnitsMax; ++j) {
                    const int u = hl + 32 * j;
                    xv[p][j] = qv[p][j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (u < units && pos[p] >= 0) {
                        xv[p][j] = *reinterpret_cast<const half8*>(X + unit_of(pos[p], u, KT) * 8);
                        qv[p][j] = *reinterpret_cast<const half8*>(Q + unit_of(q[p], u, KT) * 8);
                    }
                }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < kRescoreUnitsMax; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[p] = __builtin_fmaf((float)xv[p][j][e], (float)qv[p][j][e], acc[p]);
        } else {
            for (int p = 0; p < 2; ++p)
                for (int u = hl; u < units && pos[p] >= 0; u += 32) {
                    const half8 xw = *reinterpret_cast<const half8*>(X + unit_of(pos[p], u, KT) * 8);
                    const half8 qw = *reinterpret_cast<const half8*>(Q + unit_of(q[p], u, KT) * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[p] = __builtin_fmaf((float)xw[e], (float)qw[e], acc[p]);
                }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) acc[p] += __shfl_xor(acc[p], off, 64);  // inside the half wave
            if (hl == 0 && pos[p] >= 0) {
                const unsigned slot = atomicAdd(cand_cnt + q[p], 1u);
                if (slot < (unsigned)capq) cand_keys[(size_t)q[p] * capq + slot] = vqa_make_key(acc[p], (uint32_t)pos[p]);
                else atomicExch(overflow, 1);
            }
        }
    }
}

// per query: theta (the exact k-th best score of the first stage), ||q_lo||, ||q||, 1 / s_q -> qconst [4][256]; clears the
// candidate counters and the overflow flag of this search
__global__ void sketch_qconst_kernel(const float* __restrict__ thr, const float* __restrict__ qscale, const float* __restrict__ qlo,
                                     const float* __restrict__ qnorm, float* __restrict__ qconst,
                                     unsigned* __restrict__ cand_cnt, int* __restrict__ overflow) {
    const int q = threadIdx.x;
    qconst[q] = thr[q];
    qconst[256 + q] = qlo[q];
    qconst[512 + q] = qnorm[q];
    qconst[768 + q] = 1.0f / qscale[q];
    cand_cnt[q] = 0u;
    if (q == 0) *overflow = 0;
}

}  // namespace

int vqa_launch_sketch_qconst(const float* thr, const float* qscale, const float* qlo, const float* qnorm, float* qconst,
                             unsigned* cand_cnt, int* overflow, hipStream_t stream) {
    hipLaunchKernelGGL(sketch_qconst_kernel, dim3(1), dim3(256), 0, stream, thr, qscale, qlo, qnorm, qconst, cand_cnt, overflow);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_rescore(const unsigned long long* regions, const unsigned* counts, int cap, int nregions, const long long* stage_pos,
                       int nq, int k, const void* x16, const void* q16, int32_t d_pad16, vqa_key* cand_keys, unsigned* cand_cnt,
                       int capq, int* overflow, hipStream_t stream) {
    hipLaunchKernelGGL(rescore_kernel, dim3(32, nregions + 1), dim3(256), 0, stream, regions, counts, cap, nregions, stage_pos, nq, k,
                       reinterpret_cast<const _Float16*>(x16), reinterpret_cast<const _Float16*>(q16), d_pad16 / 32, cand_keys, cand_cnt,
                       capq, overflow);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}
