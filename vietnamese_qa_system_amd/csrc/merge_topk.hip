// K2 -- merges of sorted candidate lists (one workgroup per query), gfx950 only.
//
//  * merge_partials: the per-workgroup lists K1 flushed (vqa_key, [parts][256][k]) -> final top-k of the shard:
//    fp32 score, external id (faiss IDMap step: ids[pos] or id_base + pos; ids originate at
//    inference_pipeline/db_utils/heavy_ranker.py:76), row position; optionally the k-th best score per query
//    (threshold seed for the second scoring pass).
//  * merge_shards: the [R, B, k] (score, id) candidates after the RCCL all-gather -> [B, k_out]; ties resolve by
//    (rank asc, slot asc) which is global row position ascending for contiguous row shards.
// Selection: k rounds of "largest key strictly below the previous winner" (keys are distinct), each round one
// wavefront-shuffle max reduction + a 4-entry LDS exchange between the waves.
#include "vqa_common.h"

namespace {

constexpr int kMergeThreads = 256;
constexpr int kRadixSelectK = 32;  // merge_partials_kernel: from this k on the k-th key is found by radix descent instead of k selection rounds

// block-wide max of per-thread values; every thread gets the result.  red: LDS [2][4], `round` picks the half -- ONE barrier per
// call: the half written in round r is next written in round r + 2, behind the barrier of round r + 1 that every reader of round r
// has passed.
__device__ __forceinline__ vqa_key block_max_key(vqa_key v, vqa_key* red, int round) {
    v = vqa_wave_max_key(v);
    const int wave = threadIdx.x >> 6;
    vqa_key* slot = red + 4 * (round & 1);
    if ((threadIdx.x & 63) == 0) slot[wave] = v;
    __syncthreads();
    vqa_key m = slot[0];
#pragma unroll
    for (int w = 1; w < kMergeThreads / 64; ++w) m = slot[w] > m ? slot[w] : m;
    return m;
}

// The k largest of the m keys in LDS `keys` (distinct but for empty = 0 slots), best first, into win[0, k) (0 = none); every thread of the
// block calls it.  The k-th largest key by an 8-bit radix descent from the top byte -- 8 passes: a histogram of the current byte over the
// keys that match the bytes chosen so far, then the bin in which the count from the top reaches the rank wanted -- then the keys at or
// above it gathered and ordered by counting ranks: ~30 us whatever k, where k rounds of "largest key below the previous winner" cost
// k x (a pass over the keys + a block reduction) -- 64-173 us per selection of a k = 100 cascade.  `keys` is overwritten (scratch for
// the ordered winners); hist: [256 + 4] ints of LDS.
__device__ __forceinline__ void select_topk_radix(vqa_key* keys, int m, int k, vqa_key* win, int* hist) {
    int* sel = hist + 256;
    const int kk = k < m ? k : m;
    vqa_key prefix = 0ull;
    int need = kk;
    for (int pass = 0; pass < 8 && kk > 0; ++pass) {
        const int shift = 56 - 8 * pass;
        const vqa_key hi_mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        for (int i = threadIdx.x; i < 256; i += kMergeThreads) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += kMergeThreads) {
            const vqa_key v = keys[i];
            if ((v & hi_mask) == prefix) atomicAdd(&hist[(int)((v >> shift) & 255ull)], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // one wave: lane l owns bins 255 - 4 l .. 252 - 4 l (from the top); inclusive scan over the lanes
            const int l = threadIdx.x;
            const int h0 = hist[255 - 4 * l], h1 = hist[254 - 4 * l], h2 = hist[253 - 4 * l], h3 = hist[252 - 4 * l];
            const int mine = h0 + h1 + h2 + h3;
            int inc = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(inc, off, 64);
                if (l >= off) inc += up;
            }
            const int before = inc - mine;  // keys in bins above this lane's
            if (before < need && inc >= need) {  // the wanted rank falls into this lane's four bins: exactly one lane
                int b = 255 - 4 * l, above = before;
                if (above + h0 < need) {
                    above += h0;
                    --b;
                    if (above + h1 < need) {
                        above += h1;
                        --b;
                        if (above + h2 < need) {
                            above += h2;
                            --b;
                        }
                    }
                }
                sel[0] = b;
                sel[1] = need - above;
            }
        }
        __syncthreads();
        prefix |= (vqa_key)(unsigned)sel[0] << shift;
        need = sel[1];
    }
    if (threadIdx.x == 0) sel[2] = 0;
    __syncthreads();
    // prefix IS the kk-th largest key.  The keys above it first (fewer than kk), then those equal to it up to kk in all: exactly one
    // when the keys are distinct; the dense seed lists may hold empty (zero) slots, which tie
    if (kk > 0) {
        for (int i = threadIdx.x; i < m; i += kMergeThreads) {
            const vqa_key v = keys[i];
            if (v > prefix) win[atomicAdd(&sel[2], 1)] = v;
        }
    }
    __syncthreads();
    if (kk > 0) {
        for (int i = threadIdx.x; i < m; i += kMergeThreads) {
            const vqa_key v = keys[i];
            if (v == prefix) {
                const int sl = atomicAdd(&sel[2], 1);
                if (sl < kk) win[sl] = v;
            }
        }
    }
    __syncthreads();
    // order the kk winners (best first) by rank counting; the slots behind them are empty
    for (int t0 = 0; t0 < k; t0 += kMergeThreads) {
        const int t = t0 + threadIdx.x;
        vqa_key mine = 0ull;
        int rank = t;
        if (t < kk) {
            mine = win[t];
            rank = 0;
            for (int j = 0; j < kk; ++j) rank += (win[j] > mine || (win[j] == mine && j < t)) ? 1 : 0;
        }
        // (every chunk reads the unsorted array: the sorted one goes to `keys`, which nobody needs any more, and is copied back)
        if (t < k) keys[rank] = mine;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < k; t += kMergeThreads) win[t] = keys[t];
}

// a sketch call's flags -> the pinned mirror the host's pause logic reads: overflow of the last query tile, OR over the earlier ones, the
// most pairs any tile scored exactly, and LAST (release) the call's number, which tells the host the report is complete
__device__ __forceinline__ void publish_flags(const MergeSketchTail& tail) {
    const int f0 = __hip_atomic_load(tail.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(tail.flag_mirror, f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(tail.flag_mirror + 1, tail.overflow[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(tail.flag_mirror + 3, tail.overflow[4] > tail.overflow[3] ? tail.overflow[4] : tail.overflow[3], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(tail.flag_mirror + 2, tail.overflow[2], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kMergeThreads) void merge_partials_kernel(const vqa_key* __restrict__ partial, int parts,
                                                                       int list_len, int k,
                                                                       const long long* __restrict__ ids,
                                                                       long long id_base, float* __restrict__ out_scores,
                                                                       long long* __restrict__ out_ids,
                                                                       long long* __restrict__ out_pos,
                                                                       float* __restrict__ out_thr, float score_scale,
                                                                       int out_stride, int out_offset,
                                                                       vqa_key* __restrict__ out_last_key, int query_major,
                                                                       const int* __restrict__ gate, int row_lists,
                                                                       const unsigned* __restrict__ counts, int count_stride,
                                                                       MergeSketchTail tail, int cap_keys) {
    if (tail.flag_mirror && tail.mirror_before_gate && blockIdx.x == 0 && threadIdx.x == 0) publish_flags(tail);
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    vqa_key* keys = reinterpret_cast<vqa_key*>(smem);    // [cap_keys] (= parts * list_len unless the lists are longer than LDS: counts form)
    vqa_key* red = keys + cap_keys;                      // [2][4]
    int* fill = reinterpret_cast<int*>(red + 8);         // keys kept
    const int q = blockIdx.x;
    const int m_all = parts * list_len;
    // Only the non-empty slots are kept: a workgroup's list of a query holds ~1-3 keys after a scan seeded with good
    // thresholds, so the 512 x k slots of a two-stage search shrink to a few hundred keys and the k selection rounds below
    // touch 2 instead of 20 keys per thread (final merge 31 -> 17 us).  The order in which the keys land is arbitrary; the
    // selection does not depend on it (keys are distinct).
    // (seed lists -- 2 or 8 sub-maxima per tile, thresholds only -- are dense: they are copied as they are)
    const bool dense = out_scores == nullptr && list_len <= 8;
    if (threadIdx.x == 0) *fill = dense ? m_all : 0;
    __syncthreads();
    if (counts) {
        // a sketch search's candidate list: `parts` sub-lists of list_len slots, the first counts[q][part] of each are this search's
        // (the slots behind hold keys of earlier searches)
        // (kMergeThreads / parts threads per sub-list, all sub-lists at once: one dependent count load per thread instead of
        // `parts` in a row)
        const int tpp = parts <= kMergeThreads ? kMergeThreads / parts : 1;
        // keys below the floor cannot be among the k best (MergeSketchTail::min_score); the key of (floor, any position) is >= this one
        const vqa_key floor_key = tail.min_score ? vqa_make_key(tail.min_score[q], 0xFFFFFFFFu) : 0ull;
        for (int p = threadIdx.x / tpp; p < parts; p += kMergeThreads / tpp) {
            const unsigned cn = counts[(size_t)(q * parts + p) * count_stride];
            const int c = (int)cn < list_len ? (int)cn : list_len;
            const vqa_key* src = partial + ((size_t)q * row_lists + p) * list_len;
            for (int j = threadIdx.x % tpp; j < c; j += tpp) {
                const vqa_key v = src[j];
                if (v != 0ull && v >= floor_key) {
                    const int slot = atomicAdd(fill, 1);
                    if (slot < cap_keys) keys[slot] = v;
                }
            }
        }
    } else
    for (int i = threadIdx.x; i < m_all; i += kMergeThreads) {
        vqa_key v;
        if (query_major) {  // a query's row holds row_lists lists; the first `parts` of them are merged
            v = partial[(size_t)q * row_lists * list_len + i];
        } else {
            const int p = i / list_len, j = i - p * list_len;
            v = partial[((size_t)p * VQA_QUERY_TILE + q) * list_len + j];
        }
        if (dense) keys[i] = v;
        else if (v != 0ull) keys[atomicAdd(fill, 1)] = v;
    }
    __syncthreads();
    int m = *fill;
    if (m > cap_keys) {  // more surviving keys than LDS holds: this selection is not to be trusted -- the search's exact fallback takes over
        m = cap_keys;
        if (threadIdx.x == 0 && tail.overflow) atomicExch(tail.overflow, 1);
    }
    // The winners are collected in LDS and written out after the last round: __syncthreads waits for every outstanding memory
    // operation of the wave, so a global store per round made each round pay a store round trip (~2 us against ~0.6 us of work).
    vqa_key* win = reinterpret_cast<vqa_key*>(fill + 2);  // [k]
    if (k >= kRadixSelectK && cap_keys >= k) {
        // large k: radix descent to the k-th key instead of k selection rounds (select_topk_radix)
        select_topk_radix(keys, m, k, win, reinterpret_cast<int*>(win + k));
    } else {
    vqa_key prev = ~0ull;
    for (int r = 0; r < k; ++r) {
        vqa_key best = 0ull;
        for (int i = threadIdx.x; i < m; i += kMergeThreads) {
            const vqa_key v = keys[i];
            best = (v < prev && v > best) ? v : best;
        }
        best = block_max_key(best, red, r);
        if (threadIdx.x == 0) win[r] = best;
        prev = best;  // 0 once the candidates are exhausted: later rounds stay empty
    }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < k; r += kMergeThreads) {
        const vqa_key best = win[r];
        const bool empty = best == 0ull;
        const long long pos = empty ? -1 : (long long)vqa_key_pos(best);
        const size_t o = (size_t)q * out_stride + out_offset + r;
        if (out_scores) out_scores[o] = empty ? -INFINITY : vqa_key_score(best) * score_scale;
        if (out_ids) out_ids[o] = empty ? -1 : (ids ? ids[pos] : id_base + pos);
        if (out_pos) out_pos[o] = pos;
        if (out_thr && r == k - 1) out_thr[q] = empty ? -INFINITY : vqa_key_score(best);
        if (out_last_key && r == k - 1) out_last_key[q] = best;
    }
    // ---- the sketch search's cascade (MergeSketchTail): what used to be launches of their own between its scans
    if (tail.qconst && threadIdx.x == 0) {  // sketch.hip sketch_qconst_kernel, for this block's query, from the k-th score just selected
        const vqa_key kth = win[k - 1];
        const float thr = kth == 0ull ? -INFINITY : vqa_key_score(kth);
        const float qn = tail.qnorm[q];
        tail.qconst[q] = tail.qoff ? thr - tail.qoff[q] - tail.mu_margin * qn : thr;
        tail.qconst[256 + q] = tail.qlo[q] + tail.fp_margin * qn;
        tail.qconst[512 + q] = (tail.qrnorm ? tail.qrnorm[q] : qn) + tail.fp_margin * qn;
        tail.qconst[768 + q] = 1.0f / tail.qscale[q];
        tail.qconst[1024 + q] = tail.qalpha ? tail.qalpha[q] : 0.f;
        tail.qconst[1280 + q] = tail.fp_margin * qn;
    }
    if (tail.clear && tail.cand_cnt) {
        // the candidate counters of this query -- and, by block 0, of the queries past the batch and the overflow flags
        for (int j = threadIdx.x; j < kSketchSubLists; j += kMergeThreads) tail.cand_cnt[(size_t)(q * kSketchSubLists + j) * kSketchCntStride] = 0u;
        if (q == 0) {
            for (int j = (int)gridDim.x * kSketchSubLists + threadIdx.x; j < VQA_QUERY_TILE * kSketchSubLists; j += kMergeThreads)
                tail.cand_cnt[(size_t)j * kSketchCntStride] = 0u;
            if (threadIdx.x == 0) {
                tail.overflow[1] = tail.clear == 2 ? 0 : (tail.overflow[1] | tail.overflow[0]);
                tail.overflow[0] = 0;
                tail.overflow[2] = tail.seq;
                // [3]: pairs scored exactly for this query tile (rescore_kernel counts), [4]: the most any EARLIER tile of the call scored
                tail.overflow[4] = tail.clear == 2 ? 0 : (tail.overflow[4] > tail.overflow[3] ? tail.overflow[4] : tail.overflow[3]);
                tail.overflow[3] = 0;
            }
        }
    }
    if (tail.flag_mirror && !tail.mirror_before_gate && q == 0 && threadIdx.x == 0) publish_flags(tail);
}

// Large k: the k rounds of the selection kernel above cost ~1.5 us each; past k = 32 the lists are sorted instead --
// bitonic sort of the (zero-padded, power-of-two) key array in LDS, descending, then the first k keys are the result.
// Keys are distinct (score, position) pairs, so the sorted order is the result order.
__global__ __launch_bounds__(kMergeThreads) void merge_partials_sort_kernel(const vqa_key* __restrict__ partial, int parts,
                                                                            int list_len, int k, int m_pow2,
                                                                            const long long* __restrict__ ids, long long id_base,
                                                                            float* __restrict__ out_scores,
                                                                            long long* __restrict__ out_ids,
                                                                            long long* __restrict__ out_pos,
                                                                            float* __restrict__ out_thr, float score_scale,
                                                                            int out_stride, int out_offset,
                                                                            vqa_key* __restrict__ out_last_key, int query_major,
                                                                            const int* __restrict__ gate, int row_lists) {
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    vqa_key* keys = reinterpret_cast<vqa_key*>(smem);  // [m_pow2]
    const int q = blockIdx.x;
    const int m = parts * list_len;
    for (int i = threadIdx.x; i < m_pow2; i += kMergeThreads) {
        vqa_key v = 0ull;
        if (i < m) {
            if (query_major) {
                v = partial[(size_t)q * row_lists * list_len + i];
            } else {
                const int p = i / list_len, j = i - p * list_len;
                v = partial[((size_t)p * VQA_QUERY_TILE + q) * list_len + j];
            }
        }
        keys[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= m_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (m_pow2 >> 1); t += kMergeThreads) {
                const int lo = ((t / stride) * stride << 1) + (t % stride), hi = lo + stride;
                const bool desc = (lo & size) == 0;  // descending runs first: the final pass sorts the whole array descending
                const vqa_key a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < k; r += kMergeThreads) {
        const vqa_key best = r < m_pow2 ? keys[r] : 0ull;
        const bool empty = best == 0ull;
        const long long pos = empty ? -1 : (long long)vqa_key_pos(best);
        const size_t o = (size_t)q * out_stride + out_offset + r;
        if (out_scores) out_scores[o] = empty ? -INFINITY : vqa_key_score(best) * score_scale;
        if (out_ids) out_ids[o] = empty ? -1 : (ids ? ids[pos] : id_base + pos);
        if (out_pos) out_pos[o] = pos;
        if (out_thr && r == k - 1) out_thr[q] = empty ? -INFINITY : vqa_key_score(best);
        if (out_last_key && r == k - 1) out_last_key[q] = best;
    }
}

__global__ __launch_bounds__(kMergeThreads) void merge_shards_kernel(const float* __restrict__ scores,
                                                                     const long long* __restrict__ ids,
                                                                     long long score_rank_stride, long long id_rank_stride,
                                                                     int R, int B, int k, int k_out,
                                                                     float* __restrict__ out_scores,
                                                                     long long* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    vqa_key* keys = reinterpret_cast<vqa_key*>(smem);  // [R * k]
    vqa_key* red = keys + (size_t)R * k;  // [2][4]
    vqa_key* win = red + 8;  // [k_out] winners
    const int q = blockIdx.x;
    const int m = R * k;
    for (int i = threadIdx.x; i < m; i += kMergeThreads) {
        const int r = i / k, j = i - r * k;
        const size_t src = (size_t)q * k + j;  // inside rank r's [B, k] block
        // padded slots carry id -1: they must lose against every real candidate, including score -inf
        keys[i] = ids[(size_t)r * id_rank_stride + src] < 0 ? 0ull
                                                             : vqa_make_key(scores[(size_t)r * score_rank_stride + src], (uint32_t)i);
    }
    __syncthreads();
    if (k_out >= kRadixSelectK && m >= k_out) {
        select_topk_radix(keys, m, k_out, win, reinterpret_cast<int*>(win + k_out));
    } else {
    vqa_key prev = ~0ull;
    for (int r = 0; r < k_out; ++r) {
        vqa_key best = 0ull;
        for (int i = threadIdx.x; i < m; i += kMergeThreads) {
            const vqa_key v = keys[i];
            best = (v < prev && v > best) ? v : best;
        }
        best = block_max_key(best, red, r);
        if (threadIdx.x == 0) win[r] = best;  // written out after the last round (no global round trip inside a round: see merge_partials)
        prev = best;
    }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < k_out; r += kMergeThreads) {
        const vqa_key best = win[r];
        const bool empty = best == 0ull;
        size_t src = 0, rr = 0;
        if (!empty) {
            const int i = (int)vqa_key_pos(best);
            rr = (size_t)(i / k);
            src = (size_t)q * k + (size_t)(i - (int)rr * k);
        }
        out_scores[(size_t)q * k_out + r] = empty ? -INFINITY : scores[rr * score_rank_stride + src];
        out_ids[(size_t)q * k_out + r] = empty ? -1 : ids[rr * id_rank_stride + src];
    }
}

// One-pass large-k check (one thread per (query, workgroup list)): a FULL list whose last (smallest kept) key lies above the
// query's k-th merged key may have dropped a row that belongs to the top-k.
__global__ void verify_wide_kernel(const vqa_key* __restrict__ partial, int parts, int list_len, int nq,
                                   const vqa_key* __restrict__ kth, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= parts * nq) return;
    const int p = i / nq, q = i - p * nq;
    const vqa_key last = partial[((size_t)q * parts + p) * list_len + (list_len - 1)];  // lists are query-major
    if (last != 0ull && last > kth[q]) atomicExch(flag, 1);
}

}  // namespace

int vqa_launch_verify_wide(const vqa_key* partial, int32_t parts, int32_t list_len, int32_t nq, const vqa_key* kth, int* flag,
                           hipStream_t stream) {
    const int n = parts * nq;
    hipLaunchKernelGGL(verify_wide_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, partial, parts, list_len, nq, kth, flag);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_merge_partials(const vqa_key* partial, int32_t parts, int32_t list_len, int32_t nq, int32_t k,
                              const int64_t* ids, int64_t id_base, float* out_scores, int64_t* out_ids, int64_t* out_pos,
                              float* out_thr, float score_scale, int32_t out_stride, int32_t out_offset,
                              vqa_key* out_last_key, bool query_major, const int* gate, hipStream_t stream, int32_t row_lists,
                              const unsigned* counts, int32_t count_stride, const MergeSketchTail* tail) {
    if (row_lists <= 0) row_lists = parts;
    VQA_REQUIRE(row_lists >= parts, "merge_partials: %d lists per row but %d to merge", row_lists, parts);
    VQA_REQUIRE(parts >= 1 && list_len >= 1 && nq >= 1 && nq <= VQA_QUERY_TILE && k >= 1,
                "merge_partials: bad shape parts=%d list_len=%d nq=%d k=%d", parts, list_len, nq, k);
    // (the candidate lists of a sketch search may be longer than LDS: the kernel keeps the keys at or above tail->min_score, up to cap_keys)
    constexpr int kMaxLdsKeys = 18 * 1024;
    int cap_keys = parts * list_len;
    if (counts && cap_keys > kMaxLdsKeys) {
        VQA_REQUIRE(tail && tail->overflow, "merge_partials: %d x %d candidate keys need an overflow flag", parts, list_len);
        cap_keys = kMaxLdsKeys;
    }
    // keys, 2 x 4 reduction slots, the fill counter, k winners, the radix selection's histogram (k >= kRadixSelectK)
    const size_t lds = ((size_t)cap_keys + 10 + (size_t)k) * sizeof(vqa_key) + (256 + 4) * sizeof(int);
    VQA_REQUIRE(lds <= 160 * 1024, "merge_partials: %d lists x %d keys do not fit in LDS", parts, list_len);
    if (lds > 64 * 1024) {
        static VqaPerDeviceOnce once;
        int rc = once.run([&](int) -> int {
            VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(merge_partials_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return VQA_OK;
        });
        if (rc != VQA_OK) return rc;
    }
    if (k > 32 && !counts && !tail) {  // sort instead of k selection rounds
        int m_pow2 = 1;
        while (m_pow2 < parts * list_len) m_pow2 <<= 1;
        const size_t lds_sort = (size_t)m_pow2 * sizeof(vqa_key);
        if (lds_sort <= 64 * 1024) {
            hipLaunchKernelGGL(merge_partials_sort_kernel, dim3(nq), dim3(kMergeThreads), lds_sort, stream, partial, parts, list_len,
                               k, m_pow2, reinterpret_cast<const long long*>(ids), (long long)id_base, out_scores,
                               reinterpret_cast<long long*>(out_ids), reinterpret_cast<long long*>(out_pos), out_thr, score_scale,
                               out_stride, out_offset, out_last_key, query_major ? 1 : 0, gate, row_lists);
            VQA_HIP_CHECK(hipGetLastError());
            return VQA_OK;
        }
    }
    hipLaunchKernelGGL(merge_partials_kernel, dim3(nq), dim3(kMergeThreads), lds, stream, partial, parts, list_len, k,
                       reinterpret_cast<const long long*>(ids), (long long)id_base, out_scores,
                       reinterpret_cast<long long*>(out_ids), reinterpret_cast<long long*>(out_pos), out_thr, score_scale,
                       out_stride, out_offset, out_last_key, query_major ? 1 : 0, gate, row_lists, counts, count_stride,
                       tail ? *tail : MergeSketchTail{}, cap_keys);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

extern "C" int vqa_merge_topk(const float* scores, const int64_t* ids, int64_t score_rank_stride, int64_t id_rank_stride,
                              int32_t R, int32_t B, int32_t k, int32_t k_out, float* out_scores, int64_t* out_ids,
                              void* hip_stream) {
    VQA_REQUIRE(scores && ids && out_scores && out_ids, "vqa_merge_topk: null pointer");
    VQA_REQUIRE(R >= 1 && B >= 1 && k >= 1 && k_out >= 1, "vqa_merge_topk: bad shape R=%d B=%d k=%d k_out=%d", R, B, k,
                k_out);
    VQA_REQUIRE((long long)R * k <= 8 * 1024, "vqa_merge_topk: R*k=%lld candidates per query exceed 8192",
                (long long)R * k);
    VQA_REQUIRE(k_out <= R * k, "vqa_merge_topk: k_out=%d exceeds the R*k=%d candidates", k_out, R * k);
    if (score_rank_stride == 0) score_rank_stride = (int64_t)B * k;
    if (id_rank_stride == 0) id_rank_stride = (int64_t)B * k;
    VQA_REQUIRE(score_rank_stride >= (int64_t)B * k && id_rank_stride >= (int64_t)B * k,
                "vqa_merge_topk: rank strides %lld / %lld are smaller than one [B, k] block", (long long)score_rank_stride,
                (long long)id_rank_stride);
    // keys, 2 x 4 reduction slots, k_out winners, the radix selection's histogram: above the 64 KiB default near the R * k = 8192 limit
    const size_t lds = ((size_t)R * k + 8 + (size_t)k_out) * sizeof(vqa_key) + (256 + 4) * sizeof(int);
    if (lds > 64 * 1024) {
        static VqaPerDeviceOnce once;
        int rc = once.run([&](int) -> int {
            VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(merge_shards_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
            return VQA_OK;
        });
        if (rc != VQA_OK) return rc;
    }
    hipLaunchKernelGGL(merge_shards_kernel, dim3(B), dim3(kMergeThreads), lds, (hipStream_t)hip_stream, scores,
                       reinterpret_cast<const long long*>(ids), (long long)score_rank_stride, (long long)id_rank_stride, R, B, k,
                       k_out, out_scores,
                       reinterpret_cast<long long*>(out_ids));
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}
