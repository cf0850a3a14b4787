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
#include <stdlib.h>

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
// UM = 16-byte units per lane and pair held in registers (2, 3, 4, 6 or 8: rows of up to 32 * 8 units = 2048 fp16 / 1024 fp32 elements;
// 0 = the loop form for longer rows).  The launcher picks the smallest that covers the row: at d = 768 fp16 (3 units) the kernel
// takes 80 instead of 210 registers of the 8-unit form, i.e. 6 instead of 2 waves per SIMD to hide the row fetches behind.

// T = _Float16 (8 elements per 16-byte unit; products exact in fp32) or float (4 per unit: fp32 index, fp32 fma chain)
// RM: X is the shard's ROW-MAJOR copy (VQA_INDEX_RESCORE_ROWS: rows of KT * 64 bytes, the same stored values) -- a pair's row is one
// contiguous run of whole 128-byte lines instead of 4 KT pieces of 64 bytes at 16 KiB strides -- and Q a row-major copy of the staged
// query tile (a half wave's load is 512 contiguous bytes instead of eight 64-byte pieces 16 KiB apart: the kernel issues two loads per
// product, and the query side's were the less coalesced).  Unit u holds the same elements in both layouts, so the two forms add the
// same products in the same order: bit-equal scores.
// PH = pairs per half wave and iteration (2: four pairs per half wave measured 25 % slower at d = 768 -- 96 data registers leave
// fewer waves per SIMD): every load of an iteration -- the rows of its 2 PH pairs and the pair descriptors of the NEXT iteration
// -- is in flight before the first is consumed.  The kernel is a chain of dependent round trips per iteration -- descriptor, row,
// counter -- at ~20 waves per CU.
template <typename T, bool RM, int UM, int PH>
__global__ __launch_bounds__(256) void rescore_kernel(const unsigned long long* __restrict__ regions, const unsigned* __restrict__ counts,
                                                      int cap, int nregions, const long long* __restrict__ stage_pos, int nq, int k,
                                                      const T* __restrict__ X, const T* __restrict__ Q, int KT,
                                                      vqa_key* __restrict__ cand_keys, unsigned* __restrict__ cand_cnt, int capq,
                                                      int* __restrict__ overflow) {
    constexpr int EPU = 16 / (int)sizeof(T);
    typedef T unit_t __attribute__((ext_vector_type(EPU)));
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int y = blockIdx.y;
    const int total = y < nregions ? (int)counts[y] : nq * k;
    // (the host's profitability check: pairs this query tile scores exactly, one atomic per region and launch)
    if (blockIdx.x == 0 && threadIdx.x == 0 && y < nregions && total > 0) atomicAdd(overflow + 3, total < cap ? total : cap);
    const int units = KT * 4;
    const int sub_cap = capq / kSketchSubLists;
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
    int q[PH];
    long long pos[PH];
#pragma unroll
    for (int p = 0; p < PH; ++p) pair_of(wave * 2 * PH + 2 * p + half, q[p], pos[p]);
    for (int i0 = wave * 2 * PH; i0 < total; i0 += nwaves * 2 * PH) {
        float acc[PH];
#pragma unroll
        for (int p = 0; p < PH; ++p) acc[p] = 0.f;
        int qn[PH];
        long long posn[PH];
        if constexpr (UM > 0) {
            unit_t xv[PH][UM], qv[PH][UM];
#pragma unroll
            for (int p = 0; p < PH; ++p)
#pragma unroll
                for (int j = 0; j < UM; ++j) {
                    const int u = hl + 32 * j;
                    xv[p][j] = qv[p][j] = unit_t{};
                    if (u < units && pos[p] >= 0) {
                        xv[p][j] = *reinterpret_cast<const unit_t*>(X + (RM ? (size_t)pos[p] * units + u : unit_of(pos[p], u, KT)) * EPU);
                        qv[p][j] = *reinterpret_cast<const unit_t*>(Q + (RM ? (size_t)q[p] * units + u : unit_of(q[p], u, KT)) * EPU);
                    }
                }
#pragma unroll
            for (int p = 0; p < PH; ++p) pair_of(i0 + nwaves * 2 * PH + 2 * p + half, qn[p], posn[p]);  // the next iteration's pairs
#pragma unroll
            for (int p = 0; p < PH; ++p)
#pragma unroll
                for (int j = 0; j < UM; ++j)
#pragma unroll
                    for (int e = 0; e < EPU; ++e) acc[p] = __builtin_fmaf((float)xv[p][j][e], (float)qv[p][j][e], acc[p]);
        } else {
#pragma unroll
            for (int p = 0; p < PH; ++p) pair_of(i0 + nwaves * 2 * PH + 2 * p + half, qn[p], posn[p]);
            for (int p = 0; p < PH; ++p)
                for (int u = hl; u < units && pos[p] >= 0; u += 32) {
                    const unit_t xw = *reinterpret_cast<const unit_t*>(X + (RM ? (size_t)pos[p] * units + u : unit_of(pos[p], u, KT)) * EPU);
                    const unit_t qw = *reinterpret_cast<const unit_t*>(Q + (RM ? (size_t)q[p] * units + u : unit_of(q[p], u, KT)) * EPU);
#pragma unroll
                    for (int e = 0; e < EPU; ++e) acc[p] = __builtin_fmaf((float)xw[e], (float)qw[e], acc[p]);
                }
        }
#pragma unroll
        for (int p = 0; p < PH; ++p) {
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) acc[p] += __shfl_xor(acc[p], off, 64);  // inside the half wave
            if (hl == 0 && pos[p] >= 0) {
                // sub-list of the query's list ([query][kSketchSubLists][capq / kSketchSubLists]): picked by region + the pair's index in it
                // (groups of four), so that both a query with a few candidates per region (B = 1: every region's first pairs) and one
                // with thousands in ONE region use all the sub-lists; the appends of one search spread over 16x the counters
                const unsigned sub = (unsigned)(y + ((i0 + 2 * p) >> 2)) % kSketchSubLists;
                const unsigned lane_list = (unsigned)q[p] * kSketchSubLists + sub;
                const unsigned slot = atomicAdd(cand_cnt + (size_t)lane_list * kSketchCntStride, 1u);
                if (slot < (unsigned)sub_cap) cand_keys[(size_t)lane_list * sub_cap + slot] = vqa_make_key(acc[p], (uint32_t)pos[p]);
                else atomicExch(overflow, 1);
            }
        }
#pragma unroll
        for (int p = 0; p < PH; ++p) {
            q[p] = qn[p];
            pos[p] = posn[p];
        }
    }
}

// ---- the exact paths' last step on a large shard: the rows they selected, scored again in THIS file's arithmetic and re-ranked.
// A shard large enough for the sketch search has two ways to answer a search -- the sketch cascade (scores = rescore_kernel's fma chain)
// and the exact scan (scores = MFMA sums; during a pause, behind an overflow, for sketch = off) -- and which one runs can depend on an
// asynchronous report (include/vqa_retrieval.h: vqa_index_sketch_state).  MFMA sums and the fma chain differ in the last bits, so the
// two paths could return near-tied rows in another order ("two paths, last-bit different, chosen asynchronously").  Here the
// exact scan's `kin` best rows per query (kin = k + 2 where the scan's lists have room: rows the MFMA order ranks just below the k-th
// are candidates too) get the fma-chain score -- the same loads, the same fma order, the same butterfly as rescore_kernel -- and the
// `kout` best by (that score descending, position ascending) are the result: the same bits the sketch path returns for the same rows.
// One workgroup per query: a half wave per candidate row, then rank by counting over the <= 1024 keys in LDS.
template <typename T, bool RM>
__global__ __launch_bounds__(256) void final_rescore_kernel(const long long* __restrict__ pos_in, int in_stride, int nq, int kin, int kout,
                                                            const T* __restrict__ X, const T* __restrict__ Q, int KT,
                                                            const long long* __restrict__ ids, long long id_base,
                                                            float* __restrict__ out_scores, long long* __restrict__ out_ids,
                                                            long long* __restrict__ out_pos, int out_stride, const int* __restrict__ gate) {
    if (gate && *gate == 0) return;
    constexpr int EPU = 16 / (int)sizeof(T);
    typedef T unit_t __attribute__((ext_vector_type(EPU)));
    __shared__ vqa_key keys[VQA_MAX_K_TOTAL];
    __shared__ long long posv[VQA_MAX_K_TOTAL];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, hl = lane & 31, half = tid >> 5;  // 8 half waves
    const int units = KT * 4;
    for (int i = half; i < kin; i += 8) {
        const long long pos = pos_in[(size_t)q * in_stride + i];
        float acc = 0.f;
        if (pos >= 0)
            for (int u = hl; u < units; u += 32) {
                const unit_t xw = *reinterpret_cast<const unit_t*>(X + (RM ? (size_t)pos * units + u : unit_of(pos, u, KT)) * EPU);
                const unit_t qw = *reinterpret_cast<const unit_t*>(Q + (RM ? (size_t)q * units + u : unit_of(q, u, KT)) * EPU);
#pragma unroll
                for (int e = 0; e < EPU; ++e) acc = __builtin_fmaf((float)xw[e], (float)qw[e], acc);
            }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);  // inside the half wave
        if (hl == 0) {
            keys[i] = pos >= 0 ? vqa_make_key(acc, (uint32_t)pos) : 0ull;
            posv[i] = pos;
        }
    }
    __syncthreads();
    for (int i = tid; i < kin; i += 256) {
        const vqa_key mine = keys[i];
        if (mine == 0ull) continue;
        int rank = 0;
        for (int j = 0; j < kin; ++j) rank += keys[j] > mine ? 1 : 0;  // keys of real rows are distinct (the position is part of the key)
        if (rank < kout) {
            const size_t o = (size_t)q * out_stride + rank;
            const long long pos = posv[i];
            out_scores[o] = vqa_key_score(mine);
            out_ids[o] = ids ? ids[pos] : id_base + pos;
            if (out_pos) out_pos[o] = pos;
        }
    }
    // fewer real rows than kout: padding behind them, as the merges write it
    __shared__ int nreal;
    if (tid == 0) nreal = 0;
    __syncthreads();
    int mine_real = 0;
    for (int i = tid; i < kin; i += 256) mine_real += keys[i] != 0ull ? 1 : 0;
    if (mine_real) atomicAdd(&nreal, mine_real);
    __syncthreads();
    for (int r = nreal + tid; r < kout; r += 256) {
        const size_t o = (size_t)q * out_stride + r;
        out_scores[o] = -INFINITY;
        out_ids[o] = -1;
        if (out_pos) out_pos[o] = -1;
    }
}

// rows [first, first + count) of the tiled shard -> the row-major re-scoring copy (rows of KT * 64 bytes, zero padded as stored);
// one wave per row, 16 bytes per lane and step
__global__ __launch_bounds__(256) void rows_to_rowmajor_kernel(const uint4* __restrict__ tiled, long long first, long long count, int KT,
                                                                uint4* __restrict__ out) {
    const long long r = first + (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= first + count) return;
    for (int u = threadIdx.x & 63; u < KT * 4; u += 64) out[(size_t)r * (KT * 4) + u] = tiled[unit_of(r, u, KT)];
}

// per query: theta (the exact k-th best score of the first stage), ||q_lo||, ||z_r|| (+ fp margin), 1 / s_q, |alpha| -> qconst [5][256];
// clears the candidate counters and the overflow flag of this search
__global__ void sketch_qconst_kernel(const float* __restrict__ thr, const float* __restrict__ qscale, const float* __restrict__ qlo,
                                     const float* __restrict__ qnorm, float fp_margin, float* __restrict__ qconst,
                                     unsigned* __restrict__ cand_cnt, int* __restrict__ overflow, int clear, int seq,
                                     const float* __restrict__ qoff, float mu_margin, const float* __restrict__ qalpha,
                                     const float* __restrict__ qrnorm) {
    const int q = threadIdx.x;
    // a centred sketch bounds q . (x - mu): the threshold moves by q . mu (an fp64 dot rounded once to fp32, and this subtraction:
    // mu_margin = 3e-7 ||mu|| per unit of ||q||; the rounding of x - mu itself, 2^-24 ||x - mu||, rides in fp_margin)
    qconst[q] = qoff ? thr[q] - qoff[q] - mu_margin * qnorm[q] : thr[q];
    // The scores a search returns -- and theta -- are fp32 sums, the bound speaks of the real-number dot product: both differ
    // from it by at most gamma_d ||q|| ||x|| (d terms, unit roundoff 2^-24 per fma; for the MFMA's internal order as well),
    // ||x|| <= ||x_hi|| + ||x_lo||.  fp_margin = 2 gamma_d rides on BOTH slack terms: ||q_lo|| A + ||q|| B + fp_margin ||q|| (A + B)
    // = (||q_lo|| + fp_margin ||q||) A + (||q|| + fp_margin ||q||) B.  With the slack term split along the centre direction (convert.hip
    // sketch_rows_kernel: z = alpha w + z_r) the B term reads |alpha| C + (||z_r|| + fp_margin ||q||) B, C = max_tile |w . x_lo|.
    qconst[256 + q] = qlo[q] + fp_margin * qnorm[q];
    qconst[512 + q] = (qrnorm ? qrnorm[q] : qnorm[q]) + fp_margin * qnorm[q];
    qconst[768 + q] = 1.0f / qscale[q];
    qconst[1024 + q] = qalpha ? qalpha[q] : 0.f;  // |alpha| (split slack term) or the signed alpha (per-row form)
    qconst[1280 + q] = fp_margin * qnorm[q];       // per-row form: the margin's factor on the tile's max |beta|
    if (clear) {  // (the second scan of a cascade keeps what the first one found)
#pragma unroll
        for (int j = 0; j < kSketchSubLists; ++j) cand_cnt[(size_t)(q * kSketchSubLists + j) * kSketchCntStride] = 0u;
        if (q == 0) {
            // overflow[0]: this query tile's flag (gates its exact fallback); overflow[1]: OR over the earlier tiles of the call
            // (clear == 2: the call's first tile), so that the host's cool-down sees an overflow of ANY tile of a B > 256 call
            overflow[1] = clear == 2 ? 0 : (overflow[1] | overflow[0]);
            overflow[0] = 0;
            overflow[2] = seq;  // which call of the handle these flags belong to (the host's cool-down bookkeeping, capi.hip)
            overflow[4] = clear == 2 ? 0 : (overflow[4] > overflow[3] ? overflow[4] : overflow[3]);  // the most an earlier tile of the call scored
            overflow[3] = 0;    // pairs scored exactly for this query tile (rescore_kernel adds)
        }
    }
}

}  // namespace

float vqa_sketch_fp_margin(int32_t d, bool rotated, bool per_row) {
    // 2 gamma_d with gamma_d <= d 2^-24 / (1 - d 2^-24) < 1.2e-7 d / 2 ... kept at twice that; a rotated sketch adds the rounding of
    // the two rotations (13 butterfly stages + the normalisation: 14 2^-24 per side, kept at twice that too)
    // per-row form: beta = fl(w . y) (gamma_d ||y||, times |alpha| <= ||q||), the two element-wise projections (2^-23 each) and
    // beta (z_r . w) with |z_r . w| <= (gamma_d + 3e-6) ||z|| stand between z . y = alpha beta + z_r . y_r and what the kernels compute:
    // 2 gamma_d + 4e-6 more, on A + B and on the tile's max |beta| (qconst row 5) alike
    return 2.0f * (float)d * 1.2e-7f + (rotated ? 4.0f * 14.0f * 6e-8f : 0.f) + (per_row ? (float)d * 1.2e-7f + 4e-6f : 0.f);
}

int vqa_launch_sketch_qconst(const float* thr, const float* qscale, const float* qlo, const float* qnorm, int32_t d, float* qconst,
                             unsigned* cand_cnt, int* overflow, int clear, int seq, bool rotated, const float* qoff, float mu_margin,
                             hipStream_t stream, const float* qalpha, const float* qrnorm, bool per_row) {
    const float fp_margin = vqa_sketch_fp_margin(d, rotated, per_row);
    hipLaunchKernelGGL(sketch_qconst_kernel, dim3(1), dim3(256), 0, stream, thr, qscale, qlo, qnorm, fp_margin, qconst, cand_cnt, overflow, clear, seq, qoff,
                       mu_margin, qalpha, qrnorm);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_rows_to_rowmajor(const void* tiled, int64_t first, int64_t count, int32_t row_bytes, void* out, hipStream_t stream) {
    if (count <= 0) return VQA_OK;
    VQA_REQUIRE(row_bytes % 64 == 0, "rows_to_rowmajor: rows of %d bytes", row_bytes);
    hipLaunchKernelGGL(rows_to_rowmajor_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(tiled),
                       (long long)first, (long long)count, row_bytes / 64, reinterpret_cast<uint4*>(out));
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_rescore(const unsigned long long* regions, const unsigned* counts, int cap, int nregions, const long long* stage_pos,
                       int nq, int k, const void* x, const void* x_rowmajor, const void* q, const void* q_rowmajor, int32_t dtype, int32_t d_pad,
                       vqa_key* cand_keys, unsigned* cand_cnt, int capq, int* overflow, hipStream_t stream) {
    VQA_REQUIRE(!x_rowmajor || q_rowmajor, "rescore: the row-major form needs both copies");
    VQA_REQUIRE(dtype == VQA_F16 || dtype == VQA_F32, "rescore: storage type %d", dtype);
    // six workgroups of four waves per region (measured at 10M x 768, 142k / 214k pairs: grid x = 32 / 8 / 6 / 4 / 2 -> 94 / 74 / 58 /
    // 62 / 85 us for the first stage's pairs; fewer, longer-lived waves amortise the block dispatch, too few leave CUs idle)
    const dim3 grid(6, nregions + 1), block(256);
#define VQA_RESCORE_UM(T, RM, UMV, XP, KTV)                                                                                          \
    hipLaunchKernelGGL((rescore_kernel<T, RM, UMV, 2>), grid, block, 0, stream, regions, counts, cap, nregions, stage_pos, nq, k,    \
                       reinterpret_cast<const T*>(XP), reinterpret_cast<const T*>(RM ? q_rowmajor : q), KTV, cand_keys, cand_cnt, capq, overflow)
#define VQA_RESCORE(T, RM, XP, KTV)                                                                                                  \
    do {                                                                                                                             \
        const int per = ((KTV) * 4 + 31) / 32; /* units per lane */                                                                  \
        if (per <= 2) VQA_RESCORE_UM(T, RM, 2, XP, KTV);                                                                             \
        else if (per <= 3) VQA_RESCORE_UM(T, RM, 3, XP, KTV);                                                                        \
        else if (per <= 4) VQA_RESCORE_UM(T, RM, 4, XP, KTV);                                                                        \
        else if (per <= 6) VQA_RESCORE_UM(T, RM, 6, XP, KTV);                                                                        \
        else if (per <= 8) VQA_RESCORE_UM(T, RM, 8, XP, KTV);                                                                        \
        else VQA_RESCORE_UM(T, RM, 0, XP, KTV);                                                                                      \
    } while (0)
    if (dtype == VQA_F16) {
        if (x_rowmajor) VQA_RESCORE(_Float16, true, x_rowmajor, d_pad / 32);
        else VQA_RESCORE(_Float16, false, x, d_pad / 32);
    } else {
        if (x_rowmajor) VQA_RESCORE(float, true, x_rowmajor, d_pad / 16);
        else VQA_RESCORE(float, false, x, d_pad / 16);
    }
#undef VQA_RESCORE
#undef VQA_RESCORE_UM
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_final_rescore(const int64_t* pos_in, int in_stride, int nq, int kin, int kout, const void* x, const void* x_rowmajor,
                             const void* q, const void* q_rowmajor, int32_t dtype, int32_t d_pad, const int64_t* ids, int64_t id_base,
                             float* out_scores, int64_t* out_ids, int64_t* out_pos, int out_stride, const int* gate, hipStream_t stream) {
    VQA_REQUIRE(dtype == VQA_F16 || dtype == VQA_F32, "final_rescore: storage type %d", dtype);
    VQA_REQUIRE(kin >= kout && kin <= VQA_MAX_K_TOTAL && kout >= 1, "final_rescore: kin=%d kout=%d", kin, kout);
    const bool rm = x_rowmajor && q_rowmajor;
#define VQA_FINAL(T, RMV, XP, QP, KTV)                                                                                                  \
    hipLaunchKernelGGL((final_rescore_kernel<T, RMV>), dim3(nq), dim3(256), 0, stream, reinterpret_cast<const long long*>(pos_in), in_stride, \
                       nq, kin, kout, reinterpret_cast<const T*>(XP), reinterpret_cast<const T*>(QP), KTV,                                \
                       reinterpret_cast<const long long*>(ids), (long long)id_base, out_scores, reinterpret_cast<long long*>(out_ids),     \
                       reinterpret_cast<long long*>(out_pos), out_stride, gate)
    if (dtype == VQA_F16) {
        if (rm) VQA_FINAL(_Float16, true, x_rowmajor, q_rowmajor, d_pad / 32);
        else VQA_FINAL(_Float16, false, x, q, d_pad / 32);
    } else {
        if (rm) VQA_FINAL(float, true, x_rowmajor, q_rowmajor, d_pad / 16);
        else VQA_FINAL(float, false, x, q, d_pad / 16);
    }
#undef VQA_FINAL
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}
