// K1 -- fused query x corpus scoring (MFMA) + running top-k (LDS candidate lists), gfx950 only.
//
// Replaces the arithmetic behind `embeddings.search(query, limit)` (reference call site
// inference_pipeline/db_utils/heavy_ranker.py:98-101; in the reference that is txtai -> faiss IndexFlatIP:
// exact inner product of the query with every stored row + k largest).
//
// Shape of the work: S[q, n] = sum_j Q[q, j] * X[n, j]   with Q [256, d] (one query tile), X [N, d] row-major.
// One persistent workgroup (8 waves) walks corpus tiles of 256 rows:
//   * per K-step of 64 elements, the X slice (256 rows x 128 B, from HBM) and the Q slice (256 rows x 128 B,
//     L2 resident) are copied global -> LDS with 16-byte LDS-DMA (`global_load_lds_dwordx4`), double buffered;
//     the LDS image is XOR-swizzled through the SOURCE address so ds_read_b128 fragment reads are conflict free;
//   * wave (wm, wn) owns rows [128 wm, +128) x queries [64 wn, +64) as 8 x 4 tiles of v_mfma_f32_16x16x32_f16
//     (A = corpus rows, B = queries, so a lane's 32 accumulators of one column group all belong to ONE query);
//   * after the K loop the scores never leave registers: each lane compares its accumulators with the query's
//     current threshold (k-th best score so far, kept in LDS); survivors are appended to the query's LDS
//     candidate list with one LDS atomic, lists are compacted (rank-by-counting inside one wave) when they fill.
// No B x N score matrix is ever written.  The per-workgroup lists are flushed once at the end and merged by K2.
#include "vqa_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int kThreads = 512;           // 8 waves: 2 (row halves) x 4 (query quarters)
constexpr int kTileRows = 256;          // corpus rows per tile
constexpr int kQ = VQA_QUERY_TILE;      // 256 queries per tile
constexpr int kBK = 64;                 // K-step in elements
constexpr int kRowBytes = kBK * 2;      // 128 B of fp16 per row per K-step
constexpr int kOperandBytes = 256 * kRowBytes;  // 32 KiB: one operand slice
constexpr int kStageBytes = 2 * kOperandBytes;  // X slice + Q slice
constexpr int kPipeBytes = 2 * kStageBytes;     // double buffered: 128 KiB
constexpr int kLdsTotal = 160 * 1024;
constexpr int kCap = (kLdsTotal - kPipeBytes - 2 * kQ * 4) / (kQ * 8);  // 15 candidate slots per query
constexpr int kMaxK = kCap - 3;                                         // 12

constexpr int kOverBit = 1 << 30;  // "an append was refused" flag, kept in bit 30 of cnt[0] (LDS is fully used)
constexpr int kCntMask = 0xFFFFFF;

struct Lists {  // views into LDS: per query {threshold, count, kCap keys} = 128 B, 32 KiB in all
    float* thr;      // [256] current k-th best score per query (+inf for padded queries)
    int* cnt;        // [256] appends attempted since the last compaction (may exceed kCap when some were refused)
    vqa_key* cand;   // [256][kCap]
};

__device__ __forceinline__ int list_count(const Lists& L, int q) { return L.cnt[q] & kCntMask; }

// ---- rank-by-counting compaction of one query's list by one wave ------------------------------------------
// Keeps the k best of the (at most kCap) stored keys, sorted best first, and raises the threshold.
__device__ __forceinline__ void compact_query(const Lists& L, int q, int k, int lane) {
    const int raw = L.cnt[q];
    int n = raw & kCntMask;
    n = n < kCap ? n : kCap;
    vqa_key* c = L.cand + q * kCap;
    const vqa_key mine = lane < n ? c[lane] : 0ull;
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (c[j] > mine) ? 1 : 0;  // broadcast reads; keys are distinct rows
    if (lane < n && rank < k) c[rank] = mine;
    if (lane < n && rank == k - 1) L.thr[q] = vqa_key_score(mine);
    if (lane == 0) L.cnt[q] = (raw & kOverBit) | (n < k ? n : k);
}

// wave w compacts the queries [32 w, 32 w + 32) whose list length reached `water`
__device__ __forceinline__ void compact_pass(const Lists& L, int wave, int lane, int k, int water) {
    const int q0 = wave * 32;
    const int c = lane < 32 ? (L.cnt[q0 + lane] & kCntMask) : 0;
    unsigned long long need = __ballot(c >= water) & 0xFFFFFFFFull;
    while (need) {
        const int b = __builtin_ctzll(need);
        need &= need - 1;
        compact_query(L, q0 + b, k, lane);
    }
}

__device__ __forceinline__ bool append_candidate(const Lists& L, int q, float v, uint32_t pos) {
    const int slot = atomicAdd(&L.cnt[q], 1) & kCntMask;  // ds_add_rtn_u32
    if (slot < kCap) {
        L.cand[q * kCap + slot] = vqa_make_key(v, pos);
        return true;
    }
    return false;
}

// value of accumulator (mi, j) = bit index b of one query column group, selected without dynamic register indexing
__device__ __forceinline__ float select_acc(const f32x4 (&acc)[8][4], int ni, int b) {
    float sel = acc[0][ni][0];
#pragma unroll
    for (int i = 1; i < 32; ++i) sel = (b == i) ? acc[i >> 2][ni][i & 3] : sel;  // 31 x (v_cmp_eq + v_cndmask)
    return sel;
}

// Append this lane's pending accumulators (bit mi*4+j of pend[ni]) that still beat their query's threshold.
// Returns true when some append was refused (list full): the bit stays set for the next round.
__device__ __forceinline__ bool process_pending(const Lists& L, const f32x4 (&acc)[8][4], uint32_t (&pend)[4], int wm,
                                                int wn, int c, int g, long long row0) {
    bool refused = false;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        uint32_t bits = pend[ni];
        if (bits) {
            const int q = wn * 64 + ni * 16 + c;
            const float th = L.thr[q];
            uint32_t keep = 0;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const float v = select_acc(acc, ni, b);
                if (v >= th) {
                    const int r = wm * 128 + (b >> 2) * 16 + g * 4 + (b & 3);
                    if (!append_candidate(L, q, v, (uint32_t)(row0 + r))) keep |= 1u << b;
                }
            }
            pend[ni] = keep;
            refused |= keep != 0;
        }
    }
    return refused;
}

// kSeeded = false: first pass (thresholds start at -inf); true: main pass seeded with the k-th best scores of the
// first pass.  Same code; two instantiations so that profiles name the dominant (main) kernel separately.
template <bool kSeeded>
__global__ __launch_bounds__(kThreads) void score_topk_f16_kernel(const _Float16* __restrict__ X,
                                                                  const _Float16* __restrict__ Qs,
                                                                  const float* __restrict__ thr_init,
                                                                  vqa_key* __restrict__ partial, long long N, int D,
                                                                  int nq, int k, int tile_begin, int tile_end) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lists L;
    L.thr = reinterpret_cast<float*>(smem + kPipeBytes);
    L.cnt = reinterpret_cast<int*>(smem + kPipeBytes + kQ * 4);
    L.cand = reinterpret_cast<vqa_key*>(smem + kPipeBytes + 2 * kQ * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2;  // corpus-row half
    const int wn = wave & 3;   // query quarter
    const int c = lane & 15;
    const int g = lane >> 4;

    if (tid < kQ) {
        L.thr[tid] = tid < nq ? (kSeeded ? thr_init[tid] : -INFINITY) : INFINITY;
        L.cnt[tid] = 0;
    }

    const int KS = D / kBK;
    const int first_tile = tile_begin + blockIdx.x;
    const int ntile = first_tile < tile_end ? (tile_end - first_tile + gridDim.x - 1) / gridDim.x : 0;
    const long long total_steps = (long long)ntile * KS;

    // ---- staging geometry: wave-instruction i of wave w fills LDS units [(4w+i)*64, +64) of an operand slice
    // (unit = 16 B; 8 units per 128-B row).  Lane l -> row (4w+i)*8 + (l>>3), swizzled slot l&7, source slot
    // (l&7) ^ (row&7) = (l&7) ^ (l>>3).
    const int st_row_in = lane >> 3;
    const int st_src_slot = (lane & 7) ^ st_row_in;
    const size_t row_bytes_g = (size_t)D * 2;

    auto stage = [&](long long step) {
        const int ti = (int)(step / KS);
        const int ks = (int)(step - (long long)ti * KS);
        const long long row0 = ((long long)first_tile + (long long)ti * gridDim.x) * kTileRows;
        char* buf = smem + (step & 1) * kStageBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + st_row_in;
            long long gr = row0 + r;
            gr = gr < N ? gr : N - 1;  // tail rows re-read the last row; masked in the epilogue
            const char* src = reinterpret_cast<const char*>(X) + (size_t)gr * row_bytes_g + (size_t)ks * kRowBytes +
                              st_src_slot * 16;
            char* dst = buf + (wave * 4 + i) * 1024;  // wave-uniform base; hardware adds lane * 16
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + st_row_in;
            const char* src = reinterpret_cast<const char*>(Qs) + (size_t)r * row_bytes_g + (size_t)ks * kRowBytes +
                              st_src_slot * 16;
            char* dst = buf + kOperandBytes + (wave * 4 + i) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // ---- fragment geometry: lane reads 16 B = 8 k-contiguous halves of row (base + c), k offset 32 kk + 8 g
    const int frag_off0 = c * kRowBytes + (((0 + g) ^ (c & 7)) << 4);
    const int frag_off1 = c * kRowBytes + (((4 + g) ^ (c & 7)) << 4);
    const int a_base = wm * 128 * kRowBytes;                 // inside the X slice
    const int b_base = kOperandBytes + wn * 64 * kRowBytes;  // inside the Q slice

    if (total_steps > 0) stage(0);

    for (int ti = 0; ti < ntile; ++ti) {
        f32x4 acc[8][4];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ks = 0; ks < KS; ++ks) {
            const long long step = (long long)ti * KS + ks;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my LDS-DMA pieces of `step` have landed
            __syncthreads();                                   // everyone's have; buffer (step+1)&1 is free
            if (step + 1 < total_steps) stage(step + 1);
            const char* buf = smem + (step & 1) * kStageBytes;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int fo = kk == 0 ? frag_off0 : frag_off1;
                half8 a[8], b[4];
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    a[mi] = *reinterpret_cast<const half8*>(buf + a_base + mi * 16 * kRowBytes + fo);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    b[ni] = *reinterpret_cast<const half8*>(buf + b_base + ni * 16 * kRowBytes + fo);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }

        // ---- epilogue: threshold filter + candidate append; scores stay in registers --------------------------
        // acc[mi][ni][j] = score(row = row0 + 128 wm + 16 mi + 4 g + j, query = 64 wn + 16 ni + c)
        const long long row0 = ((long long)first_tile + (long long)ti * gridDim.x) * kTileRows;
        const long long rows_left = N - row0;  // rows of this tile that exist (may exceed 256)
        uint32_t pend[4];
        bool any = false;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const float th = L.thr[wn * 64 + ni * 16 + c];
            float m = -INFINITY;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
                m = fmaxf(m, fmaxf(fmaxf(acc[mi][ni][0], acc[mi][ni][1]), fmaxf(acc[mi][ni][2], acc[mi][ni][3])));
            uint32_t bits = 0;
            if (m >= th) {
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = wm * 128 + mi * 16 + g * 4 + j;
                        bits |= (acc[mi][ni][j] >= th && r < rows_left) ? (1u << (mi * 4 + j)) : 0u;
                    }
            }
            pend[ni] = bits;
            any |= bits != 0;
        }
        if (any && process_pending(L, acc, pend, wm, wn, c, g, row0)) atomicOr(&L.cnt[0], kOverBit);
        __syncthreads();
        for (;;) {
            const int over = L.cnt[0] & kOverBit;
            // normal tiles: compact lists that are nearly full; after a refusal: compact everything above k
            compact_pass(L, wave, lane, k, over ? k + 1 : k + (kCap - k + 1) / 2);
            __syncthreads();
            if (!over) break;
            if (tid == 0) L.cnt[0] &= ~kOverBit;
            __syncthreads();
            if (process_pending(L, acc, pend, wm, wn, c, g, row0)) atomicOr(&L.cnt[0], kOverBit);
            __syncthreads();
        }
    }

    // ---- flush: every list sorted best first, k keys per query (0 = empty) ------------------------------------
    __syncthreads();
    compact_pass(L, wave, lane, k, 1);
    __syncthreads();
    vqa_key* out = partial + (size_t)blockIdx.x * kQ * k;
    for (int i = tid; i < kQ * k; i += kThreads) {
        const int q = i / k, j = i - q * k;
        out[i] = j < list_count(L, q) ? L.cand[q * kCap + j] : 0ull;
    }
}

}  // namespace

int vqa_score_topk_lds_bytes(int dtype, int k) {
    (void)dtype;
    (void)k;
    return kPipeBytes + 2 * kQ * 4 + kQ * kCap * 8;
}

int vqa_score_topk_max_k(int dtype) {
    (void)dtype;
    return kMaxK;
}

int vqa_launch_score_topk(int dtype, const ScoreTopkArgs& a, hipStream_t stream) {
    VQA_REQUIRE(dtype == VQA_F16, "score_topk: only fp16 rows are implemented (dtype %d)", dtype);
    VQA_REQUIRE(a.d_pad > 0 && a.d_pad % kBK == 0, "score_topk: padded row length %d is not a multiple of %d", a.d_pad, kBK);
    VQA_REQUIRE(a.k >= 1 && a.k <= kMaxK, "score_topk: k=%d outside [1, %d]", a.k, kMaxK);
    VQA_REQUIRE(a.nq >= 1 && a.nq <= kQ, "score_topk: nq=%d outside [1, %d]", a.nq, kQ);
    VQA_REQUIRE(a.grid >= 1 && a.tile_end > a.tile_begin, "score_topk: empty launch");
    const int lds = vqa_score_topk_lds_bytes(dtype, a.k);
    static bool attr_set_dev[64] = {};
    int dev = 0;
    VQA_HIP_CHECK(hipGetDevice(&dev));
    bool& attr_set = attr_set_dev[dev & 63];
    if (!attr_set) {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_f16_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_f16_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    auto kern = a.thr_init ? score_topk_f16_kernel<true> : score_topk_f16_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(kThreads), lds, stream, reinterpret_cast<const _Float16*>(a.x),
                       reinterpret_cast<const _Float16*>(a.q), a.thr_init, a.partial, (long long)a.n, a.d_pad, a.nq, a.k,
                       a.tile_begin, a.tile_end);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}
