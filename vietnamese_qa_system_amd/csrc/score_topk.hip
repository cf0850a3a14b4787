// K1 -- fused query x corpus scoring (MFMA) + running top-k (LDS candidate lists), gfx950 only.
//
// Replaces the arithmetic behind `embeddings.search(query, limit)` (reference call site
// inference_pipeline/db_utils/heavy_ranker.py:98-101; in the reference that is txtai -> faiss IndexFlatIP:
// exact inner product of the query with every stored row + k largest).
//
// Shape of the work: S[q, n] = sum_j Q[q, j] * X[n, j]   with Q [256, d] (one query tile), X [N, d] row-major.
// X and Q are stored in HBM in the TILED layout of convert.hip: the slice of one 256-row tile and one K-step (64 bytes of
// every row) is a contiguous 16 KiB block, already in the XOR-swizzled image that makes every ds_read_b128 fragment
// read bank-conflict free.  One persistent workgroup (8 waves, one per CU) walks corpus tiles; per K-step the X block
// (HBM, sequential, nt) and the Q block (L2 resident) go global -> LDS by 16-byte LDS-DMA, 1 KiB contiguous per
// wave-instruction, into a 6-stage X ring and a 2-stage Q ring (counted vmcnt, never drained in the loop).
// Wave (wm, wn) owns rows [128 wm, +128) x queries [64 wn, +64) as 8 x 4 tiles of v_mfma_f32_16x16x32_f16
// (A = corpus rows, B = queries: a lane's 32 accumulators of one column group all belong to ONE query).
// fp16 / fp32: one barrier per K-step, the two wave groups (partners on each SIMD) run memory-first / matrix-first.
// fp8 (block-scaled MFMA over K-step pairs): the groups split the queries and run one slot apart (stagger_loop).
// After the K loop the scores never leave registers: each lane compares its accumulators with the query's current
// threshold (k-th best score so far, in LDS); survivors are appended to the query's LDS candidate list with one LDS
// atomic; lists are compacted (rank-by-counting inside one wave) when they fill.  No B x N score matrix exists.
//
// Two instantiations:
//   MODE 0 "seed": a few tiles per workgroup; writes, per query and tile, two sub-maxima (valid lower bounds: each is
//                  the score of a distinct row); K2 turns their k-th largest into starting thresholds.
//   MODE 1 "main": all tiles, thresholds seeded; flushes per-workgroup sorted top-k lists for K2 to merge.
//   (STAGE 1 is the same MODE 1 code under a second symbol: the first-stage launch of a two-stage search, capi.hip, so that
//   a kernel trace keeps it apart from the main launch.)
#include <type_traits>
#include <utility>

#include "vqa_common.h"

#ifndef VQA_ABLATE
// dev-only timing ablations (scripts/scan_ablation.sh), bit mask; results are wrong when != 0:
// 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMA, 8 no epilogue, 16 s_sleep in place of the MFMAs,
// 32 no Q pieces, 64 no loop barrier.  Since round 6 the int8 / fp16 slot loop honours every bit (round 5's table was taken on a
// build whose slot loop ignored them); a mask without bit 8 lets garbage accumulators flood the candidate regions: pair 2 / 4 / 32 / 64
// with 8 (profiles/r06_scan_ablation.txt)
#define VQA_ABLATE 0
#endif
#ifndef VQA_XNT
#define VQA_XNT 1  // the index stream (read once per search) is loaded with the nt policy: measured 1.7 % faster at 10M rows
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int frag_t __attribute__((ext_vector_type(4)));  // 16 bytes of one row: 8 fp16 | 16 fp8 | 4 fp32
typedef long long i64x2 __attribute__((ext_vector_type(2)));

// One K-step (64 bytes of every row) of one 16 x 16 accumulator tile.  Which k index sits in which byte is irrelevant as
// long as rows and queries are read the same way, so a lane's 16-byte fragment feeds sub-step t of every MFMA chain:
//   fp16: 1 x v_mfma_f32_16x16x32_f16 (8 halves);  fp32: 4 x v_mfma_f32_16x16x4_f32 (1 float each);  T = sub-step.
//   (fp8 pairs the fragments of two K-steps for the block-scaled MFMA, see mma_block_mx.)
template <int DT>
struct MfmaTraits;
template <>
struct MfmaTraits<VQA_F16> {
    static constexpr int kSub = 1;
    template <int T>
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <>
struct MfmaTraits<VQA_F32> {
    static constexpr int kSub = 4;
    template <int T>
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        // cast the whole fragment first: __builtin_bit_cast of a single vector ELEMENT (a[T]) reads element 0 for every T
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(af[T], bf[T], c, 0, 0, 0);
    }
};

// int8 sketch (MODE 2): v_mfma_i32_16x16x64_i8, 64 one-byte elements per 64-byte K-step at the cycles of the fp16 form, i.e.
// twice its rate per row; the int32 accumulators live in the same registers (bit casts only)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <>
struct MfmaTraits<VQA_I8_SKETCH> {
    static constexpr int kSub = 1;
    template <int T>
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b),
                                                                               __builtin_bit_cast(i32x4, c), 0, 0, 0));
    }
};

// fp8 at twice the fp16 MFMA rate: v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 127 = 2^0, byte 0 of
// the scale register, op_sel 0).  One instruction consumes 32 bytes per lane and operand = the lane's fragments of TWO
// consecutive K-steps; rows and queries are paired byte for byte (scripts/probes/mx_fp8_probe.hip: exact on integer
// data, 2.18x the plain fp8 MFMA rate).
typedef int v8i __attribute__((ext_vector_type(8)));
template <int HALF>  // write a 16-byte fragment into the low / high half of a 32-byte MFMA operand (no copy: sub-registers)
__device__ __forceinline__ void set_half(v8i& v, frag_t f) {
    v[4 * HALF + 0] = (int)f[0];
    v[4 * HALF + 1] = (int)f[1];
    v[4 * HALF + 2] = (int)f[2];
    v[4 * HALF + 3] = (int)f[3];
}
template <int M0, int M1>
__device__ __forceinline__ void mma_block_mx(f32x4 (&acc)[8][4], const v8i (&a)[8], const v8i (&b)[4]) {
    constexpr int kUnitScale = 0x7F7F7F7F;
#pragma unroll
    for (int mi = M0; mi < M1; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[mi], b[ni], acc[mi][ni], 0, 0, 0, kUnitScale, 0,
                                                                           kUnitScale);
}

// fp16 on the same K-step-pair operands (the stagger loop, LOOP = 1): the low / high 16 bytes of a lane's 32-byte operand are
// its fragments of the even / odd K-step; 2 x 32 v_mfma_f32_16x16x32_f16, K-step outermost (consecutive MFMAs hit different
// accumulators)
typedef int v4i __attribute__((ext_vector_type(4)));
template <int DT>
__device__ __forceinline__ void mma_pair(f32x4 (&acc)[8][4], const v8i (&a)[8], const v8i (&b)[4]) {
    if constexpr (DT == VQA_FP8_E4M3) {
        mma_block_mx<0, 8>(acc, a, b);
    } else {
        static_assert(DT == VQA_F16, "K-step pairs: fp8 (block-scaled MFMA) or fp16");
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    __builtin_bit_cast(half8, __builtin_shufflevector(a[mi], a[mi], 0, 1, 2, 3)),
                    __builtin_bit_cast(half8, __builtin_shufflevector(b[ni], b[ni], 0, 1, 2, 3)), acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    __builtin_bit_cast(half8, __builtin_shufflevector(a[mi], a[mi], 4, 5, 6, 7)),
                    __builtin_bit_cast(half8, __builtin_shufflevector(b[ni], b[ni], 4, 5, 6, 7)), acc[mi][ni], 0, 0, 0);
    }
}

// row groups [M0, M1) x all four query groups; sub-step outermost so consecutive MFMAs hit different accumulators
// (v_mfma_f32_16x16x4_f32 has a 40-cycle dependent latency against a 32-cycle issue)
template <int DT, int T, int M0, int M1>
__device__ __forceinline__ void mma_sub(f32x4 (&acc)[8][4], const frag_t (&a)[8], const frag_t (&b)[4]) {
#pragma unroll
    for (int mi = M0; mi < M1; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = MfmaTraits<DT>::template mma<T>(a[mi], b[ni], acc[mi][ni]);
}
template <int DT, int M0, int M1>
__device__ __forceinline__ void mma_block(f32x4 (&acc)[8][4], const frag_t (&a)[8], const frag_t (&b)[4]) {
    mma_sub<DT, 0, M0, M1>(acc, a, b);
    if constexpr (MfmaTraits<DT>::kSub > 1) mma_sub<DT, 1, M0, M1>(acc, a, b);
    if constexpr (MfmaTraits<DT>::kSub > 2) {
        mma_sub<DT, 2, M0, M1>(acc, a, b);
        mma_sub<DT, 3, M0, M1>(acc, a, b);
    }
}

// fp32 (exact f32 MFMA at 1/16 of the fp16 rate: the one scan that is bound by its MFMAs) with a partly filled query tile: only the
// wave's first `ng` query column groups are multiplied (wave-uniform; the skipped accumulators stay 0 and their queries' thresholds
// are +inf).  One question against a small fp32 index (the reference's call on BASELINE configs[0]'s shape) then costs 1/16 of a
// full tile's matrix time.  Query group outermost, sub-step and row group inside: consecutive MFMAs still hit different accumulators.
template <int DT>
__device__ __forceinline__ void mma_block_groups(f32x4 (&acc)[8][4], const frag_t (&a)[8], const frag_t (&b)[4], int ng) {
    static_assert(MfmaTraits<DT>::kSub == 4, "the f32 form");
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        if (ni < ng) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = MfmaTraits<DT>::template mma<0>(a[mi], b[ni], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = MfmaTraits<DT>::template mma<1>(a[mi], b[ni], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = MfmaTraits<DT>::template mma<2>(a[mi], b[ni], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = MfmaTraits<DT>::template mma<3>(a[mi], b[ni], acc[mi][ni]);
        }
    }
}

typedef __attribute__((address_space(3))) char* lds_char_ptr;

constexpr int kThreads = 512;           // 8 waves: 2 (row halves = ping-pong groups) x 4 (query quarters)
constexpr int kTileRows = 256;          // corpus rows per tile
constexpr int kQ = VQA_QUERY_TILE;      // 256 queries per tile
constexpr int kRowBytes = 64;           // bytes of every row per K-step (32 fp16 / 64 fp8 / 16 fp32 elements)
constexpr int kOperandBytes = 256 * kRowBytes;  // 16 KiB: one operand slice (256 rows x one K-step)
#ifndef VQA_SX
#define VQA_SX 6
#endif
constexpr int kSx = VQA_SX, kPx = VQA_SX - 1;  // X ring: stages, K-steps issued ahead (HBM stream: deep); one stage idles per K-step
constexpr int kSq = 2, kPq = 2;  // Q ring: stages, K-steps issued ahead (L2 resident: shallow)
constexpr int kXRingBytes = kSx * kOperandBytes;
constexpr int kQRingBytes = kSq * kOperandBytes;
constexpr int kPipeBytes = kXRingBytes + kQRingBytes;  // 128 KiB
constexpr int kLdsTotal = 160 * 1024;
constexpr int kCap = (kLdsTotal - kPipeBytes - 2 * kQ * 4) / (kQ * 8);  // 15 candidate slots per query
constexpr int kExt = kOperandBytes / (kQ * 8);                           // 8 more in the idle X stage during an epilogue
constexpr int kMaxK = kCap - 3;                                         // 12
constexpr int kSeedsPerTile = 2;  // seed pass: sub-maxima kept per query and tile (one per row half; 8, one per 32 rows, for tiny shards)
static_assert(kPipeBytes <= 144 * 1024, "ring sizes");

// MODE 2 (sketch scan): no candidate lists in LDS, so the X ring can take a sixth stage (LOOP 0, VQA_SKETCH_SX=6 at index create).
// Measured (round 4, interleaved A/B at 10M x 768): 1.4356 vs 1.4317 ms -- nothing: the scan is not waiting for the stream, its MFMA
// pipe sits at the rate the chip's power management allows.  The five-stage ring (LOOP 1) stays the default;
// its spare LDS holds the tile maxima of up to kSketchMaxTiles tiles per workgroup (80M-row shards).
constexpr int kSketchPipe6 = (6 + 3) * kOperandBytes;
constexpr int kSketchBetaBytes = kTileRows * 4;  // LOOP 2: the per-row betas of the tile being scanned
constexpr int kSketchMaxTiles = (kLdsTotal - kPipeBytes - kSketchQRows * kQ * 4 - 16 - kSketchBetaBytes) / 16;  // tiles per workgroup of a sketch scan (LDS: 16 B each)
constexpr int kSketchMaxTiles6 = (kLdsTotal - kSketchPipe6 - kSketchQRows * kQ * 4 - 16 - kSketchBetaBytes) / 16;
constexpr int kOverBit = 1 << 30;  // "an append was refused" flag, kept in bit 30 of cnt[0] (LDS is fully used)
constexpr int kCntMask = 0xFFFFFF;

struct Lists {  // views into LDS: per query {threshold, count, kCap keys} = 128 B, 32 KiB in all
    float* thr;     // [256] current k-th best score per query (+inf for padded queries)
    int* cnt;       // [256] appends attempted since the last compaction (may exceed the capacity when refused)
    vqa_key* cand;  // [256][kCap]
    vqa_key* ext;   // [256][kExt] spill area in the idle ring stage; only valid inside one tile's epilogue
};

__device__ __forceinline__ int list_count(const Lists& L, int q) { return L.cnt[q] & kCntMask; }

__device__ __forceinline__ vqa_key list_get(const Lists& L, int q, int i) {
    return i < kCap ? L.cand[q * kCap + i] : L.ext[q * kExt + (i - kCap)];
}

// ---- rank-by-counting compaction of one query's list by one wave ------------------------------------------
// Keeps the k best of the stored keys, sorted best first, and raises the threshold.
__device__ __forceinline__ void compact_query(const Lists& L, int q, int k, int lane, int capacity) {
    const int raw = L.cnt[q];
    int n = raw & kCntMask;
    n = n < capacity ? n : capacity;
    const vqa_key mine = lane < n ? list_get(L, q, lane) : 0ull;
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (list_get(L, q, j) > mine) ? 1 : 0;  // broadcast reads; keys are distinct
    if (lane < n && rank < k) L.cand[q * kCap + rank] = mine;
    if (lane < n && rank == k - 1) L.thr[q] = vqa_key_score(mine);
    if (lane == 0) L.cnt[q] = (raw & kOverBit) | (n < k ? n : k);
}

// wave w compacts the queries [32 w, 32 w + 32) whose list length reached `water`
__device__ __forceinline__ void compact_pass(const Lists& L, int wave, int lane, int k, int water, int capacity) {
    const int q0 = wave * 32;
    const int c = lane < 32 ? (L.cnt[q0 + lane] & kCntMask) : 0;
    unsigned long long need = __ballot(c >= water) & 0xFFFFFFFFull;
    while (need) {
        const int b = __builtin_ctzll(need);
        need &= need - 1;
        compact_query(L, q0 + b, k, lane, capacity);
    }
}

// Load one key from global memory, complete when the call returns.  Inline asm on purpose: a load hipcc can see leaves a
// "possibly pending" destination register on the paths that skip its use, and the waitcnt pass then puts an
// unconditional s_waitcnt vmcnt(0) where the K loop next writes that register -- which drained the loader waves' whole
// LDS-DMA queue every second K-step (found in the disassembly; the ring was 5 K-steps deep on paper only).
// (uniform base in SGPRs + a 32-bit lane offset: nothing 64-bit per lane stays live across the K loop)
__device__ __forceinline__ vqa_key load_key_now(const vqa_key* base, int idx) {
    vqa_key v;
    asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"((uint32_t)idx * 8u), "s"(base) : "memory");
    return v;
}

__device__ __forceinline__ bool append_candidate(const Lists& L, int q, float v, uint32_t pos, int spill) {
    const int slot = atomicAdd(&L.cnt[q], 1) & kCntMask;  // ds_add_rtn_u32
    if (slot < kCap) {
        L.cand[q * kCap + slot] = vqa_make_key(v, pos);
        return true;
    }
    if (slot < kCap + spill) {
        L.ext[q * kExt + (slot - kCap)] = vqa_make_key(v, pos);
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
// `upper` (or null): exclusive upper bound key per query -- the continuation of a search beyond kMaxK results only admits
// candidates strictly below the last key already returned.
__device__ __forceinline__ bool process_pending(const Lists& L, const f32x4 (&acc)[8][4], uint32_t (&pend)[4], int wm,
                                                int wn, int c, int g, uint32_t row0, const vqa_key* __restrict__ upper, int spill) {
    bool refused = false;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        uint32_t bits = pend[ni];
        if (bits) {
            const int q = wn * 64 + ni * 16 + c;
            const float th = L.thr[q];
            const vqa_key up = upper ? load_key_now(upper, q) : ~0ull;
            uint32_t keep = 0;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const float v = select_acc(acc, ni, b);
                if (v >= th) {
                    const uint32_t pos = row0 + (uint32_t)(wm * 128 + (b >> 2) * 16 + g * 4 + (b & 3));
                    if (vqa_make_key(v, pos) < up && !append_candidate(L, q, v, pos, spill)) keep |= 1u << b;
                }
            }
            pend[ni] = keep;
            refused |= keep != 0;
        }
    }
    return refused;
}

// ---- LDS-DMA: 4 wave-instructions, each 64 lanes x 16 B -> 1 KiB of LDS.  Global address = sbase + voff + i * 1024,
// LDS address = lds_dst + i * 1024 + lane * 16 (M0 base + instruction offset + lane * 16, added by hardware): the
// instruction offset moves both sides, so one M0 write serves the four.  Inline asm so that hipcc keeps no scoreboard
// entry for the loads: all ordering is by the counted vmcnt waits below.  sbase / lds_dst are SALU-computed.
template <bool NT>
__device__ __forceinline__ void glds16x4(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    if constexpr (NT) {
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
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
    }
}

// one wave-instruction of LDS-DMA: 64 lanes x 16 B = 1 KiB at sbase + voff -> LDS lds_dst (the per-row betas of one tile, LOOP 2)
__device__ __forceinline__ void glds16x1(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void block_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef VQA_STAMPS
// dev-only diagnostic build (scripts/stamp_timeline.py): s_memtime stamps of one workgroup's K-steps during one tile, written
// by scalar stores to a buffer no other code reads.  Stamp values arrive asynchronously (SMEM): they are only stored after
// an lgkmcnt(0).  The instrumentation itself costs ~10 % (measured), so it is never part of the product build.
constexpr int kStampWg = 5, kStampTile = 40, kStampSlots = 8;
__device__ unsigned long long g_stamps[8 * 64 * kStampSlots];
#define VQA_STAMP(J)                                                              \
    do {                                                                          \
        VQA_SB();                                                                 \
        if (stamp_on) asm volatile("s_memtime %0" : "=s"(stamp_t[J]));            \
        VQA_SB();                                                                 \
    } while (0)
#define VQA_STAMP_FLUSH(KT)                                                                                            \
    do {                                                                                                               \
        if (stamp_on) {                                                                                                \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            unsigned long long* sp_ = g_stamps + ((size_t)wave * 64 + (KT)) * kStampSlots;                             \
            asm volatile(                                                                                              \
                "s_store_dwordx2 %1, %0, 0x0\n\ts_store_dwordx2 %2, %0, 0x8\n\ts_store_dwordx2 %3, %0, 0x10\n\t"        \
                "s_store_dwordx2 %4, %0, 0x18\n\ts_store_dwordx2 %5, %0, 0x20\n\ts_store_dwordx2 %6, %0, 0x28\n\t"      \
                "s_store_dwordx2 %7, %0, 0x30\n\ts_store_dwordx2 %8, %0, 0x38"                                           \
                :: "s"(sp_), "s"(stamp_t[0]), "s"(stamp_t[1]), "s"(stamp_t[2]), "s"(stamp_t[3]), "s"(stamp_t[4]),       \
                   "s"(stamp_t[5]), "s"(stamp_t[6]), "s"(stamp_t[7]) : "memory");                                       \
        }                                                                                                              \
    } while (0)
#else
#define VQA_STAMP(J) (void)0
#define VQA_STAMP_FLUSH(KT) (void)0
#endif

template <int MODE, int DT, int STAGE = 0, int LOOP = 0>
__global__ __launch_bounds__(kThreads) void score_topk_kernel(const void* __restrict__ X, const void* __restrict__ Qs,
                                                              const float* __restrict__ thr_init,
                                                              const vqa_key* __restrict__ upper,
                                                              vqa_key* __restrict__ out, long long N, int KT, int nq, int k,
                                                              int tile_begin, int tile_end, const int* __restrict__ gate,
                                                              int row_lists, int list_offset, int seeds, SketchScanArgs sk) {
    // gated launch (fallback passes of a large-k search, capi.hip): nothing to do when the one-pass result was verified
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lists L;
    L.thr = reinterpret_cast<float*>(smem + kPipeBytes);
    L.cnt = reinterpret_cast<int*>(smem + kPipeBytes + kQ * 4);
    L.cand = reinterpret_cast<vqa_key*>(smem + kPipeBytes + 2 * kQ * 4);
    L.ext = nullptr;
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_char_ptr)smem;  // LDS byte address of the dynamic region

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fp16 / fp32: group = corpus-row half (the two groups run the K-step memory-first / matrix-first).
    // fp8: group = QUERY half (the two groups run one slot apart and own disjoint candidate lists, see stagger_loop).
    constexpr bool kStagger = DT == VQA_FP8_E4M3 || (DT == VQA_F16 && LOOP == 1);  // LOOP 1: fp16 on the K-step-pair stagger loop
    const int grp = wave >> 2;                              // waves w and w + 4 share a SIMD: one of each group
    const int wm = kStagger ? (wave >> 1) & 1 : grp;        // corpus-row half
    const int wn = kStagger ? 2 * grp + (wave & 1) : wave & 3;  // query quarter
    const int c = lane & 15;
    const int g = lane >> 4;

    if (MODE == 1 && tid < kQ) {
        L.thr[tid] = tid < nq ? (thr_init ? thr_init[tid] : -INFINITY) : INFINITY;
        L.cnt[tid] = 0;
    }
    // MODE 2 (sketch scan): per query theta, ||q_lo||, ||q||, 1 / (s_q s_x) in the list area (4 KiB) + one append counter
    constexpr int kSxSlot = (MODE == 2 && LOOP == 0) ? 6 : 5;                 // X ring stages of the slot loop (LOOP 2: five + the per-row betas)
    constexpr int kSkPipe = (kSxSlot + 3) * kOperandBytes;                    // MODE 2: where the rings end
    float* const sk_q = reinterpret_cast<float*>(smem + kSkPipe);          // [kSketchQRows][256]
    int* const sk_cnt = reinterpret_cast<int*>(smem + kSkPipe + kSketchQRows * kQ * 4);
    if (MODE == 2) {
        if (tid < kQ) {
            // queries past the batch never produce a candidate: theta = +inf and benign factors (the rows of qconst past nq are
            // whatever the device memory held -- the cascade's merges write the constants of the batch's queries only; a negative
            // garbage 1 / s_q would turn +inf into -inf and flood every region with the zero rows of the padded queries)
            const bool live = tid < nq;
            sk_q[tid] = live ? sk.qconst[tid] : INFINITY;
            sk_q[kQ + tid] = live ? sk.qconst[kQ + tid] : 0.f;
            sk_q[2 * kQ + tid] = live ? sk.qconst[2 * kQ + tid] : 0.f;
            sk_q[3 * kQ + tid] = live ? sk.qconst[3 * kQ + tid] : 1.f;
            sk_q[4 * kQ + tid] = live ? sk.qconst[4 * kQ + tid] : 0.f;
            sk_q[5 * kQ + tid] = live ? sk.qconst[5 * kQ + tid] : 0.f;
        }
        if (tid == 0) *sk_cnt = 0;
    }

    const int first_tile = tile_begin + blockIdx.x;
    const int ntile = first_tile < tile_end ? (tile_end - first_tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int total = ntile * KT;  // K-steps of this workgroup, numbered kappa = ti * KT + kt
    // MODE 2: (max ||x_hi||, max ||x_lo||, 1 / scale) of every tile this workgroup scans, in LDS (the launcher bounds ntile by kSketchMaxTiles)
    constexpr bool kBeta = MODE == 2 && LOOP == 2;  // per-row form: alpha beta added per (query, row) in the tile epilogue
    float* const sk_beta = reinterpret_cast<float*>(smem + kSkPipe + kSketchQRows * kQ * 4 + 16);  // [256] betas of the current tile
    float4* const sk_tm = reinterpret_cast<float4*>(smem + kSkPipe + kSketchQRows * kQ * 4 + 16 + kSketchBetaBytes);
    if (MODE == 2)
        for (int t = tid; t < ntile; t += kThreads) {  // .w: max |w . x_lo| of the tile (the split slack term) in place of the scale, which the scan does not use
            float4 ti = sk.tile_info[first_tile + t * (int)gridDim.x];
            ti.w = sk.tile_c ? sk.tile_c[first_tile + t * (int)gridDim.x] : 0.f;
            sk_tm[t] = ti;
        }

    // ---- LDS-DMA.  X and Q are stored in the TILED layout (convert.hip): the slice of one tile and K-step is a
    // contiguous 16 KiB block that already is the swizzled LDS image.  vmcnt retires in issue order per wave, so the
    // deep HBM stream and the shallow L2 stream are issued by DIFFERENT waves: group 1 (waves 4-7) loads X, kPx
    // K-steps ahead into a kSx-stage ring; group 0 (waves 0-3) loads Q, kPq ahead into a kSq-stage ring.  Each wave
    // moves a quarter of a slice per K-step: 4 wave-instructions reading 1 KiB contiguous each.
    const int lw = wave & 3;
    const uint32_t voff = (uint32_t)(lw * 4096 + lane * 16);
    const uint32_t xring_lds = smem_lds, qring_lds = smem_lds + kXRingBytes;
    // issue cursor of this wave: K-steps issued, K-step inside the tile, source block pointer, LDS destination
    int ik = 0, i_kt = 0;
    const size_t tile_jump = ((size_t)gridDim.x - 1) * KT * kOperandBytes;  // from a tile's last block to the next tile's
    const char* x_src = reinterpret_cast<const char*>(X) + (size_t)first_tile * KT * kOperandBytes;
    const char* q_src = reinterpret_cast<const char*>(Qs);
    uint32_t i_dst = (grp ? xring_lds : qring_lds) + lw * 4096;
    auto issue_x = [&]() {
        if (ik >= total) return;
        glds16x4<VQA_XNT != 0>(x_src, voff, i_dst);
        ++ik;
        x_src += kOperandBytes;
        if (++i_kt == KT) {
            i_kt = 0;
            x_src += tile_jump;
        }
        i_dst += kOperandBytes;
        if (i_dst >= xring_lds + kXRingBytes) i_dst -= kXRingBytes;
    };
    auto issue_q = [&]() {
        if (ik >= total) return;
        glds16x4<false>(q_src, voff, i_dst);
        ++ik;
        q_src += kOperandBytes;
        if (++i_kt == KT) {
            i_kt = 0;
            q_src = reinterpret_cast<const char*>(Qs);
        }
        i_dst += kOperandBytes;
        if (i_dst >= qring_lds + kQRingBytes) i_dst -= kQRingBytes;
    };
    // wait until this wave's pieces of K-step kappa_needed have landed; it has then issued through
    // kappa_needed + extra (4 instructions per K-step), fewer near the end of the stream
    auto wait_pieces = [&](int kappa_needed) {
        if (grp) {
            if (kappa_needed + (kPx - 2) < total) wait_vmcnt<4 * (kPx - 2)>();
            else wait_vmcnt<0>();
        } else {
            if (kappa_needed + (kPq - 2) < total) wait_vmcnt<4 * (kPq - 2)>();
            else wait_vmcnt<0>();
        }
    };

    // ---- fragment geometry: lane reads 16 B = 8 k-contiguous halves of row (base + c), k offset 8 g
    const int frag_off = c * kRowBytes + ((g ^ (((c >> 3) & 1) * 3)) << 4);
    [[maybe_unused]] const int a_base = wm * 128 * kRowBytes + frag_off;                // inside an X stage
    [[maybe_unused]] const int b_base = kXRingBytes + wn * 64 * kRowBytes + frag_off;  // inside a Q stage

    int sx = 0, sq = 0;  // ring stages of the K-step whose fragments are read next
// fragment reads of the next K-step into register set (A, B); static register names, no references (SROA-friendly)
#if VQA_ABLATE & 2
#define VQA_READ_FRAGS(A, B)                                                                  \
    do {                                                                                      \
        for (int i_ = 0; i_ < 8; ++i_) { A[i_] = frag_t{}; asm volatile("" : "+v"(A[i_])); }  \
        for (int i_ = 0; i_ < 4; ++i_) { B[i_] = frag_t{}; asm volatile("" : "+v"(B[i_])); }  \
        if (++sx == kSx) sx = 0;                                                              \
        if (++sq == kSq) sq = 0;                                                              \
    } while (0)
#else
#define VQA_READ_FRAGS(A, B)                                                                                      \
    do {                                                                                                          \
        const char* xbuf_ = smem + sx * kOperandBytes;                                                            \
        const char* qbuf_ = smem + sq * kOperandBytes;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
            B[i_] = *reinterpret_cast<const frag_t*>(qbuf_ + b_base + i_ * 16 * kRowBytes);                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
            A[i_] = *reinterpret_cast<const frag_t*>(xbuf_ + a_base + i_ * 16 * kRowBytes);                       \
        if (++sx == kSx) sx = 0;                                                                                  \
        if (++sq == kSq) sq = 0;                                                                                  \
    } while (0)
#endif
// fp8: fragments of an even / odd K-step into the low / high halves of the 32-byte MFMA operands
#define VQA_READ_HALVES(A, B, HALF)                                                                               \
    do {                                                                                                          \
        const char* xbuf_ = smem + sx * kOperandBytes;                                                            \
        const char* qbuf_ = smem + sq * kOperandBytes;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
            set_half<HALF>(B[i_], *reinterpret_cast<const frag_t*>(qbuf_ + b_base + i_ * 16 * kRowBytes));        \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
            set_half<HALF>(A[i_], *reinterpret_cast<const frag_t*>(xbuf_ + a_base + i_ * 16 * kRowBytes));        \
        if (++sx == kSx) sx = 0;                                                                                  \
        if (++sq == kSq) sq = 0;                                                                                  \
    } while (0)
#if VQA_ABLATE & 4
#define VQA_MMA_RANGE(A, B, M0, M1)                                                                       \
    do {                                                                                                  \
        _Pragma("unroll") for (int mi_ = (M0); mi_ < (M1); ++mi_)                                         \
            _Pragma("unroll") for (int ni_ = 0; ni_ < 4; ++ni_)                                           \
                asm volatile("" : "+v"(acc[mi_][ni_]) : "v"(A[mi_]), "v"(B[ni_]));                        \
    } while (0)
#else
#define VQA_MMA_RANGE(A, B, M0, M1) mma_block<DT, (M0), (M1)>(acc, A, B)
#endif
#define VQA_MMA(A, B) VQA_MMA_RANGE(A, B, 0, 8)
#ifndef VQA_SPLIT
#define VQA_SPLIT 4  // matrix-first group: row groups multiplied before its memory instructions are issued (of 8); 3 / 5 / 8
#endif               // measured 0.6 / 1.3 / 7 % slower on one device
#if VQA_ABLATE & 1
#define VQA_ISSUE() (void)0
#elif VQA_ABLATE & 32
#define VQA_ISSUE() do { if (grp) issue_x(); } while (0)
#else
#define VQA_ISSUE() do { if (grp) issue_x(); else issue_q(); } while (0)
#endif
#if VQA_ABLATE & 64
#define VQA_LOOP_BARRIER() (void)0
#else
#define VQA_LOOP_BARRIER() block_barrier()
#endif
#ifdef VQA_NO_SB
#define VQA_SB() (void)0
#else
#define VQA_SB() __builtin_amdgcn_sched_barrier(0)
#endif
// One K-step: fragments in (CA, CB) are multiplied; the next K-step's go to (NA, NB) when PREFETCH.
#define VQA_KSTEP(CA, CB, NA, NB, KAPPA, PREFETCH)                 \
    do {                                                           \
        VQA_STAMP(0);                                              \
        if (!kMemFirst) {                                          \
            VQA_MMA_RANGE(CA, CB, 0, VQA_SPLIT);                   \
            VQA_SB();                                              \
            VQA_STAMP(1);                                          \
            if (PREFETCH) VQA_READ_FRAGS(NA, NB);                  \
            VQA_STAMP(2);                                          \
            VQA_ISSUE();                                           \
            VQA_SB();                                              \
            VQA_STAMP(3);                                          \
            VQA_MMA_RANGE(CA, CB, VQA_SPLIT, 8);                   \
        } else {                                                   \
            VQA_STAMP(1);                                          \
            if (PREFETCH) VQA_READ_FRAGS(NA, NB);                  \
            VQA_STAMP(2);                                          \
            VQA_ISSUE();                                           \
            VQA_SB();                                              \
            VQA_STAMP(3);                                          \
            VQA_MMA(CA, CB);                                       \
        }                                                          \
        VQA_STAMP(4);                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        VQA_STAMP(5);                                              \
        wait_pieces((KAPPA) + 2);                                  \
        VQA_STAMP(6);                                              \
        VQA_LOOP_BARRIER();                                        \
        VQA_STAMP(7);                                              \
        VQA_STAMP_FLUSH((KAPPA) - ti * KT);                        \
    } while (0)

    // ---- epilogue pieces (shared by both loop forms); scores stay in registers -----------------------------------
    // acc[mi][ni][j] = score(row = row0 + 128 wm + 16 mi + 4 g + j, query = 64 wn + 16 ni + c)
    auto mask_ragged = [&](f32x4 (&acc)[8][4], uint32_t row0) __attribute__((always_inline)) {
        const long long left = N - (long long)row0;
        const int rows_left = left < kTileRows ? (int)left : kTileRows;
        if (rows_left < kTileRows) {  // ragged last tile of the shard: rows past the end never compete (NaN fails every
                                      // >= / > test below and is ignored by fmaxf, even against a -inf threshold)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool dead = wm * 128 + mi * 16 + g * 4 + j >= rows_left;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni][j] = dead ? __builtin_nanf("") : acc[mi][ni][j];
                }
        }
    };
    // MODE 0: two sub-maxima per query and tile (one per row half wm) -> seeds [query][seed tile][2].  Each is the score
    // of a distinct row, so the k-th largest of any k of them bounds the k-th best score from below.
    auto seed_epilogue = [&](const f32x4 (&acc)[8][4], uint32_t row0, int ti) __attribute__((always_inline)) {
        const int seed_tiles = tile_end - tile_begin;
        const int tile_local = first_tile - tile_begin + ti * (int)gridDim.x;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int q = wn * 64 + ni * 16 + c;
            const vqa_key up = upper ? load_key_now(upper, q) : ~0ull;
            float m = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = wm * 128 + mi * 16 + g * 4 + j;
                    const float v = acc[mi][ni][j];
                    if (v > m && (!upper || vqa_make_key(v, row0 + (uint32_t)r) < up)) {
                        m = v;
                        arg = r;
                    }
                }
            vqa_key key = (m > -INFINITY && q < nq) ? vqa_make_key(m, row0 + (uint32_t)arg) : 0ull;
            if (seeds == 8) {  // one seed per 32-row group: this lane's own maximum
                out[((size_t)q * seed_tiles + tile_local) * 8 + wm * 4 + g] = key;
                continue;
            }
            // max over the four lane groups g that hold the other rows of this query (lanes c, c + 16, c + 32, c + 48)
            vqa_key o = __shfl_xor(key, 16, 64);
            key = o > key ? o : key;
            o = __shfl_xor(key, 32, 64);
            key = o > key ? o : key;
            if (g == 0) out[((size_t)q * seed_tiles + tile_local) * kSeedsPerTile + wm] = key;
        }
    };
    // MODE 1: every accumulator that beats its query's threshold is appended to the query's candidate list.  Returns true
    // when some append was refused (list full); the refused accumulators are left in pend[] for process_pending.
    auto append_epilogue = [&](const f32x4 (&acc)[8][4], uint32_t row0, uint32_t (&pend)[4], int spill)
                               __attribute__((always_inline)) -> bool {
        // Common path, straight-line: the four thresholds are read first, the four maxima are reduced as independent
        // v_max3 trees (depth 4 instead of a 16-deep chain each), and ONE branch covers all four query columns -- a wave
        // whose 64 x 4 maxima all stay below their thresholds (the usual case) takes no divergent path at all.
        float th[4], m[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            pend[ni] = 0u;
            th[ni] = L.thr[wn * 64 + ni * 16 + c];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float r[8];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)  // two v_max3_f32 per accumulator vector
                r[mi] = fmaxf(fmaxf(fmaxf(acc[mi][ni][0], acc[mi][ni][1]), acc[mi][ni][2]), acc[mi][ni][3]);
            const float r01 = fmaxf(fmaxf(r[0], r[1]), r[2]), r23 = fmaxf(fmaxf(r[3], r[4]), r[5]);
            m[ni] = fmaxf(fmaxf(r01, r23), fmaxf(r[6], r[7]));
#if VQA_ABLATE & 8
            asm volatile("" : "+v"(m[ni]));
            m[ni] = -INFINITY;
#endif
        }
        bool refused = false;
        if ((m[0] >= th[0]) | (m[1] >= th[1]) | (m[2] >= th[2]) | (m[3] >= th[3])) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if (m[ni] >= th[ni]) {
                    // rare: some accumulator of this (lane, query) beats the threshold.  Two static levels (row group, then
                    // element) instead of a 32-way select: ~50 instructions for the usual single survivor.
                    const int q = wn * 64 + ni * 16 + c;
                    const vqa_key up = upper ? load_key_now(upper, q) : ~0ull;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        const float mm = fmaxf(fmaxf(acc[mi][ni][0], acc[mi][ni][1]), fmaxf(acc[mi][ni][2], acc[mi][ni][3]));
                        if (mm >= th[ni]) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v = acc[mi][ni][j];
                                if (v >= th[ni]) {
                                    const uint32_t pos = row0 + (uint32_t)(wm * 128 + mi * 16 + g * 4 + j);
                                    if (vqa_make_key(v, pos) < up && !append_candidate(L, q, v, pos, spill)) {
                                        pend[ni] |= 1u << (mi * 4 + j);
                                        refused = true;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        return refused;
    };

    // MODE 2: the accumulators are the int32 dot products D of the int8 sketches.  The exact score of (query q, row) obeys
    //     s <= s_q s_t D + ||q_lo|| max_tile ||x_hi|| + ||q|| max_tile ||x_lo||                (Cauchy-Schwarz on the two residues)
    // (s_t: the tile's scale) so a row can only belong to the top-k if D >= T_q = (theta_q - slack_q(tile)) / (s_q s_t): the test is the MODE 1 test on
    // integer maxima, with a per-tile, per-query threshold; survivors go to this workgroup's region of candidate pairs as
    // (query << 32 | row position) and are scored exactly afterwards (rescore_kernel, merge_topk.hip).  No list, no
    // compaction, no workgroup barrier.
    auto sketch_epilogue = [&](const f32x4 (&acc)[8][4], uint32_t row0, int ti) __attribute__((always_inline)) {
        if constexpr (bool(VQA_ABLATE & 8)) return;
        const float4 tmax = sk_tm[ti];
        const float a_hi = tmax.x, b_lo = tmax.y, inv_sx = tmax.z, c_w = tmax.w;
        if constexpr (kBeta) {
            // per-row form: the test is D + (alpha / (s_q s_t)) beta_row >= T, element by element (one cvt + one fma each).  One query
            // column at a time and the betas of four rows at a time, so that next to nothing stays live beside the accumulators
            // (the kernel sits at its 256 registers: hipcc spills ~90 of them otherwise).
            const float* brow = sk_beta + wm * 128 + g * 4;
            unsigned long long* region = sk.regions + (size_t)blockIdx.x * sk.cap;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int q = wn * 64 + ni * 16 + c;
                const float num = sk_q[q] - sk_q[kQ + q] * a_hi - sk_q[2 * kQ + q] * b_lo - sk_q[5 * kQ + q] * c_w;  // c_w: the tile's max |beta|
                const float inv = sk_q[3 * kQ + q] * inv_sx;
                const float t = num * inv;
                const float ab = sk_q[4 * kQ + q] * inv;  // alpha / (s_q s_t): D + ab beta is the bound's integer part
                const float Tn = t - (fabsf(t) + fabsf(ab) * c_w) * 8e-6f - 1.0f;  // + the roundings of ab and of the fma below
                float mx = -INFINITY;
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const float4 b4 = *reinterpret_cast<const float4*>(brow + mi * 16);
                    const i32x4 v = __builtin_bit_cast(i32x4, acc[mi][ni]);
                    const float v0 = __builtin_fmaf(ab, b4.x, (float)v[0]), v1 = __builtin_fmaf(ab, b4.y, (float)v[1]);
                    const float v2 = __builtin_fmaf(ab, b4.z, (float)v[2]), v3 = __builtin_fmaf(ab, b4.w, (float)v[3]);
                    mx = fmaxf(mx, fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
                }
                if (mx >= Tn) {
                    const unsigned long long qhi = (unsigned long long)q << 32;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        const i32x4 v = __builtin_bit_cast(i32x4, acc[mi][ni]);
                        const float4 b4 = *reinterpret_cast<const float4*>(brow + mi * 16);
                        const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t pos = row0 + (uint32_t)(wm * 128 + mi * 16 + g * 4 + j);
                            if (__builtin_fmaf(ab, bj[j], (float)v[j]) >= Tn && (long long)pos < N) {
                                const int slot = atomicAdd(sk_cnt, 1);
                                if (slot < sk.cap) region[slot] = qhi | pos;
                            }
                        }
                    }
                }
                VQA_SB();
            }
            return;
        }
        float T[4];
        int mi32[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int q = wn * 64 + ni * 16 + c;
            // (the fp32 summation error of the exact scores against the real-number dot product the bound speaks of rides in the
            // per-query factors: sketch_qconst_kernel; c_w: the tile's max |w . x_lo| against |alpha|, the split slack term)
            const float num = sk_q[q] - sk_q[kQ + q] * a_hi - sk_q[2 * kQ + q] * b_lo - sk_q[4 * kQ + q] * c_w;
            const float t = num * sk_q[3 * kQ + q] * inv_sx;
            T[ni] = t - fabsf(t) * 4e-6f - 0.5f;  // every rounding of this line errs towards MORE candidates; D is an integer
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            int r[8];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const i32x4 v = __builtin_bit_cast(i32x4, acc[mi][ni]);
                r[mi] = max(max(max(v[0], v[1]), v[2]), v[3]);
            }
            mi32[ni] = max(max(max(max(r[0], r[1]), r[2]), max(max(r[3], r[4]), r[5])), max(r[6], r[7]));
        }
        if (((float)mi32[0] >= T[0]) | ((float)mi32[1] >= T[1]) | ((float)mi32[2] >= T[2]) | ((float)mi32[3] >= T[3])) {
            unsigned long long* region = sk.regions + (size_t)blockIdx.x * sk.cap;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if ((float)mi32[ni] >= T[ni]) {
                    const unsigned long long qhi = (unsigned long long)(wn * 64 + ni * 16 + c) << 32;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        const i32x4 v = __builtin_bit_cast(i32x4, acc[mi][ni]);
                        if ((float)max(max(v[0], v[1]), max(v[2], v[3])) >= T[ni]) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t pos = row0 + (uint32_t)(wm * 128 + mi * 16 + g * 4 + j);
                                if ((float)v[j] >= T[ni] && (long long)pos < N) {
                                    const int slot = atomicAdd(sk_cnt, 1);
                                    if (slot < sk.cap) region[slot] = qhi | pos;
                                }
                            }
                        }
                    }
                }
            }
        }
    };

    // ---- tile epilogue shared by the fp16 / fp32 loop forms: threshold test + appends (MODE 1) or sub-maxima (MODE 0).
    // `ext_stage`: an X ring stage that stays idle until every wave has passed the next workgroup barrier
    auto finish_tile = [&](f32x4 (&acc)[8][4], int ti, int ext_stage) __attribute__((always_inline)) {
        const uint32_t row0 = ((uint32_t)first_tile + (uint32_t)ti * gridDim.x) * kTileRows;  // n < 2^32 rows
        if (MODE == 2) {
            sketch_epilogue(acc, row0, ti);
            return;
        }
        mask_ragged(acc, row0);
        if (MODE == 0) {
            seed_epilogue(acc, row0, ti);
            return;
        }
        L.ext = reinterpret_cast<vqa_key*>(smem + ext_stage * kOperandBytes);
        constexpr int kSpill = kExt;
        uint32_t pend[4];  // accumulators whose append was refused (list full): retried below
        // every wave passed the re-align barrier after its last fragment reads completed, so the idle stage (L.ext)
        // is free: a list holds up to kCap + kExt keys inside this epilogue and is back below kCap when it ends
        // (water <= kCap: every list that spilled into L.ext is compacted).
        if (append_epilogue(acc, row0, pend, kSpill)) atomicOr(&L.cnt[0], kOverBit);
        __syncthreads();
        for (;;) {
            const int over = L.cnt[0] & kOverBit;  // stable here: set before the barrier above, cleared only behind the next
            // normal tiles: compact lists that are nearly full; after a refusal: compact everything above k
            compact_pass(L, wave, lane, k, over ? k + 1 : k + (kCap - k + 1) / 2, kCap + kSpill);
            // Fast path: no trailing barrier.  What compaction wrote (thr, cnt, cand) is next read in the next tile's
            // epilogue, 24+ barriers away; the spill stage L.ext is refilled only behind the next tile's first barrier,
            // which no wave passes before every wave has finished compacting.
            if (!over) break;
            __syncthreads();
            if (tid == 0) L.cnt[0] &= ~kOverBit;
            __syncthreads();
            if (process_pending(L, acc, pend, wm, wn, c, g, row0, upper, kSpill)) atomicOr(&L.cnt[0], kOverBit);
            __syncthreads();
        }
    };

    // ---- fp16 / fp32: one barrier per K-step, the two groups memory-first / matrix-first ----------------------------
    // The whole tile loop exists twice (group 0: memory instructions first, group 1: matrix instructions first) and
    // the wave-uniform branch sits OUTSIDE it: a diamond around each K-step makes hipcc spill the accumulators.
    auto tile_loop = [&](auto mem_first_tag) __attribute__((always_inline)) {
    constexpr bool kMemFirst = decltype(mem_first_tag)::value;
    // prologue: the first K-steps of this wave's stream; K-steps 0 and 1 landed
    if (grp) {
        for (int i = 0; i < kPx; ++i) issue_x();
    } else {
        for (int i = 0; i < kPq; ++i) issue_q();
    }
    wait_pieces(1);
    block_barrier();
    frag_t a0[8], b0[4];  // fragments of the K-step about to be multiplied (carried across tiles)
    if (ntile > 0) {
        VQA_READ_FRAGS(a0, b0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    block_barrier();  // every wave holds its first fragments before a piece of K-step + kS may overwrite the stage
    for (int ti = 0; ti < ntile; ++ti) {
        f32x4 acc[8][4];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

        // One barrier per K-step.  Iteration kappa: (R) read the fragments of kappa + 1 into the other register set,
        // (D) issue this wave's LDS-DMA pieces of kappa + kP, (M) 32 MFMAs of kappa; then lgkmcnt(0) (the stage of
        // kappa + 1 may be refilled after the barrier), a counted vmcnt (this wave's pieces of kappa + 2 landed) and
        // the barrier.  Group 0 runs R, D, M and group 1 runs M, R, D, so on every SIMD one wave starts with the
        // matrix pipe while its partner starts with memory instructions.
        // (the first fragments of this tile were read before the loop / under the previous tile's last K-step, so their
        // LDS latency and the pipeline refill hide under the previous tile's epilogue)
#ifdef VQA_STAMPS
        const bool stamp_on = blockIdx.x == kStampWg && ti == kStampTile;
        unsigned long long stamp_t[kStampSlots] = {};
#endif
        if constexpr (DT == VQA_F16) {
            frag_t a1[8], b1[4];  // second register set: the next K-step's fragments load under this K-step's MFMAs
            for (int kt = 0; kt < KT; kt += 2) {  // KT is even (rows are padded to two K-steps)
                const int kappa = ti * KT + kt;
                VQA_KSTEP(a0, b0, a1, b1, kappa, true);
                VQA_KSTEP(a1, b1, a0, b0, kappa + 1, kt + 2 < KT || ti + 1 < ntile);
            }
        } else {
            // fp32: 4x the MFMAs per K-step and one register set (a second one spills): issue the DMA pieces, multiply,
            // then refill the same registers; the reads' latency hides under the partner wave's MFMAs
            // query column groups of this wave that hold a live query (4 for a full tile: the plain block)
            [[maybe_unused]] const int live_q = nq - 64 * wn;
            [[maybe_unused]] const int ng = live_q >= 64 ? 4 : live_q <= 0 ? 0 : (live_q + 15) >> 4;
            for (int kt = 0; kt < KT; ++kt) {
                const int kappa = ti * KT + kt;
                VQA_ISSUE();
                VQA_SB();
                if constexpr (DT == VQA_F32 && !(VQA_ABLATE & 4)) {
                    mma_block_groups<DT>(acc, a0, b0, ng);  // (ONE code path: a second, unconditional copy of the block made the kernel spill)
                } else {
                    VQA_MMA(a0, b0);
                }
                VQA_SB();
                if (kt + 1 < KT || ti + 1 < ntile) VQA_READ_FRAGS(a0, b0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_pieces(kappa + 2);
                VQA_LOOP_BARRIER();
            }
        }

        // X stage of the tile's last K-step: every read of it completed before the re-align barrier and its next
        // refill (K-step + kSx) is issued in the next tile's first L segment
        static_assert(kPx < kSx, "the X stage of the K-step just computed must idle until the next K-step's pieces are issued");
        finish_tile(acc, ti, (sx == 0 ? kSx - 1 : sx - 1));
    }
    };

#ifndef VQA_SLOT_EARLY
#define VQA_SLOT_EARLY 0
#endif
#ifndef VQA_SLOT
#define VQA_SLOT 1  // fp16: 1 = anti-phase slot loop below, 0 = tile_loop (one barrier per K-step; measured 4 % slower)
#endif
    // ---- fp16, anti-phase slots: two barriers per K-step, and in every slot one group only multiplies while the other
    // only moves data, so the two waves of a SIMD (one of each group) never compete for the matrix pipe and never leave
    // it idle behind each other's memory phases (stamped timeline of tile_loop: the younger wave's second MFMA half waits
    // for the older wave's whole MFMA block, then everything waits for it at the barrier).
    //     K-step kappa, slot 1:  group 0: read fragments(kappa), issue Q(kappa + 2)   | group 1: 32 MFMAs of kappa
    //                  slot 2:  group 0: 32 MFMAs of kappa                            | group 1: read fragments(kappa + 1), issue X(kappa + 5)
    // A wave never reads and multiplies at the same time, so ONE fragment register set suffices.  Rings: X 5 stages
    // (X(kappa + 5) goes where X(kappa) was: last read in slot 1), Q 3 stages (Q(kappa + 2) goes where Q(kappa - 1) was);
    // together the same 128 KiB.  Waits: group 0 ends slot 1 with vmcnt(4) (Q(kappa + 1) landed for group 1's reads in
    // slot 2); group 1 ends slot 1 with vmcnt(12) (X(kappa + 1) landed, three K-steps stay in flight).
    // Pieces are always issued (past the end of the stream: the last block again, into a stage nobody reads any more).
    // MODE 1: the X stage of a tile's last K-step is the epilogue's spill area (finish_tile), so group 1 holds that
    // K-step's issue back to the next tile's first K-step (two issues there, vmcnt(8) in between).
    auto slot_loop = [&](auto loader_q_tag) __attribute__((always_inline)) {
        constexpr bool kQ0 = decltype(loader_q_tag)::value;  // group 0: Q loader, multiplies in slot 2
        constexpr int SX = kSxSlot, SQ = 3;
        constexpr int kEarly = VQA_SLOT_EARLY;  // row groups (of 8) a group multiplies right after its memory phase, before the barrier
        static_assert((SX + SQ) * kOperandBytes == (MODE == 2 ? kSkPipe : kPipeBytes), "slot loop rings fill the pipe area");
        const uint32_t ring_lds = kQ0 ? smem_lds + SX * kOperandBytes : smem_lds;
        constexpr int kRing = (kQ0 ? SQ : SX) * kOperandBytes;
        const char* src = kQ0 ? reinterpret_cast<const char*>(Qs)
                              : reinterpret_cast<const char*>(X) + (size_t)first_tile * KT * kOperandBytes;
        uint32_t dst = ring_lds + lw * 4096;
        int n_issued = 0, src_kt = 0;
        auto issue = [&]() __attribute__((always_inline)) {
            if constexpr (kQ0) {
                if constexpr (!(VQA_ABLATE & 33)) glds16x4<false>(src, voff, dst);
            } else {
                if constexpr (!(VQA_ABLATE & 1)) glds16x4<VQA_XNT != 0>(src, voff, dst);
            }
            ++n_issued;
            if (n_issued < total) {  // past the end the cursor stays on the last block
                src += kOperandBytes;
                if (++src_kt == KT) {
                    src_kt = 0;
                    if constexpr (kQ0) src = reinterpret_cast<const char*>(Qs);
                    else src += tile_jump;
                }
            }
            dst += kOperandBytes;
            if (dst >= ring_lds + kRing) dst -= kRing;
        };
        const int a_off = wm * 128 * kRowBytes + frag_off;                          // inside an X stage
        const int b_off = SX * kOperandBytes + wn * 64 * kRowBytes + frag_off;      // inside a Q stage
        int rx = 0, rq = 0;  // stages of the K-step this group reads next
        frag_t fa[8], fb[4];
#if VQA_ABLATE & 2
        for (int i = 0; i < 8; ++i) fa[i] = frag_t{(uint32_t)lane * 2654435761u + i, 0x3c003c00u + lane, 0x12345678u * (i + 1), 0x0badcafeu ^ lane};
        for (int i = 0; i < 4; ++i) fb[i] = frag_t{(uint32_t)lane * 40503u + i, 0x38003800u + lane, 0x87654321u * (i + 1), 0x0defacedu ^ lane};
#endif
#if VQA_ABLATE & 2
#define VQA_SLOT_READ()                                                                                            \
    do {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) asm volatile("" : "+v"(fb[i_]));                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) asm volatile("" : "+v"(fa[i_]));                          \
        if (++rx == SX) rx = 0;                                                                                    \
        if (++rq == SQ) rq = 0;                                                                                    \
    } while (0)
#else
#define VQA_SLOT_READ()                                                                                            \
    do {                                                                                                           \
        const char* xbuf_ = smem + rx * kOperandBytes;                                                             \
        const char* qbuf_ = smem + rq * kOperandBytes;                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                           \
            fb[i_] = *reinterpret_cast<const frag_t*>(qbuf_ + b_off + i_ * 16 * kRowBytes);                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                           \
            fa[i_] = *reinterpret_cast<const frag_t*>(xbuf_ + a_off + i_ * 16 * kRowBytes);                        \
        if (++rx == SX) rx = 0;                                                                                    \
        if (++rq == SQ) rq = 0;                                                                                    \
    } while (0)
#endif
#ifdef VQA_SLOT_SETPRIO
#define VQA_SLOT_PRIO(P) __builtin_amdgcn_s_setprio(P)
#else
#define VQA_SLOT_PRIO(P) (void)0
#endif
// the MFMAs are register-only: these pins keep hipcc from moving them across the slot's barriers
#define VQA_SLOT_MMA(M0, M1)                                                                                       \
    do {                                                                                                           \
        asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), \
                     "+v"(fa[7]));                                                                                 \
        VQA_SB();                                                                                                  \
        VQA_SLOT_PRIO(1);                                                                                          \
        VQA_MMA_RANGE(fa, fb, M0, M1);                                                                             \
        _Pragma("unroll") for (int mi_ = (M0); mi_ < (M1); ++mi_)                                                  \
            asm volatile("" ::"v"(acc[mi_][0]), "v"(acc[mi_][1]), "v"(acc[mi_][2]), "v"(acc[mi_][3]));             \
        VQA_SLOT_PRIO(0);                                                                                          \
        VQA_SB();                                                                                                  \
    } while (0)
        // prologue: Q(0), Q(1) / X(0 .. 4) issued; K-step 0 landed; group 1 holds fragments(0)
        if constexpr (kQ0) {
            issue();
            issue();
            wait_vmcnt<4>();
        } else {
            for (int i = 0; i < SX; ++i) issue();
            wait_vmcnt<4 * (SX - 1)>();
        }
        block_barrier();
        if constexpr (!kQ0) {
            VQA_SLOT_READ();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        block_barrier();
        bool owe = false;  // group 1: an issue was held back at the end of the previous tile
        for (int ti = 0; ti < ntile; ++ti) {
            f32x4 acc[8][4];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef VQA_STAMPS
            const bool stamp_on = blockIdx.x == kStampWg && ti == kStampTile;
            unsigned long long stamp_t[kStampSlots] = {};
#endif
            int ext_stage = 0;
// One K-step; FIRST / LAST = the tile's first / last K-step (compile-time: a run-time choice between MFMA ranges makes hipcc
// spill the accumulators).  Group 1 multiplies the first kEarly row groups of a K-step in the slot before it -- except
// across a tile boundary, where the accumulators still belong to the finished tile.
#define VQA_SLOT_KSTEP(FIRST, LAST, KT_IDX)                                                                          \
    do {                                                                                                             \
        VQA_STAMP(0);                                                                                                \
        if constexpr (kQ0) {                                                                                         \
            VQA_SLOT_READ();                                                                                         \
            VQA_STAMP(1);                                                                                            \
            issue();                                                                                                 \
            VQA_STAMP(2);                                                                                            \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
            VQA_SLOT_MMA(0, kEarly); /* starts before the other group's block has drained (registers only) */        \
            wait_vmcnt<4>();                                                                                         \
        } else {                                                                                                     \
            if constexpr (FIRST) VQA_SLOT_MMA(0, 8);                                                                 \
            else VQA_SLOT_MMA(kEarly, 8);                                                                            \
            VQA_STAMP(1);                                                                                            \
            if (FIRST && owe) wait_vmcnt<4 * (SX - 3)>();                                                            \
            else wait_vmcnt<4 * (SX - 2)>();                                                                         \
            VQA_STAMP(2);                                                                                            \
        }                                                                                                            \
        VQA_STAMP(3);                                                                                                \
        VQA_LOOP_BARRIER();                                                                                          \
        VQA_STAMP(4);                                                                                                \
        if constexpr (kQ0) {                                                                                         \
            VQA_SLOT_MMA(kEarly, 8);                                                                                 \
            VQA_STAMP(5);                                                                                            \
            VQA_STAMP(6);                                                                                            \
        } else {                                                                                                     \
            if constexpr (LAST) ext_stage = rx == 0 ? SX - 1 : rx - 1; /* stage of this tile's last K-step */        \
            VQA_SLOT_READ();                                                                                         \
            VQA_STAMP(5);                                                                                            \
            if (FIRST && owe) {                                                                                      \
                issue();                                                                                             \
                owe = false;                                                                                         \
            }                                                                                                        \
            if (MODE == 1 && LAST) owe = true;                                                                       \
            else issue();                                                                                            \
            if constexpr (kBeta && FIRST) { /* the tile's 256 betas -> LDS: one more LDS-DMA piece of ONE wave.  It sits in that wave's */ \
                /* in-order vmcnt stream behind this K-step's pieces: the counted waits of the next three K-steps become stricter by  */ \
                /* one piece for it, never laxer, and the wait of K-step 4 covers it -- the epilogue reads it behind K-step KT - 1 >= 5 */ \
                if (lw == 0)                                                                                         \
                    glds16x1(sk.beta + ((size_t)first_tile + (size_t)ti * gridDim.x) * kTileRows, (uint32_t)lane * 16,  \
                             smem_lds + (uint32_t)(reinterpret_cast<char*>(sk_beta) - smem));                        \
            }                                                                                                        \
            VQA_STAMP(6);                                                                                            \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
            if constexpr (!LAST) VQA_SLOT_MMA(0, kEarly);                                                            \
        }                                                                                                            \
        VQA_LOOP_BARRIER();                                                                                          \
        VQA_STAMP(7);                                                                                                \
        VQA_STAMP_FLUSH(KT_IDX);                                                                                     \
    } while (0)
            VQA_SLOT_KSTEP(true, false, 0);
            for (int kt = 1; kt < KT - 1; ++kt) VQA_SLOT_KSTEP(false, false, kt);
            VQA_SLOT_KSTEP(false, true, KT - 1);
#undef VQA_SLOT_KSTEP
            if constexpr (kQ0) ext_stage = rx == 0 ? SX - 1 : rx - 1;  // group 0 read this tile's last K-step last
            finish_tile(acc, ti, ext_stage);
        }
        wait_vmcnt<0>();  // the pieces issued past the end of the stream land before this workgroup's LDS is released
#undef VQA_SLOT_READ
#undef VQA_SLOT_MMA
    };

    // ---- fp8: the block-scaled MFMA consumes two K-steps at once (32 bytes per lane and operand), so the loop works on
    // K-step PAIRS, and a second fragment register set does not fit beside the accumulators.  The two wave groups
    // (partners on every SIMD) therefore run ONE SLOT APART, one s_barrier per slot:
    //     group A (waves 0-3, queries   0-127):  R0 M0 R1 M1 ... R(P-1) M(P-1) E | R0 M0 ...
    //     group B (waves 4-7, queries 128-255):  -- R0 M0 R1 ...        R(P-1) M(P-1) E | R0 ...
    // R = read the 24 fragments of a pair, M = its 32 scaled MFMAs, E = threshold test + appends of the finished tile;
    // while one wave of a SIMD multiplies, its partner reads (or runs its epilogue), so the LDS reads no longer idle the
    // matrix pipe.  Groups split the QUERIES: each owns its half of every query stage (refilled one slot after it was
    // read) and its 128 candidate lists (appends, compaction and the rare refusal repair stay inside the group; the
    // other group idles through a repair's barriers).  X pairs are shared: pair j is read by A in slot a, by B in
    // slot a + 1, and refilled with pair j + 3 at slot a + 2 (3 pairs of ring = 4 slots of HBM prefetch, as before).
    // Waves 4-7 issue the X stream (their M slots), waves 0-3 the Q halves (every slot; vmcnt retires in order, so the
    // deep and the shallow stream stay in different waves).
    auto stagger_loop = [&](auto early_tag) __attribute__((always_inline)) {
        constexpr bool kEarly = decltype(early_tag)::value;
        const int P = KT >> 1;     // K-step pairs per tile (KT is even: pairs never straddle tiles)
        const int J = ntile * P;   // pairs of this workgroup
        const int fq = grp * 128, fq_other = 128 - fq;  // the groups' "append refused" flags: bit 30 of cnt[fq]
        const uint32_t voff_lane = (uint32_t)lane * 16;
        // rows [128 half, +128) of both query stages of K-step pair jj: 16 KiB = 4 waves x 4 KiB
        auto issue_q_half = [&](int half, int jj) {
            const int off = (lw >> 1) * kOperandBytes + half * (kOperandBytes / 2) + (lw & 1) * 4096;
            glds16x4<false>(reinterpret_cast<const char*>(Qs) + (size_t)(2 * jj) * kOperandBytes + off, voff_lane,
                            qring_lds + off);
        };
        v8i a2[8], b2[4];
#if VQA_ABLATE & 2
        for (int i = 0; i < 8; ++i) a2[i] = v8i{lane, 1, 2, 3, 4, 5, 6, 7};
        for (int i = 0; i < 4; ++i) b2[i] = v8i{7, 6, 5, lane, 3, 2, 1, 0};
#endif
        f32x4 acc[8][4];
        uint32_t pend[4] = {0u, 0u, 0u, 0u};
        uint32_t row0 = 0;
        // a group's refused appends: compact its lists, retry, until none is refused; every wave runs the barriers
        auto repair = [&](bool mine, int f) __attribute__((always_inline)) {
            for (;;) {
                if (!(L.cnt[f] & kOverBit)) break;  // stable: set before the last barrier, cleared only behind the next
                if (mine) compact_pass(L, wave, lane, k, k + 1, kCap);
                __syncthreads();
                if (mine && lw == 0 && lane == 0) L.cnt[f] &= ~kOverBit;
                __syncthreads();
                if (mine && process_pending(L, acc, pend, wm, wn, c, g, row0, upper, 0)) atomicOr(&L.cnt[f], kOverBit);
                __syncthreads();
            }
        };
        // All LDS-DMA is issued from R slots: the reading wave has slack, while a piece issued in front of the MFMAs costs
        // the matrix pipe ~100 cycles each (measured: M slots of 8 pieces + 32 MFMAs ran ~2000 cycles).  Per R slot a
        // wave issues the OTHER group's next query half first (it must land within the slot) and then one K-step of the X
        // stream (group A the even K-steps, group B the odd ones; forced complete two slots later by the next counted wait
        // -- vmcnt retires in order, so the shallow pieces go first and vmcnt(4) leaves the X pieces in flight).
        static_assert(kSx % 2 == 0, "the fp8 loop alternates the X K-steps between the two groups");
        int xk = grp, x_kt = grp;  // next K-step this group issues, its position inside the tile
        const char* xs = reinterpret_cast<const char*>(X) + ((size_t)first_tile * KT + grp) * kOperandBytes;
        uint32_t xd = xring_lds + grp * kOperandBytes + lw * 4096;
        auto issue_x2 = [&]() -> bool {
            if (xk >= total) return false;
            glds16x4<VQA_XNT != 0>(xs, voff, xd);
            xk += 2;
            xs += 2 * kOperandBytes;
            x_kt += 2;
            if (x_kt >= KT) {
                x_kt -= KT;
                xs += tile_jump;
            }
            xd += 2 * kOperandBytes;
            if (xd >= xring_lds + kXRingBytes) xd -= kXRingBytes;
            return true;
        };
        // prologue: the ring full (K-steps 0 .. kSx - 1), pair 0 and group A's query half of pair 0 landed; B idles in slot 0
        {
            if (kEarly && J > 0) issue_q_half(0, 0);
            int n = 0;
            for (int i = 0; i < kSx / 2; ++i) n += issue_x2() ? 1 : 0;
            if (n == kSx / 2) wait_vmcnt<4 * (kSx / 2 - 1)>();  // this group's first K-step landed
            else wait_vmcnt<0>();
        }
        block_barrier();
        if (!kEarly) block_barrier();
        int j = 0;
        for (int ti = 0; ti < ntile; ++ti) {
            for (int jj = 0; jj < P; ++jj, ++j) {
                // ---- R slot: fragments of pair j
#ifdef VQA_STAMPS
                const bool stamp_on = blockIdx.x == kStampWg && ti == kStampTile;
                unsigned long long stamp_t[kStampSlots] = {};
#endif
                VQA_STAMP(0);
                if (MODE == 1 && jj == 0 && ti > 0) compact_pass(L, wave, lane, k, k + (kCap - k + 1) / 2, kCap);
                bool xi = false;
#if !(VQA_ABLATE & 33)
                if (kEarly) {
                    issue_q_half(1, jj);  // group B's half of pair j (B read its half of pair j - 1 one slot ago)
                } else if (j + 1 < J) {
                    issue_q_half(0, jj + 1 == P ? 0 : jj + 1);  // group A's half of pair j + 1 (A read pair j one slot ago)
                }
#endif
#if !(VQA_ABLATE & 1)
                if (j >= 1) xi = issue_x2();  // into pair j - 1's stages: both groups have read them
#endif
                VQA_STAMP(1);
#if VQA_ABLATE & 2
                sx = sx + 2 >= kSx ? sx + 2 - kSx : sx + 2;
#else
                VQA_READ_HALVES(a2, b2, 0);
                VQA_READ_HALVES(a2, b2, 1);
#endif
                VQA_STAMP(2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                VQA_STAMP(3);
                if (xi) wait_vmcnt<4>();
                else wait_vmcnt<0>();
                VQA_STAMP(4);
                block_barrier();
                VQA_STAMP(5);
                if (MODE == 1 && kEarly && jj == 0 && ti > 0) repair(false, fq_other);  // group B's E slot just ended
                // ---- M slot: nothing but the 32 MFMAs
                if (jj == 0) {
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // The MFMAs are register-only, so nothing but these two pins keeps hipcc from moving them across the slot's
                // barriers (it sank them below the closing barrier: both groups then multiplied in their R slots).
                asm volatile("" : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]),
                             "+v"(a2[7]));
                VQA_SB();
#if !(VQA_ABLATE & 4)
                mma_pair<DT>(acc, a2, b2);
#endif
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    asm volatile("" ::"v"(acc[mi][0]), "v"(acc[mi][1]), "v"(acc[mi][2]), "v"(acc[mi][3]));
                VQA_STAMP(6);
                block_barrier();
                VQA_STAMP(7);
                VQA_STAMP_FLUSH(jj);
                if (MODE == 1 && !kEarly && jj == P - 1) repair(false, fq_other);  // group A's E slot just ended
            }
            // ---- E slot
            row0 = ((uint32_t)first_tile + (uint32_t)ti * gridDim.x) * kTileRows;
            mask_ragged(acc, row0);
            if (MODE == 0) {
                seed_epilogue(acc, row0, ti);
                block_barrier();
            } else {
                if (append_epilogue(acc, row0, pend, 0)) atomicOr(&L.cnt[fq], kOverBit);
                block_barrier();
                repair(true, fq);
            }
        }
        if (kEarly) {  // group B's last E slot
            block_barrier();
            if (MODE == 1) repair(false, fq_other);
        }
    };

    if constexpr (kStagger) {
        if (grp) stagger_loop(std::false_type{});
        else stagger_loop(std::true_type{});
    } else if constexpr (VQA_SLOT != 0 && (DT == VQA_F16 || DT == VQA_I8_SKETCH)) {
        if (grp) slot_loop(std::false_type{});
        else slot_loop(std::true_type{});
    } else {
        if (grp) tile_loop(std::false_type{});
        else tile_loop(std::true_type{});
    }

#ifdef VQA_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
    if (MODE == 0) return;
    if (MODE == 2) {  // pairs written to this workgroup's region; a region that filled up sends the search to its exact fallback
        __syncthreads();
        if (tid == 0) {
            const int cnt = *sk_cnt;
            sk.counts[blockIdx.x] = (unsigned)(cnt < sk.cap ? cnt : sk.cap);
            if (cnt > sk.cap) atomicExch(sk.overflow, 1);
        }
        return;
    }
    // ---- flush: every list sorted best first, k keys per query (0 = empty) ------------------------------------
    __syncthreads();
    compact_pass(L, wave, lane, k, 1, kCap);
    __syncthreads();
    // query-major [query][workgroup][k]: K2 then reads one contiguous run per query
    for (int i = tid; i < kQ * k; i += kThreads) {
        const int q = i / k, j = i - q * k;
        out[((size_t)q * row_lists + list_offset + blockIdx.x) * k + j] = j < list_count(L, q) ? L.cand[q * kCap + j] : 0ull;
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

int vqa_score_topk_seeds_per_tile() { return kSeedsPerTile; }

template <int DT, int LOOP>
static int launch_dt(const ScoreTopkArgs& a, int KT, int lds, hipStream_t stream) {
    static VqaPerDeviceOnce once;
    int rc = once.run([&](int) -> int {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<0, DT, 0, LOOP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<1, DT, 0, LOOP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<1, DT, 1, LOOP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return VQA_OK;
    });
    if (rc != VQA_OK) return rc;
    auto kern = a.seed_only ? score_topk_kernel<0, DT, 0, LOOP>
                            : a.first_stage ? score_topk_kernel<1, DT, 1, LOOP> : score_topk_kernel<1, DT, 0, LOOP>;
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(kThreads), lds, stream, a.x, a.q, a.thr_init, a.upper, a.partial, (long long)a.n,
                       KT, a.nq, a.k, a.tile_begin, a.tile_end, a.gate, a.row_lists > 0 ? a.row_lists : a.grid, a.list_offset,
                       a.seeds_per_tile, SketchScanArgs{});
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

// MODE 2: the int8 sketch scan of a large fp16 / fp32 shard (STAGE 1: the same code under a second symbol for the first-stage scan of
// the cascade, so that a kernel trace keeps the two launches of a search apart)
static int launch_sketch(const ScoreTopkArgs& a, int KT, int lds, hipStream_t stream) {
    static VqaPerDeviceOnce once;
    int rc = once.run([&](int) -> int {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2, VQA_I8_SKETCH, 0, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2, VQA_I8_SKETCH, 1, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2, VQA_I8_SKETCH, 0, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2, VQA_I8_SKETCH, 1, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2, VQA_I8_SKETCH, 0, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2, VQA_I8_SKETCH, 1, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return VQA_OK;
    });
    if (rc != VQA_OK) return rc;
    // six X stages (LOOP 0) where the workgroup's tile maxima fit beside them, five (LOOP 1) for the largest shards
    const int per_wg = (a.tile_end - a.tile_begin + a.grid - 1) / a.grid;
    const bool six = a.loop != 1 && per_wg <= kSketchMaxTiles6;  // (a.loop == 0: VQA_SKETCH_SX=6 at index create, dev / A-B switch)
    auto kern = a.sketch->beta ? (a.first_stage ? score_topk_kernel<2, VQA_I8_SKETCH, 1, 2> : score_topk_kernel<2, VQA_I8_SKETCH, 0, 2>)
                : six          ? (a.first_stage ? score_topk_kernel<2, VQA_I8_SKETCH, 1, 0> : score_topk_kernel<2, VQA_I8_SKETCH, 0, 0>)
                               : (a.first_stage ? score_topk_kernel<2, VQA_I8_SKETCH, 1, 1> : score_topk_kernel<2, VQA_I8_SKETCH, 0, 1>);
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(kThreads), lds, stream, a.x, a.q, nullptr, nullptr, nullptr, (long long)a.n, KT, a.nq, a.k,
                       a.tile_begin, a.tile_end, a.gate, a.grid, 0, 2, *a.sketch);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_score_topk_sketch_max_tiles() { return kSketchMaxTiles; }

int vqa_launch_score_topk(int dtype, const ScoreTopkArgs& a, hipStream_t stream) {
    if (a.sketch) {
        VQA_REQUIRE(dtype == VQA_I8_SKETCH && a.d_pad % (2 * kRowBytes) == 0, "score_topk: sketch scan over %d-element rows", a.d_pad);
        VQA_REQUIRE(a.nq >= 1 && a.nq <= kQ && a.grid >= 1 && a.tile_end > a.tile_begin && a.grid <= a.tile_end - a.tile_begin,
                    "score_topk: bad sketch launch");
        VQA_REQUIRE((a.tile_end - a.tile_begin + a.grid - 1) / a.grid <= kSketchMaxTiles, "score_topk: %d tiles per workgroup exceed the sketch scan's %d",
                    (a.tile_end - a.tile_begin + a.grid - 1) / a.grid, kSketchMaxTiles);
        VQA_REQUIRE(a.sketch->tile_info && a.sketch->qconst && a.sketch->regions && a.sketch->counts && a.sketch->overflow && a.sketch->cap > 0,
                    "score_topk: incomplete sketch arguments");
        VQA_REQUIRE(!a.sketch->beta || (a.sketch->tile_c && a.d_pad / kRowBytes >= 6),
                    "score_topk: the per-row form needs the tiles' max |beta| and rows of at least six K-steps");
        if (a.regq && vqa_sketch_regq_applies(a)) return vqa_launch_sketch_regq(a, stream);
        return launch_sketch(a, a.d_pad / kRowBytes, vqa_score_topk_lds_bytes(dtype, a.k), stream);
    }
    VQA_REQUIRE(dtype == VQA_F16 || dtype == VQA_FP8_E4M3 || dtype == VQA_F32, "score_topk: storage type %d", dtype);
    const int esize = dtype == VQA_F32 ? 4 : dtype == VQA_F16 ? 2 : 1;
    VQA_REQUIRE(a.d_pad > 0 && (a.d_pad * esize) % (2 * kRowBytes) == 0,
                "score_topk: padded row of %d elements x %d B is not a multiple of %d B", a.d_pad, esize, 2 * kRowBytes);
    VQA_REQUIRE(a.k >= 1 && a.k <= kMaxK, "score_topk: k=%d outside [1, %d]", a.k, kMaxK);
    VQA_REQUIRE(a.nq >= 1 && a.nq <= kQ, "score_topk: nq=%d outside [1, %d]", a.nq, kQ);
    VQA_REQUIRE(a.grid >= 1 && a.tile_end > a.tile_begin, "score_topk: empty launch");
    VQA_REQUIRE(a.seeds_per_tile == 2 || a.seeds_per_tile == 8, "score_topk: %d seeds per tile", a.seeds_per_tile);
    VQA_REQUIRE(a.grid <= a.tile_end - a.tile_begin, "score_topk: %d workgroups for %d tiles (every workgroup needs a tile)", a.grid,
                a.tile_end - a.tile_begin);
    const int lds = vqa_score_topk_lds_bytes(dtype, a.k);
    const int KT = a.d_pad * esize / kRowBytes;  // even
    if (dtype == VQA_F16) return a.loop == 1 ? launch_dt<VQA_F16, 1>(a, KT, lds, stream) : launch_dt<VQA_F16, 0>(a, KT, lds, stream);
    if (dtype == VQA_FP8_E4M3) return launch_dt<VQA_FP8_E4M3, 0>(a, KT, lds, stream);
    return launch_dt<VQA_F32, 0>(a, KT, lds, stream);
}

#ifdef VQA_STAMPS
extern "C" int vqa_debug_read_stamps(unsigned long long* out, int n) {
    const size_t bytes = sizeof(unsigned long long) * (size_t)(n < 8 * 64 * kStampSlots ? n : 8 * 64 * kStampSlots);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
