// K1r -- the int8 sketch scan with the QUERY operand resident in registers (round 6), gfx950 only.
//
// Same job, same arguments and same results as score_topk.hip MODE 2 (the sketch search's scan: D = q_int . x_int on
// v_mfma_i32_16x16x64_i8, integer threshold test per (query, tile), survivors appended to the workgroup's region of candidate
// pairs); reference call site inference_pipeline/db_utils/heavy_ranker.py:98-101 (txtai -> faiss IndexFlatIP behind it).
//
// Why a second kernel: in score_topk.hip the 256-query tile is re-streamed L2 -> LDS -> registers for EVERY corpus tile (one 16 KiB Q
// block per 16 KiB X block: rocprof TCC hits = misses), which doubles the LDS-DMA pieces, adds 4 of the 12 fragment reads per K-step
// and couples two loader groups through two barriers per K-step.  For rows of <= 768 one-byte elements a wave's share of the query
// tile fits its registers for the whole launch:
//   * workgroup = 8 waves (two per SIMD, one workgroup per CU); wave v owns queries [32 v, +32): its B fragments of all KT K-steps
//     (KT x 2 x 4 = 96 registers at d = 768) are loaded ONCE, beside 64 accumulators (128 rows x 32 queries) and ONE set of 8 A fragments;
//   * the corpus streams through a 16-stage LDS ring in steps of 8 KiB = one K-step (64 B of every row) of one HALF tile (128 rows);
//     HBM -> LDS by LDS-DMA (nt), 12 steps = 96 KiB in flight per CU, ONE 1 KiB piece per wave and step;
//   * a wave alternates a memory phase (8 ds_read_b128 = the step's A fragments, its DMA piece of step s + 12) and a matrix phase
//     (16 MFMAs); the two waves of a SIMD run these in ANTI-PHASE (group 0 = waves 0-3: memory, matrix; group 1 = waves 4-7: a phase
//     behind), so one multiplies while its partner moves data -- a DMA piece issued into a saturated memory pipe stalls its wave for
//     hundreds of cycles (measured: the 4-wave form of this kernel, where the multiplying wave issued its own pieces, ran 1.15 ms
//     against 0.77 ms for its MFMAs alone and 0.84 ms for its stream alone: profiles/r06_regq_ablation.txt);
//   * ONE workgroup barrier per four steps: ring stages are recycled a group at a time (argument at the loop);
//   * after the KT steps of a half tile: the threshold test on the accumulators (the arithmetic of score_topk.hip's sketch_epilogue),
//     under the partner wave's matrix phase.
// LDS and LDS-DMA carry X only; the Q stream, its ring, its loader group and a third of the fragment reads are gone.
// Per-row (beta) shards and rows of other lengths stay on score_topk.hip's slot loop.
#include <type_traits>

#include "vqa_common.h"

#ifndef VQA_RQ_ABLATE
// dev-only timing ablations, bit mask; results are wrong when != 0: 1 no LDS-DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue, 16 the
// threshold test without the append (rare path),
// 64 no loop barriers
#define VQA_RQ_ABLATE 0
#endif

namespace {

typedef unsigned int frag_t __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_char_ptr;

constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int kQ = VQA_QUERY_TILE;
constexpr int kRowBytes = 64;
constexpr int kBlockBytes = 256 * kRowBytes;  // one (tile, K-step) block of the tiled layout: 16 KiB
constexpr int kStepBytes = kBlockBytes / 2;   // one step: the block's half of 128 rows
constexpr int kPiece = kStepBytes / kWaves;   // 1 KiB per wave and step
constexpr int kS = 16;                        // ring stages (128 KiB)
constexpr int kLdsTotal = 160 * 1024;
constexpr int kRingBytes = kS * kStepBytes;
constexpr int kQcBytes = kSketchQRows * kQ * 4;
constexpr int kTileInfoOff = kRingBytes + kQcBytes + 16;
constexpr int kMaxTiles = (kLdsTotal - kTileInfoOff) / 16;
static_assert(kPiece == 1024, "one LDS-DMA wave-instruction per wave and step");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one wave-instruction of LDS-DMA: 64 lanes x 16 B -> 1 KiB of LDS; global address = sbase + voff, LDS address = M0 + lane * 16 (added
// by hardware).  Inline asm: hipcc keeps no scoreboard entry, all ordering is by counted vmcnt.
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// two wave-instructions (the asymmetric form: group 1's waves move 2 KiB of every step each)
__device__ __forceinline__ void glds16x2(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

#define VQA_RQ_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef VQA_RQ_STAMPS
// dev-only diagnostic build (scripts/rq_stamps.py): s_memtime stamps of one workgroup's steps during one half tile, rolling form.  Per
// barrier group of 4 steps 16 slots: 3 per step (start, in front of / behind the DMA piece), then 12 = in front of the group's waits,
// 13 = behind them, 14 = behind the barrier.  Written by scalar stores to a buffer nothing else reads; results of this build may be
// wrong (the stamps sit in lgkmcnt, which hipcc's counted waits do not know): build it with VQA_RQ_ABLATE=8.
constexpr int kStampWg = 5, kStampU = 40;
__device__ unsigned long long g_rq_stamps[8 * 3 * 16];
#define VQA_RQ_STAMP(J)                                                 \
    do {                                                                \
        if (stamp_on) asm volatile("s_memtime %0" : "=s"(stamp_t[J]));  \
    } while (0)
#else
#define VQA_RQ_STAMP(J) (void)0
#endif
#ifndef VQA_RQ_DB
#define VQA_RQ_DB 1  // 1: rolling fragment set, both wave groups run the same code; 0: the groups in explicit anti-phase (memory / matrix phases)
#endif
#ifndef VQA_RQ_PRIO
#define VQA_RQ_PRIO 2  // rolling form: the s_setprio level a wave takes on its steps of odd (kt + group); 0: never touched
#endif
#ifndef VQA_RQ_ASYM
#define VQA_RQ_ASYM 0  // rolling form: 1 = only group 1's waves issue LDS-DMA (two pieces each) and hold the higher priority throughout
#endif
#ifndef VQA_RQ_DMA_AT
#define VQA_RQ_DMA_AT 3  // rolling form: the step's DMA piece goes out behind this row group's MFMAs
#endif

template <int KT, int STAGE>
__global__ __launch_bounds__(kThreads, 1) void sketch_scan_regq_kernel(const void* __restrict__ X, const void* __restrict__ Qs, long long N,
                                                                       int nq, int tile_begin, int tile_end, const int* __restrict__ gate,
                                                                       SketchScanArgs sk) {
    static_assert(KT % 2 == 0 && KT <= 12, "rows of up to 768 one-byte elements, an even number of K-steps");
#ifdef VQA_RQ_G
    constexpr int G = KT % VQA_RQ_G == 0 ? VQA_RQ_G : 2;  // dev: another group length
#else
    constexpr int G = KT % 4 == 0 ? 4 : 2;  // steps per barrier group (KT % G == 0: a half tile is a whole number of groups)
#endif
    constexpr bool kDB = VQA_RQ_DB != 0;    // rolling form: step s reads the fragments of step s + 1 under its own MFMAs
    constexpr int kAhead = kDB ? 1 : 0;     // a group's reads reach this many steps past its last step
#ifdef VQA_RQ_P
    constexpr int P = VQA_RQ_P;             // dev: fewer steps in flight than the ring allows
    static_assert(P <= kS - G + kAhead && P >= G + 1 + kAhead, "steps in flight");
#else
    constexpr int P = kS - G + kAhead;      // steps in flight (see the ring argument below)
#endif
    constexpr int kWaitPieces = (P - G - kAhead) * ((kDB && VQA_RQ_ASYM != 0) ? 2 : 1);  // pieces that may stay in flight behind a group barrier
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_lds = (uint32_t)(size_t)(lds_char_ptr)smem;
    float* const sk_q = reinterpret_cast<float*>(smem + kRingBytes);  // [kSketchQRows][256]
    int* const sk_cnt = reinterpret_cast<int*>(smem + kRingBytes + kQcBytes);
    float4* const sk_tm = reinterpret_cast<float4*>(smem + kTileInfoOff);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD: one of each group
    const int c = lane & 15, g = lane >> 4;

    if (tid < kQ) {  // per-query constants (score_topk.hip MODE 2: queries past the batch get theta = +inf and benign factors)
        const bool live = tid < nq;
        sk_q[tid] = live ? sk.qconst[tid] : INFINITY;
        sk_q[kQ + tid] = live ? sk.qconst[kQ + tid] : 0.f;
        sk_q[2 * kQ + tid] = live ? sk.qconst[2 * kQ + tid] : 0.f;
        sk_q[3 * kQ + tid] = live ? sk.qconst[3 * kQ + tid] : 1.f;
        sk_q[4 * kQ + tid] = live ? sk.qconst[4 * kQ + tid] : 0.f;
        sk_q[5 * kQ + tid] = live ? sk.qconst[5 * kQ + tid] : 0.f;
        if (tid == 0) *sk_cnt = 0;
    }
    const int first_tile = tile_begin + blockIdx.x;
    const int ntile = first_tile < tile_end ? (tile_end - first_tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int total = ntile * 2 * KT;  // steps of this workgroup: s = (ti * 2 + h) * KT + kt
    for (int t = tid; t < ntile; t += kThreads) {  // (max ||x_hi||, max ||x_lo||, 1 / scale, max |w . x_lo|) of every tile scanned here
        float4 ti = sk.tile_info[first_tile + t * (int)gridDim.x];
        ti.w = sk.tile_c ? sk.tile_c[first_tile + t * (int)gridDim.x] : 0.f;
        sk_tm[t] = ti;
    }

    // ---- the wave's queries: B fragments of every K-step, loaded once.  Lane (c, g) of fragment ni holds bytes [16 g', +16) of query
    // 32 wave + 16 ni + c in K-step kt (g' = the tiled layout's swizzled slot: rows and queries are read the same way).
    const int frag_off = c * kRowBytes + ((g ^ (((c >> 3) & 1) * 3)) << 4);
    frag_t qb[KT][2];
    {
        const char* qsrc = reinterpret_cast<const char*>(Qs);
        const uint32_t qoff = (uint32_t)(wave * 32 * kRowBytes + frag_off);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                             : "=&v"(qb[kt][ni])
                             : "v"(qoff + (uint32_t)(kt * kBlockBytes)), "s"(qsrc), "n"(ni * 16 * kRowBytes)
                             : "memory");
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)  // the loads are invisible to hipcc's waitcnt pass: this wait orders every use behind them
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(qb[kt][0]), "+v"(qb[kt][1])::"memory");
    }

    // ---- the X stream: step s = 8 KiB at X + ((first_tile + ti grid) KT + kt) 16 KiB + h 8 KiB -> ring stage s % kS.  Wave v moves
    // bytes [v KiB, +1 KiB) of every step: one piece, issued in its memory phase of step s for step s + P.
    // Ring argument.  All fragment reads of step s and the issue of step s + P happen in the memory phase mem(s) of each wave.  A
    // workgroup barrier B(s0) stands, in every wave's program order, between mem(s0 - 1) and mem(s0) for every s0 that is a multiple
    // of G (group 0 passes it right in front of mem(s0), group 1 right behind mem(s0 - 1): the two groups sit a phase apart, the
    // barrier keeps them there).  Before B(s0) a wave waits for its pieces of steps <= s0 + G - 1, so everything read between B(s0) and
    // B(s0 + G) has landed; a piece issued in that interval goes to the stage of step s + P - kS <= s0 + G - 1 + P - kS, which every
    // wave read before B(s0) if P <= kS - G.
    constexpr bool kAsym = kDB && VQA_RQ_ASYM != 0;
    const uint32_t voff = kAsym ? (uint32_t)((wave & 3) * 2 * kPiece + lane * 16) : (uint32_t)(wave * kPiece + lane * 16);
    const char* src = reinterpret_cast<const char*>(X) + (size_t)first_tile * KT * kBlockBytes;
    const long long to_half1 = (long long)kStepBytes - (long long)KT * kBlockBytes;
    const long long to_next_tile = (long long)gridDim.x * KT * kBlockBytes - (long long)kStepBytes - (long long)KT * kBlockBytes;
    int n_issued = 0, i_kt = 0, i_h = 0;
    uint32_t dst = kAsym ? smem_lds + (wave & 3) * 2 * kPiece : smem_lds + wave * kPiece;
    auto issue = [&]() __attribute__((always_inline)) {
        if constexpr (!(VQA_RQ_ABLATE & 1)) {
            if constexpr (kAsym) {
                if (grp) glds16x2(src, voff, dst);
            } else {
                glds16(src, voff, dst);
            }
        }
        ++n_issued;
        if (n_issued < total) {  // past the end the cursor stays on the last step (its pieces land in stages nobody reads any more)
            src += kBlockBytes;
            if (++i_kt == KT) {
                i_kt = 0;
                src += i_h ? to_next_tile : to_half1;
                i_h ^= 1;
            }
        }
        dst += kStepBytes;
        if (dst >= smem_lds + kRingBytes) dst -= kRingBytes;
    };

    // prologue: P steps issued, the first group's steps landed everywhere = B(0)
    for (int i = 0; i < P; ++i) issue();
    wait_vmcnt<kWaitPieces>();
    __builtin_amdgcn_s_barrier();

#if VQA_RQ_ABLATE & 64
#define VQA_RQ_BARRIER() (void)0
#else
#define VQA_RQ_BARRIER() __builtin_amdgcn_s_barrier()
#endif

    // ---- the threshold test of one finished half tile (the arithmetic of score_topk.hip's sketch_epilogue)
    auto half_tile_epilogue = [&](const i32x4 (&acc)[8][2], int u) __attribute__((always_inline)) {
                const int ti = u >> 1, h = u & 1;
                const uint32_t row0 = ((uint32_t)first_tile + (uint32_t)ti * gridDim.x) * 256u + (uint32_t)h * 128u;
                const float4 tmax = sk_tm[ti];
                const float a_hi = tmax.x, b_lo = tmax.y, inv_sx = tmax.z, c_w = tmax.w;
                float T[2];
                int mx[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int q = wave * 32 + ni * 16 + c;
                    // (the fp32 summation error of the exact scores against the real-number dot product the bound speaks of rides in
                    // the per-query factors: sketch_qconst_kernel; c_w: the tile's max |w . x_lo| against |alpha|, the split slack term)
                    const float num = sk_q[q] - sk_q[kQ + q] * a_hi - sk_q[2 * kQ + q] * b_lo - sk_q[4 * kQ + q] * c_w;
                    const float t = num * sk_q[3 * kQ + q] * inv_sx;
                    T[ni] = t - fabsf(t) * 4e-6f - 0.5f;  // every rounding of this line errs towards MORE candidates; D is an integer
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    int r[8];
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        const i32x4 v = acc[mi][ni];
                        r[mi] = max(max(max(v[0], v[1]), v[2]), v[3]);
                    }
                    mx[ni] = max(max(max(max(r[0], r[1]), r[2]), max(max(r[3], r[4]), r[5])), max(r[6], r[7]));
                }
                if constexpr (bool(VQA_RQ_ABLATE & 16)) {  // the test without the append: what the rare path costs (results are wrong)
                    asm volatile("" ::"v"(mx[0]), "v"(mx[1]), "v"(T[0]), "v"(T[1]));
                } else
                if (((float)mx[0] >= T[0]) | ((float)mx[1] >= T[1])) {
                    unsigned long long* region = sk.regions + (size_t)blockIdx.x * sk.cap;
                    // (rare path: nothing it needs is kept in registers across the K loop -- hipcc hoists the row offsets out of the
                    // tile loop otherwise)
                    uint32_t rbase = row0 + (uint32_t)(g * 4);
                    asm volatile("" : "+v"(rbase));
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if ((float)mx[ni] >= T[ni]) {
                            const unsigned long long qhi = (unsigned long long)(wave * 32 + ni * 16 + c) << 32;
#pragma unroll
                            for (int mi = 0; mi < 8; ++mi) {
                                const i32x4 v = acc[mi][ni];
                                if ((float)max(max(v[0], v[1]), max(v[2], v[3])) >= T[ni]) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const uint32_t pos = rbase + (uint32_t)(mi * 16 + j);
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

    const bool live_wave = nq > 32 * wave;  // a wave whose 32 queries are all padding multiplies nothing (small batches)
    auto scan = [&](auto live_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr bool kLive = decltype(live_tag)::value;
        constexpr bool kMemFirst = decltype(first_tag)::value;  // group 0: B(s0) in front of mem(s0); group 1: behind mem(s0 - 1)
        frag_t fa[8];
#if VQA_RQ_ABLATE & 2
        for (int i = 0; i < 8; ++i) fa[i] = frag_t{(uint32_t)lane * 2654435761u + i, 0x3c003c00u + lane, 0x12345678u * (i + 1), 0x0badcafeu ^ lane};
#endif
        int rd = 0;  // ring stage of the step whose fragments are read next
// memory phase of one step: the step's 8 A fragments, this wave's piece of step + P
#define VQA_RQ_MEM()                                                                                                  \
    do {                                                                                                              \
        VQA_RQ_SB();                                                                                                  \
        if constexpr (kLive) {                                                                                        \
            [[maybe_unused]] const char* xb_ = smem + rd * kStepBytes + frag_off;                                     \
            rd = (rd + 1) & (kS - 1);                                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                        \
                if constexpr (bool(VQA_RQ_ABLATE & 2)) asm volatile("" : "+v"(fa[i_]));                               \
                else fa[i_] = *reinterpret_cast<const frag_t*>(xb_ + i_ * 16 * kRowBytes);                            \
            }                                                                                                         \
        }                                                                                                             \
        issue();                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        VQA_RQ_SB();                                                                                                  \
    } while (0)
// matrix phase: 16 MFMAs (8 row groups x 2 query groups), nothing else
#define VQA_RQ_MMA(KTI)                                                                                               \
    do {                                                                                                              \
        if constexpr (kLive) {                                                                                        \
            asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7])); \
            VQA_RQ_SB();                                                                                              \
            _Pragma("unroll") for (int mi_ = 0; mi_ < 8; ++mi_) _Pragma("unroll") for (int ni_ = 0; ni_ < 2; ++ni_) { \
                if constexpr (bool(VQA_RQ_ABLATE & 4)) asm volatile("" : "+v"(acc[mi_][ni_]) : "v"(fa[mi_]), "v"(qb[KTI][ni_])); \
                else acc[mi_][ni_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, fa[mi_]),        \
                                                                           __builtin_bit_cast(i32x4, qb[KTI][ni_]), acc[mi_][ni_], 0, 0, 0); \
            }                                                                                                         \
            _Pragma("unroll") for (int mi_ = 0; mi_ < 8; ++mi_) asm volatile("" ::"v"(acc[mi_][0]), "v"(acc[mi_][1])); \
            VQA_RQ_SB();                                                                                              \
        }                                                                                                             \
    } while (0)
#define VQA_RQ_GROUP_BARRIER()                                                                                        \
    do {                                                                                                              \
        wait_vmcnt<kWaitPieces>(); /* this wave's pieces of the next group's steps have landed */                           \
        VQA_RQ_BARRIER();                                                                                             \
    } while (0)

        for (int u = 0; u < 2 * ntile; ++u) {  // half tile u = 2 ti + h
            i32x4 acc[8][2];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = i32x4{0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                VQA_RQ_MEM();
                if constexpr (!kMemFirst) {
                    if ((kt + 1) % G == 0) VQA_RQ_GROUP_BARRIER();  // B(s + 1) behind mem(s)
                }
                VQA_RQ_MMA(kt);
                if constexpr (kMemFirst) {
                    if ((kt + 1) % G == 0) VQA_RQ_GROUP_BARRIER();  // B(s + 1) in front of mem(s + 1)
                }
            }
            // the half tile is complete: threshold test + appends (under the partner wave's matrix phase)
            if constexpr (kLive && !(VQA_RQ_ABLATE & 8)) half_tile_epilogue(acc, u);
        }
#undef VQA_RQ_MEM
#undef VQA_RQ_MMA
#undef VQA_RQ_GROUP_BARRIER
    };

    // ---- rolling form (VQA_RQ_DB = 1).  ONE fragment set, software-pipelined register by register: right behind the two MFMAs of row
    // group mi of step s, fa[mi] is refilled with row group mi of step s + 1 -- its next use is a whole step (14 MFMAs, > 220 cycles)
    // away, so the LDS latency never shows and no second register set is needed (a second set put the kernel at 256 registers + 21
    // spilled).  The wave's DMA piece of step s + P goes out in the middle of the step: its issue stalls the wave for ~100 cycles and
    // more when the memory pipe is saturated, and the SIMD's partner wave multiplies meanwhile.  Both groups run the same code; nothing
    // forces a phase relation, whichever wave of a SIMD has MFMAs ready feeds the pipe; no scheduling pins either (with them hipcc's
    // register allocation went to 256 + spills, and a spill reload's `s_waitcnt vmcnt(0)` drains the DMA queue).  (The anti-phase form's memory phase -- reads,
    // their latency, the DMA issue: ~350 cycles -- was longer than its 256-cycle matrix phase: 0.95 ms as soon as ONE SIMD held two live
    // waves, 0.81 ms = the stream with one live wave per SIMD; profiles/r06_regq_ablation.txt.)
    // Ring argument: B(s0) closes the group of steps [s0 - G, s0); inside [s0, s0 + G) a wave reads the fragments of steps s0 + 1 ..
    // s0 + G (it waits for its pieces of steps <= s0 + G before B(s0)) and issues step s + P into the stage of step s + P - kS <= s0,
    // read before B(s0): P = kS - G + 1.
    auto scan_db = [&](auto live_tag, auto grp_tag) __attribute__((always_inline)) {
        constexpr bool kLive = decltype(live_tag)::value;
        [[maybe_unused]] constexpr int kGrp = decltype(grp_tag)::value ? 1 : 0;
        frag_t fa[8];
#if VQA_RQ_ABLATE & 2
        for (int i = 0; i < 8; ++i) fa[i] = frag_t{(uint32_t)lane * 2654435761u + i, 0x3c003c00u + lane, 0x12345678u * (i + 1), 0x0badcafeu ^ lane};
#endif
        int rd = 0;
        if constexpr (kAsym && VQA_RQ_PRIO != 0) {  // the waves that stall on DMA issue take the pipe whenever they are ready
            if (kGrp) __builtin_amdgcn_s_setprio(VQA_RQ_PRIO);
        }
#define VQA_RQ_READ1(I)                                                                                               \
    do {                                                                                                              \
        if constexpr (bool(VQA_RQ_ABLATE & 2)) asm volatile("" : "+v"(fa[I]));                                        \
        else fa[I] = *reinterpret_cast<const frag_t*>(xb_ + (I) * 16 * kRowBytes);                                    \
    } while (0)
        if (kLive && total > 0) {
            [[maybe_unused]] const char* xb_ = smem + frag_off;
#pragma unroll
            for (int i = 0; i < 8; ++i) VQA_RQ_READ1(i);
            rd = 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave holds the fragments of step 0 before a piece may land in its stage
        for (int u = 0; u < 2 * ntile; ++u) {
            i32x4 acc[8][2];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = i32x4{0, 0, 0, 0};
#ifdef VQA_RQ_STAMPS
            const bool stamp_on = blockIdx.x == kStampWg && u == kStampU;
            unsigned long long stamp_t[16] = {};
#endif
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if constexpr (kLive) {
                    [[maybe_unused]] const char* xb_ = smem + rd * kStepBytes + frag_off;  // stage of step s + 1
                    rd = (rd + 1) & (kS - 1);
                    // Fair shares of the matrix pipe.  The SIMD's arbiter prefers its OLDER wave (group 0): stamped, group 0 ran its four
                    // steps at ~450 cycles each and then sat ~1000 cycles at the group barrier while group 1, starved until then, finished
                    // alone -- and a lone wave only keeps the pipe 57 % busy (its DMA issue and waits have nobody to hide behind).  The two
                    // waves take the higher priority on alternate steps instead: each rushes one step's MFMAs while the other fills the gaps.
                    if constexpr (VQA_RQ_PRIO != 0 && !kAsym) {
                        if (((kt & 1) ^ kGrp) != 0) __builtin_amdgcn_s_setprio(VQA_RQ_PRIO);
                        else __builtin_amdgcn_s_setprio(0);
                    }
#ifdef VQA_RQ_STAMPS
                    VQA_RQ_SB();
                    VQA_RQ_STAMP((kt % G) * 3);
                    VQA_RQ_SB();
#endif
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            if constexpr (bool(VQA_RQ_ABLATE & 4)) asm volatile("" : "+v"(acc[mi][ni]) : "v"(fa[mi]), "v"(qb[kt][ni]));
                            else acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, fa[mi]),
                                                                                    __builtin_bit_cast(i32x4, qb[kt][ni]), acc[mi][ni], 0, 0, 0);
                        }
                        VQA_RQ_READ1(mi);
                        if (mi == VQA_RQ_DMA_AT) {
                            VQA_RQ_STAMP((kt % G) * 3 + 1);
                            issue();
                            VQA_RQ_STAMP((kt % G) * 3 + 2);
                        }
                    }
                } else {
                    issue();
                }
                if ((kt + 1) % G == 0) {
                    if constexpr (kLive && bool(VQA_RQ_ABLATE & 8)) {
                        if (kt + 1 == KT) {
#pragma unroll
                            for (int mi = 0; mi < 8; ++mi) asm volatile("" ::"v"(acc[mi][0]), "v"(acc[mi][1]));
                        }
                    }
#ifdef VQA_RQ_STAMPS
                    VQA_RQ_SB();
                    VQA_RQ_STAMP(12);
#endif
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wait_vmcnt<kWaitPieces>();
                    VQA_RQ_STAMP(13);
                    VQA_RQ_BARRIER();
#ifdef VQA_RQ_STAMPS
                    VQA_RQ_STAMP(14);
                    if (stamp_on) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        unsigned long long* sp = g_rq_stamps + ((size_t)wave * 3 + kt / G) * 16;
#pragma unroll
                        for (int j = 0; j < 15; ++j) asm volatile("s_store_dwordx2 %1, %0, %2" ::"s"(sp), "s"(stamp_t[j]), "n"(j * 8) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
#endif
                }
            }
            if constexpr (kLive && !(VQA_RQ_ABLATE & 8)) half_tile_epilogue(acc, u);
        }
#undef VQA_RQ_READ1
    };
    if constexpr (kDB) {
        if (grp == 0) {
            if (live_wave) scan_db(std::true_type{}, std::false_type{});
            else scan_db(std::false_type{}, std::false_type{});
        } else {
            if (live_wave) scan_db(std::true_type{}, std::true_type{});
            else scan_db(std::false_type{}, std::true_type{});
        }
    } else if (grp == 0) {
        if (live_wave) scan(std::true_type{}, std::true_type{});
        else scan(std::false_type{}, std::true_type{});
    } else {
        if (live_wave) scan(std::true_type{}, std::false_type{});
        else scan(std::false_type{}, std::false_type{});
    }
#ifdef VQA_RQ_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
    wait_vmcnt<0>();  // pieces issued past the end of the stream land before the workgroup's LDS is released
    __syncthreads();
    if (tid == 0) {
        const int cnt = *sk_cnt;
        sk.counts[blockIdx.x] = (unsigned)(cnt < sk.cap ? cnt : sk.cap);
        if (cnt > sk.cap) atomicExch(sk.overflow, 1);
    }
}

}  // namespace

bool vqa_sketch_regq_applies(const ScoreTopkArgs& a) {
    if (!a.sketch || a.sketch->beta) return false;
    const int KT = a.d_pad / kRowBytes;
    if (a.d_pad % (2 * kRowBytes) != 0 || !(KT == 12 || KT == 6)) return false;
    const int per_wg = (a.tile_end - a.tile_begin + a.grid - 1) / a.grid;
    return per_wg <= kMaxTiles;
}

template <int KT>
static int launch_kt(const ScoreTopkArgs& a, hipStream_t stream) {
    static VqaPerDeviceOnce once;
    int rc = once.run([&](int) -> int {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sketch_scan_regq_kernel<KT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal));
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sketch_scan_regq_kernel<KT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal));
        return VQA_OK;
    });
    if (rc != VQA_OK) return rc;
    auto kern = a.first_stage ? sketch_scan_regq_kernel<KT, 1> : sketch_scan_regq_kernel<KT, 0>;
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(kThreads), kLdsTotal, stream, a.x, a.q, (long long)a.n, a.nq, a.tile_begin, a.tile_end, a.gate, *a.sketch);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_sketch_regq(const ScoreTopkArgs& a, hipStream_t stream) {
    VQA_REQUIRE(vqa_sketch_regq_applies(a), "sketch_regq: launch outside the kernel's shapes");
    return a.d_pad / kRowBytes == 12 ? launch_kt<12>(a, stream) : launch_kt<6>(a, stream);
}

#ifdef VQA_RQ_STAMPS
extern "C" int vqa_debug_read_rq_stamps(unsigned long long* out, int n) {
    const size_t bytes = sizeof(unsigned long long) * (size_t)(n < 8 * 3 * 16 ? n : 8 * 3 * 16);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rq_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
