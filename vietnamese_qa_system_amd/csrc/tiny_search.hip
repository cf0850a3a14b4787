// K4 -- ONE-launch search of a small fp16 / fp32 shard for a handful of questions, gfx950 only.
//
// The reference asks ONE question per call with limit = 1 against a corpus of a few thousand documents
// (inference_pipeline/db_utils/heavy_ranker.py:97-101).  At that size the general search
// (normalise -> query staging -> seed scan -> threshold merge -> main scan -> list merge: six dependent launches, capi.hip) is all
// launch latency: 0.05 ms of device time for a few MB of rows.  This kernel does the whole call in one launch.  Every workgroup
// (W waves of RG row groups of 16: 64 rows of a shard of up to 16 384 rows, 128 up to 131 072, else 256)
//   (1) L2-normalises the raw fp32 questions and converts them to the storage type -- the arithmetic of normalize_convert_kernel
//       followed by the staging conversion (convert.hip), bit for bit --,
//   (2) scores its rows against them on the exact scan's MFMA (fp16: one v_mfma_f32_16x16x32_f16 per K-block; fp32: four
//       v_mfma_f32_16x16x4_f32, the lane's float t of its fragment in sub-step t) -- A = corpus rows, B = questions, K-blocks and
//       sub-steps in ascending order: the exact scan's accumulation chains (score_topk.hip MfmaTraits / mma_block), hence its bits --,
//       fragments straight from the shard's tiled layout (a wave's 16 rows of a K-block are 1 KiB contiguous), every load of a chunk
//       of K-blocks in flight before its first MFMA, the first chunk requested before the questions are touched,
//   (3) keeps its k best (score, position) keys per question (k rounds of a wave maximum),
//   (4) publishes them and takes a ticket; the LAST workgroup to arrive merges all lists and writes scores, external ids and
//       positions -- straight into the caller's pinned memory.
// Limits (vqa_tiny_search_applies): fp16 / fp32 storage, <= 16 questions, k <= 32, questions x k <= 64, <= 262 144 rows.
#include <string.h>

#include <type_traits>

#include "vqa_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int frag_t __attribute__((ext_vector_type(4)));  // 16 bytes of one row: 8 fp16 | 4 fp32

constexpr int kTinyQ = 16;      // questions per call (one MFMA column group)
constexpr int kTinyK = 32;      // results per question
constexpr int kTinyMaxResults = 64;  // questions x results per call
constexpr int kTinyChunk = 12;  // K-blocks whose fragments a wave keeps in flight together (fp16, RG = 2: 96 registers; all 24 of a 768-element row at once: measured equal)
constexpr int kTinyMaxUnits = 1024;
constexpr int kTinyArgBytes = 3584;  // of questions inside the launch packet (4 KB of kernel arguments at most)
constexpr int kTinyMaxRows = 262144;
constexpr int kTinyMidRows = 131072;   // up to here: workgroups of 128 rows; beyond: 256 (the list merge takes <= 1024 workgroups)
constexpr int kTinySmallRows = 16384;  // up to here: workgroups of 64 rows (a 5000-row shard on 79 CUs instead of 40)

// the k largest of the keys a wave holds (PER per lane; real keys are distinct, 0 = empty), largest first
template <int PER, typename EMIT>
__device__ __forceinline__ void wave_topk(vqa_key (&mine)[PER], int k, EMIT&& emit) {
    for (int r = 0; r < k; ++r) {
        vqa_key best = 0ull;
#pragma unroll
        for (int i = 0; i < PER; ++i) best = mine[i] > best ? mine[i] : best;
        best = vqa_wave_max_key(best);
        if (best != 0ull) {
#pragma unroll
            for (int i = 0; i < PER; ++i) mine[i] = mine[i] == best ? 0ull : mine[i];  // exactly one lane holds it
        }
        emit(r, best);
    }
}

// the questions of a call that fit travel as a kernel argument: the launch packet is written into device memory by the host, while a
// pinned host buffer is read over the bus by EVERY workgroup (measured: normalize_convert_kernel on one mapped 3 KB question, 20 us)
struct TinyQArg {
    uint4 v[kTinyArgBytes / 16];
};

struct TinyArgs {
    const void* x;  // the shard's rows, tiled (convert.hip)
    long long n;
    int KT, d;      // K-blocks of 64 bytes per row; elements per row
    const void* q;  // questions [nq, d] fp32 / fp16 as the device sees them (unused when they travel in the launch packet)
    int q_in_args, q_is_f16, normalize, nq, k;
    const long long* ids;
    long long id_base;
    vqa_key* partial;  // [questions x k <= 64 lists][kTinyMaxUnits]: list qi k + r holds every workgroup's r-th best key of question qi
    unsigned* ticket;
    float* out_scores;
    long long* out_ids;
    long long* out_pos;
};

// one K-block of one 16 x 16 accumulator: the exact scan's instruction(s) on the lane's 16-byte fragments
template <int DT>
__device__ __forceinline__ f32x4 mma_kblock(frag_t a, frag_t b, f32x4 c) {
    if constexpr (DT == VQA_F16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    } else {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
    }
}

template <int W, int RG, int DT>
__global__ __launch_bounds__(64 * W) void tiny_search_kernel(const TinyArgs p, const TinyQArg qa) {
    constexpr int RW = 16 * RG;          // rows per wave
    constexpr int RU = RW * W;           // rows per workgroup
    constexpr int CH = kTinyChunk * 2 / RG;  // K-blocks per chunk: 96 registers of fragments whatever RG
    typedef typename std::conditional<DT == VQA_F16, _Float16, float>::type store_t;
    static_assert(RU % 64 == 0 && 256 % RU == 0, "a workgroup's rows: whole 64-key groups inside one 256-row tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int unit = blockIdx.x, units = gridDim.x;
    const int KT = p.KT, d = p.d, nq = p.nq, k = p.k;
    const int qstride = KT * 64 + 16;  // bytes per question of the LDS image (+16: the 16 questions of a fragment read start in different banks)
    vqa_key* keys = reinterpret_cast<vqa_key*>(smem + kTinyQ * qstride);  // [16 questions][RU rows]
    __shared__ unsigned last_flag;
    __shared__ int sel[W][kTinyK];

    // ---- (0) the first chunk of this wave's rows is requested before anything else: it travels while the questions are prepared
    // (addresses as a wave-uniform base per K-block + ONE per-lane byte offset: scalar registers instead of a vector pair per load)
    const long long row0 = (long long)unit * RU + wave * RW;  // first row of the wave; its rows lie in one 256-row tile
    const int tile = __builtin_amdgcn_readfirstlane((int)(row0 >> 8));
    const char* xb = reinterpret_cast<const char*>(p.x) + (size_t)tile * KT * 16384;
    f32x4 acc[RG];
#pragma unroll
    for (int mi = 0; mi < RG; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r_lo = (int)(row0 & 255) + c;  // the lane's row of the first 16; its row of the next 16 lies 16 x 64 bytes further
    const unsigned voff = (unsigned)(r_lo * 4 + (g ^ (((r_lo >> 3) & 1) * 3))) * 16u;  // slot g of the row inside a 16 KiB K-block (convert.hip: tiled_unit)
    const char* qfrag = smem + (size_t)c * qstride + g * 16;
    frag_t a[CH][RG];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kap = k0 + j < KT ? k0 + j : KT - 1;  // (past the end: the last block again, not multiplied)
            const char* blk = xb + (size_t)kap * 16384;
#pragma unroll
            for (int mi = 0; mi < RG; ++mi) a[j][mi] = *reinterpret_cast<const frag_t*>(blk + voff + mi * 1024);
        }
    };
    load_chunk(0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- (1) the questions.  Raw bytes -> LDS in 16-byte units (one round trip, whatever the source), then x / ||x|| in fp32 (sum of
    // squares strided over the lanes, xor-shuffle reduction, true division) converted to the storage type; columns past d are zeros.
    // Questions past nq are left as they are: an MFMA output column depends on ITS question only, and those columns are never read.
    char* raw = smem + kTinyQ * qstride;  // (the keys' space and beyond: lds_bytes)
    {
        const int total = nq * d * (p.q_is_f16 ? 2 : 4);
        if (p.q_in_args) {
            for (int u = tid; u * 16 < total; u += 64 * W) reinterpret_cast<uint4*>(raw)[u] = qa.v[u];
        } else if (((reinterpret_cast<uintptr_t>(p.q) | (uintptr_t)total) & 15) == 0) {
            for (int u = tid; u * 16 < total; u += 64 * W) reinterpret_cast<uint4*>(raw)[u] = reinterpret_cast<const uint4*>(p.q)[u];
        } else {
#pragma unroll 1
            for (int u = tid; u * 2 < total; u += 64 * W) reinterpret_cast<unsigned short*>(raw)[u] = reinterpret_cast<const unsigned short*>(p.q)[u];
        }
    }
    __syncthreads();
    const int cols = KT * (64 / (int)sizeof(store_t));  // padded row length in elements
    for (int qi = wave; qi < nq; qi += W) {
        store_t* dst = reinterpret_cast<store_t*>(smem + (size_t)qi * qstride);
        if (p.q_is_f16) {
            const _Float16* src = reinterpret_cast<const _Float16*>(raw) + (size_t)qi * d;
            for (int j = lane; j < cols; j += 64) dst[j] = j < d ? (store_t)src[j] : (store_t)0.f;
        } else {
            const float* src = reinterpret_cast<const float*>(raw) + (size_t)qi * d;
            float nrm = 0.f;
            if (p.normalize) {
                float ss = 0.f;
                for (int j = lane; j < d; j += 64) ss += src[j] * src[j];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
                nrm = sqrtf(ss);
            }
            for (int j = lane; j < cols; j += 64) {
                float v = 0.f;
                if (j < d) v = nrm > 0.f ? src[j] / nrm : src[j];
                dst[j] = (store_t)v;
            }
        }
    }
    __syncthreads();

    // ---- (2) scores of this wave's rows x 16 questions
    for (int k0 = 0; k0 < KT; k0 += CH) {
        if (k0) load_chunk(k0);
        __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first MFMA
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (k0 + j < KT) {
                const frag_t b = *reinterpret_cast<const frag_t*>(qfrag + (k0 + j) * 64);
#pragma unroll
                for (int mi = 0; mi < RG; ++mi) acc[mi] = mma_kblock<DT>(a[j][mi], b, acc[mi]);
            }
        }
    }
    // acc[mi][j] = score(row row0 + 16 mi + 4 g + j, question c)
#pragma unroll
    for (int mi = 0; mi < RG; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * RW + mi * 16 + g * 4 + j;
            const long long pos = (long long)unit * RU + r;
            keys[c * RU + r] = pos < p.n ? vqa_make_key(acc[mi][j], (uint32_t)pos) : 0ull;
        }
    __syncthreads();

    // ---- (3) this workgroup's k best per question -> partial[question][slot][unit] (a question's best-of-unit keys side by side)
    for (int qi = wave; qi < nq; qi += W) {
        constexpr int PER = RU / 64;
        vqa_key mine[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) mine[i] = keys[qi * RU + lane + 64 * i];
        vqa_key* dst = p.partial + (size_t)qi * k * kTinyMaxUnits + unit;  // list qi k + r of the call's <= 64
        wave_topk<PER>(mine, k, [&](int r, vqa_key best) {  // (device-scope stores: past this XCD's L2, which the others do not see)
            if (lane == 0) __hip_atomic_store(dst + (size_t)r * kTinyMaxUnits, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
    }

    // ---- (4) ticket; the last workgroup merges.  The lists are written and read with device-scope accesses and every wave waits for its
    // stores before the barrier in front of the ticket, so no fence is needed: a device-scope release / acquire pair writes back and
    // invalidates the XCD's whole L2, once per workgroup -- 103 us instead of 40 at 131 072 rows.
    // What this relies on (gfx942 / gfx950 only; outside the HIP memory model, so the file refuses other targets below): a relaxed
    // agent-scope atomic store / load of 8 bytes is emitted as `global_store_dwordx2 ... sc1` / `global_load_dwordx2 ... sc1`; an sc1
    // store writes THROUGH this XCD's L2 to the memory side before it retires from vmcnt, and an sc1 load never hits a stale line of
    // the reader's L2 -- the "every store sc1 and drained with s_waitcnt vmcnt(0) before the counter, every load sc1" hand-off that
    // /opt/skills/guides/MI355X_MICROARCH.md lists under Correctness boundaries as valid in place of the release / acquire pair.  The
    // ticket itself is a device-scope atomic at the memory side.  (A launch that aborts between its ticket and the reset below would
    // leave the ticket non-zero: the host side clears the workspace after any failed call, capi.hip.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "tiny_search.hip: the fence-free hand-off is only argued for gfx942 / gfx950"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_flag = atomicAdd(p.ticket, 1u) == (unsigned)(units - 1) ? 1u : 0u;
    __syncthreads();
    if (!last_flag) return;
    if (tid == 0) *p.ticket = 0u;  // for the next call on this handle (stream order)
    // Two levels, one wave per question.  A key of the overall top k belongs to a workgroup whose BEST key is one of the k largest
    // best keys (otherwise k keys of other workgroups beat it): select those k workgroups from the <= 1024 best keys (16 per lane, one
    // round of loads), then the answer from their k lists (k * k keys: 4 per lane up to k = 16, else 16).
    for (int qi = wave; qi < nq; qi += W) {
        const vqa_key* mine_q = p.partial + (size_t)qi * k * kTinyMaxUnits;
        auto emit = [&](int r, vqa_key best) {
            if (lane != 0) return;
            const size_t o = (size_t)qi * k + r;
            if (best == 0ull) {  // fewer than k rows: padding, as the general merge writes it
                p.out_scores[o] = -INFINITY;
                p.out_ids[o] = -1;
                if (p.out_pos) p.out_pos[o] = -1;
            } else {
                const long long pos = (long long)vqa_key_pos(best);
                p.out_scores[o] = vqa_key_score(best);
                p.out_ids[o] = p.ids ? p.ids[pos] : p.id_base + pos;
                if (p.out_pos) p.out_pos[o] = pos;
            }
        };
        vqa_key head[kTinyMaxUnits / 64];
#pragma unroll
        for (int i = 0; i < kTinyMaxUnits / 64; ++i)
            head[i] = lane + 64 * i < units ? __hip_atomic_load(mine_q + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        if (k == 1) {
            wave_topk<kTinyMaxUnits / 64>(head, 1, emit);
            continue;
        }
        wave_topk<kTinyMaxUnits / 64>(head, k, [&](int r, vqa_key best) {
            if (lane == 0) sel[wave][r] = best ? (int)(vqa_key_pos(best) / RU) : -1;
        });
        __builtin_amdgcn_wave_barrier();
        auto second_level = [&](auto per) {
            constexpr int PER2 = decltype(per)::value;
            vqa_key mine[PER2];
#pragma unroll
            for (int i = 0; i < PER2; ++i) {
                const int e = lane + 64 * i, u = e < k * k ? sel[wave][e / k] : -1;
                mine[i] = u >= 0 ? __hip_atomic_load(mine_q + (size_t)(e % k) * kTinyMaxUnits + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
            wave_topk<PER2>(mine, k, emit);
        };
        if (k <= 16) second_level(std::integral_constant<int, 4>{});
        else second_level(std::integral_constant<int, 16>{});
        __builtin_amdgcn_wave_barrier();
    }
}

// questions in pinned host memory that do not fit the launch packet: ONE wide read of them over the bus into device memory, in front
// of the search (instead of one per workgroup)
__global__ __launch_bounds__(256) void stage_questions_kernel(const void* __restrict__ src, void* __restrict__ dst, int bytes) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src) | (uintptr_t)bytes) & 15) == 0) {
        if (u * 16 < bytes) reinterpret_cast<uint4*>(dst)[u] = reinterpret_cast<const uint4*>(src)[u];
    } else {
        for (int i = u * 8; i < u * 8 + 8 && i * 2 < bytes; ++i) reinterpret_cast<unsigned short*>(dst)[i] = reinterpret_cast<const unsigned short*>(src)[i];
    }
}

// the storage-type image of 16 questions + the larger of the keys [16][rows per workgroup] and the raw questions (fp32 at most) they overlay
size_t lds_bytes(int32_t row_bytes, int32_t d_pad, int rows_per_wg, int nq) {
    const size_t keys = (size_t)kTinyQ * rows_per_wg * sizeof(vqa_key), raw = (size_t)nq * d_pad * 4;
    return (size_t)kTinyQ * (row_bytes + 16) + (keys > raw ? keys : raw);
}
constexpr size_t kTinyLdsMax = 160 * 1024 - 2048;  // (the kernel's static LDS: a flag and the merge's selections)

template <int W, int RG, int DT>
int launch(const TinyArgs& p, const TinyQArg& qa, int32_t d_pad, hipStream_t stream) {
    constexpr int RU = 16 * RG * W;
    const size_t lds = lds_bytes(p.KT * 64, d_pad, RU, p.nq);
    if (lds > 64 * 1024) {
        static VqaPerDeviceOnce once;
        int rc = once.run([&](int) -> int {
            VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_search_kernel<W, RG, DT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTinyLdsMax));
            return VQA_OK;
        });
        if (rc != VQA_OK) return rc;
    }
    hipLaunchKernelGGL((tiny_search_kernel<W, RG, DT>), dim3((unsigned)((p.n + RU - 1) / RU)), dim3(64 * W), lds, stream, p, qa);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

}  // namespace

bool vqa_tiny_search_applies(int32_t dtype, int64_t n, int32_t d_pad, int32_t B, int32_t k) {
    // (B k <= 64: every selection round is a wave-wide maximum, a question's rounds run in ONE wave, and past that the general path's
    // block-wide selections are level or ahead -- 16 questions x 16 results: 122-194 us here, 120-170 us there)
    return (dtype == VQA_F16 || dtype == VQA_F32) && n >= 1 && n <= kTinyMaxRows && B >= 1 && B <= kTinyQ && k >= 1 && k <= kTinyK &&
           B * k <= kTinyMaxResults && lds_bytes(d_pad * (dtype == VQA_F16 ? 2 : 4), d_pad, 256, kTinyQ) <= kTinyLdsMax;
}

size_t vqa_tiny_search_workspace_bytes() { return (size_t)kTinyMaxUnits * kTinyMaxResults * sizeof(vqa_key) + 64; }

// workspace: [64 lists][kTinyMaxUnits] keys, then the ticket (zero before the first call; the kernel leaves it zero).  q: the questions
// as the device sees them; q_host: the same bytes in host memory or nullptr (questions that live on the device); q_stage: device memory for
// them (host questions larger than the launch packet's share are copied there by a kernel in front)
int vqa_launch_tiny_search(const void* rows_tiled, int32_t dtype, int64_t n, int32_t d, int32_t d_pad, const void* q, const void* q_host, void* q_stage,
                           int32_t q_dtype, int32_t normalize, int32_t nq, int32_t k, const int64_t* ids, int64_t id_base, void* workspace,
                           float* out_scores, int64_t* out_ids, int64_t* out_pos, hipStream_t stream) {
    TinyArgs p;
    TinyQArg qa;
    const size_t qbytes = (size_t)nq * d * (q_dtype == VQA_F16 ? 2 : 4);
    p.q_in_args = q_host && qbytes <= sizeof(qa);
    if (p.q_in_args) {
        memcpy(&qa, q_host, qbytes);
    } else if (q_host) {
        hipLaunchKernelGGL(stage_questions_kernel, dim3((unsigned)((qbytes + 4095) / 4096)), dim3(256), 0, stream, q, q_stage, (int)qbytes);
        q = q_stage;
    }
    p.x = rows_tiled;
    p.n = n;
    p.KT = d_pad * (dtype == VQA_F16 ? 2 : 4) / 64;
    p.d = d;
    p.q = q;
    p.q_is_f16 = q_dtype == VQA_F16;
    p.normalize = normalize;
    p.nq = nq;
    p.k = k;
    p.ids = reinterpret_cast<const long long*>(ids);
    p.id_base = id_base;
    p.partial = static_cast<vqa_key*>(workspace);
    p.ticket = reinterpret_cast<unsigned*>(p.partial + (size_t)kTinyMaxUnits * kTinyMaxResults);
    p.out_scores = out_scores;
    p.out_ids = reinterpret_cast<long long*>(out_ids);
    p.out_pos = reinterpret_cast<long long*>(out_pos);
    const bool small = n <= kTinySmallRows;
    if (dtype == VQA_F16) {
        // (64 rows as 2 waves x 32; more than 4 questions: twice the waves to share their normalisation and selection rounds)
        if (small && nq <= 4) return launch<2, 2, VQA_F16>(p, qa, d_pad, stream);
        if (n <= kTinyMidRows) return launch<4, 2, VQA_F16>(p, qa, d_pad, stream);
        return launch<8, 2, VQA_F16>(p, qa, d_pad, stream);
    }
    // fp32: four MFMAs per K-block of 16 elements at 1/16 of the fp16 rate per element -- 16 rows per wave keep a wave's matrix time at 2.6 us (768 columns)
    if (small) return launch<4, 1, VQA_F32>(p, qa, d_pad, stream);
    if (n <= kTinyMidRows) return launch<8, 1, VQA_F32>(p, qa, d_pad, stream);
    return launch<8, 2, VQA_F32>(p, qa, d_pad, stream);
}
