// K4 -- ONE-launch search of a small fp16 shard for a handful of questions, gfx950 only.
//
// The reference asks ONE question per call with limit = 1 against a corpus of a few thousand documents
// (inference_pipeline/db_utils/heavy_ranker.py:97-101).  At that size the general search
// (normalise -> query staging -> seed scan -> threshold merge -> main scan -> list merge: six dependent launches, capi.hip) is all
// launch latency: 0.05 ms of device time for a few MB of rows.  This kernel does the whole call in one launch.  Every workgroup
// (W waves -- 2 up to 16 384 rows, else 4 --, one per 32 W rows of the shard)
//   (1) L2-normalises the raw fp32 questions and rounds them to fp16 -- the arithmetic of normalize_convert_kernel followed by the
//       staging conversion (convert.hip), bit for bit --,
//   (2) scores its rows against them with v_mfma_f32_16x16x32_f16 (A = corpus rows, B = questions, K-steps in ascending order: the
//       exact scan's accumulation chains, so the scores are the exact scan's bits), fragments straight from the shard's tiled layout
//       (a wave's 16 rows of a K-block are 1 KiB contiguous), every load of a 12-K-block chunk in flight before its first MFMA,
//   (3) keeps its k best (score, position) keys per question (k rounds of a wave maximum),
//   (4) publishes them and takes a ticket; the LAST workgroup to arrive merges all lists and writes scores, external ids and
//       positions -- straight into the caller's pinned memory.
// Limits (vqa_tiny_search_applies): fp16 storage, <= 16 questions, k <= 16, questions x k <= 64, <= 131 072 rows.
#include <string.h>

#include "vqa_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTinyQ = 16;      // questions per call (one MFMA column group)
constexpr int kTinyK = 16;      // results per question
constexpr int kTinyMaxResults = 64;  // questions x results per call
constexpr int kTinyChunk = 12;  // K-blocks whose fragments a wave keeps in flight together
constexpr int kTinyMaxUnits = 1024;
constexpr int kTinyArgBytes = 3584;  // of questions inside the launch packet (4 KB of kernel arguments at most)
constexpr int kTinyMaxRows = 131072;
constexpr int kTinySmallRows = 16384;  // up to here: workgroups of 64 rows (a 5000-row shard on 79 CUs instead of 20)

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

template <int W>
__global__ __launch_bounds__(64 * W) void tiny_search_kernel(const _Float16* __restrict__ X, long long n, int KT, int d, const void* __restrict__ q,
                                                             const TinyQArg qa, int q_in_args, int q_is_f16, int normalize, int nq, int k,
                                                             const long long* __restrict__ ids,
                                                             long long id_base, vqa_key* partial, unsigned* ticket,
                                                             float* __restrict__ out_scores, long long* __restrict__ out_ids,
                                                             long long* __restrict__ out_pos) {
    constexpr int RU = 32 * W;  // rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int unit = blockIdx.x, units = gridDim.x;
    const int qstride = KT * 64 + 16;  // bytes per question of the LDS image (+16: the 16 questions of a fragment read start in different banks)
    vqa_key* keys = reinterpret_cast<vqa_key*>(smem + kTinyQ * qstride);  // [16 questions][RU rows]
    __shared__ unsigned last_flag;

    // ---- (0) the first chunk of this wave's rows is requested before anything else: it travels while the questions are prepared
    const long long row0 = (long long)unit * RU + wave * 32;  // first row of the wave; its 32 rows lie in one 256-row tile
    // (addresses as a wave-uniform base per K-block + ONE per-lane byte offset: scalar registers instead of a vector pair per load)
    const int tile = __builtin_amdgcn_readfirstlane((int)(row0 >> 8));
    const char* xb = reinterpret_cast<const char*>(X) + (size_t)tile * KT * 16384;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int r_lo = (int)(row0 & 255) + c;  // the lane's row of the first 16; its row of the second 16 lies 16 x 64 bytes further
    const unsigned voff = (unsigned)(r_lo * 4 + (g ^ (((r_lo >> 3) & 1) * 3))) * 16u;  // slot g of the row inside a 16 KiB K-block (convert.hip: tiled_unit)
    const char* qfrag = smem + (size_t)c * qstride + g * 16;
    half8 a[kTinyChunk][2];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < kTinyChunk; ++j) {
            const int kap = k0 + j < KT ? k0 + j : KT - 1;  // (past the end: the last block again, not multiplied)
            const char* blk = xb + (size_t)kap * 16384;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[j][mi] = *reinterpret_cast<const half8*>(blk + voff + mi * 1024);
        }
    };
    load_chunk(0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- (1) the questions.  Raw bytes -> LDS in 16-byte units (one round trip, whatever the source), then x / ||x|| in fp32 (sum of
    // squares strided over the lanes, xor-shuffle reduction, true division) rounded to fp16; columns past d are zeros.  Questions past
    // nq are left as they are: an MFMA output column depends on ITS question only, and those columns are never read.
    char* raw = smem + kTinyQ * qstride;  // (the keys' space and beyond: lds_bytes)
    {
        const int total = nq * d * (q_is_f16 ? 2 : 4);
        if (q_in_args) {
            for (int u = tid; u * 16 < total; u += 64 * W) reinterpret_cast<uint4*>(raw)[u] = qa.v[u];
        } else if (((reinterpret_cast<uintptr_t>(q) | (uintptr_t)total) & 15) == 0) {
            for (int u = tid; u * 16 < total; u += 64 * W) reinterpret_cast<uint4*>(raw)[u] = reinterpret_cast<const uint4*>(q)[u];
        } else {
#pragma unroll 1
            for (int u = tid; u * 2 < total; u += 64 * W) reinterpret_cast<unsigned short*>(raw)[u] = reinterpret_cast<const unsigned short*>(q)[u];
        }
    }
    __syncthreads();
    for (int qi = wave; qi < nq; qi += W) {
        _Float16* dst = reinterpret_cast<_Float16*>(smem + (size_t)qi * qstride);
        if (q_is_f16) {
            const _Float16* src = reinterpret_cast<const _Float16*>(raw) + (size_t)qi * d;
            for (int j = lane; j < KT * 32; j += 64) dst[j] = j < d ? src[j] : (_Float16)0.f;
        } else {
            const float* src = reinterpret_cast<const float*>(raw) + (size_t)qi * d;
            float nrm = 0.f;
            if (normalize) {
                float ss = 0.f;
                for (int j = lane; j < d; j += 64) ss += src[j] * src[j];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
                nrm = sqrtf(ss);
            }
            for (int j = lane; j < KT * 32; j += 64) {
                float v = 0.f;
                if (j < d) v = nrm > 0.f ? src[j] / nrm : src[j];
                dst[j] = (_Float16)v;
            }
        }
    }
    __syncthreads();

    // ---- (2) scores of this wave's 32 rows x 16 questions
    for (int k0 = 0; k0 < KT; k0 += kTinyChunk) {
        if (k0) load_chunk(k0);
        __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before the first MFMA
#pragma unroll
        for (int j = 0; j < kTinyChunk; ++j) {
            if (k0 + j < KT) {
                const half8 b = *reinterpret_cast<const half8*>(qfrag + (k0 + j) * 64);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j][mi], b, acc[mi], 0, 0, 0);
            }
        }
    }
    // acc[mi][j] = score(row row0 + 16 mi + 4 g + j, question c)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 32 + mi * 16 + g * 4 + j;
            const long long pos = (long long)unit * RU + r;
            keys[c * RU + r] = pos < n ? vqa_make_key(acc[mi][j], (uint32_t)pos) : 0ull;
        }
    __syncthreads();

    // ---- (3) this workgroup's k best per question -> partial[question][slot][unit] (a question's best-of-unit keys side by side)
    for (int qi = wave; qi < nq; qi += W) {
        constexpr int PER = RU / 64;
        vqa_key mine[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) mine[i] = keys[qi * RU + lane + 64 * i];
        vqa_key* dst = partial + (size_t)qi * kTinyK * kTinyMaxUnits + unit;
        wave_topk<PER>(mine, k, [&](int r, vqa_key best) {  // (device-scope stores: past this XCD's L2, which the others do not see)
            if (lane == 0) __hip_atomic_store(dst + (size_t)r * kTinyMaxUnits, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
    }

    // ---- (4) ticket; the last workgroup merges.  The lists are written and read with device-scope accesses and every wave waits for its
    // stores before the barrier in front of the ticket, so no fence is needed: a device-scope release / acquire pair writes back and
    // invalidates the XCD's whole L2, once per workgroup -- 103 us instead of 40 at 131 072 rows.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_flag = atomicAdd(ticket, 1u) == (unsigned)(units - 1) ? 1u : 0u;
    __syncthreads();
    if (!last_flag) return;
    if (tid == 0) *ticket = 0u;  // for the next call on this handle (stream order)
    // Two levels, one wave per question.  A key of the overall top k belongs to a workgroup whose BEST key is one of the k largest
    // best keys (otherwise k keys of other workgroups beat it): select those k workgroups from the <= 1024 best keys (16 per lane, one
    // round of loads), then the answer from their k lists (k * k <= 256 keys, 4 per lane).
    __shared__ int sel[W][kTinyK];
    for (int qi = wave; qi < nq; qi += W) {
        const vqa_key* mine_q = partial + (size_t)qi * kTinyK * kTinyMaxUnits;
        auto emit = [&](int r, vqa_key best) {
            if (lane != 0) return;
            const size_t o = (size_t)qi * k + r;
            if (best == 0ull) {  // fewer than k rows: padding, as the general merge writes it
                out_scores[o] = -INFINITY;
                out_ids[o] = -1;
                if (out_pos) out_pos[o] = -1;
            } else {
                const long long pos = (long long)vqa_key_pos(best);
                out_scores[o] = vqa_key_score(best);
                out_ids[o] = ids ? ids[pos] : id_base + pos;
                if (out_pos) out_pos[o] = pos;
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
        vqa_key mine[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i, u = e < k * k ? sel[wave][e / k] : -1;
            mine[i] = u >= 0 ? __hip_atomic_load(mine_q + (size_t)(e % k) * kTinyMaxUnits + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        wave_topk<4>(mine, k, emit);
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

template <int W>
int launch(int units, size_t lds, hipStream_t stream, const void* rows, int64_t n, int KT, int32_t d, const void* q, const void* q_host, void* q_stage, int32_t q_dtype,
           int32_t normalize, int32_t nq, int32_t k, const int64_t* ids, int64_t id_base, vqa_key* partial, unsigned* ticket, float* out_scores,
           int64_t* out_ids, int64_t* out_pos) {
    TinyQArg qa;
    const size_t qbytes = (size_t)nq * d * (q_dtype == VQA_F16 ? 2 : 4);
    const int q_in_args = q_host && qbytes <= sizeof(qa);
    if (q_in_args) {
        memcpy(&qa, q_host, qbytes);
    } else if (q_host) {
        hipLaunchKernelGGL(stage_questions_kernel, dim3((unsigned)((qbytes + 4095) / 4096)), dim3(256), 0, stream, q, q_stage, (int)qbytes);
        q = q_stage;
    }
    if (lds > 64 * 1024) {
        static VqaPerDeviceOnce once;
        int rc = once.run([&](int) -> int {
            VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_search_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            return VQA_OK;
        });
        if (rc != VQA_OK) return rc;
    }
    hipLaunchKernelGGL(tiny_search_kernel<W>, dim3(units), dim3(64 * W), lds, stream, reinterpret_cast<const _Float16*>(rows), (long long)n, KT, d, q,
                       qa, q_in_args, q_dtype == VQA_F16 ? 1 : 0, normalize, nq, k, reinterpret_cast<const long long*>(ids), (long long)id_base, partial, ticket,
                       out_scores, reinterpret_cast<long long*>(out_ids), reinterpret_cast<long long*>(out_pos));
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

// the fp16 image of 16 questions + the larger of the keys [16][rows per workgroup] and the raw questions (fp32 at most) they overlay
size_t lds_bytes(int32_t d_pad, int rows_per_wg, int nq) {
    const size_t keys = (size_t)kTinyQ * rows_per_wg * sizeof(vqa_key), raw = (size_t)nq * d_pad * 4;
    return (size_t)kTinyQ * (d_pad * 2 + 16) + (keys > raw ? keys : raw);
}

}  // namespace

bool vqa_tiny_search_applies(int32_t dtype, int64_t n, int32_t d_pad, int32_t B, int32_t k) {
    // (B k <= 64: every selection round is a wave-wide maximum, a question's rounds run in ONE wave, and past that the general path's
    // block-wide selections are level or ahead -- 16 questions x 16 results: 122-194 us here, 120-170 us there)
    return dtype == VQA_F16 && n >= 1 && n <= kTinyMaxRows && B >= 1 && B <= kTinyQ && k >= 1 && k <= kTinyK && B * k <= kTinyMaxResults &&
           lds_bytes(d_pad, 128, kTinyQ) <= 160 * 1024 - 1024;
}

size_t vqa_tiny_search_workspace_bytes() { return (size_t)kTinyMaxUnits * kTinyQ * kTinyK * sizeof(vqa_key) + 64; }

// workspace: [16][16][kTinyMaxUnits] keys, then the ticket (zero before the first call; the kernel leaves it zero).  q: the questions
// as the device sees them; q_host: the same bytes in host memory or nullptr (questions that live on the device); q_stage: device memory for
// them (host questions larger than the launch packet's share are copied there by a kernel in front)
int vqa_launch_tiny_search(const void* rows_tiled, int64_t n, int32_t d, int32_t d_pad, const void* q, const void* q_host, void* q_stage, int32_t q_dtype,
                           int32_t normalize, int32_t nq, int32_t k, const int64_t* ids, int64_t id_base, void* workspace, float* out_scores,
                           int64_t* out_ids, int64_t* out_pos, hipStream_t stream) {
    const int KT = d_pad * 2 / 64;
    vqa_key* partial = static_cast<vqa_key*>(workspace);
    unsigned* ticket = reinterpret_cast<unsigned*>(partial + (size_t)kTinyMaxUnits * kTinyQ * kTinyK);
    if (n <= kTinySmallRows && nq <= 4)  // (more questions: twice the waves to share their normalisation and selection rounds)
        return launch<2>((int)((n + 63) / 64), lds_bytes(d_pad, 64, nq), stream, rows_tiled, n, KT, d, q, q_host, q_stage, q_dtype, normalize, nq, k, ids, id_base,
                         partial, ticket, out_scores, out_ids, out_pos);
    return launch<4>((int)((n + 127) / 128), lds_bytes(d_pad, 128, nq), stream, rows_tiled, n, KT, d, q, q_host, q_stage, q_dtype, normalize, nq, k, ids, id_base,
                     partial, ticket, out_scores, out_ids, out_pos);
}
