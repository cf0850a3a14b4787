// K3 -- question-encoder forward (RoBERTa / PhoBERT-base-shaped post-LN transformer), gfx950 only.
//
// Replaces the transformer forward + pooling + L2-normalise that txtai runs for every query inside
// `embeddings.search` (reference call sites inference_pipeline/db_utils/heavy_ranker.py:98-101; model chosen by `path=`
// at :80,83; DPR form at src/test.py:84-86 `q_model(input_ids).pooler_output`).  Algorithm = HF RoBERTa
// (modeling_roberta.py: embeddings with pad-offset position ids, post-LN layers, erf GELU, LN eps from the config).
//
// Kernels (all hand-written, fp16 storage, fp32 accumulation / statistics):
//   embed_ln        word + position + type embeddings -> LayerNorm                      (one wave per token)
//   gemm_tile<..>   C[M,N] = A[M,K] . W[N,K]^T + bias (+ erf GELU | + residual row) for > 320 tokens: persistent, LDS-DMA ring,
//                   anti-phase slot K loop, tile shape picked per problem (256x288 / 256x192 / 256x128 / 128x192), 16-byte stores
//                   LayerNorms folded into the GEMMs (FoldArgs); the erf GELU of the FFN1 epilogue by interpolation in an LDS table
//   gemm_tiny<..>   <= 64 tokens (ONE question: the reference's own call, heavy_ranker.py:97-101): every load of a wave in flight before its
//                   first MFMA, LayerNorms folded on raw rows, five launches per layer
//   gemm_skinny     65 .. 320 tokens (small batches): weights streamed once, fragments straight from global memory
//   gemm_nt<EPI>    the shapes in between: v_mfma_f32_16x16x32_f16, 128x128 tiles, register-staged double-buffered LDS image
//   attention       per (sequence, head): softmax(q k^T / sqrt(dh) + mask) v, K/V in LDS, wavefront-shuffle softmax
//   ln              LayerNorm (the residual add is fused into the producing GEMM's epilogue) (one wave per token)
//   pool_normalize  CLS row or masked mean -> fp32 -> x / ||x||                          (one wave per sequence)
#include <string.h>

#include <atomic>
#include <new>
#include <type_traits>
#include <vector>

#include "vqa_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- LayerNorm of one row held as kChunks x 8 values per lane: chunk i of a lane covers elements
// (lane + 64 i) * 8 .. + 7 (16-byte fp16 / 32-byte fp32 accesses; H % 8 == 0).  Slots past H hold 0.  All loops are fully
// unrolled so x[][] stays in registers.
constexpr int kChunks = 4;  // hidden <= 2048

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void row_layer_norm(float (&x)[kChunks][8], int H, int lane, const float* __restrict__ g,
                                               const float* __restrict__ b, float eps, _Float16* __restrict__ out) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kChunks; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[i][e];
    const float mu = wave_sum(s) / H;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kChunks; ++i) {
        const bool live = (lane + 64 * i) * 8 < H;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dlt = live ? x[i][e] - mu : 0.f;
            v += dlt * dlt;
        }
    }
    const float rstd = rsqrtf(wave_sum(v) / H + eps);
#pragma unroll
    for (int i = 0; i < kChunks; ++i) {
        const int j0 = (lane + 64 * i) * 8;
        if (j0 < H) {
            const f32x4v g0 = *reinterpret_cast<const f32x4v*>(g + j0), g1 = *reinterpret_cast<const f32x4v*>(g + j0 + 4);
            const f32x4v b0 = *reinterpret_cast<const f32x4v*>(b + j0), b1 = *reinterpret_cast<const f32x4v*>(b + j0 + 4);
            half8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (_Float16)((x[i][e] - mu) * rstd * g0[e] + b0[e]);
                o[4 + e] = (_Float16)((x[i][4 + e] - mu) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<half8*>(out + j0) = o;
        }
    }
}

// LayerNorm folded into the GEMMs (FoldArgs further down): a row's statistics are (sum, sum of squares) slots, P of 16 in use.
// Layout [slot][row] (row stride = the padded row count): the 16 rows a wave instruction covers are 128 contiguous bytes, for
// the GEMM that writes a slot and for the one that reads it.  ([row][slot], 8-byte pieces of a row's line written by 16 waves
// of 4 workgroups, cost 4 us per producing GEMM and 2 us per consuming one.)
constexpr int kRowStatSlots = 16;

// mean and 1 / std of a row from its slots; the whole wave calls (lanes < P load one slot each)
__device__ __forceinline__ void row_stats(const float2* __restrict__ stats, size_t stride, size_t row, int p, int lane, float inv_h,
                                          float eps, float& mean, float& rstd) {
    float2 v = make_float2(0.f, 0.f);
    if (lane < p) v = stats[lane * stride + row];
    const float sm = wave_sum(v.x), sq = wave_sum(v.y);
    mean = sm * inv_h;
    rstd = rsqrtf(fmaxf(sq * inv_h - mean * mean, 0.f) + eps);
}

// the raw row (fp16) and its statistics, of the values as stored, in slot 0 (slots 1 .. 3 cleared: P = 4)
__device__ __forceinline__ void row_raw_stats(float (&x)[kChunks][8], int H, int lane, _Float16* __restrict__ out,
                                              float2* __restrict__ stats, size_t stride, size_t row) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < kChunks; ++i) {
        const int j0 = (lane + 64 * i) * 8;
        if (j0 < H) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (_Float16)x[i][e];
                const float q = (float)o[e];
                sm += q;
                sq += q * q;
            }
            *reinterpret_cast<half8*>(out + j0) = o;
        }
    }
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    if (lane < 4) stats[lane * stride + row] = lane == 0 ? make_float2(sm, sq) : make_float2(0.f, 0.f);
}

// ---- sequence packing (round 2): a ragged batch (questions of 8-32 tokens padded to L) spends a third of every GEMM on padding
// rows.  With right-padded masks the real tokens of sequence b are l = 0 .. n_b - 1; they are stored at packed rows
// cu[b] + l and every kernel below takes its per-sequence length from cu (cu == nullptr: the padded [B, L] layout).
// Padding keys carry exactly zero attention weight in the padded form (HF adds finfo.min), so the real tokens' results are
// identical.  One workgroup: per-sequence counts -> exclusive scan -> cu[0 .. B], row_seq[packed row] = b.  A mask that is not
// right-padded, or a count of real tokens that differs from what the caller announced (more: rows would be dropped; fewer: the
// GEMMs would run on stale workspace rows past cu[B]), raises the host-visible flag (the next forward fails).
__global__ __launch_bounds__(256) void pack_kernel(const int* __restrict__ mask, int B, int L, int announced, int* __restrict__ cu,
                                                   int* __restrict__ row_seq, int* __restrict__ bad) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256, b0 = tid * per, b1 = b0 + per < B ? b0 + per : B;
    int sum = 0;
    bool ragged_ok = true;
    for (int b = b0; b < b1; ++b) {
        int n = 0;
        for (int l = 0; l < L; ++l) {
            const int m = mask[b * L + l] != 0;
            ragged_ok &= !(m && n != l);  // a real token after a padding one
            n += m;
        }
        ragged_ok &= n >= 1;
        sum += n;
    }
    part[tid] = sum;
    __syncthreads();
    for (int step = 1; step < 256; step <<= 1) {  // inclusive scan of the 256 partial sums
        const int add = tid >= step ? part[tid - step] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    if (tid == 255) {
        cu[B] = part[255];
        if (part[255] != announced) __hip_atomic_store(bad, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!ragged_ok) __hip_atomic_store(bad, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int off = part[tid] - sum;  // exclusive
    for (int b = b0; b < b1; ++b) {
        cu[b] = off;
        int n = 0;
        for (int l = 0; l < L; ++l) n += mask[b * L + l] != 0;
        for (int l = 0; l < n; ++l)
            if (off + l < announced) row_seq[off + l] = b;
        off += n;
    }
}

// Loads of activations another workgroup of the SAME launch wrote (the persistent one-question forward): relaxed device-scope atomic
// loads, i.e. `global_load ... sc1` -- never served from this CU's vector L1 or a stale L2 line (the hand-off form tiny_search.hip uses;
// /opt/skills/guides/MI355X_MICROARCH.md, Correctness boundaries).  8 bytes per instruction; COH = false: the plain load.
template <bool COH>
__device__ __forceinline__ half8 act_load16(const _Float16* p) {
    if constexpr (!COH) {
        return *reinterpret_cast<const half8*>(p);
    } else {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 v;
        v[0] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[1] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_bit_cast(half8, v);
    }
}
template <bool COH, typename T8>  // an 8-byte value (4 halves, a float2)
__device__ __forceinline__ T8 act_load8(const void* p) {
    if constexpr (!COH) {
        return *reinterpret_cast<const T8*>(p);
    } else {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_bit_cast(T8, v);
    }
}
template <bool COH>
__device__ __forceinline__ float act_load_half(const _Float16* p) {
    if constexpr (!COH) {
        return (float)*p;
    } else {
        const unsigned short v = __hip_atomic_load(reinterpret_cast<const unsigned short*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (float)__builtin_bit_cast(_Float16, v);
    }
}

// (bodies take their block / thread index as arguments: the persistent one-question forward, encoder_persist_kernel, calls them per unit)
__device__ __forceinline__ void embed_ln_body(const int* __restrict__ ids, int T, int L, int H, int pad_id, int vocab, int abs_pos,
                                              int* __restrict__ bad_ids, const int* __restrict__ cu,
                                              const int* __restrict__ row_seq, int B, const float* __restrict__ word,
                                              const float* __restrict__ pos,
                                              const float* __restrict__ type0, const float* __restrict__ g,
                                              const float* __restrict__ b, float eps, _Float16* __restrict__ out,
                                              float2* __restrict__ stats, int st_stride, int bx, int tid) {
    const int lane = tid & 63;
    const int t = bx * 4 + (tid >> 6);
    if (t >= T) return;
    int seq, l;
    if (cu) {  // packed rows: t = cu[seq] + l
        if (t >= cu[B]) return;
        seq = row_seq[t];
        l = t - cu[seq];
    } else {
        seq = t / L;
        l = t - seq * L;
    }
    // a token id outside the embedding table (tokenizer / vocabulary mismatch) must not become an out-of-bounds read:
    // it is embedded as the pad token and reported through the host-visible flag (checked by the next forward call)
    const int id_raw = ids[seq * L + l];
    const int id = (unsigned)id_raw < (unsigned)vocab ? id_raw : pad_id;
    if (id != id_raw && lane == 0) __hip_atomic_store(bad_ids, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // RoBERTa position id: pad + (number of non-pad tokens up to and including this one), pad tokens keep pad;
    // BERT (VQA_POS_ABSOLUTE: the MiniLM the reference loads): the token's index in its sequence
    int pid = l;
    if (!abs_pos) {
        int cnt = 0;
        for (int j = lane; j <= l; j += 64) cnt += ids[seq * L + j] != pad_id;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        pid = id != pad_id ? cnt + pad_id : pad_id;
    }
    float x[kChunks][8];
#pragma unroll
    for (int i = 0; i < kChunks; ++i) {
        const int j0 = (lane + 64 * i) * 8;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (j0 < H)
                v = *reinterpret_cast<const f32x4v*>(word + (size_t)id * H + j0 + 4 * hlf) +
                    *reinterpret_cast<const f32x4v*>(pos + (size_t)pid * H + j0 + 4 * hlf) +
                    *reinterpret_cast<const f32x4v*>(type0 + j0 + 4 * hlf);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[i][4 * hlf + e] = v[e];
        }
    }
    // stats != nullptr: the LayerNorm is folded into the GEMMs that follow (FoldArgs): raw sum + the row's statistics
    if (stats) row_raw_stats(x, H, lane, out + (size_t)t * H, stats, (size_t)st_stride, (size_t)t);
    else row_layer_norm(x, H, lane, g, b, eps, out + (size_t)t * H);
}
__global__ __launch_bounds__(256) void embed_ln_kernel(const int* __restrict__ ids, int T, int L, int H, int pad_id, int vocab, int abs_pos,
                                                       int* __restrict__ bad_ids, const int* __restrict__ cu,
                                                       const int* __restrict__ row_seq, int B, const float* __restrict__ word,
                                                       const float* __restrict__ pos,
                                                       const float* __restrict__ type0, const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps, _Float16* __restrict__ out,
                                                       float2* __restrict__ stats, int st_stride) {
    embed_ln_body(ids, T, L, H, pad_id, vocab, abs_pos, bad_ids, cu, row_seq, B, word, pos, type0, g, b, eps, out, stats, st_stride, (int)blockIdx.x,
                  (int)threadIdx.x);
}

// LayerNorm alone: the residual add is fused into the epilogue of the GEMM that produced `a` (EPI 2).  (Two or four rows per
// wave, all loaded before the first reduction, measured 1-2 % slower on the whole forward than one row per wave.)
template <bool COH = false>
__device__ __forceinline__ void ln_body(const _Float16* __restrict__ a, int T, int H, const float* __restrict__ g,
                                        const float* __restrict__ b, float eps, _Float16* __restrict__ out, int bx, int tid) {
    const int lane = tid & 63;
    const int t = bx * 4 + (tid >> 6);
    if (t >= T) return;
    float x[kChunks][8];
#pragma unroll
    for (int i = 0; i < kChunks; ++i) {
        const int j0 = (lane + 64 * i) * 8;
        half8 va = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (j0 < H) va = act_load16<COH>(a + (size_t)t * H + j0);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[i][e] = (float)va[e];
    }
    row_layer_norm(x, H, lane, g, b, eps, out + (size_t)t * H);
}
__global__ __launch_bounds__(256) void ln_kernel(const _Float16* __restrict__ a, int T, int H, const float* __restrict__ g,
                                                 const float* __restrict__ b, float eps, _Float16* __restrict__ out) {
    ln_body(a, T, H, g, b, eps, out, (int)blockIdx.x, (int)threadIdx.x);
}

// erf GELU, 0.5 x (1 + erf(x / sqrt 2)), with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16
// output's 5e-4): one v_rcp, one v_exp and a degree-5 Horner chain instead of libm's branchy erff (the FFN1 epilogue
// spent ~25 us per layer in it).
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = 1.0f - p * t * __expf(-z * z);  // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

// erf GELU of the tile GEMM's epilogue (FFN1: 128 values per lane of a 256 x 256 tile): x Phi(x) with Phi by LINEAR INTERPOLATION in a
// table held in LDS -- 7 VALU operations and one ds_read_b64 per value against ~15.6 VALU + v_rcp + v_exp of gelu_erf (the GELUs were
// 6 us of the packed FFN1's 37, VALU-issue-bound: profiles/r04_encoder_gemm_variants.txt).  Table: kGeluN intervals of 1 / 64 over
// [-8, 8), entry i = {Phi(x_i), Phi(x_i + 1/64) - Phi(x_i)} computed in double; interpolation error <= h^2 / 8 max |Phi''| = 7.4e-6
// in Phi, i.e. <= 7.4e-6 |x| in the GELU -- 1 / 50 of the fp16 spacing the result is stored with.  Beyond +-8 the end entries apply
// (Phi = 6e-16 / 1 - 6e-16).
constexpr int kGeluN = 1024;
constexpr int kGeluBytes = kGeluN * 8;
__device__ float2 g_gelu_tab[kGeluN];

__global__ void gelu_table_kernel() {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kGeluN) return;
    const double x0 = (double)i / 64.0 - 8.0, x1 = (double)(i + 1) / 64.0 - 8.0;
    const double p0 = 0.5 * (1.0 + erf(x0 * 0.70710678118654752440)), p1 = 0.5 * (1.0 + erf(x1 * 0.70710678118654752440));
    g_gelu_tab[i] = make_float2((float)p0, (float)(p1 - p0));
}

__device__ __forceinline__ float gelu_tab(float x, const char* __restrict__ tab_lds) {
    float t = __builtin_fmaf(x, 64.0f, 512.0f);
    t = __builtin_amdgcn_fmed3f(t, 0.0f, 1023.99994f);  // (a NaN input comes out as 0 x anything: the GEMM's operands are finite)
    const float fr = __builtin_amdgcn_fractf(t);
    const unsigned idx = (unsigned)t;  // t >= 0: truncation = floor
    const float2 e = *reinterpret_cast<const float2*>(tab_lds + (idx << 3));
    return x * __builtin_fmaf(fr, e.y, e.x);
}

// ---- GEMM: C[M, N] = A[M, K] . W[N, K]^T + bias[N]; EPI 0: identity, 1: erf GELU.  K % 64 == 0 (BK = 64) or K % 32 == 0
// (BK = 32 instantiation for small hidden sizes). -----------------------------------------------------------------------
constexpr int kGemmBM = 128, kGemmBN = 128;

// LDS image of a [128][BK] fp16 tile: 16-byte slot s of row r at r * (2 BK) + ((s ^ f(r)) << 4); conflict-free ds_read_b128
// (BK = 32: 64-B rows, f = 3 * bit3(r); BK = 64: 128-B rows, f = r & 7)
template <int BK>
__device__ __forceinline__ int swz_off(int row, int slot) {
    if (BK == 32) return row * 64 + ((slot ^ (((row >> 3) & 1) * 3)) << 4);
    return row * 128 + ((slot ^ (row & 7)) << 4);
}

template <int EPI, int BK>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                      const float* __restrict__ bias, const _Float16* __restrict__ R,
                                                      _Float16* __restrict__ C, int M, int N, int K) {
    constexpr int kTileBytes = kGemmBM * BK * 2;   // per operand per stage
    constexpr int kSlots = BK / 8;                 // 16-byte slots per row
    constexpr int kLoads = 128 * kSlots / 256;     // staging loads per thread per operand (2 or 4)
    __shared__ __attribute__((aligned(16))) char lds[4 * kTileBytes];  // [stage][A | W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int c = lane & 15, g = lane >> 4;
    const int bm = blockIdx.y * kGemmBM, bn = blockIdx.x * kGemmBN;
    int st_off[kLoads];
    const _Float16* a_src[kLoads];
    const _Float16* w_src[kLoads];
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        const int u = tid + 256 * i;
        const int row = u / kSlots, slot = u % kSlots;
        st_off[i] = swz_off<BK>(row, slot);
        const int ar = bm + row < M ? bm + row : M - 1;  // clamped rows are computed and discarded
        const int wrow = bn + row < N ? bn + row : N - 1;
        a_src[i] = A + (size_t)ar * K + slot * 8;
        w_src[i] = W + (size_t)wrow * K + slot * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int KT = K / BK;
    half8 ra[kLoads], rw[kLoads];
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        ra[i] = *reinterpret_cast<const half8*>(a_src[i]);
        rw[i] = *reinterpret_cast<const half8*>(w_src[i]);
    }
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        *reinterpret_cast<half8*>(lds + st_off[i]) = ra[i];
        *reinterpret_cast<half8*>(lds + kTileBytes + st_off[i]) = rw[i];
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const char* cur = lds + (kt & 1) * 2 * kTileBytes;
        char* nxt = lds + ((kt + 1) & 1) * 2 * kTileBytes;
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
                ra[i] = *reinterpret_cast<const half8*>(a_src[i] + (size_t)(kt + 1) * BK);
                rw[i] = *reinterpret_cast<const half8*>(w_src[i] + (size_t)(kt + 1) * BK);
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            half8 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const half8*>(cur + swz_off<BK>(wr * 64 + mi * 16 + c, kk * 4 + g));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                b[ni] = *reinterpret_cast<const half8*>(cur + kTileBytes + swz_off<BK>(wc * 64 + ni * 16 + c, kk * 4 + g));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
                *reinterpret_cast<half8*>(nxt + st_off[i]) = ra[i];
                *reinterpret_cast<half8*>(nxt + kTileBytes + st_off[i]) = rw[i];
            }
        }
        __syncthreads();
    }
    // epilogue: acc[mi][ni][j] = C[bm + wr*64 + mi*16 + 4g + j][bn + wc*64 + ni*16 + c]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = bn + wc * 64 + ni * 16 + c;
        if (n >= N) continue;
        const float bv = bias[n];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = bm + wr * 64 + mi * 16 + g * 4 + j;
                if (m >= M) continue;
                float v = acc[mi][ni][j] + bv;
                if (EPI == 1) v = gelu_erf(v);
                if (EPI == 2) v += (float)R[(size_t)m * N + n];
                C[(size_t)m * N + n] = (_Float16)v;
            }
    }
}

// ---- GEMM for the large shapes (M > 320 tokens, kTileMinM): persistent workgroups (one per CU, 8 waves), BM x BN output tiles, operands
// go global -> LDS by LDS-DMA in K-steps of 32 halves (64-byte rows, 1 KiB pieces of 16 rows) into a ring of S stages, and the
// K loop runs in ANTI-PHASE SLOTS -- the structure of the scoring kernel's slot loop (score_topk.hip): two barriers per
// K-step; in every slot one wave group only multiplies while the other only moves data (reads the fragments of its next
// K-step, issues its LDS-DMA pieces), so the two waves of a SIMD -- one of each group -- never share the matrix pipe and
// never leave it idle behind each other's memory phases.  One fragment set per wave.
//     K-step kappa, slot 1:  group 0: read fragments(kappa), issue pieces of kappa + D   | group 1: MFMAs of kappa
//                  slot 2:  group 0: MFMAs of kappa                                     | group 1: read fragments(kappa + 1), issue kappa + D
// with D = S - 1: stage kappa + D reuses the stage of kappa - 1 (last read in slot 1 (kappa - 1) / slot 2 (kappa - 2)).
// Pieces are always issued (past the end of the stream: the last K-step again, into a stage nobody reads any more), so every
// wait is a constant vmcnt.  The K-step stream runs on across output tiles (the next tile's first K-steps land under this
// tile's epilogue); XCD-aware tile order.
//
// What bounds it (ablation builds, profiles/r02_encoder_gemm_ablation.txt): with the MFMAs AND the stores removed a 256 x 128
// tile kernel ran at 60 GB/s of L2 -> LDS traffic per CU -- the per-CU LDS-DMA rate (~26 B/clk), not the matrix pipe -- so the
// lever is bytes per flop, (BM + BN) / (BM BN), and tile quantisation on 256 CUs.  The launcher picks, per (M, N), the shape
// with the fewest LDS-DMA bytes per CU: at 8192 tokens 256 x 288 for QKV (N = 2304: 256 tiles, one round), 256 x 192 for FFN1
// (N = 3072: 512 tiles, two rounds), 128 x 192 for the N = 768 projections (256 tiles).
//
// The DMA writes 64 lanes x 16 B contiguously, so the XOR swizzle of the LDS image is applied on the GLOBAL side: lane i of a
// piece fetches row i >> 2, logical 16-byte slot (i & 3) ^ 3 bit3(row); fragment reads use the same involution.  MFMA operand
// roles: A operand = weight rows, B operand = token rows, so a lane's 4 accumulator values are 4 consecutive output features
// of one token.  Token rows past M are read (and their results dropped): the caller's activation buffers are padded to 256 rows.
// EPI 0: + bias; 1: + bias, erf GELU; 2: + bias + residual row (the residual add of the post-LN block, fused here).
#ifndef VQA_GELU_TABLE
#define VQA_GELU_TABLE 1  // 0: the polynomial form in the tile kernel's epilogue too (dev / A-B switch)
#endif
#ifndef VQA_GEMM_ABLATE
#define VQA_GEMM_ABLATE 0  // dev-only timing ablations (wrong results): 1 no epilogue stores, 2 no GELU, 4 no MFMAs in the K loop
#endif
template <int BM, int BN, int BK, int RESV = 0>
struct TileGeom {
    static constexpr int kRowB = 2 * BK;  // bytes of every operand row per K-step (BK = 32 or 64 halves)
    static constexpr int kStageB = (BM + BN) * kRowB;
    static constexpr int kStages = ((160 * 1024 - RESV) / kStageB) < 8 ? ((160 * 1024 - RESV) / kStageB) : 8;
    static constexpr int kRing = kStages * kStageB;
    static constexpr int kLds = kRing + RESV;  // RESV bytes behind the ring: the GELU table of the EPI 1 form
    static constexpr int kPieceRows = 1024 / kRowB;               // rows of one 1 KiB piece: 16 or 8
    static constexpr int kPieces = (BM + BN) / kPieceRows;
    static constexpr int kPerWave = (kPieces + 7) / 8;
    static_assert(BK == 32 || BK == 64, "K-steps of 32 or 64 halves");
    static_assert(kStages >= 3, "the anti-phase ring needs three stages");
};

// N pieces of one wave behind ONE M0 write (the scoring kernel's form): piece j goes to LDS lds_dst + 1024 j (the instruction
// offset moves the LDS side) from base[j] + voff + 1024 j (... and the global side: the caller passes base[j] - 1024 j).  A
// stamped timeline of the one-piece-per-M0 form showed ~95 cycles per piece in the issuing wave (M0 write -> s_nop -> load ->
// M0 restore, each a dependent scalar hop) against ~40 per instruction for back-to-back loads under one M0.
template <int N, int NJ>
__device__ __forceinline__ void tile_dma_pieces(const char* const (&b)[NJ], uint32_t voff, uint32_t lds_dst) {
    static_assert(N >= 1 && N <= 7 && N <= NJ, "pieces per wave and K-step");
    uint32_t keep;
#define VQA_P_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
#define VQA_P_TAIL "s_mov_b32 m0, %0"
    if constexpr (N == 1)
        asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\t" VQA_P_TAIL
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b[0]) : "memory");
    else if constexpr (N == 2)
        asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t" VQA_P_TAIL
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b[0]), "s"(b[1]) : "memory");
    else if constexpr (N == 3)
        asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %5 offset:2048\n\t" VQA_P_TAIL
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b[0]), "s"(b[1]), "s"(b[2]) : "memory");
    else if constexpr (N == 4)
        asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %5 offset:2048\n\tglobal_load_lds_dwordx4 %1, %6 offset:3072\n\t" VQA_P_TAIL
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]) : "memory");
    else {
        // five to seven pieces: the instruction offset reaches +4095, so a second M0 serves pieces 4 ..
        asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %5 offset:2048\n\tglobal_load_lds_dwordx4 %1, %6 offset:3072\n\t" VQA_P_TAIL
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]) : "memory");
        const uint32_t d2 = lds_dst + 4096;
        if constexpr (N == 5)
            asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\t" VQA_P_TAIL
                         : "=&s"(keep) : "v"(voff), "s"(d2), "s"(b[4]) : "memory");
        else if constexpr (N == 6)
            asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t" VQA_P_TAIL
                         : "=&s"(keep) : "v"(voff), "s"(d2), "s"(b[4]), "s"(b[5]) : "memory");
        else
            asm volatile(VQA_P_HEAD "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %5 offset:2048\n\t" VQA_P_TAIL
                         : "=&s"(keep) : "v"(voff), "s"(d2), "s"(b[4]), "s"(b[5]), "s"(b[6]) : "memory");
    }
#undef VQA_P_HEAD
#undef VQA_P_TAIL
}

template <int N>
__device__ __forceinline__ void tile_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- LayerNorm folded into the GEMMs around it (FOLD = 1; every forward that runs on the tile kernel, encoder_launch).  The post-LN block
// h = LN(y) feeds (a) the next GEMM and (b) the next residual add.  Both can start from the RAW sum y and the row's mean /
// variance:  (a)  LN(y) . W^T = rstd (y . (gamma (.) W)^T) - rstd mean colsum(gamma (.) W) + beta . W^T  -- the GEMM runs on y
// with the weights scaled by gamma at create time (Wf, cvec = its column sums as stored in fp16, bias' = bias + W beta) and
// its epilogue applies the two per-row scalars (EPI 0 / 1);  (b)  the EPI 2 epilogue normalises the residual row it reads
// anyway, (y - mean) rstd gamma + beta, per element.  The statistics come from the GEMM that produced y (EPI 2): every wave
// adds up sum and sum of squares of its slice of the row (the fp16 values as stored) and writes them to one of 16 slots of
// the row (slot = feature slice of BN / WN columns: P = N / (BN / WN) slots in use, a multiple of 4); the consumer adds the
// slots up in a fixed order -- deterministic, no atomics, no LayerNorm kernel and no second pass over the activations.
// sum over the four lanes c, c + 16, c + 32, c + 48 (the 4 feature groups g of one token row c), in every one of them:
// v_permlane16_swap / v_permlane32_swap of a register with itself give the two halves of each pair, one VALU op per step
__device__ __forceinline__ float quad_row_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // {rows 0 0 2 2, rows 1 1 3 3}
    const float h = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned uh = __builtin_bit_cast(unsigned, h);
    const auto b = __builtin_amdgcn_permlane32_swap(uh, uh, false, false);  // {lower half twice, upper half twice}
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

#ifndef VQA_FOLD_ABLATE
#define VQA_FOLD_ABLATE 0  // dev-only timing ablations (wrong results): 1 no statistics out, 2 no statistics in, 4 no sums, 8 no extra vectors
#endif
struct FoldArgs {
    const float2* st_in;  // [16][rows] (sum, sum of squares) slots of the rows of A (EPI 0 / 1) or of R (EPI 2)
    int p_in;             // slots in use (multiple of 4)
    int st_stride;        // rows per slot of st_in / st_out (the padded row count)
    float inv_h;          // 1 / LayerNorm width
    float eps;
    const float* cvec;    // EPI 0 / 1: column sums of the scaled weights [N]
    const float* g;       // EPI 2: gamma / beta of the LayerNorm of R [N]
    const float* b;
    float2* st_out;       // EPI 2: slots of the rows of C
};

#ifdef VQA_GSTAMPS
// dev-only diagnostic build (scripts/gemm_bench.py --stamps): s_memtime stamps of one workgroup's K-steps during its first
// tile, written by scalar stores to a buffer no other code reads (the instrumentation costs ~10 %: never in the product build)
constexpr int kGStampWg = 37, kGStampSlots = 8;
__device__ unsigned long long g_gstamps[8 * 64 * kGStampSlots];
#define VQA_GSTAMP(J)                                                             \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        if (stamp_on) asm volatile("s_memtime %0" : "=s"(stamp_t[J]));            \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#define VQA_GSTAMP_FLUSH(KT)                                                                                           \
    do {                                                                                                               \
        if (stamp_on && (KT) < 64) {                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            unsigned long long* sp_ = g_gstamps + ((size_t)wave * 64 + (KT)) * kGStampSlots;                           \
            asm volatile(                                                                                              \
                "s_store_dwordx2 %1, %0, 0x0\n\ts_store_dwordx2 %2, %0, 0x8\n\ts_store_dwordx2 %3, %0, 0x10\n\t"        \
                "s_store_dwordx2 %4, %0, 0x18\n\ts_store_dwordx2 %5, %0, 0x20\n\ts_store_dwordx2 %6, %0, 0x28\n\t"      \
                "s_store_dwordx2 %7, %0, 0x30\n\ts_store_dwordx2 %8, %0, 0x38"                                           \
                :: "s"(sp_), "s"(stamp_t[0]), "s"(stamp_t[1]), "s"(stamp_t[2]), "s"(stamp_t[3]), "s"(stamp_t[4]),       \
                   "s"(stamp_t[5]), "s"(stamp_t[6]), "s"(stamp_t[7]) : "memory");                                       \
        }                                                                                                              \
    } while (0)
#else
#define VQA_GSTAMP(J) (void)0
#define VQA_GSTAMP_FLUSH(KT) (void)0
#endif

template <int EPI, int BM, int BN, int WM, int WN, int BK, int FOLD = 0, int ONEBAR = 0, int TIL = 0>
__global__ __launch_bounds__(512) void gemm_tile_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                        const float* __restrict__ bias, const _Float16* __restrict__ R,
                                                        _Float16* __restrict__ C, int M, int N, int K, int tiles_n,
                                                        int tiles_total, int nb, FoldArgs fa) {
    using G = TileGeom<BM, BN, BK, (EPI == 1 && VQA_GELU_TABLE) ? kGeluBytes : 0>;
    static_assert(WM * WN == 8 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "8 waves, whole MFMA tiles per wave");
    constexpr int MT = BM / WM / 16, NT = BN / WN / 16;  // token / feature MFMA tiles per wave
    constexpr int S = G::kStages, D = S - 1, PR = G::kPieceRows, PA = BM / PR, NJ = G::kPerWave;
    constexpr int kTileRowB = G::kRowB, KSUB = BK / 32, SP = kTileRowB / 16;  // 16-byte slots per row: 4 or 8
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef __attribute__((address_space(3))) char* lds_ptr;
    const uint32_t lds_base = (uint32_t)(size_t)(lds_ptr)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;  // token rows [wr BM/WM, +BM/WM) x output features [wc BN/WN, +BN/WN) of the tile
    const int grp = wave >> 2;                 // partners w, w + 4 share a SIMD: one of each group
#ifndef VQA_GEMM_STAGGER
#define VQA_GEMM_STAGGER 0  // 1: a group's odd waves issue their pieces before their reads (measured 7 % SLOWER: DESIGN.md)
#endif
    const bool dma_first = VQA_GEMM_STAGGER && (wave & 1);
    const int c = lane & 15, g = lane >> 4;
    const int Gd = gridDim.x, wg = blockIdx.x;
    const bool swz = (tiles_total % 8 == 0) && (Gd % 8 == 0);
    const int per = swz ? Gd >> 3 : Gd, first = swz ? wg >> 3 : wg, span = swz ? tiles_total >> 3 : tiles_total;
    const int tile0 = swz ? (wg & 7) * span : 0;
    const int my_tiles = first < span ? (span - first + per - 1) / per : 0;
    const int KT = K / BK, total = my_tiles * KT;
    if (total == 0) return;
    // Tile of this workgroup's i-th iteration.  nb > 0 (the launcher checked the divisibilities): the `per` workgroups of an
    // XCD work on a block of (per / nb) token tiles x nb feature tiles at the same time, blocks advance along the features
    // first -- so what the XCD's L2 holds at any moment is (per / nb) activation row panels + nb weight panels (a whole
    // feature-fastest run of `per` tiles spans every weight panel: FFN1's 4.7 MB of weights did not fit the 4 MiB L2).
    auto tile_of = [&](int i) __attribute__((always_inline)) -> int {
        if (nb <= 0) return tile0 + first + i * per;
        const int mb = per / nb, nbn = tiles_n / nb;
        const int m = (tile0 / tiles_n) + (i / nbn) * mb + first / nb;
        return m * tiles_n + (i % nbn) * nb + first % nb;
    };

    // ---- issue side: this wave's pieces p = wave + 8 j of every K-step; piece p < PA holds token rows PR p .., the others weight
    // rows.  Lane i of a piece: row i / SP, physical slot i % SP, which holds logical slot phys ^ f(row) -- f = 3 bit3(row) for
    // 64-byte rows, row & 7 for 128-byte rows (both conflict-free for ds_read_b128)
    const size_t row_bytes = (size_t)K * 2;
    const int prow = lane / SP, pslot = lane % SP;
    const int lslot = pslot ^ (BK == 32 ? 3 * ((prow >> 3) & 1) : (prow & 7));
    // TIL: both operands in the TILED layout (256-row tiles x 64-byte K-blocks, 16 KiB blocks: kActBlock) -- a piece is 1 KiB
    // (32-deep K-steps) or two 512-byte runs (64-deep) of whole 128-byte lines instead of 16 half lines
    const uint32_t voff = TIL ? (uint32_t)(prow * 64 + (lslot >> 2) * 16384 + ((lslot & 3) << 4)) : (uint32_t)(prow * row_bytes + (lslot << 4));
    const size_t kadv = TIL ? (size_t)(BK / 32) * 16384 : (size_t)kTileRowB;
    // this wave's pieces of every K-step: the contiguous range [p0, p0 + n_mine) (NJ or NJ - 1 of them), so that their LDS
    // destinations are 1 KiB apart and one M0 write serves them all (tile_dma_pieces)
    const int p0 = G::kPieces * wave / 8;
    const int n_mine = G::kPieces * (wave + 1) / 8 - p0;
    int is_tile = 0, is_kt = 0, is_stage = 0, is_n = 0;  // issue cursor: tile, K-step inside it, ring stage, K-steps issued
    const char* src[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) src[j] = nullptr;
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (is_kt == 0 && is_n < total) {
            const int t = tile_of(is_tile);
            const size_t bm = (size_t)(t / tiles_n) * BM, bn = (size_t)(t % tiles_n) * BN;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int p = p0 + j < G::kPieces ? p0 + j : G::kPieces - 1;
                if constexpr (TIL) {
                    const size_t row0 = p < PA ? bm + PR * p : bn + PR * (p - PA);
                    src[j] = reinterpret_cast<const char*>(p < PA ? A : W) + (row0 >> 8) * (size_t)(K / 32) * 16384 + (row0 & 255) * 64 -
                             1024 * (j & 3);
                } else
                src[j] = (p < PA ? reinterpret_cast<const char*>(A) + (bm + PR * p) * row_bytes
                                 : reinterpret_cast<const char*>(W) + (bn + PR * (p - PA)) * row_bytes) - 1024 * (j & 3);
            }
        }
        const uint32_t dst = lds_base + is_stage * G::kStageB + p0 * 1024;
        const char* at[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) at[j] = src[j] + (size_t)is_kt * kadv;
        if (n_mine == NJ) tile_dma_pieces<NJ, NJ>(at, voff, dst);
        else if constexpr (NJ > 1) tile_dma_pieces<NJ - 1, NJ>(at, voff, dst);
        ++is_n;
        if (is_n < total) {  // past the end the cursor stays on the last K-step
            if (++is_kt == KT) {
                is_kt = 0;
                ++is_tile;
            }
        }
        if (++is_stage == S) is_stage = 0;
    };
    // all but the youngest `KSTEPS` K-steps of this wave's pieces have landed
    // (`extra`: that many plain loads were issued AFTER the pieces being waited for and may stay outstanding as well)
    auto wait_keep = [&](auto ksteps_tag, auto extra_tag) __attribute__((always_inline)) {
        constexpr int KS = decltype(ksteps_tag)::value, EX = decltype(extra_tag)::value;
        if (n_mine == NJ) tile_wait_vmcnt<KS * NJ + EX>();
        else tile_wait_vmcnt<KS * (NJ - 1) + EX>();
    };
    // ---- fragment geometry: row (base + c) of a stage, sub-step ks: logical slot 4 ks + g at position slot ^ f(row); row
    // bases are multiples of 16, so f(row) = f(c)
    int frag[KSUB];
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks)
        frag[ks] = c * kTileRowB + (((4 * ks + g) ^ (BK == 32 ? 3 * ((c >> 3) & 1) : (c & 7))) << 4);
    const int x_off = wr * (BM / WM) * kTileRowB;
    const int w_off = BM * kTileRowB + wc * (BN / WN) * kTileRowB;
    int rstage = 0;  // stage of the K-step this group reads next
    half8 wf[KSUB][NT], xf[KSUB][MT];
    half8 wf2[ONEBAR == 1 ? KSUB : 1][ONEBAR == 1 ? NT : 1], xf2[ONEBAR == 1 ? KSUB : 1][ONEBAR == 1 ? MT : 1];  // ONEBAR: the other fragment set
#define VQA_T_READ_SET(WF, XF)                                                                                \
    do {                                                                                                      \
        const char* st = lds + rstage * G::kStageB;                                                           \
        if (++rstage == S) rstage = 0;                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < KSUB; ++ks) {                                                 \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                WF[ks][j] = *reinterpret_cast<const half8*>(st + w_off + j * 16 * kTileRowB + frag[ks]);      \
            _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                    \
                XF[ks][j] = *reinterpret_cast<const half8*>(st + x_off + j * 16 * kTileRowB + frag[ks]);      \
        }                                                                                                     \
    } while (0)
#define VQA_T_READ() VQA_T_READ_SET(wf, xf)
// feature tiles [N0, N1) of one K-step from the fragment set (WF, XF); the pins keep hipcc from moving the MFMAs across what
// surrounds them (they are register-only)
#define VQA_T_MMA_SET(WF, XF, N0, N1)                                                                             \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(XF[0][j]));                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (!(VQA_GEMM_ABLATE & 4)) {                                                                             \
            _Pragma("unroll") for (int ks = 0; ks < KSUB; ++ks)                                                   \
                _Pragma("unroll") for (int ni = (N0); ni < (N1); ++ni)                                            \
                    _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                                             \
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[ks][ni], XF[ks][mi], acc[ni][mi], 0, 0, 0); \
        }                                                                                                         \
        _Pragma("unroll") for (int ni = (N0); ni < (N1); ++ni)                                                    \
            _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) asm volatile("" ::"v"(acc[ni][mi]));                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
// the MFMAs are register-only: the pins keep hipcc from moving them across the slot's barriers
#define VQA_T_MMA()                                                                                               \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(xf[0][j]));                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (!(VQA_GEMM_ABLATE & 4)) {                                                                             \
            _Pragma("unroll") for (int ks = 0; ks < KSUB; ++ks)                                                   \
                _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                                                 \
                    _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                                             \
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0); \
        }                                                                                                         \
        _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                                                         \
            _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) asm volatile("" ::"v"(acc[ni][mi]));                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define VQA_T_BARRIER()                         \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
    // EPI 1: the GELU table -> LDS behind the ring, one 1 KiB LDS-DMA piece per wave, issued FIRST: every counted wait below is for
    // younger pieces, so it has landed long before the first epilogue (and the barriers in between publish it to the other waves)
    if constexpr (EPI == 1 && VQA_GELU_TABLE) {
        uint32_t keep;
        const uint32_t dst = lds_base + G::kRing + wave * 1024;
        const char* src_tab = reinterpret_cast<const char*>(g_gelu_tab) + wave * 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"((uint32_t)(lane * 16)), "s"(dst), "s"(src_tab) : "memory");
    }
    [[maybe_unused]] const char* const gelu_lds = lds + G::kRing;
    // prologue: K-steps 0 .. D - 1 issued, K-step 0 landed; group 1 holds fragments(0)
    for (int i = 0; i < D; ++i) issue_next();
    if constexpr (ONEBAR == 1) wait_keep(std::integral_constant<int, (D >= 2 ? D - 2 : 0)>{}, std::integral_constant<int, 0>{});  // K-steps 0, 1 landed
    else wait_keep(std::integral_constant<int, D - 1>{}, std::integral_constant<int, 0>{});
    VQA_T_BARRIER();
    // the tile loop exists once per group, the wave-uniform branch sits outside it (no diamond around the MFMA blocks)
    auto run = [&](auto first_group_tag) __attribute__((always_inline)) {
        constexpr bool kG0 = decltype(first_group_tag)::value;
        if constexpr (ONEBAR == 1 || !kG0) {
            VQA_T_READ();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (ONEBAR != 1) VQA_T_BARRIER();
        for (int i = 0; i < my_tiles; ++i) {
            f32x4 acc[NT][MT];  // [feature tile ni][token tile mi]
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            // FOLD: the statistics slots of this lane's rows (written by the previous kernel: an HBM / Infinity Cache round trip)
            // are requested two K-steps before the epilogue needs them (plain loads inside the LDS-DMA stream: see the K loop).
            // The 256- / 288-wide tiles (and the residual form of the 256-row tiles) have no registers to park them in: loaded in the epilogue.
            constexpr bool kStatsEarly = FOLD && BN < 256 && !(EPI == 2 && BM == 256);  // (those shapes would spill)
            f32x4 su[FOLD ? MT : 1], sw[FOLD ? MT : 1];
            auto load_stats = [&]() __attribute__((always_inline)) {
                const int t = tile_of(i);
                const int m0 = (t / tiles_n) * BM + wr * (BM / WM) + c;
#pragma unroll
                for (int mi = 0; mi < (FOLD ? MT : 0); ++mi) {
                    const float2* sp = fa.st_in + (size_t)(4 * g) * fa.st_stride + (m0 + 16 * mi);  // slots 4 g .. 4 g + 3 of the row
                    const float2 q0 = sp[0], q1 = sp[fa.st_stride], q2 = sp[2 * (size_t)fa.st_stride], q3 = sp[3 * (size_t)fa.st_stride];
                    su[mi] = f32x4{q0.x, q0.y, q1.x, q1.y};
                    sw[mi] = f32x4{q2.x, q2.y, q3.x, q3.y};
                }
            };
            // one K-step (two slots); EX: plain loads in flight that are newer than the pieces the counted waits are for
#ifdef VQA_GSTAMPS
            const bool stamp_on = blockIdx.x == kGStampWg && i == 0;
            unsigned long long stamp_t[kGStampSlots] = {};
            int stamp_kt = 0;
#endif
            auto kstep = [&](auto extra_tag) __attribute__((always_inline)) {
                // ---- slot 1
                VQA_GSTAMP(0);
                if constexpr (kG0) {
                    // (VQA_GEMM_STAGGER=1, dev: the group's odd waves issue their pieces before their reads so that one pair's reads
                    // run while the other pair's pieces drain -- measured 7 % slower: the pieces then take 445 instead of 358 cycles)
                    if (dma_first) {
                        issue_next();
                        VQA_T_READ();
                    } else {
                        VQA_T_READ();
                        VQA_GSTAMP(1);
                        issue_next();
                    }
                    VQA_GSTAMP(2);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    VQA_GSTAMP(3);
                    wait_keep(std::integral_constant<int, D - 1>{}, extra_tag);  // own pieces of kappa + 1 landed (group 1 reads them in slot 2)
                } else {
                    VQA_T_MMA();
                    VQA_GSTAMP(1);
                    VQA_GSTAMP(2);
                    VQA_GSTAMP(3);
                    wait_keep(std::integral_constant<int, D - 2>{}, extra_tag);  // own pieces of kappa + 1 landed
                }
                VQA_GSTAMP(4);
                VQA_T_BARRIER();
                VQA_GSTAMP(5);
                // ---- slot 2
                if constexpr (kG0) {
                    VQA_T_MMA();
                    VQA_GSTAMP(6);
                } else {
                    if (dma_first) {
                        issue_next();
                        VQA_T_READ();
                    } else {
                        VQA_T_READ();
                        issue_next();
                    }
                    VQA_GSTAMP(6);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                VQA_GSTAMP(7);
                VQA_T_BARRIER();
#ifdef VQA_GSTAMPS
                VQA_GSTAMP_FLUSH(stamp_kt);
                ++stamp_kt;
#endif
            };
            // ---- ONEBAR: one barrier per K-step, two fragment sets (K1's tile_loop form).  Iteration kappa: the fragments of
            // kappa + 1 go to the other set, this wave's pieces of kappa + D are issued (into the stage of kappa - 1: its
            // fragments were read an iteration ago, before the last barrier), the MFMAs of kappa run from the current set; then
            // lgkmcnt(0), a counted vmcnt (this wave's pieces of kappa + 2 landed: they are read in the next iteration, behind
            // the barrier) and the barrier.  Group 0 runs memory -> matrix, group 1 half matrix -> memory -> half matrix, so the
            // partners of a SIMD start on different pipes.  Why: the stamped timeline of the slot form (scripts/gemm_bench.py
            // --stamps) shows slots bound by their memory phase -- 10 reads + 4 LDS-DMA pieces take 600-660 cycles against 440 for
            // the other group's 24 MFMAs -- plus ~150 cycles per barrier: 1570 cycles per 32-deep K-step for 768 cycles of MFMAs.
#define VQA_T_STEP1(CW, CX, NW, NX, EXTRA)                                                                            \
    do {                                                                                                              \
        VQA_GSTAMP(0);                                                                                                \
        if constexpr (!kG0) VQA_T_MMA_SET(CW, CX, 0, NT / 2);                                                         \
        VQA_GSTAMP(1);                                                                                                \
        VQA_T_READ_SET(NW, NX);                                                                                       \
        VQA_GSTAMP(2);                                                                                                \
        issue_next();                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        VQA_GSTAMP(3);                                                                                                \
        if constexpr (kG0) VQA_T_MMA_SET(CW, CX, 0, NT);                                                              \
        else VQA_T_MMA_SET(CW, CX, NT / 2, NT);                                                                       \
        VQA_GSTAMP(4);                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        VQA_GSTAMP(5);                                                                                                \
        wait_keep(std::integral_constant<int, (D >= 2 ? D - 2 : 0)>{}, std::integral_constant<int, (EXTRA)>{});       \
        VQA_GSTAMP(6);                                                                                                \
        VQA_T_BARRIER();                                                                                              \
        VQA_GSTAMP(7);                                                                                                \
        VQA_GSTAMP_FLUSH(stamp_kt_);                                                                                  \
        ++stamp_kt_;                                                                                                  \
    } while (0)
            [[maybe_unused]] int stamp_kt_ = 0;
            if constexpr (ONEBAR == 1) {
                static_assert(ONEBAR != 1 || D >= 2, "the one-barrier loop needs three stages");
                constexpr int kEx = (kStatsEarly && !(VQA_FOLD_ABLATE & 2)) ? 4 * MT : 0;
                // statistics loads in flight during the last two K-steps: they count in a wait only when they are NEWER than the
                // pieces it is for (K-step s + 2, issued in iteration s + 2 - D; the loads go out before iteration KT - 2)
                for (int kt = 0; kt < KT - 2; kt += 2) {
                    VQA_T_STEP1(wf, xf, wf2, xf2, 0);
                    VQA_T_STEP1(wf2, xf2, wf, xf, 0);
                }
                if constexpr (kEx > 0) load_stats();
                VQA_T_STEP1(wf, xf, wf2, xf2, (D > 2 ? kEx : 0));
                VQA_T_STEP1(wf2, xf2, wf, xf, (D > 3 ? kEx : 0));
            } else
            if constexpr (kStatsEarly && !(VQA_FOLD_ABLATE & 2)) {
                // the last two K-steps run with the 4 MT statistics loads in flight.  The pieces their waits are for (K-step
                // kappa + 1, issued D - 1 steps before) are older than those loads when D >= 3 -- the loads stay outstanding
                // (+4 MT in the counts: without it they would take the places of 4 MT pieces and stall the LDS-DMA stream);
                // with a two-deep stream the last step's pieces are younger than the loads and the plain count is the right one
                for (int kt = 0; kt < KT - 2; ++kt) kstep(std::integral_constant<int, 0>{});
                load_stats();
                kstep(std::integral_constant<int, 4 * MT>{});
                kstep(std::integral_constant<int, D >= 3 ? 4 * MT : 0>{});
            } else {
                for (int kt = 0; kt < KT; ++kt) kstep(std::integral_constant<int, 0>{});
            }
            // ---- epilogue: acc[ni][mi][j] = C[token bm + wr BM/WM + 16 mi + c][feature bn + wc BN/WN + 16 ni + 4 g + j].  A lane's
            // 4 features of one tile are 8 bytes of fp16: v_permlane16_swap between the registers of feature tiles ni and ni + 1
            // (lanes g = 1 <-> g = 0, g = 3 <-> g = 2 of the same token) leaves every lane with 8 CONSECUTIVE features -- tile
            // ni + (g & 1), features 8 (g >> 1) .. + 7 -- so a pair of tiles goes out as one 16-byte store per lane (a wave
            // instruction covers 16 token rows x 64 contiguous bytes); bias, GELU and the residual row are applied in that layout.
            // An odd last feature tile keeps the 8-byte form.
            // Loads and stores share one in-order counter (vmcnt): a wait for a load also waits for every older store, and a
            // branch makes hipcc wait for everything -- the first form of this epilogue (a row bound check around every store,
            // the bias / residual loads of a pair issued behind the previous pair's stores) completed every store before the
            // next one went out (7-13 us per GEMM, profiles/r02_encoder_gemm_ablation.txt).  Hence: straight-line code, rows
            // past M are computed and stored like any other (they land in the padding rows of C: every activation buffer is
            // padded to 256 rows), and the loads of pair pr + 1 go out BEFORE the stores of pair pr.
            const int t = tile_of(i);
            const int bm = (t / tiles_n) * BM, bn = (t % tiles_n) * BN;
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            constexpr int NP = NT / 2;
            const int m_base = bm + wr * (BM / WM) + c;  // this lane's rows: m_base + 16 mi
            // FOLD: rstd and mean x rstd of this lane's MT token rows (lanes g = 0..3 of a token each add up 4 of the 16 slots)
            float rs[MT], mrs[MT], ps[MT], pss[MT];
            if constexpr (FOLD) {
                if constexpr (!kStatsEarly && !(VQA_FOLD_ABLATE & 2)) load_stats();
                const bool live = 4 * g < fa.p_in;  // slots past P hold stale values of an earlier call
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    float sm = (su[mi][0] + su[mi][2]) + (sw[mi][0] + sw[mi][2]), sq = (su[mi][1] + su[mi][3]) + (sw[mi][1] + sw[mi][3]);
                    sm = quad_row_sum(live ? sm : 0.f);
                    sq = quad_row_sum(live ? sq : 0.f);
#if VQA_FOLD_ABLATE & 2
                    sm = 0.f;
                    sq = 768.f;
#endif
                    const float mean = sm * fa.inv_h;
                    const float r = rsqrtf(fmaxf(sq * fa.inv_h - mean * mean, 0.f) + fa.eps);
                    rs[mi] = r;
                    mrs[mi] = mean * r;
                    ps[mi] = 0.f;
                    pss[mi] = 0.f;
                }
            }
            // per-feature vectors of a pair (two buffers: pair pr + 1 is loaded while pair pr is computed and stored):
            // v[0], v[1] = bias; v[2], v[3] = FOLD: column sums (EPI 0 / 1) or gamma (EPI 2); v[4], v[5] = FOLD EPI 2: beta
            constexpr int NV = FOLD ? (EPI == 2 ? 6 : 4) : 2;
            f32x4 vec[2][NV];
            half8 res[2][MT];
            auto pair_n0 = [&](int pr) __attribute__((always_inline)) {
                return bn + wc * (BN / WN) + (2 * pr + (g & 1)) * 16 + (g >> 1) * 8;
            };
            auto load_pair = [&](int pr, int buf) __attribute__((always_inline)) {
                const int n0 = pair_n0(pr);
                vec[buf][0] = *reinterpret_cast<const f32x4*>(bias + n0);
                vec[buf][1] = *reinterpret_cast<const f32x4*>(bias + n0 + 4);
                if constexpr (FOLD && bool(VQA_FOLD_ABLATE & 8)) {
                    vec[buf][2] = vec[buf][0];
                    vec[buf][3] = vec[buf][1];
                    if constexpr (EPI == 2) {
                        vec[buf][4] = vec[buf][0];
                        vec[buf][5] = vec[buf][1];
                    }
                } else if constexpr (FOLD) {
                    const float* cg = EPI == 2 ? fa.g : fa.cvec;
                    vec[buf][2] = *reinterpret_cast<const f32x4*>(cg + n0);
                    vec[buf][3] = *reinterpret_cast<const f32x4*>(cg + n0 + 4);
                    if constexpr (EPI == 2) {
                        vec[buf][4] = *reinterpret_cast<const f32x4*>(fa.b + n0);
                        vec[buf][5] = *reinterpret_cast<const f32x4*>(fa.b + n0 + 4);
                    }
                }
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        res[buf][mi] = *reinterpret_cast<const half8*>(R + (size_t)(m_base + 16 * mi) * N + n0);
                }
            };
            // the odd last feature tile's loads (8-byte form)
            f32x4 tvec[NV / 2 > 0 ? NV / 2 : 1];
            half4 tres[MT];
            auto load_tail = [&]() __attribute__((always_inline)) {
                const int n0 = bn + wc * (BN / WN) + (NT - 1) * 16 + g * 4;
                tvec[0] = *reinterpret_cast<const f32x4*>(bias + n0);
                if constexpr (FOLD) {
                    tvec[1] = *reinterpret_cast<const f32x4*>((EPI == 2 ? fa.g : fa.cvec) + n0);
                    if constexpr (EPI == 2) tvec[2] = *reinterpret_cast<const f32x4*>(fa.b + n0);
                }
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        tres[mi] = *reinterpret_cast<const half4*>(R + (size_t)(m_base + 16 * mi) * N + n0);
                }
            };
            if constexpr (NP > 0) load_pair(0, 0);
            else if constexpr (NT & 1) load_tail();
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const int buf = pr & 1;
                // (the 288-wide tile has no registers for two sets while all its accumulators are live: its pair 1 is loaded
                // after pair 0 went out -- one wait for stores per tile; from there on the accumulators of stored pairs are free)
                constexpr bool kTight = BN >= 288;
                if (kTight && pr == 1) load_pair(1, 1);
                if (pr + 1 < NP) {
                    if (!kTight || pr >= 1) load_pair(pr + 1, buf ^ 1);
                } else if constexpr (NT & 1) load_tail();
                __builtin_amdgcn_sched_barrier(0);  // hipcc sinks these loads below the stores of pair pr otherwise
                const int n0 = pair_n0(pr);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    // (cast the whole vector first: __builtin_bit_cast of a single vector ELEMENT reads element 0 for every index)
                    u32x4 ulo = __builtin_bit_cast(u32x4, acc[2 * pr][mi]), uhi = __builtin_bit_cast(u32x4, acc[2 * pr + 1][mi]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const auto r = __builtin_amdgcn_permlane16_swap(ulo[j], uhi[j], false, false);
                        ulo[j] = r[0];
                        uhi[j] = r[1];
                    }
                    const f32x4 lo = __builtin_bit_cast(f32x4, ulo), hi = __builtin_bit_cast(f32x4, uhi);
                    half8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v0, v1;
                        if constexpr (FOLD && EPI != 2) {  // rstd (y . Wf) - rstd mean colsum + bias'
                            v0 = lo[j] * rs[mi] - mrs[mi] * vec[buf][2][j] + vec[buf][0][j];
                            v1 = hi[j] * rs[mi] - mrs[mi] * vec[buf][3][j] + vec[buf][1][j];
                        } else {
                            v0 = lo[j] + vec[buf][0][j];
                            v1 = hi[j] + vec[buf][1][j];
                        }
#if !(VQA_GEMM_ABLATE & 2)
                        if (EPI == 1) {
                            if constexpr (VQA_GELU_TABLE) {
                                v0 = gelu_tab(v0, gelu_lds);
                                v1 = gelu_tab(v1, gelu_lds);
                            } else {
                                v0 = gelu_erf(v0);
                                v1 = gelu_erf(v1);
                            }
                        }
#endif
                        if constexpr (EPI == 2) {
                            if constexpr (FOLD) {  // the residual is LN(raw row)
                                v0 += ((float)res[buf][mi][j] * rs[mi] - mrs[mi]) * vec[buf][2][j] + vec[buf][4][j];
                                v1 += ((float)res[buf][mi][4 + j] * rs[mi] - mrs[mi]) * vec[buf][3][j] + vec[buf][5][j];
                            } else {
                                v0 += (float)res[buf][mi][j];
                                v1 += (float)res[buf][mi][4 + j];
                            }
                        }
                        o[j] = (_Float16)v0;
                        o[4 + j] = (_Float16)v1;
                        if constexpr (FOLD && EPI == 2 && !(VQA_FOLD_ABLATE & 4)) {  // statistics of the row AS STORED
                            const float q0 = (float)o[j], q1 = (float)o[4 + j];
                            ps[mi] += q0 + q1;
                            pss[mi] += q0 * q0 + q1 * q1;
                        }
                    }
#if VQA_GEMM_ABLATE & 1
                    asm volatile("" ::"v"(o));
#else  // plain stores: sc1 / sc0 sc1 (write-through, dropped from the XCD's L2) measured the same, nt 6 % slower on the forward
                    *reinterpret_cast<half8*>(C + (size_t)(m_base + 16 * mi) * N + n0) = o;
#endif
                }
            }
            if constexpr (NT & 1) {
                const int n0 = bn + wc * (BN / WN) + (NT - 1) * 16 + g * 4;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v;
                        if constexpr (FOLD && EPI != 2) v = acc[NT - 1][mi][j] * rs[mi] - mrs[mi] * tvec[1][j] + tvec[0][j];
                        else v = acc[NT - 1][mi][j] + tvec[0][j];
#if !(VQA_GEMM_ABLATE & 2)
                        if (EPI == 1) v = VQA_GELU_TABLE ? gelu_tab(v, gelu_lds) : gelu_erf(v);
#endif
                        if constexpr (EPI == 2) {
                            if constexpr (FOLD) v += ((float)tres[mi][j] * rs[mi] - mrs[mi]) * tvec[1][j] + tvec[2][j];
                            else v += (float)tres[mi][j];
                        }
                        o[j] = (_Float16)v;
                        if constexpr (FOLD && EPI == 2) {
                            const float q = (float)o[j];
                            ps[mi] += q;
                            pss[mi] += q * q;
                        }
                    }
#if VQA_GEMM_ABLATE & 1
                    asm volatile("" ::"v"(o));
#else
                    *reinterpret_cast<half4*>(C + (size_t)(m_base + 16 * mi) * N + n0) = o;
#endif
                }
            }
            if constexpr (FOLD && EPI == 2 && !(VQA_FOLD_ABLATE & 1)) {  // this wave's slice of every row: one (sum, sum of squares) slot
                const int slot = (bn + wc * (BN / WN)) / (BN / WN);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const float sm = quad_row_sum(ps[mi]), sq = quad_row_sum(pss[mi]);
                    if (g == 0) fa.st_out[(size_t)slot * fa.st_stride + (m_base + 16 * mi)] = make_float2(sm, sq);  // rows past M: padding
                }
            }
        }
    };
    if (grp) run(std::false_type{});
    else run(std::true_type{});
    tile_wait_vmcnt<0>();  // the pieces issued past the end of the stream land before the LDS is released
#undef VQA_T_READ
#undef VQA_T_READ_SET
#undef VQA_T_MMA
#undef VQA_T_MMA_SET
#undef VQA_T_STEP1
#undef VQA_T_BARRIER
}

// ---- GEMM for a handful of tokens (M <= 64: single queries, the reference's own calling pattern heavy_ranker.py:97-98):
// the work is streaming the weight matrix once.  One workgroup per 16 output features, its 4 waves split K four ways
// (so even N = 768 puts 192 waves on the chip), operands go global -> registers (16 B per lane, the MFMA fragment itself:
// weight rows as the A operand, token rows as the B operand), next K block prefetched while the current one multiplies;
// the four partial tiles are summed through LDS and wave 0 applies bias / GELU and stores 8 bytes per lane.
template <int EPI, int MT>  // MT token tiles of 16
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                          const float* __restrict__ bias, const _Float16* __restrict__ R,
                                                          _Float16* __restrict__ C, int M, int N, int K) {
    __shared__ f32x4 red[3][MT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * (16 * MT);  // token chunk of this workgroup (more than 64 tokens: several chunks per feature slice)
    const int kq = K >> 2;  // this wave's K range: [wave * kq, +kq), a multiple of 64
    const _Float16* wp = W + (size_t)(n0 + c) * K + wave * kq + 8 * g;
    const _Float16* ap[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int row = m0 + mi * 16 + c < M ? m0 + mi * 16 + c : M - 1;  // rows past M: a copy of the last row, dropped
        ap[mi] = A + (size_t)row * K + wave * kq + 8 * g;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 w0 = *reinterpret_cast<const half8*>(wp), w1 = *reinterpret_cast<const half8*>(wp + 32);
    half8 x0[MT], x1[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        x0[mi] = *reinterpret_cast<const half8*>(ap[mi]);
        x1[mi] = *reinterpret_cast<const half8*>(ap[mi] + 32);
    }
    for (int k = 0; k < kq; k += 64) {
        half8 nw0 = w0, nw1 = w1, nx0[MT], nx1[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            nx0[mi] = x0[mi];
            nx1[mi] = x1[mi];
        }
        if (k + 64 < kq) {  // next 64 k of this wave's range
            nw0 = *reinterpret_cast<const half8*>(wp + k + 64);
            nw1 = *reinterpret_cast<const half8*>(wp + k + 96);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                nx0[mi] = *reinterpret_cast<const half8*>(ap[mi] + k + 64);
                nx1[mi] = *reinterpret_cast<const half8*>(ap[mi] + k + 96);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0[mi], acc[mi], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x1[mi], acc[mi], 0, 0, 0);
        w0 = nw0;
        w1 = nw1;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            x0[mi] = nx0[mi];
            x1[mi] = nx1[mi];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) red[wave - 1][mi][lane] = acc[mi];
    }
    __syncthreads();
    if (wave > 0) return;
    // acc[mi][j] = C[token 16 mi + c][feature n0 + 4 g + j]
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + 4 * g);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = m0 + mi * 16 + c;
        if (m >= M) continue;
        const f32x4 sum = acc[mi] + red[0][mi][lane] + red[1][mi][lane] + red[2][mi][lane];
        half4 res = half4{0, 0, 0, 0};
        if (EPI == 2) res = *reinterpret_cast<const half4*>(R + (size_t)m * N + n0 + 4 * g);
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = sum[j] + bv[j];
            if (EPI == 1) v = gelu_erf(v);
            if (EPI == 2) v += (float)res[j];
            o[j] = (_Float16)v;
        }
        *reinterpret_cast<half4*>(C + (size_t)m * N + n0 + 4 * g) = o;
    }
}

// ---- GEMM for ONE question (M <= 64 tokens, K a multiple of 768): the latency form of the reference's own calling pattern --
// heavy_ranker.py:97-101 asks one question per call.  A forward of 32 tokens is a chain of small launches whose time is the
// weight stream's first-byte latency, so (a) every load of a wave is in flight before its first MFMA -- eight waves split K, each
// issues its weight fragments and its token fragments of its K / 8 range at once (gemm_skinny_kernel kept two K-blocks in flight
// and paid a memory round trip per 64 k) -- and (b) the two LayerNorm launches of a layer disappear: FOLDIN = 1 (QKV, FFN1) runs on
// the RAW residual sum with the gamma-scaled weights of the tile kernel's fold (Wf, column sums, bias + W beta), adds up every
// row's sum and sum of squares from the fragments it loads anyway and applies (rstd, mean rstd) in its epilogue; it also writes
// (mean, rstd) per row, with which the next EPI 2 GEMM (out-projection, FFN2) normalises the residual row it adds.  Five launches
// per layer instead of seven, each a single memory round trip deep.
// Lane roles as in gemm_skinny_kernel: A operand = 16 weight rows (features n0 + c), B operand = token rows; a K-block is 32 deep.
struct TinyArgs {
    const float* cvec = nullptr;   // FOLDIN: column sums of the folded weights [N]
    float2* st_out = nullptr;      // FOLDIN: (mean, rstd) of every input row, written by workgroup 0 [M]
    const float2* st_in = nullptr; // EPI 2 with g != nullptr: (mean, rstd) of the residual's raw rows [M]
    const float* g = nullptr;      // EPI 2: gamma / beta of the LayerNorm of the residual (nullptr: the residual is added as it is)
    const float* b = nullptr;
    float inv_k = 0.f, eps = 0.f;
};

template <int EPI, int MT, int FOLDIN, int NBC, int WV = 8, bool COH = false>  // MT token tiles of 16; NBC K-blocks per chunk (all of a chunk's loads in flight together);
                                                             // WV waves split K (8; 4 for K = 384, the MiniLM-L12 hidden size of heavy_ranker.py:80)
// (bx, by, tid: the unit and thread; threads of waves >= WV -- the persistent kernel's 512-thread workgroup running a four-wave form -- only
// take part in the barrier)
__device__ __forceinline__ void gemm_tiny_body(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                               const float* __restrict__ bias, const _Float16* __restrict__ R,
                                               _Float16* __restrict__ C, int M, int N, int K, const TinyArgs& ta, int bx, int by, int tid,
                                               const half8* wpre = nullptr /* this wave's weight fragments of its (single) chunk, loaded ahead */) {
    __shared__ f32x4 red[WV - 1][MT][64];
    __shared__ float2 sred[WV][MT][16];
    const int lane = tid & 63, wave = tid >> 6;
    if (wave >= WV) {
        __syncthreads();
        return;
    }
    const int c = lane & 15, g = lane >> 4;
    const int n0 = bx * 16;
    const int m0 = by * (16 * MT);  // by > 0: the token tiles are spread over workgroups (FFN2: 48 feature slices alone leave 208 CUs idle)
    const int kq = K / WV;  // this wave's K range [wave kq, +kq): a multiple of 32 NBC
    const int kbase = wave * kq;
    const _Float16* wp = W + (size_t)(n0 + c) * K + kbase + 8 * g;
    const _Float16* ap[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int row = m0 + mi * 16 + c < M ? m0 + mi * 16 + c : M - 1;  // rows past M: a copy of the last row, dropped
        ap[mi] = A + (size_t)row * K + kbase + 8 * g;
    }
    f32x4 acc[MT];
    float s1[MT], s2[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        s1[mi] = s2[mi] = 0.f;
    }
    // wave 0 runs the epilogue: its per-feature vectors, residual rows and row statistics are requested NOW, in front of the K loop's
    // loads (behind the barrier they would be one more memory round trip on the launch's critical path: ~1.5 us of a 5 us launch)
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, cv = bv, gv = bv, bb = bv;
    half4 res[MT];
    float2 st[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        res[mi] = half4{0, 0, 0, 0};
        st[mi] = make_float2(0.f, 1.f);
    }
    if (wave == 0) {
        bv = *reinterpret_cast<const f32x4*>(bias + n0 + 4 * g);
        if constexpr (FOLDIN) cv = *reinterpret_cast<const f32x4*>(ta.cvec + n0 + 4 * g);
        if (EPI == 2 && ta.g) {
            gv = *reinterpret_cast<const f32x4*>(ta.g + n0 + 4 * g);
            bb = *reinterpret_cast<const f32x4*>(ta.b + n0 + 4 * g);
        }
        if (EPI == 2) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const int m = m0 + mi * 16 + c < M ? m0 + mi * 16 + c : M - 1;
                res[mi] = act_load8<COH, half4>(R + (size_t)m * N + n0 + 4 * g);
                if (ta.g) st[mi] = act_load8<COH, float2>(ta.st_in + m);
            }
        }
    }
    for (int k0 = 0; k0 < kq; k0 += 32 * NBC) {
        half8 wf[NBC], xf[NBC][MT];
#pragma unroll
        for (int j = 0; j < NBC; ++j) wf[j] = wpre ? wpre[j] : *reinterpret_cast<const half8*>(wp + k0 + 32 * j);
#pragma unroll
        for (int j = 0; j < NBC; ++j)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) xf[j][mi] = act_load16<COH>(ap[mi] + k0 + 32 * j);
        // every load of the chunk is issued before the first MFMA (left alone, hipcc interleaves them with the MFMAs to save registers:
        // a memory round trip per few K-blocks again)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NBC; ++j)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[j][mi], acc[mi], 0, 0, 0);
                if constexpr (FOLDIN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (float)xf[j][mi][e];
                        s1[mi] += v;
                        s2[mi] += v * v;
                    }
                }
            }
    }
    if constexpr (FOLDIN) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {  // the four k-subblocks g of row c
            s1[mi] += __shfl_xor(s1[mi], 16, 64);
            s1[mi] += __shfl_xor(s1[mi], 32, 64);
            s2[mi] += __shfl_xor(s2[mi], 16, 64);
            s2[mi] += __shfl_xor(s2[mi], 32, 64);
            if (g == 0) sred[wave][mi][c] = make_float2(s1[mi], s2[mi]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) red[wave - 1][mi][lane] = acc[mi];
    }
    __syncthreads();
    if (wave > 0) return;
    f32x4 total[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        total[mi] = acc[mi];
#pragma unroll
        for (int w = 0; w < WV - 1; ++w) total[mi] += red[w][mi][lane];  // fixed order: deterministic
    }
    // total[mi][j] = C[token 16 mi + c][feature n0 + 4 g + j]
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = m0 + mi * 16 + c;
        const f32x4 sum = total[mi];
        float rstd = 1.f, mrs = 0.f;
        if constexpr (FOLDIN) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < WV; ++w) {
                a1 += sred[w][mi][c].x;
                a2 += sred[w][mi][c].y;
            }
            const float mean = a1 * ta.inv_k;
            rstd = rsqrtf(fmaxf(a2 * ta.inv_k - mean * mean, 0.f) + ta.eps);
            mrs = mean * rstd;
            if (ta.st_out && bx == 0 && g == 0 && m < M) ta.st_out[m] = make_float2(mean, rstd);
        }
        if (m >= M) continue;
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = FOLDIN ? sum[j] * rstd - mrs * cv[j] + bv[j] : sum[j] + bv[j];
            if (EPI == 1) v = gelu_erf(v);
            if (EPI == 2) v += ta.g ? ((float)res[mi][j] - st[mi].x) * st[mi].y * gv[j] + bb[j] : (float)res[mi][j];
            o[j] = (_Float16)v;
        }
        *reinterpret_cast<half4*>(C + (size_t)m * N + n0 + 4 * g) = o;
    }
}
template <int EPI, int MT, int FOLDIN, int NBC, int WV = 8>
__global__ __launch_bounds__(64 * WV) void gemm_tiny_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                        const float* __restrict__ bias, const _Float16* __restrict__ R,
                                                        _Float16* __restrict__ C, int M, int N, int K, TinyArgs ta) {
    gemm_tiny_body<EPI, MT, FOLDIN, NBC, WV>(A, W, bias, R, C, M, N, K, ta, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x);
}

// token ids and masks of vqa_encoder_forward_host: pinned host memory -> the device arrays a replayed graph reads
__global__ __launch_bounds__(256) void stage_tokens_kernel(const int* __restrict__ ids, const int* __restrict__ mask, int T,
                                                           int* __restrict__ ids_out, int* __restrict__ mask_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < T) {
        const int a = ids[i], b = mask[i];
        ids_out[i] = a;
        mask_out[i] = b;
    }
}

// ---- attention: one workgroup per (sequence, head); K and V of the head in LDS (fp32), one wave per query row ------
__global__ __launch_bounds__(256) void attention_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ mask,
                                                        int L, int H, int heads, _Float16* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dh = H / heads;           // <= 64... any dh <= 128 works with the loops below
    float* ks = reinterpret_cast<float*>(smem);   // [L][dh + 1]
    float* vs = ks + (size_t)L * (dh + 1);        // [L][dh + 1]
    float* ps = vs + (size_t)L * (dh + 1);        // [4 waves][L] probabilities of the wave's current row
    float* qs = ps + 4 * L;                       // [4 waves][dh]
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row_stride = (size_t)3 * H;
    const _Float16* base = qkv + (size_t)seq * L * row_stride + head * dh;
    for (int i = tid; i < L * dh; i += 256) {
        const int l = i / dh, d = i - l * dh;
        ks[l * (dh + 1) + d] = (float)base[(size_t)l * row_stride + H + d];
        vs[l * (dh + 1) + d] = (float)base[(size_t)l * row_stride + 2 * H + d];
    }
    __syncthreads();
    const float scale = rsqrtf((float)dh);
    for (int r = wave; r < L; r += 4) {
        float* q = qs + wave * dh;
        for (int d = lane; d < dh; d += 64) q[d] = (float)base[(size_t)r * row_stride + d];
        __builtin_amdgcn_wave_barrier();
        // scores for keys j = lane, lane + 64, ...
        float mx = -INFINITY;
        for (int j = lane; j < L; j += 64) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s += q[d] * ks[j * (dh + 1) + d];
            s = mask[seq * L + j] ? s * scale : -INFINITY;  // HF adds finfo.min: the masked key's weight is exactly 0
            ps[wave * L + j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float e = __expf(ps[wave * L + j] - mx);
            ps[wave * L + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / sum;
        for (int d = lane; d < dh; d += 64) {
            float o = 0.f;
            for (int j = 0; j < L; ++j) o += ps[wave * L + j] * vs[j * (dh + 1) + d];
            ctx[((size_t)seq * L + r) * H + head * dh + d] = (_Float16)(o * inv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- attention on the matrix cores (head size DH = 64 or 32, L <= 256): one workgroup per (sequence, HPW heads), one wave per head and
// block of 32 queries.  S^T = K . Q^T with v_mfma_f32_32x32x16_f16 (A = keys, B = queries: a lane then holds ONE query's scores
// for 16 keys per 32-key block in registers, so the softmax is register-local plus one cross-half shuffle); the
// normalised probabilities, converted to fp16, are used straight from the accumulator registers as the A operand of
// P . V (X^T . B form: no lane movement).  V sits in LDS ROW-major, as it lies in memory (16-byte stores), and the B operand --
// 8 keys of one head dimension per lane, in the permuted key order of the accumulator registers -- comes out of it through the
// transposing read ds_read_b64_tr_b16: per 16-lane group a block of 4 keys x 16 dimensions, lane i gets dimension i of the 4
// keys.  (Round 2 wrote a transposed image with 2-byte LDS stores: 82 % of the kernel's LDS cycles were bank conflicts.)
// Image: key row of 128 B = two 64-byte halves (dimensions 0-31 | 32-63), the halves of rows with bit 1 of the key set are
// swapped, so the four rows of a block start on banks 0 / 32 / 16 / 48: every transposed read and every store is conflict-free.
// DH = 32 (MiniLM-L12: 12 heads of 32, the model heavy_ranker.py:80 loads): key rows of 64 B, the four rows of a block are 256
// contiguous bytes = all 64 banks once, no swap needed; Q . K^T takes 2 instead of 4 k-steps, P . V one 32-dimension block.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int kAttMaxBlocks = 8;  // L <= 256
__host__ __device__ constexpr bool att_mfma_head_size(int dh) { return dh == 64 || dh == 32; }

template <int NQB, int HPW, int DH, bool COH = false>
__device__ __forceinline__ void attention_mfma_body(const _Float16* __restrict__ qkv, const int* __restrict__ mask,
                                                    int Lmax, int H, int heads, const int* __restrict__ cu,
                                                    _Float16* __restrict__ ctx, int bx, int tid, char* smem) {
    constexpr int Lp = NQB * 32;
    const int lane = tid & 63, wave = tid >> 6;
    if (wave >= NQB * HPW) {  // (a wider workgroup of the persistent kernel: the extra waves only take part in the barrier)
        __syncthreads();
        return;
    }
    const int hw = wave / NQB, qb = wave - hw * NQB;  // head inside the workgroup, query block
    const int groups = heads / HPW;
    const int seq = bx / groups, head = (bx - seq * groups) * HPW + hw;
    static_assert(DH == 64 || DH == 32, "head sizes of the matrix-core attention");
    constexpr int kRowB = DH * 2;      // bytes of a key row of the V image
    constexpr int kCh = DH / 8;        // 16-byte chunks per row
    constexpr int kVregs = DH / 16;    // 16-byte chunks of V per lane (Lp * kCh chunks over 64 NQB lanes)
    char* vimg = smem + hw * (Lp * kRowB);  // this head's V image: [Lp keys][kRowB]
    const int li = lane & 31, h = lane >> 5;
    const size_t row_stride = (size_t)3 * H;
    // packed rows: the sequence owns rows [cu[seq], cu[seq + 1]) and every one of them is a real token
    const size_t row0 = cu ? (size_t)cu[seq] : (size_t)seq * Lmax;
    const int L = cu ? cu[seq + 1] - cu[seq] : Lmax;
    const _Float16* base = qkv + row0 * row_stride + head * DH;
    // V -> registers now (the NQB waves of this head: kVregs x 16 bytes per lane; keys beyond L are zero), -> LDS behind the softmax:
    // the loads travel while Q . K^T and the softmax run
    half8 vreg[kVregs];
#pragma unroll
    for (int t = 0; t < kVregs; ++t) {
        const int i = tid - hw * 64 * NQB + t * 64 * NQB;
        const int key = i / kCh, ch = i % kCh;
        vreg[t] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (key < L) vreg[t] = act_load16<COH>(base + (size_t)key * row_stride + 2 * H + ch * 8);
    }
    const int qrow = qb * 32 + li < L ? qb * 32 + li : L - 1;  // padded query rows recompute the last row, never stored
    half8 qf[kVregs];
#pragma unroll
    for (int kk = 0; kk < kVregs; ++kk) qf[kk] = act_load16<COH>(base + (size_t)qrow * row_stride + kk * 16 + h * 8);
    f32x16 st[NQB];
    const float scale = rsqrtf((float)DH);
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NQB; ++kb) {
        const int key = kb * 32 + li;
        const int krow = key < L ? key : L - 1;
        f32x16 acc = {};
#pragma unroll
        for (int kk = 0; kk < kVregs; ++kk) {
            const half8 kf = act_load16<COH>(base + (size_t)krow * row_stride + H + kk * 16 + h * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], acc, 0, 0, 0);
        }
        // key validity of this block as a wave-uniform bit mask (bit = key index inside the block)
        const int valid = key < L && (cu != nullptr || mask[seq * Lmax + (key < L ? key : 0)] != 0);
        const unsigned long long bal = __ballot(valid);
        const unsigned int kmask = (unsigned int)(bal & 0xFFFFFFFFull);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kidx = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float sv = ((kmask >> kidx) & 1u) ? acc[r] * scale : -INFINITY;  // HF adds finfo.min: weight exactly 0
            acc[r] = sv;
            mx = fmaxf(mx, sv);
        }
        st[kb] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NQB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(st[kb][r] - mx);
            st[kb][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < kVregs; ++t) {
        const int i = tid - hw * 64 * NQB + t * 64 * NQB;
        const int key = i / kCh, ch = i % kCh;
        if constexpr (DH == 64) *reinterpret_cast<half8*>(vimg + key * 128 + (((ch >> 2) ^ ((key >> 1) & 1)) << 6) + ((ch & 3) << 4)) = vreg[t];
        else *reinterpret_cast<half8*>(vimg + key * kRowB + (ch << 4)) = vreg[t];
    }
    __syncthreads();  // V images complete
    // transposed read of this lane: block row q_ = (lane & 15) >> 2 (+ the key base), columns 4 p_ .. 4 p_ + 3 of the 16-lane
    // group's 16 dimensions 16 (group & 1) ..; rows k0 + q_ with k0 a multiple of 4, so the half swap of a row is (q_ >> 1) & 1
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int q_ = (lane & 15) >> 2, p_ = lane & 3, gi = lane >> 4;
    const int tr_off = q_ * kRowB + ((gi & 1) << 5) + (p_ << 3);  // + key base * kRowB (+ (db ^ swap) * 64 for DH = 64)
    const int swap = DH == 64 ? (q_ >> 1) & 1 : 0;
    constexpr int kDb = DH / 32;  // 32-dimension blocks of the output
    f32x16 o[kDb] = {};
#pragma unroll
    for (int kb = 0; kb < NQB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            half8 pf;  // accumulator registers 8 s2 .. 8 s2 + 7 -> k-step s2 of the A operand (X^T . B form)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (_Float16)(st[kb][8 * s2 + j] * inv);
            // element j of lane half h is key kb*32 + 16 s2 + 8 (j >> 2) + 4 h + (j & 3): V must use the same order
            const int k0 = kb * 32 + 16 * s2 + 4 * h;
#pragma unroll
            for (int db = 0; db < kDb; ++db) {
                const char* at = vimg + k0 * kRowB + ((db ^ swap) << 6) + tr_off;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(at));               // keys k0 .. k0 + 3
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(at + 8 * kRowB));   // keys k0 + 8 .. k0 + 11
                typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                // O^T = V^T . P^T: the transposed V fragment is the A operand (lane = head dimension), P the B operand (lane = query,
                // its accumulator registers as they are), so the lane that holds a query's probabilities also receives its output row
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, both), pf, o[db], 0, 0, 0);
            }
        }
    // o[db][r] = O[query qb 32 + li][dimension 32 db + 8 (r >> 2) + 4 h + (r & 3)]: four consecutive dimensions per register quad,
    // one 8-byte store each (the first form -- lane = dimension -- wrote 32 two-byte values per lane)
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const int qo = qb * 32 + li;
    if (qo < L) {
        _Float16* orow = ctx + (row0 + qo) * H + head * DH + 4 * h;
#pragma unroll
        for (int db = 0; db < kDb; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<half4*>(orow + db * 32 + 8 * g4) =
                    half4{(_Float16)o[db][4 * g4], (_Float16)o[db][4 * g4 + 1], (_Float16)o[db][4 * g4 + 2], (_Float16)o[db][4 * g4 + 3]};
    }
}
template <int NQB, int HPW, int DH>
__global__ __launch_bounds__(64 * NQB * HPW) void attention_mfma_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ mask,
                                                                         int Lmax, int H, int heads, const int* __restrict__ cu,
                                                                         _Float16* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attention_mfma_body<NQB, HPW, DH>(qkv, mask, Lmax, H, heads, cu, ctx, (int)blockIdx.x, (int)threadIdx.x, smem);
}

// CLS pooling reads one row per sequence, and after the last layer's attention nothing mixes rows any more: the last layer's
// out-projection, FFN and LayerNorms then run on the B first-token rows alone.  This gathers those rows of the attention
// output and of the layer input (the residual) into two dense [B, H] arrays; one wave per sequence, 16 bytes per lane.
// (stats != nullptr: x holds raw rows whose LayerNorm is folded into the GEMMs -- the gathered residual rows are normalised here.)
__global__ __launch_bounds__(256) void gather_first_rows_kernel(const _Float16* __restrict__ ctx, const _Float16* __restrict__ x, int B,
                                                                int Lmax, int H, const int* __restrict__ cu,
                                                                _Float16* __restrict__ ctx_out, _Float16* __restrict__ x_out,
                                                                const float2* __restrict__ stats, int st_stride, int p,
                                                                const float* __restrict__ g, const float* __restrict__ b, float eps) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= B) return;
    const size_t row = cu ? (size_t)cu[seq] : (size_t)seq * Lmax;
    float mean = 0.f, rstd = 1.f;
    if (stats) row_stats(stats, (size_t)st_stride, row, p, lane, 1.0f / H, eps, mean, rstd);
    for (int j = lane * 8; j < H; j += 512) {  // H % 8 == 0 (checked at create)
        *reinterpret_cast<half8*>(ctx_out + (size_t)seq * H + j) = *reinterpret_cast<const half8*>(ctx + row * H + j);
        half8 v = *reinterpret_cast<const half8*>(x + row * H + j);
        if (stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)(((float)v[e] - mean) * rstd * g[j + e] + b[j + e]);
        }
        *reinterpret_cast<half8*>(x_out + (size_t)seq * H + j) = v;
    }
}

// create time: one weight matrix [N][K] (fp32, device) of a GEMM whose input is a LayerNorm output -> the folded form:
// Wf = fp16(W (.) gamma), cvec[n] = sum_k Wf[n][k] (of the fp16 values the MFMAs will multiply), bf[n] = bias[n] + sum_k W[n][k] beta[k].
// One wave per output feature.
__global__ __launch_bounds__(256) void fold_weight_kernel(const float* __restrict__ W, int N, int K, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ bias,
                                                          _Float16* __restrict__ Wf, float* __restrict__ cvec,
                                                          float* __restrict__ bf) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, bs = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        const _Float16 wf = (_Float16)(w * gamma[k]);
        Wf[(size_t)n * K + k] = wf;
        cs += (float)wf;
        bs += w * beta[k];
    }
    cs = wave_sum(cs);
    bs = wave_sum(bs);
    if (lane == 0) {
        cvec[n] = cs;
        bf[n] = bias[n] + bs;
    }
}

constexpr int kMaxPer = 32;  // hidden <= 2048 (pooling keeps element j = lane + 64 i per lane)

template <bool COH = false>
__device__ __forceinline__ void pool_normalize_body(const _Float16* __restrict__ hidden, const int* __restrict__ mask,
                                                    int B, int Lmax, int H, int pooling, int normalize,
                                                    const int* __restrict__ cu, float* __restrict__ out,
                                                    const float2* __restrict__ stats, int st_stride, int p,
                                                    const float* __restrict__ g, const float* __restrict__ b, float eps, int bx, int tid) {
    const int lane = tid & 63;
    const int seq = bx * 4 + (tid >> 6);
    if (seq >= B) return;
    const size_t row0 = cu ? (size_t)cu[seq] : (size_t)seq * Lmax;
    const int L = cu ? cu[seq + 1] - cu[seq] : Lmax;
    float x[kMaxPer];
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) x[i] = 0.f;
    // stats != nullptr: `hidden` holds raw rows whose LayerNorm is folded into the GEMMs (FoldArgs); the pooled rows are
    // normalised here: sum_l ((h_l - mean_l) rstd_l) gamma + cnt beta
    float mean = 0.f, rstd = 1.f;
    int cnt = 1;
    if (pooling == VQA_POOL_CLS) {
        if (stats) row_stats(stats, (size_t)st_stride, row0, p, lane, 1.0f / H, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int j = lane + 64 * i;
            if (j < H) x[i] = (act_load_half<COH>(hidden + row0 * H + j) - mean) * rstd;
        }
    } else {
        cnt = 0;
        for (int l = 0; l < L; ++l) {
            if (!cu && !mask[seq * Lmax + l]) continue;
            ++cnt;
            if (stats) row_stats(stats, (size_t)st_stride, row0 + l, p, lane, 1.0f / H, eps, mean, rstd);
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int j = lane + 64 * i;
                if (j < H) x[i] += (act_load_half<COH>(hidden + (row0 + l) * H + j) - mean) * rstd;
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int j = lane + 64 * i;
            if (j < H) x[i] = x[i] * g[j] + (float)cnt * b[j];
        }
    }
    if (pooling != VQA_POOL_CLS) {
        const float inv = 1.0f / fmaxf((float)cnt, 1e-9f);
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) x[i] *= inv;
    }
    float nrm = 1.0f;
    if (normalize) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) ss += x[i] * x[i];
        nrm = sqrtf(wave_sum(ss));
        if (!(nrm > 0.f)) nrm = 1.0f;
    }
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
        const int j = lane + 64 * i;
        if (j < H) out[(size_t)seq * H + j] = x[i] / nrm;
    }
}
__global__ __launch_bounds__(256) void pool_normalize_kernel(const _Float16* __restrict__ hidden, const int* __restrict__ mask,
                                                             int B, int Lmax, int H, int pooling, int normalize,
                                                             const int* __restrict__ cu, float* __restrict__ out,
                                                             const float2* __restrict__ stats, int st_stride, int p,
                                                             const float* __restrict__ g, const float* __restrict__ b, float eps) {
    pool_normalize_body(hidden, mask, B, Lmax, H, pooling, normalize, cu, out, stats, st_stride, p, g, b, eps, (int)blockIdx.x, (int)threadIdx.x);
}

// hidden states for a caller (vqa_encoder_forward_hidden = HF `output_hidden_states`): row t of the activation array (packed or padded)
// -> fp32 row (seq, l) of out [B, L, H]; stats != nullptr: x holds raw rows whose LayerNorm is folded into the GEMMs, applied here.
// One wave per row.
__global__ __launch_bounds__(256) void export_hidden_kernel(const _Float16* __restrict__ x, int T, int L, int H, const int* __restrict__ cu,
                                                            const int* __restrict__ row_seq, int B, const float2* __restrict__ stats,
                                                            int st_stride, int p, const float* __restrict__ g, const float* __restrict__ b,
                                                            float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    size_t dst = (size_t)t;
    if (cu) {
        if (t >= cu[B]) return;
        const int seq = row_seq[t];
        dst = (size_t)seq * L + (t - cu[seq]);
    }
    float mean = 0.f, rstd = 1.f;
    if (stats) row_stats(stats, (size_t)st_stride, (size_t)t, p, lane, 1.0f / H, eps, mean, rstd);
    for (int j = lane; j < H; j += 64) {
        float v = (float)x[(size_t)t * H + j];
        if (stats) v = (v - mean) * rstd * g[j] + b[j];
        out[dst * H + j] = v;
    }
}

// ---- ONE launch for the whole one-question forward (round 6: the reference's own call, heavy_ranker.py:97-101 -- one question, limit 1).
// The latency form above is 62 launches of ~5.8 us each (0.36-0.44 ms for 32 tokens), every one a chain of wave start -> one memory round
// trip -> LDS reduction -> store -> kernel boundary (write-back + invalidate of the L2s).  Here the same phases -- embedding, and per layer
// QKV (LayerNorm folded in) | attention | out-projection (+ normalised residual) | FFN1 (LayerNorm folded in, GELU) | FFN2 (+ normalised
// residual), then the last LayerNorm and the pooling -- run inside ONE persistent kernel: G resident workgroups (cooperative launch), the
// phases' units (16 output features x the token tiles; 4 heads of a sequence; 4 rows) dealt round-robin, a fence-free grid barrier between
// phases.  The bodies are the launches' own (gemm_tiny_body, attention_mfma_body, ...): same arithmetic, same bits.
// Coherence without fences (a device-scope release / acquire per workgroup writes back and invalidates an XCD's L2: 12-64 us per barrier,
// scripts/probes/grid_barrier_probe.hip): every ACTIVATION buffer the phases hand to each other lives in UNCACHED device memory
// (hipDeviceMallocUncached), so plain loads and stores go to the memory side and no L2 / L1 ever holds a stale line (probe: no stale read in
// 4 x 500 phases at 48-256 workgroups, 2.9-5.1 us per phase with a 64 KB exchange); the per-CU vector L1 is invalidated by every wave behind
// each barrier (it does cache uncached memory); weights and inputs are read-only and cached as usual.
// The barrier: every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup's barrier, ONE device-scope atomic add, one lane polls.
// Spins are bounded: a workgroup that waits longer than ~50 ms raises `abort` and every workgroup leaves (the host reports the failure and
// takes the launches instead) -- a stuck grid must never hang the device.
struct PersistLayer {
    const _Float16 *wqkv_f, *wo, *w1_f, *w2;
    const float *bqkv_f, *cqkv, *bo, *b1_f, *c1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};
struct PersistArgs {
    const int *ids, *mask;
    int B, L, T, H, F, heads, layers, pad_id, vocab, abs_pos;
    int* bad_ids;
    const float *word, *pos, *type0, *emb_g, *emb_b;
    float eps;
    const PersistLayer* lay;
    _Float16 *x, *qkv, *ctx, *tmp, *ffn;  // uncached
    float2 *tiny_x, *tiny_tmp;            // uncached
    float2* st_scratch;                   // the embedding's slot statistics (written, never read here)
    int st_stride;
    int pooling, normalize;
    float* out;
    unsigned* counter;  // grid barrier: monotonic over the handle's launches
    unsigned base;      // its value when this launch starts
    unsigned* abort;    // raised by a workgroup whose wait ran out
};

// barriers of one forward: behind the embedding, five per layer, behind the last LayerNorm (the host advances its copy of the counter by
// this many x the grid per launch)
__host__ __device__ constexpr unsigned persist_barriers(int layers) { return 2u + 5u * (unsigned)layers; }

__device__ __forceinline__ bool persist_barrier(const PersistArgs& a, unsigned phase, int tid) {
    __shared__ unsigned ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores of the phase have reached the memory side
    __syncthreads();
    if (tid == 0) {
        const unsigned target = (phase + 1u) * gridDim.x;
        __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned good = 0u;
        for (unsigned spin = 0; spin < (1u << 21); ++spin) {
            if (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.base >= target) {
                good = 1u;
                break;
            }
            if ((spin & 255u) == 255u && __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!good) __hip_atomic_store(a.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = good;
    }
    __syncthreads();
    // Uncached memory bypasses the L2s, NOT this CU's vector L1: a workgroup that read a buffer in an earlier phase and was idle since can
    // still hold those lines (seen at 256 workgroups with plain loads: the ones that only ever run FFN1 units re-read the previous layer's rows;
    // `buffer_inv sc0` does not drop them, `buffer_inv sc1` does -- at 5.6 us per barrier).  The phases read activations with device-scope atomic
    // loads instead (act_load16 / act_load8: `sc1`, never served from the L1).
#ifndef VQA_PERSIST_INV
#define VQA_PERSIST_INV 0  // dev: 1 / 2 = every wave invalidates the caches behind the barrier (5.6 us per barrier; the coherent loads make it unnecessary)
#endif
#if VQA_PERSIST_INV == 1
    asm volatile("buffer_inv sc1" ::: "memory");
#elif VQA_PERSIST_INV == 2
    asm volatile("buffer_inv sc0 sc1" ::: "memory");
#endif
    return ok != 0u;
}

// this wave's weight fragments of unit `unit` of a gemm_tiny_body<.., NBC, WV> call (its whole K range is one chunk), requested AHEAD of
// the grid barrier in front of that phase: weights do not depend on the phase before, so their memory round trip passes under the barrier
template <int NBC, int WV>
__device__ __forceinline__ void persist_prefetch(const _Float16* __restrict__ W, int K, int unit, int tid, half8 (&wf)[12]) {
    const int lane = tid & 63, wave = tid >> 6;
    if (wave >= WV) return;
    const int c = lane & 15, g = lane >> 4;
    const _Float16* wp = W + (size_t)(unit * 16 + c) * K + wave * (K / WV) + 8 * g;
#pragma unroll
    for (int j = 0; j < NBC; ++j) wf[j] = *reinterpret_cast<const half8*>(wp + 32 * j);
#pragma unroll
    for (int j = 0; j < NBC; ++j) asm volatile("" : "+v"(wf[j]));  // (requested here, not behind the barrier)
}

// MT: token tiles of 16 (T <= 16 MT); FOUR: hidden size 384 (K = 384: four waves split K, heads of 32); NQB: 32-token blocks per sequence
template <int MT, bool FOUR, int NQB>
__global__ __launch_bounds__(512) void encoder_persist_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WVA = FOUR ? 4 : 8;       // waves that split K = H
    constexpr int NBCB = FOUR ? 6 : 12;     // K-blocks per chunk of FFN2 (K = F: 1536 / 3072, eight waves)
    constexpr int DH = FOUR ? 32 : 64;
    constexpr int HPW = 4;
    const int tid = threadIdx.x, G = gridDim.x, wg = blockIdx.x;
    const int H = a.H, F = a.F, T = a.T;
    const float inv_h = 1.0f / (float)H;
    unsigned phase = 0;
    half8 wnext[12];  // the next phase's weight fragments of this workgroup's first unit
#pragma unroll
    for (int j = 0; j < 12; ++j) wnext[j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#define VQA_PERSIST_BARRIER()                     \
    do {                                          \
        if (!persist_barrier(a, phase, tid)) return; \
        ++phase;                                  \
    } while (0)
    // ---- embedding: raw rows (+ slot statistics nobody reads here); 4 rows per 256 threads, 8 per unit
    for (int u = wg; u < (T + 7) / 8; u += G) embed_ln_body(a.ids, T, a.L, H, a.pad_id, a.vocab, a.abs_pos, a.bad_ids, nullptr, nullptr, a.B, a.word, a.pos, a.type0,
                                                            a.emb_g, a.emb_b, a.eps, a.x, a.st_scratch, a.st_stride, 2 * u + (tid >> 8), tid & 255);
    const float *pg = a.emb_g, *pb = a.emb_b;
    for (int l = 0; l < a.layers; ++l) {
        const PersistLayer& Ly = a.lay[l];
        if (wg < 3 * H / 16) persist_prefetch<3, WVA>(Ly.wqkv_f, H, wg, tid, wnext);
        VQA_PERSIST_BARRIER();
        {   // QKV on the raw rows, LayerNorm folded in; leaves (mean, rstd) of every row
            TinyArgs t;
            t.cvec = Ly.cqkv;
            t.st_out = a.tiny_x;
            t.inv_k = inv_h;
            t.eps = a.eps;
            if (wg < 3 * H / 16) {  // this workgroup's first unit: its weights were requested in front of the barrier
                const int u = wg;
                gemm_tiny_body<0, MT, 1, 3, WVA, true>(a.x, Ly.wqkv_f, Ly.bqkv_f, nullptr, a.qkv, T, 3 * H, H, t, u, 0, tid, wnext);
                __syncthreads();
            }
            for (int u = wg + G; u < 3 * H / 16; u += G) {
                gemm_tiny_body<0, MT, 1, 3, WVA, true>(a.x, Ly.wqkv_f, Ly.bqkv_f, nullptr, a.qkv, T, 3 * H, H, t, u, 0, tid, nullptr);
                __syncthreads();
            }
        }
        VQA_PERSIST_BARRIER();
        for (int u = wg; u < a.B * (a.heads / HPW); u += G) {
            attention_mfma_body<NQB, HPW, DH, true>(a.qkv, a.mask, a.L, H, a.heads, nullptr, a.ctx, u, tid, smem);
            __syncthreads();
        }
        if (wg < H / 16) persist_prefetch<3, WVA>(Ly.wo, H, wg, tid, wnext);
        VQA_PERSIST_BARRIER();
        {   // out-projection + LN_prev(x) as the residual -> tmp (raw)
            TinyArgs t;
            t.st_in = a.tiny_x;
            t.g = pg;
            t.b = pb;
            if (wg < H / 16) {  // this workgroup's first unit: its weights were requested in front of the barrier
                const int u = wg;
                gemm_tiny_body<2, MT, 0, 3, WVA, true>(a.ctx, Ly.wo, Ly.bo, a.x, a.tmp, T, H, H, t, u, 0, tid, wnext);
                __syncthreads();
            }
            for (int u = wg + G; u < H / 16; u += G) {
                gemm_tiny_body<2, MT, 0, 3, WVA, true>(a.ctx, Ly.wo, Ly.bo, a.x, a.tmp, T, H, H, t, u, 0, tid, nullptr);
                __syncthreads();
            }
        }
        if (wg < F / 16) persist_prefetch<3, WVA>(Ly.w1_f, H, wg, tid, wnext);
        VQA_PERSIST_BARRIER();
        {   // FFN1 on tmp with LN1 folded in, GELU; leaves (mean, rstd) of tmp's rows
            TinyArgs t;
            t.cvec = Ly.c1;
            t.st_out = a.tiny_tmp;
            t.inv_k = inv_h;
            t.eps = a.eps;
            if (wg < F / 16) {  // this workgroup's first unit: its weights were requested in front of the barrier
                const int u = wg;
                gemm_tiny_body<1, MT, 1, 3, WVA, true>(a.tmp, Ly.w1_f, Ly.b1_f, nullptr, a.ffn, T, F, H, t, u, 0, tid, wnext);
                __syncthreads();
            }
            for (int u = wg + G; u < F / 16; u += G) {
                gemm_tiny_body<1, MT, 1, 3, WVA, true>(a.tmp, Ly.w1_f, Ly.b1_f, nullptr, a.ffn, T, F, H, t, u, 0, tid, nullptr);
                __syncthreads();
            }
        }
        const int nx = H / 16, my = (T + 15) / 16;
        if (wg < nx * my) persist_prefetch<NBCB, 8>(Ly.w2, F, wg % nx, tid, wnext);
        VQA_PERSIST_BARRIER();
        {   // FFN2 + LN1(tmp) as the residual -> x (raw); one token tile per unit
            TinyArgs t;
            t.st_in = a.tiny_tmp;
            t.g = Ly.ln1_g;
            t.b = Ly.ln1_b;
            if (wg < nx * my) {  // this workgroup's first unit: its weights were requested in front of the barrier
                const int u = wg;
                gemm_tiny_body<2, 1, 0, NBCB, 8, true>(a.ffn, Ly.w2, Ly.b2, a.tmp, a.x, T, H, F, t, u % nx, u / nx, tid, wnext);
                __syncthreads();
            }
            for (int u = wg + G; u < nx * my; u += G) {
                gemm_tiny_body<2, 1, 0, NBCB, 8, true>(a.ffn, Ly.w2, Ly.b2, a.tmp, a.x, T, H, F, t, u % nx, u / nx, tid, nullptr);
                __syncthreads();
            }
        }
        pg = Ly.ln2_g;
        pb = Ly.ln2_b;
    }
    VQA_PERSIST_BARRIER();
    // ---- the last LayerNorm (x raw -> tmp), then pooling + L2 normalisation
    for (int u = wg; u < (T + 7) / 8; u += G) ln_body<true>(a.x, T, H, pg, pb, a.eps, a.tmp, 2 * u + (tid >> 8), tid & 255);
    VQA_PERSIST_BARRIER();
    for (int u = wg; u < (a.B + 7) / 8; u += G)
        pool_normalize_body<true>(a.tmp, a.mask, a.B, a.L, H, a.pooling, a.normalize, nullptr, a.out, nullptr, 0, 0, nullptr, nullptr, a.eps, 2 * u + (tid >> 8), tid & 255);
    if (phase != persist_barriers(a.layers) && wg == 0 && tid == 0)  // (the host's bookkeeping and this kernel disagree: never use the handle's counter again)
        __hip_atomic_store(a.abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#undef VQA_PERSIST_BARRIER
}

}  // namespace

struct vqa_encoder {
    int device = 0;
    vqa_encoder_config cfg{};
    int max_tokens = 0;
    // fp32 parameters
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    struct Layer {
        _Float16 *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
        float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
        float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
        // LayerNorm folded into the GEMM that reads its output (FoldArgs): weights scaled by the gamma of the LayerNorm in
        // front (QKV: the previous layer's second one, or the embedding's; FFN1: this layer's first), their column sums, bias + W beta
        _Float16 *wqkv_f = nullptr, *w1_f = nullptr;
        float *cqkv = nullptr, *c1 = nullptr, *bqkv_f = nullptr, *b1_f = nullptr;
    };
    std::vector<Layer> layers;
    bool first_rows_on = true;  // CLS pooling: last layer past the attention on the first-token rows only (encoder_launch)
    std::vector<void*> allocs;
    // activations
    _Float16 *x = nullptr, *qkv = nullptr, *ctx = nullptr, *tmp = nullptr, *ffn = nullptr;
    float2 *st_x = nullptr, *st_tmp = nullptr;  // folded LayerNorms: the statistics slots of the raw rows in x / tmp, [rows][16]
    float2 *tiny_x = nullptr, *tiny_tmp = nullptr;  // the latency form (gemm_tiny_kernel): (mean, rstd) of the raw rows in x / tmp, [64]
    bool tiny_on = true;                            // vqa_encoder_options.latency_path
    bool fold_on = true;                        // VQA_ENC_FOLD=0 at create: dev / test switch
    // small batches are launch-bound (12 layers x 7 kernels of a few microseconds each): their launch sequence is captured
    // once per (B, L, pooling, normalize) into a hipGraph over these fixed staging buffers and replayed
    int32_t *cu = nullptr, *row_seq = nullptr;    // sequence packing: [max_tokens + 1] row offsets, [max_tokens] packed row -> sequence
    int32_t *g_ids = nullptr, *g_mask = nullptr;  // [max_tokens]
    float* g_out = nullptr;                       // [max_tokens, hidden] (B <= max_tokens)
    struct Graph {
        int B, L, pooling, normalize;
        hipGraphExec_t exec;
        bool seen_once;  // the first call of a shape runs eagerly (lazy hipFuncSetAttribute calls are not capturable)
    };
    std::vector<Graph> graphs;
    hipStream_t cap_stream = nullptr;  // capture happens on a stream of our own (the caller's may be the null stream, which
                                       // cannot be captured); the instantiated graph is launched on the caller's stream
    bool use_graphs = true;
    int32_t* stage_host = nullptr;  // vqa_encoder_forward_host: pinned, device-mapped [2][max_tokens] token ids | masks
    int32_t* stage_dev = nullptr;   // ... its device alias
    hipEvent_t stage_done = nullptr;  // recorded behind the launches of the last forward_host call: the staging buffer is free again
    bool stage_pending = false;
    // the one-launch forward of one question (encoder_persist_kernel): activations in UNCACHED device memory, the layers' pointers as a
    // device array, the grid barrier's counter and its host-side value, the abort flag (pinned, device-mapped)
    struct Persist {
        bool on = false;             // options.persistent and a model / device the kernel serves
        int grid = 0;                // resident workgroups
        _Float16 *x = nullptr, *qkv = nullptr, *ctx = nullptr, *tmp = nullptr, *ffn = nullptr;
        float2 *tiny_x = nullptr, *tiny_tmp = nullptr;
        PersistLayer* layers = nullptr;
        unsigned* counter = nullptr;
        unsigned base = 0;           // the counter's value when the next launch starts (phases x grid per completed launch)
        unsigned* abort_host = nullptr;
        unsigned* abort_dev = nullptr;
    } persist;
    int* bad_ids_host = nullptr;  // pinned, device-mapped: set by embed_ln when a token id lies outside [0, vocab_size)
    int* bad_ids_dev = nullptr;
    std::atomic_flag busy = ATOMIC_FLAG_INIT;  // one forward at a time per handle (staging buffers and graphs are shared)
};

namespace {

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(dev);
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int dev_alloc(vqa_encoder* e, void** p, size_t bytes) {
    if (hipMalloc(p, bytes ? bytes : 1) != hipSuccess) {
        vqa_set_error("vqa_encoder_create: hipMalloc of %zu bytes failed", bytes);
        return VQA_ENOMEM;
    }
    e->allocs.push_back(*p);
    return VQA_OK;
}

// fp32 source (host or device) -> fp32 device copy
int upload_f32(vqa_encoder* e, const float* src, size_t n, float** out) {
    VQA_REQUIRE(src, "vqa_encoder_create: a weight pointer is null");
    int rc = dev_alloc(e, (void**)out, n * 4);
    if (rc != VQA_OK) return rc;
    VQA_HIP_CHECK(hipMemcpy(*out, src, n * 4, hipMemcpyDefault));
    return VQA_OK;
}

// fp32 source -> fp16 device copy at dst (element offset), through a temporary fp32 device buffer
int upload_f16(vqa_encoder* e, const float* src, size_t n, _Float16* dst) {
    VQA_REQUIRE(src, "vqa_encoder_create: a weight pointer is null");
    float* tmp = nullptr;
    if (hipMalloc((void**)&tmp, n * 4) != hipSuccess) {
        vqa_set_error("vqa_encoder_create: staging hipMalloc of %zu bytes failed", n * 4);
        return VQA_ENOMEM;
    }
    hipError_t err = hipMemcpy(tmp, src, n * 4, hipMemcpyDefault);
    int rc = VQA_OK;
    if (err == hipSuccess) {
        const int32_t chunk = 256;  // element-wise conversion, no normalisation: rows of 256 elements (n % 32 == 0)
        const int64_t rows = (int64_t)(n / chunk);
        if (rows) rc = vqa_normalize_convert(tmp, rows, chunk, 0, VQA_F16, dst, nullptr);
        if (rc == VQA_OK && n % chunk)
            rc = vqa_normalize_convert(tmp + rows * chunk, 1, (int32_t)(n % chunk), 0, VQA_F16, dst + rows * chunk, nullptr);
        if (rc == VQA_OK) err = hipDeviceSynchronize();
    }
    (void)hipFree(tmp);
    if (rc != VQA_OK) return rc;
    if (err != hipSuccess) {
        vqa_set_error("vqa_encoder_create: weight conversion failed: %s", hipGetErrorString(err));
        return VQA_EHIP;
    }
    return VQA_OK;
}

// fp32 source [N][K] -> the LayerNorm-folded form on the device (fold_weight_kernel), through a temporary fp32 device buffer
int upload_folded(const float* src, int N, int K, const float* gamma, const float* beta, const float* bias, _Float16* wf, float* cvec,
                  float* bf) {
    VQA_REQUIRE(src, "vqa_encoder_create: a weight pointer is null");
    float* tmp = nullptr;
    const size_t bytes = (size_t)N * K * 4;
    if (hipMalloc((void**)&tmp, bytes) != hipSuccess) {
        vqa_set_error("vqa_encoder_create: staging hipMalloc of %zu bytes failed", bytes);
        return VQA_ENOMEM;
    }
    hipError_t err = hipMemcpy(tmp, src, bytes, hipMemcpyDefault);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(fold_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, 0, tmp, N, K, gamma, beta, bias, wf, cvec, bf);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipDeviceSynchronize();
    }
    (void)hipFree(tmp);
    if (err != hipSuccess) {
        vqa_set_error("vqa_encoder_create: folding a LayerNorm into a weight matrix failed: %s", hipGetErrorString(err));
        return VQA_EHIP;
    }
    return VQA_OK;
}

constexpr int kTokenPad = 256;  // activation buffers are padded to this many rows (the tallest tile)

// The FFN1 epilogue's GELU table of this device, built once per device at encoder create (built lazily inside the first launch of a
// shape it sat on the NULL stream with a synchronise -- inside a caller's stream capture that invalidates the capture).
int ensure_gelu_table() {
    static VqaPerDeviceOnce once;
    return once.run([&](int) -> int {
        if (VQA_GELU_TABLE) {
            hipLaunchKernelGGL(gelu_table_kernel, dim3(kGeluN / 256), dim3(256), 0, nullptr);
            VQA_HIP_CHECK(hipGetLastError());
            VQA_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
        return VQA_OK;
    });
}

template <int EPI, int BM, int BN, int WM, int WN, int BK, int FOLD = 0, int ONEBAR = 0, int TIL = 0>
int launch_tile(const _Float16* A, const _Float16* W, const float* bias, const _Float16* R, _Float16* C, int M, int N, int K,
                int num_cu, hipStream_t s, const FoldArgs& fa = FoldArgs{}) {
    using G = TileGeom<BM, BN, BK, (EPI == 1 && VQA_GELU_TABLE) ? kGeluBytes : 0>;
    static VqaPerDeviceOnce once;
    int rc = once.run([&](int) -> int {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<EPI, BM, BN, WM, WN, BK, FOLD, ONEBAR, TIL>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds));
        return VQA_OK;  // (the device's GELU table is built by vqa_encoder_create_ex: ensure_gelu_table)
    });
    if (rc != VQA_OK) return rc;
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM, tiles = tiles_n * tiles_m;
    const int grid = tiles < num_cu ? tiles : num_cu;
    // blocked tile order inside an XCD (see tile_of in the kernel): needs the XCD split itself (tiles, grid multiples of 8),
    // whole token-tile rows per XCD, and a block of per = grid / 8 tiles that divides them
    static const int nb_force = vqa_dev_env("VQA_GEMM_NB") ? atoi(vqa_dev_env("VQA_GEMM_NB")) : -1;  // dev override; 0 = feature-fastest order
    int nb = 0;
    if (tiles % 8 == 0 && grid % 8 == 0 && tiles_m % 8 == 0) {
        const int per = grid / 8, rows = tiles_m / 8;
        for (int cand = 8; cand >= 1 && nb == 0; --cand)
            if (tiles_n % cand == 0 && per % cand == 0 && rows % (per / cand) == 0) nb = cand;
        if (nb_force >= 0 && (nb_force == 0 || (tiles_n % nb_force == 0 && per % nb_force == 0 && rows % (per / nb_force) == 0)))
            nb = nb_force;
    }
    hipLaunchKernelGGL((gemm_tile_kernel<EPI, BM, BN, WM, WN, BK, FOLD, ONEBAR, TIL>), dim3(grid), dim3(512), G::kLds, s, A, W, bias, R, C, M, N,
                       K, tiles_n, tiles, nb, fa);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

// LDS-DMA bytes per CU (in units of 64 KT bytes) of a tile shape on this problem: rounds of tiles x (BM + BN)
static long tile_cost(int M, int N, int BM, int BN, int num_cu) {
    if (N % BN) return -1;
    const long tiles = (long)(N / BN) * ((M + BM - 1) / BM);
    return (tiles + num_cu - 1) / num_cu * (BM + BN);
}

// Token counts: up to kSkinnyMax the skinny kernel (it splits K over its waves: 0.47 ms per 12-layer forward at 32 tokens,
// 0.79 at 256), above it the LDS-DMA tile kernel, whose forward takes a flat ~0.99 ms up to 1024 tokens (a tile walks all of
// K at ~0.9 us per K-step however few rows it has); measured crossover 320 tokens (B = 10 at L = 32).
constexpr int kGraphMaxTokens = 1024;  // calls up to this many positions replay a hipGraph (vqa_encoder_forward)
constexpr int kSkinnyMax = 320;
constexpr int kTileMinM = kSkinnyMax + 1;

// Tile shapes: {BM, BN, BK, WN}.  The cost model is LDS-DMA bytes per CU (rounds of tiles x (BM + BN)): what the K loop is
// bound by; K-steps of 32 halves pay two barriers per 32-deep step, so a 64-deep shape wins a near tie (x 0.85).
static const int kTileShapes[7][4] = {{256, 288, 32, 2}, {256, 192, 32, 2}, {256, 128, 64, 2}, {128, 192, 64, 4}, {256, 128, 32, 2},
                                      {256, 256, 32, 2}, {128, 128, 64, 2}};

// index into kTileShapes of the LDS-DMA tile kernel's shape for this problem; -1: the problem does not take that kernel
// (fewer than 1024 rows, or no shape divides N / K); -2: HIP error (message set)
static int tile_choice(int M, int N, int K, int* num_cu_out) {
    static const int force_tile = vqa_dev_env("VQA_GEMM_TILE") ? atoi(vqa_dev_env("VQA_GEMM_TILE")) : -1;  // dev override: shape index 0..6
    static const int tile_min_m = vqa_dev_env("VQA_TILE_MIN_M") ? atoi(vqa_dev_env("VQA_TILE_MIN_M")) : kTileMinM;  // dev override
    if (!(M >= tile_min_m && N % 64 == 0 && K % 32 == 0)) return -1;
    static VqaPerDeviceOnce once;
    static int num_cu[64] = {};  // written inside the once, read after it
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        vqa_set_error("hipGetDevice failed");
        return -2;
    }
    int rc = once.run([&](int d) -> int {
        hipDeviceProp_t prop;
        VQA_HIP_CHECK(hipGetDeviceProperties(&prop, d));
        num_cu[d & 63] = prop.multiProcessorCount;
        return VQA_OK;
    });
    if (rc != VQA_OK) return -2;
    const int cu = num_cu[dev & 63];
    *num_cu_out = cu;
    int best = -1;
    double best_cost = 0;
    for (int i = 0; i < 7; ++i) {
        if (i == 4) continue;  // 256 x 128 with 32-deep K-steps: forced only (the 64-deep form of the same tile always wins)
        const long cst = K % kTileShapes[i][2] ? -1 : tile_cost(M, N, kTileShapes[i][0], kTileShapes[i][1], cu);
        const double w = cst * (kTileShapes[i][2] == 64 ? 0.85 : 1.0);
        if (cst >= 0 && (best < 0 || w < best_cost)) {
            best = i;
            best_cost = w;
        }
    }
    if (force_tile >= 0 && force_tile < 7 && K % kTileShapes[force_tile][2] == 0 &&
        tile_cost(M, N, kTileShapes[force_tile][0], kTileShapes[force_tile][1], cu) >= 0)
        best = force_tile;
    return best;
}

// statistics slots per row a FOLD EPI 2 GEMM of this shape writes: N / (BN / WN); 0: the problem does not take the tile kernel
static int tile_stat_slots(int M, int N, int K) {
    int cu = 0;
    const int best = tile_choice(M, N, K, &cu);
    return best < 0 ? 0 : N / (kTileShapes[best][1] / kTileShapes[best][3]);
}

// the LayerNorm-folded forms (FoldArgs above); the caller checked with tile_stat_slots that the problem takes the tile kernel
template <int EPI>
int launch_gemm_fold(const _Float16* A, const _Float16* W, const float* bias, const _Float16* R, _Float16* C, int M, int N, int K,
                     const FoldArgs& fa, hipStream_t s) {
    int cu = 0;
    const int best = tile_choice(M, N, K, &cu);
    // the 128 x 128 tile (64-deep K-steps: the packed N = 768 projections) runs the one-barrier K loop -- its two fragment sets fit
    // beside 64 accumulators and the folded epilogue without spilling (VQA_GEMM_ONEBAR=0: the slot loop, dev / A-B switch)
    static const bool onebar6 = !(vqa_dev_env("VQA_GEMM_ONEBAR") && atoi(vqa_dev_env("VQA_GEMM_ONEBAR")) == 0);
    switch (best) {
        case 0: return launch_tile<EPI, 256, 288, 4, 2, 32, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        case 1: return launch_tile<EPI, 256, 192, 4, 2, 32, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        case 2: return launch_tile<EPI, 256, 128, 4, 2, 64, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        case 3: return launch_tile<EPI, 128, 192, 2, 4, 64, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        case 4: return launch_tile<EPI, 256, 128, 4, 2, 32, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        case 5: return launch_tile<EPI, 256, 256, 4, 2, 32, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        case 6:
            if (onebar6) return launch_tile<EPI, 128, 128, 4, 2, 64, 1, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
            return launch_tile<EPI, 128, 128, 4, 2, 64, 1>(A, W, bias, R, C, M, N, K, cu, s, fa);
        default: break;
    }
    if (best != -2) vqa_set_error("launch_gemm_fold: no tile shape for M=%d N=%d K=%d", M, N, K);
    return best == -2 ? VQA_EHIP : VQA_EINVAL;
}

template <int EPI>
int launch_gemm(const _Float16* A, const _Float16* W, const float* bias, const _Float16* R, _Float16* C, int M, int N, int K,
                hipStream_t s) {
    static const bool force_small = vqa_dev_env("VQA_GEMM_SMALL") != nullptr;  // dev override, read once
    static const int skinny_max = vqa_dev_env("VQA_SKINNY_MAX") ? atoi(vqa_dev_env("VQA_SKINNY_MAX")) : kSkinnyMax;  // dev override
    if (M <= skinny_max && N % 16 == 0 && K % 256 == 0 && !force_small) {
        const int mt = M >= 64 ? 4 : (M + 15) / 16;
        const int chunks = (M + 16 * mt - 1) / (16 * mt);
#define VQA_SKINNY(MT)                                                                                                          \
    case MT:                                                                                                                    \
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, MT>), dim3(N / 16, chunks), dim3(256), 0, s, A, W, bias, R, C, M, N, K);     \
        break;
        switch (mt) {
            VQA_SKINNY(1) VQA_SKINNY(2) VQA_SKINNY(3) VQA_SKINNY(4)
        }
#undef VQA_SKINNY
        VQA_HIP_CHECK(hipGetLastError());
        return VQA_OK;
    }
    if (!force_small) {
        int cu = 0;
        const int best = tile_choice(M, N, K, &cu);
        if (best == -2) return VQA_EHIP;
        switch (best) {
            case 0: return launch_tile<EPI, 256, 288, 4, 2, 32>(A, W, bias, R, C, M, N, K, cu, s);
            case 1: return launch_tile<EPI, 256, 192, 4, 2, 32>(A, W, bias, R, C, M, N, K, cu, s);
            case 2: return launch_tile<EPI, 256, 128, 4, 2, 64>(A, W, bias, R, C, M, N, K, cu, s);
            case 3: return launch_tile<EPI, 128, 192, 2, 4, 64>(A, W, bias, R, C, M, N, K, cu, s);
            case 4: return launch_tile<EPI, 256, 128, 4, 2, 32>(A, W, bias, R, C, M, N, K, cu, s);
            case 5: return launch_tile<EPI, 256, 256, 4, 2, 32>(A, W, bias, R, C, M, N, K, cu, s);
            case 6: return launch_tile<EPI, 128, 128, 4, 2, 64>(A, W, bias, R, C, M, N, K, cu, s);
            default: break;  // no shape divides N / K: the register-staged kernel below
        }
    }
    dim3 grid((N + kGemmBN - 1) / kGemmBN, (M + kGemmBM - 1) / kGemmBM);
    if (K % 64 == 0)
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, 64>), grid, dim3(256), 0, s, A, W, bias, R, C, M, N, K);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, 32>), grid, dim3(256), 0, s, A, W, bias, R, C, M, N, K);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

// the latency form's shapes: M <= 64 rows, N a multiple of 16, K a multiple of 768 (eight waves, K / 256 blocks each in chunks of 3, 6 or
// 12) or of 384 (four waves, K / 128 blocks each in chunks of 3: hidden size 384, paraphrase-multilingual-MiniLM-L12-v2)
constexpr int kTinyMaxM = 64;
static bool tiny_shape(int M, int N, int K) { return M >= 1 && M <= kTinyMaxM && N % 16 == 0 && K % 384 == 0; }

template <int EPI, int FOLDIN>
int launch_gemm_tiny(const _Float16* A, const _Float16* W, const float* bias, const _Float16* R, _Float16* C, int M, int N, int K,
                     const TinyArgs& ta, hipStream_t s) {
    // a GEMM of few feature slices and a long K (FFN2: 48 slices x K = 3072; 24 x 1536) spreads its token tiles over workgroups: every
    // workgroup then pulls one tile's rows of A (98 KB instead of 196) beside its 98 KB of weights (the weights are re-read from the L2)
    const int my = (!FOLDIN && N / 16 < 96 && K >= 1536 && K >= 4 * N) ? (M + 15) / 16 : 1;
    const int mt = my > 1 ? 1 : (M + 15) / 16;
    const bool four = K % 768 != 0;
    const int nbw = four ? K / 128 : K / 256;
    // all of a wave's loads at once when they fit its registers (NBC (1 + MT) fragments of 4 registers), else in two or more chunks
    const int nbc = four ? 3 : (nbw % 12 == 0 && mt <= 2) ? 12 : (nbw % 6 == 0 && mt <= 4) ? 6 : 3;
#define VQA_TINY(MT, NBC, WV)                                                                                                          \
    hipLaunchKernelGGL((gemm_tiny_kernel<EPI, MT, FOLDIN, NBC, WV>), dim3(N / 16, my), dim3(64 * WV), 0, s, A, W, bias, R, C, M, N, K, ta)
#define VQA_TINY_MT(NBC, WV)                                   \
    switch (mt) {                                              \
        case 1: VQA_TINY(1, NBC, WV); break;                   \
        case 2: VQA_TINY(2, NBC, WV); break;                   \
        case 3: VQA_TINY(3, NBC, WV); break;                   \
        default: VQA_TINY(4, NBC, WV); break;                  \
    }
    if (four) {
        VQA_TINY_MT(3, 4)
    } else if (nbc == 12) {
        if (mt == 1) VQA_TINY(1, 12, 8);
        else VQA_TINY(2, 12, 8);
    } else if (nbc == 6) {
        VQA_TINY_MT(6, 8)
    } else {
        VQA_TINY_MT(3, 8)
    }
#undef VQA_TINY
#undef VQA_TINY_MT
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

}  // namespace

#ifdef VQA_DEV
// dev-only entry points (scripts/gemm_bench.py; built with -DVQA_DEV into a variant library, never part of the product build):
// one GEMM of the encoder through a chosen tile shape (index into kTileShapes, -1 = the launcher's own choice)
extern "C" int vqa_dev_gemm(const void* A, const void* W, const float* bias, const void* R, void* C, int M, int N, int K, int epi,
                            int shape, void* stream) {
    const _Float16 *a = (const _Float16*)A, *w = (const _Float16*)W, *r = (const _Float16*)R;
    _Float16* c = (_Float16*)C;
    hipStream_t s = (hipStream_t)stream;
    if (int grc = ensure_gelu_table(); grc != VQA_OK) return grc;
    hipDeviceProp_t prop;
    int dev = 0;
    VQA_HIP_CHECK(hipGetDevice(&dev));
    VQA_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const int cu = prop.multiProcessorCount;
#define VQA_DEV_SHAPE(E)                                                                                \
    switch (shape) {                                                                                    \
        case 0: return launch_tile<E, 256, 288, 4, 2, 32>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 1: return launch_tile<E, 256, 192, 4, 2, 32>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 2: return launch_tile<E, 256, 128, 4, 2, 64>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 3: return launch_tile<E, 128, 192, 2, 4, 64>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 4: return launch_tile<E, 256, 128, 4, 2, 32>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 5: return launch_tile<E, 256, 256, 4, 2, 32>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 6: return launch_tile<E, 128, 128, 4, 2, 64>(a, w, bias, r, c, M, N, K, cu, s);            \
        case 7: return launch_tile<E, 256, 192, 4, 2, 32, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);      \
        case 8: return launch_tile<E, 256, 256, 4, 2, 32, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);      \
        case 9: return launch_tile<E, 256, 128, 4, 2, 64, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);      \
        case 10: return launch_tile<E, 128, 192, 2, 4, 64, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);     \
        case 11: return launch_tile<E, 128, 128, 4, 2, 64, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);     \
        case 12: return launch_tile<E, 256, 288, 4, 2, 32, 0, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);  \
        case 13: return launch_tile<E, 256, 192, 4, 2, 32, 0, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);  \
        case 14: return launch_tile<E, 256, 128, 4, 2, 64, 0, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);  \
        case 15: return launch_tile<E, 128, 192, 2, 4, 64, 0, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);  \
        case 16: return launch_tile<E, 256, 256, 4, 2, 32, 0, 0, 1>(a, w, bias, r, c, M, N, K, cu, s);  \
        case 17: return launch_tile<E, 128, 128, 4, 2, 64, 0, 1, 1>(a, w, bias, r, c, M, N, K, cu, s);  \
        default: return launch_gemm<E>(a, w, bias, r, c, M, N, K, s);                                   \
    }
    if (epi == 0) { VQA_DEV_SHAPE(0) }
    if (epi == 1) { VQA_DEV_SHAPE(1) }
    VQA_DEV_SHAPE(2)
#undef VQA_DEV_SHAPE
}
#ifdef VQA_GSTAMPS
extern "C" int vqa_dev_read_gstamps(unsigned long long* out, int n) {
    const size_t bytes = sizeof(unsigned long long) * (size_t)(n < 8 * 64 * kGStampSlots ? n : 8 * 64 * kGStampSlots);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gstamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
#endif

extern "C" void vqa_encoder_destroy(vqa_encoder* e) {
    if (!e) return;
    DevGuard g(e->device);
    for (void* p : e->allocs) (void)hipFree(p);
    for (auto& gr : e->graphs)
        if (gr.exec) (void)hipGraphExecDestroy(gr.exec);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->bad_ids_host) (void)hipHostFree(e->bad_ids_host);
    if (e->persist.abort_host) (void)hipHostFree(e->persist.abort_host);
    for (void* q : {(void*)e->persist.x, (void*)e->persist.qkv, (void*)e->persist.ctx, (void*)e->persist.tmp, (void*)e->persist.ffn, (void*)e->persist.tiny_x,
                    (void*)e->persist.tiny_tmp, (void*)e->persist.layers, (void*)e->persist.counter})
        if (q) (void)hipFree(q);
    if (e->stage_host) (void)hipHostFree(e->stage_host);
    if (e->stage_done) (void)hipEventDestroy(e->stage_done);
    delete e;
}

extern "C" void vqa_encoder_options_init(vqa_encoder_options* o) {
    if (!o) return;
    o->struct_size = (uint32_t)sizeof(vqa_encoder_options);
    o->fold_layernorm = 1;
    o->first_rows = 1;
    o->graphs = 1;
    o->latency_path = 1;
    o->persistent = 0;  // (built, bit-equal to the launches, and 2.2x SLOWER than them: profiles/r06_persistent_forward.txt)
    o->persistent_grid = 0;
}

extern "C" int vqa_encoder_create(vqa_encoder** out, int device, const vqa_encoder_config* cfg, const vqa_encoder_weights* w,
                                  int32_t max_tokens) {
    return vqa_encoder_create_ex(out, device, cfg, w, max_tokens, nullptr);
}

extern "C" int vqa_encoder_create_ex(vqa_encoder** out, int device, const vqa_encoder_config* cfg, const vqa_encoder_weights* w,
                                     int32_t max_tokens, const vqa_encoder_options* opt) {
    VQA_REQUIRE(out, "vqa_encoder_create: out is null");
    *out = nullptr;
    vqa_encoder_options o;
    vqa_encoder_options_init(&o);
    if (opt) {
        VQA_REQUIRE(opt->struct_size >= 8 && opt->struct_size <= 4096, "vqa_encoder_create_ex: options.struct_size=%u (vqa_encoder_options_init sets it)",
                    opt->struct_size);
        memcpy(&o, opt, opt->struct_size < sizeof(o) ? opt->struct_size : sizeof(o));
        o.struct_size = (uint32_t)sizeof(o);
    }
    if (const char* v = vqa_dev_env("VQA_ENC_FIRST_ROWS")) o.first_rows = atoi(v) != 0;
    if (const char* v = vqa_dev_env("VQA_ENC_FOLD")) o.fold_layernorm = atoi(v) != 0;
    if (const char* v = vqa_dev_env("VQA_ENCODER_GRAPH")) o.graphs = v[0] != '0';
    if (const char* v = vqa_dev_env("VQA_ENC_TINY")) o.latency_path = v[0] != '0';
    if (const char* v = vqa_dev_env("VQA_ENC_PERSIST")) o.persistent = v[0] != '0';
    if (const char* v = vqa_dev_env("VQA_ENC_PERSIST_GRID")) o.persistent_grid = atoi(v);
    VQA_REQUIRE(cfg && w && w->layer, "vqa_encoder_create: null config / weights");
    VQA_REQUIRE(cfg->hidden >= 32 && cfg->hidden <= 2048 && cfg->hidden % 32 == 0,
                "vqa_encoder_create: hidden=%d must be a multiple of 32 in [32, 2048]", cfg->hidden);
    VQA_REQUIRE(cfg->ffn >= 32 && cfg->ffn % 32 == 0, "vqa_encoder_create: ffn=%d must be a multiple of 32", cfg->ffn);
    VQA_REQUIRE(cfg->heads >= 1 && cfg->hidden % cfg->heads == 0 && cfg->hidden / cfg->heads <= 128,
                "vqa_encoder_create: heads=%d does not divide hidden=%d into head sizes <= 128", cfg->heads, cfg->hidden);
    VQA_REQUIRE(cfg->layers >= 1 && cfg->vocab_size >= 1 && cfg->max_pos >= 2 && cfg->type_vocab >= 1 && max_tokens >= 1,
                "vqa_encoder_create: bad sizes");
    VQA_REQUIRE(cfg->pad_id >= 0 && cfg->pad_id < cfg->vocab_size, "vqa_encoder_create: pad_id=%d", cfg->pad_id);
    VQA_REQUIRE(cfg->position_ids == VQA_POS_ROBERTA || cfg->position_ids == VQA_POS_ABSOLUTE, "vqa_encoder_create: position_ids=%d",
                cfg->position_ids);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        vqa_set_error("vqa_encoder_create: no HIP device visible");
        return VQA_ENODEV;
    }
    VQA_REQUIRE(device >= 0 && device < ndev, "vqa_encoder_create: device %d of %d", device, ndev);
    DevGuard guard(device);
    if (int grc = ensure_gelu_table(); grc != VQA_OK) return grc;
    vqa_encoder* e = new (std::nothrow) vqa_encoder();
    if (!e) {
        vqa_set_error("vqa_encoder_create: host allocation failed");
        return VQA_ENOMEM;
    }
    e->device = device;
    e->cfg = *cfg;
    e->max_tokens = max_tokens;
    e->first_rows_on = o.first_rows != 0;
    e->fold_on = o.fold_layernorm != 0;
    e->tiny_on = o.latency_path != 0;
    const size_t H = cfg->hidden, F = cfg->ffn;
    int rc = VQA_OK;
    do {
        if ((rc = upload_f32(e, w->word_emb, (size_t)cfg->vocab_size * H, &e->word)) != VQA_OK) break;
        if ((rc = upload_f32(e, w->pos_emb, (size_t)cfg->max_pos * H, &e->pos)) != VQA_OK) break;
        if ((rc = upload_f32(e, w->type_emb, H, &e->type0)) != VQA_OK) break;
        if ((rc = upload_f32(e, w->emb_ln_g, H, &e->emb_g)) != VQA_OK) break;
        if ((rc = upload_f32(e, w->emb_ln_b, H, &e->emb_b)) != VQA_OK) break;
        e->layers.resize(cfg->layers);
        for (int i = 0; i < cfg->layers && rc == VQA_OK; ++i) {
            const vqa_encoder_layer_weights& lw = w->layer[i];
            vqa_encoder::Layer& L = e->layers[i];
            if ((rc = dev_alloc(e, (void**)&L.wqkv, 3 * H * H * 2)) != VQA_OK) break;
            if ((rc = upload_f16(e, lw.wq, H * H, L.wqkv)) != VQA_OK) break;
            if ((rc = upload_f16(e, lw.wk, H * H, L.wqkv + H * H)) != VQA_OK) break;
            if ((rc = upload_f16(e, lw.wv, H * H, L.wqkv + 2 * H * H)) != VQA_OK) break;
            if ((rc = dev_alloc(e, (void**)&L.bqkv, 3 * H * 4)) != VQA_OK) break;
            VQA_REQUIRE(lw.bq && lw.bk && lw.bv, "vqa_encoder_create: a bias pointer of layer %d is null", i);
            if (hipMemcpy(L.bqkv, lw.bq, H * 4, hipMemcpyDefault) != hipSuccess ||
                hipMemcpy(L.bqkv + H, lw.bk, H * 4, hipMemcpyDefault) != hipSuccess ||
                hipMemcpy(L.bqkv + 2 * H, lw.bv, H * 4, hipMemcpyDefault) != hipSuccess) {
                vqa_set_error("vqa_encoder_create: copying qkv biases of layer %d failed", i);
                rc = VQA_EHIP;
                break;
            }
            if ((rc = dev_alloc(e, (void**)&L.wo, H * H * 2)) != VQA_OK) break;
            if ((rc = upload_f16(e, lw.wo, H * H, L.wo)) != VQA_OK) break;
            if ((rc = dev_alloc(e, (void**)&L.w1, F * H * 2)) != VQA_OK) break;
            if ((rc = upload_f16(e, lw.w1, F * H, L.w1)) != VQA_OK) break;
            if ((rc = dev_alloc(e, (void**)&L.w2, H * F * 2)) != VQA_OK) break;
            if ((rc = upload_f16(e, lw.w2, H * F, L.w2)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.bo, H, &L.bo)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.b1, F, &L.b1)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.b2, H, &L.b2)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.ln1_g, H, &L.ln1_g)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.ln1_b, H, &L.ln1_b)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.ln2_g, H, &L.ln2_g)) != VQA_OK) break;
            if ((rc = upload_f32(e, lw.ln2_b, H, &L.ln2_b)) != VQA_OK) break;
            if (e->fold_on) {
                // QKV reads the LayerNorm in front of this layer (the embedding's, or the previous layer's second one), FFN1 this
                // layer's first one
                const float* pg = i == 0 ? e->emb_g : e->layers[i - 1].ln2_g;
                const float* pb = i == 0 ? e->emb_b : e->layers[i - 1].ln2_b;
                if ((rc = dev_alloc(e, (void**)&L.wqkv_f, 3 * H * H * 2)) != VQA_OK) break;
                if ((rc = dev_alloc(e, (void**)&L.cqkv, 3 * H * 4)) != VQA_OK) break;
                if ((rc = dev_alloc(e, (void**)&L.bqkv_f, 3 * H * 4)) != VQA_OK) break;
                const float* part[3] = {lw.wq, lw.wk, lw.wv};
                for (int j = 0; j < 3 && rc == VQA_OK; ++j)
                    rc = upload_folded(part[j], (int)H, (int)H, pg, pb, L.bqkv + j * H, L.wqkv_f + j * H * H, L.cqkv + j * H,
                                       L.bqkv_f + j * H);
                if (rc != VQA_OK) break;
                if ((rc = dev_alloc(e, (void**)&L.w1_f, F * H * 2)) != VQA_OK) break;
                if ((rc = dev_alloc(e, (void**)&L.c1, F * 4)) != VQA_OK) break;
                if ((rc = dev_alloc(e, (void**)&L.b1_f, F * 4)) != VQA_OK) break;
                if ((rc = upload_folded(lw.w1, (int)F, (int)H, L.ln1_g, L.ln1_b, L.b1, L.w1_f, L.c1, L.b1_f)) != VQA_OK) break;
            }
        }
        if (rc != VQA_OK) break;
        // token rows padded to the large GEMM's 256-row tiles: rows past B * L are read (never written back), so clear them once
        const size_t T = ((size_t)max_tokens + kTokenPad - 1) / kTokenPad * kTokenPad;
        if ((rc = dev_alloc(e, (void**)&e->x, T * H * 2)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->qkv, T * 3 * H * 2)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->ctx, T * H * 2)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->tmp, T * H * 2)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->ffn, T * F * 2)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->st_x, T * kRowStatSlots * sizeof(float2))) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->st_tmp, T * kRowStatSlots * sizeof(float2))) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->tiny_x, kTinyMaxM * sizeof(float2))) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->tiny_tmp, kTinyMaxM * sizeof(float2))) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->cu, ((size_t)max_tokens + 1) * 4)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->row_seq, (size_t)max_tokens * 4)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->g_ids, (size_t)max_tokens * 4)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->g_mask, (size_t)max_tokens * 4)) != VQA_OK) break;
        if ((rc = dev_alloc(e, (void**)&e->g_out, (size_t)max_tokens * H * 4)) != VQA_OK) break;
        if (hipHostMalloc((void**)&e->bad_ids_host, sizeof(int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&e->bad_ids_dev, e->bad_ids_host, 0) != hipSuccess) {
            vqa_set_error("vqa_encoder_create: allocating the token-id flag failed");
            rc = VQA_ENOMEM;
            break;
        }
        *e->bad_ids_host = 0;
        // the one-launch forward (encoder_persist_kernel): the reference's two model shapes, folded weights at hand, a device that can hold
        // the grid; a failed uncached allocation only switches the path off
        if (o.persistent && e->tiny_on && e->fold_on && ((H == 768 && F == 3072 && cfg->heads == 12) || (H == 384 && F == 1536 && cfg->heads == 12))) {
            auto& P = e->persist;
            hipDeviceProp_t prop;
            int coop = 0;
            const size_t TM = kTinyMaxM;
#ifndef VQA_PERSIST_UC
#define VQA_PERSIST_UC 1
#endif
            auto uc = [&](void** q, size_t bytes) {
                return (VQA_PERSIST_UC ? hipExtMallocWithFlags(q, bytes, hipDeviceMallocUncached) : hipMalloc(q, bytes)) == hipSuccess && hipMemset(*q, 0, bytes) == hipSuccess;
            };
            bool ok = hipGetDeviceProperties(&prop, device) == hipSuccess && hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) == hipSuccess && coop;
            ok = ok && uc((void**)&P.x, TM * H * 2) && uc((void**)&P.qkv, TM * 3 * H * 2) && uc((void**)&P.ctx, TM * H * 2) && uc((void**)&P.tmp, TM * H * 2) &&
                 uc((void**)&P.ffn, TM * F * 2) && uc((void**)&P.tiny_x, TM * sizeof(float2)) && uc((void**)&P.tiny_tmp, TM * sizeof(float2));
            ok = ok && hipMalloc((void**)&P.counter, 64) == hipSuccess && hipMemset(P.counter, 0, 64) == hipSuccess &&
                 hipMalloc((void**)&P.layers, e->layers.size() * sizeof(PersistLayer)) == hipSuccess &&
                 hipHostMalloc((void**)&P.abort_host, sizeof(unsigned), hipHostMallocMapped) == hipSuccess &&
                 hipHostGetDevicePointer((void**)&P.abort_dev, P.abort_host, 0) == hipSuccess;
            if (ok) {
                std::vector<PersistLayer> hl;
                for (const vqa_encoder::Layer& Ly : e->layers)
                    hl.push_back(PersistLayer{Ly.wqkv_f, Ly.wo, Ly.w1_f, Ly.w2, Ly.bqkv_f, Ly.cqkv, Ly.bo, Ly.b1_f, Ly.c1, Ly.b2, Ly.ln1_g, Ly.ln1_b, Ly.ln2_g, Ly.ln2_b});
                ok = hipMemcpy(P.layers, hl.data(), hl.size() * sizeof(PersistLayer), hipMemcpyHostToDevice) == hipSuccess;
            }
            if (ok) {
                *P.abort_host = 0u;
                // workgroups: QKV has 3 H / 16 units, FFN1 F / 16, FFN2 and the out-projection H / 16 x token tiles; a barrier costs 1.2 us at
                // 48 workgroups and 2.9 at 192 (grid_barrier_probe): F / 32 (96 / 48) by default, options.persistent_grid overrides
                int g = o.persistent_grid > 0 ? o.persistent_grid : (int)(F / 32);
                g = std::min(g, prop.multiProcessorCount);
                P.grid = std::max(g, 1);
                P.on = true;
            } else {
                (void)hipGetLastError();
            }
        }
        {
            e->use_graphs = o.graphs != 0;
            if (e->use_graphs && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                e->cap_stream = nullptr;
                e->use_graphs = false;
            }
        }
        if (hipMemset(e->x, 0, T * H * 2) != hipSuccess || hipMemset(e->ctx, 0, T * H * 2) != hipSuccess ||
            hipMemset(e->ffn, 0, T * F * 2) != hipSuccess || hipMemset(e->tmp, 0, T * H * 2) != hipSuccess ||
            hipMemset(e->st_x, 0, T * kRowStatSlots * sizeof(float2)) != hipSuccess ||
            hipMemset(e->st_tmp, 0, T * kRowStatSlots * sizeof(float2)) != hipSuccess) {
            vqa_set_error("vqa_encoder_create: clearing the activation buffers failed");
            rc = VQA_EHIP;
            break;
        }
    } while (0);
    if (rc != VQA_OK) {
        vqa_encoder_destroy(e);
        return rc;
    }
    *out = e;
    return VQA_OK;
}

static void launch_ln(const _Float16* a, int T, int H, const float* g, const float* b, float eps, _Float16* out, hipStream_t s) {
    hipLaunchKernelGGL(ln_kernel, dim3((T + 3) / 4), dim3(256), 0, s, a, T, H, g, b, eps, out);
}

// the launch sequence of one forward pass (no validation, no allocation, no synchronisation: capturable)
// hidden_out != nullptr (vqa_encoder_forward_hidden): run `stop_layers` layers only, write the hidden state of every position as
// fp32 [B, L, H] and return without pooling.
static int encoder_launch(vqa_encoder* e, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L,
                          int32_t real_tokens, int32_t pooling, int32_t normalize, float* out, hipStream_t s,
                          int stop_layers = -1, float* hidden_out = nullptr) {
    const int H = e->cfg.hidden, F = e->cfg.ffn, heads = e->cfg.heads, dh = H / heads;
    // real_tokens > 0: the caller states how many mask entries are set (right-padded masks): only those rows are computed
    const bool packed = real_tokens > 0 && real_tokens < B * L && att_mfma_head_size(dh) && L <= 32 * kAttMaxBlocks;
    const int T = packed ? real_tokens : B * L;  // activation rows
    const int* cu = packed ? e->cu : nullptr;
    const float eps = e->cfg.ln_eps;
    const int row_blocks = (T + 3) / 4;
    // CLS pooling of a large batch: the last layer's out-projection / FFN / LayerNorms only on the first-token rows (below);
    // small calls are launch-bound and keep the plain sequence (VQA_ENC_FIRST_ROWS=0 at create: dev / test switch)
    const bool first_rows_only = hidden_out == nullptr && e->first_rows_on && pooling == VQA_POOL_CLS && T >= kTileMinM && T >= 2 * B && H % 8 == 0 && !e->layers.empty() &&
                                 (size_t)B * (3 * H + F) <= (size_t)e->max_tokens * F;  // the scratch fits the FFN array
    if (packed) {
        hipLaunchKernelGGL(pack_kernel, dim3(1), dim3(256), 0, s, attn_mask, B, L, real_tokens, e->cu, e->row_seq, e->bad_ids_dev);
        VQA_HIP_CHECK(hipGetLastError());
    }
    // LayerNorms folded into the GEMMs (FoldArgs): every GEMM of the layer must take the LDS-DMA tile kernel and the two that
    // produce statistics must fit their slices into the 16 slots of a row
    // (a reader's lane adds up FOUR slots: a slice count that is no multiple of 4 -- hidden size 384: 6 slices of 64 -- is rounded up and the
    // slots in between are zeroed once per forward, below)
    const int w_out = e->fold_on ? tile_stat_slots(T, H, H) : 0, w_ffn = e->fold_on ? tile_stat_slots(T, H, F) : 0;  // slots written
    const int p_out = (w_out + 3) & ~3, p_ffn = (w_ffn + 3) & ~3;                                                      // slots read
    const bool fold = e->fold_on && w_out >= 1 && p_out <= kRowStatSlots && w_ffn >= 1 && p_ffn <= kRowStatSlots &&
                      tile_stat_slots(T, 3 * H, H) > 0 && tile_stat_slots(T, F, H) > 0;
    const int st_stride = (e->max_tokens + kTokenPad - 1) / kTokenPad * kTokenPad;  // rows per statistics slot (the padded row count)
    // one question (<= 64 positions, hidden / FFN sizes in multiples of 384): the latency form -- raw rows all the way, LayerNorms inside
    // the GEMMs (gemm_tiny_kernel), five launches per layer
    if (fold && p_out != w_out)  // (st_tmp is written by the out-projections only)
        VQA_HIP_CHECK(hipMemsetAsync(e->st_tmp + (size_t)w_out * st_stride, 0, (size_t)(p_out - w_out) * st_stride * sizeof(float2), s));
    const bool tiny = e->tiny_on && e->fold_on && !packed && tiny_shape(T, 3 * H, H) && tiny_shape(T, F, H) && tiny_shape(T, H, F) &&
                      att_mfma_head_size(dh) && L <= 32 * kAttMaxBlocks;
    hipLaunchKernelGGL(embed_ln_kernel, dim3(row_blocks), dim3(256), 0, s, input_ids, T, L, H, e->cfg.pad_id, e->cfg.vocab_size,
                       e->cfg.position_ids == VQA_POS_ABSOLUTE ? 1 : 0, e->bad_ids_dev, cu, e->row_seq, B, e->word, e->pos, e->type0, e->emb_g, e->emb_b, eps, e->x,
                       (fold || tiny) ? e->st_x : (float2*)nullptr, st_stride);
    VQA_HIP_CHECK(hipGetLastError());
    // fold: x holds RAW rows; (pg, pb, p_x) = gamma / beta / slots in use of the LayerNorm that belongs on them
    const float *pg = e->emb_g, *pb = e->emb_b;
    int p_x = 4;
    const float inv_h = 1.0f / H;
    const size_t attn_lds = ((size_t)2 * L * (dh + 1) + 4 * L + 4 * dh) * sizeof(float);
    VQA_REQUIRE(attn_lds <= 160 * 1024, "vqa_encoder_forward: L=%d with head size %d needs %zu bytes of LDS", L, dh, attn_lds);
    if (attn_lds > 64 * 1024)
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds));
    int layers_done = 0;
    for (const vqa_encoder::Layer& Ly : e->layers) {
        if (hidden_out && layers_done++ == stop_layers) break;
        int rc;
        if (tiny) {
            TinyArgs tq;
            tq.cvec = Ly.cqkv;
            tq.st_out = e->tiny_x;
            tq.inv_k = inv_h;
            tq.eps = eps;
            rc = launch_gemm_tiny<0, 1>(e->x, Ly.wqkv_f, Ly.bqkv_f, nullptr, e->qkv, T, 3 * H, H, tq, s);
        } else
        if (fold) rc = launch_gemm_fold<0>(e->x, Ly.wqkv_f, Ly.bqkv_f, nullptr, e->qkv, T, 3 * H, H,
                                           FoldArgs{e->st_x, p_x, st_stride, inv_h, eps, Ly.cqkv, nullptr, nullptr, nullptr}, s);
        else rc = launch_gemm<0>(e->x, Ly.wqkv, Ly.bqkv, nullptr, e->qkv, T, 3 * H, H, s);
        if (rc != VQA_OK) return rc;
        if (att_mfma_head_size(dh) && L <= 32 * kAttMaxBlocks) {
            const int nqb = (L + 31) / 32;
            // heads per workgroup: as many as keep the workgroup at <= 512 threads and divide the head count
            int hpw = nqb <= 2 ? 4 : nqb <= 4 ? 2 : 1;
            static const int hpw_force = vqa_dev_env("VQA_ATT_HPW") ? atoi(vqa_dev_env("VQA_ATT_HPW")) : 0;  // dev override (A/B)
            if (hpw_force > 0 && hpw_force <= hpw) hpw = hpw_force;
            while (heads % hpw) hpw >>= 1;
            const size_t lds = (size_t)hpw * nqb * 32 * 2 * dh;  // per head: [32 nqb keys][2 dh bytes]
#define VQA_ATT(NQB, HPW)                                                                                                   \
    do {                                                                                                                    \
        if (dh == 64)                                                                                                       \
            hipLaunchKernelGGL((attention_mfma_kernel<NQB, HPW, 64>), dim3(B * (heads / HPW)), dim3(64 * NQB * HPW), lds, s, e->qkv,  \
                               attn_mask, L, H, heads, cu, e->ctx);                                                        \
        else                                                                                                                \
            hipLaunchKernelGGL((attention_mfma_kernel<NQB, HPW, 32>), dim3(B * (heads / HPW)), dim3(64 * NQB * HPW), lds, s, e->qkv,  \
                               attn_mask, L, H, heads, cu, e->ctx);                                                        \
    } while (0)
#define VQA_ATT_H(NQB)                             \
    case NQB:                                      \
        if (hpw == 4) VQA_ATT(NQB, 4);             \
        else if (hpw == 2) VQA_ATT(NQB, 2);        \
        else VQA_ATT(NQB, 1);                      \
        break;
#define VQA_ATT_L(NQB)                             \
    case NQB:                                      \
        if (hpw == 2) VQA_ATT(NQB, 2);             \
        else VQA_ATT(NQB, 1);                      \
        break;
#define VQA_ATT_1(NQB)                             \
    case NQB:                                      \
        VQA_ATT(NQB, 1);                           \
        break;
            switch (nqb) {
                VQA_ATT_H(1) VQA_ATT_H(2) VQA_ATT_L(3) VQA_ATT_L(4) VQA_ATT_1(5) VQA_ATT_1(6) VQA_ATT_1(7) VQA_ATT_1(8)
            }
#undef VQA_ATT
#undef VQA_ATT_H
#undef VQA_ATT_L
#undef VQA_ATT_1
        } else {
            hipLaunchKernelGGL(attention_kernel, dim3(B * heads), dim3(256), attn_lds, s, e->qkv, attn_mask, L, H, heads, e->ctx);
        }
        VQA_HIP_CHECK(hipGetLastError());
        if (first_rows_only && &Ly == &e->layers.back()) {
            // the last layer past its attention, on the B first-token rows only (scratch: the FFN intermediate array, idle here)
            _Float16* c_ctx = e->ffn;
            _Float16* c_x = c_ctx + (size_t)B * H;
            _Float16* c_tmp = c_x + (size_t)B * H;
            _Float16* c_ffn = c_tmp + (size_t)B * H;
            hipLaunchKernelGGL(gather_first_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, s, e->ctx, e->x, B, L, H, cu, c_ctx, c_x,
                               fold ? e->st_x : (const float2*)nullptr, st_stride, p_x, pg, pb, eps);
            VQA_HIP_CHECK(hipGetLastError());
            if ((rc = launch_gemm<2>(c_ctx, Ly.wo, Ly.bo, c_x, c_tmp, B, H, H, s)) != VQA_OK) return rc;
            launch_ln(c_tmp, B, H, Ly.ln1_g, Ly.ln1_b, eps, c_x, s);
            VQA_HIP_CHECK(hipGetLastError());
            if ((rc = launch_gemm<1>(c_x, Ly.w1, Ly.b1, nullptr, c_ffn, B, F, H, s)) != VQA_OK) return rc;
            if ((rc = launch_gemm<2>(c_ffn, Ly.w2, Ly.b2, c_x, c_tmp, B, H, F, s)) != VQA_OK) return rc;
            launch_ln(c_tmp, B, H, Ly.ln2_g, Ly.ln2_b, eps, c_x, s);
            VQA_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(pool_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, s, c_x, attn_mask, B, 1, H, pooling, normalize,
                               (const int*)nullptr, out, (const float2*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr, eps);
            VQA_HIP_CHECK(hipGetLastError());
            return VQA_OK;
        }
        if (tiny) {
            TinyArgs to, t1, t2;
            to.st_in = e->tiny_x;  // residual = LN_prev(x): (mean, rstd) left by this layer's QKV launch
            to.g = pg;
            to.b = pb;
            if ((rc = launch_gemm_tiny<2, 0>(e->ctx, Ly.wo, Ly.bo, e->x, e->tmp, T, H, H, to, s)) != VQA_OK) return rc;
            t1.cvec = Ly.c1;
            t1.st_out = e->tiny_tmp;
            t1.inv_k = inv_h;
            t1.eps = eps;
            if ((rc = launch_gemm_tiny<1, 1>(e->tmp, Ly.w1_f, Ly.b1_f, nullptr, e->ffn, T, F, H, t1, s)) != VQA_OK) return rc;
            t2.st_in = e->tiny_tmp;  // residual = LN1(tmp)
            t2.g = Ly.ln1_g;
            t2.b = Ly.ln1_b;
            if ((rc = launch_gemm_tiny<2, 0>(e->ffn, Ly.w2, Ly.b2, e->tmp, e->x, T, H, F, t2, s)) != VQA_OK) return rc;
            pg = Ly.ln2_g;
            pb = Ly.ln2_b;
            continue;
        }
        if (fold) {
            // raw rows + statistics all the way: out-projection (residual = LN(x) applied in its epilogue) -> tmp, st_tmp;
            // FFN1 on tmp with LN1 folded; FFN2 (residual = LN1(tmp)) -> x, st_x; the next layer's QKV folds LN2
            if ((rc = launch_gemm_fold<2>(e->ctx, Ly.wo, Ly.bo, e->x, e->tmp, T, H, H,
                                          FoldArgs{e->st_x, p_x, st_stride, inv_h, eps, nullptr, pg, pb, e->st_tmp}, s)) != VQA_OK)
                return rc;
            if ((rc = launch_gemm_fold<1>(e->tmp, Ly.w1_f, Ly.b1_f, nullptr, e->ffn, T, F, H,
                                          FoldArgs{e->st_tmp, p_out, st_stride, inv_h, eps, Ly.c1, nullptr, nullptr, nullptr}, s)) != VQA_OK)
                return rc;
            if (p_ffn != w_ffn && &Ly == &e->layers.front())  // (behind the last reader of the embedding's four slots, in front of FFN2's first write)
                VQA_HIP_CHECK(hipMemsetAsync(e->st_x + (size_t)w_ffn * st_stride, 0, (size_t)(p_ffn - w_ffn) * st_stride * sizeof(float2), s));
            if ((rc = launch_gemm_fold<2>(e->ffn, Ly.w2, Ly.b2, e->tmp, e->x, T, H, F,
                                          FoldArgs{e->st_tmp, p_out, st_stride, inv_h, eps, nullptr, Ly.ln1_g, Ly.ln1_b, e->st_x}, s)) != VQA_OK)
                return rc;
            pg = Ly.ln2_g;
            pb = Ly.ln2_b;
            p_x = p_ffn;
            continue;
        }
        // out-projection / FFN2 add the residual row in their epilogue (EPI 2); the LayerNorm then reads one array
        if ((rc = launch_gemm<2>(e->ctx, Ly.wo, Ly.bo, e->x, e->tmp, T, H, H, s)) != VQA_OK) return rc;
        launch_ln(e->tmp, T, H, Ly.ln1_g, Ly.ln1_b, eps, e->x, s);
        VQA_HIP_CHECK(hipGetLastError());
        if ((rc = launch_gemm<1>(e->x, Ly.w1, Ly.b1, nullptr, e->ffn, T, F, H, s)) != VQA_OK) return rc;
        if ((rc = launch_gemm<2>(e->ffn, Ly.w2, Ly.b2, e->x, e->tmp, T, H, F, s)) != VQA_OK) return rc;
        launch_ln(e->tmp, T, H, Ly.ln2_g, Ly.ln2_b, eps, e->x, s);
        VQA_HIP_CHECK(hipGetLastError());
    }
    const _Float16* x_final = e->x;
    bool x_raw = fold;  // x holds raw rows + slot statistics
    if (tiny) {  // the latency form leaves raw rows (the embedding's, or the last FFN2's): their LayerNorm, once
        if (layers_done == 0 && hidden_out) {
            x_raw = true;  // (the embedding kernel wrote slot statistics: exported as the folded form is)
        } else {
            launch_ln(e->x, T, H, pg, pb, eps, e->tmp, s);
            VQA_HIP_CHECK(hipGetLastError());
            x_final = e->tmp;
            x_raw = false;
        }
    }
    if (hidden_out) {
        if (packed) VQA_HIP_CHECK(hipMemsetAsync(hidden_out, 0, (size_t)B * L * H * sizeof(float), s));  // padding positions: not computed
        hipLaunchKernelGGL(export_hidden_kernel, dim3(row_blocks), dim3(256), 0, s, x_final, T, L, H, cu, e->row_seq, B,
                           x_raw ? e->st_x : (const float2*)nullptr, st_stride, p_x, pg, pb, eps, hidden_out);
        VQA_HIP_CHECK(hipGetLastError());
        return VQA_OK;
    }
    hipLaunchKernelGGL(pool_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, s, x_final, attn_mask, B, L, H, pooling, normalize, cu, out,
                       x_raw ? e->st_x : (const float2*)nullptr, st_stride, p_x, pg, pb, eps);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

namespace {
struct EncBusy {  // the handle's staging buffers, activations and graphs are shared: a second concurrent call is refused
    std::atomic_flag& f;
    bool ok;
    explicit EncBusy(std::atomic_flag& flag) : f(flag), ok(!flag.test_and_set(std::memory_order_acquire)) {}
    ~EncBusy() {
        if (ok) f.clear(std::memory_order_release);
    }
};
}  // namespace

extern "C" int vqa_encoder_forward_hidden(vqa_encoder* e, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L,
                                          int32_t real_tokens, int32_t n_layers, float* out_hidden, void* hip_stream) {
    VQA_REQUIRE(e, "vqa_encoder_forward_hidden: encoder is null");
    VQA_REQUIRE(input_ids && attn_mask && out_hidden, "vqa_encoder_forward_hidden: null pointer");
    VQA_REQUIRE(B >= 1 && L >= 1, "vqa_encoder_forward_hidden: B=%d L=%d", B, L);
    VQA_REQUIRE((long long)B * L <= e->max_tokens, "vqa_encoder_forward_hidden: B*L=%lld exceeds the workspace of %d tokens",
                (long long)B * L, e->max_tokens);
    const int last_pos = e->cfg.position_ids == VQA_POS_ABSOLUTE ? L - 1 : L + e->cfg.pad_id;
    VQA_REQUIRE(last_pos < e->cfg.max_pos, "vqa_encoder_forward_hidden: L=%d needs position %d, the table has %d rows", L, last_pos,
                e->cfg.max_pos);
    VQA_REQUIRE(n_layers >= 0 && n_layers <= e->cfg.layers, "vqa_encoder_forward_hidden: n_layers=%d outside [0, %d]", n_layers, e->cfg.layers);
    VQA_REQUIRE(real_tokens >= 0 && (long long)real_tokens <= (long long)B * L,
                "vqa_encoder_forward_hidden: real_tokens=%d outside [0, B*L=%lld]", real_tokens, (long long)B * L);
    EncBusy busy(e->busy);
    VQA_REQUIRE(busy.ok, "vqa_encoder_forward_hidden: this encoder handle is in use by another host thread");
    DevGuard guard(e->device);
    // always the eager launch sequence (the same kernels, tile shapes and folded LayerNorms a forward of this shape runs; no graph)
    return encoder_launch(e, input_ids, attn_mask, B, L, real_tokens, VQA_POOL_MEAN, 0, nullptr, (hipStream_t)hip_stream, n_layers, out_hidden);
}

static int forward_locked(vqa_encoder* e, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L, int32_t real_tokens,
                          int32_t pooling, int32_t normalize, float* out, hipStream_t s);

extern "C" int vqa_encoder_forward_host(vqa_encoder* e, const int32_t* ids_host, const int32_t* mask_host, int32_t B, int32_t L,
                                        int32_t pooling, int32_t normalize, float* out_dev, void* hip_stream) {
    VQA_REQUIRE(e, "vqa_encoder_forward_host: encoder is null");
    VQA_REQUIRE(ids_host && mask_host && out_dev, "vqa_encoder_forward_host: null pointer");
    VQA_REQUIRE(B >= 1 && L >= 1, "vqa_encoder_forward_host: B=%d L=%d", B, L);
    VQA_REQUIRE((long long)B * L <= e->max_tokens, "vqa_encoder_forward_host: B*L=%lld exceeds the workspace of %d tokens", (long long)B * L,
                e->max_tokens);
    hipStream_t s = (hipStream_t)hip_stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    VQA_REQUIRE(cap == hipStreamCaptureStatusNone, "vqa_encoder_forward_host: the stream is being captured (this entry waits on the host)");
    EncBusy busy(e->busy);
    VQA_REQUIRE(busy.ok, "vqa_encoder_forward_host: this encoder handle is in use by another host thread (one forward at a time per handle)");
    DevGuard guard(e->device);
    const int T = B * L;
    // host-side checks a device-pointer call can only make after the fact: ids inside the table, a right-padded mask (then packed)
    int real = 0;
    bool right_padded = true;
    for (int b = 0; b < B; ++b) {
        int n = 0;
        for (int l = 0; l < L; ++l) {
            const int32_t id = ids_host[b * L + l];
            VQA_REQUIRE(id >= 0 && id < e->cfg.vocab_size, "vqa_encoder_forward_host: token id %d outside [0, %d) (tokenizer / vocabulary mismatch?)", id,
                        e->cfg.vocab_size);
            const bool m = mask_host[b * L + l] != 0;
            right_padded = right_padded && !(m && n != l);
            n += m;
        }
        right_padded = right_padded && n >= 1;
        real += n;
    }
    if (!e->stage_host) {
        if (hipHostMalloc((void**)&e->stage_host, (size_t)2 * e->max_tokens * 4, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&e->stage_dev, e->stage_host, 0) != hipSuccess ||
            hipEventCreateWithFlags(&e->stage_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            vqa_set_error("vqa_encoder_forward_host: allocating the pinned staging buffer failed");
            return VQA_ENOMEM;
        }
    }
    if (e->stage_pending) {  // the previous call's launches may still read the staging buffer
        VQA_HIP_CHECK(hipEventSynchronize(e->stage_done));
        e->stage_pending = false;
    }
    memcpy(e->stage_host, ids_host, (size_t)T * 4);
    memcpy(e->stage_host + e->max_tokens, mask_host, (size_t)T * 4);
    int rc = forward_locked(e, e->stage_dev, e->stage_dev + e->max_tokens, B, L, right_padded ? real : 0, pooling, normalize, out_dev, s);
    if (rc != VQA_OK) return rc;
    VQA_HIP_CHECK(hipEventRecord(e->stage_done, s));
    e->stage_pending = true;
    return VQA_OK;
}

extern "C" int vqa_encoder_forward(vqa_encoder* e, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L,
                                   int32_t real_tokens, int32_t pooling, int32_t normalize, float* out, void* hip_stream) {
    VQA_REQUIRE(e, "vqa_encoder_forward: encoder is null");
    VQA_REQUIRE(input_ids && attn_mask && out, "vqa_encoder_forward: null pointer");
    VQA_REQUIRE(B >= 1 && L >= 1, "vqa_encoder_forward: B=%d L=%d", B, L);
    VQA_REQUIRE((long long)B * L <= e->max_tokens, "vqa_encoder_forward: B*L=%lld exceeds the workspace of %d tokens",
                (long long)B * L, e->max_tokens);
    const int last_pos = e->cfg.position_ids == VQA_POS_ABSOLUTE ? L - 1 : L + e->cfg.pad_id;
    VQA_REQUIRE(last_pos < e->cfg.max_pos, "vqa_encoder_forward: L=%d needs position %d, the table has %d rows", L, last_pos, e->cfg.max_pos);
    VQA_REQUIRE(pooling == VQA_POOL_CLS || pooling == VQA_POOL_MEAN, "vqa_encoder_forward: pooling %d", pooling);
    VQA_REQUIRE(real_tokens >= 0 && (long long)real_tokens <= (long long)B * L, "vqa_encoder_forward: real_tokens=%d outside [0, B*L=%lld]",
                real_tokens, (long long)B * L);
    hipStream_t s = (hipStream_t)hip_stream;
    EncBusy busy(e->busy);
    VQA_REQUIRE(busy.ok, "vqa_encoder_forward: this encoder handle is in use by another host thread (one forward at a time per handle)");
    DevGuard guard(e->device);
    return forward_locked(e, input_ids, attn_mask, B, L, real_tokens, pooling, normalize, out, s);
}

// ---- the one-launch forward (encoder_persist_kernel).  Returns VQA_PERSIST_DECLINED when the launches should run instead.
constexpr int VQA_PERSIST_DECLINED = 1;
template <int MT, bool FOUR, int NQB>
static int persist_launch_t(vqa_encoder* e, PersistArgs& pa, hipStream_t s) {
    constexpr int DH = FOUR ? 32 : 64;
    const int lds = 4 * NQB * 32 * 2 * DH;  // the attention's V images: [4 heads][32 NQB keys][2 DH bytes]
    static VqaPerDeviceOnce once;
    int rc = once.run([&](int) -> int {
        VQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_persist_kernel<MT, FOUR, NQB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return VQA_OK;
    });
    if (rc != VQA_OK) return rc;
    void* args[] = {&pa};
    const hipError_t err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(encoder_persist_kernel<MT, FOUR, NQB>), dim3(e->persist.grid), dim3(512), args,
                                                      (unsigned)lds, s);
    if (err != hipSuccess) {  // (a grid the device cannot hold resident, ...): the launches take over for good
        (void)hipGetLastError();
        e->persist.on = false;
        return VQA_PERSIST_DECLINED;
    }
    return VQA_OK;
}

static int persist_launch(vqa_encoder* e, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L, int32_t pooling, int32_t normalize,
                          float* out, hipStream_t s) {
    auto& P = e->persist;
    if (__atomic_load_n(P.abort_host, __ATOMIC_RELAXED)) {  // an earlier launch gave up at a barrier: its output was invalid
        __atomic_store_n(P.abort_host, 0u, __ATOMIC_RELAXED);
        P.on = false;
        vqa_set_error("vqa_encoder_forward: an earlier one-launch forward on this handle did not complete (a grid barrier ran out: its output is "
                      "invalid); the handle uses the launches from now on");
        return VQA_EHIP;
    }
    const int T = B * L, H = e->cfg.hidden, F = e->cfg.ffn;
    // the call's ids and masks (device pointers, or the pinned device-mapped buffer of vqa_encoder_forward_host) -> device arrays
    hipLaunchKernelGGL(stage_tokens_kernel, dim3((T + 255) / 256), dim3(256), 0, s, input_ids, attn_mask, T, e->g_ids, e->g_mask);
    VQA_HIP_CHECK(hipGetLastError());
    PersistArgs pa;
    pa.ids = e->g_ids;
    pa.mask = e->g_mask;
    pa.B = B;
    pa.L = L;
    pa.T = T;
    pa.H = H;
    pa.F = F;
    pa.heads = e->cfg.heads;
    pa.layers = (int)e->layers.size();
    pa.pad_id = e->cfg.pad_id;
    pa.vocab = e->cfg.vocab_size;
    pa.abs_pos = e->cfg.position_ids == VQA_POS_ABSOLUTE ? 1 : 0;
    pa.bad_ids = e->bad_ids_dev;
    pa.word = e->word;
    pa.pos = e->pos;
    pa.type0 = e->type0;
    pa.emb_g = e->emb_g;
    pa.emb_b = e->emb_b;
    pa.eps = e->cfg.ln_eps;
    pa.lay = P.layers;
    pa.x = P.x;
    pa.qkv = P.qkv;
    pa.ctx = P.ctx;
    pa.tmp = P.tmp;
    pa.ffn = P.ffn;
    pa.tiny_x = P.tiny_x;
    pa.tiny_tmp = P.tiny_tmp;
    pa.st_scratch = e->st_x;
    pa.st_stride = (e->max_tokens + kTokenPad - 1) / kTokenPad * kTokenPad;
    pa.pooling = pooling;
    pa.normalize = normalize;
    pa.out = out;
    pa.counter = P.counter;
    pa.base = P.base;
    pa.abort = P.abort_dev;
    const bool four = H == 384;
    const int mt = T <= 32 ? 2 : 4, nqb = L <= 32 ? 1 : 2;
    int rc;
#define VQA_PERSIST(MTV, FOURV, NQBV) rc = persist_launch_t<MTV, FOURV, NQBV>(e, pa, s)
    if (!four) {
        if (mt == 2 && nqb == 1) VQA_PERSIST(2, false, 1);
        else if (mt == 2) VQA_PERSIST(2, false, 2);
        else if (nqb == 1) VQA_PERSIST(4, false, 1);
        else VQA_PERSIST(4, false, 2);
    } else {
        if (mt == 2 && nqb == 1) VQA_PERSIST(2, true, 1);
        else if (mt == 2) VQA_PERSIST(2, true, 2);
        else if (nqb == 1) VQA_PERSIST(4, true, 1);
        else VQA_PERSIST(4, true, 2);
    }
#undef VQA_PERSIST
    if (rc == VQA_OK) P.base += persist_barriers(pa.layers) * (unsigned)P.grid;  // what the counter gains in one forward
    return rc;
}

// the body of a forward call: arguments checked by the caller, handle held, device set
static int forward_locked(vqa_encoder* e, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L, int32_t real_tokens,
                          int32_t pooling, int32_t normalize, float* out, hipStream_t s) {
    const int last_pos = e->cfg.position_ids == VQA_POS_ABSOLUTE ? L - 1 : L + e->cfg.pad_id;
    VQA_REQUIRE(last_pos < e->cfg.max_pos, "vqa_encoder_forward: L=%d needs position %d, the table has %d rows", L, last_pos, e->cfg.max_pos);
    VQA_REQUIRE(pooling == VQA_POOL_CLS || pooling == VQA_POOL_MEAN, "vqa_encoder_forward: pooling %d", pooling);
    if (__atomic_load_n(e->bad_ids_host, __ATOMIC_RELAXED)) {
        __atomic_store_n(e->bad_ids_host, 0, __ATOMIC_RELAXED);
        vqa_set_error("vqa_encoder_forward: an earlier forward on this handle saw token ids outside [0, %d) (embedded as the pad token: "
                      "tokenizer / vocabulary mismatch?) or a packed call whose mask was not right-padded / whose number of real tokens "
                      "differed from the announced one (its output is invalid)", e->cfg.vocab_size);
        return VQA_EINVAL;
    }
    const int T = B * L;
    hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap0);
    // ONE question (<= 64 positions) on one of the reference's two model shapes: the whole forward as one cooperative launch
    // (launch-bound sizes run the PADDED form whatever real_tokens says, as the graph path below does)
    if (e->persist.on && e->use_graphs && cap0 == hipStreamCaptureStatusNone && T <= kTinyMaxM && L <= 64) {
        const int prc = persist_launch(e, input_ids, attn_mask, B, L, pooling, normalize, out, s);
        if (prc != VQA_PERSIST_DECLINED) return prc;
    }
    // Launch-bound sizes replay a captured graph (unless the caller's stream is itself being captured: then the kernels
    // simply join the caller's capture).  Above 1024 tokens the kernels are long enough for eager launches to stay ahead of
    // the GPU (graph = eager: 1.10 / 1.28 / 1.41 ms at 2048 / 3072 / 4096 tokens), and eager calls can pack a ragged batch
    // (1.02 / 1.11 / 1.17 ms at the SURVEY's length distribution).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (!e->use_graphs || T > kGraphMaxTokens || cap != hipStreamCaptureStatusNone)
        return encoder_launch(e, input_ids, attn_mask, B, L, real_tokens, pooling, normalize, out, s);
    // (launch-bound sizes replay a graph of the padded form: a graph is keyed by shape, a packed row count is not one)
    vqa_encoder::Graph* gr = nullptr;
    for (auto& c : e->graphs)
        if (c.B == B && c.L == L && c.pooling == pooling && c.normalize == normalize) gr = &c;
    if (!gr) {  // first call of this shape: eager (sets the kernels' attributes), remembered -- up to 64 shapes, then eager only
        if (e->graphs.size() < 64) e->graphs.push_back({B, L, pooling, normalize, nullptr, true});
        return encoder_launch(e, input_ids, attn_mask, B, L, 0, pooling, normalize, out, s);
    }
    const size_t H = e->cfg.hidden;
    // the call's ids and masks -> the graph's input arrays: ONE small kernel (device pointers, or the pinned device-mapped buffer of
    // vqa_encoder_forward_host, read over the bus); two copy operations of a few hundred bytes took ~25 us of the stream's time
    hipLaunchKernelGGL(stage_tokens_kernel, dim3((T + 255) / 256), dim3(256), 0, s, input_ids, attn_mask, T, e->g_ids, e->g_mask);
    VQA_HIP_CHECK(hipGetLastError());
    if (!gr->exec) {
        hipGraph_t graph = nullptr;
        VQA_HIP_CHECK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = encoder_launch(e, e->g_ids, e->g_mask, B, L, 0, pooling, normalize, e->g_out, e->cap_stream);
        const hipError_t end = hipStreamEndCapture(e->cap_stream, &graph);
        if (rc != VQA_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        VQA_HIP_CHECK(end);
        const hipError_t inst = hipGraphInstantiate(&gr->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        VQA_HIP_CHECK(inst);
    }
    VQA_HIP_CHECK(hipGraphLaunch(gr->exec, s));
    VQA_HIP_CHECK(hipMemcpyAsync(out, e->g_out, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    return VQA_OK;
}
