// Element-wise helpers around the index (HBM-bound, 16 B per lane where the shape allows), gfx950 only.
//   * tile_rows / untile_rows : row-major [count, d] f32|f16 rows <-> the TILED layout of the index in its storage type
//     (fp16 | fp8-e4m3 | fp32), zero padded; also stages the 256-query tile of a search
//   * normalize_convert : fp32 rows -> L2-normalised (txtai normalises at index and at query time; cosine intent at
//     /root/reference/src/test.py:104) -> fp32 | fp16 | fp8-e4m3 (OCP, saturating, round to nearest even)
#include "vqa_common.h"

namespace {

__device__ __forceinline__ uint8_t f32_to_e4m3(float f) {
    // OCP e4m3fn, round-to-nearest-even, saturating at +-448 (NaN -> 0x7f).  Bit-exact with oracle/retrieval.py
    // e4m3_encode; written with integer arithmetic so the result does not depend on a hardware conversion mode.
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    const uint32_t sign = (u >> 24) & 0x80u;
    u &= 0x7FFFFFFFu;
    if (u > 0x7F800000u) return 0x7F;                       // NaN
    if (u >= 0x43E00000u) return (uint8_t)(sign | 0x7E);     // |f| >= 448 -> 448 (also +-inf)
    if (u < 0x3A800000u) {                                   // |f| < 2^-10: rounds to 0 or the smallest subnormal
        // spacing 2^-9: values in (2^-10, ...) handled below; here |f| <= 2^-10 -> 0 (ties to even = 0)
        return (uint8_t)sign;
    }
    const int e = (int)(u >> 23) - 127;  // unbiased exponent, >= -10
    uint32_t mant = (u & 0x7FFFFFu) | 0x800000u;  // 24-bit significand
    int shift;                                    // bits to drop so that 3 fraction bits remain (normal) or fewer
    int eb;
    if (e >= -6) {
        shift = 20;
        eb = e + 7;
    } else {  // subnormal target: value = m * 2^-9, m in 0..7
        shift = 20 + (-6 - e);
        eb = 0;
    }
    const uint32_t half = 1u << (shift - 1);
    const uint32_t rest = mant & ((1u << shift) - 1);
    uint32_t m = mant >> shift;
    if (rest > half || (rest == half && (m & 1u))) ++m;
    uint32_t code;
    if (eb == 0) {
        code = m;  // m may reach 8 -> encodes as exponent 1, mantissa 0 (= 2^-6), which is the right value
    } else {
        m -= 8;  // remove the hidden bit (m in 8..16)
        code = ((uint32_t)eb << 3) + m;  // m == 8 carries into the exponent
    }
    if (code > 0x7Eu) code = 0x7E;
    return (uint8_t)(sign | code);
}

// TILED layout: tile t (256 rows), K-step kappa (64 bytes of every row = 32 fp16 / 64 fp8 / 16 fp32 elements) is one
// contiguous 16 KiB block at unit index (t * KT + kappa) * 1024 (unit = 16 B); inside the block the unit of (row r,
// 16-byte slot s) sits at r * 4 + (s ^ 3 * bit3(r)) -- exactly the bank-conflict-free LDS image K1 wants, so K1's LDS-DMA
// reads 1 KiB contiguous per wave-instruction and a tile's K-steps stream from HBM as one sequential 256 * d_pad * esize
// byte run.  Which k index lands in which byte of a unit is irrelevant as long as rows and queries agree (a dot product
// is invariant under a common permutation of k), so elements are simply stored in their natural order.
__device__ __forceinline__ size_t tiled_unit(long long row, int kappa, int slot, int KT) {
    const long long t = row >> 8;
    const int r = (int)(row & 255);
    return ((size_t)t * KT + kappa) * 1024 + (size_t)(r * 4 + (slot ^ (((r >> 3) & 1) * 3)));
}

template <typename DST>
__device__ __forceinline__ DST store_cast(float v);
template <>
__device__ __forceinline__ float store_cast<float>(float v) { return v; }
template <>
__device__ __forceinline__ _Float16 store_cast<_Float16>(float v) { return (_Float16)v; }
template <>
__device__ __forceinline__ uint8_t store_cast<uint8_t>(float v) { return f32_to_e4m3(v); }

// rows: [valid, d] row-major, SRC element type; row i goes to index row first + i; rows valid..count-1 become zeros.
// One thread per 16-byte output unit (EPU = 16 / sizeof(DST) elements); values are multiplied by `scale` (a power of
// two: 1 for fp16/fp32 storage, the fp8 range scale otherwise) before the round-to-nearest-even conversion.
template <typename SRC, typename DST>
__global__ void tile_rows_kernel(const SRC* __restrict__ rows, long long first, long long count, long long valid, int d,
                                 int KT, float scale, DST* __restrict__ out) {
    constexpr int EPU = 16 / (int)sizeof(DST);
    const int units_per_row = KT * 4;
    const long long total = count * units_per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ri = i / units_per_row;
        const int u = (int)(i - ri * units_per_row);
        const int kappa = u >> 2, slot = u & 3;
        const int j0 = u * EPU;
        DST v[EPU];
#pragma unroll
        for (int e = 0; e < EPU; ++e)
            v[e] = store_cast<DST>((ri < valid && j0 + e < d) ? (float)rows[ri * d + j0 + e] * scale : 0.0f);
        DST* dst = out + tiled_unit(first + ri, kappa, slot, KT) * EPU;
#pragma unroll
        for (int e = 0; e < EPU; ++e) dst[e] = v[e];
    }
}

// inverse of tile_rows: TILED -> row-major [count, d] in the storage type (index export for Embeddings.save)
template <typename DST>
__global__ void untile_rows_kernel(const DST* __restrict__ tiled, long long first, long long count, int d, int KT,
                                   DST* __restrict__ out) {
    constexpr int EPU = 16 / (int)sizeof(DST);
    const long long total = count * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ri = i / d;
        const int j = (int)(i - ri * d);
        const int u = j / EPU;
        out[i] = tiled[tiled_unit(first + ri, u >> 2, u & 3, KT) * EPU + (j - u * EPU)];
    }
}

// one wave per row: sum of squares by wavefront shuffles, then scale + convert
__global__ __launch_bounds__(256) void normalize_convert_kernel(const float* __restrict__ rows, long long n, int d,
                                                                int normalize, int dtype, void* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave; r < n; r += nwaves) {
        const float* src = rows + r * d;
        float nrm = 0.f;
        if (normalize) {
            float ss = 0.f;
            for (int j = lane; j < d; j += 64) ss += src[j] * src[j];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
            nrm = sqrtf(ss);
        }
        for (int j = lane; j < d; j += 64) {
            // x / ||x|| as a true division (the oracle divides too); zero rows stay zero
            const float v = nrm > 0.f ? src[j] / nrm : src[j];
            const long long o = r * d + j;
            if (dtype == VQA_F16) reinterpret_cast<_Float16*>(out)[o] = (_Float16)v;
            else if (dtype == VQA_F32) reinterpret_cast<float*>(out)[o] = v;
            else reinterpret_cast<uint8_t*>(out)[o] = f32_to_e4m3(v);
        }
    }
}

// ---- int8 sketch of fp16 / fp32 rows (the pruning pre-pass of a large shard, score_topk.hip MODE 2).  One wave per row of a
// TILED array (index rows or the staged query tile): the stored values x -> x_int = clamp(rint(x / s), -127, 127) in the
// TILED int8 layout (K-blocks of 64 elements), with s the scale of the row's 256-row TILE (index rows: max|x| of the tile
// / 127, tile_scale_kernel) or the row's own max|x| / 127 (queries), plus what the bound
// |q . x - s_q s_x (q_int . x_int)| <= ||q_lo|| ||x_hi|| + ||q|| ||x_lo||  needs
// (x_hi = s x_int, x_lo = x - x_hi): per row the scale, ||x_lo|| and ||x|| (queries), per 256-row tile the maxima of ||x_hi||
// and ||x_lo|| (index rows; atomicMax on the bits of non-negative floats, interleaved [tile][4]).  Norms are rounded up (relative 2^-16) so that
// fp32 rounding inside this kernel can never make the bound too tight.
// 16 consecutive elements (one int8 unit u) of a row of a TILED array of SRC: two fp16 units (2 u, 2 u + 1) or four fp32 units
// (4 u .. 4 u + 3 = the four slots of K-block u); zero past the source's padding
template <typename SRC>
__device__ __forceinline__ void load_sketch_unit(const SRC* __restrict__ tiled, long long row, int u, int KTS, float (&x)[16]) {
    constexpr int EPU = 16 / (int)sizeof(SRC);  // elements per 16-byte source unit: 8 or 4
    constexpr int NU = 16 / EPU;                 // source units per int8 unit
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int us = NU * u + i;
        if (us < KTS * 4) {
            const SRC* p = tiled + tiled_unit(row, us >> 2, us & 3, KTS) * EPU;
#pragma unroll
            for (int e = 0; e < EPU; ++e) x[i * EPU + e] = (float)p[e];
        } else {
#pragma unroll
            for (int e = 0; e < EPU; ++e) x[i * EPU + e] = 0.f;
        }
    }
}

// ---- rotation of a row before it is sketched (ROT).  One scale per tile means the largest component sets the quantisation step of
// all the others: a few outlier dimensions (what RoBERTa-family encoders emit) widen the bound's slack six-fold.  The sketch is
// therefore cut from T x instead of x, T = H D: random signs D (a fixed hash of the element index) and a normalised Walsh-Hadamard
// transform H of every power-of-two block of the padded row (768 = 512 + 256; blocks of >= 128 elements), applied in fp32 to
// rows at build time and to the query tile per search.  T is orthogonal, (T q) . (T x) = q . x, so the bound holds unchanged on
// the rotated quantities (the norms below are those of T x); its coordinates are near-Gaussian whatever the input's.  Rounding:
// log2(B) <= 13 butterfly stages of one fp32 add each, |T^ x - T x| <= 13 2^-24 ||x|| -- vqa_launch_sketch_qconst adds it to the margin.
// Row layout in a wave (sketch_rows_kernel): element j = 16 (lane + 64 i) + e sits in x[i][e] of `lane`.
// (NG = register groups of 1024 elements a kernel is built for: 1, 2, 4 or 8 -- rows of up to 8 * 64 * 16 = 8192 elements)

__device__ __forceinline__ int sketch_block_of(int j, int d8) {  // size of the power-of-two block of the padded row that holds element j
    int off = 0, rem = d8;
    for (;;) {
        const int b = 1 << (31 - __builtin_clz(rem));
        if (j < off + b || rem == b) return b;
        off += b;
        rem -= b;
    }
}

template <int NG>
__device__ __forceinline__ void sketch_rotate(float (&x)[NG][16], int lane, int d8) {
    float bs[NG];  // block size of this lane's elements of register group i (0: past the row)
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int j0 = 16 * (lane + 64 * i);
        bs[i] = j0 < d8 ? (float)sketch_block_of(j0, d8) : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {  // D: sign of element j from a fixed, well-mixed hash (a bit of a plain multiplicative hash
            // of consecutive j is close to a Walsh function: H D would map a constant row onto a few huge coordinates)
            unsigned h = (unsigned)(j0 + e) + 0x9E3779B9u;
            h ^= h >> 16;
            h *= 0x85EBCA6Bu;
            h ^= h >> 13;
            h *= 0xC2B2AE35u;
            h ^= h >> 16;
            x[i][e] = (h & 1u) ? -x[i][e] : x[i][e];
        }
    }
    // bits 0-3 of j: inside the 16 registers of a group (every block has at least 128 elements)
#pragma unroll
    for (int b = 1; b < 16; b <<= 1)
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!(e & b)) {
                    const float u = x[i][e], v = x[i][e | b];
                    x[i][e] = u + v;
                    x[i][e | b] = u - v;
                }
    // bits 4-9: across lanes (partner lane ^ m; both inside the same block when 16 * 2 m <= block size)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const bool on = bs[i] >= (float)(32 * m);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = x[i][e], p = __shfl_xor(v, m, 64);
                x[i][e] = !on ? v : (lane & m) ? p - v : v + p;
            }
        }
    // bits 10-12: across the register groups (blocks of 2048 elements and more)
#pragma unroll
    for (int m = 1; m < NG; m <<= 1)
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (!(i & m)) {
                const bool on = bs[i] >= (float)(2048 * m);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float u = x[i][e], v = x[i | m][e];
                    x[i][e] = on ? u + v : u;
                    x[i | m][e] = on ? u - v : v;
                }
            }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const float nrm = bs[i] > 0.f ? rsqrtf(bs[i]) : 0.f;  // H / sqrt(B): orthonormal (B a power of two: exact for even log2, one rounding else)
#pragma unroll
        for (int e = 0; e < 16; ++e) x[i][e] *= nrm;
    }
}

// mu (or null): the shard's centre, d8 floats.  Index rows (center = 1) are sketched as T (x - mu): embeddings of one encoder share a
// large common component (mean cosine 0.5 and more), which would otherwise eat the quantiser's range; q . x = q . mu + q . (x - mu), so
// the query side (center = 0) only reports q . mu (row_off) and the scan's threshold moves by it (sketch_qconst_kernel).
// The query tile of a search (QueryStage::src != nullptr): the rows come ROW-MAJOR from the caller (fp32 or fp16, `valid` of them, the
// rest of the 256 read as zeros), are converted to the storage type SRC -- the values every exact score is computed from -- and
// written out as the staged tile (TILED, what the exact kernels read) and as its row-major copy (what the re-scoring reads), before
// the same registers are sketched: one launch instead of three (tile_rows, sketch_rows, rows_to_rowmajor: 23 -> 12 us per search).
struct QueryStage {
    const void* src = nullptr;  // [valid][d] row-major, nullptr: the rows are read from `tiled` (index rows)
    int src_f32 = 0;            // element type of src: 1 = fp32, 0 = fp16
    int valid = 0, d = 0;
    float scale = 1.0f;         // values are multiplied by it before the conversion (a power of two)
    void* stage = nullptr;      // TILED tile of SRC (KTS K-blocks)
    void* rowmajor = nullptr;   // [256][KTS * 64 bytes] or nullptr
};

template <typename SRC, bool ROT, int NG>
__global__ __launch_bounds__(256) void sketch_rows_kernel(const SRC* __restrict__ tiled, long long first, long long count,
                                                          int KTS, int KT8, const float* tile_info, int8_t* __restrict__ out8,
                                                          float* __restrict__ row_scale, float* __restrict__ row_lo,
                                                          float* __restrict__ row_norm, unsigned* tile_max /* = tile_info */,
                                                          const float* __restrict__ mu, int center, float* __restrict__ row_off,
                                                          QueryStage qs, SketchSplit sp) {
    const int lane = threadIdx.x & 63;
    const long long ri = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ri >= count) return;
    const long long row = first + ri;
    const int units8 = KT8 * 4;
    // this lane's int8 units u = lane, lane + 64, ...
    constexpr int kMaxPer = NG;
    float x[kMaxPer][16];
    float amax = 0.f;
    if (qs.src) {
        constexpr int EPU = 16 / (int)sizeof(SRC);  // elements per 16-byte unit of the storage type: 8 or 4
        constexpr int NU = 16 / EPU;
        typedef SRC unit_t __attribute__((ext_vector_type(EPU)));
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int u = lane + 64 * i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = 16 * u + e;
                float v = 0.f;
                if (u < units8 && ri < qs.valid && j < qs.d)
                    v = qs.src_f32 ? reinterpret_cast<const float*>(qs.src)[(size_t)ri * qs.d + j]
                                   : (float)reinterpret_cast<const _Float16*>(qs.src)[(size_t)ri * qs.d + j];
                x[i][e] = (float)(SRC)(v * qs.scale);  // the stored value
            }
#pragma unroll
            for (int n = 0; n < NU; ++n) {
                const int us = NU * u + n;
                if (u < units8 && us < KTS * 4) {
                    unit_t o;
#pragma unroll
                    for (int e = 0; e < EPU; ++e) o[e] = (SRC)x[i][n * EPU + e];
                    *reinterpret_cast<unit_t*>(reinterpret_cast<SRC*>(qs.stage) + tiled_unit(row, us >> 2, us & 3, KTS) * EPU) = o;
                    if (qs.rowmajor) *reinterpret_cast<unit_t*>(reinterpret_cast<SRC*>(qs.rowmajor) + ((size_t)row * (KTS * 4) + us) * EPU) = o;
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
        const int u = lane + 64 * i;
        if (u < units8) load_sketch_unit<SRC>(tiled, row, u, KTS, x[i]);
        else
#pragma unroll
            for (int e = 0; e < 16; ++e) x[i][e] = 0.f;
    }
    }
    if (mu) {
        // q . mu in DOUBLE: the scan's threshold moves by it, and an fp32 dot of d terms errs by gamma_d ||q|| ||mu|| -- for embeddings
        // collapsed onto their centre (||mu|| ~ 1) that alone was a third of the bound's slack.  768 fp64 fmas per query and search.
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int u = lane + 64 * i;
            if (u >= units8) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m = mu[16 * u + e];
                if (center) x[i][e] -= m;
                else dot = __builtin_fma((double)x[i][e], (double)m, dot);
            }
        }
        if (!center) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
            if (lane == 0 && row_off) row_off[ri] = (float)dot;
        }
    } else if (lane == 0 && row_off) {
        row_off[ri] = 0.f;
    }
    if constexpr (ROT) sketch_rotate<NG>(x, lane, units8 * 16);
    // The slack term |z . x_lo| of the bound (z = the rotated query) is split along w = the shard's rotated, normalised centre:
    // z = alpha w + z_r  =>  |z . x_lo| <= |alpha| |w . x_lo| + ||z_r|| ||x_lo||.  x_lo is a quantisation residue, nearly orthogonal to any
    // fixed direction (|w . x_lo| ~ ||x_lo|| / sqrt(d)), while the queries of a corpus whose embeddings share a large common component
    // are mostly alpha w (alpha ~ the mean cosine): the term shrinks from ||q|| ||x_lo|| to about ||z_r|| ||x_lo||.  The identity holds for
    // ANY vector w and ANY number alpha as long as z_r is z - alpha w: nothing here has to be exact except that subtraction (its
    // rounding, <= 2^-23 ||z|| ||x_lo||, rides in the margin).  Query rows report |alpha| and ||z_r||, index rows raise their tile's
    // max |w . x_lo| (below, with the codes).
    // Per-row form (sp.per_row; shards whose embeddings collapse onto the centre direction, mean cosine between two rows >= 0.6): the rank-one part leaves
    // the sketch altogether.  With y = beta w + y_r per row (beta = w . y) and z = alpha w + z_r per query,
    //     z . y = alpha (w . y) + beta (z_r . w) + z_r . y_r,
    // the sketch is cut from y_r and z_r -- a quarter of the norm, a quarter of the quantisation step --, the scan adds alpha beta per
    // (query, row) in its epilogue (score_topk.hip LOOP 2) and beta (z_r . w), of the order of the rounding of alpha, rides in the margin.
    float full2 = -1.f;  // ||z||^2 before a projection (the margins speak of the whole query)
    if (sp.wdir && (sp.per_row || (!tile_info && (sp.row_alpha || sp.row_rnorm)))) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int u = lane + 64 * i;
            if (u >= units8) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) a = __builtin_fmaf(x[i][e], sp.wdir[16 * u + e], a);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
        if (!(fabsf(a) < INFINITY)) a = 0.f;  // a row with NaN / Inf: no projection (its norms are infinite anyway)
        float r2 = 0.f, f2 = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int u = lane + 64 * i;
            if (u >= units8) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float zr = x[i][e] - a * sp.wdir[16 * u + e];
                r2 += zr * zr;
                f2 += x[i][e] * x[i][e];
                if (sp.per_row) x[i][e] = zr;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            r2 += __shfl_xor(r2, off, 64);
            f2 += __shfl_xor(f2, off, 64);
        }
        if (sp.per_row) full2 = f2;
        if (lane == 0) {
            const float up = 1.0f + 1.0f / 65536.0f;
            float aa = sp.per_row ? a : fabsf(a) * up, rn = sqrtf(r2) * up;
            if (!(rn < INFINITY)) rn = INFINITY;
            if (!tile_info) {
                if (sp.row_alpha) sp.row_alpha[ri] = aa;
                if (sp.row_rnorm) sp.row_rnorm[ri] = rn;
            } else if (sp.beta) {  // index rows: beta of the row, and the tile's max |beta| (a bound on ||y|| - ||y_r|| for the margins)
                sp.beta[row] = a;
                atomicMax(reinterpret_cast<unsigned*>(sp.tile_c) + (row >> 8), __builtin_bit_cast(unsigned, fabsf(a) * up));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(x[i][e]));
    float s = tile_info ? tile_info[4 * (row >> 8) + 3] : 0.f;  // index rows: the scale of the row's tile
    if (!(s > 0.f)) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        s = amax > 0.f ? amax / 127.0f : 1.0f;
    }
    float hi2 = 0.f, lo2 = 0.f, n2 = 0.f, wl = 0.f;
    const bool split = sp.wdir && sp.tile_c && tile_info && !sp.per_row;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
        const int u = lane + 64 * i;
        if (u >= units8) continue;
        union {
            int8_t b[16];
            uint4 v;
        } o;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float q = rintf(x[i][e] / s);
            q = fminf(fmaxf(q, -127.f), 127.f);
            o.b[e] = (int8_t)q;
            const float hi = q * s, lo = x[i][e] - hi;
            hi2 += hi * hi;
            lo2 += lo * lo;
            n2 += x[i][e] * x[i][e];
            if (split) wl = __builtin_fmaf(lo, sp.wdir[16 * u + e], wl);
        }
        *reinterpret_cast<uint4*>(out8 + tiled_unit(row, u >> 2, u & 3, KT8) * 16) = o.v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        hi2 += __shfl_xor(hi2, off, 64);
        lo2 += __shfl_xor(lo2, off, 64);
        n2 += __shfl_xor(n2, off, 64);
        wl += __shfl_xor(wl, off, 64);
    }
    if (lane == 0) {
        const float up = 1.0f + 1.0f / 65536.0f;
        float hi = sqrtf(hi2) * up, lo = sqrtf(lo2) * up + 1e-30f, nn = sqrtf(full2 >= 0.f ? full2 : n2) * up;
        // a row holding NaN / Inf cannot be bounded: an infinite norm sends its whole tile (or query) to the candidates
        if (!(hi < INFINITY)) hi = INFINITY;
        if (!(lo < INFINITY)) lo = INFINITY;
        if (!(nn < INFINITY)) nn = INFINITY;
        if (row_scale) row_scale[ri] = s;
        if (row_lo) row_lo[ri] = lo;
        if (row_norm) row_norm[ri] = nn;
        if (tile_max) {  // [tiles][4]: (max ||x_hi||, max ||x_lo||, 1 / scale, scale), read as float4 by the sketch scan
            atomicMax(tile_max + 4 * (row >> 8), __builtin_bit_cast(unsigned, hi));
            atomicMax(tile_max + 4 * (row >> 8) + 1, __builtin_bit_cast(unsigned, lo));
            if (split) {  // |w . x_lo| of this row, rounded up: the fp32 dot errs by at most gamma_d ||x_lo|| ||w||, ||w|| <= 1 + 1e-6
                float c = (fabsf(wl) + (float)(units8 * 16) * 1.3e-7f * lo) * up;
                if (!(c < INFINITY)) c = INFINITY;
                atomicMax(reinterpret_cast<unsigned*>(sp.tile_c) + (row >> 8), __builtin_bit_cast(unsigned, c));
            }
        }
    }
}

// one workgroup per tile of a TILED fp16 / fp32 array: scale = max |x| over the tile's 256 rows / 127 (a tile of zeros: 1), written
// with its reciprocal to tile_info[tile] = (0, 0, 1 / scale, scale) -- the two maxima are cleared for sketch_rows_kernel to fill
template <typename SRC, bool ROT, int NG>
__global__ __launch_bounds__(256) void tile_scale_kernel(const SRC* __restrict__ tiled, long long tile0, int KTS, int KT8,
                                                         float* __restrict__ tile_info, const float* __restrict__ mu,
                                                         float* __restrict__ tile_c, const float* __restrict__ proj_w) {
    constexpr int EPU = 16 / (int)sizeof(SRC);
    __shared__ float red[4];
    const long long tile = tile0 + blockIdx.x;
    float m = 0.f;
    if constexpr (ROT) {  // the maximum is that of the ROTATED rows: every wave rotates 64 of the tile's rows
        const int lane = threadIdx.x & 63, units8 = KT8 * 4;
        for (int r = threadIdx.x >> 6; r < 256; r += 4) {
            float x[NG][16];
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int u = lane + 64 * i;
                if (u < units8) load_sketch_unit<SRC>(tiled, tile * 256 + r, u, KTS, x[i]);
                else
#pragma unroll
                    for (int e = 0; e < 16; ++e) x[i][e] = 0.f;
            }
            if (mu) {
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int u = lane + 64 * i;
                    if (u >= units8) continue;
#pragma unroll
                    for (int e = 0; e < 16; ++e) x[i][e] -= mu[16 * u + e];
                }
            }
            sketch_rotate<NG>(x, lane, units8 * 16);
            if (proj_w) {  // per-row form: the sketch is cut from y - (w . y) w -- the same arithmetic, in the same order, as sketch_rows_kernel
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int u = lane + 64 * i;
                    if (u >= units8) continue;
#pragma unroll
                    for (int e = 0; e < 16; ++e) a = __builtin_fmaf(x[i][e], proj_w[16 * u + e], a);
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
                if (!(fabsf(a) < INFINITY)) a = 0.f;
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int u = lane + 64 * i;
                    if (u >= units8) continue;
#pragma unroll
                    for (int e = 0; e < 16; ++e) x[i][e] = x[i][e] - a * proj_w[16 * u + e];
                }
            }
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(x[i][e]));
        }
    } else {
    const SRC* base = tiled + (size_t)tile * KTS * 1024 * EPU;  // the tile's KTS blocks of 16 KiB are contiguous
    for (int u = threadIdx.x; u < KTS * 1024; u += 256) {
        const SRC* p = base + (size_t)u * EPU;
#pragma unroll
        for (int e = 0; e < EPU; ++e) m = fmaxf(m, fabsf((float)p[e]));
    }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float s = (m > 0.f && m < INFINITY) ? m / 127.0f : 1.0f;
        float* o = tile_info + 4 * tile;
        o[0] = 0.f;
        o[1] = 0.f;
        o[2] = 1.0f / s;
        o[3] = s;
        if (tile_c) tile_c[tile] = 0.f;
    }
}

// w = T mu / ||T mu|| (zeros when the centre is zero): the direction the slack term of the bound is split along (sketch_rows_kernel); one wave
template <bool ROT, int NG>
__global__ __launch_bounds__(64) void center_dir_kernel(const float* __restrict__ mu, int units8, float* __restrict__ wdir) {
    const int lane = threadIdx.x;
    float x[NG][16];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int u = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 16; ++e) x[i][e] = u < units8 ? mu[16 * u + e] : 0.f;
    }
    if constexpr (ROT) sketch_rotate<NG>(x, lane, units8 * 16);
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) n2 += x[i][e] * x[i][e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_xor(n2, off, 64);
    const float inv = n2 > 1e-30f && n2 < INFINITY ? rsqrtf(n2) : 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int u = lane + 64 * i;
        if (u >= units8) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) wdir[16 * u + e] = x[i][e] * inv;
    }
}

// centre of a shard: mean of the `count` rows first, first + stride, ... of a TILED array, element by element -> mu [d8] (zero past the row's padding);
// one workgroup per 16-element unit, 256 threads over the rows
template <typename SRC>
__global__ __launch_bounds__(256) void row_mean_kernel(const SRC* __restrict__ tiled, long long first, long long count, long long stride,
                                                       int KTS, float* __restrict__ mu, float* __restrict__ msq) {
    __shared__ float red[256][17];
    const int u = blockIdx.x;
    float acc[16], acc2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = acc2[e] = 0.f;
    for (long long r = threadIdx.x; r < count; r += 256) {
        float x[16];
        load_sketch_unit<SRC>(tiled, first + r * stride, u, KTS, x);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[e] += x[e];
            acc2[e] = __builtin_fmaf(x[e], x[e], acc2[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 16) {
        float sum = 0.f;
        for (int t = 0; t < 256; ++t) sum += red[t][threadIdx.x];
        const float m = sum / (float)count;
        mu[16 * u + threadIdx.x] = m == m && fabsf(m) < INFINITY ? m : 0.f;  // NaN / Inf rows: no centre in that dimension
    }
    if (msq) {  // mean of the squares, element by element: their sum over the row is the sample's mean ||x||^2 (how collapsed the rows are: ||mu||^2 against it)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[threadIdx.x][e] = acc2[e];
        __syncthreads();
        if (threadIdx.x < 16) {
            float sum = 0.f;
            for (int t = 0; t < 256; ++t) sum += red[t][threadIdx.x];
            const float m = sum / (float)count;
            msq[16 * u + threadIdx.x] = m == m && fabsf(m) < INFINITY ? m : 0.f;
        }
    }
}

}  // namespace

int vqa_launch_row_mean(const void* tiled, int32_t src_dtype, int64_t first, int64_t count, int64_t stride, int32_t d_pad_src, int32_t d_pad8,
                        float* mu, hipStream_t stream, float* msq) {
    VQA_REQUIRE(count > 0 && stride >= 1 && (src_dtype == VQA_F16 || src_dtype == VQA_F32), "row_mean: bad arguments");
    if (src_dtype == VQA_F16)
        hipLaunchKernelGGL(row_mean_kernel<_Float16>, dim3(d_pad8 / 16), dim3(256), 0, stream, reinterpret_cast<const _Float16*>(tiled),
                           (long long)first, (long long)count, (long long)stride, d_pad_src / 32, mu, msq);
    else
        hipLaunchKernelGGL(row_mean_kernel<float>, dim3(d_pad8 / 16), dim3(256), 0, stream, reinterpret_cast<const float*>(tiled),
                           (long long)first, (long long)count, (long long)stride, d_pad_src / 16, mu, msq);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

template <typename SRC>
static int launch_tile_rows_src(const SRC* rows, int64_t first, int64_t count, int64_t valid, int32_t d, int32_t d_pad,
                                int32_t dtype, float scale, void* out, hipStream_t stream) {
    const int esize = dtype == VQA_F32 ? 4 : dtype == VQA_F16 ? 2 : 1;
    const int KT = d_pad * esize / 64;
    const long long total = (long long)count * KT * 4;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    if (dtype == VQA_F16)
        hipLaunchKernelGGL((tile_rows_kernel<SRC, _Float16>), dim3(blocks), dim3(threads), 0, stream, rows, (long long)first,
                           (long long)count, (long long)valid, d, KT, scale, reinterpret_cast<_Float16*>(out));
    else if (dtype == VQA_F32)
        hipLaunchKernelGGL((tile_rows_kernel<SRC, float>), dim3(blocks), dim3(threads), 0, stream, rows, (long long)first,
                           (long long)count, (long long)valid, d, KT, scale, reinterpret_cast<float*>(out));
    else
        hipLaunchKernelGGL((tile_rows_kernel<SRC, uint8_t>), dim3(blocks), dim3(threads), 0, stream, rows, (long long)first,
                           (long long)count, (long long)valid, d, KT, scale, reinterpret_cast<uint8_t*>(out));
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_tile_rows(const void* rows, int32_t src_dtype, int64_t first, int64_t count, int64_t valid, int32_t d,
                         int32_t d_pad, int32_t dtype, float scale, void* out, hipStream_t stream) {
    VQA_REQUIRE(src_dtype == VQA_F32 || src_dtype == VQA_F16, "tile_rows: source element type %d is not f32/f16", src_dtype);
    if (count == 0) return VQA_OK;
    if (src_dtype == VQA_F32)
        return launch_tile_rows_src(reinterpret_cast<const float*>(rows), first, count, valid, d, d_pad, dtype, scale, out, stream);
    return launch_tile_rows_src(reinterpret_cast<const _Float16*>(rows), first, count, valid, d, d_pad, dtype, scale, out, stream);
}

int vqa_launch_untile_rows(const void* tiled, int64_t first, int64_t count, int32_t d, int32_t d_pad, int32_t dtype, void* out,
                           hipStream_t stream) {
    if (count == 0) return VQA_OK;
    const int esize = dtype == VQA_F32 ? 4 : dtype == VQA_F16 ? 2 : 1;
    const int KT = d_pad * esize / 64;
    const long long total = (long long)count * d;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    if (dtype == VQA_F16)
        hipLaunchKernelGGL(untile_rows_kernel<_Float16>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const _Float16*>(tiled), (long long)first, (long long)count, d, KT,
                           reinterpret_cast<_Float16*>(out));
    else if (dtype == VQA_F32)
        hipLaunchKernelGGL(untile_rows_kernel<float>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const float*>(tiled), (long long)first, (long long)count, d, KT,
                           reinterpret_cast<float*>(out));
    else
        hipLaunchKernelGGL(untile_rows_kernel<uint8_t>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const uint8_t*>(tiled), (long long)first, (long long)count, d, KT,
                           reinterpret_cast<uint8_t*>(out));
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

extern "C" int vqa_normalize_convert(const float* rows, int64_t n, int32_t d, int32_t normalize, int32_t dtype, void* out,
                                     void* hip_stream) {
    VQA_REQUIRE(rows && out, "vqa_normalize_convert: null pointer");
    VQA_REQUIRE(n >= 0 && d >= 1, "vqa_normalize_convert: bad shape n=%lld d=%d", (long long)n, d);
    VQA_REQUIRE(dtype == VQA_F32 || dtype == VQA_F16 || dtype == VQA_FP8_E4M3, "vqa_normalize_convert: dtype %d", dtype);
    if (n == 0) return VQA_OK;
    const int threads = 256;
    const long long waves_needed = n;
    long long blocks = (waves_needed + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(normalize_convert_kernel, dim3((int)blocks), dim3(threads), 0, (hipStream_t)hip_stream, rows,
                       (long long)n, d, normalize, dtype, out);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_sketch_rows(const void* tiled, int32_t src_dtype, int64_t first, int64_t count, int32_t d_pad_src, int32_t d_pad8,
                           const float* tile_info, void* out8, float* row_scale, float* row_lo, float* row_norm, bool rotate,
                           const float* mu, bool center, float* row_off, hipStream_t stream, const VqaQueryRows* qr, const SketchSplit* split) {
    if (count == 0) return VQA_OK;
    const SketchSplit sp = split ? *split : SketchSplit();
    QueryStage qs;
    if (qr) {
        VQA_REQUIRE(qr->rows && qr->stage && (qr->src_dtype == VQA_F32 || qr->src_dtype == VQA_F16) && first == 0 && count == VQA_QUERY_TILE,
                    "sketch_rows: bad query staging arguments");
        qs.src = qr->rows;
        qs.src_f32 = qr->src_dtype == VQA_F32;
        qs.valid = qr->valid;
        qs.d = qr->d;
        qs.scale = qr->scale;
        qs.stage = qr->stage;
        qs.rowmajor = qr->rowmajor;
    }
    VQA_REQUIRE(src_dtype == VQA_F16 || src_dtype == VQA_F32, "sketch_rows: source type %d", src_dtype);
    VQA_REQUIRE(d_pad8 / 16 <= 8 * 64, "sketch_rows: rows of %d elements are too long for the int8 sketch", d_pad8);
    const dim3 grid((unsigned)((count + 3) / 4));
    unsigned* tmax = reinterpret_cast<unsigned*>(const_cast<float*>(tile_info));
#define VQA_SKROWS_NG(T, ROT, NGV, KTSV)                                                                                          \
    hipLaunchKernelGGL((sketch_rows_kernel<T, ROT, NGV>), grid, dim3(256), 0, stream, reinterpret_cast<const T*>(tiled), (long long)first, \
                       (long long)count, KTSV, d_pad8 / 64, tile_info, reinterpret_cast<int8_t*>(out8), row_scale, row_lo, row_norm, tmax, mu,  \
                       center ? 1 : 0, row_off, qs, sp)
#define VQA_SKROWS(T, ROT, KTSV)                                                                                                  \
    do {                                                                                                                          \
        if (d_pad8 <= 1024) VQA_SKROWS_NG(T, ROT, 1, KTSV);                                                                       \
        else if (d_pad8 <= 2048) VQA_SKROWS_NG(T, ROT, 2, KTSV);                                                                  \
        else if (d_pad8 <= 4096) VQA_SKROWS_NG(T, ROT, 4, KTSV);                                                                  \
        else VQA_SKROWS_NG(T, ROT, 8, KTSV);                                                                                      \
    } while (0)
    if (src_dtype == VQA_F16) {
        if (rotate) VQA_SKROWS(_Float16, true, d_pad_src / 32);
        else VQA_SKROWS(_Float16, false, d_pad_src / 32);
    } else {
        if (rotate) VQA_SKROWS(float, true, d_pad_src / 16);
        else VQA_SKROWS(float, false, d_pad_src / 16);
    }
#undef VQA_SKROWS
#undef VQA_SKROWS_NG
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_tile_scales(const void* tiled, int32_t src_dtype, int64_t tile0, int64_t ntiles, int32_t d_pad_src, int32_t d_pad8,
                           float* tile_info, bool rotate, const float* mu, hipStream_t stream, float* tile_c, const float* proj_w) {
    if (ntiles == 0) return VQA_OK;
    VQA_REQUIRE(rotate || !proj_w, "tile_scales: the per-row form needs the rotated, centred sketch");
    VQA_REQUIRE(rotate || !mu, "tile_scales: a centre needs the rotated form");
#define VQA_TSCALE_NG(T, ROT, NGV, KTSV)                                                                                       \
    hipLaunchKernelGGL((tile_scale_kernel<T, ROT, NGV>), dim3((unsigned)ntiles), dim3(256), 0, stream, reinterpret_cast<const T*>(tiled), \
                       (long long)tile0, KTSV, d_pad8 / 64, tile_info, mu, tile_c, proj_w)
#define VQA_TSCALE(T, ROT, KTSV)                                                                                               \
    do {                                                                                                                       \
        if (d_pad8 <= 1024) VQA_TSCALE_NG(T, ROT, 1, KTSV);                                                                    \
        else if (d_pad8 <= 2048) VQA_TSCALE_NG(T, ROT, 2, KTSV);                                                               \
        else if (d_pad8 <= 4096) VQA_TSCALE_NG(T, ROT, 4, KTSV);                                                               \
        else VQA_TSCALE_NG(T, ROT, 8, KTSV);                                                                                   \
    } while (0)
    if (src_dtype == VQA_F16) {
        if (rotate) VQA_TSCALE(_Float16, true, d_pad_src / 32);
        else VQA_TSCALE(_Float16, false, d_pad_src / 32);
    } else {
        if (rotate) VQA_TSCALE(float, true, d_pad_src / 16);
        else VQA_TSCALE(float, false, d_pad_src / 16);
    }
#undef VQA_TSCALE
#undef VQA_TSCALE_NG
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_center_dir(const float* mu, int32_t d_pad8, bool rotate, float* wdir, hipStream_t stream) {
    VQA_REQUIRE(mu && wdir && d_pad8 % 16 == 0 && d_pad8 / 16 <= 8 * 64, "center_dir: bad arguments");
    const int units8 = d_pad8 / 16;
#define VQA_CDIR(ROT)                                                                                                        \
    do {                                                                                                                     \
        if (d_pad8 <= 1024) hipLaunchKernelGGL((center_dir_kernel<ROT, 1>), dim3(1), dim3(64), 0, stream, mu, units8, wdir);   \
        else if (d_pad8 <= 2048) hipLaunchKernelGGL((center_dir_kernel<ROT, 2>), dim3(1), dim3(64), 0, stream, mu, units8, wdir); \
        else if (d_pad8 <= 4096) hipLaunchKernelGGL((center_dir_kernel<ROT, 4>), dim3(1), dim3(64), 0, stream, mu, units8, wdir); \
        else hipLaunchKernelGGL((center_dir_kernel<ROT, 8>), dim3(1), dim3(64), 0, stream, mu, units8, wdir);                   \
    } while (0)
    if (rotate) VQA_CDIR(true);
    else VQA_CDIR(false);
#undef VQA_CDIR
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}
