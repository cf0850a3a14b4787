// Element-wise helpers around the index (HBM-bound, 16 B per lane where the shape allows), gfx950 only.
//   * stage_queries : [nq, d] fp32|fp16 queries -> one zero padded 256-row tile in the index element type, TILED layout
//   * tile_rows     : row-major [count, d] rows -> the TILED layout of the index (see vqa_common.h), zero padded
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

// TILED layout (fp16): tile t (256 rows), K-step kappa (32 elements) is one contiguous 16 KiB block at unit index
// (t * KT + kappa) * 1024 (unit = 16 B = 8 elements); inside the block the unit of (row r, 8-element slot s) sits at
// r * 4 + (s ^ 3 * bit3(r)) -- exactly the bank-conflict-free LDS image K1 wants, so K1's LDS-DMA reads 1 KiB
// contiguous per wave-instruction and a tile's K-steps stream from HBM as one sequential 256 * d_pad * 2 byte run.
__device__ __forceinline__ size_t tiled_unit(long long row, int kappa, int slot, int KT) {
    const long long t = row >> 8;
    const int r = (int)(row & 255);
    return ((size_t)t * KT + kappa) * 1024 + (size_t)(r * 4 + (slot ^ (((r >> 3) & 1) * 3)));
}

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// rows: [count, d] row-major, SRC element type; row i goes to index row first + i.  One thread per 16-byte output unit.
template <typename SRC>
__global__ void tile_rows_kernel(const SRC* __restrict__ rows, long long first, long long count, long long valid, int d,
                                 int KT, _Float16* __restrict__ out) {
    const int units_per_row = KT * 4;
    const long long total = count * units_per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ri = i / units_per_row;
        const int u = (int)(i - ri * units_per_row);
        const int kappa = u >> 2, slot = u & 3;
        const int j0 = u * 8;
        half8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (ri < valid && j0 + e < d) ? (_Float16)rows[ri * d + j0 + e] : (_Float16)0;
        *reinterpret_cast<half8_t*>(out + tiled_unit(first + ri, kappa, slot, KT) * 8) = v;
    }
}

// inverse of tile_rows: TILED fp16 -> row-major [count, d] fp16 (index export for Embeddings.save)
__global__ void untile_rows_kernel(const _Float16* __restrict__ tiled, long long first, long long count, int d, int KT,
                                   _Float16* __restrict__ out) {
    const long long total = count * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ri = i / d;
        const int j = (int)(i - ri * d);
        out[i] = tiled[tiled_unit(first + ri, j >> 5, (j >> 3) & 3, KT) * 8 + (j & 7)];
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

}  // namespace

int vqa_launch_tile_rows(const void* rows, int32_t src_dtype, int64_t first, int64_t count, int64_t valid, int32_t d,
                         int32_t d_pad, void* out, hipStream_t stream) {
    VQA_REQUIRE(src_dtype == VQA_F32 || src_dtype == VQA_F16, "tile_rows: source element type %d is not f32/f16", src_dtype);
    if (count == 0) return VQA_OK;
    const int KT = d_pad / 32;
    const long long total = (long long)count * KT * 4;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    if (src_dtype == VQA_F32)
        hipLaunchKernelGGL(tile_rows_kernel<float>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const float*>(rows), (long long)first, (long long)count, (long long)valid, d, KT,
                           reinterpret_cast<_Float16*>(out));
    else
        hipLaunchKernelGGL(tile_rows_kernel<_Float16>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const _Float16*>(rows), (long long)first, (long long)count, (long long)valid, d, KT,
                           reinterpret_cast<_Float16*>(out));
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_untile_rows(const void* tiled, int64_t first, int64_t count, int32_t d, int32_t d_pad, void* out,
                           hipStream_t stream) {
    if (count == 0) return VQA_OK;
    const long long total = (long long)count * d;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    hipLaunchKernelGGL(untile_rows_kernel, dim3(blocks), dim3(threads), 0, stream, reinterpret_cast<const _Float16*>(tiled),
                       (long long)first, (long long)count, d, d_pad / 32, reinterpret_cast<_Float16*>(out));
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
