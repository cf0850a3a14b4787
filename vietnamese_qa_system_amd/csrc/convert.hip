// Element-wise helpers around the index (HBM-bound, 16 B per lane where the shape allows), gfx950 only.
//   * stage_queries : [nq, d] fp32|fp16 queries -> zero padded [256, d_pad] tile in the index element type
//   * pad_rows      : [n, d] -> [n, d_pad] zero padded copy (index rows whose length is not a multiple of 64)
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

template <typename SRC>
__global__ void stage_queries_kernel(const SRC* __restrict__ q, int nq, int d, int d_pad, int dtype, void* __restrict__ out) {
    const int total = VQA_QUERY_TILE * d_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / d_pad, j = i - r * d_pad;
        const float v = (r < nq && j < d) ? (float)q[(size_t)r * d + j] : 0.0f;
        if (dtype == VQA_F16) reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
        else if (dtype == VQA_F32) reinterpret_cast<float*>(out)[i] = v;
        else reinterpret_cast<uint8_t*>(out)[i] = f32_to_e4m3(v);
    }
}

template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ rows, long long n, int d, int d_pad, T* __restrict__ out) {
    const long long total = n * d_pad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / d_pad;
        const int j = (int)(i - r * d_pad);
        out[i] = j < d ? rows[r * d + j] : T(0);
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

int vqa_launch_stage_queries(const void* q, int32_t q_dtype, int32_t nq, int32_t d, int32_t d_pad, int32_t dtype, void* out,
                             hipStream_t stream) {
    VQA_REQUIRE(q_dtype == VQA_F32 || q_dtype == VQA_F16, "query element type %d is not f32/f16", q_dtype);
    const int threads = 256, blocks = (VQA_QUERY_TILE * d_pad + threads - 1) / threads;
    if (q_dtype == VQA_F32)
        hipLaunchKernelGGL(stage_queries_kernel<float>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const float*>(q), nq, d, d_pad, dtype, out);
    else
        hipLaunchKernelGGL(stage_queries_kernel<_Float16>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const _Float16*>(q), nq, d, d_pad, dtype, out);
    VQA_HIP_CHECK(hipGetLastError());
    return VQA_OK;
}

int vqa_launch_pad_rows(const void* rows, int64_t n, int32_t d, int32_t d_pad, int32_t elem_bytes, void* out,
                        hipStream_t stream) {
    const int threads = 256;
    const long long total = (long long)n * d_pad;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    if (elem_bytes == 2)
        hipLaunchKernelGGL(pad_rows_kernel<uint16_t>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const uint16_t*>(rows), (long long)n, d, d_pad, reinterpret_cast<uint16_t*>(out));
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(pad_rows_kernel<uint32_t>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const uint32_t*>(rows), (long long)n, d, d_pad, reinterpret_cast<uint32_t*>(out));
    else
        hipLaunchKernelGGL(pad_rows_kernel<uint8_t>, dim3(blocks), dim3(threads), 0, stream,
                           reinterpret_cast<const uint8_t*>(rows), (long long)n, d, d_pad, reinterpret_cast<uint8_t*>(out));
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
