// C ABI glue: index handles, search orchestration (stage queries -> K1 pass(es) -> K2 merge), error plumbing.
// Entry points are declared in include/vqa_retrieval.h (each cites the reference interface it replaces).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "vqa_common.h"

static thread_local char g_err[1024] = "";

void vqa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vqa_last_error(void) { return g_err; }
extern "C" int vqa_version(void) { return VQA_VERSION; }

static int elem_bytes(int dtype) { return dtype == VQA_F32 ? 4 : dtype == VQA_F16 ? 2 : 1; }

struct vqa_index {
    int device = 0;
    int64_t n = 0;
    int32_t d = 0, d_pad = 0, dtype = 0;
    void* rows = nullptr;  // [n, d_pad]
    bool rows_owned = false;
    int64_t* ids = nullptr;  // [n] or null
    int64_t id_base = 0;
    int num_cu = 0;
    int max_grid = 0;
    bool two_pass = true;
    // workspace (allocated once; search never allocates)
    void* q_stage = nullptr;     // [256, d_pad] index element type
    vqa_key* partial = nullptr;  // [2 * max_grid, 256, max_k]
    float* thr0 = nullptr;       // [256]
    // opt-in kernel timing (bench.py): event pairs around the main scoring kernel
    bool timing = false;
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t ev_used = 0;
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

extern "C" void vqa_index_destroy(vqa_index* ix) {
    if (!ix) return;
    DeviceGuard g(ix->device);
    if (ix->rows_owned && ix->rows) (void)hipFree(ix->rows);
    if (ix->ids) (void)hipFree(ix->ids);
    if (ix->q_stage) (void)hipFree(ix->q_stage);
    if (ix->partial) (void)hipFree(ix->partial);
    if (ix->thr0) (void)hipFree(ix->thr0);
    for (hipEvent_t e : ix->ev) (void)hipEventDestroy(e);
    delete ix;
}

extern "C" int vqa_index_create(vqa_index** out, int device, int64_t n, int32_t d, int32_t dtype, const void* rows,
                                const int64_t* ids_or_null, int64_t id_base, uint32_t flags) {
    VQA_REQUIRE(out, "vqa_index_create: out is null");
    *out = nullptr;
    VQA_REQUIRE(rows || n == 0, "vqa_index_create: rows is null");
    VQA_REQUIRE(n >= 0 && n < 0xFFFFFFFFll, "vqa_index_create: n=%lld outside [0, 2^32-1) rows per shard", (long long)n);
    VQA_REQUIRE(d >= 1 && d <= 65536, "vqa_index_create: d=%d", d);
    VQA_REQUIRE(dtype == VQA_F32 || dtype == VQA_F16 || dtype == VQA_FP8_E4M3, "vqa_index_create: dtype %d", dtype);
    VQA_REQUIRE(dtype == VQA_F16, "vqa_index_create: only VQA_F16 rows are implemented in this build (dtype %d)", dtype);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        vqa_set_error("vqa_index_create: no HIP device visible");
        return VQA_ENODEV;
    }
    VQA_REQUIRE(device >= 0 && device < ndev, "vqa_index_create: device %d of %d", device, ndev);
    hipDeviceProp_t prop;
    VQA_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vqa_set_error("vqa_index_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return VQA_ENODEV;
    }
    DeviceGuard guard(device);
    vqa_index* ix = new (std::nothrow) vqa_index();
    if (!ix) {
        vqa_set_error("vqa_index_create: host allocation failed");
        return VQA_ENOMEM;
    }
    ix->device = device;
    ix->n = n;
    ix->d = d;
    ix->d_pad = (d + 63) / 64 * 64;
    ix->dtype = dtype;
    ix->id_base = id_base;
    ix->num_cu = prop.multiProcessorCount;
    ix->max_grid = ix->num_cu;
    const char* tp = getenv("VQA_TWO_PASS");
    ix->two_pass = !(tp && tp[0] == '0');
    const int eb = elem_bytes(dtype);
    const bool borrow = (flags & VQA_ROWS_BORROW) != 0;
    int rc = VQA_OK;
    do {
        if (borrow) {
            if (ix->d_pad != d) {
                vqa_set_error("vqa_index_create: VQA_ROWS_BORROW needs d %% 64 == 0 (d=%d)", d);
                rc = VQA_EINVAL;
                break;
            }
            ix->rows = const_cast<void*>(rows);
            ix->rows_owned = false;
        } else if (n > 0) {
            const size_t bytes = (size_t)n * ix->d_pad * eb;
            if (hipMalloc(&ix->rows, bytes) != hipSuccess) {
                vqa_set_error("vqa_index_create: hipMalloc of %zu bytes for the rows failed", bytes);
                rc = VQA_ENOMEM;
                break;
            }
            ix->rows_owned = true;
            if (ix->d_pad == d) {
                if (hipMemcpy(ix->rows, rows, bytes, hipMemcpyDefault) != hipSuccess) {
                    vqa_set_error("vqa_index_create: copying the rows failed");
                    rc = VQA_EHIP;
                    break;
                }
            } else {
                void* tmp = nullptr;
                const size_t src_bytes = (size_t)n * d * eb;
                if (hipMalloc(&tmp, src_bytes) != hipSuccess) {
                    vqa_set_error("vqa_index_create: hipMalloc of %zu staging bytes failed", src_bytes);
                    rc = VQA_ENOMEM;
                    break;
                }
                hipError_t e = hipMemcpy(tmp, rows, src_bytes, hipMemcpyDefault);
                if (e == hipSuccess) {
                    rc = vqa_launch_pad_rows(tmp, n, d, ix->d_pad, eb, ix->rows, nullptr);
                    if (rc == VQA_OK) e = hipDeviceSynchronize();
                }
                (void)hipFree(tmp);
                if (rc != VQA_OK) break;
                if (e != hipSuccess) {
                    vqa_set_error("vqa_index_create: padding the rows failed: %s", hipGetErrorString(e));
                    rc = VQA_EHIP;
                    break;
                }
            }
        }
        if (ids_or_null && n > 0) {
            if (hipMalloc((void**)&ix->ids, (size_t)n * 8) != hipSuccess) {
                vqa_set_error("vqa_index_create: hipMalloc for %lld ids failed", (long long)n);
                rc = VQA_ENOMEM;
                break;
            }
            if (hipMemcpy(ix->ids, ids_or_null, (size_t)n * 8, hipMemcpyDefault) != hipSuccess) {
                vqa_set_error("vqa_index_create: copying the ids failed");
                rc = VQA_EHIP;
                break;
            }
        }
        const int max_k = vqa_score_topk_max_k(dtype);
        if (hipMalloc(&ix->q_stage, (size_t)VQA_QUERY_TILE * ix->d_pad * eb) != hipSuccess ||
            hipMalloc((void**)&ix->partial, (size_t)2 * ix->max_grid * VQA_QUERY_TILE * max_k * sizeof(vqa_key)) != hipSuccess ||
            hipMalloc((void**)&ix->thr0, VQA_QUERY_TILE * sizeof(float)) != hipSuccess) {
            vqa_set_error("vqa_index_create: workspace allocation failed");
            rc = VQA_ENOMEM;
            break;
        }
    } while (0);
    if (rc != VQA_OK) {
        vqa_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return VQA_OK;
}

extern "C" int64_t vqa_index_size(const vqa_index* ix) { return ix ? ix->n : -1; }
extern "C" int32_t vqa_index_dim(const vqa_index* ix) { return ix ? ix->d : -1; }
extern "C" int32_t vqa_index_dtype(const vqa_index* ix) { return ix ? ix->dtype : -1; }

struct LaunchPlan {
    int tiles = 0;   // corpus tiles of 256 rows
    int grid0 = 0;   // workgroups of the seeding pass (0 = single pass)
    int grid1 = 0;   // workgroups of the main pass
};

static LaunchPlan plan_launch(const vqa_index* ix) {
    LaunchPlan p;
    p.tiles = (int)((ix->n + 255) / 256);
    p.grid1 = p.tiles < ix->max_grid ? p.tiles : ix->max_grid;
    // Two passes when every workgroup has several tiles: the first `grid` tiles are searched alone, their exact
    // k-th best score per query seeds the thresholds of the pass over the remaining tiles, so the main pass
    // appends ~10 / (256 * grid) of the scores instead of starting every workgroup from -inf.
    if (ix->two_pass && p.tiles >= 8 * ix->max_grid) {
        p.grid0 = ix->max_grid;
        const int rest = p.tiles - p.grid0;
        p.grid1 = rest < ix->max_grid ? rest : ix->max_grid;
    }
    return p;
}

extern "C" int vqa_index_launch_info(const vqa_index* ix, int32_t B, int32_t k, vqa_launch_info* out) {
    VQA_REQUIRE(ix && out, "vqa_index_launch_info: null pointer");
    const LaunchPlan p = plan_launch(ix);
    (void)B;
    out->grid = p.grid1;
    out->block = 512;
    out->lds_bytes = vqa_score_topk_lds_bytes(ix->dtype, k);
    out->rows_per_tile = 256;
    const int64_t seed_rows = (int64_t)p.grid0 * 256 < ix->n ? (int64_t)p.grid0 * 256 : ix->n;
    out->rows_per_launch = ix->n - seed_rows;
    out->bytes_per_launch = out->rows_per_launch * (int64_t)ix->d * elem_bytes(ix->dtype);
    out->flops_per_launch = 2 * (int64_t)VQA_QUERY_TILE * out->rows_per_launch * (int64_t)ix->d;
    out->seed_grid = p.grid0;
    out->reserved = 0;
    return VQA_OK;
}

extern "C" int vqa_index_set_timing(vqa_index* ix, int32_t enabled) {
    VQA_REQUIRE(ix, "vqa_index_set_timing: index is null");
    ix->timing = enabled != 0;
    ix->ev_used = 0;
    return VQA_OK;
}

extern "C" int vqa_index_get_timing(vqa_index* ix, double* kernel_ms_sum, int64_t* launches) {
    VQA_REQUIRE(ix && kernel_ms_sum && launches, "vqa_index_get_timing: null pointer");
    DeviceGuard guard(ix->device);
    double sum = 0.0;
    for (size_t i = 0; i + 1 < ix->ev_used; i += 2) {
        VQA_HIP_CHECK(hipEventSynchronize(ix->ev[i + 1]));
        float ms = 0.f;
        VQA_HIP_CHECK(hipEventElapsedTime(&ms, ix->ev[i], ix->ev[i + 1]));
        sum += ms;
    }
    *kernel_ms_sum = sum;
    *launches = (int64_t)(ix->ev_used / 2);
    ix->ev_used = 0;
    return VQA_OK;
}

static int timing_event(vqa_index* ix, hipStream_t stream) {
    if (ix->ev_used == ix->ev.size()) {
        hipEvent_t e;
        VQA_HIP_CHECK(hipEventCreate(&e));
        ix->ev.push_back(e);
    }
    VQA_HIP_CHECK(hipEventRecord(ix->ev[ix->ev_used++], stream));
    return VQA_OK;
}

extern "C" int vqa_index_search(vqa_index* ix, const void* q, int32_t q_dtype, int32_t B, int32_t k, float* out_scores,
                                int64_t* out_ids, int64_t* out_pos_or_null, void* hip_stream) {
    VQA_REQUIRE(ix, "vqa_index_search: index is null");
    VQA_REQUIRE(q && out_scores && out_ids, "vqa_index_search: null pointer");
    VQA_REQUIRE(B >= 1, "vqa_index_search: B=%d", B);
    const int max_k = vqa_score_topk_max_k(ix->dtype);
    VQA_REQUIRE(k >= 1 && k <= max_k, "vqa_index_search: k=%d outside [1, %d]", k, max_k);
    VQA_REQUIRE(q_dtype == VQA_F32 || q_dtype == VQA_F16, "vqa_index_search: q_dtype %d is not f32/f16", q_dtype);
    hipStream_t stream = (hipStream_t)hip_stream;
    DeviceGuard guard(ix->device);
    const int qeb = q_dtype == VQA_F32 ? 4 : 2;
    const LaunchPlan p = plan_launch(ix);
    for (int q0 = 0; q0 < B; q0 += VQA_QUERY_TILE) {
        const int nq = B - q0 < VQA_QUERY_TILE ? B - q0 : VQA_QUERY_TILE;
        float* os = out_scores + (size_t)q0 * k;
        int64_t* oi = out_ids + (size_t)q0 * k;
        int64_t* op = out_pos_or_null ? out_pos_or_null + (size_t)q0 * k : nullptr;
        if (ix->n == 0) {  // empty shard: every slot is padding
            // reuse the merge kernel on one all-empty partial list
            VQA_HIP_CHECK(hipMemsetAsync(ix->partial, 0, (size_t)VQA_QUERY_TILE * k * sizeof(vqa_key), stream));
            int rc = vqa_launch_merge_partials(ix->partial, 1, nq, k, ix->ids, ix->id_base, os, oi, op, nullptr, k, stream);
            if (rc != VQA_OK) return rc;
            continue;
        }
        int rc = vqa_launch_stage_queries(reinterpret_cast<const char*>(q) + (size_t)q0 * ix->d * qeb, q_dtype, nq, ix->d,
                                          ix->d_pad, ix->dtype, ix->q_stage, stream);
        if (rc != VQA_OK) return rc;
        ScoreTopkArgs a;
        a.x = ix->rows;
        a.q = ix->q_stage;
        a.n = ix->n;
        a.d_pad = ix->d_pad;
        a.nq = nq;
        a.k = k;
        int parts = 0;
        if (p.grid0 > 0) {
            a.thr_init = nullptr;
            a.partial = ix->partial;
            a.tile_begin = 0;
            a.tile_end = p.grid0;
            a.grid = p.grid0;
            rc = vqa_launch_score_topk(ix->dtype, a, stream);
            if (rc != VQA_OK) return rc;
            rc = vqa_launch_merge_partials(ix->partial, p.grid0, nq, k, nullptr, 0, nullptr, nullptr, nullptr, ix->thr0, k,
                                           stream);
            if (rc != VQA_OK) return rc;
            parts = p.grid0;
        }
        a.thr_init = p.grid0 > 0 ? ix->thr0 : nullptr;
        a.partial = ix->partial + (size_t)parts * VQA_QUERY_TILE * k;
        a.tile_begin = p.grid0;
        a.tile_end = p.tiles;
        a.grid = p.grid1;
        if (ix->timing && (rc = timing_event(ix, stream)) != VQA_OK) return rc;
        rc = vqa_launch_score_topk(ix->dtype, a, stream);
        if (rc != VQA_OK) return rc;
        if (ix->timing && (rc = timing_event(ix, stream)) != VQA_OK) return rc;
        parts += p.grid1;
        rc = vqa_launch_merge_partials(ix->partial, parts, nq, k, ix->ids, ix->id_base, os, oi, op, nullptr, k, stream);
        if (rc != VQA_OK) return rc;
    }
    return VQA_OK;
}
