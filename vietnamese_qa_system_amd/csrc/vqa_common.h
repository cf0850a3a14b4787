// Shared host/device helpers for libvqa_retrieval (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "vqa_retrieval.h"

// ---- error plumbing (thread-local message, negative codes; nothing throws across the ABI) -----------------
void vqa_set_error(const char* fmt, ...);
#define VQA_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            vqa_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VQA_EHIP;                                                                      \
        }                                                                                         \
    } while (0)
#define VQA_REQUIRE(cond, ...)       \
    do {                             \
        if (!(cond)) {               \
            vqa_set_error(__VA_ARGS__); \
            return VQA_EINVAL;       \
        }                            \
    } while (0)

// ---- development overrides.  The product build reads NO environment variable: every knob is a field of vqa_index_options /
// vqa_encoder_options.  A `-DVQA_DEV` variant library (scripts/: A/B runs, timing ablations) still overlays VQA_* variables on those
// defaults and honours the GEMM launcher's dev switches; vqa_dev_env is the one place that decides.
#include <stdlib.h>
static inline const char* vqa_dev_env(const char* name) {
#ifdef VQA_DEV
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- one-time setup per device (hipFuncSetAttribute, device queries), callable from several host threads at once:
// different handles may be used from different threads (include/vqa_retrieval.h), and every launch path comes through here.
struct VqaPerDeviceOnce {
    std::once_flag flag[64];
    int rc[64] = {};
    template <typename F>
    int run(F&& fn) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) {
            vqa_set_error("hipGetDevice failed");
            return VQA_EHIP;
        }
        const int slot = dev & 63;
        std::call_once(flag[slot], [&] { rc[slot] = fn(dev); });
        if (rc[slot] != VQA_OK) vqa_set_error("one-time kernel setup failed on device %d", dev);
        return rc[slot];
    }
};

// ---- candidate keys -----------------------------------------------------------------------------------------
// A candidate (score, row position) is one u64 whose unsigned order is the result order
// "score descending, then row position ascending":  key = ordered(score) << 32 | (0xFFFFFFFF - pos).
// key 0 is "empty" (smaller than every real candidate, including score = -inf).
typedef unsigned long long vqa_key;

__host__ __device__ __forceinline__ uint32_t vqa_f32_ordered(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float vqa_ordered_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ __forceinline__ vqa_key vqa_make_key(float score, uint32_t pos) {
    return ((vqa_key)vqa_f32_ordered(score) << 32) | (vqa_key)(0xFFFFFFFFu - pos);
}
__host__ __device__ __forceinline__ float vqa_key_score(vqa_key k) { return vqa_ordered_f32((uint32_t)(k >> 32)); }
__host__ __device__ __forceinline__ uint32_t vqa_key_pos(vqa_key k) { return 0xFFFFFFFFu - (uint32_t)k; }

// Largest key of the wave, in every lane.  Four DPP steps inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror: the partner's two halves arrive with one VALU move each), then v_permlane16_swap / v_permlane32_swap of a register
// with itself across the rows -- no LDS crossbar (six 64-bit __shfl_xor steps are twelve dependent ds_bpermute round trips, which
// was most of a selection round).
template <int CTRL>
__device__ __forceinline__ vqa_key vqa_dpp_max_step(vqa_key v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const vqa_key o = ((vqa_key)ohi << 32) | olo;
    return o > v ? o : v;
}

__device__ __forceinline__ vqa_key vqa_wave_max_key(vqa_key v) {
    v = vqa_dpp_max_step<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v = vqa_dpp_max_step<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v = vqa_dpp_max_step<0x141>(v);  // row_half_mirror
    v = vqa_dpp_max_step<0x140>(v);  // row_mirror: every lane holds its row's maximum
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // {rows 0 0 2 2, rows 1 1 3 3}
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const vqa_key x = ((vqa_key)(unsigned)b[0] << 32) | (unsigned)a[0], y = ((vqa_key)(unsigned)b[1] << 32) | (unsigned)a[1];
        v = x > y ? x : y;
    }
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);  // {lower half twice, upper half twice}
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const vqa_key x = ((vqa_key)(unsigned)b[0] << 32) | (unsigned)a[0], y = ((vqa_key)(unsigned)b[1] << 32) | (unsigned)a[1];
        v = x > y ? x : y;
    }
    return v;
}

// internal storage code of the int8 sketch of a large fp16 shard (never a public index type: include/vqa_retrieval.h)
#define VQA_I8_SKETCH 3

// MODE 2 of the scoring kernel: the sketch scan's extra arguments
// A query's candidate list of a sketch search is kept as kSketchSubLists sub-lists with a counter each; the re-scoring kernel picks
// the sub-list by the pair's index in its scan region, so the pairs of one region (near-duplicates stored side by side) spread
// over all of them.  One counter per query took every append of a search through 8 cache lines of L2 atomics: 110-130 us of a
// 290 us kernel at 10M rows.
constexpr int kSketchSubLists = 16;
// ... and every counter sits on a 128-byte line of its own (stride in counters): atomics to ONE line serialise at the L2, and 4096
// counters packed into 128 lines still took ~1700 appends per line of a 10M-row search
constexpr int kSketchCntStride = 32;

constexpr int kSketchQRows = 6;  // per-query constants of a sketch scan
struct SketchScanArgs {
    const float4* tile_info = nullptr;  // [tiles] (max ||x_hi||, max ||x_lo||, 1 / scale, scale) of every 256-row tile of the sketch
    const float* qconst = nullptr;      // [kSketchQRows][256]: theta (exact lower bound of the k-th best score), ||q_lo||, ||z_r||, 1 / s_q, |alpha|
    const float* tile_c = nullptr;      // [tiles] max |w . x_lo| of every tile (the split slack term, convert.hip) or nullptr (taken as 0);
                                        // per-row form: max |beta| of the tile
    const float* beta = nullptr;        // per-row form (LOOP 2): [tiles * 256] beta = w . y of every row; qconst row 4 = signed alpha, row 5 = margin factor
    unsigned long long* regions = nullptr;  // [grid][cap] candidate (query << 32 | row position) pairs, one region per workgroup
    unsigned* counts = nullptr;         // [grid] pairs written per region
    int* overflow = nullptr;            // set when a region filled up: the caller's exact fallback scan runs
    int cap = 0;
};

// ---- kernel launchers implemented in the .hip files ---------------------------------------------------------
struct ScoreTopkArgs {
    const void* x = nullptr;  // index rows in TILED layout (convert.hip): ceil(n/256) tiles x (d_pad/32) blocks of 16 KiB
    const void* q = nullptr;  // staged query tile, same layout (one tile), zero padded
    const float* thr_init = nullptr;  // [VQA_QUERY_TILE] starting thresholds or nullptr (-inf)
    const vqa_key* upper = nullptr;  // [VQA_QUERY_TILE] exclusive upper bound keys (continuation passes) or nullptr
    vqa_key* partial = nullptr;  // main pass: [VQA_QUERY_TILE, grid, k] per-workgroup sorted partial lists, query-major (output)
    int64_t n = 0;        // rows in the shard
    int32_t d_pad = 0;    // padded row length in elements (multiple of 64)
    int32_t nq = 0;       // valid queries in the tile (1..VQA_QUERY_TILE)
    int32_t k = 0;
    int32_t tile_begin = 0;  // first corpus tile (of 256 rows) this launch covers
    int32_t tile_end = 0;  // one past the last
    int32_t grid = 0;     // workgroups
    const int* gate = nullptr;  // device flag or nullptr: the kernel returns at once when *gate == 0
    int32_t row_lists = 0;      // main pass flush: lists per query row of `partial` (0 = grid) ...
    int32_t list_offset = 0;    // ... and the slot of this launch's workgroup 0 inside the row (two-stage search: the first
                                // stage fills slots [0, grid), the main launch [grid, 2 grid) of rows of 2 grid lists)
    bool first_stage = false;   // the first-stage launch of a two-stage search: same code, its own kernel symbol
    int32_t loop = 0;           // fp16: 0 = anti-phase slot loop, 1 = K-step-pair stagger loop (the fp8 structure); sketch scan: 1 = five
                                // X ring stages instead of six
    const SketchScanArgs* sketch = nullptr;  // not null: MODE 2 over the int8 sketch (x = sketch rows, q = sketch of the query tile)
    bool regq = true;           // sketch scans: the register-resident-query kernel (scan_regq.hip) where it applies
    bool seed_only = false;  // MODE 0: writes seeds_per_tile sub-maxima per query and tile to `partial` as [query][tile - tile_begin][.]
    int32_t seeds_per_tile = 2;  // 2 (one per 128-row half) or 8 (one per 32-row group: shards of a few tiles, where 2 per tile
                                 // are fewer than k values and leave the thresholds at -inf)
};
int vqa_launch_score_topk(int dtype, const ScoreTopkArgs& a, hipStream_t stream);
// scan_regq.hip: the sketch scan with the query operand resident in registers (rows of 768 / 384 one-byte elements, no per-row betas)
bool vqa_sketch_regq_applies(const ScoreTopkArgs& a);
int vqa_launch_sketch_regq(const ScoreTopkArgs& a, hipStream_t stream);
int vqa_score_topk_lds_bytes(int dtype, int k);
int vqa_score_topk_max_k(int dtype);
int vqa_score_topk_seeds_per_tile();
int vqa_score_topk_sketch_max_tiles();  // tiles one workgroup of the sketch scan can take (its tile maxima sit in LDS)

// What a merge of the sketch search's cascade does besides selecting (every pointer null: nothing).  qconst: the per-query constants
// of the NEXT sketch scan, from this merge's k-th score (what sketch_qconst_kernel computes as a launch of its own), with the reset of
// the candidate counters / overflow flags when `clear`.  flag_mirror: the search's last merge copies the overflow flags and the
// call's number to the host-visible mirror (system-scope stores, the number last).
struct MergeSketchTail {
    float* qconst = nullptr;
    const float *qscale = nullptr, *qlo = nullptr, *qnorm = nullptr, *qoff = nullptr;
    const float *qalpha = nullptr, *qrnorm = nullptr;  // the split slack term: |alpha|, ||z_r|| per query (nullptr: 0, ||q||)
    float fp_margin = 0.f, mu_margin = 0.f;  // mu_margin: 3e-7 ||mu|| (the rounding of q . mu, an fp64 dot, and of theta - q . mu)
    unsigned* cand_cnt = nullptr;
    int* overflow = nullptr;
    int clear = 0, seq = 0;
    int* flag_mirror = nullptr;
    int mirror_before_gate = 0;  // 1: this launch only REPORTS -- block 0 copies the flags to flag_mirror before the gate test, whatever the gate
                                 // says.  The cascade's last selection raises overflow[0] from any of its blocks, so the report is written by the
                                 // launch BEHIND it (the gated fallback's merge), when every block of the selection has finished
    const float* min_score = nullptr;  // [nq] or nullptr: keys scoring below it are dropped before the selection (the cascade's last merge:
                                       // theta1, the exact k-th best score of the first stage -- no key below it can be among the k best);
                                       // a list that still overflows the kernel's LDS raises overflow[0]: the search's exact fallback runs
};
float vqa_sketch_fp_margin(int32_t d, bool rotated, bool per_row = false);  // sketch.hip: the margin vqa_launch_sketch_qconst uses

// `parts` key lists of `list_len` keys per query ([parts][256][list_len], or query-major) -> the k best per query:
// final [nq, k] (scores, external ids, positions) and/or the k-th best score per query (-inf when fewer exist)
int vqa_launch_merge_partials(const vqa_key* partial, int32_t parts, int32_t list_len, int32_t nq, int32_t k,
                              const int64_t* ids, int64_t id_base, float* out_scores, int64_t* out_ids, int64_t* out_pos,
                              float* out_thr, float score_scale /* applied to out_scores only (power of two) */,
                              int32_t out_stride /* row stride of the out arrays */, int32_t out_offset /* first column */,
                              vqa_key* out_last_key /* [nq] k-th key per query (0 when fewer exist) or nullptr */,
                              bool query_major /* lists are [query][parts][list_len] instead of [parts][256][list_len] */,
                              const int* gate /* device flag or nullptr: no-op when *gate == 0 */, hipStream_t stream,
                              int32_t row_lists = 0 /* query-major only: lists per query row in memory (>= parts; 0 = parts):
                                                       the first `parts` lists of every row are merged */,
                              const unsigned* counts = nullptr /* [nq][parts] or nullptr (query-major lists): only the first
                                                                  counts[q][part] slots of every list hold keys of this search
                                                                  (the candidate sub-lists of a sketch search) */,
                              int32_t count_stride = 1 /* counters between two entries of `counts` */,
                              const MergeSketchTail* tail = nullptr);
// one-pass large-k check: sets *flag = 1 when some workgroup's list (list_len keys, full) ends ABOVE the query's k-th merged
// key `kth` -- that list may have dropped a row of the true top-k (capi.hip, vqa_index_search)
int vqa_launch_verify_wide(const vqa_key* partial, int32_t parts, int32_t list_len, int32_t nq, const vqa_key* kth, int* flag,
                           hipStream_t stream);

// row-major [valid, d] f32|f16 rows (device) -> TILED layout, storage type `dtype`, at rows [first, first + count);
// rows valid..count-1 are written as zeros (query tile: first = 0, count = 256, valid = nq); values are multiplied by
// `scale` before conversion
int vqa_launch_tile_rows(const void* rows, int32_t src_dtype, int64_t first, int64_t count, int64_t valid, int32_t d,
                         int32_t d_pad, int32_t dtype, float scale, void* out, hipStream_t stream);
// TILED rows [first, first + count) -> row-major [count, d] in the storage type (device)
int vqa_launch_untile_rows(const void* tiled, int64_t first, int64_t count, int32_t d, int32_t d_pad, int32_t dtype, void* out,
                           hipStream_t stream);

// ---- int8 sketch (large fp16 shards: the rigorous pruning pre-pass, score_topk.hip MODE 2) -------------------------------------
// TILED fp16 / fp32 rows [first, first + count) -> TILED int8 (K-blocks of 64 elements).  tile_info [tiles][4] floats = (max ||x_hi||,
// max ||x_lo||, 1 / scale, scale) per 256-row tile: not null (index rows) -> the rows take their tile's scale and raise its two
// maxima; null (the query tile) -> every row its own max|x| / 127, with per-row scale / ||x_lo|| / ||x|| outputs
// the query tile of a search, staged AND sketched by one launch (vqa_launch_sketch_rows with `qr`): the caller's row-major rows ->
// storage type -> the TILED tile `stage` (+ its row-major copy) -> sketch of those stored values
// The slack term |z . x_lo| split along w (convert.hip sketch_rows_kernel): index rows raise tile_c[tile] = max |w . x_lo|, query rows
// report |alpha| = |z . w| and ||z - alpha w||
struct SketchSplit {
    const float* wdir = nullptr;  // [d_pad8] the shard's rotated, normalised centre (vqa_launch_center_dir) or nullptr: no split
    float* tile_c = nullptr;      // [tiles], index rows
    float* row_alpha = nullptr;   // [count], query rows
    float* row_rnorm = nullptr;   // [count], query rows
    // per-row form: rows and queries are projected off w before they are sketched; index rows store beta = w . y (beta [tiles * 256],
    // tile_c = max |beta|), query rows report the SIGNED alpha; the scan adds alpha beta per (query, row) (score_topk.hip LOOP 2)
    int per_row = 0;
    float* beta = nullptr;
};
struct VqaQueryRows {
    const void* rows = nullptr;  // [valid][d] row-major, device
    int32_t src_dtype = VQA_F16;  // VQA_F32 | VQA_F16
    int32_t valid = 0, d = 0;
    float scale = 1.0f;
    void* stage = nullptr;     // TILED [256][d_pad] of the storage type
    void* rowmajor = nullptr;  // [256][d_pad] or nullptr
};
int vqa_launch_sketch_rows(const void* tiled, int32_t src_dtype, int64_t first, int64_t count, int32_t d_pad_src, int32_t d_pad8,
                           const float* tile_info, void* out8, float* row_scale, float* row_lo, float* row_norm,
                           bool rotate /* sketch T x instead of x (convert.hip: sketch_rotate); rows and queries alike */,
                           const float* mu /* the shard's centre [d_pad8] or nullptr */, bool center /* index rows: sketch x - mu */,
                           float* row_off /* query rows: q . mu per row, or nullptr */, hipStream_t stream,
                           const VqaQueryRows* qr = nullptr /* not null: the rows come from qr (first = 0, count = 256), `tiled` is unused */,
                           const SketchSplit* split = nullptr);
// w = T mu / ||T mu|| [d_pad8] (zeros for a zero centre)
int vqa_launch_center_dir(const float* mu, int32_t d_pad8, bool rotate, float* wdir, hipStream_t stream);
// mean of the `count` rows first, first + stride, first + 2 stride, ... of a TILED array -> mu [d_pad8]
int vqa_launch_row_mean(const void* tiled, int32_t src_dtype, int64_t first, int64_t count, int64_t stride, int32_t d_pad_src, int32_t d_pad8,
                        float* mu, hipStream_t stream, float* msq = nullptr /* [d_pad8] mean of the squares, element by element */);
// scale (= max |x| / 127) of tiles [tile0, tile0 + ntiles) of a TILED fp16 / fp32 array into tile_info; clears their two maxima
int vqa_launch_tile_scales(const void* tiled, int32_t src_dtype, int64_t tile0, int64_t ntiles, int32_t d_pad_src, int32_t d_pad8,
                           float* tile_info, bool rotate, const float* mu, hipStream_t stream, float* tile_c = nullptr /* cleared too */,
                           const float* proj_w = nullptr /* per-row form: the scale of y - (w . y) w */);
// sketch search: per-query constants of the scan + reset of the candidate counters; exact scores of the candidate pairs
int vqa_launch_sketch_qconst(const float* thr, const float* qscale, const float* qlo, const float* qnorm, int32_t d, float* qconst,
                             unsigned* cand_cnt, int* overflow /* [4]: this tile's flag, OR over the call's earlier tiles, seq, pairs scored exactly */,
                             int clear /* 0: keep; 1: reset the candidate counters and the tile's overflow flag (OR-ed into overflow[1]
                                          first); 2: the same for the first query tile of a call (overflow[1] = 0) */,
                             int seq /* written to overflow[2] when clearing: the call these flags belong to */,
                             bool rotated /* the sketch is of rotated rows: the rotation's rounding joins the margin */,
                             const float* qoff /* q . mu per query or nullptr */, float mu_margin /* 3e-7 ||mu|| */, hipStream_t stream,
                             const float* qalpha = nullptr, const float* qrnorm = nullptr /* the split slack term or nullptr: 0, ||q|| */,
                             bool per_row = false);
int vqa_launch_rescore(const unsigned long long* regions, const unsigned* counts, int cap, int nregions, const long long* stage_pos,
                       int nq, int k, const void* x, const void* x_rowmajor /* or nullptr: the tiled rows x are read */, const void* q,
                       const void* q_rowmajor /* the staged query tile, row-major (read with x_rowmajor) */, int32_t dtype, int32_t d_pad,
                       vqa_key* cand_keys, unsigned* cand_cnt, int capq, int* overflow, hipStream_t stream);
// the exact scan's kin best rows per query (positions, -1 = none) scored again by rescore_kernel's fma chain and re-ranked: kout results
// (scores, external ids, positions) -- the bits the sketch path returns for the same rows (sketch.hip)
int vqa_launch_final_rescore(const int64_t* pos_in, int in_stride, int nq, int kin, int kout, const void* x, const void* x_rowmajor,
                             const void* q, const void* q_rowmajor, int32_t dtype, int32_t d_pad, const int64_t* ids, int64_t id_base,
                             float* out_scores, int64_t* out_ids, int64_t* out_pos, int out_stride, const int* gate, hipStream_t stream);
// rows [first, first + count) of a tiled shard -> its row-major copy (rows of row_bytes = padded row length in bytes)
int vqa_launch_rows_to_rowmajor(const void* tiled, int64_t first, int64_t count, int32_t row_bytes, void* out, hipStream_t stream);

// K4 (tiny_search.hip): the whole search of a small fp16 / fp32 shard for <= 16 questions and k <= 32 (questions x k <= 64) in ONE launch -- the latency form of
// vqa_index_search_host.  `workspace`: vqa_tiny_search_workspace_bytes() of device memory, zeroed once; results go to out_* (device
// pointers, here: the handle's mapped pinned buffer).  Same bits as the general path (tests/test_gpu_embeddings.py).
bool vqa_tiny_search_applies(int32_t dtype, int64_t n, int32_t d_pad, int32_t B, int32_t k);
size_t vqa_tiny_search_workspace_bytes();
int vqa_launch_tiny_search(const void* rows_tiled, int32_t dtype, int64_t n, int32_t d, int32_t d_pad, const void* q, const void* q_host_or_null,
                           void* q_stage, int32_t q_dtype, int32_t normalize, int32_t nq, int32_t k, const int64_t* ids, int64_t id_base,
                           void* workspace, float* out_scores, int64_t* out_ids, int64_t* out_pos, hipStream_t stream);
