// C ABI glue: index handles, search orchestration (stage queries -> K1 pass(es) -> K2 merge), error plumbing.
// Entry points are declared in include/vqa_retrieval.h (each cites the reference interface it replaces).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <thread>
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

// ---- THE PLAN RULES: every constant that decides which launches a search consists of, in one place, with the measurement that
// fitted it (all on MI355X, i.i.d. unit rows unless said otherwise).  vqa_index_options_init copies them into the caller-visible
// options; plan_launch / sketch_active / profit_pairs below apply them.  tests/test_gpu_plan.py pins what they give for the five
// BASELINE per-GPU shard shapes.
constexpr int kSketchMaxK = 128;  // the sketch search serves k <= 128 (txtai's hybrid search asks the dense index for 10 x limit rows)
struct PlanRules {
    int seed_mult = 2;             // seed pass = 2 x CUs tiles: 512 tiles halve the main pass's appends against 256 (round 1, net gain) ...
    int seed_div = 16;             // ... capped at 1/16 of the shard (1M x 768 fp32: 13 % of the search in the seed pass without it, 3.37 -> 3.19 ms)
    int seed_floor_tiles = 24;     // ... never fewer than 24 tiles (20 000-row shard with 4 seed tiles: thresholds stay -inf, 0.39 ms of flooding)
    int stage_min_plain = 24;      // two stages from 24 tiles per CU on (1.6M rows; crossover between 1M rows +1 % and 2M rows -2 %: profiles/r02_sweeps.txt)
    int stage_min_f16_sketch = 16; // fp16 + sketch: the cascade pays from 16 tiles per CU (1.05M rows; 1.5M rows 0.56 vs 0.65 ms, 1M level, 0.5M 0.30 vs 0.27)
    int stage_min_f32_sketch = 8;  // fp32 + sketch: the exact scan runs at 1/16 of the fp16 matrix rate, the int8 scan does not: from 524k rows
    int stage_pct = 10;            // first stage = 10 % of the tiles (6-14 % level at 10M rows: 2.00-2.03 ms; scripts/trace_steps_env.sh, r04_stage_size_trace.txt)
    int mid_k = 16;                // a third cascade level from k = 16 on ...
    int mid_min_tiles = 128;       // ... or from 128 tiles per workgroup on at any k (10M x 768, k = 10, interleaved A/B: 1.936 -> 1.888 ms; 3M rows: +4 %)
    int mid_pct = 200;             // ... of twice the first stage's tiles (profiles/r04_cascade_levels_ab.txt)
    int pre_k = 48;                // a fourth level (the first stage's leading quarter) from k = 48 on (r04_cascade_levels_ab.txt: k = 100: 3.9 -> 3.2 ms)
    int sketch_max_k = kSketchMaxK;  // (candidates grow linearly with k: 2.0 -> 2.8 ms from k = 10 to 128 at 10M rows, r05_k_and_batch.txt; beyond: the exact forms)
    double profit_f16 = 0.75;      // pause when a query tile scores more than 0.75 n - 4e5 pairs exactly (profiles/r04_profit_probe.txt: 5 shapes x 6 k)
    double profit_f32 = 4.0;       // fp32 shards: 4 n (their exact scan is 16x slower)
    double profit_offset_f16 = 4e5;
    double per_row_ratio = 0.6;    // per-row form when ||mu||^2 >= 0.6 mean ||x||^2 (level with the centre split at 0.5, ahead from 0.7: r04_per_row_threshold.txt)
    int per_row_min_ksteps = 6;    // ... and rows of >= 6 K-steps of 64 bytes
    int per_row_min_sample = 4096; // ... decided from at least min(n, 4096) rows of the first fill (a 1-row first fill always looked collapsed)
    int cooldown = 64;             // searches without the sketch after an overflow, doubling up to 64 x (r04_fallback_scan.txt: none of 576 shapes overflow unclustered)
};
static const PlanRules kPlan;

struct vqa_index {
    int device = 0;
    int64_t n = 0;
    int32_t d = 0, d_pad = 0, dtype = 0;
    float scale = 1.0f;      // stored value = scale * given value (power of two; 16 for fp8 so unit-vector components leave
                             // the e4m3 subnormal range); queries are scaled alike and scores are multiplied by 1/scale^2
    void* rows = nullptr;    // TILED layout: ceil(n/256) tiles x (d_pad/32) blocks of 16 KiB (convert.hip)
    size_t rows_bytes = 0;
    int64_t* ids = nullptr;  // [n] or null
    int64_t id_base = 0;
    int num_cu = 0;
    int max_grid = 0;
    bool two_pass = true;
    int seed_mult = 2;  // seed pass covers seed_mult * CUs tiles (vqa_index_options.seed_mult = 1..4)
    int seed_div = 16;  // ... but at most 1 / seed_div of the shard's tiles (options.seed_div, 0 = no cap)
    int stage_min_tiles = 24;  // two-stage search when the shard has at least this many tiles per workgroup (1.6M rows: measured
                               // crossover between 1M rows, +1 % step time, and 2M rows, -2 %; options.stage_min_tiles: dev /
                               // test override, 0 disables); the first stage takes stage_pct % of the tiles (options.stage_pct)
    int stage_pct = 10;
    int f16_loop = 0;  // fp16 K loop: 0 = anti-phase slots (two barriers per K-step), 1 = K-step pairs on the stagger loop (options.f16_loop)
    // workspace (allocated once; search never allocates)
    void* q_stage = nullptr;     // one 256-row tile in TILED layout
    void* q_rm = nullptr;        // ... and row-major (sketch shards with the re-scoring copy: what rescore_kernel reads)
    void* q_rows = nullptr;      // device staging for get_rows' host path (lazy)
    size_t q_rows_bytes = 0;
    // host rows -> shard (set_rows with a host pointer): two pinned + two device staging buffers and a copy stream (lazy), so
    // that the caller's pageable rows travel CPU copy -> DMA -> transpose kernel with chunk i + 1's CPU copy under chunk i's DMA
    void* up_pinned[2] = {nullptr, nullptr};
    void* up_dev[2] = {nullptr, nullptr};
    hipEvent_t up_done[2] = {nullptr, nullptr};
    hipStream_t up_stream = nullptr;
    size_t up_bytes = 0;
    vqa_key* partial = nullptr;  // [max_grid, 256, max(max_k, seeds per query)]: seed pass output, then main pass lists
    float* thr0 = nullptr;       // [256]
    float* thr_seed = nullptr;   // [256] the cascade's theta0 (k-th largest exact seed, MFMA arithmetic): kept for its exact fallback
    vqa_key* upper = nullptr;    // [256] last key returned per query (continuation passes of a search with k > 12)
    int* wide_flag = nullptr;    // 1 = the one-pass large-k result could not be verified: the gated continuation passes run
    // the exact paths' final re-scoring on large fp16 / fp32 shards (sketch.hip final_rescore_kernel): where the merge of an exact scan leaves its
    // rows before they are scored by the re-scoring arithmetic and re-ranked into the caller's arrays
    float* fin_scores = nullptr;   // [256][VQA_MAX_K_TOTAL]
    int64_t* fin_ids = nullptr;    // [256][VQA_MAX_K_TOTAL]
    int64_t* fin_pos = nullptr;    // [256][VQA_MAX_K_TOTAL]
    bool final_fma = true;         // options.final_rescore
    int fin_min_tiles = 0;         // ... from this many tiles per workgroup on (0: never)
    bool wide = true;            // options.wide_k = 0 disables the one-pass large-k attempt
    // int8 sketch of a large fp16 shard (VQA_INDEX_SKETCH): the rigorous pruning pre-pass of the main launch (score_topk.hip MODE 2)
    bool sketch = false;
    int32_t d_pad8 = 0;             // sketch row length (multiple of 128 elements)
    void* rows8 = nullptr;          // TILED int8: ceil(n/256) tiles x (d_pad8/64) blocks of 16 KiB
    size_t rows8_bytes = 0;
    void* rows_rm = nullptr;     // VQA_INDEX_RESCORE_ROWS: row-major copy of the stored rows (rows_bytes), read by the sketch search's re-scoring
    float* tile_info = nullptr;     // [tiles][4]: max ||x_hi||, max ||x_lo||, 1 / scale, scale of every 256-row tile (x_int = rint(x / scale))
    float* tile_c = nullptr;        // [tiles] (behind tile_info, same allocation): max |w . x_lo| of every tile -- the split slack term
    void* q8_stage = nullptr;       // sketch of the staged query tile
    float* qrow = nullptr;          // [5][256] per query: scale, ||q_lo||, ||q||, |alpha| = |z . w|, ||z - alpha w||
    float* qconst = nullptr;        // [kSketchQRows][256] the scan's per-query constants
    unsigned long long* regions = nullptr;  // [max_grid][kSketchCap] candidate pairs per workgroup of the scan
    unsigned* region_cnt = nullptr;         // [max_grid]
    vqa_key* cand_keys = nullptr;           // [256][kSketchCap] exact (score, position) keys per query
    unsigned* cand_cnt = nullptr;           // [256][kSketchSubLists] x kSketchCntStride: keys in every sub-list of a query's list (a line per counter)
    int* sketch_flag = nullptr;             // [5]: [0] 1 = a candidate buffer filled up in this query tile: its exact fallback scan runs;
                                            // [1] = OR of [0] over the EARLIER query tiles of the call, [2] = the call's number (sketch_qconst_kernel),
                                            // [3] = pairs scored exactly for this query tile (rescore_kernel adds its regions' counts), [4] = the
                                            // most any earlier tile of the call scored
    double profit_ratio = 0.75;             // A query tile that scores more pairs exactly than profit_pairs() costs more than the exact scan the sketch
                                            // search replaces: the handle pauses the sketch as after an overflow.  Fitted to 5 shard shapes x 6 k
                                            // (profiles/r04_profit_probe.txt, the set measured with the radix selections): the sketch search is level
                                            // with the exact forms over a wide band and loses clearly where pairs > 0.75 n - 4e5; the f32 MFMA scan of
                                            // fp32 shards is 16x slower: 4 n.  options.sketch_profit sets the factor, 0: never
    long long* stage_pos = nullptr;         // [256][max_k] row positions of the first stage's top-k
    float* mu = nullptr;                    // [d_pad8] centre of the shard (mean of the rows of its first fill), subtracted before the sketch
    float* wdir = nullptr;                  // [d_pad8] (behind mu, same allocation) w = T mu / ||T mu||: the slack term |z . x_lo| of the bound is split
                                            // along it (convert.hip sketch_rows_kernel); options.sketch_split = 0: not (A-B switch)
    bool split = true;
    float* beta = nullptr;                  // [tiles * 256] per-row form: beta = w . y of every row (the scan adds alpha beta per (query, row))
    bool per_row = false;                   // decided with the centre, at the first fill: ||mu||^2 >= 0.6 x the sample's mean ||x||^2 (rows collapsed onto one
                                            // direction; unit rows: a mean cosine of 0.6 between two rows; measured: the per-row form is level with the centre split at 0.5 and ahead from 0.7 on,
                                            // profiles/r04_per_row_threshold.txt) and rows of >= 6 K-steps; options.sketch_per_row = 0 / 1 forces it
    int per_row_env = -1;
    float* qoff = nullptr;                  // [256] q . mu of the query tile
    float mu_norm = 0.f;
    bool mu_set = false;
    bool center = true;                     // options.sketch_center = 0: no centring (needs the rotated form)
    bool rotate = true;                     // the sketch is cut from rotated rows (convert.hip: sketch_rotate); options.sketch_rotate = 0: from the rows as they are
    bool sketch_regq = true;                // options.sketch_regq: the register-resident-query scan kernel where it applies
    bool sketch_sx5 = true;                 // the sketch scan's X ring: five stages; options.sketch_ring_stages = 6: six (A-B switch: measured equal)
    int mid_k = 16, mid_pct = 200;          // a second cascade stage of mid_pct % of the first one's tiles for k >= mid_k (options.sketch_mid_k, 0: never; options.sketch_mid_pct)
    int pre_k = 48;                         // a leading quarter of the first stage as a stage of its own for k >= pre_k (options.sketch_pre_k, 0: never)
    int mid_min_tiles = 128;                // ... and for any k on shards of that many tiles per workgroup (options.sketch_mid_min_tiles, 0: by k only)
    bool cascade = true;                    // options.sketch_cascade = 0: the exact first stage of the narrow sketch form (A-B switch)
    int* sketch_flag_dev_mirror = nullptr;  // device address of the pinned mirror below (mapped host memory: the cascade's last merge writes it)
    int* sketch_flag_host = nullptr;        // pinned mirror of sketch_flag [3], copied once behind the last query tile of a call (read by LATER calls)
    int sketch_cooldown = 0;                // searches left that skip the sketch: data the bound cannot prune would pay the sketch scan
                                            // AND the exact fallback every time (options.sketch_cooldown searches, default 64, then it tries again;
                                            // every overflow in a row doubles the pause, up to 64 x the base: data the bound never prunes
                                            // ends up paying one wasted sketch scan per 4096 searches)
    int sketch_cooldown_len = 64;           // base length
    int sketch_cooldown_cur = 64;           // length of the next pause
    int sketch_seq = 0;                     // calls of this handle that ran the sketch search; the device writes the call's number beside its
    int sketch_seq_seen = 0;                // flags (sketch_flag[2]), so the host reacts ONCE to every completed call, however far it runs ahead
    int sketch_seq_ignore = 0;              // calls up to this number were queued before the current pause began
    int k_of_seq[64] = {0};                 // k of the sketch call with number seq (seq & 63): what a report is about
    int pause_min_k = 0;                    // the current pause applies to searches with k >= this (0: to all -- an overflow; a pause started by the
                                            // profitability rule at k leaves searches of less than half that k on the sketch)
    // opt-in kernel timing (bench.py): event pairs around the main scoring kernel
    bool timing = false;
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t ev_used = 0;
    // vqa_index_search_host: pinned, device-mapped [queries | scores | ids | positions] and the device copy of the normalised queries (lazy, grow-only)
    void* hio = nullptr;
    void* hio_dev = nullptr;
    size_t hio_bytes = 0;
    void* hq_norm = nullptr;
    size_t hq_norm_bytes = 0;
    bool one_launch = true;   // options.one_launch: small fp16 shards answer vqa_index_search_host with ONE kernel (tiny_search.hip)
    void* tiny_ws = nullptr;  // its per-workgroup lists + ticket (lazy)
    std::atomic_flag busy = ATOMIC_FLAG_INIT;  // one search at a time per handle (the workspace is shared)
};

constexpr int kSketchCap = 32768;  // candidate pairs per workgroup region / keys per query list (10M rows: ~1400 per query at k = 10,
                                   // ~13 000 at k = 100; the last selection drops the keys below theta1 before it sorts: merge_topk.hip)

struct HandleBusy {  // a second concurrent call on one handle is refused instead of corrupting the shared workspace
    std::atomic_flag& f;
    bool ok;
    explicit HandleBusy(std::atomic_flag& flag) : f(flag), ok(!flag.test_and_set(std::memory_order_acquire)) {}
    ~HandleBusy() {
        if (ok) f.clear(std::memory_order_release);
    }
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
    if (ix->rows) (void)hipFree(ix->rows);
    if (ix->ids) (void)hipFree(ix->ids);
    if (ix->q_stage) (void)hipFree(ix->q_stage);
    if (ix->q_rm) (void)hipFree(ix->q_rm);
    if (ix->q_rows) (void)hipFree(ix->q_rows);
    if (ix->hio) (void)hipHostFree(ix->hio);
    if (ix->hq_norm) (void)hipFree(ix->hq_norm);
    if (ix->tiny_ws) (void)hipFree(ix->tiny_ws);
    for (int b = 0; b < 2; ++b) {
        if (ix->up_pinned[b]) (void)hipHostFree(ix->up_pinned[b]);
        if (ix->up_dev[b]) (void)hipFree(ix->up_dev[b]);
        if (ix->up_done[b]) (void)hipEventDestroy(ix->up_done[b]);
    }
    if (ix->up_stream) (void)hipStreamDestroy(ix->up_stream);
    for (void* p : {(void*)ix->rows8, (void*)ix->tile_info, ix->q8_stage, (void*)ix->qrow, (void*)ix->qconst,
                    (void*)ix->regions, (void*)ix->region_cnt, (void*)ix->cand_keys, (void*)ix->cand_cnt, (void*)ix->sketch_flag,
                    (void*)ix->stage_pos})
        if (p) (void)hipFree(p);
    if (ix->rows_rm) (void)hipFree(ix->rows_rm);
    if (ix->mu) (void)hipFree(ix->mu);
    if (ix->beta) (void)hipFree(ix->beta);
    if (ix->qoff) (void)hipFree(ix->qoff);
    if (ix->sketch_flag_host) (void)hipHostFree(ix->sketch_flag_host);
    if (ix->partial) (void)hipFree(ix->partial);
    if (ix->thr0) (void)hipFree(ix->thr0);
    if (ix->thr_seed) (void)hipFree(ix->thr_seed);
    if (ix->upper) (void)hipFree(ix->upper);
    if (ix->wide_flag) (void)hipFree(ix->wide_flag);
    if (ix->fin_scores) (void)hipFree(ix->fin_scores);
    if (ix->fin_ids) (void)hipFree(ix->fin_ids);
    if (ix->fin_pos) (void)hipFree(ix->fin_pos);
    for (hipEvent_t e : ix->ev) (void)hipEventDestroy(e);
    delete ix;
}


// ---- host rows -> shard.  What Embeddings.load() does in Python (pinned double buffering, a few copier threads) for any
// C-ABI caller: the caller's (pageable) rows are copied by kUpThreads host threads into one of two PINNED staging buffers,
// travel to the device by DMA on a private stream and are transposed into the tiled layout by the kernel queued behind the
// copy, while the host threads already fill the other buffer.  (hipMemcpy from pageable memory stages through the runtime's
// own small bounce buffers and blocks: 3-5 GB/s; this path is bound by the host copy into the pinned buffer.)
constexpr size_t kUpChunkBytes = 32u << 20;
constexpr int kUpThreads = 4;

static void parallel_copy(void* dst, const void* src, size_t bytes) {
    if (bytes < (4u << 20)) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t cut = ((bytes + kUpThreads - 1) / kUpThreads + 4095) / 4096 * 4096;
    std::thread workers[kUpThreads];
    int started = 0;
    for (size_t off = cut; off < bytes && started < kUpThreads - 1; off += cut) {
        const size_t len = std::min(cut, bytes - off);
        workers[started++] = std::thread([=] { memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
    }
    memcpy(dst, src, std::min(cut, bytes));
    for (int i = 0; i < started; ++i) workers[i].join();
}

static int upload_host_rows(vqa_index* ix, int64_t first, int64_t count, const void* rows, int32_t src_dtype) {
    const size_t row_bytes = (size_t)ix->d * (src_dtype == VQA_F32 ? 4 : 2);
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(kUpChunkBytes / row_bytes));
    const size_t need = (size_t)std::min(chunk_rows, count) * row_bytes;
    if (ix->up_bytes < need) {
        for (int b = 0; b < 2; ++b) {
            if (ix->up_pinned[b]) (void)hipHostFree(ix->up_pinned[b]);
            if (ix->up_dev[b]) (void)hipFree(ix->up_dev[b]);
            ix->up_pinned[b] = ix->up_dev[b] = nullptr;
        }
        ix->up_bytes = 0;
        for (int b = 0; b < 2; ++b) {
            if (hipHostMalloc(&ix->up_pinned[b], need, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(&ix->up_dev[b], need) != hipSuccess) {
                (void)hipGetLastError();
                vqa_set_error("vqa_index_set_rows: allocating 2 x %zu staging bytes (pinned + device) failed", need);
                return VQA_ENOMEM;
            }
        }
        ix->up_bytes = need;
    }
    if (!ix->up_stream) VQA_HIP_CHECK(hipStreamCreateWithFlags(&ix->up_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b)
        if (!ix->up_done[b]) VQA_HIP_CHECK(hipEventCreateWithFlags(&ix->up_done[b], hipEventDisableTiming));
    VQA_HIP_CHECK(hipStreamSynchronize(nullptr));  // earlier fills of this shard (device-pointer calls run on the null stream)
    bool used[2] = {false, false};
    int i = 0;
    for (int64_t c0 = 0; c0 < count; c0 += chunk_rows, ++i) {
        const int b = i & 1;
        const int64_t c = std::min(chunk_rows, count - c0);
        if (used[b]) VQA_HIP_CHECK(hipEventSynchronize(ix->up_done[b]));  // chunk i - 2 has left this buffer pair
        parallel_copy(ix->up_pinned[b], static_cast<const char*>(rows) + (size_t)c0 * row_bytes, (size_t)c * row_bytes);
        VQA_HIP_CHECK(hipMemcpyAsync(ix->up_dev[b], ix->up_pinned[b], (size_t)c * row_bytes, hipMemcpyHostToDevice, ix->up_stream));
        int rc = vqa_launch_tile_rows(ix->up_dev[b], src_dtype, first + c0, c, c, ix->d, ix->d_pad, ix->dtype, ix->scale, ix->rows,
                                      ix->up_stream);
        if (rc != VQA_OK) return rc;
        VQA_HIP_CHECK(hipEventRecord(ix->up_done[b], ix->up_stream));
        used[b] = true;
    }
    VQA_HIP_CHECK(hipStreamSynchronize(ix->up_stream));
    return VQA_OK;
}

extern "C" int vqa_index_set_rows(vqa_index* ix, int64_t first, int64_t count, const void* rows, int32_t src_dtype,
                                  const int64_t* ids_or_null) {
    VQA_REQUIRE(ix, "vqa_index_set_rows: index is null");
    VQA_REQUIRE(first >= 0 && count >= 0 && first + count <= ix->n, "vqa_index_set_rows: rows [%lld, %lld) outside [0, %lld)",
                (long long)first, (long long)(first + count), (long long)ix->n);
    VQA_REQUIRE(src_dtype == VQA_F32 || src_dtype == VQA_F16, "vqa_index_set_rows: source element type %d is not f32/f16",
                src_dtype);
    VQA_REQUIRE((ids_or_null != nullptr) == (ix->ids != nullptr) || count == 0,
                "vqa_index_set_rows: the index was created %s an id vector", ix->ids ? "with" : "without");
    if (count == 0) return VQA_OK;
    VQA_REQUIRE(rows, "vqa_index_set_rows: rows is null");
    // the rows, the sketch's codes and tile_info are what a search of this handle reads: one call at a time per handle (a search
    // still in flight on a stream is the caller's to order: this call runs on the null stream and synchronises)
    HandleBusy busy(ix->busy);
    VQA_REQUIRE(busy.ok, "vqa_index_set_rows: this index handle is in use by another host thread (one call at a time per handle)");
    DeviceGuard guard(ix->device);
    hipPointerAttribute_t attr;
    bool on_device = hipPointerGetAttributes(&attr, rows) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // an unregistered host pointer reports an error: not ours
    if (on_device) {
        int rc = vqa_launch_tile_rows(rows, src_dtype, first, count, count, ix->d, ix->d_pad, ix->dtype, ix->scale, ix->rows, nullptr);
        if (rc != VQA_OK) return rc;
    } else {
        int rc = upload_host_rows(ix, first, count, rows, src_dtype);
        if (rc != VQA_OK) return rc;
    }
    if (ix->sketch) {
        // the sketch of every tile these rows touch, from the stored fp16 values: the tile's scale (max |x| / 127 over its 256
        // rows), then its rows' int8 codes and the two maxima the pruning bound needs -- whole tiles, so a tile filled by
        // several calls always ends up consistent
        VQA_HIP_CHECK(hipStreamSynchronize(nullptr));
        const int64_t t0 = first >> 8, t1 = (first + count - 1) >> 8;
        if (ix->center && !ix->mu_set) {
            // the shard's centre: the mean of (up to 65 536 of) the rows of its FIRST fill -- a strided sample over the whole fill, so
            // that a corpus stored sorted by topic still gets its overall mean --, fixed from then on: every tile's sketch must be
            // cut against the same centre; any centre is valid (q . x = q . mu + q . (x - mu)), a good one shortens the rows
            const int64_t samples = std::min<int64_t>(count, 65536);
            // (the mean of the squares lands where w goes afterwards: mu and w share an allocation)
            int rcm = vqa_launch_row_mean(ix->rows, ix->dtype, first, samples, count / samples, ix->d_pad, ix->d_pad8, ix->mu, nullptr, ix->wdir);
            if (rcm != VQA_OK) return rcm;
            std::vector<float> h((size_t)ix->d_pad8 * (ix->wdir ? 2 : 1));
            VQA_HIP_CHECK(hipMemcpy(h.data(), ix->mu, h.size() * 4, hipMemcpyDeviceToHost));
            double n2 = 0.0, row2 = 0.0;  // ||mu||^2 and the sample's mean ||x||^2
            for (size_t j = 0; j < (size_t)ix->d_pad8; ++j) n2 += (double)h[j] * h[j];
            for (size_t j = (size_t)ix->d_pad8; j < h.size(); ++j) row2 += (double)h[j];
            ix->mu_norm = (float)(std::sqrt(n2) * (1.0 + 1e-6));
            ix->mu_set = true;
            if (ix->wdir) {
                rcm = vqa_launch_center_dir(ix->mu, ix->d_pad8, ix->rotate, ix->wdir, nullptr);
                if (rcm != VQA_OK) return rcm;
                // rows collapsed onto the centre direction (an untrained / anisotropic encoder): the per-row form (convert.hip sketch_rows_kernel)
                // (the per-row form is irreversible: a first fill of a few rows -- a producer feeding row by row -- says nothing about the
                // shard, and one row alone has ||mu||^2 == ||x||^2: at least min(n, 4096) sampled rows, or the centre split stays)
                const bool enough = samples >= std::min<int64_t>(ix->n, kPlan.per_row_min_sample);
                ix->per_row = ix->beta && (ix->per_row_env == 1 || (ix->per_row_env < 0 && enough && row2 > 0.0 && n2 >= kPlan.per_row_ratio * row2));
            }
        }
        int rc = vqa_launch_tile_scales(ix->rows, ix->dtype, t0, t1 - t0 + 1, ix->d_pad, ix->d_pad8, ix->tile_info, ix->rotate, ix->center ? ix->mu : nullptr, nullptr,
                                        ix->tile_c, ix->per_row ? ix->wdir : nullptr);
        if (rc != VQA_OK) return rc;
        const int64_t r0 = t0 * 256, r1 = std::min<int64_t>(ix->n, (t1 + 1) * 256);
        SketchSplit sp;
        sp.wdir = ix->wdir;
        sp.tile_c = ix->tile_c;
        sp.per_row = ix->per_row ? 1 : 0;
        sp.beta = ix->per_row ? ix->beta : nullptr;
        rc = vqa_launch_sketch_rows(ix->rows, ix->dtype, r0, r1 - r0, ix->d_pad, ix->d_pad8, ix->tile_info, ix->rows8, nullptr, nullptr,
                                    nullptr, ix->rotate, ix->center ? ix->mu : nullptr, true, nullptr, nullptr, nullptr, ix->wdir ? &sp : nullptr);
        if (rc != VQA_OK) return rc;
        if (ix->rows_rm) {  // the same stored values, row-major
            rc = vqa_launch_rows_to_rowmajor(ix->rows, first, count, ix->d_pad * elem_bytes(ix->dtype), ix->rows_rm, nullptr);
            if (rc != VQA_OK) return rc;
        }
    }
    if (ids_or_null) VQA_HIP_CHECK(hipMemcpy(ix->ids + first, ids_or_null, (size_t)count * 8, hipMemcpyDefault));
    VQA_HIP_CHECK(hipStreamSynchronize(nullptr));
    return VQA_OK;
}

extern "C" void vqa_index_options_init(vqa_index_options* o) {
    if (!o) return;
    o->struct_size = (uint32_t)sizeof(vqa_index_options);
    o->flags = 0;
    o->two_pass = 1;
    o->wide_k = 1;
    o->seed_mult = kPlan.seed_mult;
    o->seed_div = kPlan.seed_div;
    o->stage_min_tiles = -1;
    o->stage_pct = kPlan.stage_pct;
    o->f16_loop = 0;
    o->sketch_cascade = o->sketch_rotate = o->sketch_center = o->sketch_split = 1;
    o->sketch_per_row = -1;
    o->sketch_ring_stages = 5;
    o->sketch_mid_k = kPlan.mid_k;
    o->sketch_mid_min_tiles = kPlan.mid_min_tiles;
    o->sketch_mid_pct = kPlan.mid_pct;
    o->sketch_pre_k = kPlan.pre_k;
    o->sketch_cooldown = kPlan.cooldown;
    o->sketch_profit = -1.0f;
    o->rescore_copy = -1;
    o->poison_workspace = -1;
    o->one_launch = 1;
    o->sketch_regq = 1;
    o->final_rescore = 1;
}

// -DVQA_DEV variant libraries only (scripts/: ab_loops.py, probes): the rounds-1-4 environment switches laid over the options
static void dev_env_overlay(vqa_index_options* o) {
    auto geti = [](const char* name, int32_t* dst) {
        if (const char* v = vqa_dev_env(name)) *dst = atoi(v);
    };
    geti("VQA_TWO_PASS", &o->two_pass);
    geti("VQA_WIDE_K", &o->wide_k);
    geti("VQA_SEED_MULT", &o->seed_mult);
    geti("VQA_SEED_DIV", &o->seed_div);
    geti("VQA_STAGE_MIN", &o->stage_min_tiles);
    geti("VQA_STAGE_PCT", &o->stage_pct);
    geti("VQA_F16_LOOP", &o->f16_loop);
    geti("VQA_SKETCH_CASCADE", &o->sketch_cascade);
    geti("VQA_SKETCH_ROTATE", &o->sketch_rotate);
    geti("VQA_SKETCH_CENTER", &o->sketch_center);
    geti("VQA_SKETCH_SPLIT", &o->sketch_split);
    geti("VQA_SKETCH_PER_ROW", &o->sketch_per_row);
    geti("VQA_SKETCH_SX", &o->sketch_ring_stages);
    geti("VQA_SKETCH_MID_K", &o->sketch_mid_k);
    geti("VQA_SKETCH_MID_MIN", &o->sketch_mid_min_tiles);
    geti("VQA_SKETCH_MID_PCT", &o->sketch_mid_pct);
    geti("VQA_SKETCH_PRE_K", &o->sketch_pre_k);
    geti("VQA_SKETCH_COOLDOWN", &o->sketch_cooldown);
    if (const char* v = vqa_dev_env("VQA_SKETCH_PROFIT")) o->sketch_profit = (float)atof(v);
    geti("VQA_RESCORE_COPY", &o->rescore_copy);
    geti("VQA_ONE_LAUNCH", &o->one_launch);
    geti("VQA_SKETCH_REGQ", &o->sketch_regq);
    geti("VQA_FINAL_RESCORE", &o->final_rescore);
    if (const char* v = vqa_dev_env("VQA_POISON_WORKSPACE")) o->poison_workspace = (int)strtol(v, nullptr, 0) & 0xFF;
    if (const char* v = vqa_dev_env("VQA_SKETCH")) {
        if (v[0] == '0') o->flags &= ~(uint32_t)VQA_INDEX_SKETCH;
    }
}

extern "C" int vqa_index_create(vqa_index** out, int device, int64_t n, int32_t d, int32_t dtype, const void* rows,
                                int32_t rows_dtype, const int64_t* ids_or_null, int64_t id_base, uint32_t flags) {
    vqa_index_options o;
    vqa_index_options_init(&o);
    o.flags = flags;
    return vqa_index_create_ex(out, device, n, d, dtype, rows, rows_dtype, ids_or_null, id_base, &o);
}

extern "C" int vqa_index_create_ex(vqa_index** out, int device, int64_t n, int32_t d, int32_t dtype, const void* rows,
                                   int32_t rows_dtype, const int64_t* ids_or_null, int64_t id_base, const vqa_index_options* opt) {
    VQA_REQUIRE(out, "vqa_index_create: out is null");
    *out = nullptr;
    vqa_index_options o;
    vqa_index_options_init(&o);
    if (opt) {
        VQA_REQUIRE(opt->struct_size >= 8 && opt->struct_size <= 4096, "vqa_index_create_ex: options.struct_size=%u (vqa_index_options_init sets it)",
                    opt->struct_size);
        memcpy(&o, opt, opt->struct_size < sizeof(o) ? opt->struct_size : sizeof(o));
        o.struct_size = (uint32_t)sizeof(o);
    }
    dev_env_overlay(&o);
    // rescore_copy = 0: no row-major copy whatever the flag says (rounds 1-4 meaning of VQA_RESCORE_COPY=0); 1: the flag + no free-memory rule
    const uint32_t flags = o.rescore_copy == 0 ? (o.flags & ~(uint32_t)VQA_INDEX_RESCORE_ROWS) : o.rescore_copy == 1 ? (o.flags | VQA_INDEX_RESCORE_ROWS) : o.flags;
    VQA_REQUIRE(o.seed_mult >= 1 && o.seed_mult <= 4, "vqa_index_create_ex: seed_mult=%d outside [1, 4]", o.seed_mult);
    VQA_REQUIRE(o.seed_div >= 0, "vqa_index_create_ex: seed_div=%d", o.seed_div);
    VQA_REQUIRE(o.stage_pct >= 1 && o.stage_pct <= 50, "vqa_index_create_ex: stage_pct=%d outside [1, 50]", o.stage_pct);
    VQA_REQUIRE(o.f16_loop >= 0 && o.f16_loop <= 2, "vqa_index_create_ex: f16_loop=%d", o.f16_loop);
    VQA_REQUIRE(o.sketch_ring_stages == 5 || o.sketch_ring_stages == 6, "vqa_index_create_ex: sketch_ring_stages=%d (5 or 6)", o.sketch_ring_stages);
    VQA_REQUIRE(o.sketch_mid_pct >= 1 && o.sketch_cooldown >= 0 && o.sketch_mid_k >= 0 && o.sketch_pre_k >= 0 && o.sketch_mid_min_tiles >= 0,
                "vqa_index_create_ex: a sketch_* option is negative");
    VQA_REQUIRE(o.poison_workspace >= -1 && o.poison_workspace <= 255, "vqa_index_create_ex: poison_workspace=%d", o.poison_workspace);
    VQA_REQUIRE(n >= 0 && n < 0xFFFFFFFFll, "vqa_index_create: n=%lld outside [0, 2^32-1) rows per shard", (long long)n);
    VQA_REQUIRE(d >= 1 && d <= 65536, "vqa_index_create: d=%d", d);
    VQA_REQUIRE(dtype == VQA_F32 || dtype == VQA_F16 || dtype == VQA_FP8_E4M3, "vqa_index_create: dtype %d", dtype);
    VQA_REQUIRE((flags & ~(uint32_t)(VQA_INDEX_HAS_IDS | VQA_INDEX_SKETCH | VQA_INDEX_RESCORE_ROWS)) == 0, "vqa_index_create: unknown flags 0x%x", flags);
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
    const int pad_to = 128 / elem_bytes(dtype);  // two K-steps of 64 bytes
    ix->d_pad = (d + pad_to - 1) / pad_to * pad_to;
    ix->dtype = dtype;
    ix->scale = dtype == VQA_FP8_E4M3 ? 16.0f : 1.0f;
    ix->id_base = id_base;
    ix->num_cu = prop.multiProcessorCount;
    ix->max_grid = ix->num_cu;
    ix->two_pass = o.two_pass != 0;
    ix->wide = o.wide_k != 0;
    ix->one_launch = o.one_launch != 0;
    ix->seed_div = o.seed_div;
    ix->seed_mult = o.seed_mult;
    const bool stage_given = o.stage_min_tiles >= 0;  // (tests, A/B runs: a plan forced at a size production would not pick it at)
    ix->stage_min_tiles = stage_given ? o.stage_min_tiles : kPlan.stage_min_plain;
    // shard size (tiles per workgroup) from which a shard of this type CAN have two answer paths -- where a sketch index of the type starts its
    // sketch search: from there on every exact path ends in the re-scoring arithmetic (final_rescore_applies), with or without a sketch
    ix->fin_min_tiles = stage_given ? o.stage_min_tiles : dtype == VQA_F32 ? kPlan.stage_min_f32_sketch : kPlan.stage_min_f16_sketch;
    ix->stage_pct = o.stage_pct;
    ix->f16_loop = o.f16_loop;
    const int eb = elem_bytes(dtype);
    int rc = VQA_OK;
    do {
        const int64_t tiles = (n + 255) / 256;
        ix->rows_bytes = (size_t)tiles * 256 * ix->d_pad * eb;
        if (n > 0) {
            if (hipMalloc(&ix->rows, ix->rows_bytes) != hipSuccess) {
                vqa_set_error("vqa_index_create: hipMalloc of %zu bytes for the rows failed", ix->rows_bytes);
                rc = VQA_ENOMEM;
                break;
            }
            // rows of the ragged last tile and the padded tail of every row must read as zeros
            if (hipMemset(ix->rows, 0, ix->rows_bytes) != hipSuccess) {
                vqa_set_error("vqa_index_create: clearing the rows failed");
                rc = VQA_EHIP;
                break;
            }
            if ((ids_or_null || (flags & VQA_INDEX_HAS_IDS)) && hipMalloc((void**)&ix->ids, (size_t)n * 8) != hipSuccess) {
                vqa_set_error("vqa_index_create: hipMalloc for %lld ids failed", (long long)n);
                rc = VQA_ENOMEM;
                break;
            }
            if (ix->ids && !ids_or_null && hipMemset(ix->ids, 0xFF, (size_t)n * 8) != hipSuccess) {
                vqa_set_error("vqa_index_create: clearing the ids failed");
                rc = VQA_EHIP;
                break;
            }
        }
        const int max_k = vqa_score_topk_max_k(dtype);
        const int list_len = max_k;  // main pass lists [max_grid][256][k <= max_k]; the seed pass needs 4 * max_grid * 256 * 2 keys at most
        if (hipMalloc(&ix->q_stage, (size_t)VQA_QUERY_TILE * ix->d_pad * eb) != hipSuccess ||
            hipMalloc((void**)&ix->partial, (size_t)4 * ix->max_grid * VQA_QUERY_TILE * list_len * sizeof(vqa_key)) != hipSuccess ||
            hipMalloc((void**)&ix->thr0, VQA_QUERY_TILE * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&ix->thr_seed, VQA_QUERY_TILE * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&ix->upper, VQA_QUERY_TILE * sizeof(vqa_key)) != hipSuccess ||
            hipMalloc((void**)&ix->wide_flag, sizeof(int)) != hipSuccess ||
            ((dtype == VQA_F16 || dtype == VQA_F32) &&
             (hipMalloc((void**)&ix->fin_scores, (size_t)VQA_QUERY_TILE * VQA_MAX_K_TOTAL * 4) != hipSuccess ||
              hipMalloc((void**)&ix->fin_ids, (size_t)VQA_QUERY_TILE * VQA_MAX_K_TOTAL * 8) != hipSuccess ||
              hipMalloc((void**)&ix->fin_pos, (size_t)VQA_QUERY_TILE * VQA_MAX_K_TOTAL * 8) != hipSuccess))) {
            vqa_set_error("vqa_index_create: workspace allocation failed");
            rc = VQA_ENOMEM;
            break;
        }
        ix->final_fma = o.final_rescore != 0;
        // int8 sketch: fp16 shards only, and only where a workgroup's tile maxima fit the scan's LDS
        // (and only for shards large enough for the two-stage search, whose main launch the sketch scan replaces)
        // fp32 shards: the exact scan runs at 1/16 of the fp16 matrix rate, the sketch scan at the same int8 rate: the
        // two-stage / sketch plan pays from 8 tiles per compute unit (524k rows) on
        if ((flags & VQA_INDEX_SKETCH) && dtype == VQA_F32 && !stage_given) ix->stage_min_tiles = kPlan.stage_min_f32_sketch;
        // fp16 shards: the cascade pays from 16 tiles per compute unit (1.05M rows) on -- 1.5M rows: 0.56 vs 0.65 ms, 1M: 0.48 either
        // way, 0.5M: 0.30 vs 0.27 (the exact two-stage plan of shards without a sketch keeps its 24)
        if ((flags & VQA_INDEX_SKETCH) && dtype == VQA_F16 && !stage_given) ix->stage_min_tiles = kPlan.stage_min_f16_sketch;
        ix->sketch = (flags & VQA_INDEX_SKETCH) && (dtype == VQA_F16 || dtype == VQA_F32) && n > 0 && ix->stage_min_tiles > 0 &&
                     tiles >= (int64_t)ix->stage_min_tiles * ix->max_grid &&
                     (tiles + ix->max_grid - 1) / ix->max_grid <= vqa_score_topk_sketch_max_tiles() && d <= 8192;
        if (ix->sketch) {
            ix->d_pad8 = (d + 127) / 128 * 128;
            ix->rows8_bytes = (size_t)tiles * 256 * ix->d_pad8;
            const size_t qc = VQA_QUERY_TILE * sizeof(float);
            if (hipMalloc(&ix->rows8, ix->rows8_bytes) != hipSuccess || hipMalloc((void**)&ix->tile_info, (size_t)tiles * 20) != hipSuccess ||
                hipMalloc(&ix->q8_stage, (size_t)VQA_QUERY_TILE * ix->d_pad8) != hipSuccess ||
                hipMalloc((void**)&ix->qrow, 5 * qc) != hipSuccess || hipMalloc((void**)&ix->qconst, kSketchQRows * qc) != hipSuccess ||
                hipMalloc((void**)&ix->regions, (size_t)ix->max_grid * kSketchCap * 8) != hipSuccess ||
                hipMalloc((void**)&ix->region_cnt, (size_t)ix->max_grid * 4) != hipSuccess ||
                hipMalloc((void**)&ix->cand_keys, (size_t)VQA_QUERY_TILE * kSketchCap * sizeof(vqa_key)) != hipSuccess ||
                hipMalloc((void**)&ix->cand_cnt, (size_t)VQA_QUERY_TILE * kSketchSubLists * kSketchCntStride * 4) != hipSuccess || hipMalloc((void**)&ix->sketch_flag, 5 * sizeof(int)) != hipSuccess ||
                hipMalloc((void**)&ix->stage_pos, (size_t)VQA_QUERY_TILE * max_k * 8) != hipSuccess ||
                hipHostMalloc((void**)&ix->sketch_flag_host, 5 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&ix->sketch_flag_dev_mirror, ix->sketch_flag_host, 0) != hipSuccess) {
                vqa_set_error("vqa_index_create: allocating the int8 sketch (%zu bytes) failed", ix->rows8_bytes);
                rc = VQA_ENOMEM;
                break;
            }
            ix->sketch_flag_host[0] = ix->sketch_flag_host[1] = ix->sketch_flag_host[2] = ix->sketch_flag_host[3] = ix->sketch_flag_host[4] = 0;
            // (a shard whose size rule was given explicitly -- tests, A/B runs -- is below the size at which the sketch pays at all:
            // the profitability rule is off there unless asked for)
            if (o.sketch_profit >= 0.f) ix->profit_ratio = o.sketch_profit;
            else ix->profit_ratio = stage_given ? 0.0 : dtype == VQA_F32 ? kPlan.profit_f32 : kPlan.profit_f16;
            ix->cascade = o.sketch_cascade != 0;
            ix->mid_k = o.sketch_mid_k;
            ix->mid_pct = o.sketch_mid_pct;
            ix->mid_min_tiles = o.sketch_mid_min_tiles;
            ix->pre_k = o.sketch_pre_k;
            ix->sketch_sx5 = o.sketch_ring_stages != 6;
            ix->sketch_regq = o.sketch_regq != 0;
            ix->rotate = o.sketch_rotate != 0;
            ix->center = o.sketch_center != 0;
            ix->center = ix->center && ix->rotate;
            ix->tile_c = ix->tile_info + (size_t)tiles * 4;
            if (ix->center && (hipMalloc((void**)&ix->mu, (size_t)ix->d_pad8 * 8) != hipSuccess ||
                               hipMalloc((void**)&ix->qoff, VQA_QUERY_TILE * 4) != hipSuccess ||
                               hipMemset(ix->mu, 0, (size_t)ix->d_pad8 * 8) != hipSuccess)) {
                vqa_set_error("vqa_index_create: allocating the sketch's centre failed");
                rc = VQA_ENOMEM;
                break;
            }
            ix->split = o.sketch_split != 0 && ix->center;
            if (ix->split) ix->wdir = ix->mu + ix->d_pad8;
            ix->per_row_env = o.sketch_per_row == 1 ? 1 : o.sketch_per_row == 0 ? 0 : -1;
            if (ix->split && ix->d_pad8 / 64 >= kPlan.per_row_min_ksteps && ix->per_row_env != 0 &&
                (hipMalloc((void**)&ix->beta, (size_t)tiles * 256 * 4) != hipSuccess || hipMemset(ix->beta, 0, (size_t)tiles * 256 * 4) != hipSuccess)) {
                vqa_set_error("vqa_index_create: allocating the per-row betas failed");
                rc = VQA_ENOMEM;
                break;
            }
            ix->sketch_cooldown_len = o.sketch_cooldown;
            ix->sketch_cooldown_cur = ix->sketch_cooldown_len;
            if (hipMemset(ix->rows8, 0, ix->rows8_bytes) != hipSuccess || hipMemset(ix->tile_info, 0, (size_t)tiles * 20) != hipSuccess ||
                hipMemset(ix->sketch_flag, 0, 5 * sizeof(int)) != hipSuccess ||
                hipMemset(ix->q8_stage, 0, (size_t)VQA_QUERY_TILE * ix->d_pad8) != hipSuccess) {
                vqa_set_error("vqa_index_create: clearing the int8 sketch failed");
                rc = VQA_EHIP;
                break;
            }
            if (flags & VQA_INDEX_RESCORE_ROWS) {  // the sketch search's re-scoring then reads whole rows instead of 64-byte pieces
                // the copy only speeds the re-scoring up: a device too full for it goes without (vqa_index_device_bytes tells) -- and
                // "too full" leaves the caller room: the copy is only taken when an eighth of the device's memory stays free behind it
                // (36 GB of 288: the encoder's workspace, torch's allocations, another shard's buffers)
                size_t mem_free = 0, mem_total = 0;
                if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) {
                    (void)hipGetLastError();
                    mem_free = mem_total = 0;
                }
                const bool room = o.rescore_copy == 1 || mem_total == 0 || mem_free >= ix->rows_bytes + mem_total / 8;
                if (!room || hipMalloc(&ix->rows_rm, ix->rows_bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    ix->rows_rm = nullptr;
                } else if (hipMemset(ix->rows_rm, 0, ix->rows_bytes) != hipSuccess) {
                    vqa_set_error("vqa_index_create: clearing the row-major copy failed");
                    rc = VQA_EHIP;
                    break;
                } else if (hipMalloc(&ix->q_rm, (size_t)VQA_QUERY_TILE * ix->d_pad * eb) != hipSuccess) {
                    vqa_set_error("vqa_index_create: allocating the row-major query tile failed");
                    rc = VQA_ENOMEM;
                    break;
                }
            }
        }
        if (o.poison_workspace >= 0) {
            // test hook: every workspace a search is expected to WRITE before it reads starts as a byte pattern instead of whatever the
            // allocator hands out (a long-lived process gets recycled, dirty memory; 0xCB... reads as a large negative float)
            const int byte = o.poison_workspace;
            const int max_k2 = vqa_score_topk_max_k(dtype);
            (void)hipMemset(ix->q_stage, byte, (size_t)VQA_QUERY_TILE * ix->d_pad * eb);
            (void)hipMemset(ix->partial, byte, (size_t)4 * ix->max_grid * VQA_QUERY_TILE * max_k2 * sizeof(vqa_key));
            (void)hipMemset(ix->thr0, byte, VQA_QUERY_TILE * sizeof(float));
            (void)hipMemset(ix->thr_seed, byte, VQA_QUERY_TILE * sizeof(float));
            (void)hipMemset(ix->upper, byte, VQA_QUERY_TILE * sizeof(vqa_key));
            if (ix->sketch) {
                (void)hipMemset(ix->qrow, byte, 5 * VQA_QUERY_TILE * sizeof(float));
                (void)hipMemset(ix->qconst, byte, kSketchQRows * VQA_QUERY_TILE * sizeof(float));
                (void)hipMemset(ix->regions, byte, (size_t)ix->max_grid * kSketchCap * 8);
                (void)hipMemset(ix->region_cnt, byte, (size_t)ix->max_grid * 4);
                (void)hipMemset(ix->cand_keys, byte, (size_t)VQA_QUERY_TILE * kSketchCap * sizeof(vqa_key));
                (void)hipMemset(ix->cand_cnt, byte, (size_t)VQA_QUERY_TILE * kSketchSubLists * kSketchCntStride * 4);
                (void)hipMemset(ix->stage_pos, byte, (size_t)VQA_QUERY_TILE * max_k2 * 8);
                if (ix->qoff) (void)hipMemset(ix->qoff, byte, VQA_QUERY_TILE * 4);
                if (ix->q_rm) (void)hipMemset(ix->q_rm, byte, (size_t)VQA_QUERY_TILE * ix->d_pad * eb);
            }
            (void)hipDeviceSynchronize();
        }
        if (rows && n > 0) rc = vqa_index_set_rows(ix, 0, n, rows, rows_dtype, ids_or_null);
    } while (0);
    if (rc != VQA_OK) {
        vqa_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return VQA_OK;
}

extern "C" int vqa_index_get_rows(vqa_index* ix, int64_t first, int64_t count, void* out_rows, int64_t* out_ids_or_null) {
    VQA_REQUIRE(ix, "vqa_index_get_rows: index is null");
    VQA_REQUIRE(first >= 0 && count >= 0 && first + count <= ix->n, "vqa_index_get_rows: rows [%lld, %lld) outside [0, %lld)",
                (long long)first, (long long)(first + count), (long long)ix->n);
    if (count == 0) return VQA_OK;
    VQA_REQUIRE(out_rows, "vqa_index_get_rows: out is null");
    DeviceGuard guard(ix->device);
    hipPointerAttribute_t attr;
    const bool on_device = hipPointerGetAttributes(&attr, out_rows) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (on_device) {
        int rc = vqa_launch_untile_rows(ix->rows, first, count, ix->d, ix->d_pad, ix->dtype, out_rows, nullptr);
        if (rc != VQA_OK) return rc;
    } else {
        const int64_t chunk_rows = std::max<int64_t>(1, (64ll << 20) / ix->d);
        const size_t eb = (size_t)elem_bytes(ix->dtype);
        const size_t need = (size_t)std::min(chunk_rows, count) * ix->d * eb;
        if (ix->q_rows_bytes < need) {
            if (ix->q_rows) (void)hipFree(ix->q_rows);
            ix->q_rows = nullptr;
            ix->q_rows_bytes = 0;
            if (hipMalloc(&ix->q_rows, need) != hipSuccess) {
                vqa_set_error("vqa_index_get_rows: hipMalloc of %zu staging bytes failed", need);
                return VQA_ENOMEM;
            }
            ix->q_rows_bytes = need;
        }
        for (int64_t c0 = 0; c0 < count; c0 += chunk_rows) {
            const int64_t c = std::min(chunk_rows, count - c0);
            int rc = vqa_launch_untile_rows(ix->rows, first + c0, c, ix->d, ix->d_pad, ix->dtype, ix->q_rows, nullptr);
            if (rc != VQA_OK) return rc;
            VQA_HIP_CHECK(hipMemcpy(reinterpret_cast<char*>(out_rows) + (size_t)c0 * ix->d * eb, ix->q_rows,
                                    (size_t)c * ix->d * eb, hipMemcpyDeviceToHost));
        }
    }
    if (out_ids_or_null) {
        VQA_REQUIRE(ix->ids, "vqa_index_get_rows: the index has no id vector (ids are id_base + position)");
        VQA_HIP_CHECK(hipMemcpy(out_ids_or_null, ix->ids + first, (size_t)count * 8, hipMemcpyDefault));
    }
    VQA_HIP_CHECK(hipStreamSynchronize(nullptr));
    return VQA_OK;
}

extern "C" int64_t vqa_index_size(const vqa_index* ix) { return ix ? ix->n : -1; }
extern "C" int32_t vqa_index_dim(const vqa_index* ix) { return ix ? ix->d : -1; }
extern "C" int32_t vqa_index_dtype(const vqa_index* ix) { return ix ? ix->dtype : -1; }
// the profitability rule of the sketch search (vqa_index::profit_ratio)
static double profit_pairs(const vqa_index* ix) {
    const double n = (double)ix->n;
    const double lim = ix->profit_ratio * n - (ix->dtype == VQA_F32 ? 0.0 : kPlan.profit_offset_f16);
    return lim > 0.1 * n ? lim : 0.1 * n;
}
static int pairs_reported(const vqa_index* ix) {  // the most pairs a query tile of the last reported call scored exactly
    const int a = __atomic_load_n(ix->sketch_flag_host + 3, __ATOMIC_RELAXED), b = __atomic_load_n(ix->sketch_flag_host + 4, __ATOMIC_RELAXED);
    return a > b ? a : b;  // (the kernel-written mirror holds the maximum in [3]; the copied flags hold [3] and [4])
}

extern "C" int32_t vqa_index_sketch_state(const vqa_index* ix) {
    if (!ix || !ix->sketch) return -1;
    // (the flag of the last sketch search arrives in the pinned mirror when that search has completed)
    // a completed call's report the host has not looked at yet: what the next search will do with it
    if (__atomic_load_n(ix->sketch_flag_host + 2, __ATOMIC_RELAXED) != ix->sketch_seq_seen &&
        ((__atomic_load_n(ix->sketch_flag_host, __ATOMIC_RELAXED) | __atomic_load_n(ix->sketch_flag_host + 1, __ATOMIC_RELAXED)) != 0 ||
         (ix->profit_ratio > 0.0 && (double)pairs_reported(ix) > profit_pairs(ix))))
        return ix->sketch_cooldown_cur > 0 ? ix->sketch_cooldown_cur : 1;
    return ix->sketch_cooldown;
}

extern "C" int vqa_index_sketch_pause(vqa_index* ix, int32_t searches) {
    VQA_REQUIRE(ix, "vqa_index_sketch_pause: index is null");
    VQA_REQUIRE(ix->sketch, "vqa_index_sketch_pause: the shard keeps no sketch");
    VQA_REQUIRE(searches >= 0, "vqa_index_sketch_pause: searches=%d", searches);
    HandleBusy busy(ix->busy);
    VQA_REQUIRE(busy.ok, "vqa_index_sketch_pause: this index handle is in use by another host thread");
    ix->sketch_cooldown = searches > 0 ? searches + 1 : 0;  // (a search decrements the counter BEFORE it looks at it)
    ix->pause_min_k = 0;                        // every k
    ix->sketch_seq_ignore = ix->sketch_seq;     // reports of searches already queued say nothing about this pause
    return VQA_OK;
}

extern "C" int64_t vqa_index_device_bytes(const vqa_index* ix) {
    if (!ix) return -1;
    int64_t b = (int64_t)ix->rows_bytes + (ix->ids ? ix->n * 8 : 0);
    if (ix->sketch) b += (int64_t)ix->rows8_bytes + ((ix->n + 255) / 256) * 16;
    if (ix->rows_rm) b += (int64_t)ix->rows_bytes;
    return b;
}

struct LaunchPlan {
    int tiles = 0;  // corpus tiles of 256 rows
    int seed_tiles = 0;  // tiles of the seed pass, 0 = no seeding
    int grid0 = 0;  // its workgroups (a few tiles each)
    int grid1 = 0;  // workgroups of the main pass
    int stage_tiles = 0;  // two-stage search (k <= 12, large shards): tiles of the FIRST stage, 0 = one stage
    int mid_tiles = 0;    // sketch cascade, large k: tiles of a SECOND stage behind the first (0 = two levels)
    int pre_tiles = 0;    // sketch cascade, larger k still: the first stage's leading tiles as a stage of their own (0 = none)
    int seeds_per_tile = 2;  // 8 for shards of fewer than 24 tiles (a 1000-row shard has 4: 8 seeds per query would leave the
                             // thresholds at -inf and every list flooding: 0.43 ms per search instead of 0.08)
};

// The sketch search (capi: vqa_index_search) serves large shards -- those the two-stage plan serves -- for k up to kSketchMaxK:
// txtai's hybrid search asks the dense index for 10 x limit rows (30 at its default limit).  The candidates of a query grow with k
// (theta sits at rank k: ~860 rows at k = 10, ~2300 at k = 30, ~13 000 at k = 100 of a 10M-row shard -- half of them from the first
// stage, whose seed threshold is the weaker one) and the exact re-scoring with them; txtai's hybrid search at limit 10 asks for 100.

static LaunchPlan plan_launch(const vqa_index* ix, int k = 0) {
    LaunchPlan p;
    p.tiles = (int)((ix->n + 255) / 256);
    p.grid1 = p.tiles < ix->max_grid ? p.tiles : ix->max_grid;
    // Seed pass: the first min(tiles, seed_mult * CUs) tiles are scored once more by the MODE 0 kernel, which only keeps 2
    // sub-maxima per query and tile; their k-th largest is a valid lower bound of the k-th best score and seeds every
    // workgroup's thresholds, so the main pass appends ~k / (256 * grid) of the scores instead of flooding its
    // candidate lists on each workgroup's first tiles.  Costs <= 512 tiles of extra scoring (1.3 % at 10M rows) and halves
    // the appends of the main pass against 256 seed tiles (measured: net gain).
    if (ix->two_pass) {
        // more seed tiles = tighter starting thresholds = fewer appends; never more than 1/16 of a small shard
        // (a 1M-row fp32 shard spent 13 % of its search in the seed pass with the fixed count; 1/16: step 3.37 -> 3.19 ms)
        int want = ix->seed_mult * ix->max_grid;
        const int div = ix->seed_div;
        if (div > 0 && want > p.tiles / div) want = p.tiles / div > 0 ? p.tiles / div : 1;
        // ... but never fewer than 24 (or all there are): a tile gives 2 seeds per query, and with fewer than k of them the
        // threshold stays at -inf and every workgroup floods its lists on its first tile -- a 20 000-row shard (79 tiles,
        // 4 seed tiles by the 1/16 rule) spent 0.39 ms there, 4x what a 65 536-row shard takes
        const int floor_tiles = kPlan.seed_floor_tiles;
        if (want < floor_tiles) want = floor_tiles;
        // a large k on a small shard: the threshold is the k-th largest seed, so there must be k of them
        if (2 * want < k) want = (k + 1) / 2;
        p.seed_tiles = p.tiles < want ? p.tiles : want;
        if (p.seed_tiles < floor_tiles || 2 * p.seed_tiles < k) p.seeds_per_tile = 8;
        p.grid0 = p.seed_tiles < ix->max_grid ? p.seed_tiles : ix->max_grid;
        // Two stages: a workgroup only knows its own rows and the seeds, so ~5 candidates per tile pass its threshold test
        // and the epilogue's rare path costs 4.4 % of the scan.  The first stage_pct % of the tiles are
        // therefore scored by a launch of their own; the exact k-th best of THOSE rows (one list merge) is a 7x tighter
        // bound than the seeds' for the main launch over the remaining tiles, and the first stage only needs seeds from
        // half as many tiles.  Both launches flush into one array of 2 x grid lists per query for the final merge.
        if (ix->stage_min_tiles > 0 && p.grid1 == ix->max_grid && p.tiles >= (long long)ix->stage_min_tiles * p.grid1) {
            const int pct = ix->stage_pct;
            const int per_wg = (int)((long long)p.tiles * pct / 100 / p.grid1);
            p.stage_tiles = (per_wg > 0 ? per_wg : 1) * p.grid1;
            const int half = p.seed_tiles / 2 > 0 ? p.seed_tiles / 2 : 1;
            p.seed_tiles = half;
            p.grid0 = p.seed_tiles < ix->max_grid ? p.seed_tiles : ix->max_grid;
            // Sketch cascade, a third level for large k: most candidates of the main scan are admitted by the weakness of theta1 (the
            // k-th best of only 10 % of the rows), not by the bound's slack -- against the k-th best of 30 % of the rows the same bound
            // leaves 2.7x fewer (scripts/probes/decomp_probe.py) -- and their number grows with k.  [first | 2 x first | rest].
            // Small k: the level pays for its ~45 us (a launch ramp, a re-scoring, a selection) only on shards of >= 128 tiles per workgroup
            // (10M x 768, k = 10, interleaved A/B: 1.936 -> 1.888 ms; 3M rows: +4 %).
            if (ix->sketch && ix->cascade && ix->mid_k > 0 && k <= kSketchMaxK &&
                (k >= ix->mid_k || (ix->mid_min_tiles > 0 && p.tiles >= (long long)ix->mid_min_tiles * p.grid1))) {
                const int mid = (int)((long long)p.stage_tiles * ix->mid_pct / 100 / p.grid1) * p.grid1;
                if (mid > 0 && p.stage_tiles + mid + p.grid1 <= p.tiles) p.mid_tiles = mid;
                // k >= pre_k: the first stage runs against the seeds' threshold -- the k-th best of only 131 072 rows -- and leaves most of a
                // large-k search's pairs; its leading quarter as a stage of its own gives the other three quarters a threshold 4x as many rows deep
                if (p.mid_tiles > 0 && ix->pre_k > 0 && k >= ix->pre_k) {
                    const int pre = p.stage_tiles / 4 / p.grid1 * p.grid1;
                    if (pre >= 2 * p.grid1) p.pre_tiles = pre;  // (the exact seeds' tiles -- 2 per workgroup -- lie inside it: their k rows make its lists k deep)
                }
            }
        }
    }
    return p;
}

// the exact paths of this shard end in the re-scoring arithmetic (sketch.hip final_rescore_kernel): fp16 / fp32 shards of the size at which
// a sketch search can exist beside the exact scan (whether or not THIS handle keeps a sketch: `sketch=False` answers in the same bits)
static bool final_rescore_applies(const vqa_index* ix, const LaunchPlan& p) {
    return ix->final_fma && ix->fin_pos && (ix->dtype == VQA_F16 || ix->dtype == VQA_F32) && ix->scale == 1.0f && ix->fin_min_tiles > 0 &&
           (long long)p.tiles >= (long long)ix->fin_min_tiles * ix->max_grid;
}

static bool sketch_active(const vqa_index* ix, const LaunchPlan& p, int k) {
    return ix->sketch && p.stage_tiles > 0 && p.grid0 > 0 && k <= kSketchMaxK;
}

extern "C" int vqa_index_launch_info(const vqa_index* ix, int32_t B, int32_t k, vqa_launch_info* out) {
    VQA_REQUIRE(ix && out, "vqa_index_launch_info: null pointer");
    const LaunchPlan p = plan_launch(ix, k);
    (void)B;
    out->grid = p.grid1;
    out->block = 512;
    out->lds_bytes = vqa_score_topk_lds_bytes(ix->dtype, k);
    out->rows_per_tile = 256;
    out->sketch_scan = sketch_active(ix, p, k) ? 1 : 0;
    out->first_stage_rows = (k <= vqa_score_topk_max_k(ix->dtype) || out->sketch_scan) ? (int64_t)(p.stage_tiles + (out->sketch_scan ? p.mid_tiles : 0)) * 256 : 0;
    out->rows_per_launch = ix->n - out->first_stage_rows;
    out->levels = out->sketch_scan ? 2 + (p.mid_tiles > 0) + (p.pre_tiles > 0) : (p.stage_tiles > 0 && k <= vqa_score_topk_max_k(ix->dtype) && p.grid0 > 0) ? 2 : 1;
    out->bytes_per_launch = out->rows_per_launch * (int64_t)ix->d * (out->sketch_scan ? 1 : elem_bytes(ix->dtype));
    out->flops_per_launch = 2 * (int64_t)VQA_QUERY_TILE * out->rows_per_launch * (int64_t)ix->d;
    out->seed_grid = p.grid0;
    out->seed_tiles = p.seed_tiles;
    out->scan_kernel = 0;
    if (out->sketch_scan && ix->sketch_regq && !ix->per_row) {  // the rule of vqa_sketch_regq_applies for this shard's scans
        ScoreTopkArgs a;
        SketchScanArgs sk;
        a.sketch = &sk;
        a.d_pad = ix->d_pad8;
        a.tile_begin = 0;
        a.tile_end = p.tiles;
        a.grid = p.grid1;
        out->scan_kernel = vqa_sketch_regq_applies(a) ? 1 : 0;
    }
    return VQA_OK;
}

extern "C" int vqa_index_set_timing(vqa_index* ix, int32_t enabled) {
    VQA_REQUIRE(ix, "vqa_index_set_timing: index is null");
    ix->timing = enabled != 0;
    if (enabled == 1) ix->ev_used = 0;  // 1: from scratch; 0 / 2: stop / resume -- the pairs recorded so far stay until vqa_index_get_timing
                                        // (sampled timing: bench.py brackets every 4th step)
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

// arguments every exact launch of a search shares (the caller sets k, the tile range, the grid and the list layout)
static ScoreTopkArgs exact_launch_args(const vqa_index* ix, int nq, int k) {
    ScoreTopkArgs a;
    a.x = ix->rows;
    a.q = ix->q_stage;
    a.n = ix->n;
    a.d_pad = ix->d_pad;
    a.nq = nq;
    a.k = k;
    a.loop = ix->f16_loop == 1 ? 1 : 0;
    a.partial = ix->partial;
    return a;
}

// Seed pass: the plan's first seed_tiles tiles scored by the MODE 0 kernel (sub-maxima per query and tile) and the `rank`-th
// largest of them per query -> thr_out (ix->thr0 by default), the starting threshold of the launches behind it (`a` carries upper / gate)
static int seed_pass(vqa_index* ix, const LaunchPlan& p, ScoreTopkArgs a, int rank, const int* gate, hipStream_t stream, float* thr_out = nullptr,
                     const MergeSketchTail* tail = nullptr) {
    a.thr_init = nullptr;
    a.tile_begin = 0;
    a.tile_end = p.seed_tiles;
    a.grid = p.grid0;
    a.seed_only = true;
    a.seeds_per_tile = p.seeds_per_tile;
    int rc = vqa_launch_score_topk(ix->dtype, a, stream);
    if (rc != VQA_OK) return rc;
    return vqa_launch_merge_partials(ix->partial, p.seed_tiles, p.seeds_per_tile, a.nq, rank, nullptr, 0, nullptr, nullptr, nullptr,
                                     thr_out ? thr_out : ix->thr0, 1.0f, rank, 0, nullptr, true, gate, stream, 0, nullptr, 1, tail);
}

// One sketch scan of tiles [tile_begin, tile_end) against theta = thr[] and the exact scores of what it leaves: per-query
// constants -> int8 scan -> re-scoring of its candidate pairs (and of `stage_k` first-stage rows per query, ix->stage_pos) into
// the queries' candidate lists.  `clear`: 1 / 2 = this is the query tile's first scan (counters and overflow flag start at zero;
// 2: the first tile of the call), 0 = a later one.
// the merge in front of a sketch scan can write that scan's per-query constants itself (MergeSketchTail: the cascade), which saves the
// sketch_qconst launch: `thr` == nullptr then
static MergeSketchTail qconst_tail(const vqa_index* ix, int clear) {
    MergeSketchTail t;
    t.qconst = ix->qconst;
    t.qscale = ix->qrow;
    t.qlo = ix->qrow + VQA_QUERY_TILE;
    t.qnorm = ix->qrow + 2 * VQA_QUERY_TILE;
    t.qoff = ix->center ? ix->qoff : nullptr;
    if (ix->wdir) {
        t.qalpha = ix->qrow + 3 * VQA_QUERY_TILE;
        t.qrnorm = ix->qrow + 4 * VQA_QUERY_TILE;
    }
    t.fp_margin = vqa_sketch_fp_margin(ix->d_pad, ix->rotate, ix->per_row);
    t.mu_margin = 3e-7f * ix->mu_norm;
    t.cand_cnt = ix->cand_cnt;
    t.overflow = ix->sketch_flag;
    t.clear = clear;
    t.seq = ix->sketch_seq;
    return t;
}

static int sketch_scan_rescore(vqa_index* ix, const LaunchPlan& p, const float* thr, int tile_begin, int tile_end, int nq, int stage_k, int clear,
                               bool timed, hipStream_t stream) {
    const int seq = ix->sketch_seq;
    int rc = VQA_OK;
    if (thr)
        rc = vqa_launch_sketch_qconst(thr, ix->qrow, ix->qrow + VQA_QUERY_TILE, ix->qrow + 2 * VQA_QUERY_TILE, ix->d_pad, ix->qconst,
                                      ix->cand_cnt, ix->sketch_flag, clear, seq, ix->rotate, ix->center ? ix->qoff : nullptr, 3e-7f * ix->mu_norm, stream,
                                      ix->wdir ? ix->qrow + 3 * VQA_QUERY_TILE : nullptr, ix->wdir ? ix->qrow + 4 * VQA_QUERY_TILE : nullptr, ix->per_row);
    if (rc != VQA_OK) return rc;
    SketchScanArgs sk;
    sk.tile_info = reinterpret_cast<const float4*>(ix->tile_info);
    sk.qconst = ix->qconst;
    sk.tile_c = ix->wdir ? ix->tile_c : nullptr;
    sk.beta = ix->per_row ? ix->beta : nullptr;
    sk.regions = ix->regions;
    sk.counts = ix->region_cnt;
    sk.overflow = ix->sketch_flag;
    sk.cap = kSketchCap;
    ScoreTopkArgs b;
    b.x = ix->rows8;
    b.q = ix->q8_stage;
    b.n = ix->n;
    b.d_pad = ix->d_pad8;
    b.nq = nq;
    b.k = vqa_score_topk_max_k(ix->dtype);
    b.tile_begin = tile_begin;
    b.tile_end = tile_end;
    b.grid = p.grid1;
    b.sketch = &sk;
    b.first_stage = tile_end < p.tiles;  // (its own kernel symbol in a trace)
    b.loop = ix->sketch_sx5 ? 1 : 0;
    b.regq = ix->sketch_regq;
    timed = timed && ix->timing;
    if (timed && (rc = timing_event(ix, stream)) != VQA_OK) return rc;
    rc = vqa_launch_score_topk(VQA_I8_SKETCH, b, stream);
    if (rc != VQA_OK) return rc;
    if (timed && (rc = timing_event(ix, stream)) != VQA_OK) return rc;
    return vqa_launch_rescore(ix->regions, ix->region_cnt, kSketchCap, p.grid1, stage_k > 0 ? ix->stage_pos : nullptr, nq, stage_k, ix->rows,
                              ix->rows_rm, ix->q_stage, ix->q_rm, ix->dtype, ix->d_pad, ix->cand_keys, ix->cand_cnt, kSketchCap, ix->sketch_flag, stream);
}

// the k best of every query's candidate list -> the caller's outputs (os == nullptr: only their k-th score -> ix->thr0).  The
// caller enqueues its exact fallback behind, gated on ix->sketch_flag.
static int sketch_select(vqa_index* ix, int nq, int k, float* os, int64_t* oi, int64_t* op, hipStream_t stream, const MergeSketchTail* tail = nullptr) {
    int rc = vqa_launch_merge_partials(ix->cand_keys, kSketchSubLists, kSketchCap / kSketchSubLists, nq, k, os ? ix->ids : nullptr, ix->id_base, os,
                                       oi, op, os ? nullptr : ix->thr0, 1.0f, k, 0, nullptr, true, nullptr, stream, kSketchSubLists, ix->cand_cnt,
                                       kSketchCntStride, tail);
    return rc;
}

static int search_impl(vqa_index* ix, const void* q, int32_t q_dtype, int32_t B, int32_t k, float* out_scores, int64_t* out_ids,
                       int64_t* out_pos_or_null, hipStream_t stream);

extern "C" int vqa_index_search(vqa_index* ix, const void* q, int32_t q_dtype, int32_t B, int32_t k, float* out_scores,
                                int64_t* out_ids, int64_t* out_pos_or_null, void* hip_stream) {
    VQA_REQUIRE(ix, "vqa_index_search: index is null");
    VQA_REQUIRE(q && out_scores && out_ids, "vqa_index_search: null pointer");
    VQA_REQUIRE(B >= 1, "vqa_index_search: B=%d", B);
    VQA_REQUIRE(k >= 1 && k <= VQA_MAX_K_TOTAL, "vqa_index_search: k=%d outside [1, %d]", k, VQA_MAX_K_TOTAL);
    VQA_REQUIRE(q_dtype == VQA_F32 || q_dtype == VQA_F16, "vqa_index_search: q_dtype %d is not f32/f16", q_dtype);
    HandleBusy busy(ix->busy);
    VQA_REQUIRE(busy.ok, "vqa_index_search: this index handle is in use by another host thread (one call at a time per handle)");
    DeviceGuard guard(ix->device);
    return search_impl(ix, q, q_dtype, B, k, out_scores, out_ids, out_pos_or_null, (hipStream_t)hip_stream);
}

// ---- the latency form: host pointers in, host pointers out, ONE call (heavy_ranker.py:97-101 asks one question at a time with
// limit = 1).  Nothing is allocated per call and no copy operation is queued: the queries are copied by the CPU into a pinned,
// device-mapped buffer of the handle which the first kernel reads over the bus (3 KB per query), the last merge writes the k
// results straight into pinned memory, and the host waits by polling the stream instead of sleeping on an interrupt.
extern "C" int vqa_index_search_host(vqa_index* ix, const void* q_host, int32_t q_dtype, int32_t B, int32_t k, int32_t normalize,
                                     float* out_scores, int64_t* out_ids, int64_t* out_pos_or_null, void* hip_stream) {
    VQA_REQUIRE(ix, "vqa_index_search_host: index is null");
    VQA_REQUIRE(q_host && out_scores && out_ids, "vqa_index_search_host: null pointer");
    VQA_REQUIRE(B >= 1 && B <= 65536, "vqa_index_search_host: B=%d outside [1, 65536]", B);
    VQA_REQUIRE(k >= 1 && k <= VQA_MAX_K_TOTAL, "vqa_index_search_host: k=%d outside [1, %d]", k, VQA_MAX_K_TOTAL);
    VQA_REQUIRE(q_dtype == VQA_F32 || q_dtype == VQA_F16, "vqa_index_search_host: q_dtype %d is not f32/f16", q_dtype);
    VQA_REQUIRE(!normalize || q_dtype == VQA_F32, "vqa_index_search_host: only fp32 queries can be L2-normalised by the call");
    hipStream_t stream = (hipStream_t)hip_stream;
    HandleBusy busy(ix->busy);
    VQA_REQUIRE(busy.ok, "vqa_index_search_host: this index handle is in use by another host thread (one call at a time per handle)");
    DeviceGuard guard(ix->device);
    const size_t qbytes = (size_t)B * ix->d * (q_dtype == VQA_F32 ? 4 : 2), nres = (size_t)B * k;
    const size_t q_off = 0, s_off = (qbytes + 63) / 64 * 64, i_off = s_off + (nres * 4 + 63) / 64 * 64, p_off = i_off + nres * 8;
    const size_t need = p_off + nres * 8;
    // one launch for the whole call where K4 applies (tiny_search.hip); `timing` handles time the scan kernel of the general path
    // (a shard on the two-stage plan -- only tests bring that down to this size -- answers in the re-scoring arithmetic: the general launches)
    const bool one_launch = ix->one_launch && !ix->timing && !ix->sketch && ix->scale == 1.0f && vqa_tiny_search_applies(ix->dtype, ix->n, ix->d_pad, B, k) &&
                            !final_rescore_applies(ix, plan_launch(ix, k));
    if (one_launch && !ix->tiny_ws) {
        const size_t ws = vqa_tiny_search_workspace_bytes();
        if (hipMalloc(&ix->tiny_ws, ws) != hipSuccess) {
            (void)hipGetLastError();
            ix->tiny_ws = nullptr;
            vqa_set_error("vqa_index_search_host: allocating %zu device bytes failed", ws);
            return VQA_ENOMEM;
        }
        VQA_HIP_CHECK(hipMemsetAsync(ix->tiny_ws, 0, ws, stream));  // the ticket starts at zero; every call leaves it there
    }
    if (ix->hio_bytes < need || ((normalize || one_launch) && ix->hq_norm_bytes < qbytes)) {
        VQA_HIP_CHECK(hipStreamSynchronize(stream));  // (an earlier call's kernels may still read the buffers about to be replaced)
        if (ix->hio_bytes < need) {
            if (ix->hio) (void)hipHostFree(ix->hio);
            ix->hio = ix->hio_dev = nullptr;
            ix->hio_bytes = 0;
            const size_t cap = std::max<size_t>(need, 64 << 10);
            if (hipHostMalloc(&ix->hio, cap, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer(&ix->hio_dev, ix->hio, 0) != hipSuccess) {
                (void)hipGetLastError();
                vqa_set_error("vqa_index_search_host: allocating %zu pinned bytes failed", cap);
                return VQA_ENOMEM;
            }
            ix->hio_bytes = cap;
        }
        if ((normalize || one_launch) && ix->hq_norm_bytes < qbytes) {
            if (ix->hq_norm) (void)hipFree(ix->hq_norm);
            ix->hq_norm = nullptr;
            ix->hq_norm_bytes = 0;
            const size_t cap = std::max<size_t>(qbytes, 64 << 10);
            if (hipMalloc(&ix->hq_norm, cap) != hipSuccess) {
                (void)hipGetLastError();
                vqa_set_error("vqa_index_search_host: allocating %zu device bytes failed", cap);
                return VQA_ENOMEM;
            }
            ix->hq_norm_bytes = cap;
        }
    }
    char* h = static_cast<char*>(ix->hio);
    char* dv = static_cast<char*>(ix->hio_dev);
    // the queries may already be on the device (the question encoder's output: the text form of the reference's call, one C call for the
    // forward and this one for the search, heavy_ranker.py:98): then only the results travel through the pinned buffer
    hipPointerAttribute_t qattr;
    const bool q_on_device = hipPointerGetAttributes(&qattr, q_host) == hipSuccess && qattr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // an unregistered host pointer reports an error: not ours
    const void* q_dev = q_host;
    if (!q_on_device) {
        memcpy(h + q_off, q_host, qbytes);
        q_dev = dv + q_off;
    }
    int rc;
    if (one_launch) {
        rc = vqa_launch_tiny_search(ix->rows, ix->dtype, ix->n, ix->d, ix->d_pad, q_dev, q_on_device ? nullptr : q_host, ix->hq_norm, q_dtype, normalize, B, k, ix->ids, ix->id_base, ix->tiny_ws,
                                    reinterpret_cast<float*>(dv + s_off), reinterpret_cast<int64_t*>(dv + i_off),
                                    out_pos_or_null ? reinterpret_cast<int64_t*>(dv + p_off) : nullptr, stream);
    } else {
        if (normalize) {  // x / ||x|| by the kernel every other path uses (Embeddings.batchsearch, vqa_normalize_convert): the same bits
            rc = vqa_normalize_convert(reinterpret_cast<const float*>(q_dev), B, ix->d, 1, VQA_F32, ix->hq_norm, stream);
            if (rc != VQA_OK) return rc;
            q_dev = ix->hq_norm;
        }
        rc = search_impl(ix, q_dev, q_dtype, B, k, reinterpret_cast<float*>(dv + s_off), reinterpret_cast<int64_t*>(dv + i_off),
                         out_pos_or_null ? reinterpret_cast<int64_t*>(dv + p_off) : nullptr, stream);
    }
    if (rc != VQA_OK) return rc;
    // poll: a search of a small shard is a handful of launches of a few microseconds; sleeping on the completion interrupt costs more
    // than they take.  After ~0.2 ms of polling (large shards) the blocking wait takes over.
    hipError_t st = hipErrorNotReady;
    // (ONE short kernel: the blocking wait itself spins first and sees the completion ~3.5 us sooner than a hipStreamQuery loop --
    // scripts/probes/launch_sync_floor.hip: 10.8 against 14.6 us around an empty kernel)
    const int spins = one_launch ? 0 : 400;
    for (int spin = 0; spin < spins && st == hipErrorNotReady; ++spin) st = hipStreamQuery(stream);
    if (st == hipErrorNotReady) {
        (void)hipGetLastError();
        st = hipStreamSynchronize(stream);
    }
    if (st != hipSuccess) {
        vqa_set_error("vqa_index_search_host: waiting for the search failed: %s", hipGetErrorString(st));
        // a one-launch search that died between its ticket and the ticket's reset would leave the NEXT call's last workgroup undetected
        // (stale pinned results returned as VQA_OK): the workspace starts from zero again (best effort: the device may be gone)
        if (one_launch && ix->tiny_ws) (void)hipMemset(ix->tiny_ws, 0, vqa_tiny_search_workspace_bytes());
        (void)hipGetLastError();
        return VQA_EHIP;
    }
    memcpy(out_scores, h + s_off, nres * 4);
    memcpy(out_ids, h + i_off, nres * 8);
    if (out_pos_or_null) memcpy(out_pos_or_null, h + p_off, nres * 8);
    return VQA_OK;
}

static int search_impl(vqa_index* ix, const void* q, int32_t q_dtype, int32_t B, int32_t k, float* out_scores, int64_t* out_ids,
                       int64_t* out_pos_or_null, hipStream_t stream) {
    const int max_k = vqa_score_topk_max_k(ix->dtype);  // per pass over the index
    const int qeb = q_dtype == VQA_F32 ? 4 : 2;
    const LaunchPlan p = plan_launch(ix, k);
    bool sketch_call = false;  // some query tile of this call ran the sketch search
    bool mirror_by_kernel = false;  // ... and its last merge reports the overflow flags to the host itself (the cascade)
    // Query tiles of equal size: B = 257 runs as 129 + 128, not 256 + 1 -- the sketch scan's cost follows the waves that hold queries
    // (32 each: 0.80 ms up to 128 queries, 0.95 at 160, 1.08 at 256 on a 10M-row shard), every other launch of a tile the queries themselves
    const int q_tiles = (B + VQA_QUERY_TILE - 1) / VQA_QUERY_TILE;
    const int q_per = q_tiles > 1 ? (B + q_tiles - 1) / q_tiles : VQA_QUERY_TILE;
    for (int q0 = 0; q0 < B; q0 += q_per) {
        const int nq = B - q0 < q_per ? B - q0 : q_per;
        float* os = out_scores + (size_t)q0 * k;
        int64_t* oi = out_ids + (size_t)q0 * k;
        int64_t* op = out_pos_or_null ? out_pos_or_null + (size_t)q0 * k : nullptr;
        if (ix->n == 0) {  // empty shard: every slot is padding
            // reuse the merge kernel on one all-empty partial list, max_k columns at a time
            VQA_HIP_CHECK(hipMemsetAsync(ix->partial, 0, (size_t)VQA_QUERY_TILE * max_k * sizeof(vqa_key), stream));
            for (int done = 0; done < k; done += max_k) {
                const int kk = k - done < max_k ? k - done : max_k;
                int rc = vqa_launch_merge_partials(ix->partial, 1, max_k, nq, kk, ix->ids, ix->id_base, os, oi, op, nullptr, 1.0f, k,
                                                   done, nullptr, false, nullptr, stream);
                if (rc != VQA_OK) return rc;
            }
            continue;
        }
        int rc = VQA_OK;
        // An earlier sketch search of this handle that overflowed into its exact fallback (its flag arrives in the pinned mirror
        // some time after that call; a stale read only delays the reaction by a search) switches the sketch off for a while
        if (ix->sketch && q0 == 0) {
            // (the three words arrive by one 12-byte DMA; the number is read first, so a report caught half-way is at worst taken
            // for the previous call's and looked at again next time)
            const int seq = __atomic_load_n(ix->sketch_flag_host + 2, __ATOMIC_ACQUIRE);
            // "over": a candidate buffer overflowed (the call's exact fallback ran), or the call scored so many pairs exactly that the exact
            // scan would have been cheaper (its result stands: this only decides what the NEXT searches run)
            const bool over = (__atomic_load_n(ix->sketch_flag_host, __ATOMIC_RELAXED) | __atomic_load_n(ix->sketch_flag_host + 1, __ATOMIC_RELAXED)) != 0 ||
                              (ix->profit_ratio > 0.0 && (double)pairs_reported(ix) > profit_pairs(ix));
            if (seq != ix->sketch_seq_seen) {  // a sketch search completed since the last look
                ix->sketch_seq_seen = seq;
                // (reports of calls that were already queued when the current pause began say nothing new)
                const bool fresh = (int)((unsigned)seq - (unsigned)ix->sketch_seq_ignore) > 0;
                if (over && fresh) {
                    const bool overflowed = (__atomic_load_n(ix->sketch_flag_host, __ATOMIC_RELAXED) | __atomic_load_n(ix->sketch_flag_host + 1, __ATOMIC_RELAXED)) != 0;
                    const int min_k = overflowed ? 0 : (ix->k_of_seq[seq & 63] + 1) / 2;
                    ix->pause_min_k = ix->sketch_cooldown > 0 ? std::min(ix->pause_min_k, min_k) : min_k;
                    ix->sketch_cooldown = ix->sketch_cooldown_cur;
                    // overflow, pause, overflow again: the pause doubles (64, 128, ... 4096 searches)
                    ix->sketch_cooldown_cur = std::min(2 * ix->sketch_cooldown_cur, 64 * ix->sketch_cooldown_len);
                    ix->sketch_seq_ignore = ix->sketch_seq;
                } else if (!over && fresh && ix->sketch_cooldown == 0) {
                    ix->sketch_cooldown_cur = ix->sketch_cooldown_len;  // a sketch search stood: back to the base pause
                }
            } else if (ix->sketch_cooldown > 0 && k >= ix->pause_min_k) {
                --ix->sketch_cooldown;  // (the pause counts the searches it applies to)
            }
        }
        const bool any_sketch = sketch_active(ix, p, k) && (ix->sketch_cooldown == 0 || k < ix->pause_min_k);
        if (any_sketch && q0 == 0) {
            ++ix->sketch_seq;
            ix->k_of_seq[ix->sketch_seq & 63] = k;
            sketch_call = true;
        }
        const int sk_clear = q0 == 0 ? 2 : 1;  // the call's first query tile also clears the OR over the tiles
        const bool use_sketch = any_sketch && k <= max_k && !ix->cascade;  // round 3.s form (options.sketch_cascade = 0: A/B switch)
        const void* q_tile = reinterpret_cast<const char*>(q) + (size_t)q0 * ix->d * qeb;
        if (any_sketch) {
            // one launch: the query tile in the storage type (tiled for the exact kernels, row-major for the re-scoring) and its int8
            // sketch (every query its own scale) + ||q_lo||, ||q||, q . mu
            VqaQueryRows qr;
            qr.rows = q_tile;
            qr.src_dtype = q_dtype;
            qr.valid = nq;
            qr.d = ix->d;
            qr.scale = ix->scale;
            qr.stage = ix->q_stage;
            qr.rowmajor = ix->q_rm;
            SketchSplit sp;  // + |alpha|, ||z_r||: the query's share of the split slack term
            sp.wdir = ix->wdir;
            sp.row_alpha = ix->qrow + 3 * VQA_QUERY_TILE;
            sp.row_rnorm = ix->qrow + 4 * VQA_QUERY_TILE;
            sp.per_row = ix->per_row ? 1 : 0;
            rc = vqa_launch_sketch_rows(nullptr, ix->dtype, 0, VQA_QUERY_TILE, ix->d_pad, ix->d_pad8, nullptr, ix->q8_stage, ix->qrow,
                                        ix->qrow + VQA_QUERY_TILE, ix->qrow + 2 * VQA_QUERY_TILE, ix->rotate, ix->center ? ix->mu : nullptr,
                                        false, ix->qoff, stream, &qr, ix->wdir ? &sp : nullptr);
        } else {
            rc = vqa_launch_tile_rows(q_tile, q_dtype, 0, VQA_QUERY_TILE, nq, ix->d, ix->d_pad, ix->dtype, ix->scale, ix->q_stage, stream);
        }
        if (rc != VQA_OK) return rc;
        // k <= 12: one pass.  Larger k, first attempt: ONE pass in which every workgroup keeps its local top 12 above the
        // seeded threshold (the k-th largest seed, a valid lower bound of the k-th best score), merged to k results.  The
        // global top-k is inside the union of the local lists unless some workgroup owns more than 12 of them (with 256
        // workgroups: probability ~1e-10 at k = 256 on unclustered data); vqa_launch_verify_wide checks exactly that on
        // the device and raises wide_flag.  The exact continuation passes below are then launched GATED on the flag: they
        // return at once when the one-pass result stands (no host round trip), and overwrite it when it does not.
        const int* gate = nullptr;
        // Large fp16 / fp32 shards (the size at which a sketch search can exist for the type): every exact path ends by scoring its rows in
        // the re-scoring kernel's arithmetic and re-ranking them (sketch.hip final_rescore_kernel), so that the exact scan -- during a
        // pause of the sketch, behind an overflow, with the sketch off -- returns the bits the sketch path returns.  One pass (k <= 10):
        // the scan keeps k + 2 rows per query, so rows the MFMA order ranks just below the k-th compete too.
        const bool fin = final_rescore_applies(ix, p);
        int64_t* const wide_pos = fin && k > max_k ? (op ? op : ix->fin_pos) : op;  // k > 12: the passes' rows, re-scored in place at the end
        MergeSketchTail cascade_report;  // (set by the cascade: the report its fallback's first merge carries)
        if (any_sketch && ix->cascade) {
            // A sketch shard, k <= 64, as a cascade of bounds -- no exact scan of a first stage at all:
            //   exact seeds (2 grid tiles: sub-maxima of real rows)  ->  theta0, a valid lower bound of the k-th best score;
            //   sketch scan of the first stage's tiles against theta0, its candidates scored exactly -> theta1 = the exact k-th best
            //   score of those rows (every row of theirs that reaches theta0 is a candidate; the seeds' k rows are among them);
            //   sketch scan of the other tiles against theta1, candidates scored exactly into the same lists; the k best of a list
            //   are the result.  An overflow anywhere raises sketch_flag, and the exact search runs gated on it: for k <= 12 ONE pass
            //   over all tiles seeded with theta0 -- NOT theta1: that is a sum in the re-scoring kernel's order, and the scan's
            //   `score >= threshold` test on MFMA sums could drop the very row that defines it by an ulp; theta0 is an MFMA sum of
            //   the same kernel family (test_small_k_fallback_keeps_the_row_that_defines_theta) --, for larger k the gated exact
            //   passes below.
            LaunchPlan pc = p;
            pc.seed_tiles = std::min(2 * p.grid1, p.stage_tiles);  // (1-4 tiles per workgroup x 6-14 % first stage: 2.00-2.03 ms at 10M rows)
            pc.grid0 = std::min(pc.seed_tiles, ix->max_grid);
            pc.seeds_per_tile = 2;
            ScoreTopkArgs a = exact_launch_args(ix, nq, k <= max_k ? k : max_k);
            // (every merge also does what sits between it and the next scan: the scan's per-query constants from the threshold it
            // has just selected, the reset of the candidate counters; the call's last one reports the overflow flags to the host)
            MergeSketchTail t0 = qconst_tail(ix, sk_clear), t2;
            t2.overflow = ix->sketch_flag;
            t2.min_score = ix->thr0;  // theta1: what the first selection left there
            // the call's report to the host: written by the gated fallback's merge BEHIND the last selection (any block of that
            // selection may still raise the overflow flag when block 0 is done: ADVICE r4), by the last query tile of the call
            MergeSketchTail tm;
            tm.overflow = ix->sketch_flag;
            tm.flag_mirror = q0 + q_per >= B ? ix->sketch_flag_dev_mirror : nullptr;
            tm.mirror_before_gate = 1;
            cascade_report = tm;
            rc = seed_pass(ix, pc, a, k, nullptr, stream, ix->thr_seed, &t0);  // theta0 -> thr_seed
            if (rc != VQA_OK) return rc;
            // the stages: [leading quarter of the first (k >= pre_k) |] first | [second (k >= mid_k or a large shard) |] the rest
            int bounds[4], nb = 0;
            if (p.pre_tiles > 0) bounds[nb++] = p.pre_tiles;
            bounds[nb++] = p.stage_tiles;
            if (p.mid_tiles > 0) bounds[nb++] = p.stage_tiles + p.mid_tiles;
            bounds[nb++] = p.tiles;
            for (int l = 0, begin = 0; l < nb; begin = bounds[l], ++l) {
                const bool last = l == nb - 1;
                rc = sketch_scan_rescore(ix, p, nullptr, begin, bounds[l], nq, 0, l == 0 ? sk_clear : 0, last, stream);
                if (rc != VQA_OK) return rc;
                if (last) {
                    rc = sketch_select(ix, nq, k, os, oi, op, stream, &t2);
                } else {  // theta_l = the exact k-th best of the rows scanned so far -> thr0 (and the next scan's constants)
                    MergeSketchTail tl = qconst_tail(ix, 0);
                    if (l > 0) tl.min_score = ix->thr0;  // keys below the previous level's threshold cannot matter any more (read while the lists are gathered, before the new threshold is written)
                    rc = sketch_select(ix, nq, k, nullptr, nullptr, nullptr, stream, &tl);
                }
                if (rc != VQA_OK) return rc;
            }
            mirror_by_kernel = true;
            if (k <= max_k) {
                a.thr_init = ix->thr_seed;
                a.tile_begin = 0;
                a.tile_end = p.tiles;
                a.grid = p.grid1;
                a.gate = ix->sketch_flag;
                rc = vqa_launch_score_topk(ix->dtype, a, stream);
                if (rc != VQA_OK) return rc;
                int64_t* fpos = fin ? (op ? op : ix->fin_pos) : op;  // (the final re-scoring reads the rows' positions)
                rc = vqa_launch_merge_partials(ix->partial, p.grid1, k, nq, k, ix->ids, ix->id_base, os, oi, fpos, nullptr,
                                               1.0f / (ix->scale * ix->scale), k, 0, nullptr, true, ix->sketch_flag, stream, 0, nullptr, 1,
                                               &cascade_report);
                if (rc != VQA_OK) return rc;
                // the fallback's rows in the sketch path's arithmetic (gated like the fallback itself)
                if (fin && (rc = vqa_launch_final_rescore(fpos, k, nq, k, k, ix->rows, nullptr, ix->q_stage, nullptr, ix->dtype, ix->d_pad, ix->ids,
                                                          ix->id_base, os, oi, op, k, ix->sketch_flag, stream)) != VQA_OK)
                    return rc;
                continue;
            }
            gate = ix->sketch_flag;  // larger k: the gated exact passes below
        } else if (k > max_k && ix->wide && k <= 3 * p.grid1 && p.seed_tiles > 0) {
            ScoreTopkArgs a = exact_launch_args(ix, nq, max_k);
            VQA_HIP_CHECK(hipMemsetAsync(ix->wide_flag, 0, sizeof(int), stream));
            rc = seed_pass(ix, p, a, k, nullptr, stream);
            if (rc != VQA_OK) return rc;
            a.thr_init = ix->thr0;
            a.tile_begin = 0;
            a.tile_end = p.tiles;
            a.grid = p.grid1;
            a.seed_only = false;
            rc = vqa_launch_score_topk(ix->dtype, a, stream);
            if (rc != VQA_OK) return rc;
            rc = vqa_launch_merge_partials(ix->partial, p.grid1, max_k, nq, k, ix->ids, ix->id_base, os, oi, wide_pos, nullptr,
                                           1.0f / (ix->scale * ix->scale), k, 0, ix->upper, true, nullptr, stream);
            if (rc != VQA_OK) return rc;
            rc = vqa_launch_verify_wide(ix->partial, p.grid1, max_k, nq, ix->upper, ix->wide_flag, stream);
            if (rc != VQA_OK) return rc;
            gate = ix->wide_flag;
        }
        // Exact passes: k <= 12 needs one; larger k ceil(k / 12), each admitting only keys strictly below the last key already
        // returned (keys are distinct, so the continuation is exact); every pass is a full scan of the shard.
        const int kx = fin && k + 2 <= max_k && !use_sketch && ix->n >= k + 2 ? 2 : 0;  // margin rows of a one-pass exact search
        for (int done = 0; done < k; done += max_k) {
            const int kk = (k - done < max_k ? k - done : max_k) + kx;
            const vqa_key* upper = done > 0 ? ix->upper : nullptr;
            ScoreTopkArgs a = exact_launch_args(ix, nq, kk);
            a.upper = upper;
            a.gate = gate;
            // The gated exact passes behind a cascade (k > 12: ceil(k / 12) of them, each "returns at once" -- but an empty launch still costs
            // its ~5 us of dispatch: four launches per pass were 200 us of a 3 ms search at k = 100) need no seed pass of their own: theta0,
            // the k-th largest exact seed of the cascade, bounds the k-th best score from below and with it every pass's 12 j-th.
            // The same holds behind the one-pass attempt of a shard without a sketch (gate = wide_flag): its seed pass left the k-th
            // largest seed in thr0, and nothing overwrites it when the continuation passes do not seed themselves.
            const bool cascade_fallback = any_sketch && ix->cascade && gate == ix->sketch_flag;
            const bool wide_fallback = gate != nullptr && gate == ix->wide_flag && p.grid0 > 0;
            if (!cascade_fallback && !wide_fallback && p.grid0 > 0 && (rc = seed_pass(ix, p, a, kk, gate, stream)) != VQA_OK) return rc;
            a.thr_init = cascade_fallback ? ix->thr_seed : p.grid0 > 0 ? ix->thr0 : nullptr;
            a.tile_begin = 0;
            a.tile_end = p.tiles;
            a.grid = p.grid1;
            a.seed_only = false;
            const bool staged = p.stage_tiles > 0 && k <= max_k && p.grid0 > 0;  // one exact pass of a large shard
            int lists = p.grid1;
            if (staged) {
                // first stage: tiles [0, stage_tiles) with the seeds' thresholds -> lists [0, grid) of every query's row of 2 grid
                // lists; their merge (thresholds only) = the exact k-th best score of those rows, seeds the main launch
                lists = 2 * p.grid1;
                a.row_lists = lists;
                a.list_offset = 0;
                a.tile_end = p.stage_tiles;
                a.first_stage = true;
                rc = vqa_launch_score_topk(ix->dtype, a, stream);
                if (rc != VQA_OK) return rc;
                rc = vqa_launch_merge_partials(ix->partial, p.grid1, kk, nq, kk, nullptr, 0, nullptr, nullptr,
                                               use_sketch ? reinterpret_cast<int64_t*>(ix->stage_pos) : nullptr, ix->thr0, 1.0f, kk, 0,
                                               nullptr, true, gate, stream, lists);
                if (rc != VQA_OK) return rc;
                a.first_stage = false;
                a.tile_begin = p.stage_tiles;
                a.tile_end = p.tiles;
                a.list_offset = p.grid1;
                if (ix->f16_loop == 2) a.loop = 1;
                if (use_sketch) {
                    // Main launch over the int8 sketch (half the bytes, twice the MFMA rate per row): with theta = the first stage's
                    // exact k-th best score it leaves the (query, row) pairs whose rigorous upper bound reaches theta; those and the
                    // first stage's k rows are scored exactly and the k best of each query's list are the result.  Should a
                    // candidate buffer fill up (adversarial data: the bound prunes nothing), sketch_flag sends the search through
                    // the exact main launch below, gated on the flag.
                    rc = sketch_scan_rescore(ix, p, ix->thr0, p.stage_tiles, p.tiles, nq, kk, sk_clear, true, stream);
                    if (rc != VQA_OK) return rc;
                    MergeSketchTail tf;
                    tf.overflow = ix->sketch_flag;  // (a list longer than the selection's LDS raises it)
                    rc = sketch_select(ix, nq, kk, os, oi, op, stream, &tf);
                    if (rc != VQA_OK) return rc;
                    a.gate = ix->sketch_flag;  // the exact main launch + merge below: only when the flag is up
                }
            }
            const bool time_it = ix->timing && !(staged && use_sketch) && !(gate && gate == ix->sketch_flag);  // a sketch search times its sketch scan instead
            if (time_it && (rc = timing_event(ix, stream)) != VQA_OK) return rc;
            rc = vqa_launch_score_topk(ix->dtype, a, stream);
            if (rc != VQA_OK) return rc;
            if (time_it && (rc = timing_event(ix, stream)) != VQA_OK) return rc;
            const bool report_here = cascade_fallback && done == 0;  // (k > 12 behind a cascade: the first gated pass's merge reports)
            const bool one_pass_fin = fin && k <= max_k && !use_sketch;  // its k + kx rows go to the scratch arrays, the re-scoring writes the caller's
            rc = vqa_launch_merge_partials(ix->partial, lists, kk, nq, kk, ix->ids, ix->id_base, one_pass_fin ? ix->fin_scores : os,
                                           one_pass_fin ? ix->fin_ids : oi, one_pass_fin ? ix->fin_pos : wide_pos, nullptr,
                                           1.0f / (ix->scale * ix->scale), one_pass_fin ? kk : k, done, done + kk < k ? ix->upper : nullptr, true,
                                           staged && use_sketch ? ix->sketch_flag : gate, stream, 0, nullptr, 1,
                                           report_here ? &cascade_report : nullptr);
            if (rc != VQA_OK) return rc;
            if (one_pass_fin && (rc = vqa_launch_final_rescore(ix->fin_pos, kk, nq, kk, k, ix->rows, nullptr, ix->q_stage, nullptr, ix->dtype, ix->d_pad,
                                                               ix->ids, ix->id_base, os, oi, op, k, gate, stream)) != VQA_OK)
                return rc;
        }
        // k > 12: the passes' k rows (the verified one-pass attempt's, or the gated continuation passes') re-scored and re-ranked in place;
        // behind a cascade the passes only ran if its flag is up, and so does this
        if (fin && k > max_k && !use_sketch &&
            (rc = vqa_launch_final_rescore(wide_pos, k, nq, k, k, ix->rows, nullptr, ix->q_stage, nullptr, ix->dtype, ix->d_pad, ix->ids, ix->id_base,
                                           os, oi, op, k, gate == ix->sketch_flag ? gate : nullptr, stream)) != VQA_OK)
            return rc;
    }
    // the overflow flags of this call (its last tile's, the OR over the earlier ones, the call's number) -> the pinned mirror a later
    // call's cool-down bookkeeping reads
    if (sketch_call && !mirror_by_kernel)
        VQA_HIP_CHECK(hipMemcpyAsync(ix->sketch_flag_host, ix->sketch_flag, 5 * sizeof(int), hipMemcpyDeviceToHost, stream));
    return VQA_OK;
}

// ---- diagnostics of the sketch search (bench.py prices a step's physical bytes from the pair counts; tests/test_gpu_sketch_bound.py
// compares a tile's codes and maxima with a restatement of its own).  Both synchronise the device: never on a hot path.
extern "C" int vqa_index_sketch_stats(vqa_index* ix, int64_t* out) {
    VQA_REQUIRE(ix && out, "vqa_index_sketch_stats: null pointer");
    VQA_REQUIRE(ix->sketch, "vqa_index_sketch_stats: the shard keeps no sketch");
    DeviceGuard guard(ix->device);
    VQA_HIP_CHECK(hipDeviceSynchronize());
    std::vector<unsigned> rc(ix->max_grid), cc((size_t)VQA_QUERY_TILE * kSketchSubLists * kSketchCntStride);
    int flag[3] = {0, 0, 0};
    VQA_HIP_CHECK(hipMemcpy(rc.data(), ix->region_cnt, rc.size() * 4, hipMemcpyDeviceToHost));
    VQA_HIP_CHECK(hipMemcpy(cc.data(), ix->cand_cnt, cc.size() * 4, hipMemcpyDeviceToHost));
    VQA_HIP_CHECK(hipMemcpy(flag, ix->sketch_flag, sizeof(flag), hipMemcpyDeviceToHost));
    int64_t sum = 0, mx = 0, qsum = 0, mq = 0;
    for (unsigned v : rc) {
        sum += v;
        mx = std::max<int64_t>(mx, v);
    }
    for (size_t i = 0; i < cc.size(); i += kSketchCntStride) {
        qsum += cc[i];
        mq = std::max<int64_t>(mq, cc[i]);
    }
    out[0] = sum;   // candidate pairs the LAST sketch scan of the last query tile left (a cascade: its second, main scan)
    out[1] = mx;    // ... the fullest workgroup region of that scan (capacity kSketchCap)
    out[2] = qsum;  // pairs scored exactly for that query tile: every scan of its cascade
    out[3] = mq;    // ... the longest candidate sub-list (capacity kSketchCap / kSketchSubLists)
    out[4] = flag[0];  // 1: that tile overflowed into its exact fallback
    out[5] = flag[1];  // 1: an earlier query tile of the same call did
    out[6] = kSketchCap;
    out[7] = kSketchCap / kSketchSubLists;
    return VQA_OK;
}

extern "C" int vqa_index_get_sketch_tile(vqa_index* ix, int64_t tile, int8_t* out_codes, float* out_info, float* out_mu_or_null) {
    VQA_REQUIRE(ix && out_codes && out_info, "vqa_index_get_sketch_tile: null pointer");
    VQA_REQUIRE(ix->sketch, "vqa_index_get_sketch_tile: the shard keeps no sketch");
    const int64_t tiles = (ix->n + 255) / 256;
    VQA_REQUIRE(tile >= 0 && tile < tiles, "vqa_index_get_sketch_tile: tile %lld outside [0, %lld)", (long long)tile, (long long)tiles);
    DeviceGuard guard(ix->device);
    const size_t bytes = (size_t)256 * ix->d_pad8;
    void* dev = nullptr;
    if (hipMalloc(&dev, bytes) != hipSuccess) {
        vqa_set_error("vqa_index_get_sketch_tile: hipMalloc of %zu bytes failed", bytes);
        return VQA_ENOMEM;
    }
    // the sketch is a TILED array of one-byte elements (K-blocks of 64): the fp8 form of the untile kernel moves bytes
    int rc = vqa_launch_untile_rows(ix->rows8, tile * 256, 256, ix->d_pad8, ix->d_pad8, VQA_FP8_E4M3, dev, nullptr);
    hipError_t e = rc == VQA_OK ? hipMemcpy(out_codes, dev, bytes, hipMemcpyDeviceToHost) : hipSuccess;
    (void)hipFree(dev);
    if (rc != VQA_OK) return rc;
    VQA_HIP_CHECK(e);
    VQA_HIP_CHECK(hipMemcpy(out_info, ix->tile_info + 4 * tile, 16, hipMemcpyDeviceToHost));
    if (out_mu_or_null) {
        if (ix->mu) VQA_HIP_CHECK(hipMemcpy(out_mu_or_null, ix->mu, (size_t)ix->d_pad8 * 4, hipMemcpyDeviceToHost));
        else memset(out_mu_or_null, 0, (size_t)ix->d_pad8 * 4);
    }
    return VQA_OK;
}

extern "C" int vqa_index_get_sketch_split(vqa_index* ix, int64_t tile, float* out_c, float* out_w_or_null, float* out_beta_or_null,
                                          int32_t* out_per_row_or_null) {
    VQA_REQUIRE(ix && out_c, "vqa_index_get_sketch_split: null pointer");
    VQA_REQUIRE(ix->sketch, "vqa_index_get_sketch_split: the shard keeps no sketch");
    const int64_t tiles = (ix->n + 255) / 256;
    VQA_REQUIRE(tile >= 0 && tile < tiles, "vqa_index_get_sketch_split: tile %lld outside [0, %lld)", (long long)tile, (long long)tiles);
    DeviceGuard guard(ix->device);
    VQA_HIP_CHECK(hipDeviceSynchronize());
    if (ix->wdir) VQA_HIP_CHECK(hipMemcpy(out_c, ix->tile_c + tile, 4, hipMemcpyDeviceToHost));
    else *out_c = 0.f;
    if (out_w_or_null) {
        if (ix->wdir) VQA_HIP_CHECK(hipMemcpy(out_w_or_null, ix->wdir, (size_t)ix->d_pad8 * 4, hipMemcpyDeviceToHost));
        else memset(out_w_or_null, 0, (size_t)ix->d_pad8 * 4);
    }
    if (out_beta_or_null) {
        if (ix->per_row) VQA_HIP_CHECK(hipMemcpy(out_beta_or_null, ix->beta + tile * 256, 256 * 4, hipMemcpyDeviceToHost));
        else memset(out_beta_or_null, 0, 256 * 4);
    }
    if (out_per_row_or_null) *out_per_row_or_null = ix->per_row ? 1 : 0;
    return VQA_OK;
}
