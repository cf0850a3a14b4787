/* vqa_retrieval.h -- C ABI of libvqa_retrieval.so (hand-written HIP for gfx950 / MI355X).
 *
 * Drop-in boundary for ONE path of vTuanpham/Vietnamese_QA_System: the dense retriever its
 * inference_pipeline drives through a txtai-shaped object (reference:
 * inference_pipeline/db_utils/heavy_ranker.py:78-101).  The reference has no FFI of its own (pure Python,
 * scoring delegated to txtai -> faiss); these entry points are what a maintainer binds with ctypes to replace
 * that delegation -- see INTEGRATION.md for the stub.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative VQA_E* code on failure; the message is available from
 *     vqa_last_error() (thread-local).  No C++ exception crosses the ABI, nothing calls exit/abort.
 *   - all device work is enqueued asynchronously on the caller's hipStream_t (passed as void*); no hidden
 *     hipDeviceSynchronize / hipMalloc in search/merge/forward calls (they are hipGraph-capturable).
 *   - the caller allocates every output (device memory); the library owns only what it allocated in *_create.
 *   - different handles may be used from different host threads at once (one-time per-device kernel setup is guarded);
 *     ONE handle serves one call at a time: vqa_index_search and vqa_encoder_forward refuse a second concurrent call on
 *     the same handle with VQA_EINVAL instead of sharing its workspace.
 */
#ifndef VQA_RETRIEVAL_H
#define VQA_RETRIEVAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQA_VERSION 113 /* 0.1.13: vqa_index_options.sketch_regq (csrc/scan_regq.hip), .final_rescore, vqa_launch_info.scan_kernel, vqa_index_sketch_pause; 0.1.12: vqa_index_options.one_launch (vqa_index_search_host of a small fp16 shard is one kernel); 0.1.11: vqa_encoder_forward_host, device queries in vqa_index_search_host; 0.1.10: vqa_index_search_host; 0.1.9: vqa_index_options / vqa_index_create_ex, vqa_encoder_options / vqa_encoder_create_ex (no environment variable is read any more), vqa_launch_info.levels; 0.1.8: vqa_encoder_forward_hidden; 0.1.7: vqa_index_get_sketch_split; 0.1.6: vqa_index_sketch_stats, vqa_index_get_sketch_tile; 0.1.5: vqa_index_sketch_state; 0.1.4: VQA_INDEX_RESCORE_ROWS, vqa_index_device_bytes; 0.1.3: VQA_INDEX_SKETCH */

/* error codes */
#define VQA_OK 0
#define VQA_EINVAL (-1)  /* bad argument (shape, dtype, k, null pointer) */
#define VQA_EHIP (-2)    /* a HIP runtime call failed (message carries hipGetErrorString) */
#define VQA_ENOMEM (-3)  /* device allocation failed */
#define VQA_ENODEV (-4)  /* no gfx950 device / wrong architecture */

/* element types of index rows and query rows */
#define VQA_F32 0
#define VQA_F16 1
#define VQA_FP8_E4M3 2 /* OCP e4m3fn (gfx950), one byte per element */

/* vqa_index_create flags */
#define VQA_INDEX_HAS_IDS 1 /* reserve the id vector even though ids_or_null is NULL (filled later by vqa_index_set_rows) */
#define VQA_INDEX_SKETCH 2  /* fp16 / fp32 shards: keep an int8 sketch of the rows beside them (+50 % / +25 % memory).  Large shards (the
                             * sketch search: >= 16 (fp32: 8) tiles of 256 rows per compute unit; k <= 128) then run their main
                             * launch over the sketch -- v_mfma_i32_16x16x64_i8, half the bytes and twice the matrix rate per
                             * row -- with a rigorous upper bound on every (query, row) score, and score exactly (stored rows, fp32
                             * accumulation) only the pairs the bound cannot exclude: the results are those of the exact scan.
                             * The sketch is cut from T (x - mu): mu the mean of the shard's first fill, T random signs + a block
                             * Walsh-Hadamard transform -- anisotropic embeddings (outlier dimensions, a common component) keep pruning; shards whose rows
                             * collapse onto one direction (||mu||^2 >= 0.6 of the mean ||x||^2: a mean cosine of 0.6 between two rows) take a per-row form of the bound (vqa_index_get_sketch_split).
                             * Ignored for fp8 storage. */
#define VQA_INDEX_RESCORE_ROWS 4 /* with VQA_INDEX_SKETCH, where the sketch is kept: a second, ROW-MAJOR copy of the stored rows (+100 %
                             * of the rows' memory).  The sketch search scores its surviving (query, row) pairs exactly from the
                             * stored rows; in the scan's tiled layout a row is 2 d / 64 pieces of 64 bytes, 16 KiB apart (random
                             * half cache lines), in the copy one contiguous run -- the same values added in the same order, so
                             * the results are bit-equal with and without it.  Ignored where no sketch is kept, and when the device has no room for it. */

/* limits of the fused scoring + top-k kernel */
#define VQA_MAX_K 12       /* top-k per query found by ONE exact pass over the index (LDS candidate lists; BASELINE k = 10) */
#define VQA_MAX_K_TOTAL 1024 /* larger k: one pass keeping every workgroup's local top 12, verified on the device; when the
                              * check fails, ceil(k / 12) exact passes (each continuing strictly below the previous one) run
                              * behind a device-side gate */
#define VQA_QUERY_TILE 256 /* queries scored per pass over the index (BASELINE batch = 256) */

typedef struct vqa_index vqa_index;     /* opaque: one row shard of the corpus on one device */
typedef struct vqa_encoder vqa_encoder; /* opaque: question-encoder weights on one device */

int vqa_version(void);
const char* vqa_last_error(void);

/* ---- index: replaces txtai's ANN backend (faiss IndexFlatIP + IDMap) behind Embeddings.index/load -----------
 * heavy_ranker.py:86-94.  The index keeps its rows in HBM in a TILED layout (tiles of 256 rows x K-blocks of 64 bytes per
 * row, each a contiguous 16 KiB block in the exact bank-conflict-free LDS image of the scoring kernel), so a search
 * streams the shard as one sequential read.  Storage type `dtype`: VQA_F16 (fp16 MFMA), VQA_F32 (exact fp32 MFMA) or
 * VQA_FP8_E4M3 (OCP e4m3 MFMA; rows AND queries are stored as e4m3(16 * x) so that unit-vector components leave the
 * subnormal range, and returned scores are divided by 256 -- both factors are powers of two, hence exact).  vqa_index_create allocates a shard of n rows of length d and,
 * when `rows` is not NULL, fills it: rows [n, d] row-major, host or device pointer, element type rows_dtype
 * (VQA_F32 or VQA_F16; converted to the storage type with round-to-nearest-even, saturating for fp8), expected L2-normalised by the
 * caller (txtai normalises at index time).  ids: [n] int64 external ids (host or device) or NULL, in which case
 * id = id_base + row position (sqlite AUTOINCREMENT rowids start at 1, setup_db.py:14).
 * vqa_index_set_rows (faiss `add_with_ids` counterpart) fills rows [first, first + count) later, chunk by chunk, so
 * a 15-123 GB shard never needs a second full copy; it synchronises before returning. */
int vqa_index_create(vqa_index** out, int device, int64_t n, int32_t d, int32_t dtype, const void* rows, int32_t rows_dtype,
                     const int64_t* ids_or_null, int64_t id_base, uint32_t flags);

/* Every knob of an index handle, explicit and versioned.  The library reads NO environment variable (rounds 1-4 read 22 VQA_* at
 * create time; a `-DVQA_DEV` build -- scripts/, never the product -- still overlays them on these defaults for A/B runs).
 * vqa_index_options_init fills in the defaults [in brackets]; a caller changes what it wants and passes the struct to
 * vqa_index_create_ex.  `struct_size` = sizeof of the struct the caller was compiled against: fields a newer library knows beyond
 * it keep their defaults.  The production plan is what the defaults give (csrc/capi.hip kPlan: one table, every constant with the
 * probe that fitted it); the other values exist for tests that must reach a plan at a small size and for A/B measurements.
 * Nothing but speed depends on any of them: every plan returns the exact top-k of the stored values. */
typedef struct vqa_index_options {
    uint32_t struct_size;         /* sizeof(vqa_index_options) */
    uint32_t flags;               /* VQA_INDEX_* [0] */
    int32_t two_pass;             /* [1] seed pass (sub-maxima of the first tiles -> starting thresholds) in front of the main pass */
    int32_t wide_k;               /* [1] k > VQA_MAX_K: one verified pass before the ceil(k / 12) exact passes */
    int32_t seed_mult;            /* [2] seed pass covers seed_mult x CUs tiles (1..4) ... */
    int32_t seed_div;             /* [16] ... but at most 1 / seed_div of the shard's tiles (0: no cap) */
    int32_t stage_min_tiles;      /* [-1: by type -- 24 tiles per compute unit, 16 for an fp16 and 8 for an fp32 shard with a sketch] shard size
                                   * from which the two-stage / sketch plan applies; 0: never.  Given explicitly (tests at small sizes), the
                                   * profitability pause is off unless sketch_profit is given too */
    int32_t stage_pct;            /* [10] share of the first stage, percent (1..50) */
    int32_t f16_loop;             /* [0] fp16 K loop: 0 anti-phase slots, 1 K-step pairs, 2 pairs for the main launch of a two-stage search */
    int32_t sketch_cascade;       /* [1] cascade of bounds; 0: round 3's form with an exact first stage (k <= 12) */
    int32_t sketch_rotate;        /* [1] sketch cut from sign-flipped, block-Hadamard-rotated rows */
    int32_t sketch_center;        /* [1] ... of centred rows (needs the rotation) */
    int32_t sketch_split;         /* [1] slack term of the bound split along the centre direction */
    int32_t sketch_per_row;       /* [-1: auto, ||mu||^2 >= 0.6 mean ||x||^2 at the first fill] 0 / 1 force the per-row form off / on */
    int32_t sketch_ring_stages;   /* [5] X-ring stages of the int8 scan (6: measured equal) */
    int32_t sketch_mid_k;         /* [16] a second cascade stage from this k on (0: never) ... */
    int32_t sketch_mid_min_tiles; /* [128] ... or from this many tiles per workgroup on, whatever k (0: by k only) */
    int32_t sketch_mid_pct;       /* [200] its size, percent of the first stage */
    int32_t sketch_pre_k;         /* [48] the first stage's leading quarter as a stage of its own from this k on (0: never) */
    int32_t sketch_cooldown;      /* [64] base length of the pause after an overflow, in searches (0: no pause) */
    float sketch_profit;          /* [-1: by type -- 0.75 fp16, 4 fp32] factor f of the profitability rule (pause when a query tile scores
                                   * more than f n - 4e5 pairs exactly); 0: never */
    int32_t rescore_copy;         /* [-1] with VQA_INDEX_RESCORE_ROWS: take the copy only if an eighth of the device's memory stays free
                                   * behind it; 1: ask for the copy (sets the flag) and take it whenever the allocation succeeds;
                                   * 0: no copy, whatever the flag says */
    int32_t poison_workspace;     /* [-1: no] 0..255: fill every workspace with this byte at create (tests: nothing may read what no launch wrote) */
    int32_t one_launch;           /* [1] vqa_index_search_host on an fp16 / fp32 shard of <= 262 144 rows with <= 16 questions, k <= 32 and questions x k <= 64 runs the whole
                                   * search -- normalise, score, select, merge -- as ONE kernel (csrc/tiny_search.hip; the reference's call:
                                   * one question, limit 1, a few thousand documents); 0: the general launches (same bits; A/B switch) */
    int32_t sketch_regq;          /* [1] int8 sketch scans of rows of 768 / 384 elements run the register-resident-query kernel (csrc/scan_regq.hip:
                                   * the query tile is loaded once per launch instead of once per corpus tile); 0: score_topk.hip's slot loop
                                   * (same candidate pairs, same results; A/B switch) */
    int32_t final_rescore;        /* [1] fp16 / fp32 shards of the size at which a sketch search exists for the type (16 / 8 tiles of 256 rows per
                                   * compute unit: 1.05M / 0.52M rows; or stage_min_tiles when given): every EXACT path (sketch paused or off,
                                   * overflow fallback) ends by scoring its rows in the sketch path's re-scoring arithmetic and re-ranking them,
                                   * so both paths return the same bits (k <= 10: over the scan's k + 2 best rows; rows that tie with the k-th
                                   * beyond those two can still differ); 0: the MFMA sums as they are */
} vqa_index_options;
void vqa_index_options_init(vqa_index_options* opt);
/* vqa_index_create with explicit options (`flags` travel inside them); opt == NULL: the defaults */
int vqa_index_create_ex(vqa_index** out, int device, int64_t n, int32_t d, int32_t dtype, const void* rows, int32_t rows_dtype,
                        const int64_t* ids_or_null, int64_t id_base, const vqa_index_options* opt);
int vqa_index_set_rows(vqa_index* index, int64_t first, int64_t count, const void* rows, int32_t rows_dtype,
                       const int64_t* ids_or_null);
/* export (Embeddings.save, heavy_ranker.py:87): rows [first, first + count) back as row-major [count, d] in the STORAGE
 * type (fp16 / fp32 values, or e4m3 codes of 16 * x) into a host or device buffer and, optionally, their ids;
 * synchronises before returning. */
int vqa_index_get_rows(vqa_index* index, int64_t first, int64_t count, void* out_rows, int64_t* out_ids_or_null);
void vqa_index_destroy(vqa_index* index);
int64_t vqa_index_size(const vqa_index* index);
int32_t vqa_index_dim(const vqa_index* index);
int32_t vqa_index_dtype(const vqa_index* index);
/* device memory the shard holds: rows + id vector + sketch + re-scoring copy (workspaces of a few MB not counted); -1: null */
int64_t vqa_index_device_bytes(const vqa_index* index);
/* -1: the shard keeps no sketch; 0: its searches take the sketch search; n > 0: a sketch search overflowed into its exact fallback
 * (data the bound cannot prune) and about the next n searches of this handle run the exact scan (call it after the searches completed).
 *
 * The stateful part of a sketch shard, in full.  A sketch search whose candidate buffers fill up (any query tile of the call)
 * still returns the exact result -- its exact fallback runs behind it on the device, gated on a flag, no host round trip -- but
 * costs a sketch scan AND an exact scan.  The flags of a call (+ the call's number) are copied to a pinned mirror behind its
 * last launch; a LATER vqa_index_search on the handle that finds the report of a call it has not seen yet, with a flag up, starts
 * a pause: the next `sketch_cooldown` (vqa_index_options, default 64) searches run the exact scan only, then the sketch is tried again; every
 * further overflow doubles the pause (up to 64 x), a sketch search that stands resets it.  A search that does not overflow but
 * scores more pairs exactly in one query tile than the scan saves -- more than 0.75 n - 4e5 pairs (fp32 shards: 4 n;
 * vqa_index_options.sketch_profit sets the factor) -- is treated the same way (dense clusters; a large k on a shard of a million rows): its
 * result stands, the following searches of at least HALF its k take the exact scan (a handle that serves k = 100 and k = 10
 * searches alternately on a 1M-row shard keeps the k = 10 ones on the sketch; a pause started by an overflow applies to every k).  The host may run many searches ahead
 * of the device: reports of calls queued before a pause began are ignored, so the reaction lags by the queue depth and never
 * compounds.  Nothing but SPEED depends on this state: every path returns the exact top-k of the stored values.  Scores are
 * bit-stable within one path; between the sketch path (fp32 fma chain of the re-scoring kernel) and the exact scan (MFMA
 * accumulation) the last bits of a score may differ (<= 3e-7 on unit vectors at d = 768), and with them the order of rows
 * whose scores tie to within that.
 * Under hipGraph capture the host-side choice is frozen: a captured search replays the launch sequence chosen at capture
 * time -- captured outside a pause: the sketch search with its gated fallback (an overflow inside a replay takes the fallback,
 * correct results, and no pause ever starts because no host code runs); captured inside a pause: the exact scan. */
int32_t vqa_index_sketch_state(const vqa_index* index);
/* Pause the sketch of this handle for (at least) its next `searches` searches (they take the exact scan; 0 ends a pause at once).  What the
 * handle does by itself after an overflow, at the caller's request: for data the caller knows the bound cannot prune, and for tests /
 * benchmarks that want the exact path of the SAME handle (since 0.1.13 both paths return the same bits on the shards that have both:
 * vqa_index_options.final_rescore).  VQA_EINVAL for a shard without a sketch.  Not thread-safe against a concurrent search. */
int vqa_index_sketch_pause(vqa_index* index, int32_t searches);
/* Diagnostics of the sketch search; both SYNCHRONISE the device (never call them between the searches of a timed loop).
 * vqa_index_sketch_stats: what the last search of this handle left in its candidate buffers -- out[8] = { candidate pairs of the last
 * sketch scan (of a cascade: its second, main scan) of the call's last query tile, the fullest workgroup region of that scan, pairs
 * scored exactly for that query tile (all its scans), its longest candidate sub-list, overflow flag of that tile, OR of the flags of
 * the call's earlier tiles, capacity of a region, capacity of a sub-list }.  bench.py derives a step's physical bytes from it.
 * vqa_index_get_sketch_tile: the int8 codes of one 256-row tile as host [256, d8] row-major (d8 = d rounded up to 128), its
 * (max ||x_hi||, max ||x_lo||, 1 / scale, scale) and, optionally, the shard's centre mu [d8] (zeros when the sketch is not centred):
 * what the bound of the sketch search is computed from, for tests that restate it independently.
 * vqa_index_get_sketch_split: the tile's max |w . x_lo| and, optionally, w [d8] -- the rotated, normalised centre along which the
 * bound's slack term |z . x_lo| is split into |alpha| |w . x_lo| + ||z - alpha w|| ||x_lo|| (zeros when the shard does not split).
 * Shards whose rows collapse onto the centre direction (||mu|| >= 0.85 at the first fill: an untrained or strongly anisotropic
 * encoder) take the PER-ROW form instead (*out_per_row = 1): rows and queries are projected off w before they are sketched, every row
 * keeps beta = w . y (out_beta: the tile's 256 values; out_c: its max |beta|) and the scan adds alpha beta per (query, row). */
int vqa_index_sketch_stats(vqa_index* index, int64_t* out /* [8] */);
int vqa_index_get_sketch_tile(vqa_index* index, int64_t tile, int8_t* out_codes, float* out_info /* [4] */, float* out_mu_or_null);
int vqa_index_get_sketch_split(vqa_index* index, int64_t tile, float* out_c /* [1] */, float* out_w_or_null /* [d8] */,
                               float* out_beta_or_null /* [256] */, int32_t* out_per_row_or_null);

/* ---- search: replaces the scoring + top-k inside Embeddings.search / batchsearch (heavy_ranker.py:98,100) ---
 * q: [B, d] DEVICE pointer, element type q_dtype (VQA_F32 or VQA_F16; converted to the index storage type with
 * round-to-nearest-even inside the call).  Scores are fp32 inner products of the stored values accumulated in fp32.
 * out_scores [B, k] float and out_ids [B, k] int64 (device): best first; ties by row position ascending; when
 * the shard holds fewer than k rows the tail is padded with (-inf, -1).  out_pos_or_null [B, k] int64 receives
 * the row positions inside this shard (or NULL).  1 <= k <= VQA_MAX_K_TOTAL (k <= VQA_MAX_K: one pass over the index;
 * beyond that one verified pass, with ceil(k / VQA_MAX_K) gated exact passes as the fallback); any B >= 1 (processed in
 * ceil(B / VQA_QUERY_TILE) query tiles of equal size: B = 257 as 129 + 128; a query's result does not depend on its tile).
 * Asynchronous: the launches are queued on hip_stream and the call returns.  A handle owns ONE set of workspaces (query
 * staging, candidate lists, the sketch search's buffers): its searches must be ordered on the device -- the same stream, or
 * streams joined by events -- and one host thread at a time may be inside a call on it (a second one gets VQA_EINVAL).
 * (A scan already fills the device, so concurrent searches of one shard would gain nothing; batches are pipelined by
 * queueing them one behind the other.) */
int vqa_index_search(vqa_index* index, const void* q, int32_t q_dtype, int32_t B, int32_t k, float* out_scores,
                     int64_t* out_ids, int64_t* out_pos_or_null, void* hip_stream);

/* The latency form of vqa_index_search for the reference's own calling pattern -- one question per call, limit = 1
 * (heavy_ranker.py:97-101): HOST pointers in and out, synchronous.  q_host [B, d] fp32 / fp16 host memory; normalize = 1 (fp32
 * queries only) L2-normalises them on the device with the kernel of vqa_normalize_convert, so the results are those of
 * normalise-then-vqa_index_search bit for bit.  out_scores [B, k], out_ids [B, k] and out_pos_or_null [B, k] are host arrays,
 * valid when the call returns.  No allocation and no copy operation per call: the CPU copies the queries into a pinned,
 * device-mapped buffer the handle keeps (grown on demand), the kernels read them from there and write the results into pinned
 * memory, and the call polls the stream for completion (up to ~0.2 ms, then blocks).  The launches go to hip_stream: order it
 * with the handle's other searches as for vqa_index_search. */
/* (q_host may also be a DEVICE pointer -- the question encoder's output: vqa_encoder_forward_host + this call are the whole text
 * query of heavy_ranker.py:98; only the results then travel through the pinned buffer.) */
int vqa_index_search_host(vqa_index* index, const void* q_host, int32_t q_dtype, int32_t B, int32_t k, int32_t normalize,
                          float* out_scores, int64_t* out_ids, int64_t* out_pos_or_null, void* hip_stream);

/* ---- merge: final step after the RCCL all-gather of per-shard candidates (new in this build; the reference is
 * single-process).  scores/ids: R blocks of [B, k] on the device, each [k] list best first, padded with (-inf, -1);
 * block r starts at scores + r * score_rank_stride / ids + r * id_rank_stride (strides in ELEMENTS; 0 = B * k,
 * i.e. plain [R, B, k] arrays) -- the strides let both arrays live in ONE all-gathered buffer (one collective per
 * batch).  Shards are contiguous row ranges in rank order, so ties resolve by (rank asc, slot asc) = global row
 * position asc.  out: [B, k_out] with k_out <= R * k <= 8192. */
int vqa_merge_topk(const float* scores, const int64_t* ids, int64_t score_rank_stride, int64_t id_rank_stride, int32_t R,
                   int32_t B, int32_t k, int32_t k_out, float* out_scores, int64_t* out_ids, void* hip_stream);

/* ---- measurement hooks used by bench.py (roofline of the dominant kernel).  vqa_index_launch_info reports the
 * geometry of the main scoring kernel so the algorithmic bytes/flops per launch can be stated.  With timing
 * enabled, every vqa_index_search brackets its MAIN scoring-kernel launch with HIP events on the caller's stream
 * (events are not capturable: leave timing off under hipGraph capture); vqa_index_get_timing waits for the last
 * event, returns the summed kernel time and the number of launches since the previous call, and resets both. */
typedef struct vqa_launch_info {
    int32_t grid;          /* workgroups of the fused scoring kernel */
    int32_t block;         /* threads per workgroup */
    int32_t lds_bytes;     /* dynamic LDS per workgroup */
    int32_t rows_per_tile; /* corpus rows scored per workgroup iteration */
    int64_t rows_per_launch;  /* corpus rows the MAIN scoring kernel launch covers: n - first_stage_rows (the seeding pass scores
                               * its seed_tiles * rows_per_tile rows once more, in a launch of its own) */
    int64_t bytes_per_launch; /* algorithmic bytes of that launch: rows_per_launch * d * sizeof(element) */
    int64_t flops_per_launch; /* 2 * VQA_QUERY_TILE * rows_per_launch * d */
    int32_t seed_grid;        /* workgroups of the seeding pass (a few tiles each), 0 when the search is single pass */
    int32_t seed_tiles;       /* tiles of rows_per_tile rows the seeding pass scores */
    int64_t first_stage_rows; /* large shards, k <= VQA_MAX_K: rows scored by the first-stage launch of the same kernel (its exact
                               * k-th best scores seed the main launch's thresholds); 0 = the main launch covers every row */
    int32_t sketch_scan;      /* 1: the main launch scans the int8 sketch (VQA_INDEX_SKETCH): bytes_per_launch counts one byte per
                               * element, flops_per_launch the same multiply-adds (int8 x int8 -> int32) */
    int32_t levels;           /* scans of disjoint row ranges one pass consists of: 1 = one launch over every row; 2 = first stage + main
                               * launch; sketch cascade: 2 .. 4 (leading quarter | first stage | second stage | the rest) */
    int32_t scan_kernel;      /* which kernel the sketch scans run: 0 = csrc/score_topk.hip (MODE 2, the slot loop; also every exact scan),
                               * 1 = csrc/scan_regq.hip (query tile resident in registers: rows of 768 / 384 sketch elements, no per-row form) */
} vqa_launch_info;
int vqa_index_launch_info(const vqa_index* index, int32_t B, int32_t k, vqa_launch_info* out);
int vqa_index_set_timing(vqa_index* index, int32_t enabled); /* 0: off (recorded pairs are kept until get_timing), 1: on, from scratch,
                                                               * 2: on, keeping the pairs recorded so far (an event record between two
                                                               * launches costs the stream a few microseconds: bench.py samples) */
int vqa_index_get_timing(vqa_index* index, double* kernel_ms_sum, int64_t* launches);
/* Measurement helper (no search path uses it): GB/s of a read-only stream over `bytes` of device memory with the scan kernels' own loads
 * (LDS-DMA nt into an LDS ring, one persistent workgroup per compute unit) and nothing else -- the ceiling of THIS device the scans'
 * bandwidth fractions can be read against (csrc/diag.hip; fastest of `reps` launches behind two warm-ups; synchronises the stream). */
int vqa_measure_read_stream(const void* device_buffer, int64_t bytes, int32_t reps, double* out_gbs, void* hip_stream);

/* ---- question encoder: replaces the transformer forward + pooling + L2-normalise inside txtai
 * (model chosen by `path=` at heavy_ranker.py:80,83; DPR form at src/test.py:84-86 `.pooler_output`).
 * BERT-family post-LN encoder: RoBERTa / PhoBERT / XLM-R (paraphrase-multilingual-mpnet-base-v2, heavy_ranker.py:83: hidden 768,
 * 12 heads of 64, vocab 250 002, 514 positions) and BERT (paraphrase-multilingual-MiniLM-L12-v2, heavy_ranker.py:80: hidden 384,
 * 12 heads of 32, FFN 1536, absolute position ids); head sizes 64 and 32 run the matrix-core attention kernel.  All weight pointers are fp32, host or device, copied (and
 * converted to fp16) at create time. */
#define VQA_POS_ROBERTA 0  /* position id = pad_id + number of non-pad tokens up to and including this one (RoBERTa, XLM-R, PhoBERT) */
#define VQA_POS_ABSOLUTE 1 /* position id = index of the token in its sequence (BERT: paraphrase-multilingual-MiniLM-L12-v2, heavy_ranker.py:80) */
typedef struct vqa_encoder_config {
    int32_t vocab_size, hidden, layers, heads, ffn, max_pos, type_vocab, pad_id;
    float ln_eps;
    int32_t position_ids; /* VQA_POS_*; token_type ids are 0 for every token either way (row 0 of the type table) */
} vqa_encoder_config;

typedef struct vqa_encoder_layer_weights {
    const float *wq, *bq, *wk, *bk, *wv, *bv; /* [hidden, hidden] row-major [out, in] (torch Linear), [hidden] */
    const float *wo, *bo;                     /* attention output projection */
    const float *ln1_g, *ln1_b;               /* LayerNorm after attention residual */
    const float *w1, *b1;                     /* [ffn, hidden], [ffn] */
    const float *w2, *b2;                     /* [hidden, ffn], [hidden] */
    const float *ln2_g, *ln2_b;               /* LayerNorm after FFN residual */
} vqa_encoder_layer_weights;

typedef struct vqa_encoder_weights {
    const float *word_emb, *pos_emb, *type_emb; /* [vocab, hidden], [max_pos, hidden], [type_vocab, hidden] */
    const float *emb_ln_g, *emb_ln_b;
    const vqa_encoder_layer_weights* layer;     /* [layers] */
} vqa_encoder_weights;

#define VQA_POOL_CLS 0  /* DPR: last_hidden[:, 0, :] (transformers modeling_dpr.py DPREncoder.forward) */
#define VQA_POOL_MEAN 1 /* sentence-transformers masked mean (the models heavy_ranker.py:80,83 actually load) */

int vqa_encoder_create(vqa_encoder** out, int device, const vqa_encoder_config* cfg, const vqa_encoder_weights* w,
                       int32_t max_tokens /* B*L capacity of the activation workspace */);
/* the encoder's switches, as vqa_index_options: explicit, versioned, defaults in brackets; no environment variable is read */
typedef struct vqa_encoder_options {
    uint32_t struct_size; /* sizeof(vqa_encoder_options) */
    int32_t fold_layernorm; /* [1] calls of > 320 tokens carry raw rows + row statistics and apply the LayerNorms inside the GEMMs; 0: LayerNorm kernels */
    int32_t first_rows;     /* [1] CLS pooling of a large batch: the last layer past its attention on the first-token rows only */
    int32_t graphs;         /* [1] calls of <= 1024 positions replay a captured hipGraph */
    int32_t latency_path;   /* [1] calls of <= 64 positions (one question: heavy_ranker.py:97-101) on models whose hidden and FFN sizes are
                             * multiples of 384 (the reference's two models: hidden 384 / 768) run the latency form: five launches per layer, LayerNorms inside the GEMMs; 0: the general small-batch kernels */
    int32_t persistent;     /* [0] 1: the two model shapes of the reference (hidden 768 / FFN 3072 and hidden 384 / FFN 1536, 12 heads) run a call of <= 64
                             * positions as ONE cooperative launch: every phase of the latency form inside a persistent kernel, fence-free grid barriers
                             * between the phases, activations in uncached device memory read with device-scope loads (csrc/encoder.hip
                             * encoder_persist_kernel).  The same bits as the launches -- and, measured, 1.04 ms against their 0.44 (every workgroup of
                             * every phase pulls the activations through the memory side instead of its XCD's L2: profiles/r06_persistent_forward.txt):
                             * kept as the measured answer to "would one launch be faster", off by default */
    int32_t persistent_grid; /* [0: FFN size / 32] resident workgroups of that kernel (A/B switch) */
} vqa_encoder_options;
void vqa_encoder_options_init(vqa_encoder_options* opt);
int vqa_encoder_create_ex(vqa_encoder** out, int device, const vqa_encoder_config* cfg, const vqa_encoder_weights* w, int32_t max_tokens,
                          const vqa_encoder_options* opt /* NULL: defaults */);
void vqa_encoder_destroy(vqa_encoder* enc);
/* input_ids, attn_mask: [B, L] int32 device.  out: [B, hidden] fp32 device.
 * real_tokens: 0 = unknown (every one of the B * L positions is computed), else the number of set mask entries of a
 * RIGHT-PADDED mask (mask[b][l] = 1 exactly for l < n_b, n_b >= 1): calls of more than 1024 positions then compute only those
 * rows (sequence packing; the results of the real tokens are identical, padding keys carry zero attention weight either
 * way).  A mask that is not right-padded, or whose number of set entries differs from real_tokens (more: rows would be
 * dropped; fewer: the GEMMs would run on stale workspace rows), invalidates that call's output and makes the NEXT
 * vqa_encoder_forward on the handle fail with VQA_EINVAL.
 * Token ids outside [0, vocab_size) never index the embedding table: they are embedded as pad_id and a host-visible flag
 * is raised, which makes the NEXT vqa_encoder_forward on the handle fail with VQA_EINVAL (the call that saw them cannot
 * report it without a synchronisation). */
int vqa_encoder_forward(vqa_encoder* enc, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L,
                        int32_t real_tokens, int32_t pooling, int32_t normalize, float* out, void* hip_stream);

/* The forward for HOST token ids / masks (what a tokenizer returns; one question per call in heavy_ranker.py:97-101): the CPU copies
 * them into a pinned, device-mapped staging buffer of the handle -- no torch tensor, no copy operation per call -- and the launches are
 * queued on hip_stream like vqa_encoder_forward's (asynchronous; out is a DEVICE pointer [B, hidden] fp32, typically handed to
 * vqa_index_search_host next).  A right-padded mask is recognised here and packs calls of more than 1024 positions; token ids outside
 * the table fail the call at once.  The staging buffer is reused: a second call waits (on the host) until the first one's launches
 * have consumed it.  Not capturable into a hipGraph. */
int vqa_encoder_forward_host(vqa_encoder* enc, const int32_t* input_ids_host, const int32_t* attn_mask_host, int32_t B, int32_t L,
                             int32_t pooling, int32_t normalize, float* out_dev, void* hip_stream);

/* Hidden states of the same forward (HF `output_hidden_states`: transformers BaseModelOutput.hidden_states[n_layers], what
 * `q_model(input_ids)` of src/test.py:84-86 carries beside .pooler_output): the embedding output (n_layers = 0) or the output of layer
 * n_layers (1 .. layers; n_layers = layers: last_hidden_state), every position, LayerNorm applied, as fp32 [B, L, hidden] on the device.
 * Runs the kernels, tile shapes and folded LayerNorms a vqa_encoder_forward of this (B, L, real_tokens) runs, eagerly (no graph).
 * real_tokens as in vqa_encoder_forward; in the packed form padding positions are not computed and come back as zeros.  This is how the
 * parity tests hold the HIP layers -- not only the pooled vector -- to the HF goldens (tests/test_gpu_encoder.py). */
int vqa_encoder_forward_hidden(vqa_encoder* enc, const int32_t* input_ids, const int32_t* attn_mask, int32_t B, int32_t L,
                               int32_t real_tokens, int32_t n_layers, float* out_hidden, void* hip_stream);

/* ---- helpers used by the host side when it builds an index from fp32 embeddings ------------------------------
 * rows fp32 [n, d] device -> L2-normalised (optional) -> element type dtype, written to out (device). */
int vqa_normalize_convert(const float* rows, int64_t n, int32_t d, int32_t normalize, int32_t dtype, void* out,
                          void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VQA_RETRIEVAL_H */
