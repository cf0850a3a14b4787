// Host-side sanitizer run of the C ABI's error and cleanup paths (no GPU needed: on a box without a device every *_create
// returns VQA_ENODEV after its argument checks; with one, the create / destroy cycles run too).  Built by
// tests/test_asan_host.py with -fsanitize=address,undefined -fno-gpu-sanitize from the same sources as the product library
// (host code instrumented, device code as usual).  Exit code 0 = every expectation held and the sanitizers stayed silent.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "vqa_retrieval.h"

static int failures = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "%s:%d: %s FAILED (last error: %s)\n", __FILE__, __LINE__, #cond, vqa_last_error()); \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

int main() {
    EXPECT(vqa_version() == VQA_VERSION);
    EXPECT(vqa_last_error() != nullptr);
    // ---- index: argument validation happens before any device work
    vqa_index* ix = reinterpret_cast<vqa_index*>(0x1);
    EXPECT(vqa_index_create(nullptr, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, 0) == VQA_EINVAL);
    EXPECT(vqa_index_create(&ix, 0, -1, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, 0) == VQA_EINVAL && ix == nullptr);
    EXPECT(strstr(vqa_last_error(), "n=-1") != nullptr);
    EXPECT(vqa_index_create(&ix, 0, 10, 0, VQA_F16, nullptr, VQA_F16, nullptr, 0, 0) == VQA_EINVAL);
    EXPECT(vqa_index_create(&ix, 0, 10, 8, 7, nullptr, VQA_F16, nullptr, 0, 0) == VQA_EINVAL);
    EXPECT(vqa_index_create(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, 0xF0) == VQA_EINVAL);
    EXPECT(vqa_index_create(&ix, 0, (int64_t)1 << 33, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, 0) == VQA_EINVAL);
    // explicit options (vqa_index_create_ex): defaults, versioning by struct_size, range checks -- all before any device work
    {
        vqa_index_options o;
        memset(&o, 0xAB, sizeof(o));
        vqa_index_options_init(&o);
        EXPECT(o.struct_size == sizeof(o) && o.flags == 0 && o.stage_min_tiles == -1 && o.stage_pct == 10 && o.sketch_cooldown == 64 &&
               o.sketch_profit < 0.f && o.poison_workspace == -1 && o.rescore_copy == -1 && o.sketch_per_row == -1 && o.seed_mult == 2 && o.one_launch == 1);
        vqa_index_options_init(nullptr);
        vqa_index_options bad_o = o;
        bad_o.struct_size = 0;
        EXPECT(vqa_index_create_ex(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, &bad_o) == VQA_EINVAL && ix == nullptr);
        EXPECT(strstr(vqa_last_error(), "struct_size") != nullptr);
        bad_o = o;
        bad_o.stage_pct = 99;
        EXPECT(vqa_index_create_ex(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, &bad_o) == VQA_EINVAL);
        bad_o = o;
        bad_o.seed_mult = 0;
        EXPECT(vqa_index_create_ex(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, &bad_o) == VQA_EINVAL);
        bad_o = o;
        bad_o.poison_workspace = 300;
        EXPECT(vqa_index_create_ex(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, &bad_o) == VQA_EINVAL);
        bad_o = o;
        bad_o.flags = 0xF0;
        EXPECT(vqa_index_create_ex(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, &bad_o) == VQA_EINVAL);
        // an OLDER caller's shorter struct: the fields it does not know keep their defaults (only the prefix is read)
        vqa_index_options shorter = o;
        shorter.struct_size = 16;
        shorter.poison_workspace = 300;  // beyond the prefix: must not be looked at
        const int rc_short = vqa_index_create_ex(&ix, 0, 10, 8, VQA_F16, nullptr, VQA_F16, nullptr, 0, &shorter);
        EXPECT(rc_short == VQA_OK || rc_short == VQA_ENODEV || rc_short == VQA_EHIP);
        if (rc_short == VQA_OK) vqa_index_destroy(ix);
        vqa_encoder_options eo;
        vqa_encoder_options_init(&eo);
        EXPECT(eo.struct_size == sizeof(eo) && eo.fold_layernorm == 1 && eo.first_rows == 1 && eo.graphs == 1 && eo.latency_path == 1);
    }
    int rc = vqa_index_create(&ix, 0, 1000, 64, VQA_F16, nullptr, VQA_F16, nullptr, 1, 0);
    if (rc == VQA_OK) {  // a device is visible: create / use-after-bad-args / destroy cycles
        for (int i = 0; i < 3; ++i) {
            EXPECT(vqa_index_size(ix) == 1000 && vqa_index_dim(ix) == 64 && vqa_index_dtype(ix) == VQA_F16);
            std::vector<float> rows(10 * 64, 0.5f);
            EXPECT(vqa_index_set_rows(ix, 995, 10, rows.data(), VQA_F32, nullptr) == VQA_EINVAL);  // past the end
            EXPECT(vqa_index_set_rows(ix, 0, 10, rows.data(), 9, nullptr) == VQA_EINVAL);
            EXPECT(vqa_index_set_rows(ix, 0, 10, rows.data(), VQA_F32, nullptr) == VQA_OK);
            vqa_launch_info info;
            EXPECT(vqa_index_launch_info(ix, 4, 10, &info) == VQA_OK && info.rows_per_launch == 1000);
            vqa_index_destroy(ix);
            EXPECT(vqa_index_create(&ix, 0, 1000, 64, i % 2 ? VQA_F32 : VQA_FP8_E4M3, nullptr, VQA_F16, nullptr, 1, VQA_INDEX_HAS_IDS) == VQA_OK);
        }
        vqa_index_destroy(ix);
    } else {
        EXPECT(rc == VQA_ENODEV || rc == VQA_EHIP);
        EXPECT(ix == nullptr);
    }
    vqa_index_destroy(nullptr);
    EXPECT(vqa_index_size(nullptr) == -1 && vqa_index_dim(nullptr) == -1 && vqa_index_dtype(nullptr) == -1);
    float s[4];
    int64_t ids[4];
    EXPECT(vqa_index_search(nullptr, s, VQA_F32, 1, 1, s, ids, nullptr, nullptr) == VQA_EINVAL);
    EXPECT(vqa_index_set_rows(nullptr, 0, 1, s, VQA_F32, nullptr) == VQA_EINVAL);
    EXPECT(vqa_index_get_rows(nullptr, 0, 1, s, nullptr) == VQA_EINVAL);
    EXPECT(vqa_index_launch_info(nullptr, 1, 1, nullptr) == VQA_EINVAL);
    {
        int64_t st[8];
        int8_t codes[16];
        float info4[4];
        EXPECT(vqa_index_sketch_stats(nullptr, st) == VQA_EINVAL);
        EXPECT(vqa_index_get_sketch_tile(nullptr, 0, codes, info4, nullptr) == VQA_EINVAL);
        EXPECT(vqa_index_get_sketch_split(nullptr, 0, info4, nullptr, nullptr, nullptr) == VQA_EINVAL);
        EXPECT(vqa_index_sketch_state(nullptr) == -1 && vqa_index_device_bytes(nullptr) == -1);
    }
    EXPECT(vqa_index_set_timing(nullptr, 1) == VQA_EINVAL);
    double ms;
    int64_t n;
    EXPECT(vqa_index_get_timing(nullptr, &ms, &n) == VQA_EINVAL);
    // ---- merge: shape checks
    EXPECT(vqa_merge_topk(nullptr, ids, 0, 0, 2, 1, 2, 2, s, ids, nullptr) == VQA_EINVAL);
    EXPECT(vqa_merge_topk(s, ids, 0, 0, 0, 1, 2, 2, s, ids, nullptr) == VQA_EINVAL);
    EXPECT(vqa_merge_topk(s, ids, 0, 0, 8, 1, 2048, 10, s, ids, nullptr) == VQA_EINVAL);  // R * k > 8192
    EXPECT(vqa_merge_topk(s, ids, 0, 0, 2, 1, 2, 5, s, ids, nullptr) == VQA_EINVAL);      // k_out > R * k
    EXPECT(vqa_merge_topk(s, ids, 1, 1, 2, 1, 2, 2, s, ids, nullptr) == VQA_EINVAL);      // strides below one block
    // ---- encoder: config validation
    vqa_encoder* enc = reinterpret_cast<vqa_encoder*>(0x1);
    vqa_encoder_config cfg = {100, 64, 1, 4, 128, 40, 1, 1, 1e-5f};
    vqa_encoder_layer_weights lw;
    memset(&lw, 0, sizeof(lw));
    vqa_encoder_weights w;
    memset(&w, 0, sizeof(w));
    w.layer = &lw;
    EXPECT(vqa_encoder_create(nullptr, 0, &cfg, &w, 64) == VQA_EINVAL);
    EXPECT(vqa_encoder_create(&enc, 0, nullptr, &w, 64) == VQA_EINVAL && enc == nullptr);
    vqa_encoder_config bad = cfg;
    bad.hidden = 63;
    EXPECT(vqa_encoder_create(&enc, 0, &bad, &w, 64) == VQA_EINVAL);
    bad = cfg;
    bad.heads = 5;
    EXPECT(vqa_encoder_create(&enc, 0, &bad, &w, 64) == VQA_EINVAL);
    bad = cfg;
    bad.pad_id = 100;
    EXPECT(vqa_encoder_create(&enc, 0, &bad, &w, 64) == VQA_EINVAL);
    rc = vqa_encoder_create(&enc, 0, &cfg, &w, 64);  // null weight pointers: refused, everything allocated so far is released
    EXPECT(rc != VQA_OK && enc == nullptr);
    vqa_encoder_destroy(nullptr);
    int32_t tok[4] = {0, 1, 2, 3};
    EXPECT(vqa_encoder_forward(nullptr, tok, tok, 1, 4, 0, VQA_POOL_CLS, 1, s, nullptr) == VQA_EINVAL);
    EXPECT(vqa_encoder_forward_host(nullptr, tok, tok, 1, 4, VQA_POOL_CLS, 1, s, nullptr) == VQA_EINVAL);
    EXPECT(vqa_encoder_forward_hidden(nullptr, tok, tok, 1, 4, 0, 1, s, nullptr) == VQA_EINVAL);
    EXPECT(vqa_index_search_host(nullptr, s, VQA_F32, 1, 1, 0, s, ids, nullptr, nullptr) == VQA_EINVAL);
    EXPECT(vqa_normalize_convert(nullptr, 1, 4, 1, VQA_F16, s, nullptr) == VQA_EINVAL);
    EXPECT(vqa_normalize_convert(s, -1, 4, 1, VQA_F16, s, nullptr) == VQA_EINVAL);
    EXPECT(vqa_normalize_convert(s, 1, 4, 1, 9, s, nullptr) == VQA_EINVAL);
    if (failures) fprintf(stderr, "%d expectation(s) failed\n", failures);
    return failures ? 1 : 0;
}
