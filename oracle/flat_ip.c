/* CPU oracle, C restatement of the scoring + top-k step (TEST INFRASTRUCTURE -- NOT PRODUCT CODE).
 *
 * Restates what the reference's retriever call `embeddings.search(query, limit)`
 * (inference_pipeline/db_utils/heavy_ranker.py:98-101) computes below the txtai API: faiss `IndexFlatIP`
 * semantics = exact fp32 inner product of each query with every stored row, k largest per query kept in a
 * binary heap, results best first.  faiss is an un-vendored transitive dependency of the unpinned `txtai`
 * (requirements.txt:74); this file follows its published flat-index algorithm, it copies no source.
 * Tie order is defined here: score descending, then row position ascending.
 *
 * PARITY STATUS: parity unpinned by the reference (no tests/fixtures there); cross-checked in
 * tests/test_oracle.py against oracle/retrieval.py (numpy), torch and scikit-learn.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC oracle/flat_ip.c -o oracle/libflat_ip.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int64_t p; } cand_t;

/* "a is worse than b" in the (score desc, position asc) order */
static inline int worse(cand_t a, cand_t b) { return a.s < b.s || (a.s == b.s && a.p > b.p); }

/* min-heap on the order above: heap[0] is the worst kept candidate */
static void heap_sift_down(cand_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == i) return;
        cand_t t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
}

static void heap_push_or_replace(cand_t* h, int* n, int k, cand_t c) {
    if (*n < k) {
        int i = (*n)++;
        h[i] = c;
        while (i > 0) {
            int p = (i - 1) / 2;
            if (!worse(h[i], h[p])) break;
            cand_t t = h[i]; h[i] = h[p]; h[p] = t; i = p;
        }
    } else if (worse(h[0], c)) {
        h[0] = c;
        heap_sift_down(h, k, 0);
    }
}

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1F, m = h & 0x3FF, u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { /* subnormal */
            int sh = 0; while (!(m & 0x400)) { m <<= 1; ++sh; }
            m &= 0x3FF; u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

static float e4m3_to_float(uint8_t c) {
    int e = (c >> 3) & 0xF, m = c & 7; float v;
    if ((c & 0x7F) == 0x7F) return NAN;
    v = e == 0 ? ldexpf((float)m / 8.0f, -6) : ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return (c & 0x80) ? -v : v;
}

/* dtype: 0 f32, 1 f16 (uint16 bits), 2 fp8 e4m3fn codes.  q is always fp32 [b, d] (already holding the values the
 * GPU sees).  out_scores/out_pos are [b, k] (padded with -inf / -1 when n < k).  Returns 0. */
int flat_ip_search(const float* q, int64_t b, int64_t d, const void* x, int dtype, int64_t n, int k,
                   float* out_scores, int64_t* out_pos) {
    const int64_t BLK = 1024; /* rows decoded per block so the block stays in L2 */
    int64_t nblk = (n + BLK - 1) / BLK;
    float* xf = (float*)malloc(sizeof(float) * (size_t)BLK * (size_t)d);
    cand_t* heaps = (cand_t*)malloc(sizeof(cand_t) * (size_t)b * (size_t)k);
    int* hn = (int*)calloc((size_t)b, sizeof(int));
    if (!xf || !heaps || !hn) return -1;
    for (int64_t blk = 0; blk < nblk; ++blk) {
        int64_t r0 = blk * BLK, r1 = r0 + BLK < n ? r0 + BLK : n;
        for (int64_t r = r0; r < r1; ++r)
            for (int64_t j = 0; j < d; ++j) {
                float v;
                if (dtype == 0) v = ((const float*)x)[r * d + j];
                else if (dtype == 1) v = half_to_float(((const uint16_t*)x)[r * d + j]);
                else v = e4m3_to_float(((const uint8_t*)x)[r * d + j]);
                xf[(r - r0) * d + j] = v;
            }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < b; ++i) {
            const float* qi = q + i * d;
            cand_t* h = heaps + i * k;
            for (int64_t r = r0; r < r1; ++r) {
                const float* xr = xf + (r - r0) * d;
                float acc = 0.0f;
                for (int64_t j = 0; j < d; ++j) acc += qi[j] * xr[j];
                cand_t c = { acc, r };
                heap_push_or_replace(h, &hn[i], k, c);
            }
        }
    }
    for (int64_t i = 0; i < b; ++i) { /* heap -> best first */
        cand_t* h = heaps + i * k; int m = hn[i];
        for (int j = 0; j < k; ++j) { out_scores[i * k + j] = -INFINITY; out_pos[i * k + j] = -1; }
        for (int j = m - 1; j >= 0; --j) {
            out_scores[i * k + j] = h[0].s; out_pos[i * k + j] = h[0].p;
            h[0] = h[j]; heap_sift_down(h, j, 0);
        }
    }
    free(xf); free(heaps); free(hn);
    return 0;
}
