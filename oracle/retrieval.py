"""CPU oracle for the dense-retrieval hot path (TEST INFRASTRUCTURE -- NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``vietnamese_qa_system_amd``) never imports anything under ``oracle/`` and
fails loudly when the HIP library is missing.

What it restates
----------------
The reference delegates the whole path to ``txtai.Embeddings.search`` (call sites
``inference_pipeline/db_utils/heavy_ranker.py:91-101``); txtai (unpinned, ``requirements.txt:74``) in turn
delegates scoring/top-k to faiss ``IndexFlatIP``.  Neither library is in ``/root/reference`` nor installable
here, so this file restates their *published* algorithm:

* L2-normalise query and corpus rows in fp32 (txtai ``Embeddings.normalize``: ``e /= norm(e, axis=1)[:, None]``;
  cosine intent evidenced by the reference itself at ``src/test.py:104`` ``cosine_similarity``),
* exact fp32 inner product of every query with every stored row (faiss ``METRIC_INNER_PRODUCT``, flat index),
* the k largest per query, best first (faiss heap result order), ids mapped through the id vector
  (faiss ``IDMap``; external ids originate at ``heavy_ranker.py:76`` ``"id": row[0]`` = sqlite rowid,
  ``setup_db.py:14``),
* txtai's ``score > 0`` result filter is a host-side option (``min_score``), never part of scoring.

Tie order is *defined* here (faiss leaves it unspecified): score descending, then corpus row position
ascending.  For ``ids = arange(N) + base`` that equals "id ascending".

PARITY STATUS: **parity unpinned by the reference** -- the reference has no test, fixture or golden vector
for this path.  The oracle is instead pinned against two independent implementations
available in this container (``tests/golden/make_golden.py``): ``torch.nn.functional.cosine_similarity``
(the very call the reference makes at ``src/test.py:104``) and scikit-learn's brute-force cosine
``NearestNeighbors``.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------------------
# storage dtypes of the index (what the GPU path keeps in HBM); the oracle always scores the SAME stored
# values upcast to fp32, otherwise recall@10 = 1.0 is not well defined ("Same-dtype oracle").
# --------------------------------------------------------------------------------------------------
DTYPE_F32, DTYPE_F16, DTYPE_FP8_E4M3 = 0, 1, 2
E4M3_MAX = 448.0


def l2_normalize(x: np.ndarray) -> np.ndarray:
    """Row-wise ``x / ||x||_2`` in fp32 (txtai ``normalize``; zero rows are left as zeros)."""
    x = np.asarray(x, dtype=np.float32)
    n = np.linalg.norm(x, axis=-1, keepdims=True).astype(np.float32)
    n = np.where(n == 0, np.float32(1), n)
    return (x / n).astype(np.float32)


# ---- OCP fp8 e4m3fn (gfx950 uses OCP, not fnuz) -----------------------------------------------------

def _e4m3_table() -> np.ndarray:
    """All 256 e4m3fn code points decoded to fp32 (0x7f / 0xff are NaN)."""
    codes = np.arange(256, dtype=np.uint32)
    sign = np.where(codes & 0x80, -1.0, 1.0)
    e = (codes >> 3) & 0xF
    m = codes & 0x7
    val = np.where(e == 0, (m / 8.0) * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    val = sign * val
    val[(codes & 0x7F) == 0x7F] = np.nan
    return val.astype(np.float32)


_E4M3 = _e4m3_table()


def e4m3_decode(codes: np.ndarray) -> np.ndarray:
    return _E4M3[np.asarray(codes, dtype=np.uint8)]


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """fp32 -> e4m3fn code, round-to-nearest-even, saturating to +-448 (the conversion the GPU build
    uses when it quantises an index: ``v_cvt_pk_fp8_f32`` semantics with saturation)."""
    x = np.asarray(x, dtype=np.float32)
    sign = np.signbit(x)
    a = np.minimum(np.abs(x).astype(np.float64), E4M3_MAX)
    # spacing: subnormal/first binade 2^-9, otherwise 2^(floor(log2 a) - 3)
    with np.errstate(divide="ignore"):
        ex = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    ex = np.clip(ex, -6, 8)
    step = 2.0 ** (ex - 3)
    q = np.rint(a / step) * step  # np.rint = round half to even
    q = np.minimum(q, E4M3_MAX)
    # encode q exactly
    with np.errstate(divide="ignore"):
        e2 = np.floor(np.log2(np.where(q > 0, q, 1.0)))
    e2 = np.clip(e2, -6, 8)
    is_sub = q < 2.0 ** -6
    mant = np.where(is_sub, np.rint(q / 2.0 ** -9), np.rint((q / 2.0 ** e2 - 1.0) * 8.0))
    ebits = np.where(is_sub, 0, e2 + 7)
    code = (ebits.astype(np.uint32) << 3) | mant.astype(np.uint32)
    code = np.where(sign, code | 0x80, code)
    code = np.where(np.isnan(x), 0x7F, code)
    return code.astype(np.uint8)


def e4m3_encode_fast(x: np.ndarray) -> np.ndarray:
    """The same conversion as :func:`e4m3_encode`, written on the fp32 bit patterns (a handful of integer passes instead of
    float64 logarithms: ~20x faster; ``tests/test_oracle.py`` holds the two equal on every binade boundary and on random
    data).  Normal range: the code is the fp32 pattern re-biased by 120 exponent steps and rounded to nearest even on the 20
    dropped mantissa bits (the carry walks into the exponent by itself); below 2^-6: round(|x| * 2^9) sub-steps."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    a = u & np.uint32(0x7FFFFFFF)
    norm = ((a - np.uint32(120 << 23)) + np.uint32(0x7FFFF) + ((a >> np.uint32(20)) & np.uint32(1))) >> np.uint32(20)
    small = a < np.uint32(121 << 23)
    sub = np.rint(np.where(small, np.abs(x), np.float32(0)) * np.float32(512.0)).astype(np.uint32)  # exact scaling, half to even
    code = np.where(small, sub, np.minimum(norm, np.uint32(0x7E)))
    code |= (u >> np.uint32(24)) & np.uint32(0x80)
    return np.where(a > np.uint32(0x7F800000), np.uint32(0x7F), code).astype(np.uint8)  # NaN: one code, as e4m3_encode


def quantize_rows(x: np.ndarray, dtype: int) -> np.ndarray:
    """fp32 rows -> the values the index stores, returned in their storage representation
    (float32 / float16 arrays, uint8 e4m3 codes)."""
    x = np.asarray(x, dtype=np.float32)
    if dtype == DTYPE_F32:
        return x
    if dtype == DTYPE_F16:
        return x.astype(np.float16)
    if dtype == DTYPE_FP8_E4M3:
        return e4m3_encode(x)
    raise ValueError(f"unknown dtype {dtype}")


def stored_to_f32(xs: np.ndarray, dtype: int) -> np.ndarray:
    if dtype == DTYPE_FP8_E4M3:
        return e4m3_decode(xs)
    return np.asarray(xs).astype(np.float32)


# ---- scoring + exact top-k ------------------------------------------------------------------------------

def _order_desc_pos_asc(scores: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """Indices that sort candidates by (score desc, position asc)."""
    return np.lexsort((pos, -scores.astype(np.float64)))


def search(q: np.ndarray, x_stored: np.ndarray, k: int, *, dtype: int = DTYPE_F32, ids: np.ndarray | None = None,
           id_base: int = 0, chunk: int = 65536, acc: str = "f32"):
    """faiss ``IndexFlatIP.search`` restated: scores = Q @ X^T (fp32, or fp64 when ``acc='f64'``), k largest per
    query, best first, ties by row position ascending.

    ``q``        [B, d] values already in the index's storage family (fp32 array of the values the GPU will
                 see, i.e. already rounded to fp16/fp8 when the index is fp16/fp8).
    ``x_stored`` [N, d] stored rows (``quantize_rows`` output).
    Returns ``(scores [B, kk] float32, ids [B, kk] int64, pos [B, kk] int64)`` with ``kk = min(k, N)``.
    The N axis is processed in chunks so the B x chunk block stays cache resident.
    """
    q = np.ascontiguousarray(q, dtype=np.float64 if acc == "f64" else np.float32)
    n = x_stored.shape[0]
    b = q.shape[0]
    kk = min(k, n)
    best_s = np.full((b, 0), 0, dtype=q.dtype)
    best_p = np.zeros((b, 0), dtype=np.int64)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        xc = stored_to_f32(x_stored[c0:c1], dtype).astype(q.dtype, copy=False)
        s = q @ xc.T  # [B, c]
        c = c1 - c0
        if c > kk:
            # every score >= the kk-th largest survives (keeps all members of a boundary tie group)
            part = np.partition(s, c - kk, axis=1)[:, c - kk]
            keep = s >= part[:, None]
            width = int(keep.sum(axis=1).max())
            # gather survivors row by row, padded with -inf
            order = np.argsort(~keep, axis=1, kind="stable")[:, :width]  # survivor positions, ascending
            cs = np.take_along_axis(s, order, axis=1)
            valid = np.take_along_axis(keep, order, axis=1)
            cs = np.where(valid, cs, -np.inf)
            cp = order.astype(np.int64) + c0
        else:
            cs = s
            cp = np.broadcast_to(np.arange(c0, c1, dtype=np.int64), (b, c)).copy()
        ms = np.concatenate([best_s, cs], axis=1)
        mp = np.concatenate([best_p, cp], axis=1)
        keep_n = min(kk, ms.shape[1])  # fewer than kk candidates exist while chunk < kk
        new_s = np.empty((b, keep_n), dtype=q.dtype)
        new_p = np.empty((b, keep_n), dtype=np.int64)
        for i in range(b):
            o = _order_desc_pos_asc(ms[i], mp[i])[:keep_n]
            new_s[i] = ms[i, o]
            new_p[i] = mp[i, o]
        best_s, best_p = new_s, new_p
    out_ids = position_to_id(best_p, ids, id_base)
    return best_s.astype(np.float32), out_ids, best_p


def search_blocked(q: np.ndarray, x32: np.ndarray, k: int, *, block: int = 16384, threads: int | None = None):
    """The same search as :func:`search` written for SPEED on the host cores (the ``cpu_baseline`` leg of ``bench.py``;
    stand-in for faiss ``IndexFlatIP.search``, which blocks the corpus the same way): per block of rows one fp32 GEMM
    ``Q @ Xb^T`` (``torch.mm`` -> the host BLAS, all cores) and ``torch.topk`` on the ``[B, block]`` score block (small
    enough to stay in the last-level cache), then one merge of the per-block winners.  ``x32`` [N, d] float32, host
    resident.  Ties exactly at a block's k-th place follow ``torch.topk`` (unspecified); the final order is
    score desc, position asc.  Returns ``(scores [B, k] float32, positions [B, k] int64)``."""
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    qt = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32))
    xt = torch.from_numpy(x32) if isinstance(x32, np.ndarray) else x32
    n = int(xt.shape[0])
    kk = min(k, n)
    vals, poss = [], []
    with torch.no_grad():
        for c0 in range(0, n, block):
            xb = xt[c0:c0 + block]
            s = qt @ xb.T
            v, i = torch.topk(s, min(kk, xb.shape[0]), dim=1)
            vals.append(v)
            poss.append(i + c0)
        v = torch.cat(vals, dim=1).numpy()
        p = torch.cat(poss, dim=1).numpy()
    out_s = np.empty((q.shape[0], kk), dtype=np.float32)
    out_p = np.empty((q.shape[0], kk), dtype=np.int64)
    if v.shape[1] > 4 * kk:  # shrink with one argpartition before the exact ordering
        sel = np.argpartition(-v, kk - 1, axis=1)[:, :2 * kk] if v.shape[1] > 2 * kk else None
        if sel is not None:
            v, p = np.take_along_axis(v, sel, axis=1), np.take_along_axis(p, sel, axis=1)
    for i in range(q.shape[0]):
        o = _order_desc_pos_asc(v[i], p[i])[:kk]
        out_s[i], out_p[i] = v[i, o], p[i, o]
    return out_s, out_p


def position_to_id(pos: np.ndarray, ids: np.ndarray | None, id_base: int = 0) -> np.ndarray:
    """faiss ``IDMap``: external id of a row position (``ids[pos]``, or ``id_base + pos`` without an id vector;
    a fresh sqlite AUTOINCREMENT table gives ``id_base = 1``, ``setup_db.py:14``)."""
    pos = np.asarray(pos, dtype=np.int64)
    if ids is None:
        return pos + np.int64(id_base)
    return np.asarray(ids, dtype=np.int64)[pos]


def full_scores(q: np.ndarray, x_stored: np.ndarray, dtype: int = DTYPE_F32, acc: str = "f64") -> np.ndarray:
    """Dense [B, N] score matrix (small N only) used by the tie-aware comparator."""
    t = np.float64 if acc == "f64" else np.float32
    return np.asarray(q, dtype=t) @ stored_to_f32(x_stored, dtype).astype(t).T


def merge_shards(scores: np.ndarray, ids: np.ndarray, k: int):
    """Final merge after the all-gather: ``scores``/``ids`` are [R, B, k_r] per-shard results (each already
    sorted best first, padded with ``-inf`` / ``-1``); shards are contiguous row ranges in rank order, so the
    global tie order (score desc, row position asc) is (score desc, rank asc, slot asc)."""
    scores = np.asarray(scores, dtype=np.float32)
    ids = np.asarray(ids, dtype=np.int64)
    r, b, kr = scores.shape
    flat_s = scores.transpose(1, 0, 2).reshape(b, r * kr)
    flat_i = ids.transpose(1, 0, 2).reshape(b, r * kr)
    seq = np.arange(r * kr, dtype=np.int64)  # rank-major, slot-minor
    kk = min(k, r * kr)
    out_s = np.empty((b, kk), dtype=np.float32)
    out_i = np.empty((b, kk), dtype=np.int64)
    for i in range(b):
        o = _order_desc_pos_asc(flat_s[i], seq)[:kk]
        out_s[i] = flat_s[i, o]
        out_i[i] = flat_i[i, o]
    return out_s, out_i


def recall_at_k(got_ids: np.ndarray, ref_ids: np.ndarray) -> float:
    """mean |got ∩ ref| / k over queries."""
    got_ids = np.asarray(got_ids)
    ref_ids = np.asarray(ref_ids)
    hits = [len(set(g.tolist()) & set(r.tolist())) for g, r in zip(got_ids, ref_ids)]
    return float(np.mean(hits)) / ref_ids.shape[1]


# ---- tie-aware comparison (SURVEY.md section 8d "Parity protocol") ---------------------------------------------

def check_topk(got_scores: np.ndarray, got_pos: np.ndarray, s_full: np.ndarray, k: int, *, score_tol: float = 1e-5,
               tie_tol: float = 2e-6) -> None:
    """Assert a GPU result against the dense fp64 oracle scores ``s_full`` [B, N].

    * every returned score is within ``score_tol`` of the oracle score of the returned row,
    * returned scores are non-increasing (within ``tie_tol``), rows are distinct,
    * rank-by-rank: the id must equal the oracle id wherever the oracle's neighbouring gaps exceed ``tie_tol``;
      inside a near-tie group only set membership is required: no row outside the result may beat the result's
      last score by more than ``tie_tol``, and every returned row is within ``tie_tol`` of deserving its rank.
    """
    b, n = s_full.shape
    kk = min(k, n)
    got_scores = np.asarray(got_scores)[:, :kk]
    got_pos = np.asarray(got_pos)[:, :kk]
    order = np.argsort(-s_full, axis=1, kind="stable")  # score desc, position asc
    for i in range(b):
        gp = got_pos[i]
        assert len(set(gp.tolist())) == kk, f"query {i}: duplicate rows {gp}"
        assert gp.min() >= 0 and gp.max() < n, f"query {i}: row out of range {gp}"
        ref_sc = s_full[i, gp]
        err = np.abs(got_scores[i].astype(np.float64) - ref_sc).max()
        assert err <= score_tol, f"query {i}: score error {err} > {score_tol}"
        assert np.all(np.diff(ref_sc) <= tie_tol), f"query {i}: result not sorted: {ref_sc}"
        ro = order[i]
        ref_top = s_full[i, ro[: kk + 1]] if n > kk else np.append(s_full[i, ro[:kk]], -np.inf)
        for j in range(kk):
            lo_gap = ref_top[j] - ref_top[j + 1]
            hi_gap = ref_top[j - 1] - ref_top[j] if j > 0 else np.inf
            if lo_gap > tie_tol and hi_gap > tie_tol:
                assert gp[j] == ro[j], f"query {i} rank {j}: got row {gp[j]} want {ro[j]} (gaps {hi_gap}, {lo_gap})"
            else:
                assert abs(ref_sc[j] - ref_top[j]) <= tie_tol, (
                    f"query {i} rank {j}: row {gp[j]} score {ref_sc[j]} not within tie_tol of oracle {ref_top[j]}")
        # nothing outside the result beats its last element by more than tie_tol
        mask = np.ones(n, dtype=bool)
        mask[gp] = False
        if mask.any():
            assert s_full[i, mask].max() <= ref_sc[-1] + tie_tol, f"query {i}: a better row was left out"


# ---- int8 sketch of large fp16 / fp32 shards: CPU restatement of the pruning bound (checker only) ------------------
# The build's large shards run their main launch over an int8 sketch (csrc/score_topk.hip MODE 2, csrc/convert.hip
# sketch_rows_kernel) and score exactly only the (query, row) pairs whose upper bound reaches the threshold.  What is
# restated here is the arithmetic that decides it, so that CPU tests can hold the bound itself to float64 truth:
#   rows:    one scale per 256-row tile, s_t = max|x| over the tile / 127;  x_int = clip(rint(x / s_t), -127, 127)
#   queries: one scale per query,        s_q = max|q| / 127
#   q . x  <=  s_q s_t (q_int . x_int) + ||q_lo|| max_tile ||x_hi|| + ||q|| max_tile ||x_lo||     (Cauchy-Schwarz on the residues)
# Reference interface: the scoring inside `embeddings.search` (heavy_ranker.py:98-101) -- the sketch only decides which rows
# need the exact inner product.

def sketch_rows(x: np.ndarray, tile: int = 256):
    """Stored rows [n, d] -> (x_int int8 [n, d], scale per tile [tiles], max ||x_hi|| per tile, max ||x_lo|| per tile)."""
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    tiles = (n + tile - 1) // tile
    xi = np.zeros(x.shape, dtype=np.int8)
    scale = np.ones(tiles, dtype=np.float32)
    hi_max = np.zeros(tiles, dtype=np.float32)
    lo_max = np.zeros(tiles, dtype=np.float32)
    for t in range(tiles):
        rows = x[t * tile:(t + 1) * tile]
        m = np.float32(np.abs(rows).max()) if rows.size else np.float32(0)
        s = np.float32(m / np.float32(127.0)) if np.isfinite(m) and m > 0 else np.float32(1.0)
        q = np.clip(np.rint(rows / s), -127, 127).astype(np.float32)
        hi = q * s
        xi[t * tile:(t + 1) * tile] = q.astype(np.int8)
        scale[t] = s
        hi_max[t] = np.sqrt((hi.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32)).max() * np.float32(1 + 2.0 ** -16)
        lo_max[t] = np.sqrt(((rows - hi) ** 2).sum(axis=1, dtype=np.float32)).max() * np.float32(1 + 2.0 ** -16)
    return xi, scale, hi_max, lo_max


def sketch_queries(q: np.ndarray):
    """Query rows [b, d] -> (q_int int8, scale [b], ||q_lo|| [b], ||q|| [b])."""
    q = np.asarray(q, dtype=np.float32)
    m = np.abs(q).max(axis=1)
    s = np.where(m > 0, m / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    qi = np.clip(np.rint(q / s[:, None]), -127, 127).astype(np.float32)
    lo = q - qi * s[:, None]
    up = np.float32(1 + 2.0 ** -16)
    return qi.astype(np.int8), s, np.sqrt((lo ** 2).sum(axis=1, dtype=np.float32)) * up, np.sqrt((q ** 2).sum(axis=1, dtype=np.float32)) * up


def _sketch_signs(d8: int) -> np.ndarray:
    """The random signs D of the sketch's rotation (csrc/convert.hip: sketch_rotate): +-1 from a mixed hash of the element index."""
    h = (np.arange(d8, dtype=np.uint64) + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return np.where((h & np.uint64(1)) == 1, -1.0, 1.0)


def sketch_transform(v: np.ndarray, mu=None) -> np.ndarray:
    """T (v - mu) of csrc/convert.hip (sketch_rotate): rows padded to a multiple of 128 elements, random signs, then the normalised
    Walsh-Hadamard transform of every power-of-two block of the padded row (768 = 512 + 256).  T is orthogonal:
    sketch_transform(q) . sketch_transform(x, mu) + q . mu = q . x.  float64 here (the kernel works in fp32)."""
    v = np.asarray(v, dtype=np.float64)
    n, d = v.shape
    d8 = (d + 127) // 128 * 128
    y = np.zeros((n, d8))
    y[:, :d] = v - (0.0 if mu is None else np.asarray(mu, dtype=np.float64)[None, :d])
    y *= _sketch_signs(d8)[None, :]
    off, rem = 0, d8
    while rem > 0:
        b = 1 << (rem.bit_length() - 1)
        blk = y[:, off:off + b]
        h = 1
        while h < b:  # in-place butterflies on bit h of the element index
            w = blk.reshape(n, b // (2 * h), 2, h)
            lo, hi = w[:, :, 0, :] + w[:, :, 1, :], w[:, :, 0, :] - w[:, :, 1, :]
            w[:, :, 0, :], w[:, :, 1, :] = lo, hi
            h *= 2
        blk /= np.sqrt(b)
        off += b
        rem -= b
    return y


def sketch_upper_bounds(q: np.ndarray, x: np.ndarray, tile: int = 256, transform: bool = False, mu=None) -> np.ndarray:
    """[b, n] rigorous upper bounds of q . x (the stored values, real-number dot product) from the int8 sketches.
    ``transform``: the sketches are cut from the centred, rotated vectors (what the library does by default): rows T (x - mu),
    queries T q, and q . mu joins the bound."""
    off = 0.0
    if transform:
        mu = x[: min(len(x), 65536)].astype(np.float64).mean(axis=0) if mu is None else mu
        off = (np.asarray(q, dtype=np.float64) @ np.asarray(mu, dtype=np.float64))[:, None]
        q, x = sketch_transform(q).astype(np.float32), sketch_transform(x, mu).astype(np.float32)
        # (the float32 rounding of the transformed vectors is what the kernel quantises; its distance to the exact transform is
        # part of the margin the kernel adds: 1e-6 relative here)
        off = off + 4e-6 * np.linalg.norm(q, axis=1)[:, None] * (np.linalg.norm(x, axis=1)[None, :] + np.linalg.norm(mu))
        xi, sx, hi_max, lo_max = sketch_rows(x, tile)
        qi, sq, qlo, qn = sketch_queries(q)
        d_int = qi.astype(np.int64) @ xi.astype(np.int64).T
        t_of = np.arange(x.shape[0]) // tile
        main = sq[:, None].astype(np.float64) * sx[t_of][None, :].astype(np.float64) * d_int
        slack = qlo[:, None].astype(np.float64) * hi_max[t_of][None, :] + qn[:, None].astype(np.float64) * lo_max[t_of][None, :]
        return main + slack + off
    xi, sx, hi_max, lo_max = sketch_rows(x, tile)
    qi, sq, qlo, qn = sketch_queries(q)
    d_int = qi.astype(np.int64) @ xi.astype(np.int64).T  # exact, as the int32 MFMA accumulators are
    t_of = np.arange(x.shape[0]) // tile
    main = sq[:, None].astype(np.float64) * sx[t_of][None, :].astype(np.float64) * d_int
    slack = qlo[:, None].astype(np.float64) * hi_max[t_of][None, :] + qn[:, None].astype(np.float64) * lo_max[t_of][None, :]
    return main + slack


def sketch_upper_bounds_centre_split(q: np.ndarray, x: np.ndarray, tile: int = 256, mu=None, per_row: bool = False) -> np.ndarray:
    """[b, n] upper bounds of q . x with the rank-one structure of embeddings that share a large common component taken out of
    the slack (csrc/convert.hip sketch_rows_kernel, csrc/score_topk.hip MODE 2).  w = T mu / ||T mu||, y = T (x - mu), z = T q,
    alpha = z . w, z_r = z - alpha w.
    * split form (``per_row=False``; every centred shard): the sketches are those of y and z as before, only the slack term
      |z . x_lo| <= ||z|| ||x_lo|| becomes |alpha| max_tile |w . x_lo| + ||z_r|| max_tile ||x_lo|| (x_lo is a quantisation residue:
      nearly orthogonal to any fixed direction);
    * per-row form (shards whose rows have a mean cosine >= 0.6): beta = w . y per row, y_r = y - beta w; the sketches are cut from y_r and z_r and
      z . y = alpha beta + beta (z_r . w) + z_r . y_r: the scan adds alpha beta per (query, row), the middle term is rounding-sized."""
    x64, q64 = np.asarray(x, dtype=np.float64), np.asarray(q, dtype=np.float64)
    mu = x64[: min(len(x64), 65536)].mean(axis=0) if mu is None else np.asarray(mu, dtype=np.float64)
    off = (q64 @ mu)[:, None]
    z, y = sketch_transform(q64), sketch_transform(x64, mu)
    w = sketch_transform(mu[None, :])[0]
    w = w / max(np.linalg.norm(w), 1e-300)
    alpha = z @ w
    zr = z - alpha[:, None] * w[None, :]
    t_of = np.arange(x64.shape[0]) // tile
    margin = 4e-6 * np.linalg.norm(z, axis=1)[:, None] * (np.linalg.norm(y, axis=1)[None, :] + np.linalg.norm(mu))  # fp32 storage of the transformed vectors
    if per_row:
        beta = y @ w
        yr = (y - beta[:, None] * w[None, :]).astype(np.float32)
        zr32 = zr.astype(np.float32)
        xi, sx, hi_max, lo_max = sketch_rows(yr, tile)
        qi, sq, qlo, _ = sketch_queries(zr32)
        d_int = qi.astype(np.int64) @ xi.astype(np.int64).T
        main = sq[:, None].astype(np.float64) * sx[t_of][None, :].astype(np.float64) * d_int
        rn = np.linalg.norm(zr32.astype(np.float64), axis=1) * (1 + 2.0 ** -16)
        slack = qlo[:, None].astype(np.float64) * hi_max[t_of][None, :] + rn[:, None] * lo_max[t_of][None, :]
        cross = np.abs(zr @ w)[:, None] * np.abs(beta)[None, :]  # beta (z_r . w): zero in exact arithmetic, kept for rigour
        return off + alpha[:, None] * beta[None, :] + main + slack + cross + margin
    y32, z32 = y.astype(np.float32), z.astype(np.float32)
    xi, sx, hi_max, lo_max = sketch_rows(y32, tile)
    qi, sq, qlo, _ = sketch_queries(z32)
    d_int = qi.astype(np.int64) @ xi.astype(np.int64).T
    main = sq[:, None].astype(np.float64) * sx[t_of][None, :].astype(np.float64) * d_int
    lo = y32.astype(np.float64) - sx[t_of][:, None].astype(np.float64) * xi.astype(np.float64)
    tiles = (x64.shape[0] + tile - 1) // tile
    c_max = np.array([np.abs(lo[t * tile:(t + 1) * tile] @ w).max() for t in range(tiles)]) * (1 + 2.0 ** -16)
    rn = np.linalg.norm(zr, axis=1) * (1 + 2.0 ** -16)
    slack = qlo[:, None].astype(np.float64) * hi_max[t_of][None, :] + np.abs(alpha)[:, None] * c_max[t_of][None, :] + rn[:, None] * lo_max[t_of][None, :]
    return off + main + slack + margin

