"""The pruning bound of the int8 sketch search, held to float64 truth on the CPU (oracle/retrieval.py: sketch_*).

The GPU path prunes a (query, row) pair only when this bound stays below the running threshold, so the bound must never fall
below the true score -- for unit vectors, for tiles whose scale one outlier component sets, for unnormalised and zero rows, for
values that clip, for fp16 and fp32 stored values alike -- and it must be tight enough to prune (otherwise the sketch search
degenerates into its exact fallback)."""
import numpy as np
import pytest

from oracle import retrieval as R


def _cases():
    rng = np.random.default_rng(0)
    unit = R.l2_normalize(rng.standard_normal((1500, 96)).astype(np.float32))
    out = [("unit fp16", unit.astype(np.float16).astype(np.float32)), ("unit fp32", unit)]
    x = unit.copy()
    x[300, 7] = 800.0          # one component sets the scale of its tile
    x[301] *= 40               # a row of norm 40
    x[600:700] = 0             # zero rows (a zero tile among them: rows 512..767 are not all zero, 600..700 are)
    x[1400:] = 0
    out.append(("outliers, zero rows", x.astype(np.float16).astype(np.float32)))
    out.append(("tiny values", (unit * 1e-4).astype(np.float16).astype(np.float32)))
    out.append(("heavy tails", (rng.standard_t(2, size=(1500, 96)) * 0.05).astype(np.float16).astype(np.float32)))
    return out


@pytest.mark.parametrize("name,x", _cases(), ids=[c[0] for c in _cases()])
def test_sketch_bound_never_falls_below_the_true_score(name, x):
    rng = np.random.default_rng(1)
    q = R.l2_normalize(rng.standard_normal((33, x.shape[1])).astype(np.float32)).astype(np.float16).astype(np.float32)
    q[0] = x[300] / max(np.linalg.norm(x[300]), 1e-9)   # along the outlier row
    q[1] = 0
    q[2] = x[5]
    ub = R.sketch_upper_bounds(q, x)
    true = q.astype(np.float64) @ x.astype(np.float64).T
    assert np.all(ub >= true), (name, float((true - ub).max()))
    assert np.all(np.isfinite(ub))


def test_sketch_bound_prunes_unit_vectors():
    """d = 768 unit vectors: the slack is about half a standard deviation of the scores.  At this toy size the threshold (10th
    best of 410 rows: 2 sigma) still lets 8 % of the pairs through; at 10M rows it sits at 4.3 sigma and 1e-4 of them survive
    (220 000 of 2.3e9: the GPU path's regime)."""
    rng = np.random.default_rng(2)
    n, d, b = 4096, 768, 16
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32)).astype(np.float16).astype(np.float32)
    q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32)).astype(np.float16).astype(np.float32)
    ub = R.sketch_upper_bounds(q, x)
    true = q.astype(np.float64) @ x.astype(np.float64).T
    slack = ub - true
    sigma = 1 / np.sqrt(d)
    assert 0 <= slack.min() and slack.max() < 1.2 * sigma and 0.3 * sigma < np.median(slack) < 0.8 * sigma
    theta = np.sort(true[:, :n // 10], axis=1)[:, -10]            # exact 10th best of the first tenth of the rows
    survivors = (ub[:, n // 10:] >= theta[:, None]).mean()
    assert survivors < 0.15
    # every row of the true top-10 survives
    top = np.argsort(-true, axis=1)[:, :10]
    assert all(ub[i, j] >= theta[i] for i in range(b) for j in top[i])


def test_sketch_rows_layout_and_scales():
    x = np.zeros((300, 8), dtype=np.float32)
    x[0, 0] = 1.27
    x[299, 3] = -2.54
    xi, s, hi, lo = R.sketch_rows(x)
    assert s.shape == (2,) and np.isclose(s[0], 0.01) and np.isclose(s[1], 0.02)
    assert xi[0, 0] == 127 and xi[299, 3] == -127 and hi[0] >= 1.27 and lo.max() < 1e-6


def test_the_sketch_rotation_is_orthogonal_and_flattens_outliers():
    """T = H D of csrc/convert.hip as restated in oracle/retrieval.py: (T q) . (T (x - mu)) + q . mu = q . x for any centre mu,
    norms are kept, and a row dominated by two outlier dimensions comes out with near-Gaussian coordinates (max / rms about 3-4
    instead of 19) -- which is what keeps the quantiser's step small."""
    rng = np.random.default_rng(3)
    for d in (64, 100, 768, 1000):
        x = rng.standard_normal((40, d))
        x[:, 3] *= 20
        x[:, d // 2] += 15
        q = rng.standard_normal((7, d))
        mu = x.mean(axis=0)
        tx, tq = R.sketch_transform(x, mu), R.sketch_transform(q)
        assert np.allclose(tq @ tx.T + (q @ mu)[:, None], q @ x.T, rtol=0, atol=1e-9 * np.abs(q @ x.T).max())
        assert np.allclose(np.linalg.norm(tx, axis=1), np.linalg.norm(x - mu, axis=1))
        ratio = lambda v: np.abs(v).max(axis=1) / np.sqrt((v ** 2).mean(axis=1))
        if d >= 768:
            assert ratio(x - mu).mean() > 10 and ratio(tx).mean() < 4.5
    const = np.ones((1, 768))
    assert (np.abs(R.sketch_transform(const)).max() / np.sqrt((R.sketch_transform(const) ** 2).mean())) < 4.5  # no spike from a constant row


@pytest.mark.parametrize("name,x", _cases(), ids=[c[0] for c in _cases()])
def test_transformed_sketch_bound_never_falls_below_the_true_score(name, x):
    rng = np.random.default_rng(1)
    q = R.l2_normalize(rng.standard_normal((33, x.shape[1])).astype(np.float32)).astype(np.float16).astype(np.float32)
    q[0] = x[300] / max(np.linalg.norm(x[300]), 1e-9)
    q[1] = 0
    q[2] = x[5]
    ub = R.sketch_upper_bounds(q, x, transform=True)
    true = q.astype(np.float64) @ x.astype(np.float64).T
    assert np.all(ub >= true), (name, float((true - ub).max()))
    assert np.all(np.isfinite(ub))


def test_transformed_sketch_prunes_anisotropic_rows():
    """Rows with a common component and two outlier dimensions: the plain sketch's bound leaves most rows above a top-10 threshold,
    the centred, rotated one a few."""
    rng = np.random.default_rng(8)
    n, d = 4096, 256
    def make(m):
        v = R.l2_normalize(rng.standard_normal((m, d)).astype(np.float32)) + np.ones(d, np.float32) / np.sqrt(d)
        v[:, 9] += 0.9
        return R.l2_normalize(v).astype(np.float16).astype(np.float32)
    x, q = make(n), make(16)
    true = q.astype(np.float64) @ x.astype(np.float64).T
    thr = np.sort(true, axis=1)[:, -10][:, None]
    plain = (R.sketch_upper_bounds(q, x) >= thr).mean()
    rotated = (R.sketch_upper_bounds(q, x, transform=True) >= thr).mean()
    assert rotated < 0.25 * plain and rotated < 0.2, (plain, rotated)


def _collapsed(n, b, d, weight, seed):
    """unit rows and queries that share ONE large common component (mean cosine weight^2 / (1 + weight^2))"""
    rng = np.random.default_rng(seed)
    c = rng.standard_normal(d)
    c /= np.linalg.norm(c)

    def draw(m):
        v = rng.standard_normal((m, d))
        v = weight * c + v / np.linalg.norm(v, axis=1, keepdims=True)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float16)

    return draw(n), draw(b)


def test_centre_split_and_per_row_bounds_on_collapsed_embeddings():
    """Rows that share one large common component (mean cosine 0.9: what an untrained encoder emits).  Both refinements of the bound
    stay above the float64 score of every pair, and each prunes more than the one before: plain transformed sketch -> slack term
    split along the centre direction -> per-row rank-one term."""
    n, d, b, k = 8192, 768, 16, 10
    x, q = _collapsed(n, b, d, 3.0, 5)
    true = q.astype(np.float64) @ x.astype(np.float64).T
    theta = np.sort(true, axis=1)[:, -k]
    counts, slack = {}, {}
    for name, ub in (("plain", R.sketch_upper_bounds(q.astype(np.float32), x.astype(np.float32), transform=True)),
                     ("split", R.sketch_upper_bounds_centre_split(q, x)),
                     ("per_row", R.sketch_upper_bounds_centre_split(q, x, per_row=True))):
        assert (ub >= true).all(), name
        counts[name] = int((ub >= theta[:, None]).sum())
        slack[name] = float((ub - true).mean())
    assert slack["per_row"] < 0.6 * slack["split"] < 0.6 * slack["plain"], slack
    assert counts["per_row"] < counts["split"] < counts["plain"], counts


def test_centre_split_changes_nothing_for_isotropic_rows():
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2048, 128))
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    q = x[:8] + 0  # queries drawn like the rows
    true = q.astype(np.float64) @ x.astype(np.float64).T
    for per_row in (False, True):
        ub = R.sketch_upper_bounds_centre_split(q, x, per_row=per_row)
        assert (ub >= true).all()
    plain = R.sketch_upper_bounds(q.astype(np.float32), x.astype(np.float32), transform=True)
    split = R.sketch_upper_bounds_centre_split(q, x)
    assert np.abs(split - plain).max() < 0.2 * (plain - true).mean()  # the same bound to within a fraction of its slack

