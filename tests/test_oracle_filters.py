"""CPU tests of the oracle's restatement of the two sphere-range consumers (bilateral filter, WLOP).

The reference's own tests for them hold no numbers, only properties (test/algorithm/bilateral_filter.cpp:126-129, :148-150:
the displaced points move towards the line, the normal count is kept; test/algorithm/wlop.cpp:79-96: I points come back,
none NaN or Inf).  Those are checked here on the oracle, and the oracle's C++ loops are cross-checked against an independent
float64 numpy evaluation of the formulas as the reference writes them (brute-force ranges, vectorised sums)."""
import numpy as np
import pytest


def _line_scenario():
    pts = np.array([[-0.1, 0, 0], [-0.075, 0, 0], [-0.05, 0, 0.01], [-0.025, 0, 0], [0, 0, 0], [0.025, 0, 0], [0.05, 0, -0.01],
                    [0.075, 0, 0], [0.1, 0, 0]], np.float32)
    nrm = np.array([[0, 0, 1], [0, 0, 1], [-0.19611614, 0, 0.98058068], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0.19611614, 0, 0.98058068],
                    [0, 0, 1], [0, 0, 1]], np.float32)
    return pts, nrm


def _gauss(sigma, r):
    return np.exp(-r * r / (2 * sigma * sigma)) / (sigma * np.sqrt(2 * np.pi))


def _dgauss(sigma, r):
    return -r / (sigma ** 3 * np.sqrt(2 * np.pi)) * np.exp(-r * r / (2 * sigma * sigma))


def _ranges(pts, centres, r):
    """brute-force sphere ranges with the reference's float predicate (sphere.hpp:52-56)"""
    p32, c32 = pts.astype(np.float32), centres.astype(np.float32)
    out = []
    r2 = np.float32(r) * np.float32(r)
    for c in c32:
        d = p32 - c
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        out.append(np.nonzero(d2 <= r2)[0])
    return out


def _np_bilateral_points(pts, nrm, sf, sg):
    sf, sg = float(np.float32(sf)), float(np.float32(sg))
    P, N = pts.astype(np.float64), nrm.astype(np.float64)
    out = np.empty_like(P)
    for i, nb in enumerate(_ranges(pts, pts, 2 * np.float32(sf))):
        s = P[i]
        d = np.sum((P[nb] - s) * N[nb], axis=1)
        proj = s + d[:, None] * N[nb]
        w = _gauss(sf, np.linalg.norm(s - P[nb], axis=1)) * _gauss(sg, np.linalg.norm(proj - s, axis=1))
        out[i] = (w[:, None] * proj).sum(0) / w.sum()
    return out


def _np_bilateral_normals(pts, nrm, sf, sg):
    """compute_ni as written (bilateral_filter.hpp:103-269), including its projection Jacobian (:213-222)."""
    sf, sg = float(np.float32(sf)), float(np.float32(sg))
    P, N = pts.astype(np.float64), nrm.astype(np.float64)
    out = np.empty_like(N)

    def unit(v):
        n = np.linalg.norm(v)
        return v / n if n > 0 else v

    for i, nb in enumerate(_ranges(pts, pts, 2 * np.float32(sf))):
        s, ns = P[i], N[i]
        Jsum, pifg, gk, k = np.zeros((3, 3)), np.zeros(3), np.zeros(3), 0.0
        for j in nb:
            p, n = P[j], N[j]
            proj = s + np.dot(p - s, n) * n
            sp, sps = s - p, proj - s
            rf, rg = np.linalg.norm(sp), np.linalg.norm(sps)
            wf, wg = _gauss(sf, rf), _gauss(sg, rg)
            k += wf * wg
            pifg += wf * wg * proj
            gf = unit(sp) * _dgauss(sf, rf)
            Jpi = np.outer(n, n)
            np.fill_diagonal(Jpi, 1 - n * n)
            u = unit(sps)
            gg = (u @ Jpi - u) * _dgauss(sg, rg)
            gk += gf * wg + wf * gg
            Jsum += Jpi * wf * wg + np.outer(sps, gf) * wg + np.outer(sps * wf, gg)
        J = (Jsum * k - np.outer(pifg, gk)) / (k * k)
        out[i] = unit(J @ ns)
    return out


def test_bilateral_reference_scenario_properties(oracle):
    pts, nrm = _line_scenario()
    idx, cnt = oracle.KdTree(pts, compute_max_depth=True).knn(pts, 2)
    sigmaf = float(np.mean(oracle.mean_dist_from_knn(pts, pts, idx, cnt)))
    out = oracle.bilateral_filter_points(pts, nrm, sigmaf, sigmaf / 8.0, K=2)
    assert pts[2, 2] > out[2, 2] and pts[6, 2] < out[6, 2]  # bilateral_filter.cpp:126-129
    assert out.shape == pts.shape and np.isfinite(out).all()
    on = oracle.bilateral_filter_normals(pts, nrm, sigmaf, sigmaf / 8.0, K=2)
    assert on.shape == nrm.shape  # :148-150
    assert np.allclose(np.linalg.norm(on, axis=1), 1.0, atol=1e-5)


def test_bilateral_oracle_matches_independent_float64_evaluation(oracle):
    rng = np.random.default_rng(5)
    n = 600
    xy = rng.uniform(-1, 1, (n, 2))
    pts = np.column_stack([xy, 0.1 * np.sin(3 * xy[:, 0]) + rng.normal(0, 0.01, n)]).astype(np.float32)
    nrm = np.column_stack([-0.3 * np.cos(3 * xy[:, 0]), np.zeros(n), np.ones(n)]) + rng.normal(0, 0.05, (n, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    sf, sg = 0.12, 0.03
    ref = _np_bilateral_points(pts, nrm, sf, sg)
    assert np.abs(oracle.bilateral_filter_points(pts, nrm, sf, sg, K=1) - ref).max() <= 2e-6
    assert np.abs(oracle.bilateral_filter_points(pts, nrm, sf, sg, K=1, f64_yardstick=True) - ref).max() <= 2e-7
    refn = _np_bilateral_normals(pts, nrm, sf, sg)
    got = oracle.bilateral_filter_normals(pts, nrm, sf, sg, K=1)
    got64 = oracle.bilateral_filter_normals(pts, nrm, sf, sg, K=1, f64_yardstick=True)
    assert (1 - np.sum(got64 * refn, axis=1)).max() <= 2e-7  # (the yardstick is returned rounded to float32)
    assert (1 - np.sum(got * refn, axis=1)).max() <= 1e-4  # north_star's normal tolerance
    # K iterations = K single iterations chained (points: the tree follows the points; normals: it does not)
    two = oracle.bilateral_filter_points(oracle.bilateral_filter_points(pts, nrm, sf, sg, K=1), nrm, sf, sg, K=1)
    assert np.array_equal(oracle.bilateral_filter_points(pts, nrm, sf, sg, K=2), two)
    twon = oracle.bilateral_filter_normals(pts, oracle.bilateral_filter_normals(pts, nrm, sf, sg, K=1), sf, sg, K=1)
    assert np.array_equal(oracle.bilateral_filter_normals(pts, nrm, sf, sg, K=2), twon)


def test_bilateral_planar_cloud_is_a_fixed_point(oracle):
    rng = np.random.default_rng(3)
    pts = np.column_stack([rng.uniform(-1, 1, (800, 2)), np.full(800, 0.25)]).astype(np.float32)
    nrm = np.tile(np.float32([0, 0, 1]), (800, 1))
    out = oracle.bilateral_filter_points(pts, nrm, 0.1, 0.02, K=2)
    assert np.abs(out - pts).max() <= 1e-6 and np.all(out[:, 2] == np.float32(0.25))


def _np_wlop_step(P, x, mu, h, uniform):
    P64, x64 = P.astype(np.float64), x.astype(np.float64)
    h = float(np.float32(h))
    theta = lambda r2: np.exp(-r2 / (h * h / 16.0))  # noqa: E731

    def density(pts32, pts64):
        v = np.ones(len(pts64))
        for c, nb in enumerate(_ranges(pts32, pts32, h)):
            d = pts64[nb] - pts64[c]
            keep = ~np.all(np.abs(d) < 1e-9, axis=1)
            v[c] += theta(np.sum(d[keep] ** 2, axis=1)).sum()
        return v

    vj = density(P, P64) if uniform else np.ones(len(P))
    wi = density(x, x64) if uniform else np.ones(len(x))
    out = np.empty_like(x64)
    rp, rq = _ranges(P, x, h), _ranges(x, x, h)
    for i in range(len(x)):
        q = x64[i]
        d = P64[rp[i]] - q
        keep = ~np.all(np.abs(d) < 1e-9, axis=1)
        nb = rp[i][keep]
        r = np.linalg.norm(P64[nb] - q, axis=1)
        coeff = theta(r * r) / r / vj[nb]
        med = q if abs(coeff.sum()) < 1e-9 else (coeff[:, None] * P64[nb]).sum(0) / coeff.sum()
        d = x64[rq[i]] - q
        keep = ~np.all(np.abs(d) < 1e-9, axis=1)
        nb = rq[i][keep]
        dv = q - x64[nb]
        r = np.linalg.norm(dv, axis=1)
        coeff = wi[nb] * theta(r * r) / r
        rep = np.zeros(3) if abs(coeff.sum()) < 1e-9 else float(np.float32(mu)) / coeff.sum() * (coeff[:, None] * dv).sum(0)
        out[i] = med + rep
    return out


@pytest.mark.parametrize("uniform", [True, False])
def test_wlop_oracle_matches_independent_float64_evaluation(oracle, uniform):
    rng = np.random.default_rng(6)
    P = rng.uniform(-1, 1, (1500, 3)).astype(np.float32)
    sample = rng.permutation(1500)[1000:].astype(np.uint64)
    ref = _np_wlop_step(P, P[sample.astype(np.int64)], 0.45, 0.3, uniform)
    assert np.abs(oracle.wlop(P, sample, 0.45, 0.3, 1, uniform=uniform) - ref).max() <= 3e-6
    assert np.abs(oracle.wlop(P, sample, 0.45, 0.3, 1, uniform=uniform, f64_yardstick=True) - ref).max() <= 2e-7
    assert np.array_equal(oracle.wlop(P, sample, 0.45, 0.3, 0), P[sample.astype(np.int64)])  # no iteration: the seeds


def test_wlop_reference_scenario_properties(oracle):
    """test/algorithm/wlop.cpp:20-96 with a fixed seed: 1000 points in [-10, 10]^3, I = n / 2, k = 2, h = the mean
    distance to 15 neighbours, uniform."""
    rng = np.random.default_rng(0)
    P = rng.uniform(-10, 10, (1000, 3)).astype(np.float32)
    idx, cnt = oracle.KdTree(P, compute_max_depth=True).knn(P, 15)
    h = float(np.mean(oracle.mean_dist_from_knn(P, P, idx, cnt)))
    sample = rng.permutation(1000)[500:].astype(np.uint64)
    out = oracle.wlop(P, sample, 0.45, h, 2, uniform=True)
    assert out.shape == (500, 3)
    assert not np.isinf(out).any() and not np.isnan(out).any()
