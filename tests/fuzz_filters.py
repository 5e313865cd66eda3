#!/usr/bin/env python3
"""Randomised parity run of the range consumers on the GPU box: random cloud shapes (uniform, clustered, slab, lattice with
exact ties, duplicates, huge dynamic range), sizes, radii, normals (unit, unnormalised, zero) -- bilateral_filter_points /
_normals and WLOP, one iteration from identical inputs, against the oracle.  Test infrastructure (uses the oracle).
A row counts as ill-conditioned -- and is only counted, not judged -- when the oracle's own float32 result is further than
HALF the tolerance (half the tolerated angle: a quarter of the 1 - cos bound) from its float64 evaluation: there the
reference's arithmetic has no stable answer to be equal to (e.g. all normals (0, 0, 1): the reference's J n cancels to rounding noise),
or when the float64 evaluation reports that the magnitudes added and subtracted for the row exceed what is left of them by more than
CANCEL_MAX, or sit in float32's denormal range (oracle: bilateral_ni's cancellation factor).
usage: python tests/fuzz_filters.py [seconds] [seed]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("point-cloud-processing_amd")
from oracle import pcp_oracle as O
O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
POS_TOL, COS_TOL = 4e-6, 1e-4
CANCEL_MAX = 3000.0  # rows whose added and subtracted magnitudes exceed the result by more than this (oracle, float64) are not judged
SIZES = [1, 2, 8, 9, 65, 300, 2000, 9000, 20000]


def cloud(n):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        p = rng.random((n, 3), dtype=np.float32)
    elif kind == 1:
        p = pkg.synthetic.clustered_cloud(max(n, 64), seed=int(rng.integers(1, 1 << 30)))[:n]
    elif kind == 2:
        p = rng.random((n, 3), dtype=np.float32) * np.array([1, 1, 1e-3], np.float32)
    elif kind == 3:
        m = int(round(n ** (1 / 3))) + 1
        p = np.stack(np.meshgrid(*[np.arange(m, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n] * np.float32(0.0625)
    elif kind == 4:
        base = rng.random((max(1, n // 3), 3), dtype=np.float32)
        p = base[rng.integers(0, len(base), n)]
    else:
        p = (rng.standard_normal((n, 3)) * np.array([30, 1, 1e-2])).astype(np.float32)
    return np.ascontiguousarray(p, np.float32), kind


def normals(n):
    v = rng.standard_normal((n, 3))
    mode = int(rng.integers(0, 4))
    if mode != 1:  # 1: left unnormalised
        v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
    v = v.astype(np.float32)
    if mode == 2 and n > 3:
        v[rng.integers(0, n, max(1, n // 50))] = 0.0  # some zero normals
    if mode == 3:
        v[:] = np.float32([0, 0, 1])
    return v, mode


def spacing(pts):
    ext = np.ptp(pts.astype(np.float64), axis=0)
    ext = ext[ext > 0]
    vol = float(np.prod(ext)) if len(ext) else 1.0
    return (vol / max(len(pts), 1)) ** (1.0 / max(len(ext), 1)) if len(ext) else 1.0


t_end = time.time() + budget
cases = fails = ill = rows = 0
worst = {"points": 0.0, "normals": 0.0, "wlop": 0.0}
while time.time() < t_end:
    n = int(rng.choice(SIZES))
    pts, kind = cloud(n)
    n = len(pts)
    nrm, nmode = normals(n)
    s = spacing(pts)
    radius = min(float(s * rng.choice([0.7, 1.5, 3.0])), 0.9)  # (radius <= 1: DESIGN.md, Radius > 1)
    sigmaf, sigmag = radius / 2, radius / 2 * float(rng.choice([0.1, 0.5, 2.0]))
    ext = max(float(np.abs(pts).max()), 1e-30)
    what = {"n": n, "kind": kind, "normals": nmode, "radius": radius, "sigmag": sigmag, "seed": seed, "case": cases}
    print("case", json.dumps(what), flush=True)
    try:
        got = pkg.bilateral_filter_points(pts, nrm, sigmaf, sigmag, K=1)
        exp = O.bilateral_filter_points(pts, nrm, sigmaf, sigmag, K=1, nthreads=8)
        yard = O.bilateral_filter_points(pts, nrm, sigmaf, sigmag, K=1, f64_yardstick=True, nthreads=8)
        with np.errstate(invalid="ignore"):
            d = np.abs(got - exp).max(axis=1)
            unstable = ~(np.abs(exp - yard).max(axis=1) <= 0.5 * POS_TOL * ext)
            same_nan = np.isnan(got).any(axis=1) & np.isnan(exp).any(axis=1)
            bad = ~((d <= POS_TOL * ext) | unstable | same_nan)
        ill += int(unstable.sum()); rows += n
        worst["points"] = max(worst["points"], float(np.nanmax(np.where(unstable | same_nan, 0, d)) / ext))
        if bad.any():
            raise AssertionError("bilateral points: %d rows differ, worst %.3g (extent %.3g)" % (int(bad.sum()), float(np.nanmax(d[bad])), ext))
        gn = pkg.bilateral_filter_normals(pts, nrm, sigmaf, sigmag, K=1)
        en = O.bilateral_filter_normals(pts, nrm, sigmaf, sigmag, K=1, nthreads=8)
        yn, cf = O.bilateral_filter_normals(pts, nrm, sigmaf, sigmag, K=1, f64_yardstick=True, nthreads=8, want_cancellation=True)
        with np.errstate(invalid="ignore", divide="ignore"):
            # directions are compared (unit vectors formed here in float64): where |J n|^2 underflows in float32 the reference's
            # normalize() leaves the vector short or untouched (Eigen: divide only if the squared norm is > 0), on both sides
            def unit(a):
                a = a.astype(np.float64)
                return a / np.linalg.norm(a, axis=1, keepdims=True)
            ug, ue, uy = unit(gn), unit(en), unit(yn)
            c = 1 - np.sum(ug * ue, axis=1)
            unstable = ~((1 - np.sum(ue * uy, axis=1)) <= 0.25 * COS_TOL) | ~(cf <= CANCEL_MAX)
            both_degenerate = (np.isnan(gn).any(axis=1) & np.isnan(en).any(axis=1)) | ((np.abs(gn).max(axis=1) == 0) & (np.abs(en).max(axis=1) == 0))
            bad = ~((c <= COS_TOL) | unstable | both_degenerate)
        ill += int(unstable.sum()); rows += n
        worst["normals"] = max(worst["normals"], float(np.nanmax(np.where(unstable | both_degenerate, 0, c))))
        if bad.any():
            raise AssertionError("bilateral normals: %d rows differ, worst 1-cos %.3g" % (int(bad.sum()), float(np.nanmax(c[bad]))))
        m = int(rng.integers(1, n + 1))
        sample = rng.permutation(n)[:m].astype(np.uint64)
        if rng.random() < 0.2:
            sample[: m // 2] = sample[m // 2: 2 * (m // 2)]  # duplicated seeds
        mu = float(rng.choice([0.0, 0.2, 0.45, 0.5]))
        uniform = bool(rng.integers(0, 2))
        gw = pkg.wlop(pts, mu=mu, h=radius, k=1, uniform=uniform, sample=sample)
        ew = O.wlop(pts, sample, mu, radius, 1, uniform=uniform, nthreads=8)
        yw = O.wlop(pts, sample, mu, radius, 1, uniform=uniform, f64_yardstick=True, nthreads=8)
        with np.errstate(invalid="ignore"):
            d = np.abs(gw - ew).max(axis=1)
            unstable = ~(np.abs(ew - yw).max(axis=1) <= 0.5 * POS_TOL * ext)
            bad = ~((d <= POS_TOL * ext) | unstable)
        ill += int(unstable.sum()); rows += m
        worst["wlop"] = max(worst["wlop"], float(np.nanmax(np.where(unstable, 0, d)) / ext))
        if bad.any():
            raise AssertionError("wlop: %d rows differ, worst %.3g (extent %.3g)" % (int(bad.sum()), float(np.nanmax(d[bad])), ext))
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL", json.dumps(what), repr(e)[:300], flush=True)
    cases += 1
print(json.dumps({"cases": cases, "failures": fails, "rows_checked": rows, "rows_ill_conditioned_in_the_reference_arithmetic": ill,
                  "worst_relative_position_error": worst["points"], "worst_1_minus_cos": worst["normals"], "worst_relative_wlop_error": worst["wlop"],
                  "seed": seed, "seconds": budget}))
sys.exit(1 if fails else 0)
