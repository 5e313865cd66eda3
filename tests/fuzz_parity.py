#!/usr/bin/env python3
"""Randomised parity run on the GPU box: random cloud shapes, sizes, k, eps, radii; the product (through the Python mirror of
the C ABI) against the oracle's brute force on a sample of queries, tie-aware.  Test infrastructure (uses the oracle); the fixed
cases live in tests/test_gpu_parity.py, this looks for what they miss.   usage: python tests/fuzz_parity.py [seconds] [seed] [big]   (tests/test_gpu_parity.py runs it for a few seconds)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # conftest helpers
import numpy as np
from conftest import knn_rows_equivalent
pkg = importlib.import_module("point-cloud-processing_amd")
from oracle import pcp_oracle as O
O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
SIZES = [1, 2, 7, 8, 9, 63, 64, 65, 100, 513, 4096, 30000, 200000]
if len(sys.argv) > 3 and sys.argv[3] == "big":  # deeper trees, many resident waves' worth of groups
    SIZES = [1_000_000, 2_500_000, 5_000_000]
rng = np.random.default_rng(seed)


def cloud(n):
    kind = rng.integers(0, 6)
    if kind == 0:
        p = rng.random((n, 3), dtype=np.float32)
    elif kind == 1:
        p = pkg.synthetic.clustered_cloud(max(n, 64), seed=int(rng.integers(1, 1 << 30)))[:n]
    elif kind == 2:  # thin slab / surface
        p = rng.random((n, 3), dtype=np.float32) * np.array([1, 1, 1e-4], np.float32)
    elif kind == 3:  # lattice with exact ties
        m = int(round(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(m, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n] * np.float32(0.125)
        p = g
    elif kind == 4:  # duplicates
        base = rng.random((max(1, n // 3), 3), dtype=np.float32)
        p = base[rng.integers(0, len(base), n)]
    else:  # huge dynamic range
        p = (rng.standard_normal((n, 3)) * np.array([1e3, 1, 1e-3])).astype(np.float32)
    return np.ascontiguousarray(p, np.float32), int(kind)


t_end = time.time() + budget
cases = fails = 0
while time.time() < t_end:
    n = int(rng.choice(SIZES))
    pts, kind = cloud(n)
    n = len(pts)
    k = int(rng.choice([1, 2, 3, 8, 15, 16, 17, 31, 32, 33, 40, 70]))
    eps = float(rng.choice([1e-5, 0.0, 1e-3, 1e-7, 3e-2]))
    coarse = bool(rng.integers(0, 2))  # PCPX_BUILD_COARSE_ORDER: same results by contract
    ix = pkg.Index(pts, coarse_order=coarse)
    eps_mode = int(rng.integers(0, 3))  # where k_knn applies the eps-box test: same results by contract
    ix.debug_eps_test_mode(eps_mode)
    sel = rng.choice(n, size=min(n, 300), replace=False)
    what = {"n": n, "kind": kind, "k": k, "eps": eps, "coarse_order": coarse, "eps_test_mode": eps_mode, "seed": seed, "case": cases}
    print("case", json.dumps(what), flush=True)  # so that a hang names its case
    try:
        idx, cnt = ix.knn_self(k, eps)[:2]
        oi, oc = O.knn_bruteforce(pts, pts[sel], k, eps=eps, nthreads=8)[:2]
        ok, why = knn_rows_equivalent(pts, pts[sel], idx[sel], cnt[sel], oi, oc)
        if not ok:
            raise AssertionError("knn_self: " + why)
        q = (pts[sel] + rng.standard_normal((len(sel), 3)).astype(np.float32) * np.float32(1e-2 * max(1e-6, float(np.ptp(pts, 0).max()))))
        bi, bc = ix.knn(q, k, eps)[:2]
        oi, oc = O.knn_bruteforce(pts, q, k, eps=eps, nthreads=8)[:2]
        ok, why = knn_rows_equivalent(pts, q, bi, bc, oi, oc)
        if not ok:
            raise AssertionError("knn_batch: " + why)
        r = float(np.ptp(pts, 0).max()) * float(rng.choice([0.0, 0.01, 0.1, 0.9]))
        if r <= 1.0:  # (radius > 1: the reference's pruning quirk, DESIGN.md "Semantics")
            rc = ix.range_count_self(r)
            orc = O.range_count_bruteforce(pts, pts[sel], r, nthreads=8)
            if not np.array_equal(rc[sel], orc):
                raise AssertionError("range_count_self differs")
        # sphere / box lists against a float32 numpy restatement of contains() (sphere.hpp:27-35: d2 <= r*r;
        # axis_aligned_bounding_box.hpp:111-125: inclusive)
        m = min(len(sel), 40)
        centers = q[:m]
        if r <= 1.0:
            off, out = ix.range_sphere(centers, r)
            r2 = np.float32(r) * np.float32(r)
            for i in range(m):
                d = pts - centers[i]
                d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                want = np.flatnonzero(d2 <= r2)
                got = np.sort(out[int(off[i]):int(off[i + 1])])
                if not np.array_equal(got, want):
                    raise AssertionError("range_sphere list %d differs (%d vs %d)" % (i, len(got), len(want)))
        half = np.abs(rng.standard_normal((m, 3)).astype(np.float32)) * np.float32(0.05 * max(1e-6, float(np.ptp(pts, 0).max())))
        boxes = np.concatenate([centers - half, centers + half], axis=1).astype(np.float32)
        off, out = ix.range_aabb(boxes)
        for i in range(m):
            inside = np.all((pts >= boxes[i, :3]) & (pts <= boxes[i, 3:]), axis=1)
            got = np.sort(out[int(off[i]):int(off[i + 1])])
            if not np.array_equal(got, np.flatnonzero(inside)):
                raise AssertionError("range_aabb list %d differs" % i)
        if n >= 3:  # tangent planes and mean neighbour distances for any k
            cen, pn = ix.tangent_planes_knn_self(k, eps)
            md = ix.mean_knn_distance_self(k, eps)
            if not np.array_equal(cen[sel].view(np.uint32), O.centroids_from_knn(pts, idx[sel], cnt[sel]).view(np.uint32)):
                raise AssertionError("centroids differ")
            if not np.array_equal(md[sel].view(np.uint32), O.mean_dist_from_knn(pts, pts[sel], idx[sel], cnt[sel]).view(np.uint32)):
                raise AssertionError("mean neighbour distances differ")
        if n >= 2 and rng.random() < 0.5:  # device orientation == sequential search on the same rows
            nrm0, idx0, cnt0 = ix.normals_knn_self(k, eps, want_knn=True)
            nrm0 = np.nan_to_num(nrm0)
            host, hreach = pkg.propagate_normal_orientations(pts, idx0, nrm0, cnt0)
            dev, dreach = ix.orient_normals_knn_self(nrm0, k, eps)
            if dreach != hreach or not np.array_equal(dev.view(np.uint32), host.view(np.uint32)):
                raise AssertionError("device orientation differs from the sequential search")
        if k <= 32 and n >= 3:
            nrm = ix.normals_knn_self(k, eps)
            onrm = O.normals_from_knn(pts, idx[sel], cnt[sel])
            full = cnt[sel] >= 3
            a, b = nrm[sel][full].astype(np.float64), onrm[full].astype(np.float64)
            good = np.isfinite(a).all(1) & np.isfinite(b).all(1)
            if good.any():
                cos = np.abs((a[good] * b[good]).sum(1))
                # rank-deficient neighbourhoods (lattices, duplicates) have an arbitrary null-space direction: compare bits instead
                if not (np.array_equal(nrm[sel][full][good].view(np.uint32), onrm[full][good].view(np.uint32)) or (1 - cos).max() <= 1e-4):
                    raise AssertionError("normals differ: max 1-|cos| = %g" % (1 - cos).max())
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL", json.dumps(what), repr(e)[:300], flush=True)
    cases += 1
    ix.close()
print(json.dumps({"cases": cases, "failures": fails, "seconds": budget, "seed": seed}))
sys.exit(1 if fails else 0)
