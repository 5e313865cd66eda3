import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so import it by string)."""
    return importlib.import_module("point-cloud-processing_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pcp_oracle
    pcp_oracle.build()
    return pcp_oracle


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bunny(pkg):
    pts, _ = pkg.ply.read_ply(os.path.join(GOLDEN, "stanford_bunny.ply"))
    assert pts.shape == (35947, 3)
    return pts


@pytest.fixture(scope="session")
def bunny_golden():
    return np.load(os.path.join(GOLDEN, "bunny_k15.npz"))


EXTRA_CLOUDS = ("detergent", "spray", "fandisk")  # the reference's other example clouds (examples/data/*.ply)


@pytest.fixture(scope="session", params=EXTRA_CLOUDS)
def extra_cloud(request, pkg):
    """(name, points, golden) of one of the reference's example clouds beside the bunny: scanner noise (detergent), thin
    walls (spray), a CAD shape with sharp edges and many exactly equal distances (fandisk)."""
    pts, _ = pkg.ply.read_ply(os.path.join(GOLDEN, request.param + ".ply"))
    return request.param, pts, np.load(os.path.join(GOLDEN, request.param + "_k15.npz"))


def normals_vs_float64_eigh(pts, idx, cnt, nrm):
    """float64 symmetric eigen-solve of every row's scatter matrix against the float32 normal: (max 1 - |cos| over the rows
    whose smallest eigenvalue is separated by a relative gap >= 1e-3, fraction of rows that are not)."""
    worst, ill = 0.0, 0
    for r in range(len(idx)):
        nb = pts[idx[r, : cnt[r]].astype(np.int64)].astype(np.float64)
        c = nb - nb.mean(0)
        w, v = np.linalg.eigh(c.T @ c)
        if (w[1] - w[0]) / max(w[2], 1e-300) < 1e-3:
            ill += 1
            continue
        worst = max(worst, 1.0 - abs(float(v[:, 0] @ nrm[r].astype(np.float64))))
    return worst, ill / max(1, len(idx))


def points_match(got, expected, eps=1e-5):
    """pcp::common::are_vectors_equal on each row (include/pcp/common/vector3d_queries.hpp:47-64)."""
    got = np.asarray(got, np.float32).reshape(-1, 3)
    expected = np.asarray(expected, np.float32).reshape(-1, 3)
    return got.shape == expected.shape and bool(np.all(np.abs(got - expected) < eps))


def same_point_set(got, expected, eps=1e-5):
    got = np.asarray(got, np.float32).reshape(-1, 3)
    expected = np.asarray(expected, np.float32).reshape(-1, 3)
    if len(got) != len(expected):
        return False
    used = set()
    for g in got:
        hit = [i for i, e in enumerate(expected) if i not in used and np.all(np.abs(g - e) < eps)]
        if not hit:
            return False
        used.add(hit[0])
    return True


def knn_rows_equivalent(xyz, queries, idx_a, cnt_a, idx_b, cnt_b):
    """Tie-aware row comparison (SURVEY.md section 7 'hard parts'): the reference leaves ties at equal
    distance implementation-defined, so rows must have equal counts and equal sorted d2 lists, and
    equal index sets wherever distances are distinct."""
    xyz = np.asarray(xyz, np.float32)
    queries = np.asarray(queries, np.float32)
    if not np.array_equal(cnt_a, cnt_b):
        return False, "counts differ"
    for q in range(len(queries)):
        c = int(cnt_a[q])
        a, b = idx_a[q, :c].astype(np.int64), idx_b[q, :c].astype(np.int64)
        if np.array_equal(a, b):
            continue
        da = _d2(xyz[a], queries[q])
        db = _d2(xyz[b], queries[q])
        if not np.array_equal(da, db):
            return False, "row %d: distance lists differ" % q
    return True, ""


def _d2(p, q):
    d = p - q[None, :]
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def analytic_normal_cases():
    """Point sets whose PCA normal is known in closed form and whose scatter matrix is EXACT in float32, so that what
    is tested is the eigen-solver alone (beyond the reference's axis-aligned 7-point KAT,
    test/common/normal_estimation.cpp:11-38, whose scatter matrix is diagonal and never enters the QR iteration).

    Construction: an orthonormal basis (u, v, w) with rational entries (two Pythagorean rotations: multiples of 1/65),
    points {+-a u +- b v +- c w} over a few (a, b, c) triples that are multiples of 65/64 -- all coordinates are small
    integers / 64, so the mean is exactly 0, every product and partial sum of the scatter matrix is an exact float32, and
    Cov = R diag(8 sum a^2, 8 sum b^2, 8 sum c^2) R^T exactly.  The eigenvector of the smallest of the three is the
    corresponding basis vector.  Yields (name, points float32 (m, 3), unit normal float64 (3,), relative eigen-gap)."""
    import itertools
    from fractions import Fraction as F
    def mat(rows):
        return [[F(x) for x in r] for r in rows]
    def mul(a, b):
        return [[sum(a[i][k] * b[k][j] for k in range(3)) for j in range(3)] for i in range(3)]
    c1, s1, c2, s2 = F(3, 5), F(4, 5), F(5, 13), F(12, 13)
    rz = mat([[c1, -s1, 0], [s1, c1, 0], [0, 0, 1]])
    rx = mat([[1, 0, 0], [0, c2, -s2], [0, s2, c2]])
    ry = mat([[c2, 0, s2], [0, 1, 0], [-s2, 0, c2]])
    bases = {"zx": (mul(rz, rx), 65), "xz": (mul(rx, rz), 65), "zy": (mul(rz, ry), 65), "yxz": (mul(ry, mul(rx, rz)), 845)}
    spreads = {
        "flat": [(8, 6, 1), (4, 7, 0), (2, 3, 1)],          # thin slab: well separated smallest eigenvalue
        "mild": [(8, 6, 5), (4, 7, 4), (2, 3, 5)],          # anisotropic but thick
        "near_tie": [(8, 8, 7), (5, 5, 5), (3, 2, 2)],      # two large eigenvalues close, smallest only ~25 % below
        "needle": [(1, 9, 2), (0, 7, 1), (1, 8, 0)],        # one dominant direction: the normal is one of two small ones
    }
    for bname, (basis, den) in bases.items():
        ib = [[int(basis[r][c] * den) for c in range(3)] for r in range(3)]  # den * R: integers
        assert all(F(ib[r][c], den) == basis[r][c] for r in range(3) for c in range(3))
        for sname, triples in spreads.items():
            rows = []
            for (a, b, c) in triples:
                for sa, sb, sc in itertools.product((-1, 1), repeat=3):
                    rows.append([sa * a * ib[r][0] + sb * b * ib[r][1] + sc * c * ib[r][2] for r in range(3)])
            pts = np.array(rows, dtype=np.int64)  # integer coordinates (the cloud scaled by den); /64 keeps them exact
            assert np.abs(pts).max() < 2 ** 15     # so squares and their sums stay below 2^24 * 64^2: exact in float32
            lam = np.array([8.0 * sum(t[i] ** 2 for t in triples) for i in range(3)])
            order = np.argsort(lam)
            normal = np.array([float(basis[r][order[0]]) for r in range(3)])
            gap = (lam[order[1]] - lam[order[0]]) / lam[order[2]]
            p32 = (pts.astype(np.float64) / 64.0).astype(np.float32)
            assert np.array_equal(p32.astype(np.float64) * 64.0, pts.astype(np.float64)), "coordinates must be exact in float32"
            yield "%s/%s" % (bname, sname), p32, normal / np.linalg.norm(normal), float(gap)
