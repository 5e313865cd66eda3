"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C ABI of
libpcpx.so; the oracle (oracle/pcp_oracle.cpp) is only the checker.

Bars: kNN rows are bit-exact against the (d2, index)-sorted brute-force oracle -- indices, counts and
float32 squared distances; range results are exact sets; normals are bit-comparable with the oracle's
Eigen restatement and within 1e-4 cosine (the tolerance BASELINE.json's north_star states)."""
import importlib

import numpy as np
import pytest

from conftest import knn_rows_equivalent, normals_vs_float64_eigh, points_match, same_point_set

pytestmark = pytest.mark.gpu

COS_TOL = 1e-4  # north_star: "normals within 1e-4 cosine of reference"


def _cos_err(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    num = np.abs((a * b).sum(1))
    den = np.sqrt((a * a).sum(1) * (b * b).sum(1))
    return 1.0 - num / np.maximum(den, 1e-300)


def _check_knn_exact(pkg, oracle, pts, queries, k, eps=1e-5, self_query=False, ix=None):
    ix = ix or pkg.Index(pts)
    if self_query:
        gi, gc, gd = ix.knn_self(k, eps, want_d2=True)
        queries = pts
    else:
        gi, gc, gd = ix.knn(queries, k, eps, want_d2=True)
    oi, oc, od = oracle.knn_bruteforce(pts, queries, k, eps, nthreads=8, want_d2=True)
    _assert_rows_exact(pts, queries, k, gi, gc, gd, oi, oc, od)
    return ix


def _assert_rows_exact(pts, queries, k, gi, gc, gd, oi, oc, od):
    """Counts and float32 squared distances are bit-exact; indices are identical except where several
    points tie EXACTLY with the k-th distance (the reference leaves that choice to heap order,
    linked_octree_node.hpp:479-489): there the returned point must be a real point at that distance
    and the row must still be in (d2, index) order."""
    assert np.array_equal(gc, oc)
    valid = np.arange(k)[None, :] < oc[:, None]
    assert np.array_equal(gd[valid], od[valid])
    assert np.all(gi[~valid] == 0xFFFFFFFF)
    diff = (gi != oi) & valid
    for q in np.nonzero(diff.any(1))[0]:
        c = int(oc[q])
        last = od[q, c - 1]
        cols = np.nonzero(diff[q])[0]
        assert np.all(od[q, cols] == last), "row %d differs away from a k-th distance tie" % q
        ids = gi[q, cols].astype(np.int64)
        d = pts[ids] - np.asarray(queries, np.float32)[q][None, :]
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
        assert np.all(d2 == last), "row %d returns a point that is not at the tied distance" % q
        key = gd[q, :c].astype(np.float64) * 2.0 ** 40 + gi[q, :c].astype(np.float64)
        assert np.all(np.diff(key) > 0), "row %d not in (d2, index) order" % q
        assert len(set(gi[q, :c].tolist())) == c


# ---- the reference's own known-answer tests, through the GPU path ------------------------------------
def test_reference_knn_kats(pkg, kats):
    for case in kats["knn"]:
        pts = np.array(case["points"], np.float32)
        for tree in (pkg.LinkedOctree(pts, voxel_grid=case["voxel_grid"]), pkg.LinkedKdTree(pts)):
            idx, cnt = tree.nearest_neighbours(case["queries"], case["k"], eps=kats["eps"])
            assert list(cnt) == case["expected_counts"], case["name"]
            for q, exp in enumerate(case["expected_points"]):
                assert points_match(pts[idx[q, : cnt[q]]], exp), case["name"]
                assert np.all(idx[q, cnt[q]:] == 0xFFFFFFFF)


def test_reference_range_kats(pkg, kats):
    r = kats["range"]
    pts = np.array(r["points"], np.float32)
    for tree in (pkg.LinkedOctree(pts, voxel_grid=r["voxel_grid"]), pkg.LinkedKdTree(pts)):
        centers = [s["center"] for s in r["spheres"]]
        radii = np.array([s["radius"] for s in r["spheres"]], np.float32)
        off, idx = tree.range_sphere(centers, radii)
        for i, s in enumerate(r["spheres"]):
            assert same_point_set(pts[idx[off[i]:off[i + 1]]], s["expected_points"])
            o1, i1 = tree.range_search([s["center"]], s["radius"])
            assert same_point_set(pts[i1], s["expected_points"])
        boxes = [b["min"] + b["max"] for b in r["aabbs"]]
        off, idx = tree.range_aabb(boxes)
        for i, b in enumerate(r["aabbs"]):
            assert same_point_set(pts[idx[off[i]:off[i + 1]]], b["expected_points"])


def test_reference_insertion_kat(pkg, kats):
    c = kats["octree_insertion"]
    pts = np.array(c["inside"] + c["outside"], np.float32)
    t = pkg.LinkedOctree(pts, voxel_grid=c["voxel_grid"])
    assert t.size() == c["expected_size"]
    assert np.array_equal(t.voxel_grid(), np.array(c["voxel_grid"], np.float32))
    # dropped points are never returned, and their own rows are empty
    idx, cnt = t.knn_self(3)
    assert np.all(cnt[len(c["inside"]):] == 0) and np.all(cnt[: len(c["inside"])] == 3)
    assert idx[: len(c["inside"])].max() < len(c["inside"])


def test_reference_normal_kat(pkg, kats):
    c = kats["normal"]
    n = pkg.estimate_normal(c["points"])
    exp = np.array(c["expected_normal_up_to_sign"], np.float32)
    tol = c["component_tolerance"]
    assert np.all(np.abs(n - exp) < tol) or np.all(np.abs(n + exp) < tol)
    assert abs(float(np.sqrt((n.astype(np.float64) ** 2).sum())) - 1.0) < tol


def test_planted_neighbours(pkg, kats):
    p = kats["planted_knn"]
    rng = np.random.default_rng(7)
    for _ in range(3):
        n = int(rng.integers(*p["size_range"]))
        k = int(rng.integers(p["k_range"][0], p["k_range"][1] + 1))
        bg = rng.uniform(*p["background_range"], (n, 3)).astype(np.float32)
        planted = np.stack([rng.uniform(*p["near_range"], k), rng.uniform(*p["far_range"], k),
                            rng.uniform(*p["far_range"], k)], axis=1).astype(np.float32)
        pts = np.concatenate([bg, planted])
        t = pkg.LinkedOctree(pts, voxel_grid=p["voxel_grid"])
        assert t.size() == n + k
        idx, cnt = t.nearest_neighbours([p["reference_point"]], k)
        assert cnt[0] == k and set(idx[0].tolist()) == set(range(n, n + k))


# ---- golden fixtures ----------------------------------------------------------------------------------
def test_bunny_golden(pkg, bunny, bunny_golden):
    g = bunny_golden
    qi = g["query_index"]
    ix = pkg.LinkedOctree(bunny)
    assert ix.size() == len(bunny)
    nrm, idx, cnt = ix.normals_knn_self(15, want_knn=True)
    assert np.array_equal(idx[qi], g["knn_idx"]) and np.array_equal(cnt[qi], g["knn_cnt"])
    _, _, d2 = ix.knn_self(15, want_d2=True)
    assert np.array_equal(d2[qi], g["knn_d2"])
    assert _cos_err(nrm[qi], g["normals"]).max() <= COS_TOL
    assert np.array_equal(nrm[qi], g["normals"])  # same arithmetic, same order: bit-identical in practice
    assert np.array_equal(ix.range_count_self(0.01)[qi], g["range_count_r001"])
    assert np.array_equal(ix.range_count(bunny[qi], 0.01), g["range_count_r001"])


def test_bunny_full_against_oracle_trees(pkg, oracle, bunny):
    """BASELINE config 1: every bunny point, k=15, against the restated octree and kd-tree (tie-aware)."""
    ix = pkg.Index(bunny)
    gi, gc = ix.knn_self(15)
    for tree in (oracle.Octree(bunny), oracle.KdTree(bunny, compute_max_depth=True)):
        oi, oc = tree.knn(bunny, 15, nthreads=8)
        ok, why = knn_rows_equivalent(bunny, bunny, gi, gc, oi, oc)
        assert ok, why
    on = oracle.normals_from_knn(bunny, gi, gc, nthreads=8)
    gn = ix.normals_knn_self(15)
    assert _cos_err(gn, on).max() <= COS_TOL


def test_other_example_clouds(pkg, oracle, extra_cloud):
    """The reference's other three example inputs (examples/data/detergent.ply, spray.ply, fandisk.ply): every point, k = 15,
    through the octree and the kd-tree wrappers -- distances bit-equal to brute force and to the committed rows, indices equal
    wherever distances are distinct (fandisk is a CAD grid: ties everywhere), both restated reference trees agree, sphere
    counts exact, normals bit-equal to the oracle's on the same rows and within 1e-4 cosine of a float64 eigen-solve."""
    name, pts, g = extra_cloud
    qi = g["query_index"]
    radius = float(g["range_radius"])
    bi, bc, bd = oracle.knn_bruteforce(pts, pts, 15, nthreads=8, want_d2=True)
    for ix in (pkg.LinkedOctree(pts), pkg.LinkedKdTree(pts)):
        assert ix.size() == len(pts)
        nrm, idx, cnt = ix.normals_knn_self(15, want_knn=True)
        _, _, d2 = ix.knn_self(15, want_d2=True)
        assert np.array_equal(cnt, bc) and np.array_equal(d2, bd), name
        assert np.array_equal(d2[qi], g["knn_d2"]) and np.array_equal(cnt[qi], g["knn_cnt"]), name
        ok, why = knn_rows_equivalent(pts, pts, idx, cnt, bi, bc)
        assert ok, name + ": " + why
        for tree in (oracle.Octree(pts), oracle.KdTree(pts, compute_max_depth=True)):
            ti, tc = tree.knn(pts, 15, nthreads=8)
            ok, why = knn_rows_equivalent(pts, pts, idx, cnt, ti, tc)
            assert ok, name + ": " + why
        assert np.array_equal(ix.range_count_self(radius)[qi], g["range_count"]), name
        assert np.array_equal(ix.range_count(pts[qi], radius), g["range_count"]), name
        on = oracle.normals_from_knn(pts, idx, cnt, nthreads=8)
        assert _cos_err(nrm, on).max() <= COS_TOL, name
        worst, ill = normals_vs_float64_eigh(pts, idx[qi], cnt[qi], nrm[qi])
        print("%s: max 1-|cos| vs float64 eigh %.2e, ill-conditioned fraction %.4f" % (name, worst, ill))
        assert worst <= COS_TOL, name
        ix.close()


# ---- seeded random clouds against the brute-force oracle --------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 63, 64, 65, 257, 1000, 4099])
@pytest.mark.parametrize("k", [1, 15, 16, 17, 32, 40])
def test_knn_self_small(pkg, oracle, n, k):
    rng = np.random.default_rng(1000 * n + k)
    pts = rng.random((n, 3), dtype=np.float32)
    _check_knn_exact(pkg, oracle, pts, None, k, self_query=True)


@pytest.mark.parametrize("n,nq,k", [(1, 5, 3), (50, 1, 15), (1000, 777, 15), (20000, 3000, 10), (20000, 130, 32),
                                    (20000, 500, 33), (5000, 200, 64), (3000, 100, 100), (40, 7, 70)])
def test_knn_batch_arbitrary_queries(pkg, oracle, n, nq, k):
    rng = np.random.default_rng(n + nq + k)
    pts = rng.random((n, 3), dtype=np.float32)
    q = (rng.random((nq, 3), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)  # some outside the cloud's box
    _check_knn_exact(pkg, oracle, pts, q, k)


def test_knn_empty_and_k0(pkg):
    ix = pkg.Index(np.empty((0, 3), np.float32))
    assert ix.size() == 0 and ix.empty()
    idx, cnt = ix.knn([[0, 0, 0], [1, 1, 1]], 4)
    assert np.all(cnt == 0) and np.all(idx == 0xFFFFFFFF)
    ix = pkg.Index(np.random.default_rng(0).random((100, 3), dtype=np.float32))
    idx, cnt = ix.knn([[0.5, 0.5, 0.5]], 0)
    assert idx.shape == (1, 0) and cnt[0] == 0
    idx, cnt = ix.knn([[0.5, 0.5, 0.5]], 150)  # k > n: short row
    assert cnt[0] == 100 and np.all(idx[0, 100:] == 0xFFFFFFFF) and len(set(idx[0, :100].tolist())) == 100


def test_knn_coincident_points_and_eps(pkg, oracle):
    """eps-box exclusion removes EVERY point inside the box, not 'the query index'
    (include/pcp/octree/linked_octree_node.hpp:540)."""
    rng = np.random.default_rng(3)
    base = rng.random((500, 3), dtype=np.float32)
    dup = base[:100] + np.float32(3e-6)  # inside the 1e-5 box of their twins
    pts = np.concatenate([base, dup, base[:50]])  # and exact duplicates
    _check_knn_exact(pkg, oracle, pts, None, 8, self_query=True)
    _check_knn_exact(pkg, oracle, pts, None, 8, eps=1e-7, self_query=True)  # duplicates at d2 == 0 stay excluded
    _check_knn_exact(pkg, oracle, pts, None, 8, eps=0.0, self_query=True)   # nothing excluded: self is neighbour 0
    _check_knn_exact(pkg, oracle, pts, base[:64] + np.float32(1e-3), 8, eps=2e-3)


@pytest.mark.parametrize("k", [5, 15, 32])
def test_eps_box_test_in_either_place_gives_the_same_rows(pkg, oracle, k):
    """The throughput kernel applies the eps-box exclusion either to every candidate or to the buffered keys when they are
    folded into the best-list (pcpx_debug_eps_test_mode; the launcher picks by eps against the point spacing).  Both forms,
    forced, against brute force: tiny, ordinary and absurdly large boxes, twins inside the box, self and foreign queries."""
    rng = np.random.default_rng(17)
    base = rng.random((6000, 3), dtype=np.float32)
    twins = base[:1500] + rng.uniform(-4e-6, 4e-6, (1500, 3)).astype(np.float32)
    pts = np.concatenate([base, twins, base[:300]])
    foreign = np.concatenate([rng.random((700, 3), dtype=np.float32), base[:300] + np.float32(2e-6)])
    ix = pkg.Index(pts)
    for mode in (1, 2, 0):
        ix.debug_eps_test_mode(mode)
        for eps in (0.0, 1e-7, 1e-5, 1e-3, 0.04, 0.3):
            _check_knn_exact(pkg, oracle, pts, None, k, eps=eps, self_query=True, ix=ix)
            _check_knn_exact(pkg, oracle, pts, foreign, k, eps=eps, ix=ix)
    with pytest.raises(Exception):
        ix.debug_eps_test_mode(3)


def test_knn_lattice_ties(pkg, oracle):
    """A lattice makes exact distance ties the norm; (d2, index) ordering must still match exactly, and
    the reference trees must agree up to ties."""
    g = np.arange(12, dtype=np.float32) / np.float32(8)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(0)
    pts = pts[rng.permutation(len(pts))]
    ix = _check_knn_exact(pkg, oracle, pts, None, 15, self_query=True)
    gi, gc = ix.knn_self(15)
    oi, oc = oracle.Octree(pts).knn(pts, 15, nthreads=8)
    ok, why = knn_rows_equivalent(pts, pts, gi, gc, oi, oc)
    assert ok, why


def test_knn_clustered_and_degenerate(pkg, oracle):
    c = pkg.synthetic.clustered_cloud(30000, seed=44)
    _check_knn_exact(pkg, oracle, c, c[::37], 15)
    rng = np.random.default_rng(9)
    plane = rng.random((3000, 3), dtype=np.float32)
    plane[:, 2] = 0.25  # zero extent on one axis
    _check_knn_exact(pkg, oracle, plane, None, 15, self_query=True)
    line = np.zeros((2000, 3), np.float32)
    line[:, 0] = rng.random(2000, dtype=np.float32) * 100 - 50
    _check_knn_exact(pkg, oracle, line, None, 5, self_query=True)
    big = (rng.random((3000, 3), dtype=np.float32) * 200 - 100).astype(np.float32)  # benchmark range U(-100,100)
    _check_knn_exact(pkg, oracle, big, None, 10, self_query=True)


@pytest.mark.parametrize("k", [7, 15, 32])
def test_knn_when_few_lanes_of_a_group_need_a_leaf(pkg, oracle, k):
    """Tight blobs with stragglers between them: in most 64-query groups a few lanes reach leaves no other lane needs, which
    k_knn looks at point-per-lane for those lanes only (pcpx_query.hip: sparse_leaf); eight-fold coincident points make such
    a leaf fill a lane's append buffer in one go."""
    rng = np.random.default_rng(77 + k)
    centres = rng.random((60, 3), dtype=np.float32)
    blobs = (centres[rng.integers(0, 60, 36000)] + rng.normal(0, 2e-3, (36000, 3))).astype(np.float32)
    stragglers = rng.random((4000, 3), dtype=np.float32)
    repeated = np.repeat(rng.random((250, 3), dtype=np.float32), 8, axis=0)
    pts = np.concatenate([blobs, stragglers, repeated])[rng.permutation(42000)]
    ix = _check_knn_exact(pkg, oracle, pts, None, k, self_query=True)
    queries = np.concatenate([stragglers[:900] + np.float32(1e-3), blobs[:900], rng.random((248, 3), dtype=np.float32) * 3 - 1])
    _check_knn_exact(pkg, oracle, pts, queries[rng.permutation(len(queries))], k, eps=0.0, ix=ix)


def test_rebuild_reuses_handle(pkg, oracle):
    rng = np.random.default_rng(21)
    a = rng.random((5000, 3), dtype=np.float32)
    ix = pkg.Index(a)
    _check_knn_exact(pkg, oracle, a, None, 15, self_query=True, ix=ix)
    b = pkg.synthetic.jitter(a, seed=46)
    ix.rebuild(b)
    _check_knn_exact(pkg, oracle, b, None, 15, self_query=True, ix=ix)
    c = rng.random((9000, 3), dtype=np.float32)  # grows past the first capacity
    ix.rebuild(c)
    _check_knn_exact(pkg, oracle, c, None, 15, self_query=True, ix=ix)


# ---- radius search -------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,r", [(1, 0.5), (100, 0.3), (5000, 0.05), (50000, 0.03), (5000, 1.0)])
def test_range_count_and_lists(pkg, oracle, n, r):
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3), dtype=np.float32)
    ix = pkg.Index(pts)
    exp = oracle.range_count_bruteforce(pts, pts, r, nthreads=8)
    assert np.array_equal(ix.range_count_self(r), exp)
    q = (rng.random((300, 3), dtype=np.float32) * 1.2 - 0.1).astype(np.float32)
    expq = oracle.range_count_bruteforce(pts, q, r, nthreads=8)
    assert np.array_equal(ix.range_count(q, r), expq)
    off, idx = ix.range_sphere(q, r)
    assert np.array_equal(np.diff(off).astype(np.uint32), expq)
    tree = oracle.Octree(pts)
    for i in range(0, 300, 29):
        assert sorted(idx[off[i]:off[i + 1]].tolist()) == sorted(tree.range_sphere(q[i], r).tolist())


def test_single_range_calls_take_the_latency_path_and_agree(pkg, oracle):
    """One sphere or one box per call (the shape of the reference's range_search and of its benchmark) goes through a one-wavefront
    kernel and the pinned stage; the lists equal the batch form's, in the same order, for empty results, ordinary ones, and ones
    larger than the stage (4 096 indices: the call falls back to the batch form)."""
    rng = np.random.default_rng(17)
    pts = rng.random((60000, 3), dtype=np.float32)
    ix = pkg.Index(pts)
    centers = rng.random((24, 3), dtype=np.float32)
    centers[0] = [5, 5, 5]  # nothing there
    for r in (0.0, 0.02, 0.3):  # 0.3: ~6 800 points per sphere, more than the stage holds
        off, idx = ix.range_sphere(centers, r)
        for i in range(len(centers)):
            o1, i1 = ix.range_sphere(centers[i:i + 1], r)
            assert int(o1[1]) == int(off[i + 1] - off[i])
            assert np.array_equal(i1, idx[int(off[i]):int(off[i + 1])])
    half = np.array([0.01, 0.05, 0.4], np.float32)
    for h in half:
        boxes = np.concatenate([centers - h, centers + h], axis=1).astype(np.float32)
        off, idx = ix.range_aabb(boxes)
        for i in range(len(boxes)):
            o1, i1 = ix.range_aabb(boxes[i:i + 1])
            assert int(o1[1]) == int(off[i + 1] - off[i])
            assert np.array_equal(i1, idx[int(off[i]):int(off[i + 1])])
    one = pkg.Index(pts[:1])  # a one-point index: the root is the only leaf
    o1, i1 = one.range_sphere(pts[:1], 0.1)
    assert list(i1) == [0] and int(o1[1]) == 1


def test_ranges_wider_than_one(pkg, oracle):
    """radius > 1, the one place where results knowingly differ from the reference: its box-sphere test compares the SQUARED
    distance with `radius` (include/pcp/common/intersections.hpp:87-102, :113-130), so its trees prune boxes between
    sqrt(radius) and radius away.  The GPU returns the geometric range.  Shown here: GPU == brute force; the reference's
    octree and kd-tree (as restated) return subsets of it; and the points they miss are exactly the points the same trees
    find once that one comparison is made against radius^2 -- nothing else differs."""
    rng = np.random.default_rng(31)
    pts = rng.uniform(-6, 6, (30000, 3)).astype(np.float32)
    q = rng.uniform(-6, 6, (40, 3)).astype(np.float32)
    r = 2.5
    ix = pkg.Index(pts)
    exp = oracle.range_count_bruteforce(pts, q, r, nthreads=8)
    assert np.array_equal(ix.range_count(q, r), exp)
    off, idx = ix.range_sphere(q, r)
    assert np.array_equal(np.diff(off).astype(np.uint32), exp)
    d = pts[None, :, :].astype(np.float32) - q[:, None, :]
    d2 = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    missed_total = 0
    for tree in (oracle.Octree(pts), oracle.KdTree(pts)):
        for i in range(len(q)):
            gpu = set(idx[off[i]:off[i + 1]].tolist())
            assert gpu == set(np.nonzero(d2[i] <= np.float32(r) * np.float32(r))[0].tolist())
            ref = set(tree.range_sphere(q[i], r).tolist())
            oracle.set_geometric_prune(True)
            try:
                fixed = set(tree.range_sphere(q[i], r).tolist())
            finally:
                oracle.set_geometric_prune(False)
            assert ref <= gpu and fixed == gpu
            missed_total += len(gpu - ref)
    assert missed_total > 0  # (the case is not vacuous: at this radius the reference's prune does drop points)


def test_range_aabb_against_oracle(pkg, oracle):
    rng = np.random.default_rng(4)
    pts = rng.random((20000, 3), dtype=np.float32)
    lo = rng.random((50, 3), dtype=np.float32) * 0.9
    hi = lo + rng.random((50, 3), dtype=np.float32) * 0.2
    ix = pkg.Index(pts)
    off, idx = ix.range_aabb(np.concatenate([lo, hi], 1))
    tree = oracle.KdTree(pts)
    for i in range(50):
        assert sorted(idx[off[i]:off[i + 1]].tolist()) == sorted(tree.range_aabb(lo[i], hi[i]).tolist())


# ---- normals -------------------------------------------------------------------------------------------
def test_normals_match_oracle_bitwise_on_random_cloud(pkg, oracle):
    rng = np.random.default_rng(12)
    pts = rng.uniform(-10, 10, (20000, 3)).astype(np.float32)
    ix = pkg.Index(pts)
    nrm, idx, cnt = ix.normals_knn_self(15, want_knn=True)
    on, oev = oracle.normals_from_knn(pts, idx, cnt, nthreads=8, want_evals=True)
    assert _cos_err(nrm, on).max() <= COS_TOL
    assert np.mean(np.all(nrm == on, axis=1)) > 0.999
    n2, ev = ix.normals_from_knn(idx, cnt, want_evals=True)
    assert np.array_equal(n2, nrm)
    assert np.allclose(ev, oev, rtol=1e-5, atol=1e-6)
    assert np.all(np.abs(np.sqrt((nrm.astype(np.float64) ** 2).sum(1)) - 1) < 1e-5)


def test_normals_of_a_plane_and_short_neighbourhoods(pkg):
    rng = np.random.default_rng(2)
    pts = rng.random((4000, 3), dtype=np.float32)
    pts[:, 2] = 0.5 + 0.25 * pts[:, 0]  # plane z = 0.5 + x/4, normal ~ (-0.25, 0, 1)/|.|
    nrm = pkg.Index(pts).normals_knn_self(15)
    exp = np.array([-0.25, 0, 1.0]) / np.sqrt(1 + 0.0625)
    assert (1 - np.abs(nrm.astype(np.float64) @ exp)).max() < 1e-4
    few = rng.random((3, 3), dtype=np.float32)  # k > n - 1: rows are short, normals still finite
    n3, idx, cnt = pkg.Index(few).normals_knn_self(15, want_knn=True)
    assert list(cnt) == [2, 2, 2] and np.isfinite(n3).all()


# ---- BASELINE-size runs: size-independent properties + sampled oracle checks ------------------------
def _properties(pts, idx, cnt, d2, k):
    assert np.all(cnt == k)
    assert np.all(np.diff(d2.astype(np.float64), axis=1) >= 0)  # ascending
    assert not np.any(idx == np.arange(len(pts), dtype=np.uint32)[:, None])  # the query itself is excluded
    assert idx.max() < len(pts)
    # d2 really is the reference's float expression for the returned index
    sel = np.random.default_rng(0).integers(0, len(pts), 2000)
    for j in (0, k - 1):
        d = pts[idx[sel, j]] - pts[sel]
        ref = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        assert np.array_equal(ref.astype(np.float32), d2[sel, j])


def test_config2_uniform_1m(pkg, oracle):
    """BASELINE configs[1]: 1 M uniform points, k=15 kNN + normals."""
    pts = pkg.synthetic.uniform_cloud(1_000_000, 42)
    ix = pkg.Index(pts)
    nrm, idx, cnt = ix.normals_knn_self(15, want_knn=True)
    _, _, d2 = ix.knn_self(15, want_d2=True)
    _properties(pts, idx, cnt, d2, 15)
    sel = np.random.default_rng(1).integers(0, len(pts), 3000)
    tree = oracle.Octree(pts)  # reference defaults: capacity 32, depth 21, auto bbox
    oi, oc = tree.knn(pts[sel], 15, nthreads=8)
    ok, why = knn_rows_equivalent(pts, pts[sel], idx[sel], cnt[sel], oi, oc)
    assert ok, why
    on, oev = oracle.normals_from_knn(pts, idx[sel], cnt[sel], nthreads=8, want_evals=True)
    err = _cos_err(nrm[sel], on)
    assert err.max() <= COS_TOL
    # symmetric check of a neighbour relation through range counts: |ball(r_k)| >= k + 1
    rk = np.sqrt(d2[sel, -1].astype(np.float64)).astype(np.float32)
    off, _ = ix.range_sphere(pts[sel[:200]], rk[:200] * np.float32(1.0001))
    assert np.all(np.diff(off) >= 16)


def test_config3_range_10m_and_knn_10m_properties(pkg, oracle):
    """BASELINE configs[2] (10 M points, r = 0.01 counts) and the 10 M kNN target config: properties at
    full size plus brute-force checks on sampled queries."""
    pts = pkg.synthetic.uniform_cloud(10_000_000, 43)
    ix = pkg.Index(pts)
    assert ix.size() == len(pts)
    cnt = ix.range_count_self(0.01)
    assert cnt.min() >= 1  # the query point itself is inside its sphere
    assert abs(cnt.mean() - 1e7 * 4 / 3 * np.pi * 1e-6) < 2.5  # E ~ 41.9 minus boundary effects
    sel = np.random.default_rng(2).integers(0, len(pts), 2048)
    assert np.array_equal(cnt[sel], oracle.range_count_bruteforce(pts, pts[sel], 0.01, nthreads=8))
    idx, kc, d2 = ix.knn_self(15, want_d2=True)
    _properties(pts, idx, kc, d2, 15)
    oi, oc, od = oracle.knn_bruteforce(pts, pts[sel], 15, nthreads=8, want_d2=True)
    _assert_rows_exact(pts, pts[sel], 15, idx[sel], kc[sel], d2[sel], oi, oc, od)


@pytest.mark.parametrize("n,k", [(3000, 15), (600_000, 15), (600_000, 40)])
def test_rows_in_curve_order_equal_rows_in_input_order(pkg, n, k):
    """pcpx_normals_knn_self_curve_order (rows computed and copied slice by slice along the curve; 600 000 points = 8 slices)
    against the input-order form: the same rows, counts and normals bit for bit, found through the position table."""
    pts = pkg.synthetic.clustered_cloud(n, seed=7)
    ix = pkg.Index(pts)
    nrm, idx, cnt = ix.normals_knn_self(k, want_knn=True)
    cn, ci, cc, perm, pos = ix.normals_knn_self_curve_order(k)
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32)) and np.array_equal(perm[pos], np.arange(n, dtype=np.uint32))
    assert np.array_equal(ci[pos], idx) and np.array_equal(cc[pos], cnt)
    assert np.array_equal(cn[pos].view(np.uint32), nrm.view(np.uint32))
    ix.close()
    # points outside the voxel grid have no row: their position is 0xFFFFFFFF, the others' rows are those of the input-order form
    grid = [0.25, 0.25, 0.25, 0.75, 0.75, 0.75]
    u = pkg.synthetic.uniform_cloud(20000, 3)
    ix = pkg.Index(u, voxel_grid=grid)
    inside = np.all((u >= 0.25) & (u <= 0.75), axis=1)
    nrm, idx, cnt = ix.normals_knn_self(9, want_knn=True)
    cn, ci, cc, perm, pos = ix.normals_knn_self_curve_order(9)
    assert ix.size() == int(inside.sum()) == len(perm)
    assert np.all(pos[~inside] == 0xFFFFFFFF) and np.array_equal(perm[pos[inside]], np.nonzero(inside)[0].astype(np.uint32))
    assert np.array_equal(ci[pos[inside]], idx[inside]) and np.array_equal(cc[pos[inside]], cnt[inside])
    ix.close()


def test_coarse_order_build_gives_the_same_results(pkg):
    """PCPX_BUILD_COARSE_ORDER (top-level buckets sorted only as deep as their size asks for) changes the order of points inside small
    cells, never a query result: rows, distances, counts and normals equal the default build's bit for bit, on a uniform and on a
    clustered cloud, through creation and through an in-place rebuild; arbitrary query batches too (their seeds are looked up in the
    coarser order)."""
    for pts in (pkg.synthetic.uniform_cloud(300_000, 5), pkg.synthetic.clustered_cloud(300_000, seed=6)):
        a = pkg.Index(pts)
        b = pkg.Index(pts, coarse_order=True)
        na, ia, ca = a.normals_knn_self(15, want_knn=True)
        nb, ib, cb = b.normals_knn_self(15, want_knn=True)
        _, _, da = a.knn_self(15, want_d2=True)
        _, _, db = b.knn_self(15, want_d2=True)
        assert np.array_equal(ca, cb) and np.array_equal(da, db)
        same = np.all(ia == ib, axis=1)
        assert same.mean() > 0.999  # (rows may differ only where points tie exactly with the k-th distance)
        assert np.array_equal(na[same].view(np.uint32), nb[same].view(np.uint32))
        assert np.array_equal(a.range_count_self(0.02), b.range_count_self(0.02))
        q = np.random.default_rng(9).random((5000, 3), dtype=np.float32)
        qa, qb = a.knn(q, 9, want_d2=True), b.knn(q, 9, want_d2=True)
        assert np.array_equal(qa[1], qb[1]) and np.array_equal(qa[2], qb[2])
        b.rebuild(pts[::-1].copy(), coarse_order=True)
        nb2, ib2, cb2 = b.normals_knn_self(15, want_knn=True)
        assert np.array_equal(cb2[::-1], ca)
        a.close()
        b.close()


def test_sorted_shards_cover_the_cloud(pkg):
    """The per-rank query shards of the multi-GPU path (pcpx_shard_range + *_dev sorted slices): the union
    of the shards' rows equals the single-call result."""
    import torch
    pts = pkg.synthetic.clustered_cloud(200_000, seed=44)
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), len(pts))
    k = 15
    full_idx = torch.full((len(pts), k), -1, dtype=torch.int32, device=dev)
    full_cnt = torch.zeros(len(pts), dtype=torch.int32, device=dev)
    full_n = torch.zeros((len(pts), 3), dtype=torch.float32, device=dev)
    ix.normals_knn_self_dev(k, 1e-5, full_n.data_ptr(), full_idx.data_ptr(), full_cnt.data_ptr())
    ix.synchronize()
    sh_idx = torch.full_like(full_idx, -1)
    sh_cnt = torch.zeros_like(full_cnt)
    sh_n = torch.zeros_like(full_n)
    covered = 0
    for rank in range(8):
        first, count = pkg.shard_range(ix.size(), rank, 8)
        assert first % 64 == 0
        ix.normals_knn_self_dev(k, 1e-5, sh_n.data_ptr(), sh_idx.data_ptr(), sh_cnt.data_ptr(), first, count)
        covered += count
    ix.synchronize()
    assert covered == len(pts)
    assert torch.equal(full_idx, sh_idx) and torch.equal(full_cnt, sh_cnt) and torch.equal(full_n, sh_n)
    host_idx, host_cnt = pkg.Index(pts).knn_self(k)
    assert np.array_equal(full_idx.cpu().numpy().view(np.uint32), host_idx)


# ---- build pipeline: the hand-written radix sort -------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 100_000, 1_000_003, 5_000_001])
@pytest.mark.parametrize("first_bit", [0, 24])
def test_radix_sort_is_a_stable_sort(pkg, n, first_bit):
    """The build's onesweep radix sort (pcpx_sort.hip): words ordered by their bits [first_bit, 64), input order kept
    among words that agree on those bits -- against numpy's stable argsort."""
    import ctypes as C
    import importlib
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    lib = capi.load()
    rng = np.random.default_rng(n + first_bit)
    # few distinct values in every byte position => many ties => stability is exercised in every pass
    keys = rng.integers(0, 7, n, dtype=np.uint64) * np.uint64(0x0101010101010101) + (rng.integers(0, 3, n, dtype=np.uint64) << np.uint64(40))
    if n > 10:
        keys[rng.integers(0, n, n // 10)] = np.uint64(0xFFFFFFFFFFFFFFFF)  # outside-the-grid words
        keys[rng.integers(0, n, n // 10)] = rng.integers(0, 2 ** 63, n // 10, dtype=np.uint64)
    if first_bit:  # low bits = the element's index, like the build's words: what stability is observed through
        keys = (keys & ~np.uint64((1 << first_bit) - 1)) | (np.arange(n, dtype=np.uint64) & np.uint64((1 << first_bit) - 1))
    out = np.empty(n, np.uint64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    capi.check(lib.pcpx_debug_sort_keys(vp(keys), n, first_bit, 0, vp(out)))
    order = np.argsort(keys >> np.uint64(first_bit), kind="stable")
    assert np.array_equal(out, keys[order])


# ---- the callers right next to the normal loop ----------------------
def test_tangent_planes_and_mean_distances(pkg, oracle, bunny):
    """estimate_tangent_planes (plane = centroid of the k-neighbourhood + PCA normal) and
    average_distances_to_neighbors, fused into the kNN kernel, against the oracle's restatement."""
    # k = 24: the KCAP-32 kernel; k = 40: the multi-pass path, where the products come from the finished rows
    for pts, k in ((bunny, 15), (pkg.synthetic.clustered_cloud(50_000, seed=44), 9), (bunny[::3], 24), (bunny[::5], 40)):
        ix = pkg.Index(pts)
        idx, cnt = ix.knn_self(k)
        cen, nrm = ix.tangent_planes_knn_self(k)
        md = ix.mean_knn_distance_self(k)
        assert np.array_equal(cen, oracle.centroids_from_knn(pts, idx, cnt))
        assert _cos_err(nrm, oracle.normals_from_knn(pts, idx, cnt, nthreads=8)).max() <= COS_TOL
        assert np.array_equal(nrm, ix.normals_knn_self(k))
        assert np.array_equal(md, oracle.mean_dist_from_knn(pts, pts, idx, cnt))
        # the reference's scalar: average of the per-point means (std::reduce order is unspecified)
        assert abs(float(md.astype(np.float64).mean()) - float(oracle.mean_dist_from_knn(pts, pts, idx, cnt).astype(np.float64).mean())) < 1e-9
    # the 7-point plane KAT of test/common/plane3d.cpp:33-69: centroid (0,0,0), normal +-(0,0,1)
    seven = np.array([[0, 0, 0], [-2, 0, 0], [2, 0, 0], [0, -2, 0], [0, 2, 0], [0, 0, -1], [0, 0, 1]], np.float32)
    far = np.array([[100, 100, 100]], np.float32)  # query whose 7 neighbours are exactly the 7 points
    ix = pkg.Index(np.concatenate([seven, far]))
    cen, nrm = ix.tangent_planes_knn_self(7)
    assert np.all(np.abs(cen[7]) < 1e-6) and abs(abs(nrm[7][2]) - 1) < 1e-5


def test_large_k_multipass(pkg, oracle):
    """k > 32 runs ceil(k/32) passes with a key lower bound and stitches the rows; ties across a pass
    boundary (lattice) and normals over long rows included."""
    g = np.arange(9, dtype=np.float32) / np.float32(8)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[np.random.default_rng(1).permutation(len(lattice))]
    _check_knn_exact(pkg, oracle, lattice, None, 50, self_query=True)
    pts = pkg.synthetic.clustered_cloud(30_000, seed=44)
    ix = _check_knn_exact(pkg, oracle, pts, None, 48, self_query=True)
    nrm, idx, cnt = ix.normals_knn_self(48, want_knn=True)
    assert _cos_err(nrm, oracle.normals_from_knn(pts, idx, cnt, nthreads=8)).max() <= COS_TOL


# ---- normal orientation (propagate_normal_orientations) ---------------------------------------------------------
def test_normal_orientation_kat_through_the_abi(pkg, kats):
    """test/algorithm/estimate_normals.cpp:67-155: GPU kNN rows (k = 2) + the ABI's breadth-first pass."""
    c = kats["normal_orientation"]
    pts = np.array(c["points"], np.float32)
    ix = pkg.Index(pts, voxel_grid=c["voxel_grid"])
    idx, cnt = ix.knn_self(c["k"])[:2]
    out, reached = pkg.propagate_normal_orientations(pts, idx, np.array(c["normals"], np.float32), cnt)
    assert reached == len(pts)
    assert np.all(np.abs(out - np.array(c["expected_normal"], np.float32)) < c["component_tolerance"])


def test_bunny_orientation_matches_oracle(pkg, oracle, bunny, bunny_golden):
    """configs[0] cloud: normals and 15-NN rows from the fused kernel, orientation by the ABI; bit-identical to the
    oracle's pass over the same rows, and to the committed flip bits wherever the rows equal the oracle's."""
    ix = pkg.Index(bunny)
    nrm, idx, cnt = ix.normals_knn_self(15, want_knn=True)
    out, reached = pkg.propagate_normal_orientations(bunny, idx, nrm, cnt)
    ref, ref_reached = oracle.propagate_normal_orientations(bunny, idx, cnt, nrm)
    assert reached == ref_reached == len(bunny)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    oidx, ocnt = oracle.knn_bruteforce(bunny, bunny, 15, nthreads=8)[:2]
    onrm = oracle.normals_from_knn(bunny, oidx, ocnt)
    # the golden flip bits were generated from the oracle's rows and normals: the GPU must reproduce both exactly on
    # this cloud (no exact k-th-distance ties on the bunny), otherwise the comparison below would be vacuous
    assert np.array_equal(oidx, idx) and np.array_equal(onrm.view(np.uint32), nrm.view(np.uint32))
    flipped = np.any(np.signbit(out) != np.signbit(nrm), axis=1)
    assert np.array_equal(np.packbits(flipped), bunny_golden["orientation_flipped"])


def test_orientation_rejects_bad_rows(pkg):
    pts = np.zeros((4, 3), np.float32)
    rows = np.array([[1], [2], [3], [9]], np.uint32)
    with pytest.raises(pkg.PcpxError):
        pkg.propagate_normal_orientations(pts, rows, np.zeros((4, 3), np.float32))


def _host_orientation(pkg, pts, k):
    ix = pkg.Index(pts)
    nrm, idx, cnt = ix.normals_knn_self(k, want_knn=True)
    out, reached = pkg.propagate_normal_orientations(pts, idx, nrm, cnt)
    return ix, nrm, idx, cnt, out, reached


@pytest.mark.parametrize("case", ["bunny", "clustered", "uniform", "tiny"])
def test_device_orientation_equals_the_sequential_search(pkg, bunny, case):
    """The level-synchronous GPU search must reproduce the reference's queue order: bit-identical flips."""
    pts, k = {"bunny": (bunny, 15), "clustered": (pkg.synthetic.clustered_cloud(50_000, seed=44), 9),
              "uniform": (pkg.synthetic.uniform_cloud(200_000, 5), 15), "tiny": (pkg.synthetic.uniform_cloud(10, 3), 15)}[case]
    ix, nrm, idx, cnt, host, host_reached = _host_orientation(pkg, pts, k)
    dev, didx, dcnt, reached = ix.oriented_normals_knn_self(k, want_knn=True)
    assert np.array_equal(didx, idx) and np.array_equal(dcnt, cnt)
    assert reached == host_reached
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))
    only_normals, reached2 = ix.oriented_normals_knn_self(k)
    assert reached2 == reached and np.array_equal(only_normals.view(np.uint32), host.view(np.uint32))


def test_device_orientation_on_a_disconnected_graph(pkg):
    """Two far clusters, small k: the search cannot leave the root's cluster; the other normals stay as estimated."""
    a = pkg.synthetic.uniform_cloud(3000, 1) * 0.1
    b = pkg.synthetic.uniform_cloud(2000, 2) * 0.1 + np.float32(5.0)
    pts = np.concatenate([a, b]).astype(np.float32)
    ix, nrm, idx, cnt, host, host_reached = _host_orientation(pkg, pts, 6)
    dev, reached = ix.oriented_normals_knn_self(6)
    assert reached == host_reached and 0 < reached <= 2000  # the root (largest z) lies in cluster b
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))
    assert np.array_equal(dev[:3000].view(np.uint32), nrm[:3000].view(np.uint32))


def test_device_orientation_kat_and_bad_rows(pkg, kats):
    torch = pytest.importorskip("torch")
    c = kats["normal_orientation"]
    pts = np.array(c["points"], np.float32)
    ix = pkg.Index(pts, voxel_grid=c["voxel_grid"])
    idx, cnt = ix.knn_self(c["k"])[:2]
    dev = torch.device("cuda", 0)
    d_pts, d_idx = torch.from_numpy(pts).to(dev), torch.from_numpy(idx.astype(np.int32)).to(dev)
    d_cnt, d_nrm = torch.from_numpy(cnt.astype(np.int32)).to(dev), torch.tensor(c["normals"], dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    reached, levels = pkg.propagate_normal_orientations_dev(d_pts.data_ptr(), len(pts), d_idx.data_ptr(), d_cnt.data_ptr(), c["k"],
                                                             d_nrm.data_ptr())
    assert reached == len(pts) and levels >= 2
    assert np.all(np.abs(d_nrm.cpu().numpy() - np.array(c["expected_normal"], np.float32)) < c["component_tolerance"])
    d_bad = torch.full((len(pts), c["k"]), 99, dtype=torch.int32, device=dev)
    with pytest.raises(pkg.PcpxError):
        pkg.propagate_normal_orientations_dev(d_pts.data_ptr(), len(pts), d_bad.data_ptr(), d_cnt.data_ptr(), c["k"], d_nrm.data_ptr())


def test_orientation_of_caller_normals(pkg, bunny):
    """pcpx_orient_normals_knn_self: arbitrary (here random unit) normals, graph built on the GPU."""
    rng = np.random.default_rng(11)
    nrm = rng.normal(size=bunny.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    ix = pkg.Index(bunny)
    idx, cnt = ix.knn_self(10)[:2]
    host, host_reached = pkg.propagate_normal_orientations(bunny, idx, nrm, cnt)
    dev, reached = ix.orient_normals_knn_self(nrm, 10)
    assert reached == host_reached
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))


def test_zero_radius_cap_terminates(pkg, oracle):
    """eps = 0 and many coincident points: most lanes of a wave have k neighbours at distance 0, so the first-round
    radius cap of the wave is 0 and cannot grow by multiplication -- the lanes without enough duplicates must still
    finish (found by tests/fuzz_parity.py: the kernel used to loop forever here)."""
    rng = np.random.default_rng(4)
    base = rng.random((300, 3), dtype=np.float32)
    pts = np.concatenate([np.repeat(base, 20, axis=0), rng.random((1000, 3), dtype=np.float32)]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    ix = pkg.Index(pts)
    for k in (15, 19, 24):
        idx, cnt = ix.knn_self(k, eps=0.0)[:2]
        sel = rng.choice(len(pts), 400, replace=False)
        oi, oc = oracle.knn_bruteforce(pts, pts[sel], k, eps=0.0, nthreads=8)[:2]
        ok, why = knn_rows_equivalent(pts, pts[sel], idx[sel], cnt[sel], oi, oc)
        assert ok, why


def test_short_randomised_parity_run():
    """tests/fuzz_parity.py for a few seconds with a fixed seed (its long runs are done by hand: DESIGN.md section 2)."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "8", "99"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]


def test_randomised_parity_run_on_large_clouds():
    """tests/fuzz_parity.py in its `big` mode for 60 s with a fixed seed: 1 - 5 M points per case (deep trees, several rounds of
    query groups per resident wave), six cloud shapes, every entry point against brute force on sampled queries."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "60", "314", "big"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    summary = out.stdout.strip().splitlines()[-1]
    print(summary)
    assert '"failures": 0' in summary


def test_error_behaviour_of_the_abi(pkg):
    """Bad arguments come back as status codes with a message, never as a crash (SURVEY.md section 8(b) "Errors")."""
    import ctypes as C
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    lib = capi.load()
    pts = pkg.synthetic.uniform_cloud(1000, 1)
    ix = pkg.Index(pts)
    h = ix._h
    out_idx = np.empty((1000, 4), np.uint32)
    out_cnt = np.empty(1000, np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    # null outputs, null handle, a device that does not exist
    assert lib.pcpx_knn_self(h, 4, 1e-5, None, None, None) == capi.PCPX_ERR_INVALID
    assert lib.pcpx_knn_self(None, 4, 1e-5, vp(out_idx), vp(out_cnt), None) == capi.PCPX_ERR_INVALID
    box = np.empty(6, np.float32)
    assert lib.pcpx_bounding_box(vp(pts), 1000, 99, box.ctypes.data_as(C.POINTER(C.c_float))) == capi.PCPX_ERR_INVALID
    assert b"device" in lib.pcpx_last_error()
    # k = 0 is "no neighbours", like the reference (linked_octree_node.hpp:464): status OK, counts 0
    out_cnt[:] = 7
    assert lib.pcpx_knn_self(h, 0, 1e-5, vp(out_idx), vp(out_cnt), None) == capi.PCPX_OK
    assert np.all(out_cnt == 0)
    assert lib.pcpx_knn_self(h, 0, 1e-5, None, vp(out_cnt), None) == capi.PCPX_OK  # no row storage needed for k = 0
    # the sorted-slice forms want a 64-aligned start
    assert lib.pcpx_knn_self_dev(h, 4, 1e-5, 3, 64, None, None, None) == capi.PCPX_ERR_INVALID
    # range lists: too small a buffer reports the needed size through the offsets and PCPX_ERR_CAPACITY
    off = np.zeros(3, np.uint64)
    centers = pts[:2].copy()
    small = np.empty(1, np.uint32)
    st = lib.pcpx_range_sphere_batch(h, vp(centers), None, 0.5, 2, vp(off), vp(small), 1)
    assert st == capi.PCPX_ERR_CAPACITY and off[2] > 1
    big = np.empty(int(off[2]), np.uint32)
    assert lib.pcpx_range_sphere_batch(h, vp(centers), None, 0.5, 2, vp(off), vp(big), len(big)) == capi.PCPX_OK
    # an explicit voxel grid drops outside points silently (linked_octree_node.hpp:174-175) ...
    half = pkg.Index(pts, voxel_grid=[0, 0, 0, 0.5, 1, 1])
    assert 0 < half.size() < 1000
    # ... and the all-in-one orientation call refuses such an index instead of inventing neighbourhoods
    nrm = np.empty((1000, 3), np.float32)
    assert lib.pcpx_oriented_normals_knn_self(half._h, 5, 1e-5, vp(nrm), None, None, None) == capi.PCPX_ERR_UNSUPPORTED
    ix.close()
    half.close()


def test_aabb_and_mean_distance_kats(pkg, kats):
    """test/common/aabb.cpp and test/algorithm/average_distance_to_neighbors.cpp through the ABI."""
    tol = kats["eps"]
    for case in kats["aabb"]["cases"]:
        pts = np.array(case["points"], np.float32)
        b = pkg.bounding_box(pts)
        assert np.array_equal(b[:3], pts.min(0)) and np.array_equal(b[3:], pts.max(0))
        ix = pkg.Index(pts)
        assert np.array_equal(ix.bbox(), b)
        # bounds are inclusive (axis_aligned_bounding_box.hpp:111-125): the box range over the bounding box is everything
        assert len(ix.range_aabb(b[None, :])[1]) == len(pts)
        for q, inside in case["contains"]:
            q = np.array(q, np.float32)
            assert bool(np.all((q >= b[:3]) & (q <= b[3:]))) == inside
        for q, want in case["nearest"]:
            assert np.all(np.abs(np.clip(np.array(q, np.float32), b[:3], b[3:]) - np.array(want, np.float32)) < tol)
    c = kats["mean_neighbour_distance"]
    pts = np.array(c["points"], np.float32)
    means = pkg.Index(pts).mean_knn_distance_self(c["k"])
    mu = np.float32(means.sum(dtype=np.float32) / np.float32(len(pts)))
    assert abs(float(mu) - c["expected_mean"]) < c["tolerance"]
