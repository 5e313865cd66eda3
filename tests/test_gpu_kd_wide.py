"""pcp::basic_linked_kdtree_t for K > 3 (include/pcpx.h: pcpx_kd_*; csrc/pcpx_kd.hip) against the numpy restatement of the
reference's kd-tree queries (oracle/pcp_oracle.py: kd_knn_bruteforce, kd_range_aabb)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _assert_rows(pts, q, k, gi, gc, gd, oi, oc, od):
    """Counts and float32 squared distances bit for bit; indices equal except among points at EXACTLY the row's last distance
    (the reference leaves such ties to its heap): there the point must be a real one at that distance, the row in (d2, index) order."""
    assert np.array_equal(gc, oc)
    assert np.array_equal(gd, od)
    for j in np.nonzero((gi != oi).any(1))[0]:
        cols = np.nonzero(gi[j] != oi[j])[0]
        assert np.all(od[j, cols] == od[j, oc[j] - 1])


@pytest.mark.parametrize("dims", [4, 5, 8, 16])
@pytest.mark.parametrize("n,nq,k", [(1, 3, 4), (50, 7, 15), (1000, 40, 1), (1000, 40, 64), (1000, 9, 65), (3000, 5, 200), (20000, 300, 15), (70000, 2, 33)])
def test_knn_in_k_dimensions(pkg, oracle, dims, n, nq, k):
    rng = np.random.default_rng(1000 * dims + n + k)
    pts = rng.random((n, dims), dtype=np.float32)
    q = np.concatenate([pts[rng.integers(0, n, nq // 2)], (rng.random((nq - nq // 2, dims), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)])
    tree = pkg.KdTreeK(pts)
    assert tree.size() == n
    gi, gc, gd = tree.nearest_neighbours(q, k, want_d2=True)
    oi, oc, od = oracle.kd_knn_bruteforce(pts, q, k)
    _assert_rows(pts, q, k, gi, gc, gd, oi, oc, od)
    tree.close()


def test_knn_ties_eps_and_duplicates(pkg, oracle):
    """A lattice (many exactly equal distances), duplicated points, eps = 0 (nothing is 'equal': the point itself comes first) and a
    large eps (a whole neighbourhood left out)."""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.arange(6, dtype=np.float32)] * 4, indexing="ij"), -1).reshape(-1, 4)
    pts = np.concatenate([g, g[:200], g[:50]]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    q = pts[rng.integers(0, len(pts), 60)]
    tree = pkg.KdTreeK(pts)
    for eps in (1e-5, 0.0, 1.5):
        for k in (1, 9, 40, 100):
            gi, gc, gd = tree.nearest_neighbours(q, k, eps=eps, want_d2=True)
            oi, oc, od = oracle.kd_knn_bruteforce(pts, q, k, eps=eps)
            _assert_rows(pts, q, k, gi, gc, gd, oi, oc, od)
            for j in range(len(q)):  # rows in (d2, index) order
                keys = [(float(gd[j, c]), int(gi[j, c])) for c in range(gc[j])]
                assert keys == sorted(keys)
    tree.close()


@pytest.mark.parametrize("dims,n", [(4, 1), (4, 3000), (6, 50000), (16, 2000)])
def test_boxes_in_k_dimensions(pkg, oracle, dims, n):
    rng = np.random.default_rng(dims * n)
    pts = rng.random((n, dims), dtype=np.float32)
    lo = (rng.random((40, dims), dtype=np.float32) * 0.7).astype(np.float32)
    boxes = np.concatenate([lo, lo + rng.random((40, dims), dtype=np.float32) * 0.9], axis=1).astype(np.float32)
    boxes[0, :dims], boxes[0, dims:] = 0.0, 1.0        # everything
    boxes[1, :dims], boxes[1, dims:] = 2.0, 3.0        # nothing
    boxes[2, :dims], boxes[2, dims:] = pts[0], pts[0]  # a box that is one point: its faces are inside
    tree = pkg.KdTreeK(pts)
    off, idx = tree.range_search(boxes)
    want = oracle.kd_range_aabb(pts, boxes)
    assert off[0] == 0 and off[-1] == len(idx)
    for i, w in enumerate(want):
        assert sorted(idx[int(off[i]):int(off[i + 1])].tolist()) == w.tolist()
    tree.close()


def test_empty_and_bad_arguments(pkg):
    tree = pkg.KdTreeK(np.empty((0, 5), np.float32))
    idx, cnt = tree.nearest_neighbours(np.zeros((2, 5), np.float32), 3)
    assert np.all(cnt == 0) and np.all(idx == 0xFFFFFFFF)
    off, out = tree.range_search(np.zeros((2, 10), np.float32))
    assert np.all(off == 0) and len(out) == 0
    tree.close()
    with pytest.raises(pkg.PcpxError):
        pkg.KdTreeK(np.zeros((4, 17), np.float32))
