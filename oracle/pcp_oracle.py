"""ctypes view of oracle/libpcp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does.  See oracle/pcp_oracle.cpp for the reference citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def build(force=False):
    so = os.path.join(_HERE, "libpcp_oracle.so")
    src = os.path.join(_HERE, "pcp_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpcp_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libpcp_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_octree_create.restype = C.c_void_p
        L.orc_kdtree_create.restype = C.c_void_p
        L.orc_octree_size.restype = C.c_uint64
        L.orc_octree_range_sphere.restype = C.c_uint64
        L.orc_octree_range_aabb.restype = C.c_uint64
        L.orc_kdtree_range_sphere.restype = C.c_uint64
        L.orc_kdtree_range_aabb.restype = C.c_uint64
        L.orc_hardware_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def hardware_threads():
    return int(lib().orc_hardware_threads())


def bbox(xyz):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.zeros(6, np.float32)
    lib().orc_bbox(_p(xyz, _f32p), C.c_uint64(len(xyz)), _p(out, _f32p))
    return out


def knn_bruteforce(xyz, queries, k, eps=1e-5, nthreads=1, want_d2=False):
    xyz = _f32(xyz).reshape(-1, 3)
    q = _f32(queries).reshape(-1, 3)
    idx = np.empty((len(q), k), np.uint32)
    cnt = np.empty(len(q), np.uint32)
    d2 = np.empty((len(q), k), np.float32) if want_d2 else None
    lib().orc_knn_bruteforce(_p(xyz, _f32p), C.c_uint64(len(xyz)), _p(q, _f32p), C.c_uint64(len(q)),
                             C.c_uint32(k), C.c_float(eps), _p(idx, _u32p), _p(cnt, _u32p), _p(d2, _f32p),
                             C.c_int(nthreads))
    return (idx, cnt, d2) if want_d2 else (idx, cnt)


def set_geometric_prune(on):
    """Test switch: the restated trees' box-sphere test compares with radius^2 instead of the reference's `radius`
    (include/pcp/common/intersections.hpp:101,129).  Off by default: the oracle is the reference."""
    lib().orc_set_geometric_prune(C.c_int(1 if on else 0))


def range_count_bruteforce(xyz, queries, r, nthreads=1):
    xyz = _f32(xyz).reshape(-1, 3)
    q = _f32(queries).reshape(-1, 3)
    cnt = np.empty(len(q), np.uint32)
    lib().orc_range_count_bruteforce(_p(xyz, _f32p), C.c_uint64(len(xyz)), _p(q, _f32p), C.c_uint64(len(q)),
                                     C.c_float(r), _p(cnt, _u32p), C.c_int(nthreads))
    return cnt


class _Tree:
    _kind = None

    def __init__(self, handle, xyz):
        self._h = C.c_void_p(handle)
        self.xyz = xyz

    def knn(self, queries, k, eps=1e-5, nthreads=1, want_d2=False):
        q = _f32(queries).reshape(-1, 3)
        idx = np.empty((len(q), k), np.uint32)
        cnt = np.empty(len(q), np.uint32)
        d2 = np.empty((len(q), k), np.float32) if want_d2 else None
        fn = lib().orc_octree_knn if self._kind == 0 else lib().orc_kdtree_knn
        fn(self._h, _p(q, _f32p), C.c_uint64(len(q)), C.c_uint32(k), C.c_float(eps), _p(idx, _u32p),
           _p(cnt, _u32p), _p(d2, _f32p), C.c_int(nthreads))
        return (idx, cnt, d2) if want_d2 else (idx, cnt)

    def range_sphere(self, center, r):
        c = _f32(center)
        cap = len(self.xyz)
        out = np.empty(max(cap, 1), np.uint32)
        fn = lib().orc_octree_range_sphere if self._kind == 0 else lib().orc_kdtree_range_sphere
        n = fn(self._h, _p(c, _f32p), C.c_float(r), _p(out, _u32p), C.c_uint64(cap))
        return out[: int(n)].copy()

    def range_aabb(self, bmin, bmax):
        b = _f32(np.concatenate([np.asarray(bmin, np.float32), np.asarray(bmax, np.float32)]))
        cap = len(self.xyz)
        out = np.empty(max(cap, 1), np.uint32)
        fn = lib().orc_octree_range_aabb if self._kind == 0 else lib().orc_kdtree_range_aabb
        n = fn(self._h, _p(b, _f32p), _p(out, _u32p), C.c_uint64(cap))
        return out[: int(n)].copy()

    def estimate_normals(self, k, eps=1e-5, first=0, count=None, nthreads=1, want_idx=False):
        count = len(self.xyz) - first if count is None else count
        nrm = np.empty((count, 3), np.float32)
        idx = np.empty((count, k), np.uint32) if want_idx else None
        lib().orc_estimate_normals(self._h, C.c_int(self._kind), C.c_uint64(first), C.c_uint64(count),
                                   C.c_uint32(k), C.c_float(eps), _p(nrm, _f32p), _p(idx, _u32p),
                                   C.c_int(nthreads))
        return (nrm, idx) if want_idx else nrm


    def estimate_normals_stdpar(self, k, eps=1e-5, first=0, count=None):
        """The same loop under std::execution::par, as the reference's examples run it; returns (normals, number of
        distinct threads that executed the body) -- 1 where libstdc++ has no TBB backend."""
        count = len(self.xyz) - first if count is None else count
        nrm = np.empty((count, 3), np.float32)
        nt = C.c_int(0)
        lib().orc_estimate_normals_stdpar(self._h, C.c_int(self._kind), C.c_uint64(first), C.c_uint64(count), C.c_uint32(k),
                                          C.c_float(eps), _p(nrm, _f32p), C.byref(nt))
        return nrm, nt.value


class Octree(_Tree):
    """basic_linked_octree_t restatement (include/pcp/octree/linked_octree.hpp)."""
    _kind = 0

    def __init__(self, xyz, node_capacity=32, max_depth=21, voxel_grid=None):
        xyz = _f32(xyz).reshape(-1, 3)
        g = _f32(np.asarray(voxel_grid).reshape(6)) if voxel_grid is not None else None
        h = lib().orc_octree_create(_p(xyz, _f32p), C.c_uint64(len(xyz)), C.c_uint32(node_capacity),
                                    C.c_uint32(max_depth), C.c_int(0 if g is None else 1), _p(g, _f32p))
        super().__init__(h, xyz)

    def size(self):
        return int(lib().orc_octree_size(self._h))

    def voxel_grid(self):
        out = np.zeros(6, np.float32)
        lib().orc_octree_grid(self._h, _p(out, _f32p))
        return out

    def range_count(self, queries, r, nthreads=1):
        q = _f32(queries).reshape(-1, 3)
        cnt = np.empty(len(q), np.uint32)
        lib().orc_octree_range_count(self._h, _p(q, _f32p), C.c_uint64(len(q)), C.c_float(r), _p(cnt, _u32p),
                                     C.c_int(nthreads))
        return cnt

    def __del__(self):
        try:
            lib().orc_octree_destroy(self._h)
        except Exception:
            pass


class KdTree(_Tree):
    """basic_linked_kdtree_t restatement (include/pcp/kdtree/linked_kdtree.hpp)."""
    _kind = 1

    def __init__(self, xyz, max_depth=12, compute_max_depth=False, max_elements_per_leaf=64):
        xyz = _f32(xyz).reshape(-1, 3)
        h = lib().orc_kdtree_create(_p(xyz, _f32p), C.c_uint64(len(xyz)), C.c_uint64(max_depth),
                                    C.c_int(1 if compute_max_depth else 0), C.c_uint64(max_elements_per_leaf))
        super().__init__(h, xyz)

    def __del__(self):
        try:
            lib().orc_kdtree_destroy(self._h)
        except Exception:
            pass


def estimate_normal(points, want_evals=False):
    """pcp::estimate_normal (include/pcp/common/normals/normal_estimation.hpp:32-78)."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros(3, np.float32)
    ev = np.zeros(3, np.float32)
    lib().orc_estimate_normal(_p(pts, _f32p), None, C.c_uint64(len(pts)), _p(out, _f32p), _p(ev, _f32p))
    return (out, ev) if want_evals else out


def estimate_normal_traced(points):
    """estimate_normal plus which solver paths ran: (normal, eigenvalues, general tridiagonalisation taken, QR steps)."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros(3, np.float32)
    ev = np.zeros(3, np.float32)
    tr = (C.c_int * 2)()
    lib().orc_estimate_normal_traced(_p(pts, _f32p), C.c_uint64(len(pts)), _p(out, _f32p), _p(ev, _f32p), tr)
    return out, ev, bool(tr[0]), int(tr[1])


def normals_from_knn(xyz, nbr, cnt, nthreads=1, want_evals=False):
    xyz = _f32(xyz).reshape(-1, 3)
    nbr = np.ascontiguousarray(nbr, np.uint32)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    nq, k = nbr.shape
    out = np.empty((nq, 3), np.float32)
    ev = np.empty((nq, 3), np.float32) if want_evals else None
    lib().orc_normals_from_knn(_p(xyz, _f32p), _p(nbr, _u32p), _p(cnt, _u32p), C.c_uint64(nq), C.c_uint32(k),
                               _p(out, _f32p), _p(ev, _f32p), C.c_int(nthreads))
    return (out, ev) if want_evals else out


def centroids_from_knn(xyz, nbr, cnt):
    """center_of_geometry of every neighbour row (the tangent plane's point)."""
    xyz = _f32(xyz).reshape(-1, 3)
    nbr = np.ascontiguousarray(nbr, np.uint32)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    nq, k = nbr.shape
    out = np.empty((nq, 3), np.float32)
    lib().orc_centroids_from_knn(_p(xyz, _f32p), _p(nbr, _u32p), _p(cnt, _u32p), C.c_uint64(nq), C.c_uint32(k), _p(out, _f32p))
    return out


def propagate_normal_orientations(xyz, nbr, cnt, normals):
    """propagate_normal_orientations over explicit neighbour rows; returns (oriented normals, vertices reached)."""
    xyz = _f32(xyz).reshape(-1, 3)
    nbr = np.ascontiguousarray(nbr, np.uint32)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    n, k = nbr.shape
    out = np.array(normals, dtype=np.float32, order="C", copy=True).reshape(-1, 3)
    fn = lib().orc_propagate_normal_orientations
    fn.restype = C.c_uint64
    reached = fn(_p(xyz, _f32p), C.c_uint64(n), _p(nbr, _u32p), _p(cnt, _u32p), C.c_uint32(k), _p(out, _f32p))
    return out, int(reached)


def mean_dist_from_knn(xyz, queries, nbr, cnt):
    """average_distances_to_neighbors: mean Euclidean distance from each query to its neighbour row."""
    xyz = _f32(xyz).reshape(-1, 3)
    q = _f32(queries).reshape(-1, 3)
    nbr = np.ascontiguousarray(nbr, np.uint32)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    nq, k = nbr.shape
    out = np.empty(nq, np.float32)
    lib().orc_mean_dist_from_knn(_p(xyz, _f32p), _p(q, _f32p), _p(nbr, _u32p), _p(cnt, _u32p), C.c_uint64(nq), C.c_uint32(k),
                                 _p(out, _f32p))
    return out


def bilateral_filter_points(xyz, normals, sigmaf, sigmag, K=1, f64_yardstick=False, nthreads=1):
    """pcp::algorithm::bilateral_filter_points (bilateral_filter.hpp:303-428); f64_yardstick: same in double."""
    xyz = _f32(xyz).reshape(-1, 3)
    nrm = _f32(normals).reshape(-1, 3)
    out = np.empty_like(xyz)
    lib().orc_bilateral_filter_points(_p(xyz, _f32p), _p(nrm, _f32p), C.c_uint64(len(xyz)), C.c_double(sigmaf), C.c_double(sigmag),
                                      C.c_uint64(K), _p(out, _f32p), C.c_int(int(f64_yardstick)), C.c_int(nthreads))
    return out


def bilateral_filter_normals(xyz, normals, sigmaf, sigmag, K=1, f64_yardstick=False, nthreads=1, want_cancellation=False):
    """pcp::algorithm::bilateral_filter_normals (bilateral_filter.hpp:460-574).  want_cancellation: also the per-row factor by
    which the quotient rule's two terms exceed their difference in the last iteration (rows where the formula is unstable)."""
    xyz = _f32(xyz).reshape(-1, 3)
    nrm = _f32(normals).reshape(-1, 3)
    out = np.empty_like(nrm)
    cf = np.zeros(len(xyz), np.float32) if want_cancellation else None
    lib().orc_bilateral_filter_normals(_p(xyz, _f32p), _p(nrm, _f32p), C.c_uint64(len(xyz)), C.c_double(sigmaf), C.c_double(sigmag),
                                       C.c_uint64(K), _p(out, _f32p), C.c_int(int(f64_yardstick)), C.c_int(nthreads), _p(cf, _f32p))
    return (out, cf) if want_cancellation else out


def wlop(xyz, sample, mu, h, K, uniform=True, f64_yardstick=False, nthreads=1):
    """pcp::algorithm::wlop::wlop (wlop.hpp:287-428) from a given initial sample (indices into xyz)."""
    xyz = _f32(xyz).reshape(-1, 3)
    sample = np.ascontiguousarray(sample, np.uint64)
    out = np.empty((len(sample), 3), np.float32)
    lib().orc_wlop(_p(xyz, _f32p), C.c_uint64(len(xyz)), _p(sample, C.POINTER(C.c_uint64)), C.c_uint64(len(sample)), C.c_double(mu),
                   C.c_double(h), C.c_uint64(K), C.c_int(int(uniform)), _p(out, _f32p), C.c_int(int(f64_yardstick)), C.c_int(nthreads))
    return out


# ---- K-dimensional kd-tree queries (numpy restatement; K > 3 has no caller in the reference's tests or examples, so these are
#      pinned through K = 3: tests/test_oracle.py checks them against the C++ restatement above, which the reference's KATs pin) ----
def kd_knn_bruteforce(points, queries, k, eps=1e-5):
    """pcp::basic_linked_kdtree_t::nearest_neighbours for any K (include/pcp/kdtree/linked_kdtree.hpp:200-262, recurse_knn :436-540):
    the k points of smallest squared distance -- std::inner_product of the coordinate differences with itself from 0, in float,
    axis by axis (include/pcp/common/norm.hpp:123-141) --, nearest first, a point with |p[a] - q[a]| < eps on every axis skipped
    (common::floating_point_equals, vector3d_queries.hpp:31-35).  Equal distances in index order (the reference: heap order).
    Returns idx (nq, k) padded with 0xFFFFFFFF, count (nq,), d2 (nq, k) padded with +inf."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, pts.shape[1])
    n, dims = pts.shape
    idx = np.full((len(q), k), 0xFFFFFFFF, np.uint32)
    d2o = np.full((len(q), k), np.inf, np.float32)
    cnt = np.zeros(len(q), np.uint32)
    eps32 = np.float32(eps) if eps > 0 else np.float32(0)
    for j in range(len(q)):
        acc = np.zeros(n, np.float32)
        same = np.ones(n, bool)
        for a in range(dims):
            d = pts[:, a] - q[j, a]
            acc = acc + d * d
            same &= np.abs(d) < eps32
        keep = np.nonzero(~same & ~np.isnan(acc))[0]
        order = keep[np.lexsort((keep, acc[keep]))][:k]
        cnt[j] = len(order)
        idx[j, :len(order)] = order
        d2o[j, :len(order)] = acc[order]
    return idx, cnt, d2o


def kd_range_aabb(points, boxes):
    """range_search with a kd_axis_aligned_bounding_box_t (linked_kdtree.hpp:270-311; contains: min <= p <= max on every axis,
    include/pcp/common/axis_aligned_bounding_box.hpp): per box the sorted indices of the points inside."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    dims = pts.shape[1]
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 2 * dims)
    return [np.nonzero(((pts >= bb[:dims]) & (pts <= bb[dims:])).all(1))[0] for bb in b]
