"""Host-side mirror (Python) of the reference's spatial containers and normal estimation, in batched
form, over the C ABI of libpcpx.so.  The C++17 drop-in headers live in include/pcp/; this module
exists so that tests and bench.py can drive the same entry points from Python.

Names follow the reference: LinkedOctree ~ pcp::basic_linked_octree_t
(include/pcp/octree/linked_octree.hpp:40-41), LinkedKdTree ~ pcp::basic_linked_kdtree_t
(include/pcp/kdtree/linked_kdtree.hpp:64-65), estimate_normals ~ pcp::algorithm::estimate_normals
(include/pcp/algorithm/estimate_normals.hpp:58-65).  Elements are indices into the input array.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import BuildParams, PcpxError, check  # noqa: F401

INVALID = np.uint32(0xFFFFFFFF)


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    n = C.c_int(0)
    check(_capi.load().pcpx_device_count(C.byref(n)))
    return n.value


def bounding_box(xyz, device=0):
    """pcp::bounding_box (include/pcp/common/axis_aligned_bounding_box.hpp:214-251) on the GPU."""
    xyz = _f32(xyz, 3)
    out = np.zeros(6, np.float32)
    check(_capi.load().pcpx_bounding_box(_vp(xyz), len(xyz), device, out.ctypes.data_as(_capi.f32p)))
    return out


def shard_range(n, rank, world):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(_capi.load().pcpx_shard_range(n, rank, world, C.byref(a), C.byref(b)))
    return a.value, b.value


def shard_cuts_by_cost(n, world, group_stride, events):
    """pcpx_shard_cuts_by_cost: world + 1 curve positions (multiples of 64) that cut the order into shards of equal estimated WORK;
    events = the table of Index.knn_group_costs (nsamples x 4, uint32)."""
    ev = np.ascontiguousarray(events, dtype=np.uint32).reshape(-1, 4)
    out = (C.c_uint64 * (world + 1))()
    check(_capi.load().pcpx_shard_cuts_by_cost(n, world, group_stride, _vp(ev) if len(ev) else None, len(ev), out))
    return [int(v) for v in out]


def estimate_normal(points, device=0):
    """pcp::estimate_normal (include/pcp/common/normals/normal_estimation.hpp:32-78)."""
    pts = _f32(points, 3)
    out = np.zeros(3, np.float32)
    check(_capi.load().pcpx_estimate_normal(_vp(pts), len(pts), device, out.ctypes.data_as(_capi.f32p)))
    return out


class Index:
    """Owns a pcpx_index handle: the device-resident curve-sorted implicit AABB tree."""

    def __init__(self, xyz, voxel_grid=None, device=0, coarse_order=False, shard=None, k_hint=0):
        self._lib = _capi.load()
        self._h = C.c_void_p(None)
        self.device = device
        xyz = _f32(xyz, 3)
        self.n_in = len(xyz)
        p = self._params(voxel_grid, coarse_order, shard, k_hint)
        check(self._lib.pcpx_index_create(_vp(xyz), len(xyz), p, device, C.byref(self._h)))

    @staticmethod
    def _params(voxel_grid, coarse_order=False, shard=None, k_hint=0, borrow=False, shard_range=None):
        """pcpx_build_params: voxel_grid = the explicit grid (PCPX_BUILD_USE_GRID); coarse_order = PCPX_BUILD_COARSE_ORDER (an index that
        is rebuilt after a query pass or two: one radix pass fewer on a uniform cloud, same results); shard = (rank, world):
        PCPX_BUILD_SHARD, the rank-local index of the multi-GPU path (k_hint sizes its halo; borrow = PCPX_BUILD_BORROW_CLOUD)."""
        if shard_range is not None and shard is None:
            shard = (0, 1)
        if voxel_grid is None and not coarse_order and shard is None:
            return None
        p = BuildParams()
        p.struct_size = C.sizeof(BuildParams)
        p.flags = (_capi.PCPX_BUILD_USE_GRID if voxel_grid is not None else 0) | (_capi.PCPX_BUILD_COARSE_ORDER if coarse_order else 0)
        if shard is not None:
            p.flags |= _capi.PCPX_BUILD_SHARD | (_capi.PCPX_BUILD_BORROW_CLOUD if borrow else 0)
            p.shard_rank, p.shard_world = int(shard[0]), int(shard[1])
            p.shard_k_hint = int(k_hint)
            if shard_range is not None:  # PCPX_BUILD_SHARD_RANGE: explicit curve positions (first, count), e.g. a cut by work
                p.flags |= _capi.PCPX_BUILD_SHARD_RANGE
                p.shard_first, p.shard_count = int(shard_range[0]), int(shard_range[1])
        if voxel_grid is None:
            return C.pointer(p)
        g = np.asarray(voxel_grid, np.float32).reshape(6)
        for a in range(3):
            p.grid_min[a] = float(g[a])
            p.grid_max[a] = float(g[3 + a])
        return C.pointer(p)

    def rebuild(self, xyz, voxel_grid=None, coarse_order=False, shard=None, k_hint=0):
        xyz = _f32(xyz, 3)
        check(self._lib.pcpx_index_rebuild(self._h, _vp(xyz), len(xyz), self._params(voxel_grid, coarse_order, shard, k_hint)))
        self.n_in = len(xyz)

    def shard_info(self):
        """A rank-local index described (pcpx_index_shard_info)."""
        out = (C.c_uint64 * 8)()
        check(self._lib.pcpx_index_shard_info(self._h, out))
        names = ["local_points", "core_first", "core_count", "shard_first", "shard_count", "halo_cells", "last_failed", "enlargements"]
        return {k: int(out[i]) for i, k in enumerate(names)}

    def close(self):
        if self._h:
            self._lib.pcpx_index_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- container queries ----
    def size(self):
        n = C.c_uint64(0)
        check(self._lib.pcpx_index_size(self._h, C.byref(n)))
        return n.value

    def empty(self):
        return self.size() == 0

    def bbox(self):
        out = np.zeros(6, np.float32)
        check(self._lib.pcpx_index_bbox(self._h, out.ctypes.data_as(_capi.f32p)))
        return out

    # ---- kNN ----
    def knn_self(self, k, eps=1e-5, want_d2=False):
        idx = np.empty((self.n_in, k), np.uint32)
        cnt = np.empty(self.n_in, np.uint32)
        d2 = np.empty((self.n_in, k), np.float32) if want_d2 else None
        check(self._lib.pcpx_knn_self(self._h, k, eps, _vp(idx), _vp(cnt), _vp(d2)))
        return (idx, cnt, d2) if want_d2 else (idx, cnt)

    def knn(self, queries, k, eps=1e-5, want_d2=False):
        q = _f32(queries, 3)
        idx = np.empty((len(q), k), np.uint32)
        cnt = np.empty(len(q), np.uint32)
        d2 = np.empty((len(q), k), np.float32) if want_d2 else None
        check(self._lib.pcpx_knn_batch(self._h, _vp(q), len(q), k, eps, _vp(idx), _vp(cnt), _vp(d2)))
        return (idx, cnt, d2) if want_d2 else (idx, cnt)

    # ---- radius search ----
    def range_count_self(self, radius):
        cnt = np.empty(self.n_in, np.uint32)
        check(self._lib.pcpx_range_count_self(self._h, radius, _vp(cnt)))
        return cnt

    def range_count(self, queries, radius):
        q = _f32(queries, 3)
        cnt = np.empty(len(q), np.uint32)
        check(self._lib.pcpx_range_count_batch(self._h, _vp(q), len(q), radius, _vp(cnt)))
        return cnt

    def range_sphere(self, centers, radius):
        """CSR (offsets, indices) of the points inside each sphere; radius scalar or per-sphere."""
        q = _f32(centers, 3)
        radii = None
        r = 0.0
        if np.ndim(radius) == 0:
            r = float(radius)
        else:
            radii = _f32(radius).reshape(-1)
            assert len(radii) == len(q)
        off = np.zeros(len(q) + 1, np.uint64)
        st = self._lib.pcpx_range_sphere_batch(self._h, _vp(q), _vp(radii), r, len(q), _vp(off), None, 0)
        if st == _capi.PCPX_OK:
            return off, np.empty(0, np.uint32)
        if st != _capi.PCPX_ERR_CAPACITY:
            check(st)
        out = np.empty(int(off[-1]), np.uint32)
        check(self._lib.pcpx_range_sphere_batch(self._h, _vp(q), _vp(radii), r, len(q), _vp(off), _vp(out), len(out)))
        return off, out

    def range_aabb(self, boxes):
        b = _f32(boxes, 6)
        off = np.zeros(len(b) + 1, np.uint64)
        st = self._lib.pcpx_range_aabb_batch(self._h, _vp(b), len(b), _vp(off), None, 0)
        if st == _capi.PCPX_OK:
            return off, np.empty(0, np.uint32)
        if st != _capi.PCPX_ERR_CAPACITY:
            check(st)
        out = np.empty(int(off[-1]), np.uint32)
        check(self._lib.pcpx_range_aabb_batch(self._h, _vp(b), len(b), _vp(off), _vp(out), len(out)))
        return off, out

    # ---- normals ----
    def normals_knn_self(self, k, eps=1e-5, want_knn=False):
        nrm = np.empty((self.n_in, 3), np.float32)
        idx = np.empty((self.n_in, k), np.uint32) if want_knn else None
        cnt = np.empty(self.n_in, np.uint32) if want_knn else None
        check(self._lib.pcpx_normals_knn_self(self._h, k, eps, _vp(nrm), _vp(idx), _vp(cnt)))
        return (nrm, idx, cnt) if want_knn else nrm

    def normals_knn_self_curve_order(self, k, eps=1e-5, want_normals=True):
        """(normals, idx, cnt, perm, position_of) with the rows in curve order: row p belongs to input point perm[p]; position_of is
        the inverse table (0xFFFFFFFF for a point outside the voxel grid).  The copies overlap the kernels (include/pcpx.h)."""
        n, n_in = self.size(), self.n_in
        nrm = np.empty((n, 3), np.float32) if want_normals else None
        idx = np.empty((n, k), np.uint32)
        cnt = np.empty(n, np.uint32)
        perm = np.empty(n, np.uint32)
        pos = np.empty(n_in, np.uint32)
        check(self._lib.pcpx_normals_knn_self_curve_order(self._h, k, eps, _vp(nrm), _vp(idx), _vp(cnt), _vp(perm), _vp(pos)))
        return nrm, idx, cnt, perm, pos

    def oriented_normals_knn_self(self, k, eps=1e-5, want_knn=False):
        """estimate_normals followed by propagate_normal_orientations, both on the GPU (the rows stay there).
        Returns normals (and rows, counts if want_knn) plus the number of points reached from the root."""
        n = self.n_in
        nrm = np.empty((n, 3), np.float32)
        idx = np.empty((n, k), np.uint32) if want_knn else None
        cnt = np.empty(n, np.uint32) if want_knn else None
        reached = C.c_uint64(0)
        check(self._lib.pcpx_oriented_normals_knn_self(self._h, k, eps, _vp(nrm), _vp(idx) if want_knn else None,
                                                       _vp(cnt) if want_knn else None, C.byref(reached)))
        return (nrm, idx, cnt, int(reached.value)) if want_knn else (nrm, int(reached.value))

    def orient_normals_knn_self(self, normals, k, eps=1e-5):
        """propagate_normal_orientations for normals the caller has (n x 3, input order); kNN graph built on the GPU."""
        out = np.array(normals, dtype=np.float32, order="C", copy=True).reshape(-1, 3)
        if out.shape[0] != self.n_in:
            raise ValueError("normals must be n x 3")
        reached = C.c_uint64(0)
        check(self._lib.pcpx_orient_normals_knn_self(self._h, k, eps, _vp(out), C.byref(reached)))
        return out, int(reached.value)

    def tangent_planes_knn_self(self, k, eps=1e-5):
        """pcp::algorithm::estimate_tangent_planes: (centroids, normals) of every point's k-neighbourhood."""
        cen = np.empty((self.n_in, 3), np.float32)
        nrm = np.empty((self.n_in, 3), np.float32)
        check(self._lib.pcpx_tangent_planes_knn_self(self._h, k, eps, _vp(cen), _vp(nrm)))
        return cen, nrm

    def mean_knn_distance_self(self, k, eps=1e-5):
        """pcp::algorithm::average_distances_to_neighbors: mean distance to the k nearest neighbours, per point."""
        out = np.empty(self.n_in, np.float32)
        check(self._lib.pcpx_mean_knn_distance_self(self._h, k, eps, _vp(out)))
        return out

    def normals_from_knn(self, nbr, cnt, want_evals=False):
        nbr = np.ascontiguousarray(nbr, np.uint32)
        cnt = np.ascontiguousarray(cnt, np.uint32)
        nq, k = nbr.shape
        nrm = np.empty((nq, 3), np.float32)
        ev = np.empty((nq, 3), np.float32) if want_evals else None
        check(self._lib.pcpx_normals_from_knn(self._h, _vp(nbr), _vp(cnt), nq, k, _vp(nrm), _vp(ev)))
        return (nrm, ev) if want_evals else nrm

    # ---- device-pointer forms (torch tensors / raw pointers), used by bench.py ----
    @classmethod
    def from_device(cls, d_xyz_ptr, n, device=0, stream=None, voxel_grid=None, coarse_order=False, shard=None, k_hint=0, borrow=False, shard_range=None):
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self._h = C.c_void_p(None)
        self.device = device
        self.n_in = n
        check(self._lib.pcpx_index_create_dev(C.c_void_p(d_xyz_ptr), n, cls._params(voxel_grid, coarse_order, shard, k_hint, borrow, shard_range), device,
                                              C.c_void_p(stream) if stream else None, C.byref(self._h)))
        return self

    def rebuild_dev(self, d_xyz_ptr, n, voxel_grid=None, coarse_order=False, shard=None, k_hint=0, borrow=False, shard_range=None):
        check(self._lib.pcpx_index_rebuild_dev(self._h, C.c_void_p(d_xyz_ptr), n, self._params(voxel_grid, coarse_order, shard, k_hint, borrow, shard_range)))
        self.n_in = n

    def knn_self_curve_order_dev(self, k, eps, d_idx, d_cnt, d_d2=None, d_normals=None, first=0, count=_capi.UINT64_MAX):
        """Rows by curve position (row p = the p-th point of the curve order = input point perm[p], perm_dev)."""
        check(self._lib.pcpx_knn_self_curve_order_dev(self._h, k, eps, first, count, C.c_void_p(d_idx), C.c_void_p(d_cnt),
                                                      C.c_void_p(d_d2) if d_d2 else None, C.c_void_p(d_normals) if d_normals else None))

    def perm_dev(self, d_perm=None, d_position_of=None):
        check(self._lib.pcpx_index_perm_dev(self._h, C.c_void_p(d_perm) if d_perm else None, C.c_void_p(d_position_of) if d_position_of else None))

    def knn_self_dev(self, k, eps, d_idx, d_cnt, d_d2=None, first=0, count=_capi.UINT64_MAX):
        check(self._lib.pcpx_knn_self_dev(self._h, k, eps, first, count, C.c_void_p(d_idx), C.c_void_p(d_cnt),
                                          C.c_void_p(d_d2) if d_d2 else None))

    def knn_self_strided_dev(self, k, eps, row_stride, d_idx, d_cnt, d_d2=None, first=0, count=_capi.UINT64_MAX):
        """knn_self_dev with row_stride entries between rows (16 for k = 15 / 16: a row is one aligned 64-byte piece)."""
        check(self._lib.pcpx_knn_self_strided_dev(self._h, k, eps, first, count, row_stride, C.c_void_p(d_idx), C.c_void_p(d_cnt),
                                                  C.c_void_p(d_d2) if d_d2 else None))

    def normals_knn_self_strided_dev(self, k, eps, row_stride, d_normals, d_idx=None, d_cnt=None, first=0, count=_capi.UINT64_MAX):
        check(self._lib.pcpx_normals_knn_self_strided_dev(self._h, k, eps, first, count, row_stride, C.c_void_p(d_normals),
                                                          C.c_void_p(d_idx) if d_idx else None, C.c_void_p(d_cnt) if d_cnt else None))

    def knn_group_costs(self, k, eps=1e-5, group_stride=16):
        """pcpx_knn_group_costs_dev: event counts (nsamples x 4, uint32, on the host) of one query group in every group_stride of the
        curve order -- what shard_cuts_by_cost cuts by.  Deterministic: every rank gets the same table from the same cloud and grid."""
        ns = C.c_uint64(0)
        st = self._lib.pcpx_knn_group_costs_dev(self._h, k, eps, group_stride, None, 0, C.byref(ns))
        if st not in (_capi.PCPX_OK, _capi.PCPX_ERR_CAPACITY):
            check(st)
        out = np.zeros((int(ns.value), 4), np.uint32)
        if ns.value == 0:
            return out
        d = C.c_void_p(None)
        check(self._lib.pcpx_device_malloc(out.nbytes, self.device, C.byref(d)))
        try:
            check(self._lib.pcpx_knn_group_costs_dev(self._h, k, eps, group_stride, d, ns.value, C.byref(ns)))
            self.synchronize()
            check(self._lib.pcpx_device_download(_vp(out), d, out.nbytes, self.device, None))
        finally:
            self._lib.pcpx_device_free(d, self.device)
        return out

    def debug_set(self, name, value):
        """pcpx_debug_set: 'long_groups_first', 'gather_outputs' (how work is done, never what comes out)."""
        check(self._lib.pcpx_debug_set(self._h, name.encode(), int(value)))

    def debug_get(self, name):
        """pcpx_debug_get: 'build_redos', 'full_buckets', 'schedule_state'."""
        v = C.c_int64(0)
        check(self._lib.pcpx_debug_get(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def debug_group_times(self):
        """Ticks / 64 per query group of the last recorded self-kNN launch (uint32 array; empty: nothing recorded)."""
        ng = C.c_uint64(0)
        st = self._lib.pcpx_debug_group_times(self._h, None, 0, C.byref(ng))
        if st not in (_capi.PCPX_OK, _capi.PCPX_ERR_CAPACITY):
            check(st)
        out = np.zeros(int(ng.value), np.uint32)
        if ng.value:
            check(self._lib.pcpx_debug_group_times(self._h, _vp(out), ng.value, C.byref(ng)))
        return out

    def knn_batch_dev(self, d_queries, nq, k, eps, d_idx, d_cnt, d_d2=None):
        """kNN of nq arbitrary device-resident query points (curve-sorted internally, rows in query order)."""
        check(self._lib.pcpx_knn_batch_dev(self._h, d_queries, nq, k, eps, d_idx, d_cnt, d_d2))

    def normals_knn_self_dev(self, k, eps, d_normals, d_idx=None, d_cnt=None, first=0, count=_capi.UINT64_MAX):
        check(self._lib.pcpx_normals_knn_self_dev(self._h, k, eps, first, count, C.c_void_p(d_normals),
                                                  C.c_void_p(d_idx) if d_idx else None,
                                                  C.c_void_p(d_cnt) if d_cnt else None))

    def range_count_self_dev(self, radius, d_cnt, first=0, count=_capi.UINT64_MAX):
        check(self._lib.pcpx_range_count_self_dev(self._h, radius, first, count, C.c_void_p(d_cnt)))

    def range_count_self_curve_order_dev(self, radius, d_cnt, first=0, count=_capi.UINT64_MAX):
        """Counts at curve positions (d_cnt[p] = count around the p-th point of the index's order; perm_dev gives the order)."""
        check(self._lib.pcpx_range_count_self_curve_order_dev(self._h, radius, first, count, C.c_void_p(d_cnt)))

    def range_lists_self_dev(self, radius, d_offsets, d_idx=None, capacity=0):
        """pcpx_range_lists_self_dev: CSR lists of every indexed point's sphere range, device resident.  Returns the total; with d_idx
        None (or too small) only the offsets are filled -- allocate `total` indices and call again."""
        total = C.c_uint64(0)
        st = self._lib.pcpx_range_lists_self_dev(self._h, radius, C.c_void_p(d_offsets), C.c_void_p(d_idx) if d_idx else None, capacity, C.byref(total))
        if st != _capi.PCPX_OK and not (st == _capi.PCPX_ERR_CAPACITY and (not d_idx or capacity < total.value)):
            check(st)
        return int(total.value)

    def synchronize(self):
        check(self._lib.pcpx_index_synchronize(self._h))

    def debug_eps_test_mode(self, mode):
        """0: automatic; 1: eps-box test on buffered keys (compaction); 2: on every candidate.  Same results, different cost."""
        check(self._lib.pcpx_debug_eps_test_mode(self._h, int(mode)))

    def debug_knn_stats(self, k, eps=1e-5, want_waves=False, floor=False):
        """floor=True: every lane starts from its true k-th distance (what a perfect visiting order could reach)."""
        cap = 16 + 5 * 65536
        out = (C.c_uint64 * cap)()
        check(self._lib.pcpx_debug_knn_stats(self._h, k, eps, out, cap | ((1 << 63) if floor else 0)))
        names = ["leaves", "expansions", "compactions", "appended", "waves", "seed_leaves", "second_round_groups",
                 "cycles_walk", "cycles_compact", "cycles_leaf", "cycles_search_loop", "cycles_group",
                 "seed_compactions", "seed_appended", "sparse_leaves", "sparse_leaf_lanes"]
        d = {n: int(out[i]) for i, n in enumerate(names)}
        d["cycles_later_rounds"] = int(out[cap - 8])   # walk rounds after a group's first (lanes whose k-th distance lay beyond the cap)
        d["lanes_in_later_rounds"] = int(out[cap - 7])
        if want_waves:
            w = np.frombuffer(out, dtype=np.uint64)[16:16 + 5 * 65534].reshape(-1, 5)  # (the last two records' words hold the round-4 counters)
            d["wave_times"] = w[w[:, 1] > 0].copy()
        return d

    def profile_begin(self):
        check(self._lib.pcpx_profile_begin(self._h))

    def profile_end(self):
        """{family: (launches, total_ms)} from hipEvents recorded on the index's stream."""
        p = _capi.Profile()
        check(self._lib.pcpx_profile_end(self._h, C.byref(p)))
        names = ["build", "knn", "normals", "range", "query_prep"]
        return {n: (int(p.launches[i]), float(p.total_ms[i])) for i, n in enumerate(names)}


class LinkedOctree(Index):
    """pcp::basic_linked_octree_t over index elements.  node_capacity / max_depth are accepted for
    signature parity (include/pcp/octree/linked_octree_node.hpp:31-40) but do not shape the GPU
    structure: only query results are observable."""

    def __init__(self, xyz, node_capacity=32, max_depth=21, voxel_grid=None, device=0):
        assert node_capacity > 0 and max_depth > 0  # linked_octree_node.hpp:87-88
        super().__init__(xyz, voxel_grid=voxel_grid, device=device)

    def voxel_grid(self):
        return self.bbox()

    def nearest_neighbours(self, targets, k, eps=1e-5):
        return self.knn(targets, k, eps)

    def range_search(self, centers, radius):
        return self.range_sphere(centers, radius)


class LinkedKdTree(Index):
    """pcp::basic_linked_kdtree_t over index elements (construction_params_t accepted, unused)."""

    def __init__(self, xyz, max_depth=12, compute_max_depth=False, max_elements_per_leaf=64, device=0):
        super().__init__(xyz, voxel_grid=None, device=device)

    def aabb(self):
        return self.bbox()

    def nearest_neighbours(self, targets, k, eps=1e-5):
        return self.knn(targets, k, eps)

    def range_search(self, centers, radius):
        return self.range_sphere(centers, radius)


class KdTreeK:
    """pcp::basic_linked_kdtree_t for K > 3 coordinates (4 ... 16): nearest_neighbours and range_search with a kd box, by
    exhaustive search on the GPU (include/pcpx.h: pcpx_kd_*; csrc/pcpx_kd.hip).  K <= 3 goes through LinkedKdTree."""

    def __init__(self, points, device=0):
        pts = np.ascontiguousarray(points, dtype=np.float32)
        if pts.ndim != 2:
            raise ValueError("points: an (n, K) array")
        self.n_in, self.dims = int(pts.shape[0]), int(pts.shape[1])
        self._lib = _capi.load()
        h = C.c_void_p()
        check(self._lib.pcpx_kd_create(_vp(pts), self.n_in, self.dims, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pcpx_kd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self._lib.pcpx_kd_size(self._h))

    def nearest_neighbours(self, targets, k, eps=1e-5, want_d2=False):
        """(idx (nq, k) uint32 padded with 0xFFFFFFFF, count (nq,)[, d2 (nq, k) padded with +inf]): ascending (d2, index)."""
        q = _f32(targets, self.dims)
        idx = np.empty((len(q), k), np.uint32)
        cnt = np.empty(len(q), np.uint32)
        d2 = np.empty((len(q), k), np.float32) if want_d2 else None
        check(self._lib.pcpx_kd_knn_batch(self._h, _vp(q), len(q), k, eps, _vp(idx), _vp(cnt), _vp(d2)))
        return (idx, cnt, d2) if want_d2 else (idx, cnt)

    def range_search(self, boxes):
        """CSR (offsets, indices) of the points inside each kd box; boxes (nb, 2 K): min then max."""
        b = _f32(boxes, 2 * self.dims)
        off = np.zeros(len(b) + 1, np.uint64)
        st = self._lib.pcpx_kd_range_aabb_batch(self._h, _vp(b), len(b), _vp(off), None, 0)
        if st == _capi.PCPX_OK:
            return off, np.empty(0, np.uint32)
        if st != _capi.PCPX_ERR_CAPACITY:
            check(st)
        out = np.empty(int(off[-1]), np.uint32)
        check(self._lib.pcpx_kd_range_aabb_batch(self._h, _vp(b), len(b), _vp(off), _vp(out), len(out)))
        return off, out


def estimate_normals(tree, k, eps=1e-5):
    """pcp::algorithm::estimate_normals with knn_map = tree.nearest_neighbours(point, k): one normal
    per indexed point (examples/simple_example.cpp:83-99)."""
    return tree.normals_knn_self(k, eps)


def propagate_normal_orientations_dev(d_xyz, n, d_knn_idx, d_knn_count, k, d_normals, device=0, stream=None):
    """Device-pointer form (level-synchronous search on the GPU, same flips as the host form); returns
    (vertices reached, BFS depth).  Normals are updated in place."""
    reached, levels = C.c_uint64(0), C.c_uint32(0)
    check(_capi.load().pcpx_propagate_normal_orientations_dev(d_xyz, n, d_knn_idx, d_knn_count, k, d_normals, device, stream,
                                                              C.byref(reached), C.byref(levels)))
    return int(reached.value), int(levels.value)


def propagate_normal_orientations(points, knn_idx, normals, knn_count=None):
    """pcp::algorithm::propagate_normal_orientations (include/pcp/algorithm/estimate_normals.hpp:187-302) over
    the kNN rows the query kernels return (knn_idx n x k, optional per-row counts): root = first point of
    largest z with normal (0,0,1), breadth-first flips.  Returns (oriented normals, vertices reached)."""
    pts = _f32(points, 3)
    nbr = np.ascontiguousarray(knn_idx, dtype=np.uint32)
    if nbr.ndim != 2 or nbr.shape[0] != pts.shape[0]:
        raise ValueError("knn_idx must be n x k")
    out = np.array(normals, dtype=np.float32, order="C", copy=True).reshape(-1, 3)
    if out.shape[0] != pts.shape[0]:
        raise ValueError("normals must be n x 3")
    cnt = None if knn_count is None else np.ascontiguousarray(knn_count, dtype=np.uint32)
    reached = C.c_uint64(0)
    check(_capi.load().pcpx_propagate_normal_orientations(_vp(pts), pts.shape[0], _vp(nbr), None if cnt is None else _vp(cnt),
                                                          nbr.shape[1], _vp(out), C.byref(reached)))
    return out, int(reached.value)
