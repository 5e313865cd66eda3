"""The rank-local index of the multi-GPU path (PCPX_BUILD_SHARD, csrc/pcpx_shard.hip) on ONE GPU: `world` rank-local
handles, one after the other, must reproduce the whole-cloud index's answers BIT FOR BIT -- neighbour indices, counts,
squared distances, normals -- on uniform, clustered and adversarial clouds (far-apart blobs with stragglers between them,
where the initial halo cannot be enough and the coverage check must send queries round again), at BASELINE configs[3] and
configs[4] full size, for k <= 32 and the multi-pass k > 32, for radius counts, and for the rows-by-curve-position form.

The whole-cloud index itself is pinned against the brute-force oracle elsewhere (test_gpu_parity.py, test_gpu_configs.py);
sampled rows are checked against brute force here too.  Reference semantics: include/pcp/octree/linked_octree_node.hpp:453-570
(nearest_neighbours), :581-614 (range_search); benchmark shape: benchmark/spatial_data_structures_benchmark.cpp:108-148, :243-264."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    return pytest.importorskip("torch")


def _whole(pkg, torch, d_pts, n, k, grid, want_normals=True):
    dev = d_pts.device
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid)
    idx = torch.full((n, k), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d2 = torch.full((n, k), float("inf"), dtype=torch.float32, device=dev)
    nrm = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    ix.knn_self_dev(k, 1e-5, idx.data_ptr(), cnt.data_ptr(), d2.data_ptr())
    if want_normals and k <= 32:
        tmp_i, tmp_c = torch.empty_like(idx), torch.empty_like(cnt)
        ix.normals_knn_self_dev(k, 1e-5, nrm.data_ptr(), tmp_i.data_ptr(), tmp_c.data_ptr())
    ix.synchronize()
    return ix, idx, cnt, d2, nrm


def _by_shards(pkg, torch, d_pts, n, k, grid, world, want_normals=True, k_hint=None, borrow=False, info=None):
    """The same outputs from `world` rank-local handles (one handle, rebuilt per rank)."""
    dev = d_pts.device
    idx = torch.full((n, k), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d2 = torch.full((n, k), float("inf"), dtype=torch.float32, device=dev)
    nrm = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    ix = None
    covered = 0
    for rank in range(world):
        kw = dict(voxel_grid=grid, shard=(rank, world), k_hint=k if k_hint is None else k_hint, borrow=borrow)
        if ix is None:
            ix = pkg.Index.from_device(d_pts.data_ptr(), n, **kw)
        else:
            ix.rebuild_dev(d_pts.data_ptr(), n, **kw)
        size = ix.size()
        first, count = pkg.shard_range(size, rank, world)
        si = ix.shard_info()
        assert si["shard_first"] == first and si["shard_count"] == count
        assert si["core_first"] <= first and si["core_first"] + si["core_count"] >= first + count
        ix.knn_self_dev(k, 1e-5, idx.data_ptr(), cnt.data_ptr(), d2.data_ptr(), first, count)
        if want_normals and k <= 32:
            tmp_i, tmp_c = torch.empty((n, k), dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
            ix.normals_knn_self_dev(k, 1e-5, nrm.data_ptr(), tmp_i.data_ptr(), tmp_c.data_ptr(), first, count)
        ix.synchronize()
        if info is not None:
            info.append(ix.shard_info())
        covered += count
    ix.close()
    return idx, cnt, d2, nrm, covered


def _blobs_with_stragglers(n, seed):
    """Tight blobs far apart and a thin scatter of stragglers between them: a straggler's k-th neighbour is far away, in
    cells no halo of a sensible width reaches."""
    rng = np.random.default_rng(seed)
    nb = 24
    centres = rng.uniform(0.08, 0.92, (nb, 3))
    per = (n - n // 200) // nb
    parts = [c + rng.normal(0, 0.004, (per, 3)) for c in centres]
    parts.append(rng.uniform(0, 1, (n - per * nb, 3)))
    pts = np.clip(np.concatenate(parts), 0, 1).astype(np.float32)
    return pts[rng.permutation(len(pts))]


CLOUDS = {
    "uniform": lambda pkg, n: pkg.synthetic.uniform_cloud(n, 7),
    "clustered": lambda pkg, n: pkg.synthetic.clustered_cloud(n, 8),
    "blobs": lambda pkg, n: _blobs_with_stragglers(n, 9),
}


@pytest.mark.parametrize("kind,n,k,world", [
    ("uniform", 200_000, 15, 8), ("uniform", 1_000_000, 32, 8), ("uniform", 300_000, 7, 3), ("uniform", 5_000, 15, 8),
    ("clustered", 1_000_000, 15, 8), ("clustered", 400_000, 32, 2),
    ("blobs", 600_000, 15, 8), ("blobs", 300_000, 32, 5),
    ("uniform", 150_000, 40, 4), ("blobs", 200_000, 70, 8),
])
def test_rank_local_indexes_reproduce_the_whole_cloud_index(pkg, oracle, kind, n, k, world):
    torch = _torch()
    dev = torch.device("cuda", 0)
    pts = CLOUDS[kind](pkg, n)
    n = len(pts)
    grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix, idx, cnt, d2, nrm = _whole(pkg, torch, d_pts, n, k, grid)
    info = []
    sidx, scnt, sd2, snrm, covered = _by_shards(pkg, torch, d_pts, n, k, grid, world, info=info)
    assert covered == n
    assert torch.equal(scnt, cnt) and torch.equal(sidx, idx) and torch.equal(sd2, d2)
    assert torch.equal(snrm, nrm)
    if kind == "blobs":
        # the point of that cloud: the first halo is not enough, the coverage check must have sent queries round again
        assert sum(i["enlargements"] for i in info) > 0
    if kind == "uniform" and n >= 200_000:
        # ... and on a uniform cloud it is: a rank-local tree holds a fraction of the cloud
        assert max(i["local_points"] for i in info) < 0.6 * n
    from test_gpu_parity import _assert_rows_exact
    sel = np.random.default_rng(1).integers(0, n, 256)
    oi, oc, od = oracle.knn_bruteforce(pts, pts[sel], k, nthreads=16, want_d2=True)
    t = torch.from_numpy(sel).to(dev)
    _assert_rows_exact(pts, pts[sel], k, sidx[t].cpu().numpy().view(np.uint32), scnt[t].cpu().numpy().view(np.uint32), sd2[t].cpu().numpy(), oi, oc, od)
    ix.close()


def test_rank_local_index_with_points_outside_the_grid_and_duplicates(pkg):
    """Points outside the voxel grid are not inserted (linked_octree_node.hpp:174-175) and take no part in the shards;
    eight-fold coincident points tie exactly at every distance, so any change of order between the rank-local and the
    whole-cloud sort would show."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    base = rng.uniform(-0.2, 1.2, (40_000, 3)).astype(np.float32)
    pts = np.repeat(base, 8, axis=0)[rng.permutation(320_000)]
    grid = np.array([0, 0, 0, 1, 1, 1], np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    n, k = len(pts), 15
    ix, idx, cnt, d2, nrm = _whole(pkg, torch, d_pts, n, k, grid)
    assert ix.size() < n
    sidx, scnt, sd2, snrm, covered = _by_shards(pkg, torch, d_pts, n, k, grid, 8)
    assert covered == ix.size()
    assert torch.equal(scnt, cnt) and torch.equal(sidx, idx) and torch.equal(sd2, d2) and torch.equal(snrm, nrm)
    ix.close()


def test_rank_local_index_reads_a_borrowed_cloud_and_survives_rebuilds(pkg):
    """PCPX_BUILD_BORROW_CLOUD + the streaming shape: the same handle rebuilt on a moved cloud keeps the cells it has learnt."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, k, world, rank = 500_000, 15, 4, 1
    base = _blobs_with_stragglers(n, 11)
    grid = np.array([-0.01, -0.01, -0.01, 1.01, 1.01, 1.01], np.float32)
    ix = None
    learnt = []
    for it in range(3):
        pts = pkg.synthetic.jitter(base, 20 + it)
        d_pts = torch.from_numpy(pts).to(dev)
        wix, idx, cnt, d2, nrm = _whole(pkg, torch, d_pts, n, k, grid, want_normals=False)
        kw = dict(voxel_grid=grid, shard=(rank, world), k_hint=k, borrow=True)
        if ix is None:
            ix = pkg.Index.from_device(d_pts.data_ptr(), n, **kw)
        else:
            ix.rebuild_dev(d_pts.data_ptr(), n, **kw)
        first, count = pkg.shard_range(ix.size(), rank, world)
        sidx = torch.full((n, k), -1, dtype=torch.int32, device=dev)
        scnt = torch.zeros(n, dtype=torch.int32, device=dev)
        ix.knn_self_dev(k, 1e-5, sidx.data_ptr(), scnt.data_ptr(), None, first, count)
        ix.synchronize()
        learnt.append(ix.shard_info())
        rows = scnt > 0
        assert int(rows.sum()) == count
        assert torch.equal(sidx[rows], idx[rows]) and torch.equal(scnt[rows], cnt[rows])
        wix.close()
    assert learnt[0]["last_failed"] > 0
    assert learnt[2]["last_failed"] < learnt[0]["last_failed"]  # what was learnt on the first cloud serves the moved ones
    ix.close()


def test_rank_local_radius_counts(pkg, oracle):
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, world = 400_000, 8
    pts = pkg.synthetic.clustered_cloud(n, 5)
    grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid)
    d_perm = torch.empty(n, dtype=torch.int32, device=dev)
    ix.perm_dev(d_perm.data_ptr())
    for radius in (0.004, 0.03, 0.11):  # the last one is far wider than the first halo
        want = torch.zeros(n, dtype=torch.int32, device=dev)
        ix.range_count_self_dev(radius, want.data_ptr())
        # counts at curve positions (pcpx_range_count_self_curve_order_dev): the same counts, found through the order
        by_pos = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ix.range_count_self_curve_order_dev(radius, by_pos.data_ptr())
        ix.synchronize()
        assert torch.equal(by_pos, want[d_perm.long()])
        got = torch.zeros(n, dtype=torch.int32, device=dev)
        got_pos = torch.full((n,), -1, dtype=torch.int32, device=dev)
        sx = None
        for rank in range(world):
            kw = dict(voxel_grid=grid, shard=(rank, world), k_hint=8)
            if sx is None:
                sx = pkg.Index.from_device(d_pts.data_ptr(), n, **kw)
            else:
                sx.rebuild_dev(d_pts.data_ptr(), n, **kw)
            first, count = pkg.shard_range(n, rank, world)
            sx.range_count_self_dev(radius, got.data_ptr(), first, count)
            sx.range_count_self_curve_order_dev(radius, got_pos.data_ptr(), first, count)  # a rank's piece of the whole order
            sx.synchronize()
        sx.close()
        ix.synchronize()
        assert torch.equal(got, want) and torch.equal(got_pos, by_pos)
        sel = np.random.default_rng(2).integers(0, n, 200)
        assert np.array_equal(got.cpu().numpy().view(np.uint32)[sel], oracle.range_count_bruteforce(pts, pts[sel], radius, nthreads=16))
    ix.close()


def test_rows_by_curve_position_on_the_device(pkg):
    """pcpx_knn_self_curve_order_dev + pcpx_index_perm_dev: row p belongs to input point perm[p]; whole-cloud and rank-local."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, k, world = 700_000, 15, 8
    pts = pkg.synthetic.uniform_cloud(n, 13)
    grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix, idx, cnt, d2, nrm = _whole(pkg, torch, d_pts, n, k, grid)
    c_idx = torch.empty_like(idx)
    c_cnt = torch.empty_like(cnt)
    c_d2 = torch.empty_like(d2)
    c_nrm = torch.empty_like(nrm)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    pos = torch.empty(n, dtype=torch.int32, device=dev)
    ix.knn_self_curve_order_dev(k, 1e-5, c_idx.data_ptr(), c_cnt.data_ptr(), c_d2.data_ptr(), c_nrm.data_ptr())
    ix.perm_dev(perm.data_ptr(), pos.data_ptr())
    ix.synchronize()
    p = perm.long()
    assert torch.equal(torch.sort(p).values, torch.arange(n, device=dev))
    assert torch.equal(pos.long()[p], torch.arange(n, device=dev))
    assert torch.equal(c_idx, idx[p]) and torch.equal(c_cnt, cnt[p]) and torch.equal(c_d2, d2[p]) and torch.equal(c_nrm, nrm[p])
    s_idx = torch.full_like(idx, -1)
    s_cnt = torch.zeros_like(cnt)
    s_nrm = torch.zeros_like(nrm)
    s_perm = torch.full((n,), -1, dtype=torch.int32, device=dev)
    sx = None
    for rank in range(world):
        kw = dict(voxel_grid=grid, shard=(rank, world), k_hint=k)
        if sx is None:
            sx = pkg.Index.from_device(d_pts.data_ptr(), n, **kw)
        else:
            sx.rebuild_dev(d_pts.data_ptr(), n, **kw)
        first, count = pkg.shard_range(n, rank, world)
        sx.knn_self_curve_order_dev(k, 1e-5, s_idx.data_ptr(), s_cnt.data_ptr(), None, s_nrm.data_ptr(), first, count)
        sx.perm_dev(s_perm.data_ptr(), None)
        sx.synchronize()
    sx.close()
    assert torch.equal(s_perm, perm) and torch.equal(s_idx, c_idx) and torch.equal(s_cnt, c_cnt) and torch.equal(s_nrm, c_nrm)
    ix.close()


def test_what_a_rank_local_index_refuses(pkg):
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    pts = pkg.synthetic.uniform_cloud(50_000, 3)
    ix = pkg.Index(pts, shard=(1, 4), k_hint=15)
    assert ix.size() == len(pts)
    with pytest.raises(pkg.PcpxError) as e:
        ix.knn(pts[:10], 5)
    assert e.value.status == capi.PCPX_ERR_UNSUPPORTED
    with pytest.raises(pkg.PcpxError) as e:
        ix.knn_self(5)
    assert e.value.status == capi.PCPX_ERR_UNSUPPORTED
    with pytest.raises(pkg.PcpxError) as e:
        ix.range_sphere(pts[:4], 0.1)
    assert e.value.status == capi.PCPX_ERR_UNSUPPORTED
    torch = _torch()
    dev = torch.device("cuda", 0)
    d_idx = torch.empty((len(pts), 5), dtype=torch.int32, device=dev)
    d_cnt = torch.empty(len(pts), dtype=torch.int32, device=dev)
    with pytest.raises(pkg.PcpxError) as e:  # positions outside this rank's shard
        ix.knn_self_dev(5, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, 0, 64)
    assert e.value.status == capi.PCPX_ERR_INVALID
    ix.rebuild(pts)  # the same handle as a whole-cloud index again
    idx, cnt = ix.knn_self(5)
    assert np.all(cnt == 5)
    ix.close()
    p = capi.BuildParams()
    p.struct_size = C.sizeof(capi.BuildParams)
    p.flags = capi.PCPX_BUILD_SHARD
    p.shard_rank, p.shard_world = 4, 4
    h = C.c_void_p(None)
    assert capi.load().pcpx_index_create(pts.ctypes.data_as(C.c_void_p), len(pts), C.byref(p), 0, C.byref(h)) == capi.PCPX_ERR_INVALID


def test_coarse_order_with_a_grid_and_points_outside_it(pkg, oracle):
    """PCPX_BUILD_COARSE_ORDER sorted the last top-digit bucket -- where the words of points OUTSIDE the voxel grid
    go -- on its upper digits only, so an inside point of the curve's last cells could end up behind an outside word and be
    dropped.  Outside points first in the input, inside points crowded into the corner where the curve ends."""
    rng = np.random.default_rng(17)
    n_out, n_corner, n_rest = 30_000, 500, 100_000
    outside = rng.uniform(1.5, 2.5, (n_out, 3)).astype(np.float32)
    grid = np.array([0, 0, 0, 1, 1, 1], np.float32)
    # the curve ends in one of the grid's corner cells ((xmax, ymin, zmin) for Skilling's transpose; tests/cpp/test_curve.hip
    # pins the curve): points within 1 / 16384 of the extent of EVERY corner, and on the corners themselves
    corners = []
    for c in range(8):
        at = np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1], np.float32)
        blob = at + rng.uniform(0, 1.0 / 16384, (n_corner, 3)).astype(np.float32) * (1 - 2 * at)
        blob[0] = at
        corners.append(blob)
    rest = rng.uniform(0, 1, (n_rest, 3)).astype(np.float32)
    pts = np.concatenate([outside] + corners + [rest]).astype(np.float32)
    inside = np.all((pts >= 0) & (pts <= 1), axis=1)
    full = pkg.Index(pts, voxel_grid=grid)
    coarse = pkg.Index(pts, voxel_grid=grid, coarse_order=True)
    assert full.size() == coarse.size() == int(inside.sum())
    k = 8
    fi, fc, fd = full.knn_self(k, want_d2=True)
    ci, cc, cd = coarse.knn_self(k, want_d2=True)
    assert np.array_equal(fc, cc) and np.array_equal(fd[inside], cd[inside])
    assert np.all(cc[inside] == k) and np.all(cc[~inside] == 0)
    assert not np.isin(ci[inside], np.nonzero(~inside)[0]).any()
    sel = np.concatenate([np.arange(n_out, n_out + 8 * n_corner, 37), rng.integers(0, len(pts), 64)])
    sel = sel[inside[sel]]
    oi, oc, od = oracle.knn_bruteforce(pts[inside], pts[sel], k, nthreads=8, want_d2=True)
    assert np.array_equal(cd[sel], od)
    full.close()
    coarse.close()


def test_config4_cloud_rank_local_indexes_bit_for_bit(pkg, oracle):
    """BASELINE configs[3]'s cloud at full size: 10 M clustered points, k = 15, 8 rank-local indexes == the whole-cloud index."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, k, world = 10_000_000, 15, 8
    pts = pkg.synthetic.clustered_cloud(n, 44)
    grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix, idx, cnt, d2, nrm = _whole(pkg, torch, d_pts, n, k, grid)
    info = []
    sidx, scnt, sd2, snrm, covered = _by_shards(pkg, torch, d_pts, n, k, grid, world, info=info)
    assert covered == n
    assert torch.equal(scnt, cnt) and torch.equal(sidx, idx) and torch.equal(sd2, d2) and torch.equal(snrm, nrm)
    print("C4 rank-local trees:", [(i["local_points"], i["last_failed"], i["enlargements"]) for i in info])
    ix.close()


def test_config5_cloud_rank_local_indexes_bit_for_bit(pkg, oracle):
    """BASELINE configs[4] at full size: 50 M uniform points, k = 32; 8 rank-local indexes == the whole-cloud index, and 2 048
    sampled rows == brute force."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, k, world = 50_000_000, 32, 8
    pts = pkg.synthetic.jitter(pkg.synthetic.uniform_cloud(n, 45), 46)
    grid = np.array([-0.001, -0.001, -0.001, 1.001, 1.001, 1.001], np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix, idx, cnt, d2, nrm = _whole(pkg, torch, d_pts, n, k, grid, want_normals=False)
    ix.close()
    info = []
    sidx, scnt, sd2, snrm, covered = _by_shards(pkg, torch, d_pts, n, k, grid, world, want_normals=False, borrow=True, info=info)
    assert covered == n
    assert torch.equal(scnt, cnt) and torch.equal(sidx, idx) and torch.equal(sd2, d2)
    print("C5 rank-local trees:", [(i["local_points"], i["last_failed"], i["enlargements"]) for i in info])
    assert max(i["local_points"] for i in info) < 0.25 * n
    from test_gpu_parity import _assert_rows_exact
    sel = np.random.default_rng(55).integers(0, n, 2048)
    t = torch.from_numpy(sel).to(dev)
    oi, oc, od = oracle.knn_bruteforce(pts, pts[sel], k, nthreads=16, want_d2=True)
    _assert_rows_exact(pts, pts[sel], k, sidx[t].cpu().numpy().view(np.uint32), scnt[t].cpu().numpy().view(np.uint32), sd2[t].cpu().numpy(), oi, oc, od)


def test_the_host_build_of_the_sort_word_orders_points_like_the_device(pkg):
    """tests/test_multirank_cpu.py cuts its shards in the order of csrc/pcpx_curve.h compiled for the host: that order is the
    device's (pcpx_index_perm_dev), points outside the grid included."""
    torch = _torch()
    from test_multirank_cpu import _curve_order
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    pts = np.concatenate([pkg.synthetic.clustered_cloud(300_000, 9), rng.uniform(-0.3, 1.3, (20_000, 3)).astype(np.float32)])
    pts = pts[rng.permutation(len(pts))]
    grid = np.array([0, 0, 0, 1, 1, 1], np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), len(pts), voxel_grid=grid)
    perm = torch.empty(ix.size(), dtype=torch.int32, device=dev)
    ix.perm_dev(perm.data_ptr(), None)
    ix.synchronize()
    order, words = _curve_order(pts, grid)
    assert np.array_equal(perm.cpu().numpy().view(np.uint32), order[: ix.size()].astype(np.uint32))
    ix.close()
