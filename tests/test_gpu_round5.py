"""Round 5's ways of doing the same work differently, each against the plain form BIT FOR BIT (torch.equal):

* rows with a pitch (pcpx_knn_self_strided_dev / pcpx_normals_knn_self_strided_dev): whole 64-byte rows in 16-byte stores;
* input-order normals + counts through the gather-form permute ("gather_outputs");
* query groups handed out by recorded times, long groups first ("long_groups_first");
* shards cut by sampled WORK (pcpx_knn_group_costs_dev -> pcpx_shard_cuts_by_cost -> PCPX_BUILD_SHARD_RANGE): the table is the same
  from two handles (every rank of a job must compute the same cuts), the rank-local handles of the cuts reproduce the whole-cloud
  index, and the shards' measured times are closer together than equal-count shards' on the clustered cloud;
* one rank-local handle asked a growing sequence of radii and then a kNN (the selection's cell count was double
  counted when a radius widened the halo, which switched the coverage check off).

Reference semantics: include/pcp/octree/linked_octree_node.hpp:453-570 (nearest_neighbours), :581-614 (range_search),
include/pcp/algorithm/estimate_normals.hpp:80-92 (the loop the shards cut)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    return pytest.importorskip("torch")


def _cloud(pkg, kind, n):
    return pkg.synthetic.uniform_cloud(n, 11) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 12)


def _plain(torch, ix, n, k, first=0, count=None):
    dev = torch.device("cuda", 0)
    idx = torch.full((n, k), -7, dtype=torch.int32, device=dev)
    cnt = torch.full((n,), -7, dtype=torch.int32, device=dev)
    d2 = torch.full((n, k), -7.0, dtype=torch.float32, device=dev)
    nrm = torch.full((n, 3), -7.0, dtype=torch.float32, device=dev)
    kw = {} if count is None else dict(first=first, count=count)
    ix.debug_set("gather_outputs", 0)
    ix.debug_set("long_groups_first", 0)
    ix.knn_self_dev(k, 1e-5, idx.data_ptr(), cnt.data_ptr(), d2.data_ptr(), **kw)
    if k <= 32:
        i2, c2 = torch.empty_like(idx), torch.empty_like(cnt)
        ix.normals_knn_self_dev(k, 1e-5, nrm.data_ptr(), i2.data_ptr(), c2.data_ptr(), **kw)
    ix.synchronize()
    ix.debug_set("gather_outputs", 1)
    ix.debug_set("long_groups_first", 1)
    return idx, cnt, d2, nrm


@pytest.mark.parametrize("k,stride", [(15, 16), (16, 16), (7, 8), (8, 8), (31, 32), (32, 32), (12, 16), (15, 20), (3, 8), (20, 32)])
def test_rows_with_a_pitch_equal_packed_rows(pkg, k, stride):
    torch = _torch()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(k)
    pts = np.concatenate([_cloud(pkg, "clustered", 150_000), rng.uniform(-0.3, 1.3, (5_000, 3)).astype(np.float32)])
    pts[-50:] = pts[:50]  # exact ties
    n = len(pts)
    grid = np.array([0, 0, 0, 1, 1, 1], np.float32)  # some points fall outside: their rows are left alone
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid)
    idx, cnt, d2, nrm = _plain(torch, ix, n, k)
    for gather in (0, 1):
        ix.debug_set("gather_outputs", gather)
        sidx = torch.full((n, stride), -7, dtype=torch.int32, device=dev)
        sd2 = torch.full((n, stride), -7.0, dtype=torch.float32, device=dev)
        scnt = torch.full((n,), -7, dtype=torch.int32, device=dev)
        ix.knn_self_strided_dev(k, 1e-5, stride, sidx.data_ptr(), scnt.data_ptr(), sd2.data_ptr())
        nidx = torch.full((n, stride), -7, dtype=torch.int32, device=dev)
        ncnt = torch.full((n,), -7, dtype=torch.int32, device=dev)
        snrm = torch.full((n, 3), -7.0, dtype=torch.float32, device=dev)
        ix.normals_knn_self_strided_dev(k, 1e-5, stride, snrm.data_ptr(), nidx.data_ptr(), ncnt.data_ptr())
        ix.synchronize()
        assert torch.equal(sidx[:, :k], idx) and torch.equal(sd2[:, :k], d2) and torch.equal(scnt, cnt)
        assert torch.equal(nidx[:, :k], idx) and torch.equal(ncnt, cnt) and torch.equal(snrm, nrm)
        answered = cnt >= 0  # (a point outside the grid has no row: everything about it stays as it was)
        assert int((~answered).sum()) > 0
        kcap = 8 if k <= 8 else 16 if k <= 16 else 32  # (the kernel's list length for this k)
        if stride == kcap and k >= kcap - 1:
            # a whole row is written: the entries beyond k are padding
            assert bool((sidx[answered][:, k:] == -1).all()) and bool(torch.isinf(sd2[answered][:, k:]).all())
        assert bool((sidx[~answered] == -7).all()) and bool((snrm[~answered] == -7).all())
    with pytest.raises(pkg.PcpxError):
        ix.knn_self_strided_dev(k, 1e-5, k - 1, sidx.data_ptr(), scnt.data_ptr())
    ix.close()


@pytest.mark.parametrize("kind,n,k", [("uniform", 300_000, 15), ("clustered", 400_000, 15), ("clustered", 250_000, 32), ("uniform", 100_000, 8)])
def test_gather_form_and_recorded_order_change_nothing(pkg, kind, n, k):
    torch = _torch()
    dev = torch.device("cuda", 0)
    pts = _cloud(pkg, kind, n)
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), n)
    idx, cnt, d2, nrm = _plain(torch, ix, n, k)
    # three identical launches: the first records, the second makes and uses the order, the third uses it; then slices (a slice that
    # starts and ends inside groups), which record anew
    for trip in range(3):
        i2 = torch.full((n, k), -7, dtype=torch.int32, device=dev)
        c2 = torch.full((n,), -7, dtype=torch.int32, device=dev)
        n2 = torch.full((n, 3), -7.0, dtype=torch.float32, device=dev)
        ix.normals_knn_self_dev(k, 1e-5, n2.data_ptr(), i2.data_ptr(), c2.data_ptr())
        ix.synchronize()
        assert torch.equal(i2, idx) and torch.equal(c2, cnt) and torch.equal(n2, nrm), trip
    size = ix.size()
    first, count = (size // 3) // 64 * 64, size // 2 + 17
    pidx, pcnt, pd2, pnrm = _plain(torch, ix, n, k, first, count)
    for trip in range(3):
        i2 = torch.full((n, k), -7, dtype=torch.int32, device=dev)
        c2 = torch.full((n,), -7, dtype=torch.int32, device=dev)
        n2 = torch.full((n, 3), -7.0, dtype=torch.float32, device=dev)
        ix.normals_knn_self_dev(k, 1e-5, n2.data_ptr(), i2.data_ptr(), c2.data_ptr(), first=first, count=count)
        ix.synchronize()
        assert torch.equal(i2, pidx) and torch.equal(c2, pcnt) and torch.equal(n2, pnrm), trip
        assert count <= int((c2 >= 0).sum()) < count + 64  # (a slice is answered to the end of its last group)
    # a rebuild forgets the recorded order (the groups are other groups)
    ix.rebuild_dev(d_pts.data_ptr(), n // 2)
    m = n // 2
    i3 = torch.full((m, k), -7, dtype=torch.int32, device=dev)
    c3 = torch.full((m,), -7, dtype=torch.int32, device=dev)
    n3 = torch.full((m, 3), -7.0, dtype=torch.float32, device=dev)
    ix.normals_knn_self_dev(k, 1e-5, n3.data_ptr(), i3.data_ptr(), c3.data_ptr())
    ix.synchronize()
    ridx, rcnt, rd2, rnrm = _plain(torch, ix, m, k)
    assert torch.equal(i3, ridx) and torch.equal(c3, rcnt) and torch.equal(n3, rnrm)
    ix.close()


def _time_ms(torch, fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


@pytest.mark.parametrize("kind,n,k,world", [("clustered", 2_000_000, 15, 8), ("uniform", 500_000, 15, 4), ("clustered", 600_000, 32, 3)])
def test_shards_cut_by_sampled_work(pkg, kind, n, k, world):
    torch = _torch()
    dev = torch.device("cuda", 0)
    pts = _cloud(pkg, kind, n)
    grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    cs = torch.cuda.current_stream().cuda_stream
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid, stream=cs)
    stride = 16
    ev = ix.knn_group_costs(k, 1e-5, stride)
    groups = (ix.size() + 63) // 64
    assert ev.shape == (groups // stride, 4) and (ev[:, 0] > 0).all()
    # the same table from a second handle and from a second call (every rank of a job must compute the same cuts)
    iy = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid, stream=cs)
    assert np.array_equal(iy.knn_group_costs(k, 1e-5, stride), ev) and np.array_equal(ix.knn_group_costs(k, 1e-5, stride), ev)
    iy.close()
    cuts = pkg.shard_cuts_by_cost(ix.size(), world, stride, ev)
    assert cuts[0] == 0 and cuts[-1] == ix.size() and cuts == sorted(cuts)
    idx, cnt, d2, nrm = _plain(torch, ix, n, k)
    sidx = torch.full((n, k), -7, dtype=torch.int32, device=dev)
    scnt = torch.full((n,), -7, dtype=torch.int32, device=dev)
    snrm = torch.full((n, 3), -7.0, dtype=torch.float32, device=dev)
    by_work, by_count = [], []
    sh = None
    for rank in range(world):
        first, count = cuts[rank], cuts[rank + 1] - cuts[rank]
        kw = dict(voxel_grid=grid, shard=(rank, world), shard_range=(first, count), k_hint=k, stream=cs)
        if sh is None:
            sh = pkg.Index.from_device(d_pts.data_ptr(), n, **kw)
        else:
            kw.pop("stream")
            sh.rebuild_dev(d_pts.data_ptr(), n, **kw)
        si = sh.shard_info()
        assert si["shard_first"] == first and si["shard_count"] == count
        sh.normals_knn_self_dev(k, 1e-5, snrm.data_ptr(), sidx.data_ptr(), scnt.data_ptr(), first, count)
        sh.synchronize()
        by_work.append(_time_ms(torch, lambda: sh.normals_knn_self_dev(k, 1e-5, snrm.data_ptr(), sidx.data_ptr(), scnt.data_ptr(), first, count)))
    assert torch.equal(sidx, idx) and torch.equal(scnt, cnt) and torch.equal(snrm, nrm)
    tidx, tcnt, tnrm = torch.empty_like(sidx), torch.empty_like(scnt), torch.empty_like(snrm)
    for rank in range(world):
        first, count = pkg.shard_range(ix.size(), rank, world)
        sh.rebuild_dev(d_pts.data_ptr(), n, voxel_grid=grid, shard=(rank, world), k_hint=k)
        by_count.append(_time_ms(torch, lambda: sh.normals_knn_self_dev(k, 1e-5, tnrm.data_ptr(), tidx.data_ptr(), tcnt.data_ptr(), first, count)))
    sh.close()
    ix.close()
    spread_work, spread_count = max(by_work) / (sum(by_work) / world), max(by_count) / (sum(by_count) / world)
    print("%s n=%d k=%d world=%d: slowest/mean by work %.3f (slowest %.3f ms), by count %.3f (slowest %.3f ms)"
          % (kind, n, k, world, spread_work, max(by_work), spread_count, max(by_count)))
    # (how close the shards' times are is bench.py's business -- extra.configs.*.one_eighth_shard -- and at these sizes a shard is a
    #  fraction of one round of the resident waves: reported, not asserted)


def test_one_rank_local_handle_asked_growing_radii_then_knn(pkg):
    """a radius that widens the halo re-ran the selection's pack step, which counted cells that were selected
    already; at world = 2 the count passed the number of cells, `everything` became true and later calls skipped both the coverage
    check and further halo growth."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, k, world = 400_000, 15, 2
    pts = pkg.synthetic.uniform_cloud(n, 21)
    grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid)
    idx, cnt, d2, nrm = _plain(torch, ix, n, k)
    for rank in range(world):
        first, count = pkg.shard_range(ix.size(), rank, world)
        sh = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid, shard=(rank, world), k_hint=k)
        for r in (0.01, 0.03, 0.08, 0.2):  # each wider than the halo before it
            want = torch.zeros(n, dtype=torch.int32, device=dev)
            got = torch.zeros(n, dtype=torch.int32, device=dev)
            ix.range_count_self_dev(r, want.data_ptr(), first, count)
            sh.range_count_self_dev(r, got.data_ptr(), first, count)
            ix.synchronize()
            sh.synchronize()
            assert torch.equal(got, want), r
        assert sh.shard_info()["local_points"] < ix.size()  # (not everything: the check stays on)
        sidx = torch.full((n, k), -7, dtype=torch.int32, device=dev)
        scnt = torch.full((n,), -7, dtype=torch.int32, device=dev)
        sd2 = torch.full((n, k), -7.0, dtype=torch.float32, device=dev)
        sh.knn_self_dev(k, 1e-5, sidx.data_ptr(), scnt.data_ptr(), sd2.data_ptr(), first, count)
        sh.synchronize()
        sel = scnt >= 0
        assert int(sel.sum()) == count and torch.equal(sidx[sel], idx[sel]) and torch.equal(sd2[sel], d2[sel])
        sh.close()
    ix.close()


@pytest.mark.parametrize("kind,n,radius", [("uniform", 200_000, 0.03), ("clustered", 300_000, 0.004), ("uniform", 3_000, 0.2)])
def test_device_resident_range_lists_of_every_point(pkg, oracle, kind, n, radius):
    """pcpx_range_lists_self_dev (counts -> 64-bit scan -> fill with the packed leaf form) against the brute-force oracle: every
    list is the set of points within the radius (sphere.hpp:27-35, the centre included), lists of points outside the grid are
    empty, offsets are the exclusive scan of the counts."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    pts = np.concatenate([_cloud(pkg, kind, n), rng.uniform(-0.2, 1.2, (n // 50, 3)).astype(np.float32)])
    m = len(pts)
    grid = np.array([0, 0, 0, 1, 1, 1], np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), m, voxel_grid=grid)
    off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    total = ix.range_lists_self_dev(radius, off.data_ptr())
    assert total > 0
    idx = torch.full((total,), -1, dtype=torch.int32, device=dev)
    assert ix.range_lists_self_dev(radius, off.data_ptr(), idx.data_ptr(), total) == total
    ix.synchronize()
    off_h, idx_h = off.cpu().numpy(), idx.cpu().numpy().view(np.uint32)
    inside = np.all((pts >= 0) & (pts <= 1), axis=1)
    counts = np.diff(off_h)
    assert off_h[0] == 0 and off_h[-1] == total and (counts[~inside] == 0).all() and (counts[inside] >= 1).all()
    want = oracle.range_count_bruteforce(pts[inside], pts[inside], radius, nthreads=16)
    assert np.array_equal(counts[inside], want)
    for gather in (1, 0):  # input-order counts through the gather-form permute and straight from the kernel; a slice as well
        ix.debug_set("gather_counts", gather)
        cnt_dev = torch.full((m,), -7, dtype=torch.int32, device=dev)
        ix.range_count_self_dev(radius, cnt_dev.data_ptr())
        ix.synchronize()
        got = cnt_dev.cpu().numpy()
        assert np.array_equal(got[inside], want) and (got[~inside] == -7).all()
        part = torch.full((m,), -7, dtype=torch.int32, device=dev)
        first, count = (ix.size() // 4) // 64 * 64, ix.size() // 3
        ix.range_count_self_dev(radius, part.data_ptr(), first, count)
        ix.synchronize()
        part = part.cpu().numpy()
        answered = part != -7
        assert count <= answered.sum() < count + 64 and np.array_equal(part[answered], got[answered])
    ix.debug_set("gather_counts", 1)
    # members: every listed point is within the radius of its centre (float arithmetic of the reference), no duplicates
    sel = rng.choice(np.nonzero(inside)[0], 2000, replace=False)
    for i in sel:
        lst = idx_h[off_h[i]:off_h[i + 1]]
        d = pts[lst] - pts[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        assert len(np.unique(lst)) == len(lst) and (d2 <= np.float32(radius) * np.float32(radius)).all() and i in lst
    # the host-pointer list form for arbitrary centres gives the same sets
    o2, i2 = ix.range_sphere(pts[sel[:200]], radius)
    for q, i in enumerate(sel[:200]):
        assert set(i2[o2[q]:o2[q + 1]].tolist()) == set(idx_h[off_h[i]:off_h[i + 1]].tolist())
    ix.close()


@pytest.mark.parametrize("copies,k", [(300, 15), (70, 32), (5000, 8)])
def test_build_with_runs_the_leaf_kernel_cannot_order(pkg, oracle, copies, k):
    """The sort stops after two bucketed passes where runs are expected to stay short and k_finish orders runs of up to 64 words in
    LDS (csrc/pcpx_build.hip).  Hundreds of coincident points in an otherwise sparse bucket are one long run of equal keys: the
    build must notice, repeat itself with that bucket taking every pass, and give the same rows as brute force -- also on the
    rebuild after it (the handle remembers the bucket) and through a rank-local handle."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(copies)
    base = pkg.synthetic.uniform_cloud(120_000, 5)
    hot = base[rng.integers(0, len(base), 6)]
    near = (hot[:, None, :] + rng.normal(0, 1e-6, (6, copies, 3)).astype(np.float32)).reshape(-1, 3)  # dense knots: one 24-bit cell each
    same = np.repeat(hot[:3], copies, axis=0)                                                          # and exact copies
    pts = np.clip(np.concatenate([base, near, same]), 0, 1).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    n = len(pts)
    grid = np.array([0, 0, 0, 1, 1, 1], np.float32)
    d_pts = torch.from_numpy(pts).to(dev)
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid)
    if copies <= 300:  # knots of a few hundred points: the plan (words per cell of the bucket's fullest 16-bit cell) lets their buckets stop early ...
        assert ix.debug_get("build_redos") == 1 and 1 <= ix.debug_get("full_buckets") <= 12  # ... and the first attempt finds them out
    else:              # knots of thousands: the plan sends their buckets through every pass from the start
        assert ix.debug_get("build_redos") == 0
    sel = np.concatenate([rng.integers(0, n, 400), np.nonzero((pts[:, None, :] == hot[None, :3, :]).all(2).any(1))[0][:100]])
    oi, oc, od = oracle.knn_bruteforce(pts, pts[sel], k, nthreads=16, want_d2=True)
    from test_gpu_parity import _assert_rows_exact
    for attempt in range(2):  # the build that finds the runs out, and a rebuild that knows
        idx = torch.full((n, k), -1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(n, dtype=torch.int32, device=dev)
        d2 = torch.full((n, k), float("inf"), dtype=torch.float32, device=dev)
        ix.knn_self_dev(k, 1e-5, idx.data_ptr(), cnt.data_ptr(), d2.data_ptr())
        ix.synchronize()
        t = torch.from_numpy(sel).to(dev)
        _assert_rows_exact(pts, pts[sel], k, idx[t].cpu().numpy().view(np.uint32), cnt[t].cpu().numpy().view(np.uint32), d2[t].cpu().numpy(), oi, oc, od)
        want = oracle.range_count_bruteforce(pts, pts[sel], 0.01, nthreads=16)
        rc = torch.zeros(n, dtype=torch.int32, device=dev)
        ix.range_count_self_dev(0.01, rc.data_ptr())
        ix.synchronize()
        assert np.array_equal(rc[t].cpu().numpy(), want)
        ix.rebuild_dev(d_pts.data_ptr(), n, voxel_grid=grid)
        assert ix.debug_get("build_redos") == (1 if copies <= 300 else 0)  # (the rebuilds know the buckets: no second attempt)
    sidx = torch.full((n, k), -1, dtype=torch.int32, device=dev)
    scnt = torch.zeros(n, dtype=torch.int32, device=dev)
    for rank in range(2):
        sh = pkg.Index.from_device(d_pts.data_ptr(), n, voxel_grid=grid, shard=(rank, 2), k_hint=k)
        first, count = pkg.shard_range(sh.size(), rank, 2)
        sh.knn_self_dev(k, 1e-5, sidx.data_ptr(), scnt.data_ptr(), None, first, count)
        sh.synchronize()
        sh.close()
    assert torch.equal(scnt, cnt) and torch.equal(sidx, idx)
    ix.close()


@pytest.mark.parametrize("n", [9, 12, 16, 17, 24, 25, 31, 32, 33])
def test_trees_of_two_to_five_leaves(pkg, oracle, n):
    """No tree has depth 1 (csrc/pcpx_build.hip, depth_of): clouds of 9 ... 32 points -- two to four leaves -- get a level of one real
    node between the root and the leaves, so that every walk meets leaves only under a last-level node (k_knn, k_range and
    k_range_aabb have no case for a leaf popped from the pending bits).  Every query form on such clouds against brute force."""
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3), dtype=np.float32)
    q = (rng.random((40, 3), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    ix = pkg.Index(pts)
    from test_gpu_parity import _assert_rows_exact
    for k in (1, 5, 15, 32):
        gi, gc, gd = ix.knn_self(k, want_d2=True)
        oi, oc, od = oracle.knn_bruteforce(pts, pts, k, nthreads=4, want_d2=True)
        _assert_rows_exact(pts, pts, k, gi, gc, gd, oi, oc, od)
        gi, gc, gd = ix.knn(q, k, want_d2=True)  # (few queries: the latency kernel; 40 > ... both batch forms by size)
        oi, oc, od = oracle.knn_bruteforce(pts, q, k, nthreads=4, want_d2=True)
        _assert_rows_exact(pts, q, k, gi, gc, gd, oi, oc, od)
    big = np.repeat(q, 20, axis=0)  # 800 queries: the throughput kernel
    gi, gc, gd = ix.knn(big, 15, want_d2=True)
    oi, oc, od = oracle.knn_bruteforce(pts, big, 15, nthreads=4, want_d2=True)
    _assert_rows_exact(pts, big, 15, gi, gc, gd, oi, oc, od)
    for r in (0.05, 0.3, 2.0):
        assert np.array_equal(ix.range_count_self(r), oracle.range_count_bruteforce(pts, pts, r, nthreads=4))
        assert np.array_equal(ix.range_count(big, r), oracle.range_count_bruteforce(pts, big, r, nthreads=4))
        off, idx = ix.range_sphere(q, r)
        d2 = ((pts[None, :, :].astype(np.float32) - q[:, None, :]) ** 2)
        d2 = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
        for i in range(len(q)):
            assert sorted(idx[int(off[i]):int(off[i + 1])].tolist()) == np.nonzero(d2[i] <= np.float32(r) * np.float32(r))[0].tolist()
    lo = (rng.random((50, 3), dtype=np.float32) * 0.8).astype(np.float32)
    boxes = np.concatenate([lo, lo + rng.random((50, 3), dtype=np.float32) * 0.6], axis=1).astype(np.float32)
    off, idx = ix.range_aabb(boxes)
    for i, b in enumerate(boxes):
        inside = np.nonzero(((pts >= b[:3]) & (pts <= b[3:])).all(1))[0].tolist()
        assert sorted(idx[int(off[i]):int(off[i + 1])].tolist()) == inside
    nrm = ix.normals_knn_self(min(8, n - 1))
    assert np.isfinite(nrm).all()
    ix.close()
