"""BASELINE.json's configs at FULL size on the GPU (run with -m gpu): size-independent properties checked on the device
over every row, sampled rows bit-exact against the brute-force oracle, and the evidence for the normals beyond the
reference's single degenerate known-answer test.

  C4  10 M clustered (Gaussian mixture) points, k = 15, fused kNN + normals, 8 query shards
  C5  50 M uniform points, k = 32, jitter -> index rebuild -> kNN, two iterations (streaming)
  normals: analytic known-answer clouds, float64 eigh cross-check of GPU normals on the bunny / C2 / C4 clouds
"""
import importlib

import numpy as np
import pytest

from conftest import analytic_normal_cases

pytestmark = pytest.mark.gpu

COS_TOL = 1e-4   # north_star: "normals within 1e-4 cosine of reference"
GAP_TOL = 1e-3   # rows with relative eigen-gap (l1 - l0) / l2 below this are "ill-conditioned
                 # in the reference itself" and are reported separately


def _device_row_properties(torch, d_pts, d_idx, d_cnt, d_d2, k, chunk=2_000_000):
    """Every row: k neighbours, ascending d2, never the query itself, valid indices, and d2 bit-equal to the reference's
    float expression dx*dx + dy*dy + dz*dz (include/pcp/common/norm.hpp:102-112; torch evaluates each product and sum as
    its own float32 kernel, so nothing is fused) for the returned index."""
    n = d_pts.shape[0]
    assert bool((d_cnt == k).all())
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = d_idx[a:b].long()
        d2 = d_d2[a:b]
        assert bool((d2[:, 1:] >= d2[:, :-1]).all()), "rows must be ascending"
        assert int(idx.min()) >= 0 and int(idx.max()) < n
        rows = torch.arange(a, b, device=d_pts.device).unsqueeze(1)
        assert not bool((idx == rows).any()), "the query point itself must be excluded"
        q = d_pts[a:b].unsqueeze(1)
        d = d_pts[idx] - q
        dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
        ref = (dx * dx + dy * dy) + dz * dz
        assert torch.equal(ref, d2), "d2 must be the reference expression of the returned index"
        del idx, d, dx, dy, dz, ref


def _sampled_rows_exact(oracle, pts, sel, k, gi, gc, gd):
    from test_gpu_parity import _assert_rows_exact
    oi, oc, od = oracle.knn_bruteforce(pts, pts[sel], k, nthreads=16, want_d2=True)
    _assert_rows_exact(pts, pts[sel], k, gi, gc, gd, oi, oc, od)


def test_config5_streaming_50m_k32(pkg, oracle):
    """BASELINE configs[4] on one GPU: 50 M uniform points (seed 45), each iteration jitters the cloud by U(-1e-3, 1e-3)
    (seed 45 + it), rebuilds the index in place and answers k = 32 for every point."""
    torch = pytest.importorskip("torch")
    n, k = 50_000_000, 32
    dev = torch.device("cuda", 0)
    base = pkg.synthetic.uniform_cloud(n, 45)
    d_idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_d2 = torch.empty((n, k), dtype=torch.float32, device=dev)
    ix = None
    for it in (1, 2):
        pts = pkg.synthetic.jitter(base, 45 + it)
        d_pts = torch.from_numpy(pts).to(dev)
        torch.cuda.synchronize()
        if ix is None:
            ix = pkg.Index.from_device(d_pts.data_ptr(), n)
        else:
            ix.rebuild_dev(d_pts.data_ptr(), n)  # the streaming step: same handle, new coordinates
        assert ix.size() == n
        d_cnt.zero_()
        ix.knn_self_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), d_d2.data_ptr())
        ix.synchronize()
        _device_row_properties(torch, d_pts, d_idx, d_cnt, d_d2, k)
        sel = np.random.default_rng(50 + it).integers(0, n, 2048)
        t_sel = torch.from_numpy(sel).to(dev)
        _sampled_rows_exact(oracle, pts, sel, k, d_idx[t_sel].cpu().numpy().view(np.uint32), d_cnt[t_sel].cpu().numpy().view(np.uint32),
                            d_d2[t_sel].cpu().numpy())
        del d_pts
    ix.close()


def _eigh_check(pts, rows, idx, nrm):
    """float64 numpy.linalg.eigh on the centred scatter matrix of each sampled row's neighbours (the definition of
    pcp::estimate_normal, normal_estimation.hpp:41-77, in double precision) against the GPU's float32 normals.
    Returns (max 1-|cos| over the well-conditioned rows, ill-conditioned fraction)."""
    nb = pts[idx[rows].astype(np.int64)].astype(np.float64)          # rows x k x 3
    v = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("rki,rkj->rij", v, v)
    w, vec = np.linalg.eigh(cov)                                      # ascending
    ref = vec[:, :, 0]
    gap = (w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300)
    well = gap >= GAP_TOL
    cos = np.abs((ref * nrm[rows].astype(np.float64)).sum(1))
    err = 1.0 - cos
    return float(err[well].max()), float(1.0 - well.mean()), int(well.sum())


def test_config4_clustered_10m_k15_sharded(pkg, oracle):
    """BASELINE configs[3] cloud at full size on one GPU: fused kNN + normals, the 8 Morton shards of the multi-GPU
    path equal the single call, rows have the kNN properties, sampled rows are bit-exact against brute force."""
    torch = pytest.importorskip("torch")
    n, k = 10_000_000, 15
    dev = torch.device("cuda", 0)
    pts = pkg.synthetic.clustered_cloud(n, 44)
    d_pts = torch.from_numpy(pts).to(dev)
    torch.cuda.synchronize()
    ix = pkg.Index.from_device(d_pts.data_ptr(), n)
    assert ix.size() == n
    full_idx = torch.full((n, k), -1, dtype=torch.int32, device=dev)
    full_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    full_nrm = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    d_d2 = torch.empty((n, k), dtype=torch.float32, device=dev)
    ix.normals_knn_self_dev(k, 1e-5, full_nrm.data_ptr(), full_idx.data_ptr(), full_cnt.data_ptr())
    tmp_cnt = torch.zeros_like(full_cnt)
    tmp_idx = torch.empty_like(full_idx)
    ix.knn_self_dev(k, 1e-5, tmp_idx.data_ptr(), tmp_cnt.data_ptr(), d_d2.data_ptr())
    ix.synchronize()
    assert torch.equal(tmp_idx, full_idx) and torch.equal(tmp_cnt, full_cnt)
    _device_row_properties(torch, d_pts, full_idx, full_cnt, d_d2, k)
    # the 8 query shards (pcpx_shard_range) cover the cloud exactly once and reproduce the single call
    sh_idx = torch.full_like(full_idx, -1)
    sh_cnt = torch.zeros_like(full_cnt)
    sh_nrm = torch.zeros_like(full_nrm)
    covered = 0
    for rank in range(8):
        first, count = pkg.shard_range(n, rank, 8)
        ix.normals_knn_self_dev(k, 1e-5, sh_nrm.data_ptr(), sh_idx.data_ptr(), sh_cnt.data_ptr(), first, count)
        covered += count
    ix.synchronize()
    assert covered == n
    assert torch.equal(sh_idx, full_idx) and torch.equal(sh_cnt, full_cnt) and torch.equal(sh_nrm, full_nrm)
    # sampled rows against the oracle: brute-force kNN, the restated Eigen solver, and float64 eigh
    idx = full_idx.cpu().numpy().view(np.uint32)
    cnt = full_cnt.cpu().numpy().view(np.uint32)
    nrm = full_nrm.cpu().numpy()
    sel = np.random.default_rng(44).integers(0, n, 2048)
    _sampled_rows_exact(oracle, pts, sel, k, idx[sel], cnt[sel], d_d2.cpu().numpy()[sel])
    big = np.random.default_rng(45).integers(0, n, 120_000)
    on = oracle.normals_from_knn(pts, idx[big], cnt[big], nthreads=16)
    cos = np.abs((on.astype(np.float64) * nrm[big].astype(np.float64)).sum(1))
    assert (1.0 - cos).max() <= COS_TOL
    assert np.mean(np.all(on == nrm[big], axis=1)) > 0.999  # same arithmetic in the same order: bit-identical in practice
    err, ill, nwell = _eigh_check(pts, big, idx, nrm)
    print("C4 clustered 10M: %d sampled rows, %d well-conditioned, max 1-|cos| vs float64 eigh %.3e, ill-conditioned fraction %.5f"
          % (len(big), nwell, err, ill))
    assert err <= COS_TOL
    ix.close()


@pytest.mark.parametrize("cloud", ["bunny", "uniform_1m"])
def test_gpu_normals_against_float64_eigh(pkg, bunny, cloud):
    """Every bunny point (configs[0]) and 120 000 sampled rows of the 1 M uniform cloud (configs[1]): the GPU's
    float32 normals against numpy.linalg.eigh in float64 on the same neighbourhoods, with the ill-conditioned fraction
    reported (uniform clouds are where near-isotropic neighbourhoods -- meaningless normals -- occur)."""
    if cloud == "bunny":
        pts, rows = bunny, np.arange(len(bunny))
    else:
        pts = pkg.synthetic.uniform_cloud(1_000_000, 42)
        rows = np.random.default_rng(42).integers(0, len(pts), 120_000)
    ix = pkg.Index(pts)
    nrm, idx, cnt = ix.normals_knn_self(15, want_knn=True)
    assert np.all(cnt == 15)
    err, ill, nwell = _eigh_check(pts, rows, idx, nrm)
    print("%s: %d rows, %d well-conditioned, max 1-|cos| vs float64 eigh %.3e, ill-conditioned fraction %.5f"
          % (cloud, len(rows), nwell, err, ill))
    assert err <= COS_TOL
    assert nwell >= 0.9 * len(rows)


def test_estimate_normal_forms_agree_with_the_oracle_bit_for_bit(pkg, oracle):
    """pcpx_estimate_normal has three forms by neighbourhood size -- up to 64 points in the kernel arguments with a polled
    completion word (the reference's per-point shape), up to 4096 through the pinned stage, more through device buffers: the
    same sums in the same order, so the same bits as the oracle's restatement (normal_estimation.hpp:41-77) at every size,
    on both sides of each boundary, and call after call (the completion word is reused)."""
    rng = np.random.default_rng(23)
    for m in (1, 2, 3, 7, 15, 63, 64, 65, 100, 4096, 4097, 6000, 10, 64, 9):
        pts = (rng.standard_normal((m, 3)) * np.array([1.0, 0.3, 0.02]) + np.array([5.0, -2.0, 0.5])).astype(np.float32)
        n_gpu = pkg.estimate_normal(pts)
        n_orc = oracle.estimate_normal(pts)
        assert np.array_equal(n_gpu.view(np.uint32), n_orc.view(np.uint32)), m


def test_gpu_normals_on_analytic_cases(pkg, oracle):
    """Closed-form normals over exact scatter matrices that force the solver's Householder step and >= 2 QR steps
    (conftest.analytic_normal_cases): GPU == analytic within 1e-6 cosine and == the oracle bit for bit, both through
    pcpx_estimate_normal and through the fused kNN kernel (the cloud plus one far query point whose neighbours are
    exactly the cloud)."""
    for name, pts, normal, gap in analytic_normal_cases():
        n_gpu = pkg.estimate_normal(pts)
        n_orc = oracle.estimate_normal(pts)
        assert np.array_equal(n_gpu.view(np.uint32), n_orc.view(np.uint32)), name
        assert 1.0 - abs(float(n_gpu.astype(np.float64) @ normal)) <= 1e-6, name
        far = (pts.astype(np.float64).mean(0) + 1000.0 * normal).astype(np.float32)[None, :]
        ix = pkg.Index(np.concatenate([pts, far]))
        nrm, idx, cnt = ix.normals_knn_self(len(pts), want_knn=True)
        assert cnt[-1] == len(pts) and sorted(idx[-1].tolist()) == list(range(len(pts)))
        # the fused kernel sums the neighbours in row (distance) order, so only the analytic bar applies here
        assert 1.0 - abs(float(nrm[-1].astype(np.float64) @ normal)) <= 1e-6, name
        on = oracle.normals_from_knn(np.concatenate([pts, far]), idx[-1:], cnt[-1:])
        assert np.array_equal(on[0].view(np.uint32), nrm[-1].view(np.uint32)), name


def test_host_calls_equal_the_device_resident_path(pkg):
    """Host-pointer self queries stage through the handle's pooled device buffers (pcpx_api.hip, self_queries_to_host);
    the rows, counts, distances and normals must be the ones the *_dev forms write into the caller's device memory --
    bit for bit -- and stay so after the pool has been trimmed."""
    torch = pytest.importorskip("torch")
    n, k = 2_600_000, 15
    dev = torch.device("cuda", 0)
    pts = pkg.synthetic.clustered_cloud(n, 44)
    ix = pkg.Index(pts)
    nrm, idx, cnt = ix.normals_knn_self(k, want_knn=True)
    idx2, cnt2, d2 = ix.knn_self(k, want_d2=True)
    assert np.array_equal(idx, idx2) and np.array_equal(cnt, cnt2)
    d_pts = torch.from_numpy(pts).to(dev)
    torch.cuda.synchronize()
    dix = pkg.Index.from_device(d_pts.data_ptr(), n)
    d_idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_d2 = torch.empty((n, k), dtype=torch.float32, device=dev)
    dix.normals_knn_self_dev(k, 1e-5, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr())
    dix.knn_self_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), d_d2.data_ptr())
    dix.synchronize()
    assert np.array_equal(d_idx.cpu().numpy().view(np.uint32), idx)
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint32), cnt)
    assert np.array_equal(d_d2.cpu().numpy().view(np.uint32), d2.view(np.uint32))
    assert np.array_equal(d_nrm.cpu().numpy().view(np.uint32), nrm.view(np.uint32))
    # the handle keeps its staging buffers between calls; trim gives them back
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    capi.check(capi.load().pcpx_index_trim(ix._h))
    assert np.array_equal(ix.knn_self(k)[0], idx)


def test_latency_path_equals_the_general_path(pkg, oracle):
    """pcpx_knn_batch with a handful of queries takes the one-wavefront-per-query kernel (pcpx_few.hip); more than 512
    queries take the sorted batch path.  Same rows (up to exact k-th-distance ties), both against brute force."""
    from test_gpu_parity import _assert_rows_exact
    rng = np.random.default_rng(77)
    for pts in (pkg.synthetic.uniform_cloud(300_000, 9), pkg.synthetic.clustered_cloud(200_000, 44),
                rng.uniform(-100, 100, (50_000, 3)).astype(np.float32), pkg.synthetic.uniform_cloud(37, 3)):
        ix = pkg.Index(pts)
        lo, hi = pts.min(0), pts.max(0)
        q = (lo + (hi - lo) * (rng.random((300, 3)) * 1.3 - 0.15)).astype(np.float32)  # some outside the cloud's box
        q[:40] = pts[rng.integers(0, len(pts), 40)]  # some coincide with indexed points
        for k in (1, 10, 15, 32):
            gi, gc, gd = ix.knn(q, k, want_d2=True)           # latency path (300 <= 512 queries)
            oi, oc, od = oracle.knn_bruteforce(pts, q, k, nthreads=16, want_d2=True)
            _assert_rows_exact(pts, q, k, gi, gc, gd, oi, oc, od)
            one_i, one_c = ix.knn(q[7:8], k)                  # a single query
            assert np.array_equal(one_c, gc[7:8]) and np.array_equal(one_i, gi[7:8])
        big = np.concatenate([q, q, q])[:700]
        bi, bc, bd = ix.knn(big, 15, want_d2=True)            # general path
        gi, gc, gd = ix.knn(q, 15, want_d2=True)
        assert np.array_equal(bd[:300], gd) and np.array_equal(bc[:300], gc)
    # duplicates and eps = 0: the query's own coordinates are then a neighbour at distance 0
    base = rng.random((2000, 3), dtype=np.float32)
    pts = np.concatenate([base, base[:500], base[:100]])
    ix = pkg.Index(pts)
    for eps in (1e-5, 0.0):
        gi, gc, gd = ix.knn(base[:200], 8, eps=eps, want_d2=True)
        oi, oc, od = oracle.knn_bruteforce(pts, base[:200], 8, eps=eps, nthreads=16, want_d2=True)
        _assert_rows_exact(pts, base[:200], 8, gi, gc, gd, oi, oc, od)


@pytest.mark.parametrize("cloud", ["uniform_10m_seed43", "clustered_10m_seed44"])
def test_contiguous_curve_block_against_the_restated_octree(pkg, oracle, cloud):
    """BASELINE configs[2] / [3] at full size: ONE CONTIGUOUS block of a million rows of the curve order (what a rank's shard
    is), not scattered samples, against the oracle's restatement of basic_linked_octree_t (capacity 32, depth 21, auto
    bounding box; test/octree/octree_knn.cpp:184-254 is the reference's own shape of this check): k = 15 rows row for row,
    r = 0.01 counts count for count (clustered cloud: the block's first 2^16 rows); and every one of the 10 M normals is finite and of unit length
    (test/algorithm/estimate_normals.cpp:48-64 checks exactly that)."""
    torch = pytest.importorskip("torch")
    n, k, block = 10_000_000, 15, 1_000_000
    dev = torch.device("cuda", 0)
    pts = pkg.synthetic.uniform_cloud(n, 43) if cloud.startswith("uniform") else pkg.synthetic.clustered_cloud(n, 44)
    d_pts = torch.from_numpy(pts).to(dev)
    torch.cuda.synchronize()
    ix = pkg.Index.from_device(d_pts.data_ptr(), n)
    d_idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_d2 = torch.empty((n, k), dtype=torch.float32, device=dev)
    d_nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_rc = torch.empty(n, dtype=torch.int32, device=dev)
    d_perm = torch.empty(n, dtype=torch.int32, device=dev)
    ix.knn_self_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), d_d2.data_ptr())
    ix.normals_knn_self_dev(k, 1e-5, d_nrm.data_ptr())
    ix.range_count_self_dev(0.01, d_rc.data_ptr())
    ix.perm_dev(d_perm.data_ptr())
    d_rc_pos = torch.empty(n, dtype=torch.int32, device=dev)
    ix.range_count_self_curve_order_dev(0.01, d_rc_pos.data_ptr())
    ix.synchronize()
    assert torch.equal(d_rc_pos, d_rc[d_perm.long()])  # the counts at curve positions are the same counts
    del d_rc_pos
    # every normal: finite, unit length
    assert bool(torch.isfinite(d_nrm).all())
    assert float(((d_nrm * d_nrm).sum(1) - 1.0).abs().max()) <= 1e-5
    # the block: positions [first, first + block) of the curve order, deliberately not aligned to a query group
    first = 3_333_337
    ids = d_perm[first:first + block].long()
    assert int(ids.min()) >= 0 and int(ids.max()) < n and int(torch.unique(ids).numel()) == block
    gi = d_idx[ids].cpu().numpy().view(np.uint32)
    gc = d_cnt[ids].cpu().numpy().view(np.uint32)
    gd = d_d2[ids].cpu().numpy()
    grc = d_rc[ids].cpu().numpy().view(np.uint32)
    ids = ids.cpu().numpy()
    q = pts[ids]
    tree = oracle.Octree(pts)  # reference defaults
    assert tree.size() == n
    oi, oc, od = tree.knn(q, k, nthreads=16, want_d2=True)
    # counts and distance lists bit for bit; indices equal except inside runs of EXACTLY equal distance, whose order the
    # reference leaves to its heap (linked_octree_node.hpp:479-489): there the index sets of the run are equal, or -- at the
    # k-th distance, where the run may be cut -- the returned point really is at that distance
    assert np.array_equal(gc, oc) and int(oc.min()) == k
    assert np.array_equal(gd, od)
    differ = np.nonzero((gi != oi).any(1))[0]
    assert len(differ) < block // 100
    for r in differ:
        for d in np.unique(gd[r][gi[r] != oi[r]]):
            run = gd[r] == d
            if d != gd[r, -1]:
                assert set(gi[r][run].tolist()) == set(oi[r][run].tolist()), "row %d: different points at distance %r" % (r, d)
            else:
                e = pts[gi[r][run].astype(np.int64)] - q[r][None, :]
                assert np.all(((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]).astype(np.float32) == d)
                assert len(set(gi[r].tolist())) == k
    # (r = 0.01 holds ~10^5 points inside a cluster: the restated octree counts the block's first 2^16 rows there -- contiguous
    #  as well -- in the time it counts the whole block in the uniform cloud)
    rc_rows = block if cloud.startswith("uniform") else 1 << 16
    assert np.array_equal(grc[:rc_rows], tree.range_count(q[:rc_rows], 0.01, nthreads=16))
    ix.close()
