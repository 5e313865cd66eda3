#!/usr/bin/env python3
"""bench.py -- kNN + normal-estimation throughput of the pcpx hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): the north-star target configuration of BASELINE.json -- a 10 M-point
uniform synthetic cloud, k = 15, every point queries the cloud (kNN) and gets its PCA normal.  The
1 M-point configs[1] figure is reported under "extra" with --with-1m (DESIGN.md "Measurement" says why
10 M is the headline: it is the cloud the target is quoted on and 1 M is 3 query groups per resident wave).

One STEP = one pass of the query hot path over the whole cloud with the index resident in HBM: one
k_knn launch (k nearest neighbours of every point, rows written to HBM, with the 3x3 scatter matrix +
eigen-solve of every point's neighbourhood fused in, normals written to HBM).  The index build is timed
separately and reported in "extra" (the reference's own benchmarks also build once and time queries:
benchmark/spatial_data_structures_benchmark.cpp:243-264); "extra.value_incl_build" gives the rate with a
rebuild inside every step.

Other workloads (--workload; never the driver's default): clustered_10m_k15 (configs[3]'s cloud) and
uniform_50m_k32_stream (configs[4]: every step re-jitters the cloud, rebuilds the index and answers
k = 32 for every point -- the rebuild is INSIDE the step there).

Multi-GPU: one process per GPU.  Every rank holds the whole cloud (120 MB; exact kNN needs all
candidates), computes the bounding box of ITS slice of the input, the per-rank boxes are all-gathered
with RCCL (24 B per rank, the only collective), and every rank builds a RANK-LOCAL index on the union box
(PCPX_BUILD_SHARD: curve keys for every point, sort + leaves + boxes only for the cells its contiguous
64-aligned shard of the curve-sorted queries can reach; results are bit-identical to the whole-cloud
index, csrc/pcpx_shard.hip) and answers that shard.  Total work is fixed => "strong".
PCPX_BENCH_REPLICATED=1 makes every rank index the whole cloud instead (rounds 1-3).

The default single-GPU run also reports BASELINE.json's other configs under extra.configs -- each
measured in a child process of its own, so that the headline kernel's per-launch average stays that of the
headline workload -- with, for the multi-GPU configs, one rank's share of an 8-rank step timed on this GPU
for every rank in turn (a PROJECTION of the 8-GPU step, labelled as such), the reference's own benchmark
shapes (single-query kNN / range latency) and one end-to-end figure through the host-pointer ABI.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
PROFILE_TAG = "r05"  # the committed counter files this run quotes (profiles/<tag>_hbm_traffic.json, <tag>_issue.json)

WORKLOADS = {
    # name: (generator, n, seed, k)
    "uniform_10m_k15": ("uniform", 10_000_000, 43, 15),
    "uniform_1m_k15": ("uniform", 1_000_000, 42, 15),
    "clustered_10m_k15": ("clustered", 10_000_000, 44, 15),
    "uniform_10m_k8": ("uniform", 10_000_000, 43, 8),
    "uniform_50m_k32_stream": ("uniform", 50_000_000, 45, 32),
    "uniform_10m_k32_stream": ("uniform", 10_000_000, 45, 32),
}
STREAMING = ("uniform_50m_k32_stream", "uniform_10m_k32_stream")  # rebuild inside the step, kNN rows only


def make_cloud(pkg, kind, n, seed):
    if kind == "uniform":
        return pkg.synthetic.uniform_cloud(n, seed)
    return pkg.synthetic.clustered_cloud(n, seed)


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (the GPU box gives one GPU's
    share of a many-core host)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(pts, k, budget_s=12.0):
    """Reference algorithms and parameters (octree capacity 32 / depth 21 / auto bbox + estimate_normals,
    driven like examples/simple_example.cpp:83-99) from the oracle restatement, on all host threads, over a
    bounded sample of the same cloud's queries.  kind = "port": the real reference cannot be built here.
    Beside it, the figure "as the reference would get it on this box": the same loop under std::execution::par
    (the policy the reference's examples pass), which libstdc++ runs on ONE thread when it has no TBB backend."""
    from oracle import pcp_oracle as O
    threads = host_cores()
    t0 = time.perf_counter()
    tree = O.Octree(pts)  # sequential insertion, like the reference
    build_s = time.perf_counter() - t0
    n = len(pts)
    probe = min(n, 4000 * threads)
    t0 = time.perf_counter()
    tree.estimate_normals(k, first=0, count=probe, nthreads=threads)
    dt = time.perf_counter() - t0
    rate = probe / dt
    sample = int(min(n - probe, max(probe, rate * budget_s)))
    if sample > 0:
        first = (n - sample) // 2
        t0 = time.perf_counter()
        tree.estimate_normals(k, first=first, count=sample, nthreads=threads)
        dt = time.perf_counter() - t0
        rate = sample / dt
    else:
        sample = probe
    par_n = int(min(n, max(2000, rate / threads * 3.0)))  # ~3 s on one thread
    t0 = time.perf_counter()
    _, par_threads = tree.estimate_normals_stdpar(k, first=(n - par_n) // 2, count=par_n)
    par_rate = par_n / (time.perf_counter() - t0)
    return {"value": rate / 1e6, "unit": "Mqueries/s", "cores": threads, "kind": "port",
            "sample": "%d of %d queries (kNN k=%d + PCA normal each) on the full %d-point cloud, oracle octree "
                      "(capacity 32, depth 21), %d threads; octree build %.1f s not included" % (sample, n, k, n, threads, build_s),
            "std_execution_par": {"value": par_rate / 1e6, "unit": "Mqueries/s", "threads_observed": par_threads,
                                  "sample": "%d queries under std::transform(std::execution::par, ...); libstdc++ without TBB "
                                            "runs the policy on the calling thread" % par_n}}


def row_pitch(k):
    """Entries between the rows of the index output: a row of k = 15 neighbours is written with a pitch of 16 -- one aligned 64-byte
    piece, in 16-byte stores (pcpx_knn_self_strided_dev; the reference returns a std::vector per query, so the pitch is the
    library's to choose).  PCPX_BENCH_ROW_PITCH=0: packed rows of k entries, as rounds 1-4 measured."""
    env = os.environ.get("PCPX_BENCH_ROW_PITCH", "auto")
    if env != "auto":
        return int(env)
    kcap = 8 if k <= 8 else 16 if k <= 16 else 32
    return kcap if k in (kcap - 1, kcap) else 0


def shard_cuts(pkg, torch, d_pts, n, k, grid, world, stream, dev_index, by_work):
    """world + 1 curve positions.  by_work: every rank indexes the whole cloud once (set-up, outside the step), runs the k-NN walk
    for one query group in 16 (pcpx_knn_group_costs_dev: deterministic event counts -- every rank computes the same table from its
    replica of the cloud, nothing is exchanged) and cuts the curve order into shards of equal estimated work
    (pcpx_shard_cuts_by_cost); otherwise equal query counts (pcpx_shard_range)."""
    if not by_work:
        return [pkg.shard_range(n, r, world)[0] for r in range(world)] + [n], None
    t0 = time.perf_counter()
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=dev_index, stream=stream, voxel_grid=grid)
    ev = ix.knn_group_costs(k, 1e-5, 16)
    cuts = pkg.shard_cuts_by_cost(ix.size(), world, 16, ev)
    ix.close()
    torch.cuda.synchronize()
    return cuts, (time.perf_counter() - t0) * 1e3


def _reduce(torch, dist, value, op, dev, rehearse):
    """All-reduce one number over the ranks (RCCL on the device; CPU tensor over gloo in the one-GPU rehearsal)."""
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if rehearse else dev)
    dist.all_reduce(t, op=op)
    return float(t.item())


def run_workload(pkg, torch, dist, name, rank, world, steps, warmup, want_profile, rehearse=False, side=True, collective=False,
                 at_curve_positions=False):
    kind, n, seed, k = WORKLOADS[name]
    dev = torch.device("cuda", torch.cuda.current_device())
    pts = make_cloud(pkg, kind, n, seed)
    d_pts = torch.from_numpy(pts).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    lib = capi.load()

    # per-rank bounding box of this rank's slice of the input, all-gathered over RCCL
    mg = importlib.import_module("point-cloud-processing_amd.multigpu")
    lo, hi = mg.input_slice(n, rank, world)
    d_box = torch.empty(6, dtype=torch.float32, device=dev)
    capi.check(lib.pcpx_bounding_box_dev(d_pts.data_ptr() + 12 * lo, hi - lo, dev.index, stream, d_box.data_ptr()))
    torch.cuda.current_stream().synchronize()
    d_box = mg.global_grid(d_box.cpu() if rehearse else d_box, dist, world, always=collective)  # the one collective: 24 B per rank over RCCL
    grid = d_box.cpu().numpy()

    streaming = name in STREAMING
    # N > 1: the rank-local index (PCPX_BUILD_SHARD); the cloud is read in place (PCPX_BUILD_BORROW_CLOUD) when it is rebuilt every step
    local = world > 1 and os.environ.get("PCPX_BENCH_REPLICATED") != "1"
    # ... of a shard cut by WORK for a static index (the streaming cloud moves every step and is uniform: equal counts)
    by_work = local and not streaming and os.environ.get("PCPX_BENCH_CUT", "work") == "work"
    cuts, cut_ms = shard_cuts(pkg, torch, d_pts, n, k, grid, world, stream, dev.index, by_work)
    first, count = cuts[rank], cuts[rank + 1] - cuts[rank]
    build_kw = dict(voxel_grid=grid, shard=(rank, world), k_hint=k, borrow=streaming) if local else dict(voxel_grid=grid)
    if by_work:
        build_kw["shard_range"] = (first, count)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=dev.index, stream=stream, **build_kw)
    torch.cuda.synchronize()
    first_build_ms = (time.perf_counter() - t0) * 1e3
    assert ix.size() == n
    if not by_work:
        first, count = mg.query_shard(n, rank, world)

    pitch = row_pitch(k)
    d_idx = torch.empty((n, pitch or k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)  # rows outside this rank's shard stay 0
    d_nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)

    if streaming:
        # configs[4]: the cloud moves between iterations (jitter U(-1e-3, 1e-3), seed 45 + it); the cloud and a jittered
        # copy alternate so every step rebuilds on coordinates that differ from the previous step's
        variants = [d_pts] + [torch.from_numpy(np.clip(pkg.synthetic.jitter(pts, seed + it), grid[:3], grid[3:])).to(dev)
                              for it in (1,)]
        it_no = [0]
        if not local:
            build_kw = dict(voxel_grid=grid, coarse_order=True)  # (this index answers one query pass before the next rebuild)

        def step():
            it_no[0] += 1
            ix.rebuild_dev(variants[it_no[0] % len(variants)].data_ptr(), n, **build_kw)
            ix.knn_self_strided_dev(k, 1e-5, pitch, d_idx.data_ptr(), d_cnt.data_ptr(), None, first, count)
    elif at_curve_positions:  # (--rows-at-curve-positions: the same launch with rows and normals at curve positions; for the counter passes)
        def step():
            ix.knn_self_curve_order_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, d_nrm.data_ptr(), first, count)
    else:
        def step():
            ix.normals_knn_self_strided_dev(k, 1e-5, pitch, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(), first, count)

    # the first step on its own: it includes what later ones do not repeat on a static index -- the rank-local index's coverage
    # check (N > 1) and the recording of the query groups' times that the later steps' order is made from
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_step_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(0, warmup - 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if want_profile:
        ix.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ix.profile_end() if want_profile else None
    complete = int((d_cnt == k).sum().item())  # rows this rank filled with k neighbours
    if world > 1:
        elapsed = _reduce(torch, dist, elapsed, dist.ReduceOp.MAX, dev, rehearse)
        complete = int(_reduce(torch, dist, complete, dist.ReduceOp.SUM, dev, rehearse))

    # rebuild cost (same grid), for the "incl. build" figure
    rebuild_ms = None
    if side:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reb = 5
        for _ in range(reb):
            ix.rebuild_dev(d_pts.data_ptr(), n, **build_kw)
        torch.cuda.synchronize()
        rebuild_ms = (time.perf_counter() - t0) * 1e3 / reb

    # config 3 shape on the same cloud: radius count r = 0.01 around every point (device resident)
    range_ms = range_pos_ms = None
    if side and name == "uniform_10m_k15" and world == 1:
        d_rc = torch.empty(n, dtype=torch.int32, device=dev)
        ix.range_count_self_dev(0.01, d_rc.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.range_count_self_dev(0.01, d_rc.data_ptr())
        torch.cuda.synchronize()
        range_ms = (time.perf_counter() - t0) * 1e3 / 3
        # the same counts written at curve positions (pcpx_range_count_self_curve_order_dev: one 256-B store per query group)
        ix.range_count_self_curve_order_dev(0.01, d_rc.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.range_count_self_curve_order_dev(0.01, d_rc.data_ptr())
        torch.cuda.synchronize()
        range_pos_ms = (time.perf_counter() - t0) * 1e3 / 3

    # ... and the LISTS of the same ranges (the reference's range_search returns the points, linked_octree_node.hpp:581-614): counts ->
    # 64-bit scan -> fill, device resident (pcpx_range_lists_self_dev); ~42 indices per list at r = 0.01
    range_lists_ms = range_lists_total = None
    if side and name == "uniform_10m_k15" and world == 1:
        d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        total = ix.range_lists_self_dev(0.01, d_off.data_ptr())
        d_lists = torch.empty(total, dtype=torch.int32, device=dev)
        ix.range_lists_self_dev(0.01, d_off.data_ptr(), d_lists.data_ptr(), total)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.range_lists_self_dev(0.01, d_off.data_ptr(), d_lists.data_ptr(), total)
        torch.cuda.synchronize()
        range_lists_ms, range_lists_total = (time.perf_counter() - t0) * 1e3 / 3, total
        del d_lists, d_off

    # the step with rows and normals written at curve positions (pcpx_knn_self_curve_order_dev) instead of input indices
    rows_pos_ms = None
    if side and not streaming and world == 1:
        ix.knn_self_curve_order_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, d_nrm.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.knn_self_curve_order_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, d_nrm.data_ptr())
        torch.cuda.synchronize()
        rows_pos_ms = (time.perf_counter() - t0) * 1e3 / 5

    res = {"n": n, "k": k, "elapsed": elapsed, "range_lists_ms": range_lists_ms, "range_lists_total": range_lists_total, "range_ms": range_ms, "range_pos_ms": range_pos_ms, "rows_pos_ms": rows_pos_ms, "steps": steps, "ms_per_step": elapsed * 1e3 / steps,
           "mqps": n * steps / elapsed / 1e6, "first_build_ms": first_build_ms, "rebuild_ms": rebuild_ms,
           "profile": prof, "shard": (first, count), "pts": pts, "complete": complete,
           "min_count": int(d_cnt.min().item()) if world == 1 else None, "grid": grid,
           "local_index": ix.shard_info() if local else None, "first_step_ms": first_step_ms, "row_pitch": pitch,
           "cut": ("work: one group in 16 sampled on a whole-cloud index, %.1f ms of set-up per rank" % cut_ms) if by_work else "equal query counts"}
    ix.close()
    return res


def eighth_shard_projection(pkg, torch, name, pts, grid, steps=3, world=8):
    """One rank's share of an 8-rank step on THIS GPU, for every rank in turn, with the rank-local index: the step time of the
    8-GPU job would be the slowest rank's.  A projection (one GPU, no RCCL), labelled so wherever it is quoted."""
    kind, n, seed, k = WORKLOADS[name]
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    streaming = name in STREAMING
    d_pts = torch.from_numpy(pts).to(dev)
    pitch = row_pitch(k)
    d_idx = torch.empty((n, pitch or k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_nrm = None if streaming else torch.empty((n, 3), dtype=torch.float32, device=dev)
    per_rank, builds, trees, firsts = [], [], [], []
    by_work = not streaming and os.environ.get("PCPX_BENCH_CUT", "work") == "work"
    cuts, cut_ms = shard_cuts(pkg, torch, d_pts, n, k, grid, world, stream, dev.index, by_work)
    ix = None
    for rank in range(world):
        first, count = cuts[rank], cuts[rank + 1] - cuts[rank]
        kw = dict(voxel_grid=grid, shard=(rank, world), k_hint=k, borrow=streaming)
        if by_work:
            kw["shard_range"] = (first, count)
        t0 = time.perf_counter()
        if ix is None:
            ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=dev.index, stream=stream, **kw)
        else:
            ix.rebuild_dev(d_pts.data_ptr(), n, **kw)

        def step():
            if streaming:
                ix.rebuild_dev(d_pts.data_ptr(), n, **kw)
                ix.knn_self_strided_dev(k, 1e-5, pitch, d_idx.data_ptr(), d_cnt.data_ptr(), None, first, count)
            else:
                ix.normals_knn_self_strided_dev(k, 1e-5, pitch, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(), first, count)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        firsts.append((time.perf_counter() - t0) * 1e3)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        per_rank.append((time.perf_counter() - t0) * 1e3 / steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            ix.rebuild_dev(d_pts.data_ptr(), n, **kw)
        torch.cuda.synchronize()
        builds.append((time.perf_counter() - t0) * 1e3 / steps)
        trees.append(ix.shard_info()["local_points"])
    ix.close()
    return {"projection": "one rank's share of an %d-rank step, every rank in turn on this one GPU (rank-local index, no RCCL); not a "
                          "measurement on %d GPUs.  ms_*_rank: the steady state of a static index asked the same question again (coverage "
                          "verified once, query groups handed out by the times the first step recorded); ms_first_step_slowest_rank: a "
                          "rank's first step, with both" % (world, world),
            "cut": ("shards of equal estimated work (pcpx_shard_cuts_by_cost; %.1f ms of set-up per rank)" % cut_ms) if by_work else "shards of equal query counts",
            "ms_slowest_rank": round(max(per_rank), 4), "ms_mean_rank": round(sum(per_rank) / world, 4),
            "slowest_over_mean": round(max(per_rank) / (sum(per_rank) / world), 3), "ms_first_step_slowest_rank": round(max(firsts), 4),
            "rank_local_build_ms_slowest": round(max(builds), 4), "rank_local_tree_points_max": max(trees),
            "rank_local_tree_fraction_of_cloud": round(max(trees) / n, 4)}


def config_child(pkg, name, steps, warmup):
    """extra.configs[name]: one of BASELINE.json's other configurations, in a process of its own."""
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    res = run_workload(pkg, torch, dist, name, 0, 1, steps, warmup, want_profile=True, side=True)
    kind, n, seed, k = WORKLOADS[name]
    streaming = name in STREAMING
    bytes_per_q = 12 + 4 * k + (0 if streaming else 12)
    out = {"points": n, "k": k, "mqps": round(res["mqps"], 1), "ms_per_step": round(res["ms_per_step"], 4), "steps": steps,
           "step": "index rebuild + kNN rows" if streaming else "kNN rows + normals",
           "index_rebuild_ms": round(res["rebuild_ms"], 4) if res["rebuild_ms"] is not None else None}
    prof = res["profile"]
    if prof and prof["knn"][0] > 0:
        avg_s = prof["knn"][1] / prof["knn"][0] / 1e3
        out["k_knn_avg_launch_ms"] = round(avg_s * 1e3, 4)
        out["knn_only_mqps"] = round(n / avg_s / 1e6, 1)
        out["frac"] = round(n * bytes_per_q / avg_s / HBM_PEAK, 6)
        out["algorithmic_bytes_per_query"] = bytes_per_q
    if name != "uniform_1m_k15":
        out["one_eighth_shard"] = shard_extras(pkg, torch, name, res)
    print(json.dumps(out), flush=True)


def shard_extras(pkg, torch, name, res, steps=3):
    """The 8-rank projection of a one-GPU run: one rank's share with the rank-local index, and what it makes of the speed-up."""
    shard = eighth_shard_projection(pkg, torch, name, res["pts"], res["grid"], steps=steps)
    shard["speedup_projected_8_ranks"] = round(res["ms_per_step"] / shard["ms_slowest_rank"], 2)
    if name in STREAMING and res["rebuild_ms"] is not None:
        # the build is inside the step: the ceiling of the 8-rank speed-up if the queries scaled perfectly, with the whole-cloud
        # build replicated on every rank (rounds 1-3) and with the rank-local build
        q_ms = res["ms_per_step"] - res["rebuild_ms"]
        shard["amdahl_ceiling_8_ranks_build_in_step"] = round(res["ms_per_step"] / (q_ms / 8.0 + shard["rank_local_build_ms_slowest"]), 2)
        shard["amdahl_ceiling_8_ranks_replicated_build_in_step"] = round(res["ms_per_step"] / (q_ms / 8.0 + res["rebuild_ms"]), 2)
    return shard


def end_to_end_child(pkg, name):
    """H2D of the cloud + index build + fused kNN + normals + D2H of rows and normals, through the
    host-pointer ABI (what a caller of the drop-in headers pays from a cold host array to results on the host); best of 3."""
    import ctypes as C
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    lib = capi.load()
    kind, n, seed, k = WORKLOADS[name]
    pts = make_cloud(pkg, kind, n, seed)
    nrm = np.zeros((n, 3), np.float32)
    idx = np.zeros((n, k), np.uint32)
    cnt = np.zeros(n, np.uint32)
    pos = np.zeros(n, np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {}
    for form in ("input_order", "curve_order"):
        best = None
        for rep in range(4):
            h = C.c_void_p(None)
            t0 = time.perf_counter()
            capi.check(lib.pcpx_index_create(vp(pts), n, None, 0, C.byref(h)))
            t1 = time.perf_counter()
            if form == "input_order":
                capi.check(lib.pcpx_normals_knn_self(h, k, 1e-5, vp(nrm), vp(idx), vp(cnt)))
            else:
                capi.check(lib.pcpx_normals_knn_self_curve_order(h, k, 1e-5, vp(nrm), vp(idx), vp(cnt), None, vp(pos)))
            t2 = time.perf_counter()
            lib.pcpx_index_destroy(h)
            t3 = time.perf_counter()
            if rep and (best is None or t3 - t0 < best[0]):
                best = (t3 - t0, t1 - t0, t2 - t1)
        out[form] = {"total_ms": round(best[0] * 1e3, 3), "create_from_host_ms": round(best[1] * 1e3, 3), "query_to_host_ms": round(best[2] * 1e3, 3),
                     "mqps": round(n / best[0] / 1e6, 1)}
    out["workload"] = name
    out["what"] = "pcpx_index_create(host xyz) + pcpx_normals_knn_self[_curve_order](host normals, rows, counts) + pcpx_index_destroy; pageable host memory"
    print(json.dumps(out), flush=True)


def latency_child():
    """The reference's own benchmark shapes (benchmark/spatial_data_structures_benchmark.cpp:108-148, :169-213, :243-264: one
    operation per iteration) through the drop-in C++ headers: tools/latency_bench.cpp, 2^20 points."""
    import subprocess
    pkg_dir = os.path.join(ROOT, "point-cloud-processing_amd")
    exe = "/tmp/pcpx_latency_bench_%d" % os.getpid()
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "latency_bench.cpp"), "-o", exe,
                    "-L", pkg_dir, "-lpcpx", "-Wl,-rpath," + pkg_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"], check=True, timeout=120)
    r = subprocess.run([exe, str(1 << 20), "2000"], capture_output=True, text=True, check=True, timeout=200)
    os.unlink(exe)
    print(r.stdout.strip().splitlines()[-1], flush=True)


def host_api_rates(pkg, pts, k):
    """What a caller of the host-pointer C ABI (= the drop-in C++ headers) gets on the bench cloud: H2D of the points
    is excluded (the index exists), outputs are host arrays allocated and touched beforehand, PCIe included."""
    import ctypes as C
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    lib = capi.load()
    n = len(pts)
    ix = pkg.Index(pts)
    nrm = np.zeros((n, 3), np.float32)
    idx = np.zeros((n, k), np.uint32)
    cnt = np.zeros(n, np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    pos = np.zeros(n, np.uint32)
    out = {}
    for name, fn in (("host_api_normals", lambda: lib.pcpx_normals_knn_self(ix._h, k, 1e-5, vp(nrm), None, None)),
                     # rows in curve order + the table of positions: finished slices are copied while later ones are computed
                     ("host_api_rows_curve_order", lambda: lib.pcpx_normals_knn_self_curve_order(ix._h, k, 1e-5, vp(nrm), vp(idx), vp(cnt), None, vp(pos))),
                     ("host_api_rows", lambda: lib.pcpx_normals_knn_self(ix._h, k, 1e-5, vp(nrm), vp(idx), vp(cnt)))):
        for _ in range(3):  # the GPU has idled while the host arrays were made: let its clocks come back up
            capi.check(fn())
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            capi.check(fn())
            best = min(best, time.perf_counter() - t0)
        out[name + "_ms"] = round(best * 1e3, 3)
        out[name + "_mqps"] = round(n / best / 1e6, 1)
    out["host_api_rows_bytes_to_host"] = (12 + 4 * k + 4) * n
    out["host_api_rows_curve_order_bytes_to_host"] = (12 + 4 * k + 4 + 4) * n
    for name in ("host_api_rows", "host_api_rows_curve_order"):
        out[name + "_GBps"] = round(out[name + "_bytes_to_host"] / out[name + "_ms"] / 1e6, 2)
    ix.close()
    return out, nrm, idx


def normals_evidence(pts, idx, nrm, sample=100_000):
    """SURVEY.md section 8(d) parity gate, on GPU output: float64 eigh of each sampled row's scatter matrix against the
    float32 normal; rows with relative eigen-gap (l1 - l0) / l2 < 1e-3 are "ill-conditioned in the reference itself"."""
    rows = np.random.default_rng(1).integers(0, len(pts), sample)
    nb = pts[idx[rows].astype(np.int64)].astype(np.float64)
    v = nb - nb.mean(axis=1, keepdims=True)
    w, vec = np.linalg.eigh(np.einsum("rki,rkj->rij", v, v))
    gap = (w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300)
    well = gap >= 1e-3
    err = 1.0 - np.abs((vec[:, :, 0] * nrm[rows].astype(np.float64)).sum(1))
    return {"rows_sampled": int(sample), "max_1_minus_abs_cos_vs_float64_eigh": float(err[well].max()),
            "ill_conditioned_fraction": float(1.0 - well.mean()), "tolerance": 1e-4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="uniform_10m_k15", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rows-at-curve-positions", action="store_true",
                    help="the step writes rows and normals at curve positions (pcpx_knn_self_curve_order_dev) instead of input indices: "
                         "for counter passes of that form; the default line is the input-order form")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the side measurements (rebuild, range count, host-pointer ABI rates, normal evidence): the "
                         "process then launches nothing but the timed steps, which is what the profiling scripts want")
    ap.add_argument("--host-api-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--config-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--end-to-end-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--latency-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--shard-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--base-ms", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--base-rebuild-ms", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip extra.configs (BASELINE.json's other configurations, the latency shapes and the end-to-end figure, each "
                         "in a child process: about two minutes)")
    ap.add_argument("--with-range-consumers", action="store_true",
                    help="also run tools/filter_bench.py (bilateral filter, WLOP: extras outside the hot path) and report it under extra")
    ap.add_argument("--with-1m", action="store_true",
                    help="also time configs[1] (1 M points, k=15) and report it under extra; off by default so that the "
                         "default command launches k_knn on the headline workload only (its rocprofv3 average then "
                         "equals roofline.avg_launch_ms)")
    args = ap.parse_args()

    pkg = importlib.import_module("point-cloud-processing_amd")
    if args.config_child:
        return config_child(pkg, args.config_child, 3 if args.config_child in STREAMING else 10, 2)
    if args.end_to_end_child:
        return end_to_end_child(pkg, args.workload)
    if args.latency_child:
        return latency_child()
    if args.shard_child:
        import torch
        torch.cuda.set_device(0)
        kind, n, seed, k = WORKLOADS[args.workload]
        pts = make_cloud(pkg, kind, n, seed)
        res = {"pts": pts, "grid": np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32), "ms_per_step": args.base_ms,
               "rebuild_ms": args.base_rebuild_ms if args.workload in STREAMING else None}
        print(json.dumps(shard_extras(pkg, torch, args.workload, res, steps=3 if args.workload in STREAMING else 5)), flush=True)
        return
    if args.host_api_child:
        # The host-pointer ABI side measurement runs in a process of its own (started by the main run below): its k_knn
        # launches follow 2-14 ms of copy each, the GPU clocks sag in between (5.6-6.3 ms per launch against 5.2 back to
        # back), and in the main process they would blur the per-kernel average that roofline.avg_launch_ms is checked against.
        kind, n, seed, k = WORKLOADS[args.workload]
        pts = make_cloud(pkg, kind, n, seed)
        rates, h_nrm, h_idx = host_api_rates(pkg, pts, k)
        rates["normals_check"] = normals_evidence(pts, h_idx, h_nrm)
        print(json.dumps(rates), flush=True)
        return

    import torch  # before libpcpx so that both share one HIP runtime
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the pcpx hot path has no CPU fallback")
    # PCPX_BENCH_REHEARSE=1: run the N > 1 code path on ONE GPU (all ranks on device 0, collectives over gloo) --
    # a functional rehearsal of sharding and reduction for tests, not a measurement.
    # PCPX_BENCH_COLLECTIVE=1: initialise RCCL and run the bounding-box all-gather even with one rank (what a one-GPU
    # box can exercise of the real multi-GPU branch: communicator set-up and a device-tensor collective on hardware).
    rehearse = os.environ.get("PCPX_BENCH_REHEARSE") == "1"
    collective = os.environ.get("PCPX_BENCH_COLLECTIVE") == "1"
    torch.cuda.set_device(0 if rehearse else local_rank)
    if world > 1 or collective:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    side = not args.no_extra
    main_res = run_workload(pkg, torch, dist, args.workload, rank, world, args.steps, args.warmup, want_profile=True,
                            rehearse=rehearse, side=side, collective=collective, at_curve_positions=args.rows_at_curve_positions)
    n, k = main_res["n"], main_res["k"]

    extra = {"index_build_ms_first": round(main_res["first_build_ms"], 3), "shard_of_rank0": list(main_res["shard"]),
             "first_step_ms": round(main_res["first_step_ms"], 4), "row_pitch_entries": main_res["row_pitch"] or k, "shard_cut": main_res["cut"]}
    if main_res["rebuild_ms"] is not None:
        reb = main_res["rebuild_ms"]
        extra["index_rebuild_ms"] = round(reb, 3)
        extra["value_incl_build"] = round(n / ((main_res["ms_per_step"] + (0.0 if args.workload in STREAMING else reb)) / 1e3) / 1e6, 3)
        # the build's algorithmic traffic is 30 B/point (12 read + 12 sorted xyz + 4 perm + node boxes);
        # the build IS bandwidth-shaped work, so this is the fraction that means something for it
        extra["build_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_point": 30, "achieved_GBps": round(30 * n / reb / 1e6, 1),
                                   "frac": round(30 * n / (reb * 1e-3) / HBM_PEAK, 5)}
        # replicated build inside the step (configs[4]): Amdahl ceiling of the 8-rank speed-up from this rank's numbers
        q_ms = main_res["ms_per_step"] - (reb if args.workload in STREAMING else 0.0)
        extra["amdahl_ceiling_8_ranks_replicated_build_in_step"] = round((q_ms + reb) / (q_ms / 8.0 + reb), 2)  # (rounds 1-3; see one_eighth_shard)
    if main_res["min_count"] is not None:
        extra["min_neighbours_found"] = main_res["min_count"]
    if main_res["local_index"]:
        extra["rank_local_index_of_rank0"] = main_res["local_index"]
    extra["rows_with_k_neighbours_all_ranks"] = main_res["complete"]  # = points when the shards cover the cloud exactly once
    if rehearse:
        extra["rehearsal"] = "one GPU, gloo: functional check of the N > 1 path, not a measurement"
    if collective:
        extra["collective"] = "bounding-box all-gather over %s with %d rank(s)" % ("gloo" if rehearse else "RCCL (backend nccl)", world)
    if main_res.get("range_ms"):
        extra["config3_range_count_r0.01_ms"] = round(main_res["range_ms"], 3)
        extra["config3_range_count_r0.01_mqps"] = round(n / main_res["range_ms"] / 1e3, 1)
    if main_res.get("range_pos_ms"):
        extra["config3_range_count_r0.01_curve_order_ms"] = round(main_res["range_pos_ms"], 3)
    if main_res.get("range_lists_ms"):
        extra["config3_range_lists_r0.01"] = {"ms": round(main_res["range_lists_ms"], 3), "indices": main_res["range_lists_total"],
                                              "what": "pcpx_range_lists_self_dev: counts + 64-bit scan + fill of every point's list, device resident",
                                              "Mlists_per_s": round(n / main_res["range_lists_ms"] / 1e3, 1),
                                              "output_GBps": round(4 * main_res["range_lists_total"] / main_res["range_lists_ms"] / 1e6, 1)}
    if main_res.get("rows_pos_ms"):
        extra["step_rows_at_curve_positions_ms"] = round(main_res["rows_pos_ms"], 4)
        extra["step_rows_at_curve_positions_mqps"] = round(n / main_res["rows_pos_ms"] / 1e3, 1)

    roofline = None
    prof = main_res["profile"]
    if prof and prof["knn"][0] > 0:
        launches, total_ms = prof["knn"]
        avg_s = total_ms / launches / 1e3
        q_per_launch = main_res["shard"][1]
        # kNN + normals with the indices also written = 12 B read + 4k B index
        # write + 12 B normal write per query (84 B at k = 15); the fused k_knn launch does exactly that
        bytes_per_q = 12 + 4 * k + (0 if args.workload in STREAMING else 12)
        achieved = q_per_launch * bytes_per_q / avg_s
        roofline = {"bound": "hbm", "kernel": "k_knn", "achieved": round(achieved / 1e9, 3), "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 6), "traffic": None,
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches": launches,
                    "algorithmic_bytes_per_query": bytes_per_q, "queries_per_launch": q_per_launch,
                    "note": "`bound`, `achieved`, `peak`, `frac` are the contract's HBM figures: algorithmic bytes / HIP-event time of the launch on its "
                            "stream, measured in this run.  The kernel is a tree search and is NOT bounded by HBM: what binds it is instruction issue, "
                            "and how full that is is in `issue` (DESIGN.md section 5).  traffic = PMC HBM bytes per launch from the committed rocprofv3 "
                            "--pmc passes of this command (profiles/%s_hbm_traffic.json), null when that file is not for this workload" % PROFILE_TAG}
        # What binds the kernel: the scalar issue slot of every SIMD (one scalar-class instruction per 4.1 cycles whatever the number of
        # resident waves, profiles/r03_valu_issue_rates.txt) and, close behind, the vector pipe.  Instruction counts per launch from the
        # committed counter passes of this command; the cycles are this run's launch time at the counter run's clock.
        issue_file = os.path.join(ROOT, "profiles", PROFILE_TAG + "_issue.json")
        if os.path.exists(issue_file) and world == 1:
            try:
                iss = json.load(open(issue_file))
                if iss.get("workload") == args.workload:
                    cyc = avg_s * iss["shader_clock_hz"]
                    simds = iss["simds"]
                    sc = iss["scalar_class_instructions_per_launch"] / simds * iss["cycles_per_scalar_instruction"] / cyc
                    vlo = iss["valu_instructions_per_launch"] / simds * iss["cycles_per_valu_instruction_range"][0] / cyc
                    vhi = iss["valu_instructions_per_launch"] / simds * iss["cycles_per_valu_instruction_range"][1] / cyc
                    roofline["issue"] = {"bound": "scalar issue slots (SALU + SMEM + branches + s_waitcnt / s_nop), one per SIMD per ~4.1 cycles",
                                         "scalar_issue_frac": round(sc, 3), "valu_issue_frac_range": [round(vlo, 3), round(vhi, 3)],
                                         "scalar_class_instructions_per_launch": iss["scalar_class_instructions_per_launch"],
                                         "valu_instructions_per_launch": iss["valu_instructions_per_launch"],
                                         "instructions_per_query": round(iss["instructions_per_launch"] / q_per_launch, 1),
                                         "source": "profiles/%s_issue.json (counters of the committed --pmc passes; launch time of THIS run; issue costs "
                                                   "measured by tools/valu_rate.hip)" % PROFILE_TAG}
            except Exception:
                pass
        traffic_file = os.path.join(ROOT, "profiles", PROFILE_TAG + "_hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("workload") == args.workload and world == 1:
                    roofline["traffic"] = tr.get("k_knn_hbm_bytes_per_launch")
                    roofline["traffic_source"] = "profiles/%s_hbm_traffic.json (separate --pmc passes, not this run)" % PROFILE_TAG
                    roofline["hbm_measured_GBps"] = round(roofline["traffic"] / avg_s / 1e9, 1)
                    roofline["hbm_measured_frac"] = round(roofline["traffic"] / avg_s / HBM_PEAK, 5)
            except Exception:
                pass
        nl, nms = prof["normals"]
        if nl:
            extra["k_normals_avg_launch_ms"] = round(nms / nl, 4)
        extra["k_knn_avg_launch_ms"] = round(avg_s * 1e3, 4)
        extra["knn_only_mqps"] = round(q_per_launch * world / avg_s / 1e6, 3)

    if rank == 0 and world == 1 and args.with_1m and args.workload != "uniform_1m_k15":
        side_res = run_workload(pkg, torch, dist, "uniform_1m_k15", 0, 1, max(args.steps, 10), args.warmup, want_profile=False, side=False)
        extra["configs1_uniform_1m_k15_mqps"] = round(side_res["mqps"], 3)
        extra["configs1_uniform_1m_k15_ms_per_step"] = round(side_res["ms_per_step"], 4)

    if rank == 0 and world == 1 and side and args.workload not in STREAMING:
        import subprocess
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-api-child", "--workload", args.workload],
                                   capture_output=True, text=True, timeout=600)
            extra.update(json.loads([l for l in child.stdout.splitlines() if l.startswith("{")][-1]))
        except Exception as e:  # a side measurement must not take the bench line down
            extra["host_api_error"] = repr(e)[:200]
        if args.workload == "uniform_10m_k15" and args.with_range_consumers:
            # the in-library consumers of sphere ranges on the same cloud (bilateral filter, WLOP), own process
            # for the same reason
            try:
                child = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "filter_bench.py"),
                                        str(n), "2"], capture_output=True, text=True, timeout=600)
                extra["range_consumers"] = json.loads([l for l in child.stdout.splitlines() if l.startswith("{")][-1])
            except Exception as e:
                extra["range_consumers_error"] = repr(e)[:200]

    if rank == 0 and world == 1 and side and not args.no_configs and args.workload == "uniform_10m_k15":
        import subprocess
        me = [sys.executable, os.path.abspath(__file__)]
        configs = {}

        def child(key, argv, timeout):
            t0 = time.perf_counter()
            try:
                c = subprocess.run(me + argv, capture_output=True, text=True, timeout=timeout)
                configs[key] = json.loads([l for l in c.stdout.splitlines() if l.startswith("{")][-1])
            except Exception as e:  # a side measurement must not take the bench line down
                configs[key] = {"error": repr(e)[:200]}
            configs[key]["child_wall_s"] = round(time.perf_counter() - t0, 1)

        child("uniform_1m_k15", ["--config-child", "uniform_1m_k15"], 120)            # configs[1]
        child("clustered_10m_k15", ["--config-child", "clustered_10m_k15"], 200)      # configs[3]'s cloud: one GPU + one rank's share of 8
        child("uniform_50m_k32_stream", ["--config-child", "uniform_50m_k32_stream"], 300)  # configs[4]: one GPU + one rank's share of 8
        child("single_operation_latency_2^20_points", ["--latency-child"], 300)
        child("end_to_end_host_abi", ["--end-to-end-child", "--workload", args.workload], 200)
        extra["configs"] = configs
        e2e = configs.get("end_to_end_host_abi", {})
        if "curve_order" in e2e:
            extra["end_to_end_mqps"] = e2e["curve_order"]["mqps"]

    if rank == 0 and world == 1 and side and args.workload != "uniform_1m_k15":
        # one rank's share of an 8-rank step of THIS workload (a projection; the driver's SCALE run is the measurement), in a
        # process of its own like the other side measurements (its short k_knn launches would blur this process's average)
        import subprocess
        try:
            c = subprocess.run([sys.executable, os.path.abspath(__file__), "--shard-child", "--workload", args.workload, "--base-ms",
                                repr(main_res["ms_per_step"]), "--base-rebuild-ms", repr(main_res["rebuild_ms"] or 0.0)],
                               capture_output=True, text=True, timeout=400)
            extra["one_eighth_shard"] = json.loads([l for l in c.stdout.splitlines() if l.startswith("{")][-1])
            if "amdahl_ceiling_8_ranks_build_in_step" in extra["one_eighth_shard"]:
                extra["amdahl_ceiling_8_ranks_build_in_step"] = extra["one_eighth_shard"]["amdahl_ceiling_8_ranks_build_in_step"]
        except Exception as e:
            extra["one_eighth_shard"] = {"error": repr(e)[:200]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(main_res["pts"], k)

    if world > 1:
        dist.barrier()
    if rank == 0:
        metric = ("kNN+normal-estimation Mqueries/s (k=%d)" % k) if args.workload not in STREAMING else \
                 ("index rebuild + kNN Mqueries/s (k=%d)" % k)
        line = {"metric": metric, "value": round(main_res["mqps"], 3),
                "unit": "Mqueries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(main_res["ms_per_step"], 4), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": args.workload, "points": n, "queries": n, "k": k,
                           "parallelism": ("replicated cloud, rank-local index, curve-sorted query shards x%d, bbox all-gather (RCCL)" if main_res["local_index"]
                                            else "replicated index, curve-sorted query shards x%d, bbox all-gather (RCCL)") % world},
                "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
