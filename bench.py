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
with RCCL (24 B per rank, the only collective), every rank builds the same index on the union box and
answers a contiguous 64-aligned shard of the Morton-sorted queries.  Total work is fixed => "strong".
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
    bounded sample of the same cloud's queries.  kind = "port": the real reference cannot be built here."""
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
    return {"value": rate / 1e6, "unit": "Mqueries/s", "cores": threads, "kind": "port",
            "sample": "%d of %d queries (kNN k=%d + PCA normal each) on the full %d-point cloud, oracle octree "
                      "(capacity 32, depth 21), %d threads; octree build %.1f s not included" % (sample, n, k, n, threads, build_s)}


def _reduce(torch, dist, value, op, dev, rehearse):
    """All-reduce one number over the ranks (RCCL on the device; CPU tensor over gloo in the one-GPU rehearsal)."""
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if rehearse else dev)
    dist.all_reduce(t, op=op)
    return float(t.item())


def run_workload(pkg, torch, dist, name, rank, world, steps, warmup, want_profile, rehearse=False):
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
    d_box = mg.global_grid(d_box.cpu() if rehearse else d_box, dist, world)  # the one collective: 24 B per rank over RCCL
    grid = d_box.cpu().numpy()

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=dev.index, stream=stream, voxel_grid=grid)
    torch.cuda.synchronize()
    first_build_ms = (time.perf_counter() - t0) * 1e3
    assert ix.size() == n
    first, count = mg.query_shard(n, rank, world)

    d_idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)  # rows outside this rank's shard stay 0
    d_nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)

    streaming = name in STREAMING
    if streaming:
        # configs[4]: the cloud moves between iterations (jitter U(-1e-3, 1e-3), seed 45 + it); two jittered
        # copies alternate so every step rebuilds on coordinates that differ from the previous step's
        variants = [d_pts] + [torch.from_numpy(np.clip(pkg.synthetic.jitter(pts, seed + it), grid[:3], grid[3:])).to(dev)
                              for it in (1, 2)]
        it_no = [0]

        def step():
            it_no[0] += 1
            ix.rebuild_dev(variants[it_no[0] % len(variants)].data_ptr(), n, voxel_grid=grid)
            ix.knn_self_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, first, count)
    else:
        def step():
            ix.normals_knn_self_dev(k, 1e-5, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(), first, count)

    for _ in range(warmup):
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
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reb = 3
    for _ in range(reb):
        ix.rebuild_dev(d_pts.data_ptr(), n, voxel_grid=grid)
    torch.cuda.synchronize()
    rebuild_ms = (time.perf_counter() - t0) * 1e3 / reb

    # config 3 shape on the same cloud: radius count r = 0.01 around every point (device resident)
    range_ms = None
    if name == "uniform_10m_k15" and world == 1:
        d_rc = torch.empty(n, dtype=torch.int32, device=dev)
        ix.range_count_self_dev(0.01, d_rc.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.range_count_self_dev(0.01, d_rc.data_ptr())
        torch.cuda.synchronize()
        range_ms = (time.perf_counter() - t0) * 1e3 / 3

    res = {"n": n, "k": k, "elapsed": elapsed, "range_ms": range_ms, "steps": steps, "ms_per_step": elapsed * 1e3 / steps,
           "mqps": n * steps / elapsed / 1e6, "first_build_ms": first_build_ms, "rebuild_ms": rebuild_ms,
           "profile": prof, "shard": (first, count), "pts": pts, "complete": complete,
           "min_count": int(d_cnt.min().item()) if world == 1 else None}
    ix.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="uniform_10m_k15", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (rebuild, range count)")
    ap.add_argument("--with-1m", action="store_true",
                    help="also time configs[1] (1 M points, k=15) and report it under extra; off by default so that the "
                         "default command launches k_knn on the headline workload only (its rocprofv3 average then "
                         "equals roofline.avg_launch_ms)")
    args = ap.parse_args()

    import torch  # before libpcpx so that both share one HIP runtime
    import torch.distributed as dist
    pkg = importlib.import_module("point-cloud-processing_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the pcpx hot path has no CPU fallback")
    # PCPX_BENCH_REHEARSE=1: run the N > 1 code path on ONE GPU (all ranks on device 0, collectives over gloo) --
    # a functional rehearsal of sharding and reduction for tests, not a measurement
    rehearse = os.environ.get("PCPX_BENCH_REHEARSE") == "1"
    torch.cuda.set_device(0 if rehearse else local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    main_res = run_workload(pkg, torch, dist, args.workload, rank, world, args.steps, args.warmup, want_profile=True,
                            rehearse=rehearse)
    n, k = main_res["n"], main_res["k"]

    extra = {"index_build_ms_first": round(main_res["first_build_ms"], 3),
             "index_rebuild_ms": round(main_res["rebuild_ms"], 3),
             "value_incl_build": round(n / ((main_res["ms_per_step"] + main_res["rebuild_ms"]) / 1e3) / 1e6, 3),
             "shard_of_rank0": list(main_res["shard"])}
    if main_res["min_count"] is not None:
        extra["min_neighbours_found"] = main_res["min_count"]
    extra["rows_with_k_neighbours_all_ranks"] = main_res["complete"]  # = points when the shards cover the cloud exactly once
    if rehearse:
        extra["rehearsal"] = "one GPU, gloo: functional check of the N > 1 path, not a measurement"
    if main_res.get("range_ms"):
        extra["config3_range_count_r0.01_ms"] = round(main_res["range_ms"], 3)
        extra["config3_range_count_r0.01_mqps"] = round(n / main_res["range_ms"] / 1e3, 1)

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
                    "note": "fused kNN+normals kernel; it is VALU-issue bound, not HBM bound (DESIGN.md 'Roofline'); traffic = "
                            "PMC HBM bytes per launch of an earlier profiled run (profiles/r01_hbm_traffic.json)"}
        traffic_file = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("workload") == args.workload and world == 1:
                    roofline["traffic"] = tr.get("k_knn_hbm_bytes_per_launch")
                    # report the measured HBM rate beside the algorithmic one
                    roofline["hbm_measured_GBps"] = round(roofline["traffic"] / avg_s / 1e9, 1)
                    roofline["hbm_measured_frac"] = round(roofline["traffic"] / avg_s / HBM_PEAK, 5)
            except Exception:
                pass
        model_file = os.path.join(ROOT, "profiles", "r01_valu_issue_model.json")
        if os.path.exists(model_file) and world == 1:
            try:
                m = json.load(open(model_file))
                if m.get("workload") == args.workload:  # the bound that matters: share of SIMD cycles spent issuing VALU
                    roofline["valu_issue_frac_model"] = round(m["valu_issue_cycles_per_group_total"] * (q_per_launch / 64) /
                                                              (avg_s * 2.4e9 * 1024), 3)
            except Exception:
                pass
        nl, nms = prof["normals"]
        if nl:
            extra["k_normals_avg_launch_ms"] = round(nms / nl, 4)
        extra["k_knn_avg_launch_ms"] = round(avg_s * 1e3, 4)
        extra["knn_only_mqps"] = round(q_per_launch * world / avg_s / 1e6, 3)

    if rank == 0 and world == 1 and args.with_1m and args.workload != "uniform_1m_k15":
        side = run_workload(pkg, torch, dist, "uniform_1m_k15", 0, 1, max(args.steps, 10), args.warmup, want_profile=False)
        extra["configs1_uniform_1m_k15_mqps"] = round(side["mqps"], 3)
        extra["configs1_uniform_1m_k15_ms_per_step"] = round(side["ms_per_step"], 4)

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
                           "parallelism": "replicated index, Morton-sorted query shards x%d, bbox all-gather (RCCL)" % world},
                "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
