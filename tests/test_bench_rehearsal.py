"""The N > 1 path of bench.py itself (rank plumbing, per-rank boxes -> all-gather -> identical index, Morton-sorted query
shards, MAX / SUM reductions) rehearsed with two processes on the one GPU of the test box: collectives over gloo, both
ranks on device 0 (PCPX_BENCH_REHEARSE=1).  Functional only -- the timing of such a run means nothing."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_bench_two_ranks_on_one_gpu(world):
    env = dict(os.environ, PCPX_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--workload", "uniform_1m_k15"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["workload"] == "uniform_1m_k15"
    # every point got its k neighbours exactly once across the ranks' shards
    assert d["extra"]["rows_with_k_neighbours_all_ranks"] == 1_000_000
    first, count = d["extra"]["shard_of_rank0"]
    assert first == 0 and count % 64 == 0 and 0 < count < 1_000_000


def test_bench_with_the_real_rccl_backend_on_one_rank():
    """What a one-GPU box can run of the REAL multi-GPU branch: bench.py launched by torch.distributed.run (before any GPU
    call), backend "nccl" (= RCCL) initialised on the device, the per-rank bounding box all-gathered as a DEVICE tensor
    through RCCL (PCPX_BENCH_COLLECTIVE=1 makes the collective run with a single rank too), device-side reductions."""
    env = dict(os.environ, PCPX_BENCH_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PCPX_BENCH_REHEARSE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra", "--workload", "uniform_1m_k15"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert "RCCL" in d["extra"]["collective"]
    assert d["extra"]["rows_with_k_neighbours_all_ranks"] == 1_000_000


def test_rccl_communicator_behind_the_c_abi(pkg):
    """pcpx_comm_*: the bounding-box all-gather behind the C ABI (librccl bound at first use), with one rank on hardware:
    unique id -> communicator -> box of a device slice, all-gather, union -> the grid; and an index built on that grid
    answers like the auto-bbox index."""
    import ctypes as C
    import importlib
    import numpy as np
    torch = pytest.importorskip("torch")
    capi = importlib.import_module("point-cloud-processing_amd._capi")
    lib = capi.load()
    uid = C.create_string_buffer(128)
    capi.check(lib.pcpx_comm_unique_id(uid))
    comm = C.c_void_p(None)
    capi.check(lib.pcpx_comm_init_rank(uid, 1, 0, 0, C.byref(comm)))
    pts = pkg.synthetic.clustered_cloud(300_000, 44)
    d_pts = torch.from_numpy(pts).cuda()
    torch.cuda.synchronize()
    grid = np.zeros(6, np.float32)
    capi.check(lib.pcpx_comm_global_grid_dev(comm, C.c_void_p(d_pts.data_ptr()), len(pts), None, grid.ctypes.data_as(capi.f32p)))
    assert np.array_equal(grid[:3], pts.min(0)) and np.array_equal(grid[3:], pts.max(0))
    # the raw collective: one box in, world boxes out
    d_all = torch.zeros(6, dtype=torch.float32, device="cuda")
    d_loc = torch.from_numpy(grid).cuda()
    capi.check(lib.pcpx_comm_allgather_boxes_dev(comm, C.c_void_p(d_loc.data_ptr()), C.c_void_p(d_all.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(d_all.cpu().numpy(), grid)
    # an empty slice contributes the empty box
    capi.check(lib.pcpx_comm_global_grid_dev(comm, None, 0, None, grid.ctypes.data_as(capi.f32p)))
    assert grid[0] > 1e38 and grid[3] < -1e38
    lib.pcpx_comm_destroy(comm)
    assert lib.pcpx_comm_init_rank(uid, 2, 5, 0, C.byref(comm)) == capi.PCPX_ERR_INVALID
