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
