"""The throughput kNN kernels live at the edge of their register budgets (72 VGPRs = 7 waves per SIMD for k <= 16, 64 = 8 for k <= 8,
128 = 4 for k <= 32): one value kept across the search too many and hipcc moves it to scratch, which costs the kernel ~10 %.  This
compiles csrc/pcpx_query.hip for gfx950 (device side only, no GPU needed) and checks the code object's metadata: no single-pass
k_knn kernel may use scratch, and each must fit the registers its occupancy needs."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(600)
def test_single_pass_knn_kernels_use_no_scratch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "pcpx_query.hip", "k_knn<"],
                         capture_output=True, text=True, timeout=580, check=True).stdout
    rows = re.findall(r"k_knn<(\d+), (true|false), (\d), (true|false), (true|false), (\d)>\S*\s+vgpr\s+(\d+) sgpr\s+(\d+) sspill\s+(\d+) vspill\s+(\d+) scratch\s+(\d+)", out)
    assert len(rows) >= 18, out
    limit = {"8": 64, "16": 72, "32": 128}
    for kcap, _self, diag, multi, _eps_each, _nz, vgpr, _sgpr, sspill, _vspill, scratch in rows:
        if multi == "true" or diag == "1":
            continue  # (the k > 32 multi-pass and the diagnostic builds keep their own budgets)
        assert int(scratch) == 0, (kcap, diag, scratch, out)
        assert int(vgpr) <= limit[kcap], (kcap, vgpr)
        if diag == "0":
            assert int(sspill) <= 20, (kcap, sspill)  # (the event-counting build, DIAG = 2, keeps a few counters more)
