#!/bin/bash
# Build-path check on the GPU box: sort + parity tests, rebuild timings (10 M / 50 M / clustered), per-kernel stats.
#   gpurun --timeout 900 -- 'bash tools/rebuild_check.sh <tag> [tests|notests]'
set -u
tag=${1:-rebuild}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > "$out/tests.log" 2>&1
  rc=$?
  echo "pytest rc=$rc"; tail -4 "$out/tests.log"
  if [ $rc -ne 0 ]; then exit $rc; fi
fi
timeout -k 10 200 python3 tools/rebuild_loop.py 1e7 20 > "$out/rebuild_10m.json" 2> "$out/err.log" && cat "$out/rebuild_10m.json" &&
timeout -k 10 200 python3 tools/rebuild_loop.py 1e7 20 clustered > "$out/rebuild_10m_clustered.json" 2>> "$out/err.log" && cat "$out/rebuild_10m_clustered.json" &&
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 10 > "$out/rebuild_50m.json" 2>> "$out/err.log" && cat "$out/rebuild_50m.json" &&
timeout -k 10 200 python3 tools/rebuild_loop.py 1e7 20 uniform coarse > "$out/rebuild_10m_coarse.json" 2>> "$out/err.log" && cat "$out/rebuild_10m_coarse.json" &&
timeout -k 10 200 python3 tools/rebuild_loop.py 1e7 20 clustered coarse > "$out/rebuild_10m_clustered_coarse.json" 2>> "$out/err.log" && cat "$out/rebuild_10m_clustered_coarse.json" &&
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 10 uniform coarse > "$out/rebuild_50m_coarse.json" 2>> "$out/err.log" && cat "$out/rebuild_50m_coarse.json" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace10" -- python3 tools/rebuild_loop.py 1e7 10 > "$out/prof10.json" 2>> "$out/err.log" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace50" -- python3 tools/rebuild_loop.py 5e7 5 > "$out/prof50.json" 2>> "$out/err.log" || { echo "failed"; tail -5 "$out/err.log"; exit 1; }
for t in trace10 trace50; do
  f=$(find "$out/$t" -name '*kernel_stats.csv' | head -1)
  echo "== $t"; python3 tools/rocprof_summary.py "$f" | tee "$out/${t}_kernel_stats.txt"
done
# variants built beforehand (libpcpx_<tag>.so): rebuild time with each
for lib in point-cloud-processing_amd/libpcpx_*.so; do
  [ -f "$lib" ] || continue
  for sz in 1e7 5e7; do
    echo "variant $lib $sz: $(PCPX_LIB=$PWD/$lib timeout -k 10 200 python3 tools/rebuild_loop.py $sz 10 2>&1 | tail -1)"
  done
done
