#!/bin/bash
# per-kernel times of the filter bench (GPU box): gpurun -- 'bash tools/filter_profile.sh'
out=gpurun_out/filt
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/filter_bench.py 1e7 1 > "$out/bench_under_prof.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
f=$(ls -t $out/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-100s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
