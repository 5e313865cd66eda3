#!/usr/bin/env python3
"""Build tuning variants of libpcpx (extra -D flags) and time them back to back on the bench workload.
usage: tools/ab_variants.py build|run  name:flag,flag  name2:flag ...   (flags without the leading -)"""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
variants = []
for a in sys.argv[2:]:
    name, _, flags = a.partition(":")
    variants.append((name, ["-" + f for f in flags.split(",") if f]))
b = importlib.import_module("point-cloud-processing_amd.build")
if mode == "build":
    for name, flags in variants:
        print(name, b.build(extra_flags=flags, tag=name), flush=True)
else:
    steps = os.environ.get("AB_STEPS", "5")
    for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
        for name, flags in variants:
            env = dict(os.environ, PCPX_LIB=os.path.join(ROOT, "point-cloud-processing_amd", "libpcpx_%s.so" % name))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1",
                                  "--no-cpu-baseline", "--no-extra"] + os.environ.get("AB_ARGS", "").split(),
                                 env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                print("%-14s round %d: %8.2f Mq/s  %7.3f ms/step  knn %s ms" % (name, rnd, d["value"], d["ms_per_step"], d["extra"].get("k_knn_avg_launch_ms")), flush=True)
            except Exception as e:
                print(name, "FAILED", out.stderr[-500:], flush=True)
