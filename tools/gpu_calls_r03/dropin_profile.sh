#!/bin/bash
set -e
mkdir -p gpurun_out
python - > gpurun_out/dropin_profile.txt 2>&1 <<'PY'
import cProfile, pstats, sys, io, runpy
sys.argv = ["tools/dropin_latency.py", "7", "50"]
g = runpy.run_path("tools/dropin_latency.py")
for name in ("host_eval_only", "host_body", "device_body"):
    pr = cProfile.Profile()
    fn = g[name]
    pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("=====", name)
    print(s.getvalue()[:6000])
PY
grep -v amdgpu.ids gpurun_out/dropin_profile.txt | head -150
