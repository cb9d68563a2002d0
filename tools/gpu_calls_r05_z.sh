#!/bin/bash
# the remaining bench lines of the round at the final kernels: config 5 with link geometry (runs on the pinv default since the 2-dof lift),
# the single-GPU emulation of the 8-rank jobs (labelled predictions, not measurements)
O=gpurun_out/r05/final; mkdir -p $O
python bench.py --workload config5 --link-geometry --no-cpu-baseline > $O/bench_config5_link_geometry.json 2> $O/bench_config5_link_geometry.err || { tail -3 $O/bench_config5_link_geometry.err; exit 1; }
python bench.py --workload config4 --emulate-world 8 --no-cpu-baseline > $O/emulated_scaling_config4.json 2> $O/emulated_scaling_config4.err || { tail -3 $O/emulated_scaling_config4.err; exit 1; }
python bench.py --workload config5 --emulate-world 8 --no-cpu-baseline > $O/emulated_scaling_config5.json 2> $O/emulated_scaling_config5.err || { tail -3 $O/emulated_scaling_config5.err; exit 1; }
python - <<'PY'
import json
for n in ("bench_config5_link_geometry", "emulated_scaling_config4", "emulated_scaling_config5"):
    j = json.loads(open(f"gpurun_out/r05/final/{n}.json").read().strip().splitlines()[-1])
    print(n, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("value", "ms_per_step", "n_gpus", "predicted_value", "emulated_world")}, str(j.get("emulation", ""))[:300])
PY
