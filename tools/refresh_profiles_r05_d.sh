#!/bin/bash
# Round-5 evidence at the round's last kernel hash, call 2 of 2: the other workloads' bench lines, the accuracy survey, the two-rank
# rehearsal, then the fuzz campaign for whatever is left of the call.
O=gpurun_out/r05/final; mkdir -p $O
for wl in config3b config3c config3l config2 config4 config5; do
  python bench.py --workload $wl --no-cpu-baseline --no-secondary > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
done
python bench.py --workload config3 --solve auto --no-cpu-baseline --no-secondary > $O/bench_config3_auto.json 2> /dev/null || exit 1
python bench.py --workload config5 --link-geometry --no-cpu-baseline > $O/bench_config5_link_geometry.json 2> /dev/null || exit 1
timeout -k 10 400 python tools/accuracy_survey.py 2048 $O/accuracy_survey > $O/accuracy_survey.txt 2> $O/accuracy_survey.err || { tail -5 $O/accuracy_survey.err; exit 1; }
tail -16 $O/accuracy_survey.txt | cut -c1-330
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --rank-timeout 250 --no-cpu-baseline --no-secondary > $O/rehearsal_2ranks_config4.json 2> $O/rehearsal_2ranks_config4.err || { tail -5 $O/rehearsal_2ranks_config4.err; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/final/bench_*.json")) + sorted(glob.glob("gpurun_out/r05/final/rehearsal_2ranks_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f.split("/")[-1][:-5].ljust(34), f"{j['ms_per_step']*1e3:8.2f} us  {j['value']/1e6:8.1f} M/s  {str(j['config'].get('solve')):5s} {r['bound']:4s} frac {r['frac']:.3f}",
          "exec", None if r.get("executed_frac") is None else round(r["executed_frac"], 3),
          {k: round(v["ms_per_step"]*1e3, 2) for k, v in j.items() if k.startswith("solve_")}, j.get("world1_same_workload_ms"))
PY
M=${FUZZ_MINUTES:-6}
timeout -k 10 $((M * 60 + 80)) python tools/fuzz_parity.py --seeds 3000000 3060000 --minutes $M --log $O/fuzz_parity.log > $O/fuzz_parity.json 2>&1; tail -42 $O/fuzz_parity.json | cut -c1-200
