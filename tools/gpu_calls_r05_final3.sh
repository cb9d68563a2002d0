#!/bin/bash
# Round-5 evidence at the hash of 5a65afb, call 3: the four bench lines that quote a stored PMC traffic file, again -- call 2 was
# sent before call 1's traffic_*.json had been copied into profiles/, so its lines read the files of the previous hash and (rightly)
# did not quote them; the two-rank rehearsal likewise -- then a soak of the fuzz campaign on fresh seeds and the pair grid for what
# is left of the call.
O=gpurun_out/r05/final; mkdir -p $O
for wl in config3b config2 config4; do
  python bench.py --workload $wl --no-cpu-baseline --no-secondary > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
done
python bench.py --workload config3 --solve auto --no-cpu-baseline --no-secondary > $O/bench_config3_auto.json 2> /dev/null || exit 1
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --rank-timeout 250 --no-cpu-baseline --no-secondary > $O/rehearsal_2ranks_config4.json 2> $O/rehearsal_2ranks_config4.err || { tail -5 $O/rehearsal_2ranks_config4.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_config3b", "bench_config2", "bench_config4", "bench_config3_auto", "rehearsal_2ranks_config4"):
    j = json.loads(open(f"gpurun_out/r05/final/{f}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print(f.ljust(28), f"{j['ms_per_step']*1e3:8.2f} us  frac {r['frac']:.3f} traffic {r['traffic']}", (r.get("traffic_source") or {}).get("note"))
PY
M=${FUZZ_MINUTES:-8}
timeout -k 10 $((M * 60 + 90)) python tools/fuzz_parity.py --seeds 8000000 8200000 --minutes $M --log $O/fuzz_parity_soak.log > $O/fuzz_parity_soak.json 2>&1; tail -45 $O/fuzz_parity_soak.json | cut -c1-200
timeout -k 10 100 python tools/fuzz_parity.py --pairs --seeds 5000 5400 --minutes 1.2 > $O/fuzz_parity_pair_grid.json 2>&1; tail -12 $O/fuzz_parity_pair_grid.json | cut -c1-200
