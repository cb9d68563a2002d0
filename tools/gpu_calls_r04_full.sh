#!/bin/bash
# the whole GPU suite, then the bench lines of the round
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite.log; tail -8 $O/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
for wl in config3b config3c config3l config2; do
  python bench.py --workload $wl --no-cpu-baseline --no-secondary > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1
done
python bench.py --workload config3 --solve pinv --no-cpu-baseline --no-secondary > $O/bench_config3_pinv.json 2> $O/bench_config3_pinv.err || exit 1
python bench.py --workload config2 --solve pinv --no-cpu-baseline --no-secondary > $O/bench_config2_pinv.json 2> $O/bench_config2_pinv.err || exit 1
python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2> $O/bench_config4.err || exit 1
python bench.py --workload config5 --no-cpu-baseline --no-secondary > $O/bench_config5.json 2> $O/bench_config5.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_*.json')):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable', e); continue
    rc=j.get('result_check',{})
    print(f.split('/')[-1], round(j['ms_per_step']*1e3,2),'us/step', '| frac', round(j['roofline']['frac'],3), j['roofline']['bound'], '|', j['roofline']['kernel'][:60], '|', rc.get('admitted_by', {k:v.get('admitted_by') for k,v in rc.items() if isinstance(v,dict)}))
PY
