#!/bin/bash
# Round-5 evidence at the round's LAST kernel hash (the dispatch commit a2c73e5 changed rmp2_hip.hip after the previous run, so the
# hash-guarded counter files no longer matched the library bench.py loads), call 1 of 2: the GPU suite, smoke(), rocprofv3 kernel stats,
# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes), SQ counters + stamps -> executed_config3.json, then the contract line,
# which then carries `traffic` and `executed` measured at the hash it runs on.  Most important first: a call is at most 20 minutes.
# The program after "--" is python3 itself; --pmc is never combined with tracing.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/final; mkdir -p $O
C=${COMMIT:-unknown}
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite.log; tail -4 $O/gpu_suite.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
python __graft_entry__.py smoke > $O/smoke.txt 2>&1 || { tail -3 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline > /dev/null 2>&1
cp $(ls -t $O/kt3/*/*kernel_stats.csv | head -1) $O/config3_R65536_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3 $O/w3 config3 65536 $O/traffic_config3.json $C
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq2 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
(python tools/pmc_sq.py $O/sq1; python tools/pmc_sq.py $O/sq2) > $O/sq_counters_config3_R65536.txt 2>&1
(RMP2_KERNEL=quad python tools/stamps.py 65536; RMP2_KERNEL=hex python tools/stamps.py 4096) > $O/stamps.txt 2>/dev/null
python tools/executed.py $O/sq1 $O/sq2 $O/stamps.txt config3 65536 $O/executed_config3.json $C
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f2 $O/w2 config2 4096 $O/traffic_config2.json $C
cp $O/traffic_config2.json $O/traffic_config3.json $O/executed_config3.json profiles/
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style_steps20.json 2> /dev/null || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
cp $(ls -t $O/kt3b/*/*kernel_stats.csv | head -1) $O/config3b_R65536_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3b $O/w3b config3b 65536 $O/traffic_config3b.json $C
rm -rf $O/kt*/*/*.db $O/f3 $O/w3 $O/f3b $O/w3b $O/f2 $O/w2 $O/sq1 $O/sq2
for f in $O/*kernel_stats.csv; do echo $f; head -3 $f | cut -c1-100,180-330; done
cat $O/traffic_config3.json $O/traffic_config3b.json $O/executed_config3.json | cut -c1-200
python - <<'PY'
import json
for f in ("bench_default", "bench_driver_style_steps20"):
    j = json.loads(open(f"gpurun_out/r05/final/{f}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print(f, f"{j['ms_per_step']*1e3:.2f} us {j['value']/1e9:.3f} G/s frac {r['frac']:.3f} traffic {r['traffic']} executed_frac {r.get('executed_frac')}",
          j["result_check"]["admitted_by"], "stale" , (r.get("executed") or {}).get("stale"))
PY
