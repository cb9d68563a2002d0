#!/bin/bash
# interface B, two-wave form: the odd slot of every SIMD starts its first wave late (RMP2_STREAM_STAGGER=n, units of ~3.4 us)
O=gpurun_out/r05; mkdir -p $O
{ echo "# config3b, 65 536 robots, default two-wave form: us per step against the start offset of the odd wave slot (first round only)"
for s in 0 2 4 6 8 10 12 16; do
  RMP2_STREAM_STAGGER=$s python bench.py --workload config3b --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $s'.ljust(14), '%8.2f us' % (j['ms_per_step']*1e3), ' rejected', j['result_check']['rejected'], ' [' + str(j['config'].get('kernel', ''))[:70] + ']')"
done
for s in 0 6; do
  RMP2_STREAM_STAGGER=$s python bench.py --workload config3b --robots 131072 --steps 50 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('131072 robots, stagger $s'.ljust(28), '%8.2f us' % (j['ms_per_step']*1e3))"
done; } > $O/interface_b_stagger_two_wave.txt 2>&1
cat $O/interface_b_stagger_two_wave.txt | cut -c1-200
# the two-rank rehearsal in the driver's launcher shape (hung in the round's first evidence run: the world-1 leg of a job that had
# fallen back to the torch-driven exchange gathered over the job's default group)
F=$O/final; mkdir -p $F
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $F/rehearsal_2ranks_config4_torchrun.json 2> $F/rehearsal_2ranks_config4_torchrun.err; echo "torchrun rehearsal rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --rank-timeout 250 --no-cpu-baseline --no-secondary > $F/rehearsal_2ranks_config4.json 2> $F/rehearsal_2ranks_config4.err; echo "launcher rehearsal rc=$?"
wc -l $F/rehearsal_2ranks_config4_torchrun.json $F/rehearsal_2ranks_config4.json
python - <<'PY'
import json
for n in ("config4_torchrun", "config4"):
    try:
        j = json.loads(open(f"gpurun_out/r05/final/rehearsal_2ranks_{n}.json").read())
        print(n, {k: j.get(k) for k in ("value", "ms_per_step", "exchange", "rccl_nranks", "world1_same_workload_ms")}, j["world1_same_workload"]["what"][:90])
    except Exception as e:
        print(n, "no line", repr(e)[:200])
PY
tail -3 $F/rehearsal_2ranks_config4_torchrun.err | cut -c1-300
