#!/bin/bash
# The N > 1 control flow of bench.py on a ONE-GPU box: two rank processes on device 0 over torch's gloo backend
# (bench.py --rehearse-one-gpu).  (1) bench.py's own launcher (spawn_ranks + supervise), config 4: the native exchange's
# communicator fails on both ranks (RCCL refuses two ranks on one device), the ranks agree and all take the torch-driven
# exchange; (2) the same for config 5 (no data-path collective); (3) the driver's launcher shape (torch.distributed.run).
# Outputs under gpurun_out/r04/; each line is labelled a rehearsal, none is a scaling measurement.
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 400 python bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --rank-timeout 300 --no-secondary \
    > $O/rehearsal_2ranks_config4.json 2> $O/rehearsal_2ranks_config4.err &&
timeout -k 10 400 python bench.py --gpus 2 --rehearse-one-gpu --workload config5 --steps 200 --warmup 20 --rank-timeout 300 \
    > $O/rehearsal_2ranks_config5.json 2> $O/rehearsal_2ranks_config5.err &&
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --no-secondary \
    > $O/rehearsal_2ranks_config4_torchrun.json 2> $O/rehearsal_2ranks_config4_torchrun.err
rc=$?
for f in $O/rehearsal_2ranks_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: l.get(k) for k in ("value", "ms_per_step", "n_gpus", "exchange", "exchange_fell_back", "rccl_nranks")}, l.get("rehearsal", {}).get("ranks"),
          {k: (v if not isinstance(v, dict) else v.get("rejected")) for k, v in (l.get("result_check") or {}).items() if k != "tolerance"})
except Exception as e:
    print("no line:", e)
PY
done
tail -5 $O/rehearsal_2ranks_config4.err
exit $rc
