#!/bin/bash
O=gpurun_out/r05/final; mkdir -p $O
python tools/calibrate_costs.py pinv > $O/cost_calibration.json 2> $O/cost_calibration.err || { tail -3 $O/cost_calibration.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05/final/cost_calibration.json"))
for k, v in j["curves"].items(): print(k, v["us"])
PY
