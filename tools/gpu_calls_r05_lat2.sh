#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
RMP2_QUAD_LATENCY_BLOCKS=0 python tools/calibrate_costs.py pinv > $O/cost_calibration_nolatency.json 2>/dev/null
python - <<'PY'
import json
a = json.load(open("gpurun_out/r05/final/cost_calibration.json")) if __import__("os").path.exists("gpurun_out/r05/final/cost_calibration.json") else json.load(open("profiles/r05_cost_calibration.json"))
b = json.load(open("gpurun_out/r05/cost_calibration_nolatency.json"))
for k in ("two_joint", "panda"):
    print(k); print(" robots ", a["curves"][k]["robots"]); print(" default", a["curves"][k]["us"]); print(" no lat.", b["curves"][k]["us"])
PY
