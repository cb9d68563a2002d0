"""Measure the config-5 cost model (ns of kernel time per robot and per (control point, obstacle) pair, per robot type)
that fleet.MixedFleetShard.plan cuts the mixed fleet with.  Prints one JSON object; the tracked copy is
profiles/r03_cost_calibration.json and the constants in fleet.MixedFleetShard.DEFAULT_COST.
usage: calibrate_costs.py [robots ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd.fleet import MixedFleetShard
sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768, 65536]
out = {"unit": "ns of kernel time", "by_fleet_size": {}}
for R in sizes:
    c = MixedFleetShard.calibrate_costs(0, robots=R)
    out["by_fleet_size"][str(R)] = {k: {"per_robot": v[0], "per_pair": v[1],
                                       "robot_with_mean_list_ns": v[0] + v[1] * 16 * MixedFleetShard.CONTROL_POINTS[k]} for k, v in c.items()}
print(json.dumps(out, indent=1))
