"""Generate tests/golden/kinematic_tables.json from the REFERENCE's own URDF parser.

Runs only in the build container (needs /root/reference).  It imports the reference's
stdlib-only helper/urdf_parsing.py (the one hot-path module that imports without
TensorFlow/PyBullet), parses the reference's two URDFs and records
what kinematics.py:163-209 derives from the tree: frame order, backward paths, per-frame
rpy / xyz / axis / joint type / has_collision and the q re-ordering for the PyBullet motor
joint order.  The JSON is data (inputs + expected outputs), not reference source.

    python tests/golden/make_kinematic_tables.py
"""
import importlib.util
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kinematic_tables.json")

spec = importlib.util.spec_from_file_location("ref_urdf_parsing", os.path.join(REF, "helper", "urdf_parsing.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

ROBOTS = {
    "panda": (os.path.join(REF, "urdf", "franka_panda", "panda.urdf"),
              [f"panda_joint{i}" for i in range(1, 8)] + ["panda_finger_joint1", "panda_finger_joint2"]),
    "two_joint": (os.path.join(REF, "urdf", "TwoJointRobot_wo_fixedJoints.urdf"), ["joint_1", "joint_2"]),
}

out = {}
for key, (path, order) in ROBOTS.items():
    tree = ref.UrdfTree(path)
    paths = tree.get_backward_paths()
    names = [p[-1] for p in paths]           # kinematics.py:170-171
    elems = [tree.get_element_by_name(name=n) for n in names]
    out[key] = {
        "order": order,
        "frame_names": names,
        "backward_paths": paths,
        "q_reordering": [order.index(n) if n in order else len(order) for n in names],  # kinematics.py:197
        "rpy": [e.rpy for e in elems],
        "xyz": [e.xyz for e in elems],
        "axis": [e.axis for e in elems],
        "joint_type": [e.joint_type for e in elems],
        "has_collision": [bool(e.has_collision) for e in elems],
        "link_names": [e.link_name for e in elems],
    }

with open(OUT, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", OUT, {k: len(v["frame_names"]) for k, v in out.items()})
