"""Driver entry points: build() compiles every native component, smoke() runs one tiny
control step on cuda:0 and checks it against the CPU oracle."""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "riemannian_motion_policies_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "librmp2_hip.so")
HIP_SOURCES = ["rmp2_hip.hip"]
HIP_HEADERS = ["rmp2_device.h", "rmp2_solve.h", "rmp2_quad.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"]


def _src_hash(deps) -> str:
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_hip(force: bool = False) -> str:
    """(Re)build librmp2_hip.so when its sources changed.  Staleness is decided by a content hash
    stored next to the library (file mtimes do not survive the copy to the GPU box)."""
    deps = [os.path.join(CSRC, f) for f in HIP_SOURCES + HIP_HEADERS] + [os.path.join(ROOT, "include", "rmp2.h")]
    stamp = LIB + ".srchash"
    want = _src_hash(deps)
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or not os.path.exists(LIB) or have != want:
        hipcc = os.environ.get("HIPCC", "hipcc")
        cmd = [hipcc, *HIPCC_FLAGS, "-o", LIB] + [os.path.join(CSRC, f) for f in HIP_SOURCES]
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:
            f.write(want)
    return LIB


def build() -> None:
    """Compile the HIP engine for gfx950 (hipcc cross-compiles without a GPU), the CPU oracle
    (test infrastructure; building the checker is not using it) and import the package."""
    build_hip()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    # The reference is pure Python on TensorFlow/PyBullet: there is nothing to compile into oracle/_ref.
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import riemannian_motion_policies_amd  # noqa: F401
    from riemannian_motion_policies_amd import _native
    import ctypes
    lib = ctypes.CDLL(LIB)  # symbol/ABI check only -- no compute without a GPU
    for sym in ("rmp2_abi_version", "rmp2_sizeof_desc", "rmp2_sizeof_obstacles", "rmp2_create", "rmp2_destroy",
                "rmp2_last_error", "rmp2_step", "rmp2_forward_kinematics", "rmp2_differentiate"):
        getattr(lib, sym)
    _native.lib()


def smoke() -> None:
    """One small control step (Panda cluttered set, 256 robots, 32 shared spheres) on cuda:0,
    checked against the CPU oracle."""
    import numpy as np
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    build_hip()
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    import oracle as O

    g = np.load(os.path.join(ROOT, "tests", "golden", "config3.npz"))
    _, desc = Cf.config3()
    eng = Engine(desc, 0)
    reps = 4
    q, qd, goal = (np.tile(g[k], (reps, 1)) for k in ("q", "qd", "goal"))
    out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(goal),
                   obstacles=eng.obstacles(spheres=torch.from_numpy(g["spheres"])))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = O.step(desc, q, qd, goal, spheres=g["spheres"])["qdd64"]
    err = np.abs(got - want).max(axis=1)
    tol = 1e-5 * np.maximum(1.0, np.abs(want).max(axis=1))
    assert np.all(np.isfinite(got)) and np.all(err <= tol), f"smoke mismatch: max err {err.max():.3e}"
    print(f"smoke ok: R={len(q)} max|qdd_hip - qdd_oracle| = {err.max():.3e}")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
