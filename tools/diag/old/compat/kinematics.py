"""Drop-in shim: `import kinematics` resolves to the MI355X engine's class surface
(same names as the reference's top-level kinematics.py).  Put this directory first on sys.path."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from riemannian_motion_policies_amd.kinematics import *  # noqa: F401,F403,E402
from riemannian_motion_policies_amd import kinematics as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
