"""The CPU oracle evaluated in float64: oracle.py run a second time against libgs_oracle_f64.so (gs_oracle.c built with
-DGSO_REAL_DOUBLE: same formulas, same float-valued constants, every float a double).  Same functions, float64 arrays.

TEST INFRASTRUCTURE ONLY.  It is the arbiter of the randomised parity test: where the fp32 oracle is itself far from
this evaluation (an ill-conditioned scene), the HIP result is held to the stated tolerance against THIS one.
"""
import os

_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle.py")
_ns = {"__name__": __name__, "__file__": _src, "GSO_DOUBLE": True}
with open(_src) as _f:
    exec(compile(_f.read(), _src, "exec"), _ns)
globals().update({k: v for k, v in _ns.items() if not k.startswith("__")})
