"""multi_agent_pkgs_amd — MI355X-native batched solver for the per-agent trajectory optimisation
(the "HDSM inner loop") of lis-epfl/multi_agent_pkgs' multi_agent_planner.

The product is the C-ABI library declared in include/hdsm.h (HIP kernels for gfx950 + host C++).
This package holds its sources (csrc/), the build recipe, and a thin ctypes binding used by the tests,
the benchmark and the multi-GPU harness. There is no CPU fallback: importing :mod:`.lib` without the built
``libhdsm.so`` raises.
"""
from .params import HdsmParams, RefConfig, make_params, agile_params, default_params, agile_ref_config  # noqa: F401
