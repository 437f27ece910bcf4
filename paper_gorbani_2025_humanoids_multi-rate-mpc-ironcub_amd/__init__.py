"""MI355X-native batched multi-rate MPC solve path (drop-in for the reference's
VariableSamplingMPC update()+solveMPC(), momentum-based-linear-mpc-lib/.../variableSamplingMPC.cpp:88-112).

The compute lives in the HIP library behind include/vsmpc.h (csrc/); this package holds the thin
host side: record layouts, the ctypes binding, the reference-shaped wrapper class and the
synthetic workload generator.  Importing the package does not load the HIP library; creating a
solver does, and fails loudly if it is missing.
"""
from . import layout  # noqa: F401
from .layout import MPCConfig, paper_config, horizon2x_config  # noqa: F401
