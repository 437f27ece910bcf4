"""Import-name shim: the reference's harness does `from momentum_based_mpc.bindingsMPC import VariableSamplingMPC`
(src/variable_sampling_mpc.py:5).  With this repository's root on sys.path that line resolves to the MI355X-backed module
(paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd/csrc/bindings_mpc.cpp) and the harness stays unchanged.
Nothing else of the reference's `momentum_based_mpc` package is provided."""
