"""quadruped-gym_amd -- MI355X-native batched quadruped simulator.

Drop-in for the one hot path of antopio26/quadruped-gym, ``QuadrupedEnv.step()``
(``src/envs/quadruped.py:153-182``): a hand-written HIP rigid-body pipeline for
the repo's fixed 12-DoF quadruped behind a C ABI (``include/quadgym.h``), with
the reference's Gymnasium-style ``QuadrupedEnv`` API on top.

The directory name carries a hyphen (it mirrors the reference's name), so it is
imported through the ``quadruped_gym_amd`` shim package at the repository root.
"""
__version__ = "0.1.0"
