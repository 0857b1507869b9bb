"""Shared cases of the line-search tests (CPU and GPU)."""
import numpy as np

from gtsam_ndt_amd import synth


def poor_inits():
    """Initial guesses up to 0.6 m / 0.12 rad off: far enough for steps that score worse."""
    T = np.array(synth.T_STAR)
    rng = np.random.default_rng(0)
    return [tuple(T + off) for off in rng.uniform([-0.6, -0.6, -0.12], [0.6, 0.6, 0.12], size=(16, 3))]


# (hessian mode, index into poor_inits()): trajectories with 21, 7, 1 and 13 rejected trials that
# are stable under float32/float64 evaluation differences (Newton mode from most other poor
# starts is chaotic at that level - DESIGN.md section 2.5)
LS_CASES = [(0, 6), (0, 12), (1, 3), (1, 12)]

# the device evaluates in float32: Newton trajectories from poor starts are only compared where
# they are stable at that level
LS_CASES_GPU = [(0, 6), (0, 12), (1, 3)]
