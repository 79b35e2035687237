"""Dev experiment (CPU): primal-dual active-set iterations over a closed-loop C3 rollout for different first guesses at every step:
(a) cold: violated rows of v_unc; warm: the previous face shifted by one stage (what the kernels use)
(b) saturated time-varying LQR roll-forward at every step
(c) cold: the roll; warm: the shifted face."""
import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import synth
from cold_start_guess import pdas  # noqa
