"""Parity tolerances of the dispersion path and where each comes from.

Every IEEE operation of the search is replayed exactly (tests/test_hostsim.py: bit-identical to the
oracle when the device cores run with glibc's sin/cos/exp).  What differs on the GPU is the last
bit of some sin/cos/exp values.  Models whose velocity increases with depth do not notice (0
values differ in 2.75 M, profiles/r01_parity_campaign.txt).  With low-velocity zones the Rayleigh
period equation is ill-conditioned near its root; the last Neville/bisection iterate is then
decided by rounding noise and may land anywhere inside the reference's own stopping bracket

    |c1 - c2| <= 1e-6 * c1                                   (surfdisp96.f:614)

Bounds derived from that criterion:

  phase velocity  c  : |dc| / c <= 1e-6, plus half an fp32 ulp of the stored value (6e-8)
  group velocity  U  : U = (1/ta - 1/tb) / (1/(ta c0) - 1/(tb c1)), ta,tb = t/(1 +- h), h = 0.005
                       (surfdisp96.f:232-235,306); dc0, dc1 <= 1e-6 c  =>  |dU| / U <= 1e-6 U / (c h)
                       = 2e-4 U/c, with U <= c for normal dispersion; the fp32 evaluation of the
                       cancelling denominator adds ~1e-5.

(Two runs each stop somewhere inside their own final bracket around the same root, so in the worst case
two results are the sum of two brackets apart: 2e-6 c in phase velocity and 4e-4 (U/c) U in group
velocity -- relative to U that is unbounded where the finite-difference denominator nearly vanishes, i.e.
at the anomalous points of low-velocity-zone dispersion curves where U >> c.  Random campaigns over
10.8 M searches incl. higher modes, earth flattening, water layers and irregular periods --
tests/scenarios/kernel_fuzz.py, profiles/r02_kernel_fuzz.txt -- saw at most 1.99e-6 in phase velocity,
and 2.1e-4 .. 1.5e-2 in group velocity depending on whether a seed hits such a point (3-layer LVZ model,
60 periods); the test sets below stay at 1.05e-6 and 2.1e-4.  All kernel forms agree with each other bit
for bit everywhere.)

Evidence that this is the reference's own reproducibility limit, not an implementation difference
(profiles/r02_libm_selfdiff.txt, tests/scenarios/libm_selfdiff.py): the reference's native code
run against ITSELF with glibc's non-FMA sin/cos/exp builds instead of the FMA ones differs on
low-velocity-zone models by up to 1.02e-6 relative in phase velocity and 4.3e-4 (1.4e-4 relative)
in group velocity, 99.69-99.99 % of the values identical; Love and monotone models: identical.
The device against the reference (profiles/parity_r01.txt): 1.01e-6, 5.7e-4 (2.0e-4 relative),
99.2-99.9 % identical.

Round 3 closed the question for the worst values of those campaigns (tests/scenarios/lvz_worst_cases.py,
tests/golden/lvz_worst_cases.npz, profiles/r03_lvz_worst_cases.txt): the models behind the 1.51e-2, 4.5e-3
(group) and 1.99e-6, 1.88e-6 (phase) lines were re-derived from the campaign seeds and handed to the
reference's own binary under a libm whose sin/cos/exp are moved by <= 1 ulp (an LD_PRELOAD shim; 24 noise
seeds).  At exactly those models and periods the REFERENCE then returns the device's value, bit for bit,
in some of the runs and its usual value in the others: 610.66 vs 619.89 km/s (U/c = 257), -234.96 vs
-236.02 (U/c = -68, second mode), 2.51754999 vs 2.51754498.  The deviation class is therefore asserted,
not just printed, against the bound that includes the conditioning term:

  phase  |dc|/c <= 2e-6 (two stopping brackets) + fp32 rounding of the stored value
  group  |dU|/|U| <= 4e-4 * |U/c|: with U = (2h/t) / ((1+h)/(t c0) - (1-h)/(t c1)), a relative change e0, e1
         of c0, c1 changes U by (|e0| + |e1|) |U| / (2 h c); e <= 2e-6 each, h = 0.005 (surfdisp96.f:232-306)
"""
import numpy as np

TOL_PHASE_2BRACKETS = 2.2e-6    # two runs, each anywhere in its own 1e-6 bracket, + fp32 output rounding


def group_bound(U, c):
    """Largest relative difference of two group velocities computed from phase velocities that each sit
    anywhere inside the search's own stopping bracket (surfdisp96.f:614), as a function of U/c; never
    below the single-bracket figure asserted on ordinary models."""
    U, c = np.asarray(U, dtype=np.float64), np.asarray(c, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = np.where(c != 0, np.abs(U / c), np.inf)
    return np.maximum(TOL_GROUP_REL, 4.2e-4 * ratio)


TOL_PHASE_REL = 1.2e-6          # derived: 1e-6 (stopping bracket) + fp32 rounding of the output
TOL_GROUP_REL = 2.5e-4          # derived: 2e-4 * U/c + fp32 cancellation
TOL_PHASE = TOL_PHASE_REL * 5.0  # absolute forms for c, U <= 5 km/s (vs prior 2..5 km/s)
TOL_GROUP = TOL_GROUP_REL * 5.0
MIN_IDENTICAL_LVZ = 0.99        # large sets (>= 1000 models); the reference itself: >= 0.9969
MIN_IDENTICAL_LVZ_SMALL = 0.97  # 24-model golden sets: a single unlucky model moves 4 %
MIN_IDENTICAL_MONOTONE = 1.0    # velocity increasing with depth: bit-identical
TOL_RF = 1.0e-10                # receiver function, absolute at amplitudes <= ~8 (observed 4.4e-12)


def rf_bound(ref_error):
    """Bound on |rf - oracle| / scale for a model on which the accuracy of the reference itself is known:
    `ref_error` is the distance of the fp64 oracle from the same algorithm in extended precision (tests/hp_oracle.py,
    tests/rf_extreme.py: oracle_error).  About one model in 3e5 of the random campaigns -- a near-singular layer
    stack -- amplifies rounding 1e5 to 1e8 times: the reference's own double-precision trace is then good to 2.5e-9
    ... 1e-7 only, and any other fp64 evaluation of the same formulas (the device: 0.8 - 1.25e-10; the CPU replay of
    the device program with glibc math: up to 1.8e-10) may differ from it by a fraction of that.  Two evaluations
    each within e of the exact result differ by at most 2e; half the reference's own error is asserted.  Everywhere
    else TOL_RF holds (typical models: oracle 1e-13 from the exact trace, device 1e-15 from the oracle)."""
    return max(TOL_RF, 0.5 * ref_error)


TOL_MISFIT = 1.0e-6             # north_star: RMS misfit on the tutorial dataset
