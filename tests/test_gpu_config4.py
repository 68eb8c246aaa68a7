"""BASELINE configs[3] (10 000 cameras x 2 000 000 landmarks x 10 000 000 observations) and its geometry.

With 10 000 cameras on the radius-6 ring of SURVEY.md section 8(d), neighbouring cameras are 3.8 mm apart and a
five-view track has a 15 mm baseline at 6 m: 1.25 px of parallax against 0.5 px of pixel noise.  The depth of a
landmark and the local scale of the camera chain are then barely observable, Levenberg-Marquardt converges
sub-linearly (landmarks whose best depth lies beyond infinity drift outwards for ever) and no CPU reference exists at
this size.  Parity is therefore pinned in two steps: the LM path of the HIP solver against the numpy restatement,
iteration by iteration, on a down-scaled scene with the same 3.8 mm geometry; and size-independent properties at full
size, at the reference's tolerances (bundle_adjustment.py:207-209) and at the library's defaults."""

import os
import subprocess
import sys

import numpy as np
import pytest

import scenes
from oracle import lm_schur_numpy as lm
from src import _native

pytestmark = pytest.mark.gpu


def test_config4_geometry_lockstep_with_oracle():
    """120 cameras on an arc with the spacing of the 10 000-camera ring, tracks of 5: K iterations of the HIP solver
    (exact banded Cholesky, as config 4 uses) and K of the numpy LM from the same guess must take the same accept /
    reject decisions, shorten the same landmark steps (the per-landmark cheirality guard fires on this geometry) and
    arrive at the same cost (1e-6 relative; fp64 sums in different orders), although the problem is far from
    converged; run to the end with the defaults, the HIP solver stops by tolerance below that."""
    K = 30
    sc = scenes.ring_scene(120, 6000, 5, spacing_of=10000)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    a = (sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    opt = lm.LMOptions(max_iters=K, ftol=0.0, xtol=0.0, gtol=0.0)
    _, _, o = lm.lm_solve(sc["poses_guess"], sc["points_guess"], *a, opt=opt, uniform_tracks=True)
    with _native.Solver(max_iters=K, ftol=0.0, xtol=0.0, gtol=0.0, linear_solver=3) as s:
        s.set_problem(C, L, *a)
        assert s.reduced_solver() == "banded_cholesky"
        _, _, rep = s.solve(sc["poses_guess"], sc["points_guess"])
    assert rep["iterations"] == K and rep["accepted"] == o.accepted
    assert rep["bad_pivots"] == 0
    assert rep["shortened_steps"] == o.shortened_steps and o.shortened_steps > 0
    assert rep["guard_rejections"] == o.guard_rejections
    np.testing.assert_allclose(rep["lambda_history"], o.lambdas, rtol=1e-6)
    assert abs(rep["cost"] - o.cost) <= 1e-6 * o.cost, (rep["cost"], o.cost)
    with _native.Solver() as s:
        s.set_problem(C, L, *a)
        P, X, full = s.solve(sc["poses_guess"], sc["points_guess"])
        _, cost_eval, _ = s.evaluate(P, X, want_residuals=False)
    assert full["status"] in (1, 2, 3) and full["cost"] < o.cost
    assert abs(cost_eval - full["cost"]) <= 1e-12 * full["cost"]


@pytest.fixture(scope="module")
def config4():
    sc = scenes.ring_scene(10000, 2000000, 5)
    return sc


@pytest.mark.parametrize("tolerances", ["reference", "defaults"])
def test_config4_full_size_properties(config4, tolerances):
    """status in {gtol, ftol, xtol} within the reference's cap of 2000 evaluations (max_nfev, bundle_adjustment.py:210);
    the reported cost is 0.5 |r|^2 of ba_evaluate at the estimate; every accepted step lowered the cost; nothing sits
    behind a camera; the RMS reprojection error is at the noise level (0.5 px per axis: 0.707 px minus what 6 M
    parameters absorb).  The estimate is NOT a fixed point in the strict sense: convergence is sub-linear on this
    geometry (module docstring), the ftol test fires at the first accepted step that gains less than ftol * cost, and a
    fresh solve from the estimate still finds 1.5e-3 of the cost (measured); asserted: it never raises the cost and
    finds less than 1e-2 of it."""
    sc = config4
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    opts = dict(ftol=1e-6, xtol=1e-6, gtol=1e-6) if tolerances == "reference" else {}
    with _native.Solver(**opts) as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        assert s.reduced_solver() == "cyclic_reduction"
        P, X, rep = s.solve(sc["poses_guess"], sc["points_guess"])
        _, cost_eval, nb = s.evaluate(P, X, want_residuals=False)
        _, _, rep2 = s.solve(P, X) if tolerances == "reference" else (None, None, None)
    assert rep["status"] in (1, 2, 3), rep["status_name"]
    assert rep["iterations"] <= 2000 and rep["bad_pivots"] == 0
    assert abs(cost_eval - rep["cost"]) <= 1e-12 * rep["cost"]
    assert nb == 0
    h = rep["cost_history"]
    # (the linearisation re-sums the cost of an accepted point in another order: 1e-12 relative of slack)
    assert all(b <= a * (1 + 1e-12) for a, b in zip(h, h[1:]))
    assert 0.55 < rep["rms_reproj"] < 0.62
    assert np.allclose(np.linalg.norm(P[:, 3:7], axis=1), 1.0, atol=1e-12)
    if rep2 is not None:
        assert rep2["cost"] <= rep["cost"] * (1 + 1e-12) and rep["cost"] - rep2["cost"] <= 1e-2 * rep["cost"]
    print(f"config 4 at {tolerances} tolerances: {rep['iterations']} iterations ({rep['accepted']} accepted, "
          f"{rep['guard_rejections']} guard rejections, {rep['shortened_steps']} shortened landmark steps), {rep['solve_ms']:.0f} ms, cost {rep['cost']:.6f}, status {rep['status_name']}")


def test_config4_is_reproducible_across_processes():
    """The full-size solve at the reference's tolerances, in two fresh processes: the same iteration count, the same
    accept / reject sequence and a bit-identical cost (round 2: 680 to 1900 iterations and 2.4e-5 of the cost between
    two runs, from the arrival order of fp64 atomics; round 3 sums U / g_c and the statistics in a fixed order)."""
    worker = os.path.join(os.path.dirname(__file__), "_determinism_worker.py")
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, worker, "10000", "2000000", "5", "1e-6"], capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1], outs
    it, status, cost_hex, digest = outs[0].split()
    assert int(it) > 50 and int(status) in (1, 2, 3)
    print("config 4, two processes:", outs[0])
