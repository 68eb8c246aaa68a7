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


def test_config4_geometry_arc_against_the_reference(golden):
    """The same 120-camera arc pinned against THE REFERENCE (round 3; tests/golden/arc_120x6000.npz, oracle/make_golden.py:
    the reference's residual_function driven by the reference's least_squares settings, bundle_adjustment.py:201-219,
    218 evaluations / 200 s, cost 4928.5457; and the numpy LM with direct solves run on from there to ftol 1e-12,
    725 iterations, cost 4922.0099 -- scipy's own tight-tolerance variants do not finish on this valley: dense exact
    trust region: 9 GB Jacobian; LSMR at 1e-12: stopped after 2 h 20 min).
    What holds on a valley this flat, and is asserted:
      * the generator rebuilds the fixture's scene bit for bit; the residuals at the guess are the reference's (1e-9 px);
        the cost of the REFERENCE'S OWN RESULT, evaluated on the GPU, is the reference's cost (1e-12): same objective;
      * started at the reference's result, the solver reaches the polished minimum from there, two-sided 1e-6
        (measured 4e-9);
      * started at the guess with the defaults it ends BELOW the reference's cost (4926.44 < 4928.55), and with exact
        linear solves and ftol 1e-12 below the polished value too (4919.76 < 4922.01: the valley goes on);
      * at the reference's own tolerances it stops where an LM step first gains less than 1e-6 of the cost: 4931.42,
        5.8e-4 ABOVE the reference's 4928.55 (scipy's trust region takes 218 evaluations to get there, this LM stops after
        113 iterations): asserted within 1e-3, not one-sided.  The minimiser on this geometry is path-dependent at the
        1e-3 level for every solver involved; the objective is pinned to 1e-12."""
    g = golden("arc_120x6000")
    sc = scenes.ring_scene(120, 6000, 5, spacing_of=10000)
    assert np.array_equal(g["obs_cam"], sc["obs_cam"]) and np.array_equal(g["obs_pt"], sc["obs_pt"])
    for k in ("obs_uv", "poses_guess", "points_guess"):
        np.testing.assert_array_equal(g[k], sc[k])
    C, L = 120, 6000
    a = (sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    ref, pol = float(g["ref_cost"]), float(g["polished_cost"])
    assert int(g["ref_status"]) == 2 and pol < ref
    ref_poses = np.vstack([sc["poses_guess"][0][None], g["ref_x"][: 7 * (C - 1)].reshape(-1, 7)])
    ref_pts = g["ref_x"][7 * (C - 1):].reshape(-1, 3)
    with _native.Solver() as s:
        s.set_problem(C, L, *a)
        r0, _, _ = s.evaluate(sc["poses_guess"], sc["points_guess"])
        np.testing.assert_allclose(r0, g["residual_at_x0"], rtol=0, atol=1e-9)
        _, cost_at_ref, _ = s.evaluate(ref_poses, ref_pts, want_residuals=False)
        assert abs(cost_at_ref - ref) <= 1e-12 * ref, (cost_at_ref, ref)
        _, _, dflt = s.solve(sc["poses_guess"], sc["points_guess"])
    assert dflt["status"] in (1, 2, 3) and dflt["cost"] <= ref * (1 + 1e-6), (dflt["cost"], ref)
    with _native.Solver(ftol=1e-12, max_iters=4000) as s:
        s.set_problem(C, L, *a)
        _, _, on = s.solve(ref_poses, ref_pts)
    assert on["status"] in (1, 2, 3) and abs(on["cost"] - pol) <= 1e-6 * pol, (on["cost"], pol)
    with _native.Solver(ftol=1e-12, max_iters=4000, linear_solver=3) as s:
        s.set_problem(C, L, *a)
        _, _, exact = s.solve(sc["poses_guess"], sc["points_guess"])
    assert exact["status"] in (1, 2, 3) and exact["bad_pivots"] == 0 and exact["cost"] <= pol * (1 + 1e-6), (exact["cost"], pol)
    with _native.Solver(ftol=1e-6, xtol=1e-6, gtol=1e-6) as s:
        s.set_problem(C, L, *a)
        _, _, at_ref_tol = s.solve(sc["poses_guess"], sc["points_guess"])
    assert at_ref_tol["status"] in (1, 2, 3) and abs(at_ref_tol["cost"] - ref) <= 1e-3 * ref, (at_ref_tol["cost"], ref)
    print(f"arc 120 x 6000: reference {ref:.4f}, polished {pol:.4f}; GPU defaults {dflt['cost']:.4f} ({dflt['iterations']} it), "
          f"from the reference's result {on['cost']:.6f}, exact solves {exact['cost']:.4f}, reference tolerances {at_ref_tol['cost']:.4f}")


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
