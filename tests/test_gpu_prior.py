"""Quadratic pose prior (ba_set_pose_prior) and the sliding-window driver with marginalisation (SURVEY.md section 8
row f2), through the C ABI, against the numpy oracle (oracle/lm_schur_numpy.py: PosePrior, window_prior).  The
reference has no counterpart of either, so the oracle is the whole of the parity evidence for this row (its own
mathematics is pinned on the CPU by tests/test_oracle_prior.py)."""

import numpy as np
import pytest

import scenes
from oracle import lm_schur_numpy as lm
from src import _native
from src.camera_pose import CameraPose
from src.local_ba import Keyframe, SlidingWindowBundleAdjustment, build_windows
from src.validation_error import ValidationError

pytestmark = pytest.mark.gpu

K_MAT = np.array([[500.0, 0, 320.0], [0, 500.0, 240.0], [0, 0, 1.0]])


def _random_prior(rng, poses, cams, scale=50.0):
    n = len(cams)
    A = rng.normal(size=(6 * n + 3, 6 * n))
    ref = lm.retract(poses[cams], 0.03 * rng.normal(size=(n, 6)), np.ones(n, bool))
    return lm.PosePrior(cams=np.asarray(cams), ref=ref, H=scale * A.T @ A, g=scale * rng.normal(size=6 * n), c0=7.0)


@pytest.mark.parametrize("solver", [0, 2])  # AUTO (one-workgroup PCG here) and the dense Cholesky
def test_prior_solve_matches_oracle(solver):
    """One solve with a random prior over three cameras (one pair of them shares no landmark): normal equations at the
    guess against the oracle's (1e-9), then the whole LM run: same accept / reject sequence, same lambdas, costs to
    1e-9 with the exact solver, the minimum to 1e-8 with the inexact one."""
    rng = np.random.default_rng(11)
    sc = scenes.ring_scene(12, 300, 4)
    a = (sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    P0, X0 = sc["poses_guess"], sc["points_guess"]
    pr = _random_prior(rng, P0, [2, 3, 8])
    tight = dict(ftol=1e-14, gtol=1e-9, max_iters=60)
    Po, Xo, o = lm.lm_solve(P0, X0, *a, opt=lm.LMOptions(**tight), prior=pr)
    with _native.Solver(linear_solver=solver, pcg_tol=1e-10 if solver == 0 else 3e-2, **tight) as s:
        s.set_pose_prior(pr.cams, pr.ref, pr.H, pr.g, pr.c0)
        s.set_problem(12, 300, *a)
        lin = s.debug_linearization(P0, X0, 1e-3)
        P, X, rep = s.solve(P0, X0)
        _, reproj, _ = s.evaluate(P, X, want_residuals=False)
    # the oracle's blocks at the guess, prior included
    free = np.ones(12, bool)
    free[0] = False
    r, Jc, Jp = lm.linearize(P0, X0, *a)
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, cost = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], 12, 300)
    e, gj, Hj = lm.prior_terms(P0, pr)
    idx = (6 * pr.cams[:, None] + np.arange(6)).reshape(-1)
    for k, c in enumerate(pr.cams):
        U[c] += Hj[6 * k:6 * k + 6, 6 * k:6 * k + 6]
        gc[c] += gj[6 * k:6 * k + 6]
    S, rhs, _ = lm.schur_system(U, V, W, gc, gp, sc["obs_cam"], sc["obs_pt"], 1e-3, free)
    Hoff = Hj.copy()
    for k in range(3):
        Hoff[6 * k:6 * k + 6, 6 * k:6 * k + 6] = 0.0
    S[np.ix_(idx, idx)] += Hoff
    assert np.abs(S[12:18, 48:54]).max() > 0 and np.abs(S[12:18, 48:54] - Hoff[0:6, 12:18]).max() == 0  # cameras 2, 8: prior only
    np.testing.assert_allclose(lin["S"], S, rtol=0, atol=1e-9 * np.abs(S).max())
    np.testing.assert_allclose(lin["rhs"], rhs, rtol=0, atol=1e-9 * np.abs(rhs).max())
    np.testing.assert_allclose(lin["U"], U, rtol=0, atol=1e-9 * np.abs(U).max())
    dc = np.linalg.solve(S, -rhs).reshape(12, 6)
    np.testing.assert_allclose(lin["dc"], dc, rtol=0, atol=(1e-7 if solver == 0 else 1e-9) * np.abs(dc).max())
    # the LM run
    assert abs(rep["cost0"] - o.cost0) <= 1e-12 * o.cost0
    assert rep["status"] in (1, 2, 3) and o.status in (1, 2, 3)
    assert abs(rep["cost"] - o.cost) <= 1e-8 * o.cost, (rep["cost"], o.cost)
    if solver == 2:
        assert rep["accepted"] == o.accepted
        n = min(len(o.lambdas), len(rep["lambda_history"]))
        np.testing.assert_allclose(rep["lambda_history"][:n], o.lambdas[:n], rtol=1e-6)
    # the reported cost is re-projection + prior energy at the estimate
    d, _ = lm.prior_chart(P[pr.cams], pr.ref)
    d = d.reshape(-1)
    assert abs(reproj + pr.c0 + d @ (pr.g + 0.5 * pr.H @ d) - rep["cost"]) <= 1e-10 * rep["cost"]
    assert np.abs(P - Po).max() <= 1e-5 and np.abs(X - Xo).max() <= 1e-4


def test_prior_argument_errors():
    sc = scenes.ring_scene(6, 60, 3)
    a = (sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    rng = np.random.default_rng(0)
    pr = _random_prior(rng, sc["poses_guess"], [0, 2])
    with _native.Solver() as s:
        s.set_pose_prior(pr.cams, pr.ref, pr.H, pr.g)
        with pytest.raises(ValidationError, match="fixed"):
            s.set_problem(6, 60, *a)  # camera 0 is the fixed one
        with pytest.raises(ValidationError, match="symmetric"):
            H = pr.H.copy()
            H[0, 1] += 1.0
            s.set_pose_prior(pr.cams, pr.ref, H, pr.g)
        with pytest.raises(ValidationError, match="twice"):
            s.set_pose_prior([2, 2], pr.ref, pr.H, pr.g)
        with pytest.raises(ValidationError, match="at most"):
            s.set_pose_prior(np.arange(17), np.tile(pr.ref[:1], (17, 1)), np.eye(102), np.zeros(102))
        s.set_pose_prior([1, 9], pr.ref, pr.H, pr.g)
        with pytest.raises(ValidationError, match="not a camera"):
            s.set_problem(6, 60, *a)
        with pytest.raises(RuntimeError, match="pose prior"):
            s.solve_batched([dict(obs_cam=sc["obs_cam"], obs_pt=sc["obs_pt"], obs_uv=sc["obs_uv"],
                                  poses=sc["poses_guess"], points=sc["points_guess"])], sc["intr"])
        s.set_pose_prior(None, None, None, None)  # removed: the plain problem again
        s.set_problem(6, 60, *a)
        _, _, rep = s.solve(sc["poses_guess"], sc["points_guess"])
        assert rep["status"] in (1, 2, 3)


def _keyframes(sc, poses):
    out = []
    for c in range(len(poses)):
        sel = sc["obs_cam"] == c
        out.append(Keyframe(float(c), CameraPose(position=poses[c, :3], orientation=poses[c, 3:7], timestamp=float(c)),
                            K_MAT, sc["obs_uv"][sel], sc["landmark_ids"][sc["obs_pt"][sel]]))
    return out


def test_sliding_window_with_marginalisation_matches_oracle():
    """14 keyframes, windows of 8, stride 2 (4 windows).  The same chain on the oracle: numpy LM with the prior, numpy
    marginalisation.  Every window's prior (H, g, c0: 1e-6 of their scale), frozen map rows, cost before and after
    (1e-7) and the final trajectory must agree; and the chain must leave the whole trajectory at a lower
    re-projection cost than it found it."""
    sc = scenes.ring_scene(14, 420, 5)
    kfs = _keyframes(sc, sc["poses_guess"])
    tight = dict(ftol=1e-13, gtol=1e-9, max_iters=80)
    swba = SlidingWindowBundleAdjustment(window=8, stride=2, min_views=2, **tight)
    poses, pts = swba.optimize(kfs, sc["points_guess"], sc["landmark_ids"], K_MAT)
    wins = swba.last_windows
    assert [w.first for w in wins] == [0, 2, 4, 6]
    assert swba.last_priors[0] is None and all(p is not None for p in swba.last_priors[1:])
    # the oracle's chain
    P_all, X_all = sc["poses_guess"].copy(), sc["points_guess"].copy()
    frozen = np.zeros(len(X_all), bool)
    prior = None  # (global keyframes, PosePrior over local indices is rebuilt per window)
    raw = build_windows(kfs, sc["landmark_ids"], 8, 2, 2)
    for b, w0 in enumerate(raw):
        live = ~frozen[w0.landmark_rows]
        sel = live[w0.obs_pt]
        remap = np.cumsum(live) - 1
        rows = w0.landmark_rows[live]
        oc, op, uv = w0.obs_cam[sel], remap[w0.obs_pt[sel]], w0.obs_uv[sel]
        np.testing.assert_array_equal(rows, wins[b].landmark_rows)
        free = w0.keyframes >= 2  # the first two keyframes of the list are the gauge (pose and monocular scale)
        pr = None
        if prior is not None:
            pr = lm.PosePrior(cams=np.searchsorted(w0.keyframes, prior[0]), ref=prior[1].ref, H=prior[1].H,
                              g=prior[1].g, c0=prior[1].c0)
            mine = swba.last_priors[b]
            np.testing.assert_array_equal(mine.keyframes, prior[0])
            np.testing.assert_allclose(mine.H, pr.H, rtol=0, atol=1e-6 * np.abs(pr.H).max())
            np.testing.assert_allclose(mine.g, pr.g, rtol=0, atol=1e-6 * max(np.abs(pr.g).max(), np.abs(pr.H).max() * 1e-3))
            assert abs(mine.c0 - pr.c0) <= 1e-6 * abs(pr.c0)
            np.testing.assert_allclose(mine.ref, pr.ref, rtol=0, atol=1e-6)
        P, X, o = lm.lm_solve(P_all[w0.keyframes], X_all[rows], oc, op, uv, sc["intr"], free=free,
                              opt=lm.LMOptions(**tight), prior=pr)
        rep = swba.last_reports[b]
        assert abs(rep["cost0"] - o.cost0) <= 1e-6 * o.cost0, (b, rep["cost0"], o.cost0)
        assert abs(rep["cost"] - o.cost) <= 1e-7 * o.cost, (b, rep["cost"], o.cost)
        assert rep["status"] in (1, 2, 3)
        P_all[w0.keyframes] = P
        X_all[rows] = X
        if b + 1 < len(raw):
            n_drop = int(np.searchsorted(w0.keyframes, raw[b + 1].first))
            npr, mrows = lm.window_prior(P, X, oc, op, uv, sc["intr"], free, np.arange(n_drop), pr)
            frozen[rows[mrows]] = True
            prior = (w0.keyframes[npr.cams], npr)
        np.testing.assert_array_equal(np.flatnonzero(frozen), swba.last_frozen[b])
    Pm = np.array([np.r_[p.position, p.quaternion] for p in poses])
    assert np.abs(Pm - P_all).max() <= 1e-5 and np.abs(pts - X_all).max() <= 1e-4
    np.testing.assert_array_equal(Pm[:2], sc["poses_guess"][:2])
    # the whole trajectory, re-projection cost over ALL observations: far below the guess's
    c0 = lm.normal_blocks(lm.linearize(sc["poses_guess"], sc["points_guess"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"],
                                       sc["intr"], want_jac=False)[0], np.zeros((len(sc["obs_cam"]), 2, 6)),
                          np.zeros((len(sc["obs_cam"]), 2, 3)), sc["obs_cam"], sc["obs_pt"], 14, 420)[5]
    c1 = 0.5 * float(np.sum(lm.linearize(Pm, pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"], want_jac=False)[0] ** 2))
    assert c1 < 0.01 * c0
    rms = np.sqrt(2 * c1 / len(sc["obs_cam"]))
    assert rms < 2.0  # pixel noise 0.5 px; two keyframes pinned at their noisy guesses, frozen map points and
    #                   linearised priors cost the rest
