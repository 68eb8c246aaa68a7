"""The oracle's pose prior and window marginalisation (oracle/lm_schur_numpy.py), checked
on the CPU against finite differences, a brute-force minimisation and the stationarity that marginalisation must keep.
The reference has no counterpart (its keyframe list is never optimised): parity for this row is against the oracle."""

import numpy as np

import scenes
from oracle import lm_schur_numpy as lm


def _random_prior(rng, poses, cams, scale=50.0):
    n = len(cams)
    A = rng.normal(size=(6 * n + 3, 6 * n))
    ref = lm.retract(poses[cams], 0.05 * rng.normal(size=(n, 6)), np.ones(n, bool))
    return lm.PosePrior(cams=np.asarray(cams), ref=ref, H=scale * A.T @ A, g=scale * rng.normal(size=6 * n), c0=3.0)


def _energy(poses, pr):
    d, _ = lm.prior_chart(poses[pr.cams], pr.ref)
    d = d.reshape(-1)
    return pr.c0 + float(d @ (pr.g + 0.5 * pr.H @ d))


def test_prior_gradient_and_model_against_finite_differences():
    rng = np.random.default_rng(5)
    sc = scenes.ring_scene(6, 60, 3)
    poses = sc["poses_guess"]
    pr = _random_prior(rng, poses, [1, 3, 4])
    e, g, H = lm.prior_terms(poses, pr)
    assert abs(e - _energy(poses, pr)) <= 1e-12 * abs(e)
    free = np.ones(6, bool)
    eps = 1e-6
    gn = np.zeros(18)
    for k in range(18):
        dc = np.zeros((6, 6))
        dc[pr.cams[k // 6], k % 6] = eps
        gn[k] = (_energy(lm.retract(poses, dc, free), pr) - _energy(lm.retract(poses, -dc, free), pr)) / (2 * eps)
    np.testing.assert_allclose(g, gn, rtol=1e-6, atol=1e-6 * np.abs(g).max())
    # the Gauss-Newton model: exact gradient; the Hessian drops the chart's second derivative (O(|a|) relative)
    step = 1e-3 * rng.normal(size=(3, 6))
    dc = np.zeros((6, 6))
    dc[pr.cams] = step
    model = e + g @ step.reshape(-1) + 0.5 * step.reshape(-1) @ H @ step.reshape(-1)
    assert abs(_energy(lm.retract(poses, dc, free), pr) - model) <= 1e-3 * abs(model - e)


def test_marginalize_is_the_minimum_over_the_dropped_entries():
    rng = np.random.default_rng(2)
    A = rng.normal(size=(14, 9))
    S, b, c = A.T @ A, rng.normal(size=9), 1.5
    S2, b2, c2 = lm.marginalize(S, b, c, np.array([6, 7, 8]))
    x = rng.normal(size=6)
    y = -np.linalg.solve(S[6:, 6:], b[6:] + S[6:, :6] @ x)  # the minimiser over the dropped entries
    z = np.r_[x, y]
    assert abs((c + b @ z + 0.5 * z @ S @ z) - (c2 + b2 @ x + 0.5 * x @ S2 @ x)) <= 1e-12


def test_window_prior_keeps_the_window_stationary():
    """Solve a window to its minimum, marginalise its two oldest keyframes, and solve what is left (the other
    keyframes, the map points the dropped ones did not see) WITH the prior: nothing may move, and the minimum of
    the whole must be the prior's constant plus what the remaining observations cost."""
    sc = scenes.ring_scene(8, 160, 4)
    a = (sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    tight = lm.LMOptions(ftol=1e-15, gtol=1e-9, max_iters=60)
    P, X, rep = lm.lm_solve(sc["poses_guess"], sc["points_guess"], *a, opt=tight)
    free = np.ones(8, bool)
    free[0] = False
    prior, rows = lm.window_prior(P, X, *a, free, np.array([0, 1]))
    assert prior is not None and len(rows) > 0 and not set(prior.cams) & {0, 1}
    assert np.linalg.eigvalsh(prior.H).min() > -1e-8 * np.linalg.eigvalsh(prior.H).max()
    # the rest of the window: keyframes 2..7 renumbered 0..5, the map points that keyframes 0 and 1 do not see
    dead = np.zeros(len(X), bool)
    dead[rows] = True
    keep = ~dead[sc["obs_pt"]]
    assert not np.any(sc["obs_cam"][keep] < 2)
    remap = np.cumsum(~dead) - 1
    pr2 = lm.PosePrior(cams=prior.cams - 2, ref=prior.ref, H=prior.H, g=prior.g, c0=prior.c0)
    P2, X2, rep2 = lm.lm_solve(P[2:], X[~dead], sc["obs_cam"][keep] - 2, remap[sc["obs_pt"][keep]], sc["obs_uv"][keep],
                               sc["intr"], free=np.ones(6, bool), opt=tight, prior=pr2)
    assert abs(rep2.cost0 - rep.cost) <= 1e-9 * rep.cost  # c0 + remaining observations == the window's minimum
    assert rep2.cost <= rep2.cost0 and rep2.cost0 - rep2.cost <= 1e-9 * rep.cost
    assert np.abs(P2 - P[2:]).max() <= 1e-6 and np.abs(X2 - X[~dead]).max() <= 1e-5
