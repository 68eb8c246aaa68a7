"""The oracle (oracle/ref_scipy_path.py) against the fixtures recorded from the reference itself
(oracle/make_golden.py).  CPU only."""

import numpy as np
import pytest

from oracle import lm_schur_numpy as lm
from oracle import ref_scipy_path as orc

SMALL = ["test_scene_convergence", "test_scene_landmark_accuracy", "demo_scene", "noisy_10x300"]


def _args(g):
    n_poses = len(g["poses_gt"]) - 1
    return n_poses, (g["pose0"], n_poses, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])


@pytest.mark.parametrize("name", SMALL)
def test_residual_matches_reference(golden, name):
    g = golden(name)
    n_poses, a = _args(g)
    r = orc.residual_packed(g["x0"], *a)
    # floating point: the reference multiplies (P - t) @ R through BLAS; tolerance 1e-9 px absolute
    np.testing.assert_allclose(r, g["residual_at_x0"], rtol=0, atol=1e-9)
    C = n_poses + 1
    ids = [g["landmark_ids"][g["obs_pt"][g["obs_cam"] == c]] for c in range(C)]
    uv = [g["obs_uv"][g["obs_cam"] == c] for c in range(C)]
    r2 = orc.residual_per_image(g["x0"], g["pose0"], n_poses, g["landmark_ids"], ids, uv, g["intr"])
    np.testing.assert_allclose(r2, g["residual_at_x0"], rtol=0, atol=1e-9)
    # and at the reference's solution
    np.testing.assert_allclose(orc.residual_packed(g["ref_x"], *a), g["ref_residual"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", SMALL)
def test_sparsity_mask_matches_reference(golden, name):
    g = golden(name)
    n_poses, _ = _args(g)
    m = orc.sparsity_mask(n_poses, len(g["landmark_ids"]), g["obs_cam"], g["obs_pt"])
    m.sort_indices()
    assert np.array_equal(m.indptr, g["mask_indptr"])
    assert np.array_equal(m.indices, g["mask_indices"])
    nobs, nobs0 = len(g["obs_cam"]), int(np.sum(g["obs_cam"] == 0))
    assert m.nnz == 2 * nobs * 10 - 2 * nobs0 * 7


@pytest.mark.parametrize("name", ["test_scene_convergence", "test_scene_landmark_accuracy", "noisy_10x300"])
def test_scipy_path_reproduces_reference_solution(golden, name):
    g = golden(name)
    n_poses, a = _args(g)
    res, _, _ = orc.solve_scipy_path(g["x0"], *a)
    assert res.status == int(g["ref_status"])
    assert res.nfev == int(g["ref_nfev"])
    if float(g["ref_cost"]) > 1e-6:
        # same scipy call, but 1e-13 rounding differences in the residual (BLAS matmul in the
        # reference vs explicit sums here) pass through finite differences and an inexact LSMR
        # step: agreement is 1e-8 relative, asserted at 1e-7 (north_star bar: 1e-6)
        assert abs(res.cost - float(g["ref_cost"])) <= 1e-7 * float(g["ref_cost"])
        np.testing.assert_allclose(res.x, g["ref_x"], rtol=0, atol=1e-5)
    else:  # noise-free scene: cost ~ 0, compare absolutely
        assert res.cost < 1e-9


def test_analytic_jacobian_vs_finite_differences(golden):
    g = golden("noisy_10x300")
    n_poses, a = _args(g)
    x = g["ref_x"]
    J = orc.jacobian_packed(x, *a).toarray()
    rng = np.random.default_rng(1)
    for col in rng.choice(len(x), 40, replace=False):
        h = 1e-6 * max(1.0, abs(x[col]))
        e = np.zeros_like(x)
        e[col] = h
        fd = (orc.residual_packed(x + e, *a) - orc.residual_packed(x - e, *a)) / (2 * h)
        scale = max(np.abs(fd).max(), 1e-12)
        assert np.abs(J[:, col] - fd).max() <= 1e-6 * scale, col


def test_local_jacobian_vs_finite_differences(golden):
    """2x6 pose / 2x3 point blocks of the LM oracle against central differences of the REFERENCE-path
    residual through the retraction (north_star: analytic vs FD to 1e-6 relative)."""
    g = golden("test_scene_convergence")
    n_poses, a = _args(g)
    C = n_poses + 1
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    r, Jc, Jp = lm.linearize(poses, pts, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    free = np.ones(C, bool)
    free[0] = False

    def res_at(poses_, pts_):
        x = np.concatenate([poses_[1:].reshape(-1), pts_.reshape(-1)])
        return orc.residual_packed(x, *a).reshape(-1, 2)

    h = 1e-6
    for cam in (1, 3):
        for k in range(6):
            d = np.zeros((C, 6))
            d[cam, k] = h
            fd = (res_at(lm.retract(poses, d, free), pts) - res_at(lm.retract(poses, -d, free), pts)) / (2 * h)
            sel = g["obs_cam"] == cam
            scale = np.abs(fd[sel]).max()
            assert np.abs(Jc[sel][:, :, k] - fd[sel]).max() <= 1e-6 * scale
    for lmk in (0, 17):
        for k in range(3):
            d = np.zeros_like(pts)
            d[lmk, k] = h
            fd = (res_at(poses, pts + d) - res_at(poses, pts - d)) / (2 * h)
            sel = g["obs_pt"] == lmk
            scale = np.abs(fd[sel]).max()
            assert np.abs(Jp[sel][:, :, k] - fd[sel]).max() <= 1e-6 * scale


@pytest.mark.parametrize("name", ["noisy_10x300"])
def test_lm_oracle_reaches_polished_minimum(golden, name):
    g = golden(name)
    n_poses, a = _args(g)
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    _, _, rep = lm.lm_solve(poses, pts, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    pol = float(g["polished_cost"])
    assert abs(rep.cost - pol) <= 1e-6 * pol
    assert rep.cost <= float(g["ref_cost"]) * (1 + 1e-6)
