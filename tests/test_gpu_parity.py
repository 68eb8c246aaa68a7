"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against the
oracle and the fixtures recorded from the reference (tests/golden)."""

import numpy as np
import pytest

import scenes
from oracle import lm_schur_numpy as lm
from oracle import ref_scipy_path as orc
from src import _native

pytestmark = pytest.mark.gpu


def _state(g):
    n_poses = len(g["poses_gt"]) - 1
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    return n_poses, poses, pts


def _solver(g, **opt):
    s = _native.Solver(**opt)
    s.set_problem(len(g["poses_gt"]), len(g["landmark_ids"]), g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    return s


SMALL = ["test_scene_convergence", "test_scene_landmark_accuracy", "demo_scene", "noisy_10x300"]


@pytest.mark.parametrize("name", SMALL)
def test_residual_kernel_matches_reference(golden, name):
    """ba_evaluate vs BundleAdjustment.residual_function of the reference at x0 and at its solution.
    fp64: 1e-9 px absolute (pixels are O(1e2-1e3), so ~1e-12 relative)."""
    g = golden(name)
    n_poses, poses, pts = _state(g)
    with _solver(g) as s:
        r, cost, nb = s.evaluate(poses, pts)
        np.testing.assert_allclose(r, g["residual_at_x0"], rtol=0, atol=1e-9)
        assert abs(cost - 0.5 * float(g["residual_at_x0"] @ g["residual_at_x0"])) <= 1e-12 * max(cost, 1.0)
        xr = g["ref_x"]
        poses_r = np.vstack([g["pose0"][None], xr[: 7 * n_poses].reshape(-1, 7)])
        r2, cost2, _ = s.evaluate(poses_r, xr[7 * n_poses:].reshape(-1, 3))
        np.testing.assert_allclose(r2, g["ref_residual"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", ["test_scene_convergence", "noisy_10x300"])
def test_jacobian_blocks_vs_finite_differences(golden, name):
    """Analytic 2x6 / 2x3 blocks vs central differences of the oracle's reference-path residual
    (north_star: 1e-6 relative), and vs the numpy blocks (1e-11 relative)."""
    g = golden(name)
    n_poses, poses, pts = _state(g)
    C = n_poses + 1
    a = (g["pose0"], n_poses, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    with _solver(g) as s:
        Jc, Jp = s.jacobian_blocks(poses, pts)
    _, Jc_o, Jp_o = lm.linearize(poses, pts, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    Jc_o[g["obs_cam"] == 0] = 0.0
    np.testing.assert_allclose(Jc, Jc_o, rtol=0, atol=1e-11 * np.abs(Jc_o).max())
    np.testing.assert_allclose(Jp, Jp_o, rtol=0, atol=1e-11 * np.abs(Jp_o).max())
    free = np.ones(C, bool)
    free[0] = False

    def res_at(poses_, pts_):
        return orc.residual_packed(np.concatenate([poses_[1:].reshape(-1), pts_.reshape(-1)]), *a).reshape(-1, 2)

    h = 1e-6
    for cam in (1, C - 1):
        sel = g["obs_cam"] == cam
        for k in range(6):
            d = np.zeros((C, 6))
            d[cam, k] = h
            fd = (res_at(lm.retract(poses, d, free), pts) - res_at(lm.retract(poses, -d, free), pts)) / (2 * h)
            assert np.abs(Jc[sel][:, :, k] - fd[sel]).max() <= 1e-6 * np.abs(fd[sel]).max()
    for lmk in (0, len(pts) - 1):
        sel = g["obs_pt"] == lmk
        for k in range(3):
            d = np.zeros_like(pts)
            d[lmk, k] = h
            fd = (res_at(poses, pts + d) - res_at(poses, pts - d)) / (2 * h)
            assert np.abs(Jp[sel][:, :, k] - fd[sel]).max() <= 1e-6 * np.abs(fd[sel]).max()


@pytest.mark.parametrize("lin", ["default", "flat", "window", "group", "matrix_free"])
@pytest.mark.parametrize("name", ["test_scene_convergence", "noisy_10x300"])
def test_normal_equation_blocks_match_oracle(golden, name, lin, monkeypatch):
    """U, V, W, gradients, Schur complement S, reduced rhs and the LM step against the numpy
    restatement; fp64 sums in different orders: 1e-10 relative to the largest entry.  With each of the
    linearisation kernels: the default (k_linearize_rows, round 3: fixed-order sums), round 2's kernels that stay in
    the tree behind BA_LIN_ROWS=0 (k_linearize_flat for few cameras, k_linearize_win for more than 64) and the
    group-per-track k_linearize<G> (tracks longer than 32); and on the matrix-free path."""
    if lin == "flat":
        monkeypatch.setenv("BA_LIN_ROWS", "0")
    elif lin == "window":
        monkeypatch.setenv("BA_LIN_ROWS", "0")
        monkeypatch.setenv("BA_LIN_WIN", "1")
    elif lin == "group":
        monkeypatch.setenv("BA_LIN_ROWS", "0")
        monkeypatch.setenv("BA_LIN_WIN", "0")
        monkeypatch.setenv("BA_LIN_FLAT", "0")
    elif lin == "matrix_free":  # no stored W: the Schur and back-substitution kernels rebuild it (the default from 2 M observations)
        monkeypatch.setenv("BA_MATRIX_FREE", "1")
    g = golden(name)
    n_poses, poses, pts = _state(g)
    C, L = n_poses + 1, len(pts)
    lam = 1e-3
    with _solver(g, pcg_tol=1e-13, pcg_max_iter=2000) as s:
        flags = s.problem_info()["linearize_flat"]  # bits: 1 k_linearize_flat possible, 8 k_linearize_win, 16 k_linearize_rows
        assert bool(flags & 16) == (lin in ("default", "matrix_free")), flags
        assert bool(flags & 8) == (lin == "window"), flags
        if lin == "flat":
            assert flags & 1 and not flags & 8, flags
        if lin == "group":
            assert not flags & (1 | 8 | 16), flags
        out = s.debug_linearization(poses, pts, lam)
    r, Jc, Jp = lm.linearize(poses, pts, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    free = np.ones(C, bool)
    free[0] = False
    Jc = Jc * free[g["obs_cam"]][:, None, None]
    U, V, W, gc, gp, cost = lm.normal_blocks(r, Jc, Jp, g["obs_cam"], g["obs_pt"], C, L)

    def close(a, b, tol=1e-10):
        np.testing.assert_allclose(a, b, rtol=0, atol=tol * max(np.abs(b).max(), 1e-300))

    close(out["U"], U); close(out["V"], V); close(out["W"], W); close(out["gc"], gc); close(out["gp"], gp)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, g["obs_cam"], g["obs_pt"], lam, free)
    close(out["S"], S, 1e-9)
    close(out["rhs"], rhs, 1e-9)
    dc = np.linalg.solve(S, -rhs).reshape(C, 6) * free[:, None]
    dp = lm.back_substitute(dc, Vi, W, gp, g["obs_cam"], g["obs_pt"], L)
    close(out["dc"], dc, 1e-7)
    close(out["dp"], dp, 1e-7)


def test_windowed_linearisation_on_an_unstructured_scene():
    """k_linearize_win (round 2's kernel, behind BA_LIN_ROWS=0) where its LDS window overflows: 200 cameras with random
    visibility, so the four chunks of a workgroup touch far more than 32 cameras and most camera groups go straight to
    the global slab.  U, g_c, V, g_p, W against the numpy restatement (1e-10 of the largest entry), and the windowed
    kernel against the group-per-track one and against today's k_linearize_rows."""
    import os
    sc = scenes.ring_scene(200, 4000, 5, visibility="random")
    C, L = 200, 4000
    a = (sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    outs = {}
    for mode in ("1", "0", "rows"):
        if mode != "rows":
            os.environ["BA_LIN_ROWS"] = "0"
            os.environ["BA_LIN_WIN"] = mode
        try:
            with _native.Solver(linear_solver=2) as s:
                s.set_problem(C, L, *a)
                flags = s.problem_info()["linearize_flat"]  # bits: 8 k_linearize_win, 16 k_linearize_rows
                assert bool(flags & 16) == (mode == "rows") and bool(flags & 8) == (mode == "1"), (mode, flags)
                outs[mode] = s.debug_linearization(sc["poses_guess"], sc["points_guess"], 1e-3, want_S=False)
        finally:
            os.environ.pop("BA_LIN_ROWS", None)
            os.environ.pop("BA_LIN_WIN", None)
    r, Jc, Jp = lm.linearize(sc["poses_guess"], sc["points_guess"], *a)
    free = np.ones(C, bool)
    free[0] = False
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, _ = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], C, L)
    for mode, out in outs.items():
        for k, ref in (("U", U), ("V", V), ("W", W), ("gc", gc), ("gp", gp)):
            np.testing.assert_allclose(out[k], ref, rtol=0, atol=1e-10 * np.abs(ref).max(), err_msg=f"{k} (BA_LIN_WIN={mode})")
    for mode in ("1", "rows"):
        np.testing.assert_allclose(outs[mode]["dc"], outs["0"]["dc"], rtol=0, atol=1e-9 * np.abs(outs["0"]["dc"]).max())


@pytest.mark.parametrize("name", ["noisy_10x300", "ring_50x2000"])
def test_converged_cost_matches_reference(golden, name):
    """north_star: converged cost within 1e-6 relative of the reference's scipy path: one-sided
    against the reference's loosely converged cost, two-sided against the polished minimum."""
    g = golden(name)
    if name == "ring_50x2000":
        C, L, d = (int(v) for v in g["ring_params"])
        sc = scenes.ring_scene(C, L, d)
    else:
        sc = g
    n_poses, poses, pts = _state(sc)
    with _solver(sc) as s:
        P, X, rep = s.solve(poses, pts)
        _, cost_check, _ = s.evaluate(P, X, want_residuals=False)
    ref, pol = float(g["ref_cost"]), float(g["polished_cost"])
    assert abs(rep["cost"] - cost_check) <= 1e-12 * pol
    assert rep["cost"] <= ref * (1 + 1e-6)
    assert abs(rep["cost"] - pol) <= 1e-6 * pol, (rep["cost"], pol)
    rms_pol = np.sqrt(2 * pol / len(sc["obs_cam"]))
    assert abs(rep["rms_reproj"] - rms_pol) <= 1e-6 * rms_pol
    assert np.allclose(np.linalg.norm(P[:, 3:7], axis=1), 1.0, atol=1e-12)
    np.testing.assert_array_equal(P[0], poses[0])  # fixed pose untouched


@pytest.mark.parametrize("name", ["test_scene_convergence", "test_scene_landmark_accuracy", "demo_scene"])
def test_noise_free_scenes_reach_zero_cost(golden, name):
    """Noise-free reference scenes: minimum cost is 0, so relative parity is meaningless
    (SURVEY.md section 7); assert absolute RMS reprojection <= 1e-6 px (reference: 1e-6..1e-9 px)
    and gauge-aligned estimates within 1e-5 of the reference's."""
    g = golden(name)
    n_poses, poses, pts = _state(g)
    with _solver(g) as s:
        P, X, rep = s.solve(poses, pts)
    assert rep["rms_reproj"] <= 1e-6, rep
    assert rep["cost"] <= max(float(g["ref_cost"]), 1e-12)
    # gauge: scale about camera 0's centre (camera-0 frame: about the origin)
    Xr = g["ref_x"][7 * n_poses:].reshape(-1, 3)
    tr = g["ref_x"][: 7 * n_poses].reshape(-1, 7)
    scale = np.linalg.norm(Xr[0]) / np.linalg.norm(X[0])
    assert np.abs(X * scale - Xr).max() <= 1e-5 * np.abs(Xr).max()
    assert np.abs(P[1:, :3] * scale - tr[:, :3]).max() <= 1e-5
    qr = tr[:, 3:7] / np.linalg.norm(tr[:, 3:7], axis=1, keepdims=True)
    sign = np.sign(np.sum(qr * P[1:, 3:7], axis=1, keepdims=True))
    assert np.abs(P[1:, 3:7] * sign - qr).max() <= 1e-5


def test_estimates_match_polished_after_gauge_alignment(golden):
    """Pose / point estimates within 1e-5 of the polished reference-path minimum after aligning the
    free scale (north_star); noisy scene, so the minimum is unique up to that gauge."""
    g = golden("noisy_10x300")
    n_poses, poses, pts = _state(g)
    with _solver(g, ftol=1e-14, gtol=1e-9) as s:
        P, X, rep = s.solve(poses, pts)
    xp = g["polished_x"]
    Xp = xp[7 * n_poses:].reshape(-1, 3)
    tp = xp[: 7 * n_poses].reshape(-1, 7)
    scale = np.sum(Xp * X) / np.sum(X * X)
    assert np.abs(X * scale - Xp).max() <= 1e-5 * np.abs(Xp).max()
    assert np.abs(P[1:, :3] * scale - tp[:, :3]).max() <= 1e-5
    qp = tp[:, 3:7] / np.linalg.norm(tp[:, 3:7], axis=1, keepdims=True)
    sign = np.sign(np.sum(qp * P[1:, 3:7], axis=1, keepdims=True))
    assert np.abs(P[1:, 3:7] * sign - qp).max() <= 1e-5


def test_ragged_unsorted_and_unobserved(golden):
    """Edge cases the reference cannot run: shuffled observation order,
    different counts per image, a landmark nobody observes, an extra fixed pose."""
    g = golden("noisy_10x300")
    n_poses, poses, pts = _state(g)
    rng = np.random.default_rng(5)
    keep = rng.random(len(g["obs_cam"])) > 0.2
    keep &= g["obs_pt"] != 7  # landmark 7 becomes unobserved
    perm = rng.permutation(np.flatnonzero(keep))
    oc, op, uv = g["obs_cam"][perm], g["obs_pt"][perm], g["obs_uv"][perm]
    fixed = np.zeros(n_poses + 1, np.uint8)
    fixed[[0, 4]] = 1
    with _native.Solver() as s:
        s.set_problem(n_poses + 1, len(pts), oc, op, uv, g["intr"], fixed)
        r, cost, _ = s.evaluate(poses, pts)
        x = np.concatenate([poses[1:].reshape(-1), pts.reshape(-1)])
        np.testing.assert_allclose(r, orc.residual_packed(x, g["pose0"], n_poses, oc, op, uv, g["intr"]), atol=1e-9)
        P, X, rep = s.solve(poses, pts)
    np.testing.assert_array_equal(X[7], pts[7])
    np.testing.assert_array_equal(P[4], poses[4])
    free = fixed == 0
    Po, Xo, repo = lm.lm_solve(poses, pts, oc, op, uv, g["intr"], free=free)
    assert abs(rep["cost"] - repo.cost) <= 1e-6 * repo.cost


def test_error_mapping(golden):
    from src.validation_error import ProcessingError, ValidationError
    g = golden("test_scene_convergence")
    n_poses, poses, pts = _state(g)
    with _native.Solver() as s:
        with pytest.raises(ValidationError):
            s.set_problem(n_poses + 1, len(pts), g["obs_cam"] + 100, g["obs_pt"], g["obs_uv"], g["intr"])
        s.set_problem(n_poses + 1, len(pts), g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
        bad = pts.copy()
        bad[3, 1] = np.nan
        with pytest.raises(ValidationError):
            s.solve(poses, bad)
        # a landmark exactly on camera 0's principal plane: z == 0 -> ProcessingError (camera.py:129-132)
        flat = pts.copy()
        flat[3, 2] = 0.0
        ident = poses.copy()
        ident[0] = [0, 0, 0, 1, 0, 0, 0]
        with pytest.raises(ProcessingError):
            s.evaluate(ident, flat)
        with pytest.raises(ProcessingError):
            s.solve(ident, flat)


def test_single_rank_communicator_path_matches_plain_solve():
    """The multi-GPU code path (RCCL all-reduce of [S | rhs | U,gc], every rank solving the same all-reduced system in
    a fixed summation order -- no broadcast of the camera step --, all-reduce of the statistics) with a one-rank
    communicator must reproduce the plain solve; ba_comm_info reports what the communicator itself says."""
    sc = scenes.ring_scene(50, 2000, 10)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    with _native.Solver() as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        P0, X0, rep0 = s.solve(sc["poses_guess"], sc["points_guess"])
    with _native.Solver() as s:
        assert s.comm_info() == (0, 1)  # no communicator yet
        s.comm_init(0, 1, _native.comm_unique_id())
        assert s.comm_info() == (0, 1)  # ncclCommUserRank / ncclCommCount of the one-rank communicator
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        P1, X1, rep1 = s.solve(sc["poses_guess"], sc["points_guess"])
    assert rep1["iterations"] == rep0["iterations"] and rep1["status"] == rep0["status"]
    assert abs(rep1["cost"] - rep0["cost"]) <= 1e-10 * rep0["cost"]
    assert abs(rep1["cost0"] - rep0["cost0"]) <= 1e-12 * rep0["cost0"]
    assert rep1["nobs"] == len(sc["obs_cam"])
    np.testing.assert_allclose(P1, P0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(X1, X0, rtol=0, atol=1e-8)


def test_large_system_path_matches_small_path():
    """The multi-kernel PCG path (used when the camera system does not fit one workgroup) against the
    one-workgroup path on the same scene."""
    import os
    sc = scenes.ring_scene(50, 2000, 10)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    out = []
    for small in ("1", "0"):
        os.environ["BA_SMALL_SOLVE"] = small
        try:
            with _native.Solver(pcg_tol=1e-6, linear_solver=1) as s:
                s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
                assert (s.problem_info()["reduced_solver"] == 1) == (small == "1")  # 1: k_solve_small, 0: the multi-kernel PCG
                out.append(s.solve(sc["poses_guess"], sc["points_guess"]))
        finally:
            os.environ.pop("BA_SMALL_SOLVE", None)
    assert abs(out[0][2]["cost"] - out[1][2]["cost"]) <= 1e-9 * out[0][2]["cost"]
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=1e-6)


def test_config2_full_size_properties(golden):
    """BASELINE configs[1] at full size (50 cameras x 20 000 landmarks x 200 000 observations), where no CPU oracle
    finishes in seconds: against the scalars recorded in the build container (tests/golden/ring_50x20000.npz: the
    reference's least_squares call on this scene and the tight-tolerance minimum of the numpy LM) -- one-sided against
    the reference's loosely converged cost, TWO-sided 1e-6 against the polished minimum -- and size-independent
    properties: cost == 0.5 |r|^2 of ba_evaluate at the estimate, the estimate is a fixed point (a second solve from
    it stops at once without lowering the cost), and the result does not depend on the observation order."""
    g = golden("ring_50x20000")
    ref, pol = float(g["ref_cost"]), float(g["polished_cost"])
    sc = scenes.ring_scene(50, 20000, 10)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    with _native.Solver() as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        P, X, rep = s.solve(sc["poses_guess"], sc["points_guess"])
        r, cost_eval, nb = s.evaluate(P, X)
        P2, X2, rep2 = s.solve(P, X)
    assert rep["status"] in (1, 2, 3) and rep["bad_pivots"] == 0
    assert rep["cost"] <= ref * (1 + 1e-6)
    assert abs(rep["cost"] - pol) <= 1e-6 * pol, (rep["cost"], pol)
    assert abs(0.5 * float(r @ r) - rep["cost"]) <= 1e-12 * rep["cost"] and abs(cost_eval - rep["cost"]) <= 1e-12 * rep["cost"]
    assert nb == 0
    assert rep2["iterations"] <= 2 and rep2["cost"] <= rep["cost"] * (1 + 1e-12)
    assert abs(rep2["cost"] - rep["cost"]) <= 1e-7 * rep["cost"]
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(sc["obs_cam"]))
    with _native.Solver() as s:
        s.set_problem(C, L, sc["obs_cam"][perm], sc["obs_pt"][perm], sc["obs_uv"][perm], sc["intr"])
        _, _, rep3 = s.solve(sc["poses_guess"], sc["points_guess"])
    assert abs(rep3["cost"] - rep["cost"]) <= 1e-9 * rep["cost"]
    assert np.allclose(np.linalg.norm(P[:, 3:7], axis=1), 1.0, atol=1e-12)


def test_config3_full_size_properties(golden):
    """BASELINE configs[2] (1000 cameras x 200 000 landmarks x 1 000 000 observations; AUTO takes block cyclic
    reduction), run to termination with the defaults: one-sided against the cost the reference's least_squares call
    reached on this scene (17 evaluations, 400 s on one core; it stops on ftol = 1e-6 about 1e-4 above the minimum)
    and TWO-sided 1e-6 against the polished minimum (tests/golden/ring_1000x200000.npz); cost == ba_evaluate's; every
    accepted step lowered the cost; a second solve from the estimate changes the cost by less than 1e-7."""
    g = golden("ring_1000x200000")
    ref, pol = float(g["ref_cost"]), float(g["polished_cost"])
    sc = scenes.ring_scene(1000, 200000, 5)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    with _native.Solver() as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        assert s.reduced_solver() == "cyclic_reduction"
        P, X, rep = s.solve(sc["poses_guess"], sc["points_guess"])
        _, cost_eval, nb = s.evaluate(P, X, want_residuals=False)
        _, _, rep2 = s.solve(P, X)
    assert rep["status"] in (1, 2, 3) and rep["bad_pivots"] == 0 and nb == 0
    assert rep["cost"] < ref
    assert abs(rep["cost"] - pol) <= 1e-6 * pol, (rep["cost"], pol)
    assert abs(cost_eval - rep["cost"]) <= 1e-12 * rep["cost"]
    h = rep["cost_history"]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(h, h[1:]))
    assert rep2["cost"] <= rep["cost"] * (1 + 1e-12) and rep["cost"] - rep2["cost"] <= 1e-7 * rep["cost"]
    # within the first 10 iterations the cost is already below the reference's final one
    assert h[9] < ref


def test_unstructured_visibility_matches_oracle():
    """Every landmark seen by a random camera subset (no two tracks need share a camera signature: the
    staged Schur kernel then writes a contribution tile at almost every step, and S is dense) against the
    oracle's LM on the same scene; also the atomics Schur path (BA_SCHUR_STAGED=0) against the staged one."""
    import os
    sc = scenes.ring_scene(12, 600, 5, visibility="random")
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    _, _, o = lm.lm_solve(sc["poses_guess"], sc["points_guess"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    costs = []
    for staged in ("1", "0"):
        os.environ["BA_SCHUR_STAGED"] = staged
        try:
            with _native.Solver() as s:
                s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
                assert (s.problem_info()["contrib_rows"] > 0) == (staged == "1")  # the staged kernel's tiles exist or not
                out = s.debug_linearization(sc["poses_guess"], sc["points_guess"], 1e-3, want_S=True)
                _, _, rep = s.solve(sc["poses_guess"], sc["points_guess"])
        finally:
            os.environ.pop("BA_SCHUR_STAGED", None)
        assert abs(rep["cost"] - o.cost) <= 1e-6 * o.cost, (staged, rep["cost"], o.cost)
        costs.append((rep["cost"], out["S"], out["rhs"]))
    np.testing.assert_allclose(costs[0][1], costs[1][1], rtol=0, atol=1e-10 * np.abs(costs[1][1]).max())
    np.testing.assert_allclose(costs[0][2], costs[1][2], rtol=0, atol=1e-10 * np.abs(costs[1][2]).max())


def test_run_to_run_determinism_of_the_cost():
    """Round 3: no sum on the solver's path depends on an arrival order any more (U / g_c: per-chunk rows and a
    fixed-order fold, k_linearize_rows / fold_u_rows; the scalar statistics: per-block partial sums and k_fold_stats; S,
    the landmark blocks and the reduced solvers were fixed-order already).  N solves of one problem agree BIT FOR BIT:
    cost, cost history, damping history, iteration count, poses and points -- with the one-workgroup PCG, with the
    cyclic reduction, and with a robust loss."""
    for C, L, d, opts in ((50, 2000, 10, {}), (240, 12000, 5, {}), (50, 2000, 10, dict(loss="huber", f_scale=1.0))):
        sc = scenes.ring_scene(C, L, d)
        with _native.Solver(**opts) as s:
            s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
            s.upload_guess(sc["poses_guess"], sc["points_guess"])
            reps, states = [], []
            for _ in range(5):
                reps.append(s.solve_resident())
                states.append(s.download_estimate(sc["poses_guess"], sc["points_guess"]))
        r0 = reps[0]
        assert r0["status"] in (1, 2, 3) and r0["iterations"] > 3
        for r, (P, X) in zip(reps[1:], states[1:]):
            assert r["cost"] == r0["cost"] and r["cost0"] == r0["cost0"] and r["iterations"] == r0["iterations"]
            assert r["cost_history"] == r0["cost_history"] and r["lambda_history"] == r0["lambda_history"]
            assert r["accepted_history"] == r0["accepted_history"] and r["pcg_history"] == r0["pcg_history"]
            assert np.array_equal(P, states[0][0]) and np.array_equal(X, states[0][1])


def test_residual_function_callback_paths(golden):
    """BundleAdjustment.residual_function as scipy would call it, repeatedly: the small problem goes through the
    entry point that allocates nothing, a problem beyond its limits through one cached handle; both equal the
    reference's recorded residuals / the oracle (1e-9 px)."""
    from src.bundle_adjustment import BundleAdjustment
    BundleAdjustment.clear_residual_cache()  # (whatever an earlier test left)
    g = golden("noisy_10x300")
    feats, pose0, cam, guesses, lms = scenes.to_containers(scenes.noisy_scene(10, 300))
    n_poses = len(guesses)
    for _ in range(3):
        r = BundleAdjustment.residual_function(g["x0"], pose0, n_poses, g["landmark_ids"], feats, cam)
        np.testing.assert_allclose(r, g["residual_at_x0"], rtol=0, atol=1e-9)
    assert BundleAdjustment._residual_solver is None
    big = scenes.ring_scene(40, 16000, 10)  # 160 000 observations > BA_STATIC_MAX_OBS
    feats, pose0, cam, guesses, lms = scenes.to_containers(big)
    ref = orc.residual_packed(big["x0"], big["pose0"], 39, big["obs_cam"], big["obs_pt"], big["obs_uv"], big["intr"])
    for k in range(3):
        x = big["x0"] + 1e-4 * k
        r = BundleAdjustment.residual_function(x, pose0, 39, big["landmark_ids"], feats, cam)
        if k == 0:
            np.testing.assert_allclose(r, ref, rtol=0, atol=1e-9)
            first = BundleAdjustment._residual_solver[1]
    assert BundleAdjustment._residual_solver[1] is first  # one handle for the three calls
    BundleAdjustment.clear_residual_cache()
    assert BundleAdjustment._residual_solver is None and first._h is None


def test_stale_hip_error_is_not_taken_for_a_failed_launch():
    """HIP's "last error" is per host thread and sticky.  An error left behind by an earlier call on the thread (here:
    a hipMalloc that cannot be satisfied, made straight through the runtime) must not make the next healthy solve /
    evaluate / static evaluate report 'a kernel launch failed'."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    sc = scenes.noisy_scene(6, 120)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])

    def poison():
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 60))
        assert rc != 0  # out of memory: the runtime remembers it until somebody asks

    with _native.Solver() as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        _, _, ref = s.solve(sc["poses_guess"], sc["points_guess"])
        poison()
        _, _, rep = s.solve(sc["poses_guess"], sc["points_guess"])
        assert rep["cost"] == ref["cost"]
        poison()
        r, cost, _ = s.evaluate(sc["poses_guess"], sc["points_guess"])
        poison()
        s.jacobian_blocks(sc["poses_guess"], sc["points_guess"])
        poison()
        s.debug_linearization(sc["poses_guess"], sc["points_guess"], 1e-3)
        poison()
        s.triangulate(sc["poses_guess"])
    poison()
    r2, cost2, _ = _native.evaluate_static(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"], sc["poses_guess"],
                                           sc["points_guess"])
    assert abs(cost2 - cost) <= 1e-12 * cost
    dev = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0 and dev.value == 0  # the static entry point leaves the caller's device current
