"""GPU tests of the batched local bundle adjustment (ba_solve_batched, BASELINE config 5) and of the
dense Cholesky linear solver it uses (k_solve_components), through the C ABI."""

import numpy as np
import pytest

import scenes
from oracle import lm_schur_numpy as lm
from src import _native
from src.validation_error import ValidationError

pytestmark = pytest.mark.gpu


def _window(seed, K=10, L=120):
    sc = scenes.local_window_scene(seed, K=K, L=L)
    return dict(obs_cam=sc["obs_cam"], obs_pt=sc["obs_pt"], obs_uv=sc["obs_uv"], poses=sc["poses_guess"],
                points=sc["points_guess"]), sc


def test_dense_cholesky_step_matches_oracle(golden):
    """linear_solver = dense Cholesky: the LM step of one linearisation against numpy's dense solve of the
    oracle's reduced system (exact solver, so 1e-9 relative instead of the PCG test's 1e-7)."""
    g = golden("noisy_10x300")
    n_poses = len(g["poses_gt"]) - 1
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    C, L, lam = n_poses + 1, len(pts), 1e-3
    with _native.Solver(linear_solver=2) as s:
        s.set_problem(C, L, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
        out = s.debug_linearization(poses, pts, lam)
    r, Jc, Jp = lm.linearize(poses, pts, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    free = np.ones(C, bool)
    free[0] = False
    Jc = Jc * free[g["obs_cam"]][:, None, None]
    U, V, W, gc, gp, _ = lm.normal_blocks(r, Jc, Jp, g["obs_cam"], g["obs_pt"], C, L)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, g["obs_cam"], g["obs_pt"], lam, free)
    dc = np.linalg.solve(S, -rhs).reshape(C, 6) * free[:, None]
    dp = lm.back_substitute(dc, Vi, W, gp, g["obs_cam"], g["obs_pt"], L)
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-9 * np.abs(dc).max())
    np.testing.assert_allclose(out["dp"], dp, rtol=0, atol=1e-9 * np.abs(dp).max())


def test_dense_cholesky_converges_like_pcg(golden):
    g = golden("noisy_10x300")
    n_poses = len(g["poses_gt"]) - 1
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    with _native.Solver(linear_solver=2) as s:
        s.set_problem(n_poses + 1, len(pts), g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
        _, _, rep = s.solve(poses, pts)
    pol = float(g["polished_cost"])
    assert rep["pcg_iters"] == 0
    assert abs(rep["cost"] - pol) <= 1e-6 * pol


def test_whole_gpu_dense_cholesky_matches_oracle(golden):
    """Camera systems with one large connected component: the blocked whole-GPU Cholesky (potrf / trsm /
    MFMA update / backward sweep, ba_dense.h).  Step of one linearisation against numpy's dense solve
    (1e-9 relative), converged cost against the polished minimum of the fixture (1e-6)."""
    g = golden("ring_50x2000")
    C, L, d = (int(v) for v in g["ring_params"])
    sc = scenes.ring_scene(C, L, d)
    poses, pts, lam = sc["poses_guess"], sc["points_guess"], 1e-3
    with _native.Solver(linear_solver=2) as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        out = s.debug_linearization(poses, pts, lam, want_S=False)
        _, _, rep = s.solve(poses, pts)
    r, Jc, Jp = lm.linearize(poses, pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    free = np.ones(C, bool)
    free[0] = False
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, _ = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], C, L)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sc["obs_cam"], sc["obs_pt"], lam, free)
    dc = np.linalg.solve(S, -rhs).reshape(C, 6) * free[:, None]
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-9 * np.abs(dc).max())
    pol = float(g["polished_cost"])
    assert rep["pcg_iters"] == 0
    assert abs(rep["cost"] - pol) <= 1e-6 * pol


def test_batched_windows_match_individual_solves():
    """B independent windows solved as one block-diagonal problem: every window must land on the minimum
    that its own solve finds (1e-6 relative in cost, north_star's bar), which the oracle's LM confirms for
    the first one; poses / points agree after the solve to 1e-5."""
    seeds = [0, 1, 2, 3, 4, 5, 6]
    sizes = [120, 80, 200, 120, 100, 150, 90]  # ragged batch
    probs, scs = zip(*[_window(s, L=l) for s, l in zip(seeds, sizes)])
    intr = scs[0]["intr"]
    with _native.Solver() as s:
        P, X, cost0, cost, rep = s.solve_batched(probs, intr)
    assert rep["pcg_iters"] == 0 and rep["status"] in (1, 2, 3)
    assert abs(cost.sum() - rep["cost"]) <= 1e-10 * rep["cost"]
    assert abs(cost0.sum() - rep["cost0"]) <= 1e-10 * rep["cost0"]
    for b, (p, sc) in enumerate(zip(probs, scs)):
        with _native.Solver() as s1:
            s1.set_problem(len(p["poses"]), len(p["points"]), p["obs_cam"], p["obs_pt"], p["obs_uv"], intr)
            P1, X1, rep1 = s1.solve(p["poses"], p["points"])
        assert abs(cost[b] - rep1["cost"]) <= 1e-6 * rep1["cost"], (b, cost[b], rep1["cost"])
        assert abs(cost0[b] - rep1["cost0"]) <= 1e-12 * rep1["cost0"]
        np.testing.assert_array_equal(P[b][0], np.asarray(p["poses"])[0])  # first camera of a window is fixed
        # monocular scale about camera 0 is a free gauge direction: compare after aligning it
        sc1 = np.linalg.norm(X1 - P1[0, :3]) / np.linalg.norm(X[b] - P[b][0, :3])
        np.testing.assert_allclose(P[b][:, :3] * sc1, P1[:, :3], atol=1e-5 * max(1.0, np.abs(P1[:, :3]).max()))
        np.testing.assert_allclose(X[b] * sc1, X1, atol=1e-5 * np.abs(X1).max())
    # the oracle's LM (numpy, exact reduced solve) on window 0
    p, sc = probs[0], scs[0]
    o = lm.lm_solve(np.asarray(p["poses"]), np.asarray(p["points"]), p["obs_cam"], p["obs_pt"], p["obs_uv"], intr)
    cost_o = o[2].cost
    assert abs(cost[0] - cost_o) <= 1e-6 * cost_o


@pytest.mark.parametrize("matrix_free", ["0", "1"], ids=["stored_w", "matrix_free"])
def test_batched_windows_keep_their_own_damping(monkeypatch, matrix_free):
    """Levenberg-Marquardt per window (CompLm): every window of a batch has its own damping sequence and its own
    accept / reject decisions, so the batch is B independent solves that share their launches.  Five ordinary windows
    and one whose guess is far off (rejected steps, damping raised): every window must land where its own solve
    (same exact reduced solver) lands, to 1e-9 in cost, the batch must take exactly as many iterations as its slowest
    window alone; with ONE damping for the batch (BA_COMP_LM=0, the round-1 behaviour) the minima are the same (1e-6)
    and the road is no shorter."""
    monkeypatch.setenv("BA_MATRIX_FREE", matrix_free)  # both ways of getting W (batches of >= 2 M observations rebuild it)
    seeds, sizes = [0, 1, 2, 3, 4, 5], [120, 80, 200, 120, 100, 150]
    probs, scs = zip(*[_window(s, L=l) for s, l in zip(seeds, sizes)])
    probs = [dict(p) for p in probs]
    rng = np.random.default_rng(42)
    far = np.asarray(probs[3]["poses"]).copy()
    far[1:, :3] += 0.25 * rng.normal(size=(len(far) - 1, 3))
    q = far[1:, 3:7] + 0.08 * rng.normal(size=(len(far) - 1, 4))
    far[1:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    probs[3]["poses"] = far
    probs[3]["points"] = np.asarray(probs[3]["points"]) + 1.0 * rng.normal(size=np.asarray(probs[3]["points"]).shape)
    intr = scs[0]["intr"]
    solo = []
    for p in probs:
        with _native.Solver(linear_solver=2) as s1:
            s1.set_problem(len(p["poses"]), len(p["points"]), p["obs_cam"], p["obs_pt"], p["obs_uv"], intr)
            solo.append(s1.solve(p["poses"], p["points"])[2])
    assert all(r["status"] == 2 for r in solo)
    its = [r["iterations"] for r in solo]
    assert solo[3]["accepted"] < solo[3]["iterations"] and its[3] > max(its[:3] + its[4:])  # the far window struggles
    with _native.Solver() as s:
        P, X, cost0, cost, rep = s.solve_batched(probs, intr)
    assert rep["status"] == 2 and rep["iterations"] == max(its), (rep["iterations"], its)
    for b, r in enumerate(solo):
        assert abs(cost0[b] - r["cost0"]) <= 1e-12 * r["cost0"]
        assert abs(cost[b] - r["cost"]) <= 1e-9 * r["cost"], (b, cost[b], r["cost"])
    assert abs(rep["cost"] - sum(r["cost"] for r in solo)) <= 1e-9 * rep["cost"]
    assert abs(rep["cost0"] - sum(r["cost0"] for r in solo)) <= 1e-12 * rep["cost0"]
    monkeypatch.setenv("BA_COMP_LM", "0")
    with _native.Solver() as s:
        _, _, _, cost_shared, rep_shared = s.solve_batched(probs, intr)
    assert rep_shared["iterations"] >= rep["iterations"], (rep_shared["iterations"], rep["iterations"])
    for b, r in enumerate(solo):  # the same minima, by a longer road
        assert abs(cost_shared[b] - r["cost"]) <= 1e-6 * r["cost"]


def test_batched_rejects_bad_input():
    probs, scs = zip(*[_window(s) for s in (0, 1)])
    bad = dict(probs[1])
    bad["obs_cam"] = np.asarray(bad["obs_cam"]).copy()
    bad["obs_cam"][3] = 99  # camera index outside its window
    with _native.Solver() as s:
        with pytest.raises(ValidationError):
            s.solve_batched([probs[0], bad], scs[0]["intr"])


def _open_trajectory(C, L, d):
    """Ring scene without the observations that wrap around: an open trajectory (ragged tracks at its end)."""
    sc = scenes.ring_scene(C, L, d)
    keep = sc["obs_cam"] >= (sc["obs_pt"] % C)
    return dict(sc, obs_cam=sc["obs_cam"][keep], obs_pt=sc["obs_pt"][keep], obs_uv=sc["obs_uv"][keep])


@pytest.mark.parametrize("segments", [0, 4], ids=["one_level", "four_segments"])
@pytest.mark.parametrize("make", [lambda: scenes.ring_scene(60, 1200, 5), lambda: _open_trajectory(60, 1200, 5)],
                         ids=["closed_loop", "open_trajectory"])
def test_banded_cholesky_step_matches_oracle(make, segments, monkeypatch):
    """One LM step of the banded + arrow Cholesky (ba_banded.h), in one level and cut into four segments that
    are eliminated in parallel, against a dense numpy solve of the oracle's reduced system: 1e-9 relative."""
    monkeypatch.setenv("BA_BSEG_P", str(segments))
    sc = make()
    C, L = len(sc["poses_gt"]), len(sc["points_gt"])
    poses, pts = sc["poses_guess"], sc["points_guess"]
    lam = 1e-3
    with _native.Solver(linear_solver=3) as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        assert s.reduced_solver() == "banded_cholesky"
        out = s.debug_linearization(poses, pts, lam)
    r, Jc, Jp = lm.linearize(poses, pts, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    free = np.ones(C, bool)
    free[0] = False
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, _ = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], C, L)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sc["obs_cam"], sc["obs_pt"], lam, free)
    dc = np.linalg.solve(S, -rhs).reshape(C, 6) * free[:, None]
    dp = lm.back_substitute(dc, Vi, W, gp, sc["obs_cam"], sc["obs_pt"], L)
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-9 * np.abs(dc).max())
    np.testing.assert_allclose(out["dp"], dp, rtol=0, atol=1e-9 * np.abs(dp).max())


def test_banded_cholesky_is_chosen_and_converges_like_dense():
    """The banded solver on a long trajectory (AUTO now takes cyclic reduction there, tests/test_gpu_cyclic_reduction.py):
    the solve follows the dense Cholesky's path."""
    sc = scenes.ring_scene(300, 6000, 5)
    reps = {}
    for name, opt in (("auto", dict(linear_solver=3)), ("dense", dict(linear_solver=2))):
        with _native.Solver(**opt) as s:
            s.set_problem(300, 6000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
            reps[name] = (s.reduced_solver(), s.solve(sc["poses_guess"], sc["points_guess"])[2])
    assert reps["auto"][0] == "banded_cholesky" and reps["dense"][0] == "dense_cholesky"
    a, b = reps["auto"][1], reps["dense"][1]
    assert a["iterations"] == b["iterations"] and a["status"] == b["status"]
    assert abs(a["cost"] - b["cost"]) <= 1e-9 * b["cost"]


def test_banded_cholesky_rejects_unstructured_problems():
    sc = scenes.ring_scene(200, 4000, 5, visibility="random")
    with _native.Solver(linear_solver=3) as s:
        with pytest.raises(Exception):
            s.set_problem(200, 4000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])


@pytest.mark.parametrize("d,segments", [(3, 4), (3, 8), (4, 6), (7, 6)])
def test_banded_cholesky_other_bandwidths(d, segments, monkeypatch):
    """Track lengths 3 / 4 / 7 (half-bandwidths 2 / 3 / 6) with forced segment counts; where the separators'
    system does not fit the LDS window (d = 7) the solver stays on one level.  Same bar: 1e-9 of a dense solve."""
    monkeypatch.setenv("BA_BSEG_P", str(segments))
    sc = scenes.ring_scene(120, 2400, d)
    C, L = 120, 2400
    poses, pts = sc["poses_guess"], sc["points_guess"]
    lam = 1e-2
    with _native.Solver(linear_solver=3) as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        out = s.debug_linearization(poses, pts, lam, want_S=True)
    S, rhs = out["S"], out["rhs"]
    dc = np.linalg.solve(S, -rhs).reshape(C, 6)
    dc[0] = 0.0
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-9 * np.abs(dc).max())


def test_banded_cholesky_on_a_small_dense_pattern(golden):
    """Forced on a problem whose cameras all see each other (9 free cameras: the band is the whole matrix) and
    on a two-camera problem (one free camera): still an exact solver, converged cost as the default solver's."""
    g = golden("noisy_10x300")
    n_poses = len(g["poses_gt"]) - 1
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    with _native.Solver(linear_solver=3) as s:
        s.set_problem(n_poses + 1, len(pts), g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
        assert s.reduced_solver() == "banded_cholesky"
        _, _, rep = s.solve(poses, pts)
    pol = float(g["polished_cost"])
    assert abs(rep["cost"] - pol) <= 1e-6 * pol
    two = g["obs_cam"] < 2
    costs = []
    for ls in (0, 3):
        with _native.Solver(linear_solver=ls, ftol=1e-12) as s:
            s.set_problem(2, len(pts), g["obs_cam"][two], g["obs_pt"][two], g["obs_uv"][two], g["intr"])
            costs.append(s.solve(poses[:2], pts)[2]["cost"])
    assert abs(costs[0] - costs[1]) <= 1e-8 * costs[0]


def test_batched_at_config5_size(golden):
    """BASELINE configs[4] at its stated size: 1024 windows of 10 keyframes x 100 landmarks (1 024 000 observations) in
    one call.  Every window: its cost fell, is 0.5 |r|^2 of its own residuals, sits at the noise level (0.5 px per
    axis), first keyframe untouched.  Four of the windows are pinned against THE REFERENCE ITSELF (round 3:
    tests/golden/window_<seed>_10x100.npz hold what the reference's BundleAdjustment.optimize, bundle_adjustment.py:
    156-288, reaches on exactly these windows, and the dense exact-trust-region minimum from there): one-sided against
    the reference's cost, two-sided 1e-6 against the polished minimum; and against the oracle's LM on the window alone
    (every window runs its own damping sequence, accept / reject decisions and ftol test, CompLm, so each must reach
    its own minimum whatever the others do)."""
    B, L = 1024, 100
    probs, scs = zip(*[_window(s, L=L) for s in range(B)])
    intr = scs[0]["intr"]
    pinned = {}
    for b in (0, 333, 700, 1023):
        g = golden(f"window_{b}_10x100")
        p = probs[b]
        # the fixture's window IS window b of the batch (same generator, same seed)
        assert np.array_equal(g["obs_cam"], p["obs_cam"]) and np.array_equal(g["obs_pt"], p["obs_pt"])
        np.testing.assert_allclose(g["obs_uv"], p["obs_uv"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(g["poses_guess"], p["poses"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(g["points_guess"], p["points"], rtol=0, atol=1e-12)
        pinned[b] = g
    with _native.Solver() as s:
        P, X, cost0, cost, rep = s.solve_batched(probs, intr)
        assert s.reduced_solver() == "component_cholesky"
    assert rep["status"] in (1, 2, 3) and rep["bad_pivots"] == 0
    assert rep["nobs"] == B * 10 * L
    assert abs(cost.sum() - rep["cost"]) <= 1e-10 * rep["cost"]
    assert np.all(cost < cost0)
    rms = np.sqrt(2 * cost / (10 * L))
    assert np.all((rms > 0.55) & (rms < 0.80)), (rms.min(), rms.max())
    for b in range(0, B, 97):
        np.testing.assert_array_equal(P[b][0], np.asarray(probs[b]["poses"])[0])
    for b, g in pinned.items():
        p = probs[b]
        ref_cost, pol = float(g["ref_cost"]), float(g["polished_cost"])
        assert cost[b] <= ref_cost * (1 + 1e-6), (b, cost[b], ref_cost)
        assert abs(cost[b] - pol) <= 1e-6 * pol, (b, cost[b], pol)
        o = lm.lm_solve(np.asarray(p["poses"]), np.asarray(p["points"]), p["obs_cam"], p["obs_pt"], p["obs_uv"], intr)[2]
        assert abs(cost[b] - o.cost) <= 1e-6 * o.cost, (b, cost[b], o.cost)
    print(f"1024 windows: {rep['solve_ms']:.2f} ms, {rep['iterations']} LM iterations, "
          f"{1e3 * rep['solve_ms'] / B:.2f} us per window")


def test_batched_8192_windows_properties():
    """The upper end of SURVEY.md section 8(d)'s range for configs[4]: B = 8192 windows (10 keyframes x 100 landmarks,
    8.2 M observations) in one call: by properties only (every window's cost fell and sits at the noise level, the sum
    of the per-window costs is the reported cost, first keyframes untouched, no bad pivot, terminated by a tolerance);
    windows repeat with period 1024 seeds, so window b must land where window b mod 1024 lands (1e-9 relative: the
    same window in another position of the batch)."""
    B, L, period = 8192, 100, 1024
    base = [_window(s, L=L) for s in range(period)]
    probs = [base[b % period][0] for b in range(B)]
    intr = base[0][1]["intr"]
    with _native.Solver() as s:
        P, X, cost0, cost, rep = s.solve_batched(probs, intr)
        assert s.reduced_solver() == "component_cholesky"
    assert rep["status"] in (1, 2, 3) and rep["bad_pivots"] == 0 and rep["nobs"] == B * 10 * L
    assert abs(cost.sum() - rep["cost"]) <= 1e-10 * rep["cost"]
    assert np.all(cost < cost0)
    rms = np.sqrt(2 * cost / (10 * L))
    assert np.all((rms > 0.55) & (rms < 0.80)), (rms.min(), rms.max())
    for b in range(0, B, 509):
        np.testing.assert_array_equal(P[b][0], np.asarray(probs[b]["poses"])[0])
    c = cost.reshape(B // period, period)
    assert np.all(np.abs(c - c[0]) <= 1e-9 * c[0])
    print(f"8192 windows: {rep['solve_ms']:.2f} ms, {rep['iterations']} LM iterations, "
          f"{1e3 * rep['solve_ms'] / B:.2f} us per window")


def test_batched_solve_repeats_bit_for_bit():
    """Round 3: the per-window statistics (model decrease, trial cost, trial cheirality count) are per-chunk partial sums
    added up in a fixed order (comp_store / comp_fold_stats_block), not atomics: two batched solves of the same windows give
    the same bits (costs, estimates, iteration count), on one handle and on a fresh one."""
    B, L = 96, 100
    probs, scs = zip(*[_window(s, L=L) for s in range(B)])
    intr = scs[0]["intr"]
    runs = []
    with _native.Solver() as s:
        runs.append(s.solve_batched(probs, intr))
        runs.append(s.solve_batched(probs, intr))
    with _native.Solver() as s:
        runs.append(s.solve_batched(probs, intr))
    P0, X0, c00, c0, rep0 = runs[0]
    for P, X, c0i, c, rep in runs[1:]:
        assert rep["iterations"] == rep0["iterations"] and rep["cost"] == rep0["cost"]
        np.testing.assert_array_equal(c, c0)
        np.testing.assert_array_equal(c0i, c00)
        for b in range(B):
            np.testing.assert_array_equal(P[b], P0[b])
            np.testing.assert_array_equal(X[b], X0[b])
