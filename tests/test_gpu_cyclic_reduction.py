"""Block cyclic reduction of a trajectory's reduced camera system (csrc/ba_bcr.h) through the C ABI: one LM step against a
dense numpy solve of the system the library itself exports (1e-9 relative, the bar of the banded Cholesky it replaces),
over closed loops and open trajectories, bandwidths 2 ... 6, camera counts that leave padded super-blocks and every
small block count (3 ... 9: the recursion's odd / even / merged levels), and whole solves against the other solvers."""

import numpy as np
import pytest

import scenes
from oracle import lm_schur_numpy as lm
from src import _native

pytestmark = pytest.mark.gpu


def _open_trajectory(C, L, d):
    sc = scenes.ring_scene(C, L, d)
    keep = sc["obs_cam"] >= (sc["obs_pt"] % C)
    return dict(sc, obs_cam=sc["obs_cam"][keep], obs_pt=sc["obs_pt"][keep], obs_uv=sc["obs_uv"][keep])


def _step(sc, lam, **opt):
    C, L = len(sc["poses_gt"]), len(sc["points_gt"])
    with _native.Solver(linear_solver=4, **opt) as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        assert s.reduced_solver() == "cyclic_reduction"
        info = s.problem_info()
        out = s.debug_linearization(sc["poses_guess"], sc["points_guess"], lam, want_S=True)
    dc = np.linalg.solve(out["S"], -out["rhs"]).reshape(C, 6)
    dc[0] = 0.0
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-9 * np.abs(dc).max())
    return out, info


@pytest.mark.parametrize("make", [lambda: scenes.ring_scene(60, 1200, 5), lambda: _open_trajectory(60, 1200, 5)],
                         ids=["closed_loop", "open_trajectory"])
def test_cyclic_reduction_step_matches_oracle(make):
    """Against the oracle's own reduced system and back-substitution (not only the library's exported S)."""
    sc = make()
    C, L = len(sc["poses_gt"]), len(sc["points_gt"])
    lam = 1e-3
    out, info = _step(sc, lam)
    assert info["band_bw"] == 4  # cameras per super-block = the half-bandwidth: 59 = 3 + 14 x 4, the short block first
    r, Jc, Jp = lm.linearize(sc["poses_guess"], sc["points_guess"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    free = np.ones(C, bool)
    free[0] = False
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, _ = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], C, L)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sc["obs_cam"], sc["obs_pt"], lam, free)
    dc = np.linalg.solve(S, -rhs).reshape(C, 6) * free[:, None]
    dp = lm.back_substitute(dc, Vi, W, gp, sc["obs_cam"], sc["obs_pt"], L)
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-9 * np.abs(dc).max())
    np.testing.assert_allclose(out["dp"], dp, rtol=0, atol=1e-9 * np.abs(dp).max())


@pytest.mark.parametrize("C,d", [(13, 5), (17, 5), (21, 5), (25, 5), (29, 5), (33, 5), (37, 5), (41, 5),   # 3 ... 10 blocks of 4
                                 (120, 3), (121, 4), (123, 7), (200, 6), (257, 5),
                                 (90, 2), (148, 8), (152, 8)])   # 1, 7 and 7 or 8 cameras per block: 1 x 1 and 3 x 3 matrix-instruction tiles
def test_cyclic_reduction_block_counts_and_bandwidths(C, d):
    """C - 1 free cameras in super-blocks of d - 1: every small block count (levels with an odd count, an even count, the
    merged two-block level, the single last block), padded blocks (C - 1 not a multiple), bandwidths 1 ... 8 (block sizes
    6 ... 48: every tile count of the matrix-instruction products, with and without padding inside the 16-wide tiles)."""
    L = 20 * C
    out, info = _step(scenes.ring_scene(C, L, d), 1e-2)
    # blocks of exactly g = d - 1 cameras with a SHORT FIRST block where that keeps the loop closure between the first and
    # the last block: the cameras that reach back through the closure are the first g - 1 (camera 0, fixed, sits in the
    # gap), so the short block must hold g - 1 of them; otherwise round 2's cut, (C - 1) // g blocks of g or g + 1 cameras
    Cf, g = C - 1, d - 1
    if Cf % g in (0, g - 1):
        assert info["band_segments"] == -(-Cf // g) and info["band_bw"] == g
    else:
        assert info["band_segments"] == Cf // g and info["band_bw"] == g + 1


@pytest.mark.parametrize("C,d", [(60, 5), (121, 4), (257, 5)])
def test_cyclic_reduction_equal_split_layout(C, d, monkeypatch):
    """Round 2's cut of the cameras (nearly equal blocks, one camera more than the half-bandwidth in some) stays
    available (BA_BCR_SHORT_FIRST=0) and is what the solver falls back to when a short first block breaks the
    block-tridiagonal pattern: same step."""
    monkeypatch.setenv("BA_BCR_SHORT_FIRST", "0")
    out, info = _step(scenes.ring_scene(C, 20 * C, d), 1e-2)
    assert info["band_segments"] == (C - 1) // (d - 1)


def test_cyclic_reduction_is_chosen_and_follows_the_other_direct_solvers():
    """AUTO takes cyclic reduction for a long trajectory; the solve follows the dense and the banded Cholesky's path
    (same iterations, cost to 1e-9)."""
    sc = scenes.ring_scene(300, 6000, 5)
    reps = {}
    for name, opt in (("auto", {}), ("dense", dict(linear_solver=2)), ("banded", dict(linear_solver=3))):
        with _native.Solver(**opt) as s:
            s.set_problem(300, 6000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
            reps[name] = (s.reduced_solver(), s.solve(sc["poses_guess"], sc["points_guess"])[2])
    assert [reps[k][0] for k in ("auto", "dense", "banded")] == ["cyclic_reduction", "dense_cholesky", "banded_cholesky"]
    a = reps["auto"][1]
    for other in ("dense", "banded"):
        b = reps[other][1]
        assert a["iterations"] == b["iterations"] and a["status"] == b["status"] and a["bad_pivots"] == 0
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * b["cost"]


def test_cyclic_reduction_is_bitwise_reproducible():
    """No atomics anywhere in the solver: two solves of one problem give the same bits in the camera step (what lets
    every rank of a multi-GPU solve factor the all-reduced system by itself, without a broadcast)."""
    sc = scenes.ring_scene(120, 2400, 5)
    outs = []
    for _ in range(2):
        with _native.Solver(linear_solver=4) as s:
            s.set_problem(120, 2400, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
            outs.append(s.debug_linearization(sc["poses_guess"], sc["points_guess"], 1e-3, want_S=True))
    # (round 3: U / gc no longer come through atomics either, so the system itself repeats bit for bit)
    np.testing.assert_array_equal(outs[0]["S"], outs[1]["S"])
    np.testing.assert_array_equal(outs[0]["rhs"], outs[1]["rhs"])
    np.testing.assert_array_equal(outs[0]["dc"], outs[1]["dc"])
    np.testing.assert_array_equal(outs[0]["dp"], outs[1]["dp"])


def test_cyclic_reduction_rejects_unstructured_problems():
    sc = scenes.ring_scene(200, 4000, 5, visibility="random")
    with _native.Solver(linear_solver=4) as s:
        with pytest.raises(Exception):
            s.set_problem(200, 4000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
