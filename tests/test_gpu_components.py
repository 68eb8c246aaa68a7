"""k_solve_components (csrc/ba_components.h): the block Cholesky per connected component through the C ABI, one LM step
against a dense numpy solve of the system the library itself exports, for every component size (1 ... 12 free
cameras: one block step, partly filled last tiles of the trailing update, the 73-row case that needs two passes of
the wave) and for a batch of components of different sizes in one call."""
import numpy as np
import pytest

import scenes
from src import _native

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C", [2, 3, 4, 7, 10, 12, 13])
def test_component_step_matches_a_dense_solve(C):
    sc = scenes.ring_scene(C, 60 * C, min(C, 6))
    L = len(sc["points_gt"])
    with _native.Solver(linear_solver=2) as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        assert s.reduced_solver() == "component_cholesky"
        out = s.debug_linearization(sc["poses_guess"], sc["points_guess"], 1e-3, want_S=True)
    dc = np.linalg.solve(out["S"], -out["rhs"]).reshape(C, 6)
    dc[0] = 0.0
    np.testing.assert_allclose(out["dc"], dc, rtol=0, atol=1e-10 * np.abs(dc).max())


def test_components_of_different_sizes_in_one_batch():
    """Windows of 3, 6, 9 and 12 keyframes in one batched call: each window's result equals its own solve."""
    wins = []
    for k, C in enumerate((3, 6, 9, 12, 5, 12)):
        sc = scenes.ring_scene(C, 40 * C, min(C, 5), seed=10 + k)
        wins.append(dict(poses=sc["poses_guess"], points=sc["points_guess"], obs_cam=sc["obs_cam"], obs_pt=sc["obs_pt"],
                         obs_uv=sc["obs_uv"]))
        intr = sc["intr"]
    with _native.Solver() as s:
        _, _, _, cost, _ = s.solve_batched(wins, intr)
    for w, cb in zip(wins, cost):
        C, L = len(w["poses"]), len(w["points"])
        with _native.Solver(linear_solver=2) as s1:
            s1.set_problem(C, L, w["obs_cam"], w["obs_pt"], w["obs_uv"], intr)
            _, _, rep = s1.solve(w["poses"], w["points"])
        assert abs(cb - rep["cost"]) <= 1e-9 * rep["cost"]
