"""GPU parity tests of ba_triangulate through the C ABI: the reference test's
known answer, the numpy restatement on noisy multi-view tracks, distortion, edge cases, and the
triangulated points as the initial guess of the BA solve."""

import numpy as np
import pytest

import scenes
from oracle import triangulate_numpy as tri
from src import _native
from src.camera import PinHoleCamera
from src.camera_pose import CameraPose
from src.feature_observation import Features2D, ImageFeatures
from src.triangulation import triangulate_points, triangulate_tracks
from src.validation_error import ValidationError

pytestmark = pytest.mark.gpu


def test_reference_known_answer_through_the_mirrored_entry_point(golden):
    """tests/test_triangulation.py:120-131 of the reference, same call, same two assertions."""
    g = golden("known_answers")
    K = np.array([[500.0, 0, 320.0], [0, 500.0, 240.0], [0, 0, 1.0]])
    points_3d, reprojection_error = triangulate_points(g["tri_rot"][0].T, g["tri_poses"][0, :3], g["tri_rot"][1].T,
                                                       g["tri_poses"][1, :3], K, g["tri_uv"][0], g["tri_uv"][1])
    assert np.linalg.norm(points_3d - g["tri_landmarks"]) < 1e-6
    assert np.linalg.norm(reprojection_error) < 1e-6


def _problem(sc, **opt):
    s = _native.Solver(**opt)
    s.set_problem(len(sc["poses_gt"]), len(sc["points_gt"]), sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    return s


@pytest.mark.parametrize("name", ["noisy_10x300", "distorted_10x300"])
def test_matches_numpy_restatement(golden, name):
    """10 noisy views per landmark (sigma 0.5 px), ground-truth poses.  Linear estimate: the smallest
    eigenvector of A^T A (Jacobi, GPU) vs the smallest right singular vector of A (numpy SVD): 1e-9
    relative.  Refined: analytic Gauss-Newton vs the restatement's finite-difference one: 1e-6."""
    g = golden(name)
    poses = g["poses_gt"]
    L = len(g["points_gt"])
    with _problem(g) as s:
        X0, e0, v0 = s.triangulate(poses)
        X1, e1, v1 = s.triangulate(poses, refine_iters=10, min_depth=0.1)
    o0 = tri.triangulate_tracks(poses, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"], L)
    o1 = tri.triangulate_tracks(poses, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"], L, refine_iters=10, min_depth=0.1)
    assert v0.all() and v1.all() and o0[2].all() and o1[2].all()
    scale = np.abs(o0[0]).max()
    assert np.abs(X0 - o0[0]).max() <= 1e-9 * scale
    assert np.abs(e0 - o0[1]).max() <= 1e-8
    assert np.abs(X1 - o1[0]).max() <= 1e-6 * scale
    assert np.abs(e1 - o1[1]).max() <= 1e-6
    # refinement minimises the squared reprojection error; every point stays near the truth
    assert np.abs(X1 - g["points_gt"]).max() < 0.02


def test_edge_cases():
    """A landmark with one view, one with none, one behind a camera; untouched output rows."""
    sc = scenes.ring_scene(6, 42, 3)
    keep = np.ones(len(sc["obs_cam"]), bool)
    k4 = np.flatnonzero(sc["obs_pt"] == 4)
    keep[k4[1:]] = False          # landmark 4: one view
    keep[sc["obs_pt"] == 9] = False  # landmark 9: no view
    uv = sc["obs_uv"].copy()
    sc2 = dict(sc, obs_cam=sc["obs_cam"][keep], obs_pt=sc["obs_pt"][keep], obs_uv=uv[keep])
    sentinel = np.full((42, 3), 7.0)
    with _problem(sc2) as s:
        X, err, valid = s.triangulate(sc["poses_gt"], points=sentinel)
        # flip camera 0 around: its landmarks end up behind it
        poses_b = sc["poses_gt"].copy()
        q = poses_b[1, 3:7]
        poses_b[1, 3:7] = np.array([-q[2], q[3], q[0], -q[1]])  # q * (0,0,1,0): half turn about the camera's y axis
        Xb, errb, validb = s.triangulate(poses_b, points=sentinel, min_depth=0.0)
        with pytest.raises(ValidationError):
            s.triangulate(np.full((6, 7), np.nan))
    assert not valid[4] and not valid[9] and valid.sum() == 40
    assert np.isnan(err[4]) and np.isnan(err[9])
    assert np.all(X[4] == 7.0) and np.all(X[9] == 7.0)
    assert np.abs(X[valid] - sc["points_gt"][valid]).max() < 0.05
    seen_by_1 = np.unique(sc2["obs_pt"][sc2["obs_cam"] == 1])
    assert not validb[seen_by_1].any() or errb[seen_by_1][validb[seen_by_1]].min() > 1.0


def test_triangulated_points_start_the_solve_closer(golden):
    """Input side of the path: points triangulated from the pose guesses replace a poor landmark guess;
    the solve then reaches the same minimum in fewer iterations."""
    g = golden("noisy_10x300")
    n_poses = len(g["poses_gt"]) - 1
    poses0 = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts0 = g["x0"][7 * n_poses:].reshape(-1, 3)
    with _problem(g) as s:
        _, _, rep_a = s.solve(poses0, pts0)
        pts_t, err, valid = s.triangulate(poses0, refine_iters=5, min_depth=0.1)
        assert valid.all()
        _, cost_t, _ = s.evaluate(poses0, pts_t, want_residuals=False)
        _, _, rep_b = s.solve(poses0, pts_t)
    pol = float(g["polished_cost"])
    assert cost_t < rep_a["cost0"]
    assert abs(rep_b["cost"] - pol) <= 1e-6 * pol and abs(rep_a["cost"] - pol) <= 1e-6 * pol
    assert rep_b["iterations"] <= rep_a["iterations"]


def test_batched_tracks_entry_point(golden):
    g = golden("noisy_10x300")
    C = len(g["poses_gt"])
    cam = PinHoleCamera(*(float(v) for v in g["intr"]))
    feats = [ImageFeatures(float(c), 0, 640, 480, Features2D(g["obs_uv"][g["obs_cam"] == c],
                                                             g["landmark_ids"][g["obs_pt"][g["obs_cam"] == c]]))
             for c in range(C)]
    poses = [CameraPose(position=p[:3], orientation=p[3:7], timestamp=float(i)) for i, p in enumerate(g["poses_gt"])]
    lms, err = triangulate_tracks(feats, poses, cam, refine_iters=10, min_depth=0.1, max_reprojection_error=5.0)
    assert np.array_equal(lms.get_ids(), g["landmark_ids"])
    assert np.abs(lms.get_points_3d() - g["points_gt"]).max() < 0.02 and err.max() < 2.0


def test_jacobi_fallback_agrees_with_rayleigh_iteration(golden, monkeypatch):
    """The eigen-solve has two routes (Rayleigh-quotient iteration; cyclic Jacobi when that one's verdict
    is negative): BA_TRI_JACOBI=1 forces the second for every landmark."""
    g = golden("noisy_10x300")
    with _problem(g) as s:
        X0, e0, _ = s.triangulate(g["poses_gt"])
        monkeypatch.setenv("BA_TRI_JACOBI", "1")
        X1, e1, _ = s.triangulate(g["poses_gt"])
    assert np.abs(X0 - X1).max() <= 1e-9 * np.abs(X0).max() and np.abs(e0 - e1).max() <= 1e-8
