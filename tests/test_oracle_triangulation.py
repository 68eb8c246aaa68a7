"""oracle/triangulate_numpy.py against the known answer of the reference's own triangulation test
(tests/test_triangulation.py:130-131; inputs recorded with the reference's classes in
tests/golden/known_answers.npz).  CPU only."""

import numpy as np

import scenes
from oracle import triangulate_numpy as tri


def test_dlt_reproduces_the_reference_tests_known_answer(golden):
    g = golden("known_answers")
    pts = np.array([tri.dlt_point(g["tri_poses"], g["tri_uv"][:, j], g["tri_intr"]) for j in range(8)])
    err = np.array([tri.reprojection_error(g["tri_poses"], g["tri_uv"][:, j], g["tri_intr"], pts[j]) for j in range(8)])
    assert np.linalg.norm(pts - g["tri_landmarks"]) < 1e-6  # the reference's assertions, verbatim bars
    assert np.linalg.norm(err) < 1e-6
    # the projection matrix convention: rot_F1_F = R^T, t = -R^T c (visual_odometry.py:547-553)
    P = tri.projection_matrix(g["tri_poses"][1], g["tri_intr"])
    x = P @ np.r_[g["tri_landmarks"][3], 1.0]
    np.testing.assert_allclose(x[:2] / x[2], g["tri_uv"][1, 3], rtol=0, atol=1e-9)
    np.testing.assert_allclose(P[:, :3], np.diag([1, 1, 1.0]) @ np.array(
        [[500, 0, 320], [0, 500, 240], [0, 0, 1.0]]) @ g["tri_rot"][1].T, rtol=0, atol=1e-9)


def test_gauss_newton_refinement_lowers_the_reprojection_error():
    sc = scenes.ring_scene(12, 60, 5)
    poses = sc["poses_gt"]
    worse = 0
    for j in range(0, 60, 7):
        k = np.flatnonzero(sc["obs_pt"] == j)
        P, uv = poses[sc["obs_cam"][k]], sc["obs_uv"][k]
        X0 = tri.dlt_point(P, uv, sc["intr"])
        X1 = tri.gauss_newton_point(P, uv, sc["intr"], x_init=X0)
        e0, e1 = (sum(tri.reprojection_error(P[i:i + 1], uv[i:i + 1], sc["intr"], X) ** 2 for i in range(len(k)))
                  for X in (X0, X1))
        worse += e1 > e0 * (1 + 1e-9)
        assert np.linalg.norm(X1 - sc["points_gt"][j]) < 0.05
    assert worse == 0


def test_tracks_with_one_view_are_skipped():
    sc = scenes.ring_scene(6, 24, 3)
    keep = ~((sc["obs_pt"] == 4) & (sc["obs_cam"] != sc["obs_cam"][sc["obs_pt"] == 4][0]))
    pts, err, valid = tri.triangulate_tracks(sc["poses_gt"], sc["obs_cam"][keep], sc["obs_pt"][keep], sc["obs_uv"][keep],
                                             sc["intr"], 24)
    assert not valid[4] and np.isnan(err[4]) and valid.sum() == 23
