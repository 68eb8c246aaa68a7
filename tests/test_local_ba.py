"""Sliding-window local BA over a keyframe list: window construction on the CPU,
the batched solve against per-window individual solves on the GPU."""

import numpy as np
import pytest

import scenes
from src.camera_pose import CameraPose
from src.local_ba import Keyframe, LocalBundleAdjustment, build_windows
from src.validation_error import ValidationError

K_MAT = np.array([[500.0, 0, 320.0], [0, 500.0, 240.0], [0, 0, 1.0]])


def _keyframes(sc, poses):
    out = []
    for c in range(len(poses)):
        sel = sc["obs_cam"] == c
        out.append(Keyframe(float(c), CameraPose(position=poses[c, :3], orientation=poses[c, 3:7], timestamp=float(c)),
                            K_MAT, sc["obs_uv"][sel], sc["landmark_ids"][sc["obs_pt"][sel]]))
    return out


def test_window_construction():
    sc = scenes.ring_scene(25, 100, 4)
    kfs = _keyframes(sc, sc["poses_gt"])
    # the map lacks a few of the observed ids and is not sorted
    rng = np.random.default_rng(3)
    map_ids = rng.permutation(sc["landmark_ids"][sc["landmark_ids"] % 11 != 0])
    wins = build_windows(kfs, map_ids, window=10, stride=6, min_views=2)
    assert [w.first for w in wins] == [0, 6, 12, 15]  # the last window is aligned to the end of the list
    for w in wins:
        assert list(w.keyframes) == list(range(w.first, w.first + 10))
        # brute force: ids seen by >= 2 keyframes of the window and present in the map
        seen = {}
        for k in w.keyframes:
            for i in kfs[k].points_2d_ids:
                seen[int(i)] = seen.get(int(i), 0) + 1
        expect = sorted(i for i, n in seen.items() if n >= 2 and i in set(map_ids.tolist()))
        assert list(w.landmark_ids) == expect
        assert np.array_equal(map_ids[w.landmark_rows], w.landmark_ids)
        assert len(w.obs_cam) == sum(seen[i] for i in expect)
        # every kept observation is the keyframe's pixel of that id
        for c, p, uv in zip(w.obs_cam[::17], w.obs_pt[::17], w.obs_uv[::17]):
            kf = kfs[w.keyframes[c]]
            j = int(np.flatnonzero(kf.points_2d_ids == w.landmark_ids[p])[0])
            assert np.array_equal(kf.points_2d[j], uv)
    assert build_windows(kfs[:1], map_ids) == []
    assert [w.first for w in build_windows(kfs[:7], map_ids, window=10)] == [0]
    with pytest.raises(ValidationError):
        build_windows(kfs, np.r_[map_ids, map_ids[:1]])
    with pytest.raises(ValidationError):
        build_windows(kfs, map_ids, window=1)


def test_sliding_window_refuses_an_oversized_prior_before_solving():
    """window 18, stride 2 passes the constructor (16 keyframes handed on per shift), but with 25 keyframes the last
    window is aligned to the end of the list and shifts by ONE keyframe: 17 would go into the prior, one more than it
    holds.  That must surface before any window is solved (no GPU is touched: the check runs on the window list)."""
    from src.local_ba import SlidingWindowBundleAdjustment
    sc = scenes.ring_scene(25, 100, 4)
    kfs = _keyframes(sc, sc["poses_gt"])
    assert [w.first for w in build_windows(kfs, sc["landmark_ids"], window=18, stride=2)] == [0, 2, 4, 6, 7]
    sw = SlidingWindowBundleAdjustment(window=18, stride=2)
    with pytest.raises(ValidationError, match="hands 17 keyframes"):
        sw.optimize(kfs, sc["points_gt"], sc["landmark_ids"], K_MAT)
    assert sw.last_reports == []
    with pytest.raises(ValidationError):
        SlidingWindowBundleAdjustment(window=19, stride=2)


def test_window_prior_chart_matches_oracle():
    """The host-side chart of the sliding-window prior (src/local_ba.py: _chart, WindowPrior.energy) against the oracle's
    prior_chart / prior_terms: the same coordinates and the same energy, and d(retract(ref, delta)) == delta."""
    from oracle import lm_schur_numpy as lm
    from src.local_ba import WindowPrior, _chart

    rng = np.random.default_rng(8)
    sc = scenes.ring_scene(6, 60, 3)
    ref = sc["poses_guess"][1:5]
    delta = 0.2 * rng.normal(size=(4, 6))
    poses = lm.retract(ref, delta, np.ones(4, bool))
    np.testing.assert_allclose(_chart(poses, ref), delta, rtol=0, atol=1e-13)
    np.testing.assert_allclose(_chart(poses, ref), lm.prior_chart(poses, ref)[0], rtol=0, atol=1e-14)
    A = rng.normal(size=(30, 24))
    H, g = A.T @ A, rng.normal(size=24)
    wp = WindowPrior(keyframes=np.arange(1, 5), ref=ref, H=H, g=g, c0=2.5)
    full = sc["poses_guess"].copy()
    full[1:5] = poses
    e = lm.prior_terms(full, lm.PosePrior(cams=np.arange(1, 5), ref=ref, H=H, g=g, c0=2.5))[0]
    assert abs(wp.energy(poses) - e) <= 1e-12 * abs(e)


@pytest.mark.gpu
def test_batched_windows_match_individual_solves():
    from oracle import lm_schur_numpy as lm
    from src import _native

    sc = scenes.ring_scene(20, 400, 5)
    kfs = _keyframes(sc, sc["poses_guess"])
    # min_views = 3: with two-view points the first window has a direction along which the cost is flat to
    # 12 digits (a point near the baseline of its two keyframes), so estimates are only comparable without them
    lba = LocalBundleAdjustment(window=10, stride=10, min_views=3, ftol=1e-13, max_iters=100)
    poses, pts = lba.optimize(kfs, sc["points_guess"], sc["landmark_ids"], K_MAT)
    wins = lba.last_windows
    assert [w.first for w in wins] == [0, 10] and lba.last_report["status"] != 0
    c0, c1 = lba.last_costs
    assert np.all(c1 < 0.01 * c0)
    assert [p.timestamp for p in poses] == [float(c) for c in range(20)]
    P = np.array([np.r_[p.position, p.quaternion] for p in poses])
    for b, w in enumerate(wins):
        np.testing.assert_array_equal(P[w.first], sc["poses_guess"][w.first])  # the oldest keyframe is the prior
        with _native.Solver(ftol=1e-13, max_iters=100) as s:
            s.set_problem(10, len(w.landmark_ids), w.obs_cam, w.obs_pt, w.obs_uv, sc["intr"])
            Pi, Xi, rep = s.solve(sc["poses_guess"][w.keyframes], sc["points_guess"][w.landmark_rows])
        assert abs(c1[b] - rep["cost"]) <= 1e-6 * rep["cost"], (b, c1[b], rep["cost"])
        # ... and the ORACLE's LM (numpy, exact reduced solve) on that window alone: same minimum, 1e-6 in cost
        o = lm.lm_solve(sc["poses_guess"][w.keyframes], sc["points_guess"][w.landmark_rows], w.obs_cam, w.obs_pt,
                        w.obs_uv, sc["intr"], opt=lm.LMOptions(ftol=1e-13, max_iters=100))[2]
        assert abs(c1[b] - o.cost) <= 1e-6 * o.cost, (b, c1[b], o.cost)
        assert abs(c0[b] - o.cost0) <= 1e-12 * o.cost0
        # one fixed keyframe leaves the scale about its centre free: align it before comparing
        c = Pi[0, :3]
        A, B = pts[w.landmark_rows] - c, Xi - c
        scale = np.sum(A * B) / np.sum(A * A)
        assert abs(scale - 1) < 0.05
        assert np.abs(A * scale - B).max() <= 1e-5 * np.abs(B).max()
        assert np.abs((P[w.keyframes, :3] - c) * scale - (Pi[:, :3] - c)).max() <= 1e-5 * np.abs(Pi[:, :3] - c).max()
        assert np.abs(P[w.keyframes, 3:7] - Pi[:, 3:7]).max() <= 1e-6
    untouched = np.setdiff1d(np.arange(400), np.concatenate([w.landmark_rows for w in wins]))
    np.testing.assert_array_equal(pts[untouched], sc["points_guess"][untouched])


@pytest.mark.gpu
def test_overlapping_windows_and_two_fixed_keyframes():
    sc = scenes.ring_scene(16, 320, 4)
    kfs = _keyframes(sc, sc["poses_guess"])
    lba = LocalBundleAdjustment(window=8, stride=4, n_fixed=2)
    poses, pts = lba.optimize(kfs, sc["points_guess"], sc["landmark_ids"], K_MAT)
    assert [w.first for w in lba.last_windows] == [0, 4, 8]
    c0, c1 = lba.last_costs
    assert np.all(c1 < 0.05 * c0)
    P = np.array([np.r_[p.position, p.quaternion] for p in poses])
    np.testing.assert_array_equal(P[:2], sc["poses_guess"][:2])  # fixed in the first window, in no later one
    assert np.all(np.abs(P[2:] - sc["poses_guess"][2:]).max(axis=1) > 0)
    rms = np.sqrt(2 * c1 / np.array([len(w.obs_cam) for w in lba.last_windows]))
    # pixel noise is 0.5 px; two keyframes pinned at their noisy guesses (5 cm, 0.01 rad) cost the rest
    assert np.all(rms < 3.0)
    # oracle leg: every window against the numpy LM with the same two keyframes held constant (1e-6 in cost)
    from oracle import lm_schur_numpy as lm
    for b, w in enumerate(lba.last_windows):
        free = np.ones(len(w.keyframes), bool)
        free[:2] = False
        o = lm.lm_solve(sc["poses_guess"][w.keyframes], sc["points_guess"][w.landmark_rows], w.obs_cam, w.obs_pt,
                        w.obs_uv, sc["intr"], free=free)[2]
        assert abs(c0[b] - o.cost0) <= 1e-12 * o.cost0
        assert abs(c1[b] - o.cost) <= 1e-6 * o.cost, (b, c1[b], o.cost)
