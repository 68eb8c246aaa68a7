"""Adapters on the input side of the path: the reference's second container
family (src/feature_observation.py:292-483) and BAL text files.  CPU only, except the last test."""

import numpy as np
import pytest

import scenes
from src import adapters as ad
from src.feature_observation import Features2D, ImageFeatures
from src.validation_error import ValidationError


def _image_features(g):
    C = len(g["poses_gt"])
    return [ImageFeatures(float(c), 0, 640, 480, Features2D(g["obs_uv"][g["obs_cam"] == c],
                                                           g["landmark_ids"][g["obs_pt"][g["obs_cam"] == c]]))
            for c in range(C)]


def test_container_semantics_follow_the_reference():
    tr = ad.FeatureTrack(3)
    coords = np.array([1.0, 2.0])
    ob = ad.FeatureObservation(3, coords, camera_id=0, timestamp=0.5)
    coords[0] = 9.0
    assert ob.image_coords[0] == 1.0  # the constructor copies (feature_observation.py:311)
    tr.add_observation(ob)
    with pytest.raises(ValueError):  # :334-337
        tr.add_observation(ad.FeatureObservation(4, coords, 0))
    assert tr.length() == 1 and tr.get_timestamps() == [0.5]
    ts = ad.Tracks()
    ts.add_track(tr)
    assert len(ts) == 1 and ts.get_track(3) is tr and ts.get_track(7) is None and ts.get_all_tracks() == [tr]
    assert ts.get_observations_for_camera(0) == [ob] and ts.get_observations_for_camera(1) == []
    io = ad.ImageObservations(camera_id=0, timestamp=0.5)
    assert (io.image_width, io.image_height) == (640, 480) and io.get_features() == []
    io.add_feature(ob)
    assert io.get_visible_landmarks() == [3] and io.get_observations_for_landmark(3) is ob
    assert io.get_observations_for_landmark(8) is None


def test_tracks_round_trip_and_packing(golden):
    g = golden("noisy_10x300")
    feats = _image_features(g)
    tracks = ad.tracks_from_image_features(feats)
    assert len(tracks) == 300 and all(t.length() == 10 for t in tracks.get_all_tracks())
    back = ad.image_features_from_tracks(tracks, [f.get_timestamp() for f in feats])
    for a, b in zip(feats, back):
        assert np.array_equal(a.get_points_2d().get_ids(), b.get_points_2d().get_ids())
        assert np.array_equal(a.get_points_2d().get_points_2d(), b.get_points_2d().get_points_2d())
    cam, pt, uv = ad.pack_image_features(back, g["landmark_ids"])
    assert np.array_equal(cam, g["obs_cam"]) and np.array_equal(pt, g["obs_pt"]) and np.array_equal(uv, g["obs_uv"])
    # unsorted observations inside a frame come out id-sorted; short tracks are dropped on request
    fr = ad.ImageObservations(0, 0.0, [ad.FeatureObservation(5, np.array([1.0, 1.0]), 0),
                                       ad.FeatureObservation(2, np.array([2.0, 2.0]), 0)])
    f = ad.image_features_from_observations([fr])[0]
    assert list(f.get_points_2d().get_ids()) == [2, 5] and f.get_points_2d().get_points_2d()[0, 0] == 2.0
    with pytest.raises(ValidationError):
        ad.image_features_from_observations([ad.ImageObservations(0, 0.0, [fr.feature_observations[0]] * 2)])
    with pytest.raises(ValidationError):
        ad.pack_image_features(back, g["landmark_ids"][:-1])


def test_bal_round_trip(tmp_path):
    sc = scenes.ring_scene(6, 24, 3)
    path = str(tmp_path / "problem.txt")
    ad.write_bal(path, sc["poses_gt"], sc["points_gt"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    b = ad.read_bal(path)
    assert np.array_equal(b["obs_cam"], sc["obs_cam"]) and np.array_equal(b["obs_pt"], sc["obs_pt"])
    np.testing.assert_allclose(b["points"], sc["points_gt"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(b["poses"][:, :3], sc["poses_gt"][:, :3], rtol=0, atol=1e-13)
    assert np.abs(np.abs(np.sum(b["poses"][:, 3:7] * sc["poses_gt"][:, 3:7], axis=1)) - 1).max() < 1e-14
    np.testing.assert_allclose(b["obs_uv"] + sc["intr"][2:4], sc["obs_uv"], rtol=0, atol=1e-12)
    assert list(b["intr"]) == [500.0, 500.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    # BAL's own projection of its own cameras reproduces the pixels it stores
    cam = b["bal_cameras"][2]
    k = np.flatnonzero(b["obs_cam"] == 2)
    P = (ad._rodrigues(cam[:3]) @ b["points"][b["obs_pt"][k]].T).T + cam[3:6]
    p = -P[:, :2] / P[:, 2:3]
    n2 = np.sum(p * p, axis=1, keepdims=True)
    px = cam[6] * (1 + cam[7] * n2 + cam[8] * n2**2) * p
    assert np.abs(px - b["obs_uv"][k]).max() < 2.5  # sigma 0.5 px noise on the stored pixels
    # malformed files
    bad = tmp_path / "bad.txt"
    bad.write_text("2 1 1\n0 0 1.0 2.0\n")
    with pytest.raises(ValidationError):
        ad.read_bal(str(bad))
    cams2 = b["bal_cameras"].copy()
    cams2[1, 6] *= 1.1
    lines = open(path).read().split("\n")
    n = len(sc["obs_cam"])
    body = lines[: 1 + n] + [repr(float(v)) for v in cams2.reshape(-1)] + [repr(float(v)) for v in b["points"].reshape(-1)]
    bad.write_text("\n".join(body) + "\n")
    with pytest.raises(ValidationError):
        ad.read_bal(str(bad))
    assert ad.read_bal(str(bad), intrinsics=[500, 500, 0, 0, 0, 0, 0])["intr"][0] == 500
    with pytest.raises(ValidationError):
        ad.write_bal(path, sc["poses_gt"], sc["points_gt"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"],
                     [500, 400, 0, 0, 0, 0, 0])


@pytest.mark.gpu
def test_bal_problem_solves_like_the_original(tmp_path):
    from src import _native

    sc = scenes.ring_scene(8, 160, 4)
    path = str(tmp_path / "ring.txt")
    ad.write_bal(path, sc["poses_guess"], sc["points_guess"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    b = ad.read_bal(path)
    costs = []
    for prob, poses, pts in ((sc, sc["poses_guess"], sc["points_guess"]), (b, b["poses"], b["points"])):
        with _native.Solver() as s:
            s.set_problem(8, 160, prob["obs_cam"], prob["obs_pt"], prob["obs_uv"], prob["intr"])
            costs.append(s.solve(poses, pts)[2])
    assert abs(costs[0]["cost0"] - costs[1]["cost0"]) <= 1e-9 * costs[0]["cost0"]
    assert abs(costs[0]["cost"] - costs[1]["cost"]) <= 1e-8 * costs[0]["cost"]


@pytest.mark.gpu
def test_adapter_fed_solves_match_the_oracle(golden, tmp_path):
    """Oracle legs for the adapters: the scene goes in through the second container family (ImageObservations ->
    ImageFeatures -> BundleAdjustment.optimize) and through a BAL file (write_bal -> read_bal -> the C ABI); both
    converged costs against the polished minimum the fixture recorded from the reference path and against the numpy
    LM run on the arrays the adapter produced (1e-6 relative)."""
    from oracle import lm_schur_numpy as lm
    from src import _native
    from src.bundle_adjustment import BundleAdjustment
    from src.camera import PinHoleCamera
    from src.camera_pose import CameraPose
    from src.feature_observation import Landmarks3D

    g = golden("noisy_10x300")
    n_poses = len(g["poses_gt"]) - 1
    poses = np.vstack([g["pose0"][None], g["x0"][: 7 * n_poses].reshape(-1, 7)])
    pts = g["x0"][7 * n_poses:].reshape(-1, 3)
    pol = float(g["polished_cost"])
    # (1) frames of FeatureObservation objects, shuffled inside every frame
    rng = np.random.default_rng(1)
    frames = []
    for c in range(n_poses + 1):
        sel = np.flatnonzero(g["obs_cam"] == c)
        rng.shuffle(sel)
        frames.append(ad.ImageObservations(c, float(c), [ad.FeatureObservation(int(g["landmark_ids"][g["obs_pt"][k]]),
                                                                                g["obs_uv"][k], c, float(c)) for k in sel]))
    feats = ad.image_features_from_observations(frames)
    cam = PinHoleCamera.from_vector(np.array([float(v) for v in g["intr"]]))
    ba = BundleAdjustment()
    _, lms, out_poses = ba.optimize(feats, CameraPose(poses[0, :3], poses[0, 3:7], 0.0), cam,
                                    [CameraPose(p[:3], p[3:7], float(i + 1)) for i, p in enumerate(poses[1:])],
                                    Landmarks3D(pts, g["landmark_ids"]))
    assert abs(ba.last_report["cost"] - pol) <= 1e-6 * pol
    oc, op, uv = ad.pack_image_features(feats, g["landmark_ids"])
    o = lm.lm_solve(poses, pts, oc, op, uv, g["intr"])[2]
    assert abs(ba.last_report["cost"] - o.cost) <= 1e-6 * o.cost
    # (2) a BAL file of the same problem (BAL's camera convention: the adapter converts poses and pixels both ways)
    path = str(tmp_path / "noisy.txt")
    ad.write_bal(path, poses, pts, g["obs_cam"], g["obs_pt"], g["obs_uv"], g["intr"])
    b = ad.read_bal(path)
    with _native.Solver() as s:
        s.set_problem(n_poses + 1, len(pts), b["obs_cam"], b["obs_pt"], b["obs_uv"], b["intr"])
        rep = s.solve(b["poses"], b["points"])[2]
    ob = lm.lm_solve(b["poses"], b["points"], b["obs_cam"], b["obs_pt"], b["obs_uv"], b["intr"])[2]
    assert abs(rep["cost0"] - ob.cost0) <= 1e-9 * ob.cost0
    assert abs(rep["cost"] - ob.cost) <= 1e-6 * ob.cost and abs(rep["cost"] - pol) <= 1e-6 * pol
