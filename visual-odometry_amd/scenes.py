"""Synthetic bundle-adjustment scenes (the build's own generator).

Every scene is returned *packed* -- plain arrays in the layout the C ABI takes -- plus
``to_containers`` to turn it into the reference-style argument objects of
``BundleAdjustment.optimize``.  Scenes:

* ``reference_test_scene``  -- the fixture of reference tests/test_bundle_adjustment.py:27-165
  (4 poses x 50 landmarks, noise-free pixels, np.random.seed(42) guess noise).
* ``demo_scene``            -- reference src/run_bundle_adjustement.py:29-174 (8 poses x 100
  landmarks; BASELINE config 1).
* ``noisy_scene``           -- P poses x L landmarks, every landmark in every image, sigma_px = 0.5
  (SURVEY.md Appendix C ``noisy_PxL``).
* ``ring_scene``            -- C cameras on a circle looking at the centre, landmark j seen by the
  d cameras (j+k) mod C, sigma_px = 0.5 (SURVEY.md Appendix C ``ring_CxL``; BASELINE configs 2-4).
* ``local_window_scene``    -- one 10-keyframe sliding-window problem (BASELINE config 5).

Draw orders follow SURVEY.md Appendix C so that the reference, imported in the build container
(oracle/make_golden.py), produces the same scenes; tests/golden/*.npz pin that.
All frames are expressed in camera 0's frame; camera 0 is the fixed pose.
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from src.camera import PinHoleCamera
from src.camera_pose import CameraPose
from src.feature_observation import Features2D, ImageFeatures, Landmarks3D
from src.geometry import GeometryUtils as G

Scene = Dict[str, np.ndarray]


# ----------------------------------------------------------------------------- vector helpers
def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Hamilton product on (..., 4) arrays, same term order as GeometryUtils.quaternion_multiply."""
    a0, a1, a2, a3 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    b0, b1, b2, b3 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack(
        [
            a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
            a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
            a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
            a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0,
        ],
        axis=-1,
    )


def quat_to_rot(q: np.ndarray) -> np.ndarray:
    """(N, 4) unit quaternions -> (N, 3, 3) camera->world rotations (geometry.py:192-213 formula)."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * y * y - 2 * z * z
    R[:, 0, 1] = 2 * x * y - 2 * w * z
    R[:, 0, 2] = 2 * x * z + 2 * w * y
    R[:, 1, 0] = 2 * x * y + 2 * w * z
    R[:, 1, 1] = 1 - 2 * x * x - 2 * z * z
    R[:, 1, 2] = 2 * y * z - 2 * w * x
    R[:, 2, 0] = 2 * x * z - 2 * w * y
    R[:, 2, 1] = 2 * y * z + 2 * w * x
    R[:, 2, 2] = 1 - 2 * x * x - 2 * y * y
    return R


def _to_camera0_frame(poses_w: np.ndarray, pts_w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Express world poses (C,7) and points (L,3) in the frame of pose 0 (convert_to_new_frame)."""
    p0 = CameraPose(poses_w[0, :3].copy(), poses_w[0, 3:7].copy())
    R0 = p0.rotation_matrix
    pts = (pts_w - poses_w[0, :3].reshape(1, 3)) @ R0
    pos = (poses_w[:, :3] - poses_w[0, :3].reshape(1, 3)) @ R0
    qinv = G.quaternion_inverse(p0.quaternion)
    quat = quat_mul(np.broadcast_to(qinv, (len(poses_w), 4)), poses_w[:, 3:7])
    return np.hstack([pos, quat]), pts


def _project(points_cam: np.ndarray, intr: np.ndarray) -> np.ndarray:
    return PinHoleCamera.from_vector(np.array([float(v) for v in intr])).project(points_cam)


def _pack(intr, poses_gt, pts_gt, ids, obs_cam, obs_pt, obs_uv, poses_guess, pts_guess, name) -> Scene:
    C = len(poses_gt)
    x0 = np.concatenate([poses_guess[1:].reshape(-1), pts_guess.reshape(-1)])
    return {
        "name": np.array(name),
        "intr": np.asarray(intr, dtype=np.float64),
        "pose0": poses_gt[0].copy(),
        "poses_gt": poses_gt,
        "points_gt": pts_gt,
        "landmark_ids": np.asarray(ids, dtype=np.int64),
        "obs_cam": np.asarray(obs_cam, dtype=np.int32),
        "obs_pt": np.asarray(obs_pt, dtype=np.int32),
        "obs_uv": np.ascontiguousarray(obs_uv, dtype=np.float64),
        "poses_guess": np.vstack([poses_gt[:1], poses_guess[1:]]),
        "points_guess": pts_guess,
        "x0": x0,
        "num_poses": np.array(C - 1),
    }


# ----------------------------------------------------------------------------- reference scenes
_QUAT_CAM_EULER = np.array([-np.pi / 2, 0.0, -np.pi / 2])

_TEST_POSES = [  # reference tests/test_bundle_adjustment.py:40-81 (position, euler offset, timestamp)
    ([-0.5, 0.1, 0.0], [0.02, 0.03, 0.0], 0.0),
    ([0.0, -0.2, 0.0], [0.04, -0.1, 0.2], 1.0),
    ([0.0, 0.0, 0.3], [0.06, 0.02, 0.4], 2.0),
    ([0.0, 0.0, -0.3], [0.08, 0.3, 0.7], 3.0),
]
_DEMO_EXTRA_POSES = [  # reference src/run_bundle_adjustement.py:73-104
    ([0.0, 0.0, 0.0], [0.1, 0.05, -0.2], 0.0),
    ([0.0, -0.2, 0.0], [0.12, 0.0, 0.0], 1.0),
    ([0.0, 0.0, 0.3], [0.14, 0.0, 0.0], 2.0),
    ([0.0, 0.0, -0.3], [0.16, 0.0, 0.0], 3.0),
]


def _box_landmarks(n: int, seed: int) -> np.ndarray:
    from src.landmarks import LandmarkGenerator

    return LandmarkGenerator((2.8, 3.2), (-0.5, 0.5), (-0.5, 0.5)).generate_random(num_landmarks=n, seed=seed)


def _listed_poses(spec) -> Tuple[np.ndarray, np.ndarray]:
    quat_cam = G.quaternion_from_euler_angles(_QUAT_CAM_EULER)
    rows, stamps = [], []
    for pos, eul, ts in spec:
        q = G.quaternion_multiply(quat_cam, G.quaternion_from_euler_angles(np.array(eul)))
        rows.append(np.concatenate([np.array(pos, dtype=float), q]))
        stamps.append(ts)
    return np.array(rows), np.array(stamps)


def _all_visible_obs(poses_w, pts_w, intr):
    C, L = len(poses_w), len(pts_w)
    uv = []
    for c in range(C):
        pose = CameraPose(poses_w[c, :3].copy(), poses_w[c, 3:7].copy())
        uv.append(_project(pose.transform_points_world_to_camera(pts_w), intr))
    obs_cam = np.repeat(np.arange(C), L)
    obs_pt = np.tile(np.arange(L), C)
    return obs_cam, obs_pt, np.vstack(uv)


def _global_seed_guess(poses_cf0, pts_cf0, lm_std, pos_std, rot_std):
    """Guess noise drawn from the *global* numpy stream after np.random.seed(42), in the
    reference's order: landmarks first, then per free pose position(3) then euler(3)."""
    np.random.seed(42)
    pts_guess = pts_cf0 + np.random.randn(len(pts_cf0), 3) * lm_std
    poses_guess = poses_cf0.copy()
    for i in range(1, len(poses_cf0)):
        poses_guess[i, :3] = poses_cf0[i, :3] + np.random.randn(3) * pos_std
        poses_guess[i, 3:7] = G.quaternion_multiply(
            poses_cf0[i, 3:7], G.quaternion_from_euler_angles(np.random.randn(3) * rot_std)
        )
    return poses_guess, pts_guess


def reference_test_scene(lm_std: float = 0.2, pos_std: float = 0.1, rot_std: float = 0.05) -> Scene:
    """4 x 50 scene of the reference test; default noise = test_bundle_adjustment_convergence,
    (0.3, 0.05, 0.02) = test_bundle_adjustment_landmark_accuracy."""
    intr = np.array([1000.0, 1000.0, 320.0, 240.0, 0.0, 0.0, 0.0])
    pts_w = _box_landmarks(50, 42)
    poses_w, stamps = _listed_poses(_TEST_POSES)
    obs_cam, obs_pt, uv = _all_visible_obs(poses_w, pts_w, intr)
    poses_cf0, pts_cf0 = _to_camera0_frame(poses_w, pts_w)
    poses_guess, pts_guess = _global_seed_guess(poses_cf0, pts_cf0, lm_std, pos_std, rot_std)
    s = _pack(intr, poses_cf0, pts_cf0, np.arange(50), obs_cam, obs_pt, uv, poses_guess, pts_guess, "test_scene")
    s["timestamps"] = stamps
    return s


def demo_scene() -> Scene:
    """BASELINE config 1: the run_bundle_adjustement.py scene (noise 0.5 / 0.5 / 0.1)."""
    intr = np.array([1000.0, 1000.0, 320.0, 240.0, 0.0, 0.0, 0.0])
    pts_w = _box_landmarks(100, 42)
    poses_w, stamps = _listed_poses(_TEST_POSES + _DEMO_EXTRA_POSES)
    obs_cam, obs_pt, uv = _all_visible_obs(poses_w, pts_w, intr)
    poses_cf0, pts_cf0 = _to_camera0_frame(poses_w, pts_w)
    poses_guess, pts_guess = _global_seed_guess(poses_cf0, pts_cf0, 0.5, 0.5, 0.1)
    s = _pack(intr, poses_cf0, pts_cf0, np.arange(100), obs_cam, obs_pt, uv, poses_guess, pts_guess, "demo_scene")
    s["timestamps"] = stamps
    return s


def noisy_scene(P: int = 10, L: int = 300, sigma_px: float = 0.5) -> Scene:
    """SURVEY.md Appendix C ``noisy_PxL`` (rng = default_rng(0))."""
    rng = np.random.default_rng(0)
    intr = np.array([1000.0, 1000.0, 320.0, 240.0, 0.0, 0.0, 0.0])
    pts_w = _box_landmarks(L, 0)
    quat_cam = G.quaternion_from_euler_angles(_QUAT_CAM_EULER)
    rows = []
    for _ in range(P):
        p = rng.uniform(-0.4, 0.4, 3)
        e = rng.uniform(-0.1, 0.1, 3)
        rows.append(np.concatenate([p, G.quaternion_multiply(quat_cam, G.quaternion_from_euler_angles(e))]))
    poses_w = np.array(rows)
    obs_cam, obs_pt, uv = _all_visible_obs(poses_w, pts_w, intr)
    for c in range(P):
        uv[c * L:(c + 1) * L] += rng.standard_normal((L, 2)) * sigma_px
    poses_cf0, pts_cf0 = _to_camera0_frame(poses_w, pts_w)
    pts_guess = pts_cf0 + rng.standard_normal((L, 3)) * 0.1
    poses_guess = poses_cf0.copy()
    for i in range(1, P):
        poses_guess[i, :3] = poses_cf0[i, :3] + rng.standard_normal(3) * 0.05
        poses_guess[i, 3:7] = G.quaternion_multiply(
            poses_cf0[i, 3:7], G.quaternion_from_euler_angles(rng.standard_normal(3) * 0.02)
        )
    s = _pack(intr, poses_cf0, pts_cf0, np.arange(L), obs_cam, obs_pt, uv, poses_guess, pts_guess,
              f"noisy_{P}x{L}")
    s["timestamps"] = np.arange(P, dtype=float)
    return s


# ----------------------------------------------------------------------------- ring scenes (configs 2-4)
def _euler_to_quat_batch(e: np.ndarray) -> np.ndarray:
    """(N,3) [roll,pitch,yaw] -> (N,4); small angles only (trace > 0 branch of
    quaternion_from_rotation_matrix), same arithmetic as the scalar path."""
    roll, pitch, yaw = e[:, 0], e[:, 1], e[:, 2]
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    n = len(e)
    rx = np.zeros((n, 3, 3)); ry = np.zeros((n, 3, 3)); rz = np.zeros((n, 3, 3))
    rx[:, 0, 0] = 1; rx[:, 1, 1] = cr; rx[:, 1, 2] = -sr; rx[:, 2, 1] = sr; rx[:, 2, 2] = cr
    ry[:, 0, 0] = cp; ry[:, 0, 2] = sp; ry[:, 1, 1] = 1; ry[:, 2, 0] = -sp; ry[:, 2, 2] = cp
    rz[:, 0, 0] = cy; rz[:, 0, 1] = -sy; rz[:, 1, 0] = sy; rz[:, 1, 1] = cy; rz[:, 2, 2] = 1
    m = rz @ ry @ rx
    tr = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    assert np.all(tr > 0)
    s = 0.5 / np.sqrt(tr + 1.0)
    return np.stack([0.25 / s, (m[:, 2, 1] - m[:, 1, 2]) * s, (m[:, 0, 2] - m[:, 2, 0]) * s,
                     (m[:, 1, 0] - m[:, 0, 1]) * s], axis=1)


def ring_poses(C: int, radius: float = 6.0, height: float = 1.0, spacing_of: int = 0) -> np.ndarray:
    """World poses of the ring trajectory: reference camera_pose.py:282-367 with time_step 0.1,
    angular_velocity 2*pi/(C*0.1), look_at_center, offset = look-straight quaternion.
    ``spacing_of`` = N > 0 puts the C cameras on an ARC with the camera spacing of an N-camera ring
    (down-scaled scenes with the geometry of a much larger ring, e.g. the 3.8 mm baselines of config 4)."""
    ts = 0.1
    av = 2 * np.pi / ((spacing_of or C) * 0.1)
    offset = CameraPose.create_look_straight_pose().quaternion
    i = np.arange(C)
    angle = 0.0 + i * av * ts
    x = 0.0 + radius * np.cos(angle)
    y = 0.0 + radius * np.sin(angle)
    ld = np.stack([0.0 - x, 0.0 - y, np.zeros(C)], axis=1)
    ld = ld / np.linalg.norm(ld, axis=1, keepdims=True)
    yaw = np.arctan2(ld[:, 1], ld[:, 0])
    half = yaw / 2.0
    qz = np.stack([np.cos(half), np.zeros(C), np.zeros(C), 1.0 * np.sin(half)], axis=1)
    q = quat_mul(qz, np.broadcast_to(offset, (C, 4)))
    return np.hstack([np.stack([x, y, np.full(C, height)], axis=1), q])


def ring_observation_index(C: int, L: int, d: int) -> Tuple[np.ndarray, np.ndarray]:
    """Image-major flat index: camera c sees the ascending ids {j : (j+k) mod C == c, k < d}."""
    if L % C != 0:
        raise ValueError("ring scenes need C | L so every image has the same observation count")
    j = np.arange(L)
    cams = (j[:, None] + np.arange(d)[None, :]) % C  # (L, d)
    flat_cam = cams.reshape(-1)
    flat_pt = np.repeat(j, d)
    order = np.lexsort((flat_pt, flat_cam))  # by camera, then ascending landmark id
    return flat_cam[order].astype(np.int32), flat_pt[order].astype(np.int32)


def random_observation_index(C: int, L: int, d: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """Unstructured visibility: landmark j is seen by d cameras drawn at random (no two landmarks need
    share a camera set); image-major, ascending landmark id inside an image like the ring index."""
    rng = np.random.default_rng(seed)
    cams = np.argsort(rng.random((L, C)), axis=1)[:, :d]  # d distinct cameras per landmark
    flat_cam = cams.reshape(-1)
    flat_pt = np.repeat(np.arange(L), d)
    order = np.lexsort((flat_pt, flat_cam))
    return flat_cam[order].astype(np.int32), flat_pt[order].astype(np.int32)


def ring_scene(C: int, L: int, d: int, sigma_px: float = 0.5, seed: int = 0, f: float = 500.0,
               visibility: str = "ring", spacing_of: int = 0) -> Scene:
    """SURVEY.md Appendix C ``ring_CxL``: cfg2 = (50, 20000, 10), cfg3 = (1000, 200000, 5),
    cfg4 = (10000, 2000000, 5).  Fully vectorised (no per-landmark Python loops).  ``visibility="random"``
    keeps the geometry but lets every landmark be seen by d random cameras (images then have unequal
    observation counts: a stress case for the solver, not a reference-compatible scene)."""
    rng = np.random.default_rng(seed)
    intr = np.array([f, f, 320.0, 240.0, 0.0, 0.0, 0.0])
    poses_w = ring_poses(C, spacing_of=spacing_of)
    r = 1.5 * np.sqrt(rng.uniform(0, 1, L))
    th = rng.uniform(0, 2 * np.pi, L)
    z = rng.uniform(0, 2, L)
    pts_w = np.stack([r * np.cos(th), r * np.sin(th), z], axis=1)
    obs_cam, obs_pt = (ring_observation_index(C, L, d) if visibility == "ring"
                       else random_observation_index(C, L, d, seed + 1000))
    R = quat_to_rot(poses_w[:, 3:7])
    dd = pts_w[obs_pt] - poses_w[obs_cam, :3]
    Rc = R[obs_cam]
    xc = dd[:, 0:1] * Rc[:, 0, :] + dd[:, 1:2] * Rc[:, 1, :] + dd[:, 2:3] * Rc[:, 2, :]
    assert np.all(xc[:, 2] > 0), "ring scene: a landmark fell behind a camera"
    uv = _project(xc, intr) + rng.standard_normal((len(obs_cam), 2)) * sigma_px  # per camera in order
    poses_cf0, pts_cf0 = _to_camera0_frame(poses_w, pts_w)
    pts_guess = pts_cf0 + rng.standard_normal((L, 3)) * 0.05
    noise = rng.standard_normal((C - 1, 6))  # per pose: position(3) then euler(3)
    poses_guess = poses_cf0.copy()
    poses_guess[1:, :3] = poses_cf0[1:, :3] + noise[:, :3] * 0.05
    poses_guess[1:, 3:7] = quat_mul(poses_cf0[1:, 3:7], _euler_to_quat_batch(noise[:, 3:] * 0.01))
    s = _pack(intr, poses_cf0, pts_cf0, np.arange(L), obs_cam, obs_pt, uv, poses_guess, pts_guess,
              f"ring_{C}x{L}_d{d}")
    s["timestamps"] = np.arange(C) * 0.1
    return s


# ----------------------------------------------------------------------------- config 5 windows
def local_window_scene(seed: int, K: int = 10, L: int = 200, sigma_px: float = 0.5) -> Scene:
    """One sliding-window local BA: K keyframes on a short arc, L landmarks all co-visible
    (BASELINE config 5; seeds = problem index)."""
    rng = np.random.default_rng(seed)
    intr = np.array([500.0, 500.0, 320.0, 240.0, 0.0, 0.0, 0.0])
    quat_cam = G.quaternion_from_euler_angles(_QUAT_CAM_EULER)
    ang = np.linspace(-0.15, 0.15, K)
    pos = np.stack([2.0 * (1 - np.cos(ang)), 2.0 * np.sin(ang), 0.02 * rng.standard_normal(K)], axis=1)
    eul = np.stack([0.01 * rng.standard_normal(K), 0.01 * rng.standard_normal(K), -ang * 0.5], axis=1)
    poses_w = np.hstack([pos, quat_mul(np.broadcast_to(quat_cam, (K, 4)), _euler_to_quat_batch(eul))])
    pts_w = np.stack([rng.uniform(3.0, 6.0, L), rng.uniform(-1.5, 1.5, L), rng.uniform(-1.0, 1.0, L)], axis=1)
    obs_cam, obs_pt, uv = _all_visible_obs(poses_w, pts_w, intr)
    uv = uv + rng.standard_normal(uv.shape) * sigma_px
    poses_cf0, pts_cf0 = _to_camera0_frame(poses_w, pts_w)
    pts_guess = pts_cf0 + rng.standard_normal((L, 3)) * 0.05
    noise = rng.standard_normal((K - 1, 6))
    poses_guess = poses_cf0.copy()
    poses_guess[1:, :3] += noise[:, :3] * 0.02
    poses_guess[1:, 3:7] = quat_mul(poses_cf0[1:, 3:7], _euler_to_quat_batch(noise[:, 3:] * 0.01))
    s = _pack(intr, poses_cf0, pts_cf0, np.arange(L), obs_cam, obs_pt, uv, poses_guess, pts_guess,
              f"window_{seed}_{K}x{L}")
    s["timestamps"] = np.arange(K) * 0.1
    return s


# ----------------------------------------------------------------------------- container view
def to_containers(scene: Scene):
    """Packed scene -> (image_features_list, camera_pose_initial, camera_model, camera_poses_guess,
    points3d_guess): the positional arguments of BundleAdjustment.optimize."""
    intr = scene["intr"]
    cam = PinHoleCamera.from_vector(np.array([float(v) for v in intr]))
    C = len(scene["poses_gt"])
    ids = scene["landmark_ids"]
    stamps = scene.get("timestamps", np.arange(C, dtype=float))
    feats: List[ImageFeatures] = []
    for c in range(C):
        sel = scene["obs_cam"] == c
        feats.append(ImageFeatures(float(stamps[c]), 0, 640, 480,
                                   Features2D(scene["obs_uv"][sel], ids[scene["obs_pt"][sel]])))
    pg = scene["poses_guess"]
    pose0 = CameraPose(pg[0, :3].copy(), pg[0, 3:7].copy(), float(stamps[0]))
    guesses = []
    for c in range(1, C):
        guesses.append(CameraPose(pg[c, :3].copy(), pg[c, 3:7].copy(), float(stamps[c])))
    return feats, pose0, cam, guesses, Landmarks3D(scene["points_guess"], ids)
