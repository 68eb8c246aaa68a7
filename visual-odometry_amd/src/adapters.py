"""Adapters on the input side of the bundle-adjustment path.

* The reference's second family of observation containers -- ``FeatureObservation`` /
  ``FeatureTrack`` / ``Tracks`` / ``ImageObservations`` (src/feature_observation.py:292-483), which its
  renderer and simulator produce (image_renderer.py:137-177) -- with the same attribute and method
  names, and converters to the ``ImageFeatures`` list that ``BundleAdjustment.optimize`` takes and to
  the flat index arrays of the C ABI.
* Bundle-Adjustment-in-the-Large text files (``read_bal`` / ``write_bal``) for external problems.  A BAL
  camera is (Rodrigues r, t, f, k1, k2) with P = R X + t, p = -P / P.z, pixel = f (1 + k1 |p|^2 + k2 |p|^4) p;
  the reference's camera looks along +z and has ONE set of intrinsics, so a file is accepted when all its
  cameras share f, k1, k2 (or when the caller supplies the intrinsics) and is turned half a turn about
  the optical axis: pose = [c, q] with R(q) = (D R)^T, c = -(D R)^T D t, D = diag(-1, -1, 1).
"""

from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .feature_observation import Features2D, ImageFeatures
from .geometry import GeometryUtils
from .validation_error import ValidationError


class FeatureObservation:
    def __init__(self, landmark_id: int, image_coords: np.ndarray, camera_id: int, timestamp: float = 0.0):
        self.landmark_id = landmark_id
        self.image_coords = np.array(image_coords, dtype=float)
        self.camera_id = camera_id
        self.timestamp = timestamp


class FeatureTrack:
    def __init__(self, landmark_id: int):
        self.landmark_id = landmark_id
        self.observations: List[FeatureObservation] = []

    def add_observation(self, observation: FeatureObservation) -> None:
        if observation.landmark_id != self.landmark_id:
            raise ValueError(f"Observation landmark_id ({observation.landmark_id}) does not match track "
                             f"landmark_id ({self.landmark_id})")
        self.observations.append(observation)

    def length(self) -> int:
        return len(self.observations)

    def get_timestamps(self) -> List[float]:
        return [o.timestamp for o in self.observations]


class Tracks:
    def __init__(self) -> None:
        self.tracks: Dict[int, FeatureTrack] = {}
        self.camera_ids: List[int] = []
        self.timestamps: List[float] = []

    def add_track(self, track: FeatureTrack) -> None:
        self.tracks[track.landmark_id] = track

    def get_track(self, landmark_id: int) -> Optional[FeatureTrack]:
        return self.tracks.get(landmark_id)

    def get_all_tracks(self) -> List[FeatureTrack]:
        return list(self.tracks.values())

    def get_observations_for_camera(self, camera_id: int) -> List[FeatureObservation]:
        return [o for t in self.tracks.values() for o in t.observations if o.camera_id == camera_id]

    def __len__(self) -> int:
        return len(self.tracks)


class ImageObservations:
    def __init__(self, camera_id: int, timestamp: float = 0.0,
                 feature_observations: Optional[List[FeatureObservation]] = None, image_width: int = 640,
                 image_height: int = 480):
        self.camera_id = camera_id
        self.timestamp = timestamp
        self.feature_observations = feature_observations if feature_observations is not None else []
        self.image_width = image_width
        self.image_height = image_height

    def get_visible_landmarks(self) -> List[int]:
        return [o.landmark_id for o in self.feature_observations]

    def get_observations_for_landmark(self, landmark_id: int) -> Optional[FeatureObservation]:
        for o in self.feature_observations:
            if o.landmark_id == landmark_id:
                return o
        return None

    def add_feature(self, feature: FeatureObservation) -> None:
        self.feature_observations.append(feature)

    def get_features(self) -> List[FeatureObservation]:
        return self.feature_observations


# ----------------------------------------------------------------------------------------------- converters
def image_features_from_observations(frames: Sequence[ImageObservations]) -> List[ImageFeatures]:
    """One ImageFeatures per ImageObservations, ids ascending inside an image (the order on which the
    reference's residual_function is self-consistent)."""
    out = []
    for fr in frames:
        ids = np.array([o.landmark_id for o in fr.feature_observations], dtype=int)
        uv = np.array([o.image_coords for o in fr.feature_observations], dtype=float).reshape(-1, 2)
        if len(np.unique(ids)) != len(ids):
            raise ValidationError(f"image at t={fr.timestamp} observes a landmark twice")
        order = np.argsort(ids, kind="stable")
        out.append(ImageFeatures(fr.timestamp, fr.camera_id, fr.image_width, fr.image_height,
                                 Features2D(uv[order], ids[order])))
    return out


def image_features_from_tracks(tracks: Tracks, timestamps: Sequence[float], min_track_length: int = 2,
                               image_width: int = 640, image_height: int = 480) -> List[ImageFeatures]:
    """Group the observations of all tracks with >= min_track_length observations by frame: frame i is the
    one whose timestamp is timestamps[i] (the renderer stamps every observation with its frame's time,
    image_renderer.py:137-177)."""
    frame_of = {float(t): i for i, t in enumerate(timestamps)}
    per_frame: List[List[Tuple[int, np.ndarray]]] = [[] for _ in timestamps]
    cam_of = [0] * len(timestamps)
    for tr in tracks.get_all_tracks():
        if tr.length() < min_track_length:
            continue
        for o in tr.observations:
            if float(o.timestamp) not in frame_of:
                raise ValidationError(f"observation of landmark {o.landmark_id} at t={o.timestamp} belongs to no frame")
            i = frame_of[float(o.timestamp)]
            per_frame[i].append((o.landmark_id, o.image_coords))
            cam_of[i] = o.camera_id
    frames = [ImageObservations(cam_of[i], float(t), [FeatureObservation(l, uv, cam_of[i], float(t)) for l, uv in obs],
                                image_width, image_height) for i, (t, obs) in enumerate(zip(timestamps, per_frame))]
    return image_features_from_observations(frames)


def tracks_from_image_features(image_features_list: Iterable[ImageFeatures]) -> Tracks:
    tracks = Tracks()
    for f in image_features_list:
        tracks.camera_ids.append(f.get_camera_id())
        tracks.timestamps.append(f.get_timestamp())
        f2 = f.get_points_2d()
        for lid, uv in zip(f2.get_ids(), f2.get_points_2d()):
            tr = tracks.get_track(int(lid))
            if tr is None:
                tr = FeatureTrack(int(lid))
                tracks.add_track(tr)
            tr.add_observation(FeatureObservation(int(lid), uv, f.get_camera_id(), f.get_timestamp()))
    return tracks


def pack_image_features(image_features_list: Sequence[ImageFeatures], landmark_ids: np.ndarray):
    """Flat (obs_cam, obs_pt, obs_uv) of the C ABI (include/ba_amd.h: ba_set_problem) from an ImageFeatures
    list; obs_pt indexes `landmark_ids`."""
    index_of = {int(v): i for i, v in enumerate(np.asarray(landmark_ids))}
    cams, pts, uvs = [], [], []
    for i, f in enumerate(image_features_list):
        f2 = f.get_points_2d()
        ids = f2.get_ids()
        missing = [int(v) for v in ids if int(v) not in index_of]
        if missing:
            raise ValidationError(f"image {i} observes landmarks without a 3-D point: {missing[:5]}")
        cams.append(np.full(len(ids), i, dtype=np.int32))
        pts.append(np.array([index_of[int(v)] for v in ids], dtype=np.int32))
        uvs.append(np.asarray(f2.get_points_2d(), dtype=float).reshape(-1, 2))
    return np.concatenate(cams), np.concatenate(pts), np.vstack(uvs)


# ----------------------------------------------------------------------------------------------- BAL files
_D = np.diag([-1.0, -1.0, 1.0])


def _rodrigues(r: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(r)
    if th < 1e-300:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def _rodrigues_inverse(R: np.ndarray) -> np.ndarray:
    axis, angle = GeometryUtils.axis_angle_from_quaternion(GeometryUtils.quaternion_from_rotation_matrix(R))
    return np.asarray(axis, dtype=float) * float(angle)


def read_bal(path: str, intrinsics: Optional[Sequence[float]] = None) -> dict:
    """Parse a BAL problem.  Returns dict(poses C x 7, points L x 3, obs_cam, obs_pt, obs_uv, intr[7],
    bal_cameras C x 9).  Pixels keep BAL's origin (principal point 0, 0)."""
    with open(path) as fh:
        tok = fh.read().split()
    if len(tok) < 3:
        raise ValidationError("not a BAL file: header missing")
    C, L, n = int(tok[0]), int(tok[1]), int(tok[2])
    need = 3 + 4 * n + 9 * C + 3 * L
    if len(tok) != need:
        raise ValidationError(f"BAL file has {len(tok)} numbers, header implies {need}")
    obs = np.array(tok[3: 3 + 4 * n], dtype=float).reshape(n, 4)
    obs_cam, obs_pt = obs[:, 0].astype(np.int32), obs[:, 1].astype(np.int32)
    if n and (obs_cam.min() < 0 or obs_cam.max() >= C or obs_pt.min() < 0 or obs_pt.max() >= L):
        raise ValidationError("BAL observation refers to a camera or point out of range")
    cams = np.array(tok[3 + 4 * n: 3 + 4 * n + 9 * C], dtype=float).reshape(C, 9)
    points = np.array(tok[3 + 4 * n + 9 * C:], dtype=float).reshape(L, 3)
    if intrinsics is None:
        if C and np.abs(cams[:, 6:9] - cams[0, 6:9]).max() > 1e-12 * max(1.0, np.abs(cams[0, 6:9]).max()):
            raise ValidationError("BAL cameras have different f, k1, k2: the solver has one camera model; pass "
                                  "`intrinsics` (fx, fy, cx, cy, k1, k2, k3) to override")
        f, k1, k2 = cams[0, 6:9] if C else (1.0, 0.0, 0.0)
        intr = np.array([f, f, 0.0, 0.0, k1, k2, 0.0])
    else:
        intr = np.asarray(intrinsics, dtype=float).reshape(7)
    poses = np.empty((C, 7))
    for c in range(C):
        Rw2c = _D @ _rodrigues(cams[c, :3])
        poses[c, :3] = -Rw2c.T @ (_D @ cams[c, 3:6])
        poses[c, 3:7] = GeometryUtils.quaternion_from_rotation_matrix(Rw2c.T)
    return dict(poses=poses, points=points, obs_cam=obs_cam, obs_pt=obs_pt, obs_uv=obs[:, 2:4].copy(), intr=intr,
                bal_cameras=cams)


def write_bal(path: str, poses: np.ndarray, points: np.ndarray, obs_cam, obs_pt, obs_uv, intr: Sequence[float]) -> None:
    """Write a problem as a BAL file.  BAL pixels are relative to the principal point and its model has
    one focal length and k1, k2 only: fx must equal fy and k3 must be 0; (cx, cy) is subtracted."""
    fx, fy, cx, cy, k1, k2, k3 = (float(v) for v in intr)
    if fx != fy or k3 != 0.0:
        raise ValidationError("BAL cameras have one focal length and no k3")
    poses = np.asarray(poses, dtype=float).reshape(-1, 7)
    points = np.asarray(points, dtype=float).reshape(-1, 3)
    uv = np.asarray(obs_uv, dtype=float).reshape(-1, 2) - np.array([cx, cy])
    with open(path, "w") as fh:
        fh.write(f"{len(poses)} {len(points)} {len(uv)}\n")
        for c, p, (u, v) in zip(np.asarray(obs_cam), np.asarray(obs_pt), uv):
            fh.write(f"{int(c)} {int(p)} {float(u)!r} {float(v)!r}\n")
        for pose in poses:
            Rw2c = GeometryUtils.rotation_matrix_from_quaternion(pose[3:7] / np.linalg.norm(pose[3:7])).T
            R_bal = _D @ Rw2c
            t_bal = -R_bal @ pose[:3]
            for v in (*_rodrigues_inverse(R_bal), *t_bal, fx, k1, k2):
                fh.write(f"{float(v)!r}\n")
        for X in points:
            for v in X:
                fh.write(f"{float(v)!r}\n")
