"""Triangulation on the GPU: the step on the input side of the bundle-adjustment path.

Mirrors, with the same argument meaning and return values,

* ``VisualOdometry.triangulate_points(rot_F1_F, originF1_F, rot_F2_F, originF2_F, camera_matrix, pts1, pts2)
  -> (points_3d, reprojection_error)`` of the reference (src/visual_odometry.py:508-570: two views,
  ``cv2.triangulatePoints`` + mean reprojection distance), and
* ``triangulate_landmark(observations, K)`` of src/initialization.py:52-157 in its batched form
  (``triangulate_tracks``): all landmarks of a set of images at once, any number of views each, optional
  Gauss-Newton refinement, minimum-depth validity.

Both run ``ba_triangulate`` of libba_amd.so (include/ba_amd.h); there is no CPU fallback.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .camera import PinHoleCamera
from .camera_pose import CameraPose
from .feature_observation import ImageFeatures, Landmarks3D
from .geometry import GeometryUtils
from .validation_error import ValidationError


def _intrinsics_from_matrix(camera_matrix: np.ndarray) -> np.ndarray:
    camera_matrix = np.asarray(camera_matrix, dtype=float)
    if camera_matrix.shape != (3, 3):
        raise ValidationError(f"camera matrix must be 3x3, got {camera_matrix.shape}")
    return np.array([camera_matrix[0, 0], camera_matrix[1, 1], camera_matrix[0, 2], camera_matrix[1, 2], 0.0, 0.0, 0.0])


def triangulate_points(rot_F1_F: np.ndarray, originF1_F: np.ndarray, rot_F2_F: np.ndarray, originF2_F: np.ndarray,
                       camera_matrix: np.ndarray, pts1: np.ndarray, pts2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Two-view triangulation (reference: VisualOdometry.triangulate_points).

    rot_Fi_F: rotation from frame F to camera frame i (= R_i^T); originFi_F: camera centre in F;
    pts1 / pts2: N x 2 pixels of the same N points.  Returns (N x 3 points in F, N reprojection errors)."""
    GeometryUtils.validate_rotation_matrix(np.asarray(rot_F1_F))
    GeometryUtils.validate_rotation_matrix(np.asarray(rot_F2_F))
    pts1 = np.asarray(pts1, dtype=float)
    pts2 = np.asarray(pts2, dtype=float)
    if pts1.ndim != 2 or pts1.shape[1] != 2 or pts1.shape != pts2.shape:
        raise ValidationError(f"pts1 and pts2 must both be N x 2, got {pts1.shape} and {pts2.shape}")
    n = len(pts1)
    poses = np.array([np.concatenate([np.asarray(o, dtype=float).reshape(3),
                                      GeometryUtils.quaternion_from_rotation_matrix(np.asarray(r, dtype=float).T)])
                      for r, o in ((rot_F1_F, originF1_F), (rot_F2_F, originF2_F))])
    obs_cam = np.repeat(np.arange(2, dtype=np.int32), n)
    obs_pt = np.tile(np.arange(n, dtype=np.int32), 2)
    with _native.Solver() as s:
        s.set_problem(2, n, obs_cam, obs_pt, np.vstack([pts1, pts2]), _intrinsics_from_matrix(camera_matrix))
        points, err, valid = s.triangulate(poses, points=np.full((n, 3), np.nan), min_depth=-np.finfo(float).max)
    return points, err


def triangulate_tracks(image_features_list: Sequence[ImageFeatures], camera_poses: Sequence[CameraPose],
                       camera_model: PinHoleCamera, refine_iters: int = 10, min_depth: float = 0.1,
                       max_reprojection_error: Optional[float] = None) -> Tuple[Landmarks3D, np.ndarray]:
    """All landmarks seen in >= 2 of the images, from known poses: the initial `points3d_guess` of
    BundleAdjustment.optimize.  Returns (Landmarks3D of the accepted ids in ascending order, their
    reprojection errors)."""
    if len(image_features_list) != len(camera_poses):
        raise ValidationError("need one pose per image")
    ids = np.unique(np.concatenate([f.get_points_2d().get_ids() for f in image_features_list])) \
        if image_features_list else np.empty(0, dtype=int)
    index_of = {int(v): i for i, v in enumerate(ids)}
    cams, pts, uvs = [], [], []
    for i, f in enumerate(image_features_list):
        fid = f.get_points_2d().get_ids()
        cams.append(np.full(len(fid), i, dtype=np.int32))
        pts.append(np.array([index_of[int(v)] for v in fid], dtype=np.int32))
        uvs.append(f.get_points_2d().get_points_2d().reshape(-1, 2))
    poses = np.array([np.concatenate([p.position, p.quaternion]) for p in camera_poses])
    with _native.Solver() as s:
        s.set_problem(len(poses), len(ids), np.concatenate(cams), np.concatenate(pts), np.vstack(uvs),
                      camera_model.to_vector())
        points, err, valid = s.triangulate(poses, refine_iters=refine_iters, min_depth=min_depth)
    if max_reprojection_error is not None:
        valid &= err < max_reprojection_error
    return Landmarks3D(points[valid], ids[valid]), err[valid]
