"""BundleAdjustment -- drop-in for reference ``src/bundle_adjustment.py``.

Same class, method names, argument order and return triple as the reference
(bundle_adjustment.py:12-27, 89-92, 156-163, 288); the optimisation itself runs on an MI355X
through libba_amd.so (hand-written HIP kernels: analytic Jacobians, U/V/W block assembly, Schur
complement, block-Jacobi PCG, back-substitution, Levenberg-Marquardt) instead of
``scipy.optimize.least_squares`` with a ``jac_sparsity`` mask (:201-219).

Deliberate differences from the reference:
* observations are paired with landmarks *by id* (the reference pairs id-sorted projections with
  stored-order observations, which is only right for ascending ids);
* images may hold different numbers of observations (the reference's ``np.array(residuals)``
  needs equal counts, :86); landmarks nobody observes are returned untouched;
* ``bundle_adjustment_sparsity`` returns the mask without printing it densely (:153).
"""

import atexit
import hashlib
from typing import List, Optional, Tuple

import numpy as np
from scipy.sparse import coo_matrix, lil_matrix

from . import _native
from .camera import PinHoleCamera
from .camera_pose import CameraPose
from .feature_observation import Features2D, ImageFeatures, Landmarks3D
from .validation_error import ValidationError


def _pack_observations(observations: List[ImageFeatures], landmark_ids: np.ndarray):
    """Flatten a list of ImageFeatures into (obs_cam, obs_pt, obs_uv), image-major, stored order.
    obs_pt holds the landmark *index* in ``landmark_ids`` (ids are matched explicitly)."""
    landmark_ids = np.asarray(landmark_ids)
    order = np.argsort(landmark_ids, kind="stable")
    sorted_ids = landmark_ids[order]
    if len(sorted_ids) > 1 and np.any(sorted_ids[1:] == sorted_ids[:-1]):
        raise ValidationError("landmark ids must be unique")
    id_sets = [feats.get_points_2d().get_ids() for feats in observations]
    counts = np.array([len(ids) for ids in id_sets])
    if counts.sum() == 0:
        raise ValidationError("no observations")
    all_ids = np.concatenate(id_sets)
    # id -> index in landmark_ids: a direct table when the ids are small non-negative integers (the usual case: one
    # gather instead of a binary search per observation), else a search in the sorted ids
    table = None
    if np.issubdtype(landmark_ids.dtype, np.integer) and np.issubdtype(all_ids.dtype, np.integer) and len(sorted_ids) > 0 \
            and sorted_ids[0] >= 0 and sorted_ids[-1] < 4 * len(sorted_ids) + 1024:
        table = np.full(int(sorted_ids[-1]) + 1, -1, dtype=np.int64)
        table[landmark_ids] = np.arange(len(landmark_ids))
        inside = (all_ids >= 0) & (all_ids < len(table))
        index = np.where(inside, table[np.where(inside, all_ids, 0)], -1)
        bad = index < 0
    else:
        pos = np.clip(np.searchsorted(sorted_ids, all_ids), 0, len(sorted_ids) - 1)
        bad = sorted_ids[pos] != all_ids
        index = order[pos]
    if np.any(bad):
        k = int(np.argmax(bad))
        image = int(np.searchsorted(np.cumsum(counts), k, side="right"))
        raise ValidationError(f"image {image} observes landmark id {all_ids[k]} that is not in points3d_guess")
    cams = np.repeat(np.arange(len(observations), dtype=np.int32), counts)
    uvs = np.vstack([feats.get_points_2d().get_points_2d() for feats, c in zip(observations, counts) if c > 0])
    return cams, index.astype(np.int32), uvs


def _pose_rows(camera_pose_initial: CameraPose, poses: List[CameraPose]) -> np.ndarray:
    return np.array([np.concatenate([p.position, p.quaternion]) for p in [camera_pose_initial] + list(poses)])


class BundleAdjustment:
    """Bundle adjustment of 3-D landmarks and camera poses (first pose fixed)."""

    _residual_solver = None  # (key of the observations, Solver): residual_function's handle for large problems

    def __init__(self, **solver_options) -> None:
        # the reference constructor takes nothing (bundle_adjustment.py:15-16); options are an extension
        self.solver_options = solver_options
        self.last_report: Optional[dict] = None

    # ------------------------------------------------------------------ residuals (reference :19-87)
    @staticmethod
    def residual_function(params: np.ndarray, camera_pose_initial: CameraPose, num_poses: int,
                          landmark_ids: np.ndarray, observations: List[ImageFeatures],
                          camera_model: PinHoleCamera) -> np.ndarray:
        """Residuals [u0-u0', v0-v0', ...] of parameter vector [7*num_poses poses | 3*L landmarks],
        image-major, evaluated by the HIP residual kernel."""
        params = np.asarray(params, dtype=np.float64)
        L = len(landmark_ids)
        if params.shape != (7 * num_poses + 3 * L,):
            raise ValidationError(f"params must have {7 * num_poses + 3 * L} entries, got {params.shape}")
        if len(observations) != num_poses + 1:
            raise ValidationError("observations must hold num_poses + 1 images")
        obs_cam, obs_pt, obs_uv = _pack_observations(observations, landmark_ids)
        poses = np.vstack([np.concatenate([camera_pose_initial.position, camera_pose_initial.quaternion])[None],
                           params[: 7 * num_poses].reshape(num_poses, 7)])
        points = params[7 * num_poses:].reshape(L, 3)
        intr = camera_model.to_vector()
        # scipy calls this tens of times per Jacobian: no handle, no sort, no allocation when the problem fits the
        # library's static scratch (ba_evaluate_static); larger problems keep ONE handle for as long as the
        # observations stay the same (ba_set_problem sorts and uploads them: once, not per call)
        if _native.fits_static(num_poses + 1, L, len(obs_cam)):
            return _native.evaluate_static(num_poses + 1, L, obs_cam, obs_pt, obs_uv, intr, poses, points)[0]
        # the key is a cryptographic digest of the observation arrays (Python's hash() of their bytes is 64 bits and
        # salted per process: a collision would silently return the residuals of another problem)
        dig = hashlib.blake2b(digest_size=32)
        for a in (obs_cam, obs_pt, obs_uv, np.asarray(intr, dtype=np.float64)):
            dig.update(np.ascontiguousarray(a).tobytes())
        key = (num_poses + 1, L, len(obs_cam), dig.digest())
        cache = BundleAdjustment._residual_solver
        if cache is None or cache[0] != key:
            BundleAdjustment.clear_residual_cache()
            s = _native.Solver()
            s.set_problem(num_poses + 1, L, obs_cam, obs_pt, obs_uv, intr)
            BundleAdjustment._residual_solver = cache = (key, s)
        return cache[1].evaluate(poses, points)[0]

    @staticmethod
    def clear_residual_cache() -> None:
        """Close the GPU handle that residual_function keeps for problems beyond the static entry point's limits
        (also registered with atexit, so that no handle is open when the interpreter shuts the HIP runtime down)."""
        cache, BundleAdjustment._residual_solver = BundleAdjustment._residual_solver, None
        if cache is not None:
            cache[1].close()

    # ------------------------------------------------------------------ mask (reference :89-154)
    @staticmethod
    def bundle_adjustment_sparsity(num_poses: int, num_landmarks: int,
                                   observations: List[ImageFeatures]) -> lil_matrix:
        """0/1 Jacobian pattern in the reference's 7-parameter layout.  Kept for API parity; the
        GPU solver does not use it (its pattern is the observation index itself)."""
        ids = np.unique(np.concatenate([o.get_points_2d().get_ids() for o in observations]))
        obs_cam, obs_pt, _ = _pack_observations(observations, ids)  # rank among observed ids (:118-123)
        nobs = len(obs_cam)
        rows2 = np.arange(2 * nobs)
        k = rows2 // 2
        r = [np.repeat(rows2, 3)]
        c = [((num_poses * 7 + 3 * obs_pt[k])[:, None] + np.arange(3)[None, :]).reshape(-1)]
        free = obs_cam[k] > 0
        r.append(np.repeat(rows2[free], 7))
        c.append((((obs_cam[k][free] - 1) * 7)[:, None] + np.arange(7)[None, :]).reshape(-1))
        r, c = np.concatenate(r), np.concatenate(c)
        return coo_matrix((np.ones(len(r), dtype=int), (r, c)),
                          shape=(2 * nobs, num_poses * 7 + num_landmarks * 3)).tolil()

    # ------------------------------------------------------------------ optimise (reference :156-288)
    def optimize(self, image_features_list: List[ImageFeatures], camera_pose_initial: CameraPose,
                 camera_model: PinHoleCamera, camera_poses_guess: List[CameraPose],
                 points3d_guess: Landmarks3D) -> Tuple[List[ImageFeatures], Landmarks3D, List[CameraPose]]:
        if len(image_features_list) != len(camera_poses_guess) + 1:
            raise ValidationError("image_features_list must hold len(camera_poses_guess) + 1 images")
        landmark_ids = points3d_guess.get_ids()
        points0 = points3d_guess.get_points_3d()
        obs_cam, obs_pt, obs_uv = _pack_observations(image_features_list, landmark_ids)
        poses0 = _pose_rows(camera_pose_initial, camera_poses_guess)
        n_cams = len(image_features_list)

        with _native.Solver(**self.solver_options) as solver:
            solver.set_problem(n_cams, len(landmark_ids), obs_cam, obs_pt, obs_uv, camera_model.to_vector())
            poses, points, report = solver.solve(poses0, points0)
        self.last_report = report

        # unpack (reference :221-244): fixed pose first, original timestamps, unit quaternions
        optimized_camera_poses = [camera_pose_initial]
        for i, guess in enumerate(camera_poses_guess):
            row = poses[i + 1]
            optimized_camera_poses.append(CameraPose(row[:3].copy(), row[3:7] / np.linalg.norm(row[3:7]),
                                                     guess.timestamp))
        optimized_points3d = Landmarks3D(points, landmark_ids)

        # re-project the observed landmarks through the optimised poses (reference :246-286);
        # width/height come from image 0 (:190-191).  The packed observations are image-major: image i owns one slice.
        image_width = image_features_list[0].image_width
        image_height = image_features_list[0].image_height
        starts = np.searchsorted(obs_cam, np.arange(n_cams + 1))
        # same arithmetic as CameraPose.transform_points_world_to_camera per image ((P - t) @ R on that image's slice),
        # then ONE call of the camera model's project over all observations (it works point by point: same bits as a
        # call per image, without fifty rounds of argument checks)
        pts_obs = points[obs_pt]
        pts_cam = np.empty_like(pts_obs)
        for i in range(n_cams):
            if starts[i] < starts[i + 1]:
                pose = optimized_camera_poses[i]
                pts_cam[starts[i]:starts[i + 1]] = (pts_obs[starts[i]:starts[i + 1]] - pose.position.reshape(1, 3)) @ pose.rotation_matrix
        uv_all = camera_model.project(pts_cam) if len(pts_cam) else np.zeros((0, 2))
        optimized_image_features = []
        for i, feats in enumerate(image_features_list):
            if starts[i] == starts[i + 1]:
                optimized_image_features.append(feats)
                continue
            optimized_image_features.append(
                ImageFeatures(feats.get_timestamp(), feats.get_camera_id(), image_width, image_height,
                              Features2D(uv_all[starts[i]:starts[i + 1]], feats.get_points_2d().get_ids())))
        return optimized_image_features, optimized_points3d, optimized_camera_poses


atexit.register(BundleAdjustment.clear_residual_cache)
