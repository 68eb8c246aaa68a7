"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's triangulation
steps, the input side of the bundle-adjustment path.

Only tests/ may import this module, and only as the checker.

* ``dlt_point`` -- the linear estimate behind ``VisualOdometry.triangulate_points``
  (src/visual_odometry.py:546-556).  The reference calls ``cv2.triangulatePoints(P1, P2, pts1, pts2)``;
  OpenCV (third-party, un-pinned by the reference, not installed in this image) documents and
  implements it as: per point, stack the rows ``u P[2] - P[0]`` and ``v P[2] - P[1]`` of every view into
  A, take the right singular vector of the smallest singular value of A, de-homogenise.  Restated here
  with numpy's SVD for any number of views.  **Parity unpinned against cv2 itself**; anchored on the
  known answer of the reference's own test (tests/test_triangulation.py:130-131, fixture
  tests/golden/known_answers.npz ``tri_*``).
* ``reprojection_error`` -- (|e1| + |e2|) / 2 of src/visual_odometry.py:558-568, for N views the mean.
* ``gauss_newton_point`` -- ``initialization.triangulate_landmark`` (src/initialization.py:76-157) as
  written: start at the centroid of the camera centres, at most 10 Gauss-Newton steps with a forward
  finite-difference Jacobian (eps 1e-6) solved by lstsq, stop at |delta| < 1e-6, reject the point
  unless it is deeper than 0.1 in every view.  That module cannot be imported (it needs ``.features``,
  ``se3_exp`` and ``project_point``, none of which exist in the reference), so its ``project_point`` is
  taken to be the pin-hole projection of the rest of the reference (camera.py:134-139) through the
  pose convention of the BA path, Xc = R^T (X - c).
"""

from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .ref_scipy_path import project, quat_to_rot


def projection_matrix(pose: np.ndarray, intr: Sequence[float]) -> np.ndarray:
    """P = K [R^T | -R^T c] for pose = [c, q] (visual_odometry.py:547-553 with rot_F1_F = R^T)."""
    fx, fy, cx, cy = intr[:4]
    q = pose[3:7] / np.linalg.norm(pose[3:7])
    Rt = quat_to_rot(q).T
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    return K @ np.hstack([Rt, (-Rt @ pose[:3]).reshape(3, 1)])


def undistort_pixel(uv: np.ndarray, intr: Sequence[float], iters: int = 20) -> np.ndarray:
    fx, fy, cx, cy, k1, k2, k3 = intr
    if k1 == 0 and k2 == 0 and k3 == 0:
        return np.asarray(uv, dtype=float)
    xd, yd = (uv[0] - cx) / fx, (uv[1] - cy) / fy
    xn, yn = xd, yd
    for _ in range(iters):
        r2 = xn * xn + yn * yn
        f = 1 + k1 * r2 + k2 * r2**2 + k3 * r2**3
        xn, yn = xd / f, yd / f
    return np.array([fx * xn + cx, fy * yn + cy])


def dlt_point(poses: np.ndarray, uvs: np.ndarray, intr: Sequence[float]) -> np.ndarray:
    """Homogeneous linear triangulation of ONE point seen at pixel uvs[i] by the camera with poses[i]."""
    rows = []
    for pose, uv in zip(poses, uvs):
        P = projection_matrix(pose, intr)
        u, v = undistort_pixel(uv, intr)
        rows.append(u * P[2] - P[0])
        rows.append(v * P[2] - P[1])
    _, _, Vt = np.linalg.svd(np.array(rows))
    X = Vt[-1]
    return X[:3] / X[3]


def camera_points(poses: np.ndarray, X: np.ndarray) -> np.ndarray:
    out = []
    for pose in poses:
        q = pose[3:7] / np.linalg.norm(pose[3:7])
        out.append((X - pose[:3]) @ quat_to_rot(q))
    return np.array(out)


def reprojection_error(poses: np.ndarray, uvs: np.ndarray, intr: Sequence[float], X: np.ndarray) -> float:
    proj = np.vstack([project(xc.reshape(1, 3), intr) for xc in camera_points(poses, X)])
    return float(np.mean(np.linalg.norm(proj - uvs, axis=1)))


def gauss_newton_point(poses: np.ndarray, uvs: np.ndarray, intr: Sequence[float],
                       x_init: Optional[np.ndarray] = None, max_iterations: int = 10, tolerance: float = 1e-6,
                       min_depth: float = 0.1) -> Optional[np.ndarray]:
    def proj(pose, X):
        q = pose[3:7] / np.linalg.norm(pose[3:7])
        return project(((X - pose[:3]) @ quat_to_rot(q)).reshape(1, 3), intr)[0]

    X = np.mean(poses[:, :3], axis=0) if x_init is None else np.array(x_init, dtype=float)  # :77-94
    for _ in range(max_iterations):
        A, b = [], []
        for pose, uv in zip(poses, uvs):
            p0 = proj(pose, X)
            J = []
            for i in range(3):  # :112-119 forward differences
                Xp = X.copy()
                Xp[i] += 1e-6
                J.append((proj(pose, Xp) - p0) / 1e-6)
            A.append(np.array(J).T)
            b.append(-(p0 - uv))
        delta = np.linalg.lstsq(np.vstack(A), np.concatenate(b), rcond=None)[0]
        X = X + delta
        if np.linalg.norm(delta) < tolerance:
            break
    if np.any(camera_points(poses, X)[:, 2] <= min_depth):  # :148-155
        return None
    return X


def triangulate_tracks(poses, obs_cam, obs_pt, obs_uv, intr, n_points, refine_iters=0, min_depth=0.0):
    """Every landmark with >= 2 observations: DLT (+ Gauss-Newton from the DLT point).  Returns
    (points, reprojection error, valid) in landmark order; NaN / False where nothing was computed."""
    pts = np.full((n_points, 3), np.nan)
    err = np.full(n_points, np.nan)
    valid = np.zeros(n_points, bool)
    order = np.argsort(obs_pt, kind="stable")
    bounds = np.flatnonzero(np.r_[True, np.diff(obs_pt[order]) != 0, True])
    for s, e in zip(bounds[:-1], bounds[1:]):
        k = order[s:e]
        if len(k) < 2:
            continue
        j = obs_pt[k[0]]
        P, uv = poses[obs_cam[k]], obs_uv[k]
        X = dlt_point(P, uv, intr)
        if refine_iters > 0:
            Xr = gauss_newton_point(P, uv, intr, x_init=X, max_iterations=refine_iters, min_depth=-np.inf)
            X = X if Xr is None else Xr
        pts[j] = X
        err[j] = reprojection_error(P, uv, intr, X)
        valid[j] = bool(np.all(np.isfinite(X)) and np.all(camera_points(P, X)[:, 2] > min_depth))
    return pts, err, valid
