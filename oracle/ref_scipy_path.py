"""ORACLE (test infrastructure, not product code): CPU restatement of the reference BA path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline -- never as the thing shipped.  numpy +
scipy only; nothing here imports the product package or the HIP library.

What it restates (reference = /root/reference, AubreyC/visual-odometry):

* ``residual_per_image``  -- ``BundleAdjustment.residual_function`` exactly as written
  (src/bundle_adjustment.py:41-87): per image, select landmarks with ``np.intersect1d`` (id-SORTED,
  src/feature_observation.py:117-120), world->camera ``(P - t) @ R(q)`` (src/camera_pose.py:85-86,
  src/geometry.py:185-213), pin-hole projection (src/camera.py:134-165), ``observed - projected``.
  Pure-Python loop over images: use for small cases.
* ``residual_packed``     -- the same arithmetic vectorised over a flat observation list
  (image-major, ids ascending inside an image: the only layout on which the reference itself is
  self-consistent).  Bit-identical to ``residual_per_image`` on such input
  (tests/test_oracle_golden.py).
* ``sparsity_mask``       -- ``bundle_adjustment_sparsity`` (src/bundle_adjustment.py:107-151)
  vectorised, without the dense ``print(A.toarray())`` of :153.
* ``solve_scipy_path``    -- the ``least_squares`` call of src/bundle_adjustment.py:201-219 with the
  reference's literal settings (trf, jac_sparsity, x_scale='jac', ftol=xtol=gtol=1e-6,
  max_nfev=2000).  The solver arithmetic lives in scipy (third-party, not vendored by the
  reference, un-pinned there; scipy 1.15.3 in this image).
* ``polish``              -- tight-tolerance run used as the "true minimum" oracle (SURVEY.md
  Appendix C): dense exact trust-region for small scenes, analytic sparse Jacobian for large.

Parity pin: tests/golden/*.npz were produced by oracle/make_golden.py, which imports the
reference itself in the build container and records its outputs; tests/test_oracle_golden.py
checks every function here against them.
"""

from __future__ import annotations

import time
from typing import Optional, Sequence

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import coo_matrix, csr_matrix

POSE_SIZE = 7  # 3 position + 4 quaternion [w, x, y, z]   (bundle_adjustment.py:42)
REF_LSQ_KWARGS = dict(method="trf", x_scale="jac", ftol=1e-6, xtol=1e-6, gtol=1e-6, max_nfev=2000, verbose=0)


# --------------------------------------------------------------------------- small helpers
def quat_to_rot(q: np.ndarray) -> np.ndarray:
    """R(q) camera->world, q = [w,x,y,z]; normalise only if | ||q||-1 | > 1e-6 (geometry.py:185-213)."""
    n = np.linalg.norm(q)
    if abs(n - 1.0) > 1e-6:
        q = q / n
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
            [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
            [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y],
        ]
    )


def project(points_cam: np.ndarray, intr: Sequence[float]) -> np.ndarray:
    """Pin-hole projection with optional radial distortion (camera.py:134-165). intr = fx,fy,cx,cy,k1,k2,k3."""
    fx, fy, cx, cy, k1, k2, k3 = intr
    z = points_cam[:, 2]
    if np.any(z == 0):
        raise RuntimeError("zero Z coordinate")  # reference: ProcessingError (camera.py:129-132)
    xn = points_cam[:, 0] / z
    yn = points_cam[:, 1] / z
    if k1 != 0.0 or k2 != 0.0 or k3 != 0.0:
        r2 = xn**2 + yn**2
        r = np.sqrt(r2)
        factor = 1.0 + k1 * r2 + k2 * r2**2 + k3 * r2**3
        with np.errstate(invalid="ignore", divide="ignore"):
            s = np.where(r == 0, 1.0, (r * factor) / r)
        xn = xn * s
        yn = yn * s
    return np.vstack(((fx * xn) + cx, (fy * yn) + cy)).T


# --------------------------------------------------------------------------- residuals
def residual_per_image(params, pose0, num_poses, landmark_ids, image_ids, image_uv, intr) -> np.ndarray:
    """Literal restatement of bundle_adjustment.py:41-87 on plain arrays.

    pose0: (7,) fixed pose [t, q];  image_ids[i]: ids observed in image i (stored order);
    image_uv[i]: (n_i, 2) observations in the same stored order.
    """
    pts_all = params[num_poses * POSE_SIZE:].reshape(-1, 3)
    out = []
    for i, (ids_i, uv_i) in enumerate(zip(image_ids, image_uv)):
        if i == 0:
            t, q = pose0[:3], pose0[3:7]
        else:
            p = params[(i - 1) * POSE_SIZE: i * POSE_SIZE]
            t, q = p[:3], p[3:7]
            q = q / np.linalg.norm(q)  # :63
        _, sel, _ = np.intersect1d(landmark_ids, ids_i, return_indices=True)  # id-sorted selection
        xc = (pts_all[sel] - t.reshape(1, 3)) @ quat_to_rot(q)
        out.append(uv_i.flatten() - project(xc, intr).flatten())
    return np.array(out).flatten()  # :86 -- ragged images raise here, as in the reference


def rotations_from_params(params, pose0, num_poses) -> tuple[np.ndarray, np.ndarray]:
    """All C = num_poses+1 rotations (C,3,3) and positions (C,3); pose 0 is the fixed one."""
    poses = np.vstack([pose0.reshape(1, 7), params[: num_poses * POSE_SIZE].reshape(num_poses, 7)])
    t = poses[:, :3]
    q = poses[:, 3:7].copy()
    q[1:] = q[1:] / np.linalg.norm(q[1:], axis=1, keepdims=True)  # bundle_adjustment.py:63 (free poses only)
    n = np.linalg.norm(q, axis=1)
    off = np.abs(n - 1.0) > 1e-6  # geometry.py:189-190
    q[off] = q[off] / n[off, None]
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
    return R, t


def camera_frame_points(params, pose0, num_poses, obs_cam, obs_pt):
    R, t = rotations_from_params(params, pose0, num_poses)
    pts = params[num_poses * POSE_SIZE:].reshape(-1, 3)
    d = pts[obs_pt] - t[obs_cam]
    Rc = R[obs_cam]
    # row form (P - t) @ R  ==  R^T (P - t); summed in the same j = 0,1,2 order as the matmul
    xc = d[:, 0:1] * Rc[:, 0, :] + d[:, 1:2] * Rc[:, 1, :] + d[:, 2:3] * Rc[:, 2, :]
    return xc, Rc


def residual_packed(params, pose0, num_poses, obs_cam, obs_pt, obs_uv, intr) -> np.ndarray:
    """Vectorised residual over a flat observation list (obs k: image obs_cam[k], landmark *index*
    obs_pt[k], pixel obs_uv[k]).  Output order = input order, [u0, v0, u1, v1, ...]."""
    xc, _ = camera_frame_points(params, pose0, num_poses, obs_cam, obs_pt)
    return (obs_uv - project(xc, intr)).reshape(-1)


# --------------------------------------------------------------------------- sparsity mask
def sparsity_mask(num_poses: int, num_landmarks: int, obs_cam, obs_pt) -> csr_matrix:
    """0/1 pattern of bundle_adjustment.py:107-151: residual rows 2k, 2k+1 depend on the 7 columns
    of pose obs_cam[k]-1 (if obs_cam[k] > 0) and the 3 columns of landmark obs_pt[k].

    The reference maps a landmark id to its rank among the *observed* ids (:118-123); with sorted
    ids and every landmark observed (all reference scenes) that rank is the landmark index, which is
    what obs_pt holds.
    """
    obs_cam = np.asarray(obs_cam)
    obs_pt = np.asarray(obs_pt)
    nobs = len(obs_cam)
    rows2 = np.arange(2 * nobs)
    k_of_row = rows2 // 2
    lm_rows = np.repeat(rows2, 3)
    lm_cols = (num_poses * POSE_SIZE + 3 * obs_pt[k_of_row])[:, None] + np.arange(3)[None, :]
    free = obs_cam[k_of_row] > 0
    pr = rows2[free]
    pose_rows = np.repeat(pr, POSE_SIZE)
    pose_cols = ((obs_cam[k_of_row][free] - 1) * POSE_SIZE)[:, None] + np.arange(POSE_SIZE)[None, :]
    r = np.concatenate([pose_rows, lm_rows])
    c = np.concatenate([pose_cols.reshape(-1), lm_cols.reshape(-1)])
    m = coo_matrix((np.ones(len(r), dtype=np.int8), (r, c)),
                   shape=(2 * nobs, num_poses * POSE_SIZE + 3 * num_landmarks))
    return m.tocsr()


# --------------------------------------------------------------------------- analytic Jacobian (polish only)
def jacobian_packed(params, pose0, num_poses, obs_cam, obs_pt, obs_uv, intr) -> csr_matrix:
    """Analytic d residual / d params in the reference's 7-parameter pose layout (k = 0 only).

    Not part of the reference (it uses finite differences, scipy/_numdiff.py); used only by
    ``polish`` for large scenes.  Checked against 2-point finite differences of residual_packed in
    tests/test_oracle_golden.py.
    """
    fx, fy = intr[0], intr[1]
    assert intr[4] == 0 and intr[5] == 0 and intr[6] == 0
    obs_cam = np.asarray(obs_cam)
    obs_pt = np.asarray(obs_pt)
    nobs = len(obs_cam)
    xc, Rc = camera_frame_points(params, pose0, num_poses, obs_cam, obs_pt)
    x, y, z = xc[:, 0], xc[:, 1], xc[:, 2]
    Jpi = np.zeros((nobs, 2, 3))
    Jpi[:, 0, 0] = fx / z
    Jpi[:, 0, 2] = -fx * x / z**2
    Jpi[:, 1, 1] = fy / z
    Jpi[:, 1, 2] = -fy * y / z**2
    RT = np.transpose(Rc, (0, 2, 1))
    JP = -np.einsum("nij,njk->nik", Jpi, RT)  # d r / d P   (Xc = R^T (P - t))
    Jt = -JP  # d r / d t
    # d r / d q (raw 4-vector, residual normalises q): Xc = R(q/|q|)^T d,  d = P - t
    poses = params[: num_poses * POSE_SIZE].reshape(num_poses, 7)
    qraw = np.vstack([pose0[3:7].reshape(1, 4), poses[:, 3:7]])
    qn = np.linalg.norm(qraw, axis=1)
    qu = qraw / qn[:, None]
    q = qu[obs_cam]
    w, a, b, c = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    pts = params[num_poses * POSE_SIZE:].reshape(-1, 3)
    tt = np.vstack([pose0[:3].reshape(1, 3), poses[:, :3]])
    d = pts[obs_pt] - tt[obs_cam]
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    # rows of R^T d wrt unit quaternion (w,a,b,c) -> dXc/dq_unit, shape (n,3,4)
    D = np.empty((nobs, 3, 4))
    # Xc0 = R00 dx + R10 dy + R20 dz
    D[:, 0, 0] = 2 * c * dy - 2 * b * dz
    D[:, 0, 1] = 2 * b * dy + 2 * c * dz
    D[:, 0, 2] = -4 * b * dx + 2 * a * dy - 2 * w * dz
    D[:, 0, 3] = -4 * c * dx + 2 * w * dy + 2 * a * dz
    # Xc1 = R01 dx + R11 dy + R21 dz
    D[:, 1, 0] = -2 * c * dx + 2 * a * dz
    D[:, 1, 1] = 2 * b * dx - 4 * a * dy + 2 * w * dz
    D[:, 1, 2] = 2 * a * dx + 2 * c * dz
    D[:, 1, 3] = -2 * w * dx - 4 * c * dy + 2 * b * dz
    # Xc2 = R02 dx + R12 dy + R22 dz
    D[:, 2, 0] = 2 * b * dx - 2 * a * dy
    D[:, 2, 1] = 2 * c * dx - 2 * w * dy - 4 * a * dz
    D[:, 2, 2] = 2 * w * dx + 2 * c * dy - 4 * b * dz
    D[:, 2, 3] = 2 * a * dx + 2 * b * dy
    # chain through normalisation: d q_unit / d q_raw = (I - q q^T) / |q_raw|
    P = (np.eye(4)[None] - q[:, :, None] * q[:, None, :]) / qn[obs_cam][:, None, None]
    Jq = -np.einsum("nij,njk,nkl->nil", Jpi, D, P)
    free = obs_cam > 0
    kk = np.nonzero(free)[0]
    rows = []
    cols = []
    vals = []
    for r_ in range(2):
        for c_ in range(3):
            rows.append(2 * kk + r_)
            cols.append((obs_cam[kk] - 1) * 7 + c_)
            vals.append(Jt[kk, r_, c_])
        for c_ in range(4):
            rows.append(2 * kk + r_)
            cols.append((obs_cam[kk] - 1) * 7 + 3 + c_)
            vals.append(Jq[kk, r_, c_])
        for c_ in range(3):
            rows.append(2 * np.arange(nobs) + r_)
            cols.append(num_poses * 7 + 3 * obs_pt + c_)
            vals.append(JP[:, r_, c_])
    J = coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                   shape=(2 * nobs, num_poses * 7 + 3 * len(pts)))
    return J.tocsr()


# --------------------------------------------------------------------------- solves
def solve_scipy_path(x0, pose0, num_poses, obs_cam, obs_pt, obs_uv, intr, num_landmarks: Optional[int] = None,
                     max_nfev: Optional[int] = None, loss: str = "linear", f_scale: float = 1.0):
    """The reference's optimiser call (bundle_adjustment.py:201-219) on packed arrays.

    ``loss`` / ``f_scale`` are scipy's robust-loss hook, which the reference leaves at 'linear' / 1.0.
    Returns (OptimizeResult, seconds inside least_squares, seconds building the mask).
    """
    if num_landmarks is None:
        num_landmarks = (len(x0) - num_poses * POSE_SIZE) // 3
    t0 = time.perf_counter()
    mask = sparsity_mask(num_poses, num_landmarks, obs_cam, obs_pt)
    t1 = time.perf_counter()
    kw = dict(REF_LSQ_KWARGS)
    if max_nfev is not None:
        kw["max_nfev"] = max_nfev
    kw.update(loss=loss, f_scale=f_scale)
    res = least_squares(residual_packed, x0, jac_sparsity=mask,
                        args=(pose0, num_poses, np.asarray(obs_cam), np.asarray(obs_pt), obs_uv, intr), **kw)
    t2 = time.perf_counter()
    return res, t2 - t1, t1 - t0


def polish(x_start, pose0, num_poses, obs_cam, obs_pt, obs_uv, intr, dense: bool, max_nfev: int = 200,
           loss: str = "linear", f_scale: float = 1.0):
    """Tight-tolerance minimum used as the two-sided cost oracle."""
    args = (pose0, num_poses, np.asarray(obs_cam), np.asarray(obs_pt), obs_uv, intr)
    if dense:
        return least_squares(residual_packed, x_start, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15,
                             gtol=1e-12, max_nfev=max_nfev, args=args, loss=loss, f_scale=f_scale)
    return least_squares(residual_packed, x_start, jac=jacobian_packed, method="trf", x_scale="jac",
                         tr_solver="lsmr", tr_options={"atol": 1e-12, "btol": 1e-12},
                         ftol=1e-14, xtol=1e-14, gtol=1e-12, max_nfev=max_nfev, args=args, loss=loss, f_scale=f_scale)


def robust_cost(residuals: np.ndarray, loss: str = "linear", f_scale: float = 1.0) -> float:
    """scipy's cost for a residual vector: 0.5 * f_scale^2 * sum rho((f / f_scale)^2)
    (scipy/optimize/_lsq/least_squares.py:186-236)."""
    z = (np.asarray(residuals) / f_scale) ** 2
    rho = {"linear": lambda z: z, "soft_l1": lambda z: 2 * (np.sqrt(1 + z) - 1),
           "huber": lambda z: np.where(z <= 1, z, 2 * np.sqrt(np.maximum(z, 1e-300)) - 1),
           "cauchy": np.log1p, "arctan": np.arctan}[loss](z)
    return float(0.5 * f_scale**2 * np.sum(rho))


def rms_reprojection(residuals: np.ndarray) -> float:
    """sqrt(mean over observations of |r_k|^2), r_k the 2-vector residual of observation k."""
    return float(np.sqrt(np.sum(residuals**2) / (len(residuals) // 2)))
