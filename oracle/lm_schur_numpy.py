"""ORACLE (test infrastructure, not product code): numpy restatement of the Schur-complement LM
solver that the HIP kernels implement.

The reference has NO counterpart of this algorithm (it calls scipy TRF/LSMR with finite
differences, src/bundle_adjustment.py:201-219; the Schur trick is only described in
notes/schur_complement_trick.md:85-177).  This file exists so that every intermediate the GPU
produces (residuals, 2x6 / 2x3 Jacobian blocks, U, V, W, gradients, S, rhs, step, trial cost) has a
CPU value to be compared with, and as the tight-tolerance "polished minimum" oracle for scenes too
large for scipy's dense exact trust region.  The maths is SURVEY.md Appendix A, in the reference's
conventions (Xc = R(q)^T (P - t), r = observed - projected, cost = 0.5 * sum r^2).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ----------------------------------------------------------------------------- geometry
def quat_to_rot(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
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


def quat_mul(a, b):
    a0, a1, a2, a3 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    b0, b1, b2, b3 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                     a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], axis=-1)


def retract(poses: np.ndarray, dc: np.ndarray, free: np.ndarray) -> np.ndarray:
    """t <- t + dt ; q <- normalise(q (x) [1, dtheta/2])  (SURVEY.md Appendix A)."""
    out = poses.copy()
    out[:, :3] += dc[:, :3] * free[:, None]
    dq = np.concatenate([np.ones((len(poses), 1)), 0.5 * dc[:, 3:6] * free[:, None]], axis=1)
    q = quat_mul(poses[:, 3:7], dq)
    out[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return out


# ----------------------------------------------------------------------------- linearisation
def linearize(poses, points, obs_cam, obs_pt, obs_uv, intr, want_jac=True):
    """r (n,2), Jc = d r/d[dt, dtheta] (n,2,6), Jp = d r/dP (n,2,3); radial distortion as the
    reference's PinHoleCamera.project applies it (src/camera.py:141-159)."""
    fx, fy, cx, cy = intr[:4]
    k1, k2, k3 = (intr[4], intr[5], intr[6]) if len(intr) >= 7 else (0.0, 0.0, 0.0)
    R = quat_to_rot(poses[:, 3:7])[obs_cam]
    d = points[obs_pt] - poses[obs_cam, :3]
    xc = np.einsum("nji,nj->ni", R, d)  # R^T d
    x, y, z = xc[:, 0], xc[:, 1], xc[:, 2]
    iz = 1.0 / z
    xn, yn = x * iz, y * iz
    r2 = xn * xn + yn * yn
    f = 1.0 + k1 * r2 + k2 * r2**2 + k3 * r2**3
    r = np.stack([obs_uv[:, 0] - (fx * xn * f + cx), obs_uv[:, 1] - (fy * yn * f + cy)], axis=1)
    if not want_jac:
        return r, int(np.sum(z <= 0)), None
    n = len(x)
    fp = k1 + 2 * k2 * r2 + 3 * k3 * r2**2  # d f / d r2
    # d pixel / d (xn, yn), then d (xn, yn) / d Xc = [iz 0 -xn iz; 0 iz -yn iz]
    a00, a01 = fx * (f + 2 * xn * xn * fp), fx * (2 * xn * yn * fp)
    a10, a11 = fy * (2 * xn * yn * fp), fy * (f + 2 * yn * yn * fp)
    Jpi = np.zeros((n, 2, 3))
    Jpi[:, 0, 0] = a00 * iz
    Jpi[:, 0, 1] = a01 * iz
    Jpi[:, 0, 2] = -(a00 * xn + a01 * yn) * iz
    Jpi[:, 1, 0] = a10 * iz
    Jpi[:, 1, 1] = a11 * iz
    Jpi[:, 1, 2] = -(a10 * xn + a11 * yn) * iz
    RT = np.transpose(R, (0, 2, 1))
    A = np.einsum("nij,njk->nik", Jpi, RT)  # Jpi R^T
    skew = np.zeros((n, 3, 3))
    skew[:, 0, 1] = -z; skew[:, 0, 2] = y
    skew[:, 1, 0] = z; skew[:, 1, 2] = -x
    skew[:, 2, 0] = -y; skew[:, 2, 1] = x
    Jc = np.concatenate([A, -np.einsum("nij,njk->nik", Jpi, skew)], axis=2)
    return r, Jc, -A


def rho_functions(loss: str, z: np.ndarray):
    """(rho, rho', rho'') of scipy's implemented losses (scipy/optimize/_lsq/least_squares.py:186-236)."""
    if loss == "linear":
        return z, np.ones_like(z), np.zeros_like(z)
    if loss == "soft_l1":
        t = 1 + z
        return 2 * (np.sqrt(t) - 1), t**-0.5, -0.5 * t**-1.5
    if loss == "huber":
        big = z > 1
        zs = np.where(big, z, 1.0)
        return (np.where(big, 2 * np.sqrt(zs) - 1, z), np.where(big, zs**-0.5, 1.0),
                np.where(big, -0.5 * zs**-1.5, 0.0))
    if loss == "cauchy":
        return np.log1p(z), 1 / (1 + z), -1 / (1 + z) ** 2
    if loss == "arctan":
        t = 1 + z * z
        return np.arctan(z), 1 / t, -2 * z / t**2
    raise ValueError(loss)


def robustify(r, Jc, Jp, loss: str, f_scale: float):
    """Re-weighting for a robust loss on every scalar residual f (z = (f / C)^2): the Jacobian row and f
    are both multiplied by sqrt(rho'(z)), so that J^T f is the gradient of the robust cost
    0.5 C^2 sum rho(z) -- the objective of scipy's `loss=` -- and J^T J = J^T rho' J majorises its
    Hessian (all implemented rho are concave in z).  scipy itself uses Triggs' second-order term,
    s^2 = max(rho' + 2 rho'' z, eps) (scale_for_robust_loss_function, scipy/optimize/_lsq/common.py);
    clamped like that the rows of outliers vanish from J^T J, the Marquardt diagonal with them, and LM
    rejects every step from a poor initial guess, so the solver (HIP and this restatement) keeps the
    first-order weight only.  Same objective and minimisers, different Gauss-Newton model.
    Returns the scaled (r, Jc, Jp) and the robust cost."""
    z = (r / f_scale) ** 2
    rho0, rho1, _ = rho_functions(loss, z)
    cost = 0.5 * f_scale**2 * float(np.sum(rho0))
    if loss == "linear":
        return r, Jc, Jp, cost
    s = np.sqrt(rho1)
    return r * s, (None if Jc is None else Jc * s[:, :, None]), (None if Jp is None else Jp * s[:, :, None]), cost


def normal_blocks(r, Jc, Jp, obs_cam, obs_pt, C, L):
    """U (C,6,6), V (L,3,3), W (n,6,3), gc (C,6), gp (L,3), cost = 0.5 sum r^2."""
    U = np.zeros((C, 6, 6)); V = np.zeros((L, 3, 3)); gc = np.zeros((C, 6)); gp = np.zeros((L, 3))
    np.add.at(U, obs_cam, np.einsum("nki,nkj->nij", Jc, Jc))
    np.add.at(V, obs_pt, np.einsum("nki,nkj->nij", Jp, Jp))
    np.add.at(gc, obs_cam, np.einsum("nki,nk->ni", Jc, r))
    np.add.at(gp, obs_pt, np.einsum("nki,nk->ni", Jp, r))
    W = np.einsum("nki,nkj->nij", Jc, Jp)
    return U, V, W, gc, gp, 0.5 * float(np.sum(r * r))


def damped_v_inverse(V, lam, min_diag=1e-12):
    Vd = V.copy()
    idx = np.arange(3)
    Vd[:, idx, idx] += lam * np.maximum(V[:, idx, idx], min_diag)
    return np.linalg.inv(Vd)


def schur_system(U, V, W, gc, gp, obs_cam, obs_pt, lam, free, min_diag=1e-12):
    """Dense reduced camera system over ALL C cameras (fixed ones get identity rows):
    S = (U + lam D_U) - sum_tracks W Vd^-1 W^T,  rhs = gc - sum W Vd^-1 gp   (solve S dc = -rhs)."""
    C = len(U)
    Vi = damped_v_inverse(V, lam, min_diag)
    Y = np.einsum("nij,njk->nik", W, Vi[obs_pt])  # (n,6,3)
    S = np.zeros((C, C, 6, 6))
    idx = np.arange(6)
    for c in range(C):
        S[c, c] = U[c]
        S[c, c, idx, idx] += lam * np.maximum(U[c, idx, idx], min_diag)
    rhs = gc - _scatter(np.einsum("nij,nj->ni", Y, gp[obs_pt]), obs_cam, C)
    order = np.argsort(obs_pt, kind="stable")
    starts = np.flatnonzero(np.r_[True, np.diff(obs_pt[order]) != 0, True])
    for s, e in zip(starts[:-1], starts[1:]):
        k = order[s:e]
        blk = np.einsum("aij,bkj->abik", Y[k], W[k])  # Y_a W_b^T
        ca = obs_cam[k]
        S[ca[:, None], ca[None, :]] -= blk
    S = S.transpose(0, 2, 1, 3).reshape(6 * C, 6 * C)
    fmask = np.repeat(free, 6)
    S[~fmask, :] = 0; S[:, ~fmask] = 0
    S[~fmask, ~fmask] = 1.0
    rhs = rhs * free[:, None]
    return S, rhs.reshape(-1), Vi


def _scatter(vals, index, n):
    out = np.zeros((n,) + vals.shape[1:])
    np.add.at(out, index, vals)
    return out


def schur_system_sparse(U, V, W, gc, gp, obs_cam, obs_pt, lam, free, min_diag=1e-12):
    """Same as schur_system but assembled as a scipy CSR matrix via vectorised pair expansion;
    requires every track to have the same length (ring / all-visible scenes)."""
    C = len(U)
    Vi = damped_v_inverse(V, lam, min_diag)
    Y = np.einsum("nij,njk->nik", W, Vi[obs_pt])
    rhs = (gc - _scatter(np.einsum("nij,nj->ni", Y, gp[obs_pt]), obs_cam, C)) * free[:, None]
    order = np.argsort(obs_pt, kind="stable")
    L = len(V)
    d = len(order) // L
    assert d * L == len(order) and np.all(np.bincount(obs_pt, minlength=L) == d)
    k = order.reshape(L, d)
    a = np.repeat(k, d, axis=1).reshape(-1)
    b = np.tile(k, (1, d)).reshape(-1)
    blk = -np.einsum("nij,nkj->nik", Y[a], W[b])  # (L d^2, 6, 6)
    ca, cb = obs_cam[a], obs_cam[b]
    Ud = U.copy()
    idx = np.arange(6)
    Ud[:, idx, idx] += lam * np.maximum(U[:, idx, idx], min_diag)
    blocks = np.concatenate([blk, Ud])
    ra = np.concatenate([ca, np.arange(C)])
    rb = np.concatenate([cb, np.arange(C)])
    keep = free[ra] & free[rb]
    blocks, ra, rb = blocks[keep], ra[keep], rb[keep]
    rows = (6 * ra[:, None, None] + idx[None, :, None]) + 0 * idx[None, None, :]
    cols = (6 * rb[:, None, None] + idx[None, None, :]) + 0 * idx[None, :, None]
    S = sp.coo_matrix((blocks.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(6 * C, 6 * C)).tocsr()
    fixed = np.flatnonzero(~np.repeat(free, 6))
    S = S + sp.coo_matrix((np.ones(len(fixed)), (fixed, fixed)), shape=S.shape).tocsr()
    return S, rhs.reshape(-1), Vi


def back_substitute(dc, Vi, W, gp, obs_cam, obs_pt, L):
    """dp_j = -Vd_j^-1 (gp_j + sum_k W_k^T dc_{cam k})."""
    t = gp + _scatter(np.einsum("nij,ni->nj", W, dc[obs_cam]), obs_pt, L)
    return -np.einsum("lij,lj->li", Vi, t)


GUARD_KEEP = 0.1  # == GUARD_KEEP of csrc/ba_kernels.h


def guard_landmark_steps(poses, points, dp, obs_cam, obs_pt):
    """The per-landmark cheirality guard of k_backsub: a landmark step that would take the depth of the landmark in one
    of its cameras (at the CURRENT poses) below GUARD_KEEP times its value is shortened to the point where it does.
    Returns the shortened steps and how many were shortened."""
    r3 = quat_to_rot(poses[:, 3:7])[:, :, 2][obs_cam]  # the cameras' optical axes in the world frame
    z0 = np.sum((points[obs_pt] - poses[obs_cam, :3]) * r3, axis=1)
    dz = np.sum(dp[obs_pt] * r3, axis=1)
    hit = (z0 > 0) & (z0 + dz < GUARD_KEEP * z0)
    a = np.ones(len(z0))
    a[hit] = (1.0 - GUARD_KEEP) * z0[hit] / -dz[hit]
    alpha = np.ones(len(points))
    np.minimum.at(alpha, obs_pt, a)
    return dp * alpha[:, None], int(np.sum(alpha < 1.0))


def block_jacobi_pcg(S, b, tol, max_iter, C):
    """Preconditioned CG on S x = b with 6x6 block-Jacobi; stops on sqrt(r^T z) <= tol * sqrt(r0^T z0)."""
    Sd = S.toarray() if sp.issparse(S) and S.shape[0] <= 6000 else None
    if sp.issparse(S):
        diag_blocks = np.stack([S[6 * c:6 * c + 6, 6 * c:6 * c + 6].toarray() for c in range(C)])
    else:
        diag_blocks = np.stack([S[6 * c:6 * c + 6, 6 * c:6 * c + 6] for c in range(C)])
    del Sd
    Minv = np.linalg.inv(diag_blocks)
    x = np.zeros_like(b)
    r = b.copy()
    z = np.einsum("cij,cj->ci", Minv, r.reshape(C, 6)).reshape(-1)
    p = z.copy()
    rz = r @ z
    rz0 = rz
    it = 0
    while it < max_iter and rz > tol * tol * rz0 and rz > 0:
        q = S @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = np.einsum("cij,cj->ci", Minv, r.reshape(C, 6)).reshape(-1)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
    return x, it, r


# ----------------------------------------------------------------------------- pose prior (sliding-window marginalisation)
@dataclass
class PosePrior:
    """E = c0 + g^T d + 0.5 d^T H d over the chart coordinates d of `cams` around `ref` (include/ba_amd.h,
    ba_set_pose_prior): d_a = [t - tref ; 2 vec(qref^-1 q) / w(qref^-1 q)], the inverse of `retract` at ref."""
    cams: np.ndarray  # (n,)
    ref: np.ndarray   # (n, 7)
    H: np.ndarray     # (6n, 6n)
    g: np.ndarray     # (6n,)
    c0: float = 0.0


def prior_chart(poses, ref):
    """Chart coordinates d (n,6) of `poses` around `ref` and the Jacobians d d / d step (n,6,6) of the retraction
    (identity on the translation; I + [a]x + a a^T on the rotation, a = vec / w)."""
    qc = ref[:, 3:7] * np.array([1.0, -1.0, -1.0, -1.0])
    qrel = quat_mul(qc, poses[:, 3:7])
    a = qrel[:, 1:] / qrel[:, :1]
    d = np.concatenate([poses[:, :3] - ref[:, :3], 2.0 * a], axis=1)
    J = np.zeros((len(poses), 6, 6))
    J[:, :3, :3] = np.eye(3)
    ax = np.zeros((len(poses), 3, 3))
    ax[:, 0, 1] = -a[:, 2]; ax[:, 0, 2] = a[:, 1]
    ax[:, 1, 0] = a[:, 2]; ax[:, 1, 2] = -a[:, 0]
    ax[:, 2, 0] = -a[:, 1]; ax[:, 2, 1] = a[:, 0]
    J[:, 3:, 3:] = np.eye(3) + ax + a[:, :, None] * a[:, None, :]
    return d, J


def prior_terms(poses, prior: PosePrior):
    """(energy, gradient (6n,), Gauss-Newton Hessian (6n,6n)) of the prior in the step coordinates at `poses`."""
    d, J = prior_chart(poses[prior.cams], prior.ref)
    dv = d.reshape(-1)
    v = prior.H @ dv + prior.g
    n = len(prior.cams)
    Jb = np.zeros((6 * n, 6 * n))
    for a in range(n):
        Jb[6 * a:6 * a + 6, 6 * a:6 * a + 6] = J[a]
    energy = prior.c0 + float(dv @ (0.5 * (v + prior.g)))
    return energy, Jb.T @ v, Jb.T @ prior.H @ Jb


def marginalize(S, b, c, drop_rows):
    """Schur complement of a quadratic c + b^T x + 0.5 x^T S x over the entries `drop_rows` (minimised out):
    returns (S', b', c') over the remaining entries, in their order."""
    n = len(b)
    drop = np.zeros(n, bool)
    drop[drop_rows] = True
    k, r = np.flatnonzero(~drop), np.flatnonzero(drop)
    Srr = S[np.ix_(r, r)]
    Skr = S[np.ix_(k, r)]
    sol = np.linalg.solve(Srr, np.column_stack([Skr.T, b[r]]))
    return S[np.ix_(k, k)] - Skr @ sol[:, :-1], b[k] - Skr @ sol[:, -1], c - 0.5 * float(b[r] @ sol[:, -1])


def window_prior(poses, points, obs_cam, obs_pt, obs_uv, intr, free, drop_cams, prior: Optional["PosePrior"] = None):
    """What a sliding window keeps of the cameras `drop_cams` it drops: the observations of
    every landmark that a dropped camera sees (by all cameras of the window) plus the window's own prior, linearised at
    (poses, points), with those landmarks and the dropped cameras minimised out.  A dropped camera that is fixed is a
    constant: conditioned on, not eliminated.  Returns (PosePrior over the kept free cameras that the factor touches,
    in camera order; the landmark rows that went into it) or (None, rows) when nothing couples."""
    C = len(poses)
    drop = np.zeros(C, bool)
    drop[drop_cams] = True
    lm_drop = np.zeros(len(points), bool)
    lm_drop[obs_pt[drop[obs_cam]]] = True
    sel = lm_drop[obs_pt]
    oc, op, uv = obs_cam[sel], obs_pt[sel], obs_uv[sel]
    rows = np.flatnonzero(lm_drop)
    remap = np.full(len(points), -1)
    remap[rows] = np.arange(len(rows))
    r, Jc, Jp = linearize(poses, points, oc, op, uv, intr)
    Jc = Jc * free[oc][:, None, None]
    U, V, W, gc, gp, cost = normal_blocks(r, Jc, Jp, oc, remap[op], C, len(rows))
    S, rhs, Vi = schur_system(U, V, W, gc, gp, oc, remap[op], 0.0, free, 0.0)
    c = cost - 0.5 * float(np.einsum("li,lij,lj->", gp, Vi, gp))
    fm = np.repeat(free, 6)
    S[~fm, ~fm] = 0.0  # schur_system's identity rows of the fixed cameras
    if prior is not None:
        e, gj, Hj = prior_terms(poses, prior)
        idx = (6 * prior.cams[:, None] + np.arange(6)[None, :]).reshape(-1)
        S[np.ix_(idx, idx)] += Hj
        rhs[idx] += gj
        c += e
    touched = np.array([np.any(S[6 * k:6 * k + 6] != 0.0) or np.any(rhs[6 * k:6 * k + 6] != 0.0) for k in range(C)])
    elim = np.flatnonzero(drop & free & touched)
    keep = np.flatnonzero(~drop & free & touched)
    if len(keep) == 0:
        return None, rows
    use = np.concatenate([keep, elim])
    idx = (6 * use[:, None] + np.arange(6)[None, :]).reshape(-1)
    S2, b2 = S[np.ix_(idx, idx)], rhs[idx]
    if len(elim):
        S2, b2, c = marginalize(S2, b2, c, np.arange(6 * len(keep), 6 * len(use)))
    S2 = 0.5 * (S2 + S2.T)
    return PosePrior(cams=keep, ref=poses[keep].copy(), H=S2, g=b2, c0=c), rows


# ----------------------------------------------------------------------------- LM driver
@dataclass
class LMOptions:
    max_iters: int = 100
    ftol: float = 1e-10
    xtol: float = 1e-12
    gtol: float = 1e-10
    lambda0: float = 1e-4
    lambda_min: float = 1e-12
    lambda_max: float = 1e12
    min_diag: float = 1e-12
    accept_ratio: float = 1e-4
    guard_cheirality: bool = True  # shorten landmark steps that cross a camera's principal plane; reject a step that
    #                                still puts more observations behind their camera
    linear_solver: str = "direct"  # "direct" | "pcg"
    pcg_tol: float = 1e-2
    pcg_max_iter: int = 500
    loss: str = "linear"  # scipy's names: linear | soft_l1 | huber | cauchy | arctan
    f_scale: float = 1.0


@dataclass
class LMReport:
    cost0: float = 0.0
    cost: float = 0.0
    iterations: int = 0  # linearisations
    accepted: int = 0
    status: int = 0  # 1 gtol, 2 ftol, 3 xtol, 0 max_iters
    costs: List[float] = field(default_factory=list)
    lambdas: List[float] = field(default_factory=list)
    pcg_iters: List[int] = field(default_factory=list)
    grad_inf: float = 0.0
    shortened_steps: int = 0  # landmark steps shortened by the per-landmark cheirality guard, summed over iterations
    guard_rejections: int = 0


def lm_solve(poses, points, obs_cam, obs_pt, obs_uv, intr, free: Optional[np.ndarray] = None,
             opt: Optional[LMOptions] = None, uniform_tracks: bool = False, prior: Optional[PosePrior] = None):
    """Levenberg-Marquardt with Marquardt diagonal scaling, Schur elimination of the points,
    Nielsen's lambda update.  Mirrors the control flow of csrc/ba_solver.hip::ba_solve."""
    opt = opt or LMOptions()
    C, L = len(poses), len(points)
    if free is None:
        free = np.ones(C, dtype=bool)
        free[0] = False
    poses = poses.copy()
    points = points.copy()
    rep = LMReport()
    lam, nu = opt.lambda0, 2.0
    r, Jc, Jp = linearize(poses, points, obs_cam, obs_pt, obs_uv, intr)
    Jc = Jc * free[obs_cam][:, None, None]
    r, Jc, Jp, cost = robustify(r, Jc, Jp, opt.loss, opt.f_scale)
    U, V, W, gc, gp, _ = normal_blocks(r, Jc, Jp, obs_cam, obs_pt, C, L)
    Hoff = None  # the prior's camera-to-camera blocks (its diagonal blocks go into U, its gradient into gc)

    def add_prior(U, gc):
        n = len(prior.cams)
        e, gj, Hj = prior_terms(poses, prior)
        Hoff = np.zeros((6 * C, 6 * C))
        idx = (6 * prior.cams[:, None] + np.arange(6)[None, :]).reshape(-1)
        Hoff[np.ix_(idx, idx)] = Hj
        for a in range(n):
            ca = prior.cams[a]
            U[ca] += Hj[6 * a:6 * a + 6, 6 * a:6 * a + 6]
            gc[ca] += gj[6 * a:6 * a + 6]
            Hoff[6 * ca:6 * ca + 6, 6 * ca:6 * ca + 6] = 0.0
        return e, Hoff

    if prior is not None:
        assert np.all(free[prior.cams]), "the prior's cameras must be free"
        e, Hoff = add_prior(U, gc)
        cost += e
    rep.cost0 = cost
    rep.costs.append(cost)
    _, nbehind, _ = linearize(poses, points, obs_cam, obs_pt, obs_uv, intr, want_jac=False)
    it = 0
    while it < opt.max_iters:
        it += 1
        ginf = max(np.abs(gc).max(), np.abs(gp).max())
        rep.grad_inf = ginf
        if ginf <= opt.gtol:
            rep.status = 1
            break
        if uniform_tracks or C * 6 > 1500:
            S, rhs, Vi = schur_system_sparse(U, V, W, gc, gp, obs_cam, obs_pt, lam, free, opt.min_diag)
        else:
            S, rhs, Vi = schur_system(U, V, W, gc, gp, obs_cam, obs_pt, lam, free, opt.min_diag)
        if Hoff is not None:
            S = S + (sp.csr_matrix(Hoff) if sp.issparse(S) else Hoff)
        if opt.linear_solver == "pcg":
            dcv, nit, _ = block_jacobi_pcg(S, -rhs, opt.pcg_tol, opt.pcg_max_iter, C)
            rep.pcg_iters.append(nit)
        elif sp.issparse(S):
            dcv = spla.spsolve(S.tocsc(), -rhs)
        else:
            dcv = np.linalg.solve(S, -rhs)
        dc = dcv.reshape(C, 6) * free[:, None]
        dp = back_substitute(dc, Vi, W, gp, obs_cam, obs_pt, L)
        if opt.guard_cheirality:  # the model decrease below is that of the step actually taken
            dp, n_short = guard_landmark_steps(poses, points, dp, obs_cam, obs_pt)
            rep.shortened_steps += n_short
        # predicted decrease of 0.5|r + J d|^2: -(g.d) - 0.5 d^T H d, from the blocks
        gd = float(np.sum(gc * dc) + np.sum(gp * dp))
        Wdp = np.einsum("nij,nj->ni", W, dp[obs_pt])
        dHd = float(np.einsum("ci,cij,cj->", dc, U, dc) + np.einsum("li,lij,lj->", dp, V, dp)
                    + 2.0 * np.sum(dc[obs_cam] * Wdp))
        if Hoff is not None:
            dHd += float(dc.reshape(-1) @ Hoff @ dc.reshape(-1))
        pred = -gd - 0.5 * dHd
        poses_t = retract(poses, dc, free)
        points_t = points + dp
        rt, nbehind_t, _ = linearize(poses_t, points_t, obs_cam, obs_pt, obs_uv, intr, want_jac=False)
        cost_t = robustify(rt, None, None, opt.loss, opt.f_scale)[3]
        if prior is not None:
            dt_, _ = prior_chart(poses_t[prior.cams], prior.ref)
            dt_ = dt_.reshape(-1)
            cost_t += prior.c0 + float(dt_ @ (prior.g + 0.5 * (prior.H @ dt_)))
        rho = (cost - cost_t) / pred if pred > 0 else -1.0
        if opt.guard_cheirality and nbehind_t > nbehind:
            rho = -1.0
            rep.guard_rejections += 1
        step_norm = np.sqrt(np.sum(dc * dc) + np.sum(dp * dp))
        x_norm = np.sqrt(np.sum(poses[:, :3] ** 2) + np.sum(points ** 2) + C)
        rep.lambdas.append(lam)
        if np.isfinite(cost_t) and rho > opt.accept_ratio:
            dF = cost - cost_t
            poses, points, cost, nbehind = poses_t, points_t, cost_t, nbehind_t
            rep.accepted += 1
            rep.costs.append(cost)
            lam = max(opt.lambda_min, lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            nu = 2.0
            if dF <= opt.ftol * cost:
                rep.status = 2
                break
            if step_norm <= opt.xtol * (x_norm + opt.xtol):
                rep.status = 3
                break
            r, Jc, Jp = linearize(poses, points, obs_cam, obs_pt, obs_uv, intr)
            Jc = Jc * free[obs_cam][:, None, None]
            r, Jc, Jp, _ = robustify(r, Jc, Jp, opt.loss, opt.f_scale)
            U, V, W, gc, gp, _ = normal_blocks(r, Jc, Jp, obs_cam, obs_pt, C, L)
            if prior is not None:
                _, Hoff = add_prior(U, gc)
        else:
            lam = min(opt.lambda_max, lam * nu)
            nu *= 2.0
            if step_norm <= opt.xtol * (x_norm + opt.xtol):
                rep.status = 3
                break
    rep.cost = cost
    rep.iterations = it
    return poses, points, rep
