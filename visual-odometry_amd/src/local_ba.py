"""Sliding-window (local) bundle adjustment over a keyframe list: the caller of the batched solve
(BASELINE configs[4]).

The reference's visual odometry keeps ``VisualOdometry.keyframes`` (a list of ``Keyframe(timestamp, pose,
camera_matrix, points_2d, points_2d_ids)``, src/visual_odometry.py:13-25,39,395-403) and a map
``points_3d`` / ``points_3d_ids`` (:31-32), but never calls its bundle adjustment on them
(``VisualOdometry.bundle_adjustment`` at :572 is a stub).  This module closes that loop: it cuts the
keyframe list into windows, gathers for every window the map points seen by at least ``min_views`` of its
keyframes, and solves ALL windows in one GPU call as one block-diagonal problem (one dense Cholesky per
window on the reduced camera system, csrc/ba_kernels.h: k_solve_components).  The oldest ``n_fixed``
keyframes of a window are held constant: a hard prior in place of marginalisation (the reference fixes
image 0 the same way, bundle_adjustment.py:51-52).  No CPU fallback.

``SlidingWindowBundleAdjustment`` is the sequential variant with a marginalisation prior: windows are solved in
order, and what a window knows about the keyframes it is about to drop (their observations, and those of every map
point they see) is kept as a quadratic prior on the remaining keyframes (include/ba_amd.h: ba_set_pose_prior); the
map points that went into the prior are frozen.  The first ``n_fixed`` keyframes of the list are held constant wherever
they appear (two by default: six degrees of freedom and the monocular scale; with one, the scale stays a free gauge
that every prior inherits as a null direction, and the chain of windows does not converge).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .camera_pose import CameraPose
from .validation_error import ValidationError


class Keyframe:
    """Same constructor and attributes as the reference's Keyframe (visual_odometry.py:13-25; like the
    reference, camera_matrix is accepted and not stored)."""

    def __init__(self, timestamp: float, pose: CameraPose, camera_matrix: np.ndarray, points_2d: np.ndarray,
                 points_2d_ids: np.ndarray) -> None:
        self.timestamp = timestamp
        self.pose = pose
        self.points_2d = points_2d
        self.points_2d_ids = points_2d_ids


@dataclass
class Window:
    first: int                 # index of the window's first keyframe in the keyframe list
    keyframes: np.ndarray      # indices into the keyframe list
    landmark_ids: np.ndarray   # ids of the map points in the window (ascending)
    landmark_rows: np.ndarray  # their rows in points_3d
    obs_cam: np.ndarray        # int32, local keyframe index
    obs_pt: np.ndarray         # int32, local landmark index
    obs_uv: np.ndarray         # n x 2


def build_windows(keyframes: Sequence[Keyframe], points_3d_ids: np.ndarray, window: int = 10,
                  stride: Optional[int] = None, min_views: int = 2) -> List[Window]:
    """Windows [s, s + window) for s = 0, stride, 2 stride, ... (the last one is aligned to the end of the
    list); a window keeps the map points observed by >= min_views of its keyframes and only their
    observations.  Windows with fewer than 2 keyframes or no such point are skipped."""
    if window < 2:
        raise ValidationError("a window needs at least 2 keyframes")
    stride = window if stride is None else stride
    if stride < 1:
        raise ValidationError("stride must be >= 1")
    ids_map = np.asarray(points_3d_ids)
    order = np.argsort(ids_map, kind="stable")
    sorted_ids = ids_map[order]
    if len(sorted_ids) > 1 and np.any(sorted_ids[1:] == sorted_ids[:-1]):
        raise ValidationError("points_3d_ids holds an id twice")
    K = len(keyframes)
    starts = list(range(0, max(K - window, 0) + 1, stride))
    if K >= 2 and starts and starts[-1] + window < K:
        starts.append(K - window)
    out: List[Window] = []
    for s in starts:
        kfs = np.arange(s, min(s + window, K))
        if len(kfs) < 2:
            continue
        cams, lids, uvs = [], [], []
        for local, k in enumerate(kfs):
            ids = np.asarray(keyframes[k].points_2d_ids)
            uv = np.asarray(keyframes[k].points_2d, dtype=float).reshape(-1, 2)
            if len(ids) != len(uv):
                raise ValidationError(f"keyframe {k}: {len(ids)} ids for {len(uv)} points")
            pos = np.searchsorted(sorted_ids, ids)
            in_map = (pos < len(sorted_ids)) & (sorted_ids[np.minimum(pos, len(sorted_ids) - 1)] == ids) \
                if len(sorted_ids) else np.zeros(len(ids), bool)
            cams.append(np.full(int(in_map.sum()), local, dtype=np.int32))
            lids.append(ids[in_map])
            uvs.append(uv[in_map])
        cams, lids, uvs = np.concatenate(cams), np.concatenate(lids), np.vstack(uvs)
        uniq, inverse, counts = np.unique(lids, return_inverse=True, return_counts=True)
        keep_lm = counts >= min_views
        if not keep_lm.any():
            continue
        keep = keep_lm[inverse]
        new_index = np.cumsum(keep_lm) - 1
        win_ids = uniq[keep_lm]
        rows = order[np.searchsorted(sorted_ids, win_ids)]
        out.append(Window(first=int(s), keyframes=kfs, landmark_ids=win_ids, landmark_rows=rows,
                          obs_cam=cams[keep], obs_pt=new_index[inverse[keep]].astype(np.int32), obs_uv=uvs[keep]))
    return out


class LocalBundleAdjustment:
    def __init__(self, window: int = 10, stride: Optional[int] = None, min_views: int = 2, n_fixed: int = 1,
                 **solver_options) -> None:
        if n_fixed < 1:
            raise ValidationError("at least the oldest keyframe of a window must be fixed (gauge)")
        self.window, self.stride, self.min_views, self.n_fixed = window, stride, min_views, n_fixed
        self.solver_options = solver_options
        self.last_report: Optional[dict] = None
        self.last_windows: List[Window] = []
        self.last_costs: Optional[Tuple[np.ndarray, np.ndarray]] = None

    @staticmethod
    def _intrinsics(camera_matrix: np.ndarray) -> np.ndarray:
        Kc = np.asarray(camera_matrix, dtype=float)
        if Kc.shape != (3, 3):
            raise ValidationError(f"camera matrix must be 3x3, got {Kc.shape}")
        return np.array([Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2], 0.0, 0.0, 0.0])

    def optimize(self, keyframes: Sequence[Keyframe], points_3d: np.ndarray, points_3d_ids: np.ndarray,
                 camera_matrix: np.ndarray, intrinsics: Optional[np.ndarray] = None
                 ) -> Tuple[List[CameraPose], np.ndarray]:
        """Solve every window in one GPU call.  Returns (poses, points_3d): a new CameraPose per keyframe
        (timestamps kept) and a copy of points_3d with the refined rows.  Where windows overlap, the
        newest window's estimate of a keyframe / map point wins; keyframes fixed in the window that
        would overwrite them keep the value they had."""
        points_3d = np.array(points_3d, dtype=float).reshape(-1, 3)
        intr = self._intrinsics(camera_matrix) if intrinsics is None else np.asarray(intrinsics, dtype=float)
        wins = build_windows(keyframes, points_3d_ids, self.window, self.stride, self.min_views)
        self.last_windows = wins
        poses_all = np.array([np.concatenate([k.pose.position, k.pose.quaternion]) for k in keyframes]).reshape(-1, 7)
        if not wins:
            self.last_report, self.last_costs = None, None
            return [k.pose.copy() for k in keyframes], points_3d
        cam_off = np.cumsum([0] + [len(w.keyframes) for w in wins])
        lm_off = np.cumsum([0] + [len(w.landmark_ids) for w in wins])
        obs_off = np.cumsum([0] + [len(w.obs_cam) for w in wins])
        obs_cam = np.concatenate([w.obs_cam + cam_off[b] for b, w in enumerate(wins)]).astype(np.int32)
        obs_pt = np.concatenate([w.obs_pt + lm_off[b] for b, w in enumerate(wins)]).astype(np.int32)
        obs_uv = np.vstack([w.obs_uv for w in wins])
        poses = np.vstack([poses_all[w.keyframes] for w in wins])
        pts = np.vstack([points_3d[w.landmark_rows] for w in wins])
        fixed = np.zeros(cam_off[-1], dtype=np.uint8)
        for b, w in enumerate(wins):
            fixed[cam_off[b]: cam_off[b] + min(self.n_fixed, len(w.keyframes) - 1)] = 1
        with _native.Solver(**self.solver_options) as s:
            s.set_problem(int(cam_off[-1]), int(lm_off[-1]), obs_cam, obs_pt, obs_uv, intr, fixed=fixed)
            r0, _, _ = s.evaluate(poses, pts)
            P, X, rep = s.solve(poses, pts)
            r1, _, _ = s.evaluate(P, X)
        per = lambda r: np.array([0.5 * float(np.sum(r[2 * obs_off[b]: 2 * obs_off[b + 1]] ** 2)) for b in range(len(wins))])
        self.last_report, self.last_costs = rep, (per(r0), per(r1))
        out_pts = points_3d.copy()
        out_poses = poses_all.copy()
        for b, w in enumerate(wins):  # ascending start: newer windows overwrite older ones
            free = fixed[cam_off[b]: cam_off[b + 1]] == 0
            out_poses[w.keyframes[free]] = P[cam_off[b]: cam_off[b + 1]][free]
            out_pts[w.landmark_rows] = X[lm_off[b]: lm_off[b + 1]]
        return ([CameraPose(position=p[:3], orientation=p[3:7], timestamp=k.timestamp)
                 for p, k in zip(out_poses, keyframes)], out_pts)


def _chart(poses: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """Coordinates of `poses` in the chart of the solver's retraction around `ref` (ba_set_pose_prior's d), n x 6."""
    rw, rv = ref[:, 3:4], ref[:, 4:7]
    qw, qv = poses[:, 3:4], poses[:, 4:7]
    w = rw * qw + np.sum(rv * qv, axis=1, keepdims=True)          # conj(ref) (x) q
    v = rw * qv - qw * rv - np.cross(rv, qv)
    return np.hstack([poses[:, :3] - ref[:, :3], 2.0 * v / w])


@dataclass
class WindowPrior:
    keyframes: np.ndarray  # indices into the keyframe list
    ref: np.ndarray        # n x 7 chart origins (the estimates at marginalisation time)
    H: np.ndarray          # 6n x 6n
    g: np.ndarray          # 6n
    c0: float

    def energy(self, poses: np.ndarray) -> float:
        d = _chart(poses, self.ref).reshape(-1)
        return self.c0 + float(d @ (self.g + 0.5 * (self.H @ d)))


class SlidingWindowBundleAdjustment:
    """Fixed-lag smoothing over a keyframe list: windows [s, s + window), s = 0, stride, 2 stride, ..., solved one
    after the other on the GPU; the keyframes that leave a window are marginalised into a prior on the ones that stay."""

    def __init__(self, window: int = 10, stride: int = 1, min_views: int = 2, n_fixed: int = 2,
                 **solver_options) -> None:
        if window - stride > _native.Solver.PRIOR_MAX_CAMS:
            raise ValidationError(f"a prior holds at most {_native.Solver.PRIOR_MAX_CAMS} keyframes")
        if n_fixed < 1 or n_fixed >= window:
            raise ValidationError("between 1 and window - 1 keyframes at the head of the list must be fixed (gauge)")
        self.window, self.stride, self.min_views, self.n_fixed = window, stride, min_views, n_fixed
        self.solver_options = solver_options
        self.last_windows: List[Window] = []
        self.last_reports: List[dict] = []
        self.last_priors: List[Optional[WindowPrior]] = []  # the prior each window was solved WITH
        self.last_frozen: List[np.ndarray] = []             # map rows frozen after each window

    def _marginalize(self, w: Window, P: np.ndarray, X: np.ndarray, intr: np.ndarray, fixed: np.ndarray,
                     prior: Optional[WindowPrior], n_drop: int) -> Tuple[Optional[WindowPrior], np.ndarray]:
        """The factor that the first n_drop keyframes of window w leave behind, at the window's estimate (P, X).  The
        reduced camera system of the sub-problem (the map points those keyframes see + the window's own prior) comes
        from the GPU (lambda = 0); the dropped free keyframes are then eliminated from it here (a 6n x 6n system)."""
        C = len(w.keyframes)
        drop = np.zeros(C, bool)
        drop[:n_drop] = True
        lm_drop = np.zeros(len(w.landmark_ids), bool)
        lm_drop[w.obs_pt[drop[w.obs_cam]]] = True
        rows = np.flatnonzero(lm_drop)
        if len(rows) == 0:
            return None, rows
        sel = lm_drop[w.obs_pt]
        remap = np.full(len(lm_drop), -1, dtype=np.int32)
        remap[rows] = np.arange(len(rows), dtype=np.int32)
        with _native.Solver(**self.solver_options) as s:
            if prior is not None:
                local = np.searchsorted(w.keyframes, prior.keyframes)
                s.set_pose_prior(local, prior.ref, prior.H, prior.g, prior.c0)
            s.set_problem(C, len(rows), w.obs_cam[sel], remap[w.obs_pt[sel]], w.obs_uv[sel], intr, fixed=fixed)
            lin = s.debug_linearization(P, X[rows], 0.0)
            _, cost, _ = s.evaluate(P, X[rows], want_residuals=False)
        S, b = lin["S"], lin["rhs"]
        free = fixed == 0
        fm = np.repeat(free, 6)
        S[np.ix_(~fm, ~fm)] = 0.0  # the export's identity rows of the fixed keyframes
        c = cost - 0.5 * float(np.einsum("li,li->", lin["gp"], np.linalg.solve(lin["V"], lin["gp"][:, :, None])[:, :, 0]))
        if prior is not None:
            c += prior.energy(P[np.searchsorted(w.keyframes, prior.keyframes)])
        blocks = np.abs(S).reshape(C, 6, C, 6).sum(axis=(1, 2, 3)) + np.abs(b).reshape(C, 6).sum(axis=1)
        touched = blocks > 0.0
        keep = np.flatnonzero(~drop & free & touched)
        elim = np.flatnonzero(drop & free & touched)
        if len(keep) == 0:
            return None, rows
        ik = (6 * keep[:, None] + np.arange(6)).reshape(-1)
        H, g = S[np.ix_(ik, ik)], b[ik]
        if len(elim):
            ie = (6 * elim[:, None] + np.arange(6)).reshape(-1)
            sol = np.linalg.solve(S[np.ix_(ie, ie)], np.column_stack([S[np.ix_(ie, ik)], b[ie]]))
            H = H - S[np.ix_(ik, ie)] @ sol[:, :-1]
            g = g - S[np.ix_(ik, ie)] @ sol[:, -1]
            c -= 0.5 * float(b[ie] @ sol[:, -1])
        return WindowPrior(keyframes=w.keyframes[keep], ref=P[keep].copy(), H=0.5 * (H + H.T), g=g, c0=c), rows

    def optimize(self, keyframes: Sequence[Keyframe], points_3d: np.ndarray, points_3d_ids: np.ndarray,
                 camera_matrix: np.ndarray, intrinsics: Optional[np.ndarray] = None
                 ) -> Tuple[List[CameraPose], np.ndarray]:
        points = np.array(points_3d, dtype=float).reshape(-1, 3)
        intr = LocalBundleAdjustment._intrinsics(camera_matrix) if intrinsics is None else np.asarray(intrinsics, dtype=float)
        wins = build_windows(keyframes, points_3d_ids, self.window, self.stride, self.min_views)
        # the keyframes that a window hands on to the next one carry the prior: check every actual shift BEFORE anything
        # is solved (the last window is aligned to the end of the list and may shift by fewer than `stride` keyframes)
        for w, nxt in zip(wins, wins[1:]):
            kept = int(np.sum(w.keyframes >= nxt.first))
            if kept > _native.Solver.PRIOR_MAX_CAMS:
                raise ValidationError(f"the window starting at keyframe {w.first} hands {kept} keyframes on to the next one; "
                                      f"a prior holds at most {_native.Solver.PRIOR_MAX_CAMS}")
        poses = np.array([np.concatenate([k.pose.position, k.pose.quaternion]) for k in keyframes]).reshape(-1, 7)
        poses[:, 3:7] /= np.linalg.norm(poses[:, 3:7], axis=1, keepdims=True)
        frozen = np.zeros(len(points), bool)
        prior: Optional[WindowPrior] = None
        self.last_windows, self.last_reports, self.last_priors, self.last_frozen = [], [], [], []
        for b, w in enumerate(wins):
            # the window without the frozen map points (their information is in the prior)
            live = ~frozen[w.landmark_rows]
            sel = live[w.obs_pt]
            remap = (np.cumsum(live) - 1).astype(np.int32)
            w = Window(first=w.first, keyframes=w.keyframes, landmark_ids=w.landmark_ids[live],
                       landmark_rows=w.landmark_rows[live], obs_cam=w.obs_cam[sel], obs_pt=remap[w.obs_pt[sel]],
                       obs_uv=w.obs_uv[sel])
            C = len(w.keyframes)
            fixed = (w.keyframes < self.n_fixed).astype(np.uint8)
            if prior is None and not fixed.any():
                fixed[0] = 1  # nothing anchors this window (no map point links it to the ones before)
            with _native.Solver(**self.solver_options) as s:
                if prior is not None:
                    s.set_pose_prior(np.searchsorted(w.keyframes, prior.keyframes), prior.ref, prior.H, prior.g, prior.c0)
                s.set_problem(C, len(w.landmark_ids), w.obs_cam, w.obs_pt, w.obs_uv, intr, fixed=fixed)
                P, X, rep = s.solve(poses[w.keyframes], points[w.landmark_rows])
            poses[w.keyframes] = P
            points[w.landmark_rows] = X
            self.last_windows.append(w)
            self.last_reports.append(rep)
            self.last_priors.append(prior)
            if b + 1 < len(wins):
                n_drop = int(np.searchsorted(w.keyframes, wins[b + 1].first))
                new_prior, rows = self._marginalize(w, P, X, intr, fixed, prior, n_drop)
                frozen[w.landmark_rows[rows]] = True
                prior = new_prior
            self.last_frozen.append(np.flatnonzero(frozen))
        return ([CameraPose(position=p[:3], orientation=p[3:7], timestamp=k.timestamp)
                 for p, k in zip(poses, keyframes)], points)
