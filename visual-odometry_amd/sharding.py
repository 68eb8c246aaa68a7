"""Landmark-track sharding for the multi-GPU solve.

Every observation belongs to exactly one landmark, so V, g_p, W, the per-track Schur products,
back-substitution and the per-observation cost are local to the GPU that owns the landmark; camera
poses are replicated.  A shard is a contiguous range of landmark indices, chosen so that the
ranks hold (nearly) equal numbers of observations.  Host logic only -- tested on CPU with gloo
(tests/test_sharding_gloo.py); the device side is csrc/ba_solver.hip (ba_comm_init)."""

from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def landmark_ranges(obs_pt: np.ndarray, n_landmarks: int, world: int) -> np.ndarray:
    """(world + 1,) boundaries b with rank r owning landmarks [b[r], b[r+1]): the cumulative
    observation count is cut at multiples of nobs / world."""
    if world < 1:
        raise ValueError("world must be >= 1")
    counts = np.bincount(np.asarray(obs_pt), minlength=n_landmarks)
    cum = np.concatenate([[0], np.cumsum(counts)])
    targets = np.arange(1, world) * (cum[-1] / world)
    cuts = np.searchsorted(cum, targets, side="left")
    return np.concatenate([[0], np.clip(cuts, 0, n_landmarks), [n_landmarks]]).astype(np.int64)


def shard_arrays(obs_cam, obs_pt, obs_uv, points, rank: int, world: int) -> Tuple[Dict[str, np.ndarray], np.ndarray]:
    """This rank's observations (obs_pt re-indexed to the local landmark array) and local points.
    Returns (shard dict, boundaries)."""
    obs_pt = np.asarray(obs_pt)
    b = landmark_ranges(obs_pt, len(points), world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    sel = (obs_pt >= lo) & (obs_pt < hi)
    shard = {
        "obs_cam": np.ascontiguousarray(np.asarray(obs_cam)[sel], dtype=np.int32),
        "obs_pt": np.ascontiguousarray(obs_pt[sel] - lo, dtype=np.int32),
        "obs_uv": np.ascontiguousarray(np.asarray(obs_uv)[sel]),
        "points": np.ascontiguousarray(np.asarray(points)[lo:hi]),
        "lm_lo": lo, "lm_hi": hi,
    }
    return shard, b
