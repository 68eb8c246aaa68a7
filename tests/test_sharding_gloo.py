"""world_size-2 CPU (gloo) test of the multi-GPU decomposition: each rank linearises ITS landmark
shard with the numpy oracle, the partial reduced camera systems are summed with an all-reduce, and
the result must equal the unsharded system; the ranks then back-substitute their own landmarks.
This is exactly the exchange csrc/ba_solver.hip performs with RCCL: sum of [S | rhs | U, gc]."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes
import sharding
from oracle import lm_schur_numpy as lm


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.ring_scene(20, 400, 5)
        C, L = len(sc["poses_guess"]), len(sc["points_guess"])
        free = np.ones(C, bool)
        free[0] = False
        lam = 1e-3
        sh, b = sharding.shard_arrays(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["points_guess"], rank, world)
        Lr = len(sh["points"])
        r, Jc, Jp = lm.linearize(sc["poses_guess"], sh["points"], sh["obs_cam"], sh["obs_pt"], sh["obs_uv"], sc["intr"])
        Jc = Jc * free[sh["obs_cam"]][:, None, None]
        U, V, W, gc, gp, cost = lm.normal_blocks(r, Jc, Jp, sh["obs_cam"], sh["obs_pt"], C, Lr)
        # partial reduced system WITHOUT damping of U (damping needs the full diagonal of U)
        S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sh["obs_cam"], sh["obs_pt"], lam, free)
        idx = np.arange(6 * C)
        Ud = np.zeros(6 * C)
        for c in range(C):
            Ud[6 * c:6 * c + 6] = np.maximum(np.diag(U[c]), 1e-12)
        fm = np.repeat(free, 6)
        S[idx[fm], idx[fm]] -= lam * Ud[fm]      # undo the local damping
        S[idx[~fm], idx[~fm]] = 0.0              # identity rows are added once, after the sum
        buf = torch.from_numpy(np.concatenate([S.reshape(-1), rhs, Ud, [cost]]))
        dist.all_reduce(buf)                     # the one exchange step per LM iteration
        buf = buf.numpy()
        n2 = (6 * C) ** 2
        S_full = buf[:n2].reshape(6 * C, 6 * C).copy()
        rhs_full, Ud_full, cost_full = buf[n2:n2 + 6 * C], buf[n2 + 6 * C:n2 + 12 * C], buf[-1]
        S_full[idx[fm], idx[fm]] += lam * Ud_full[fm]
        S_full[idx[~fm], idx[~fm]] = 1.0
        dc = np.linalg.solve(S_full, -rhs_full).reshape(C, 6) * free[:, None]
        dp = lm.back_substitute(dc, Vi, W, gp, sh["obs_cam"], sh["obs_pt"], Lr)
        out.put((rank, int(b[rank]), int(b[rank + 1]), dc, dp, cost_full, len(sh["obs_cam"])))
    finally:
        dist.destroy_process_group()


def test_landmark_ranges_balance_and_cover():
    sc = scenes.ring_scene(20, 400, 5)
    for world in (1, 2, 3, 8):
        b = sharding.landmark_ranges(sc["obs_pt"], 400, world)
        assert b[0] == 0 and b[-1] == 400 and np.all(np.diff(b) >= 0)
        counts = [np.sum((sc["obs_pt"] >= b[r]) & (sc["obs_pt"] < b[r + 1])) for r in range(world)]
        assert sum(counts) == len(sc["obs_pt"])
        assert max(counts) - min(counts) <= 5 * 2  # at most a couple of tracks of imbalance
    # ragged tracks, some landmarks unobserved
    rng = np.random.default_rng(0)
    pt = np.sort(rng.integers(0, 50, 300))
    b = sharding.landmark_ranges(pt, 60, 4)
    assert b[-1] == 60 and sum(np.sum((pt >= b[r]) & (pt < b[r + 1])) for r in range(4)) == 300


def test_two_rank_sharded_step_equals_unsharded():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded reference
    sc = scenes.ring_scene(20, 400, 5)
    C, L = len(sc["poses_guess"]), len(sc["points_guess"])
    free = np.ones(C, bool)
    free[0] = False
    r, Jc, Jp = lm.linearize(sc["poses_guess"], sc["points_guess"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, cost = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], C, L)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sc["obs_cam"], sc["obs_pt"], 1e-3, free)
    dc = np.linalg.solve(S, -rhs).reshape(C, 6) * free[:, None]
    dp = lm.back_substitute(dc, Vi, W, gp, sc["obs_cam"], sc["obs_pt"], L)
    assert sum(res[6] for res in results) == len(sc["obs_cam"])
    for rank, lo, hi, dc_r, dp_r, cost_r, _ in results:
        np.testing.assert_allclose(dc_r, dc, rtol=0, atol=1e-9 * np.abs(dc).max())
        np.testing.assert_allclose(dp_r, dp[lo:hi], rtol=0, atol=1e-9 * np.abs(dp).max())
        assert abs(cost_r - cost) <= 1e-12 * cost
