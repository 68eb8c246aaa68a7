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


# ----------------------------------------------------------------------------------------------------------------------
# The reduced solve sharded by camera segment (BA_SHARDED_SOLVE=1, csrc/ba_solver.hip do_reduced_solve): the same five
# exchanges on the CPU with gloo.  Cameras on a ring with short tracks give a cyclic banded S; P segments, each led by a
# separator of `bw` cameras; rank t owns the segments [t P / world, (t + 1) P / world).
def _segments(Cf, P, bw):
    sp = [q * Cf // P for q in range(P + 1)]
    sep = [np.arange(sp[q], sp[q] + bw) for q in range(P)]
    interior = [np.arange(sp[q] + bw, sp[q + 1]) for q in range(P)]
    return sep, interior


def _rows(cams):
    return (6 * np.asarray(cams)[:, None] + np.arange(6)[None, :]).reshape(-1)


def _worker_segments(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C, d, P, lam = 41, 3, 4, 1e-3
        bw = d - 1
        sc = scenes.ring_scene(C, 20 * C, d)
        L = len(sc["points_guess"])
        free = np.ones(C, bool)
        free[0] = False
        sh, _ = sharding.shard_arrays(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["points_guess"], rank, world)
        Lr = len(sh["points"])
        r, Jc, Jp = lm.linearize(sc["poses_guess"], sh["points"], sh["obs_cam"], sh["obs_pt"], sh["obs_uv"], sc["intr"])
        Jc = Jc * free[sh["obs_cam"]][:, None, None]
        U, V, W, gc, gp, cost = lm.normal_blocks(r, Jc, Jp, sh["obs_cam"], sh["obs_pt"], C, Lr)
        S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sh["obs_cam"], sh["obs_pt"], lam, free)
        Ud = np.concatenate([np.maximum(np.diag(U[c]), 1e-12) for c in range(C)])
        fr = _rows(np.arange(1, C))                       # rows of the free cameras (camera 0 is fixed)
        Sp = S[np.ix_(fr, fr)].copy()                     # this rank's PARTIAL system over the free cameras ...
        Sp[np.arange(len(fr)), np.arange(len(fr))] -= lam * Ud[fr]   # ... without the damping it added locally
        # (1) all-reduce of [rhs | diag U] only
        small = torch.from_numpy(np.concatenate([rhs[fr], Ud[fr]]))
        dist.all_reduce(small)
        rhs_f, Ud_f = small.numpy()[: len(fr)].copy(), small.numpy()[len(fr):].copy()
        Cf = C - 1
        sep, interior = _segments(Cf, P, bw)
        per = P // world
        own = range(rank * per, (rank + 1) * per)
        # (2) + (3): the rows of rank t's cameras are summed onto rank t (one reduce per rank), which then adds the damping
        mine = None
        for t in range(world):
            cams_t = np.concatenate([np.concatenate([sep[q], interior[q]]) for q in range(t * per, (t + 1) * per)])
            block = torch.from_numpy(Sp[_rows(cams_t), :].copy())
            dist.reduce(block, dst=t)
            if t == rank:
                mine = (cams_t, block.numpy())
        cams_mine, rows_mine = mine
        A = np.zeros_like(Sp)                              # complete rows for this rank's cameras, zero elsewhere
        A[_rows(cams_mine), :] = rows_mine
        A[_rows(cams_mine), _rows(cams_mine)] += lam * Ud_f[_rows(cams_mine)]
        # (4) eliminate the interiors of the own segments; what they leave on their separators joins the separator system
        all_sep = np.concatenate(sep)
        sr = _rows(all_sep)
        S2 = np.zeros((len(sr), len(sr)))
        b2 = np.zeros(len(sr))
        pos = {int(c): i for i, c in enumerate(all_sep)}
        own_sep_rows = _rows(np.concatenate([sep[q] for q in own]))
        loc = np.array([6 * pos[int(c // 6)] + c % 6 for c in own_sep_rows])
        S2[loc, :] = A[np.ix_(own_sep_rows, sr)]          # the separator rows this rank owns (complete, damped)
        b2[loc] = -rhs_f[own_sep_rows]
        keep = {}
        for q in own:
            I, B = _rows(interior[q]), _rows(np.concatenate([sep[q], sep[(q + 1) % P]]))
            AII, AIB = A[np.ix_(I, I)], A[np.ix_(I, B)]
            assert np.abs(A[np.ix_(I, np.setdiff1d(np.arange(A.shape[1]), np.concatenate([I, B])))]).max() == 0.0
            X = np.linalg.solve(AII, np.column_stack([AIB, -rhs_f[I]]))
            lb = np.array([6 * pos[int(c // 6)] + c % 6 for c in B])
            S2[np.ix_(lb, lb)] -= AIB.T @ X[:, :-1]
            b2[lb] -= AIB.T @ X[:, -1]
            keep[q] = (I, B, lb, X)
        # (5) all-reduce of the separator system, (6) solved on every rank
        t2 = torch.from_numpy(np.concatenate([S2.reshape(-1), b2]))
        dist.all_reduce(t2)
        t2 = t2.numpy()
        S2f, b2f = t2[: S2.size].reshape(S2.shape), t2[S2.size:]
        x2 = np.linalg.solve(S2f, b2f)
        # (7) own interiors back-substituted into a zeroed step, own left separators copied, the step summed
        dcf = np.zeros(6 * Cf)
        for q in own:
            I, B, lb, X = keep[q]
            dcf[I] = X[:, -1] - X[:, :-1] @ x2[lb]
            dcf[_rows(sep[q])] = x2[[6 * pos[int(c)] + e for c in sep[q] for e in range(6)]]
        t3 = torch.from_numpy(dcf)
        dist.all_reduce(t3)
        out.put((rank, t3.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_reduced_solve_sharded_by_camera_segment():
    world, port = 2, 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker_segments, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    C, d = 41, 3
    sc = scenes.ring_scene(C, 20 * C, d)
    L = len(sc["points_guess"])
    free = np.ones(C, bool)
    free[0] = False
    r, Jc, Jp = lm.linearize(sc["poses_guess"], sc["points_guess"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    Jc = Jc * free[sc["obs_cam"]][:, None, None]
    U, V, W, gc, gp, cost = lm.normal_blocks(r, Jc, Jp, sc["obs_cam"], sc["obs_pt"], C, L)
    S, rhs, Vi = lm.schur_system(U, V, W, gc, gp, sc["obs_cam"], sc["obs_pt"], 1e-3, free)
    dc = np.linalg.solve(S, -rhs)[6:]
    np.testing.assert_array_equal(results[0][1], results[1][1])  # the summed step is the same bits on both ranks
    np.testing.assert_allclose(results[0][1], dc, rtol=0, atol=1e-9 * np.abs(dc).max())
