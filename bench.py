#!/usr/bin/env python3
"""bench.py -- observations/sec per LM iteration + time-to-converge of the MI355X BA solver.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|small]

A "step" is one complete solve (ba_solve_resident) of the synthetic ring scene from the same initial
guess, with observations, poses and points already resident in HBM; value = observations x LM
iterations summed over the K timed steps / wall time of those steps (max over ranks).  N = 1 runs
BASELINE.json configs[1] (50 cameras x 20k landmarks x 200k observations).  For N > 1 (launched by
torch.distributed.run, one rank per GPU) the N GPUs solve one scene with N times the landmarks and
observations, sharded by landmark track (weak scaling; DESIGN.md "Multi-GPU").  Rank 0 prints ONE
JSON line.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {  # name -> (cameras, landmarks, track length)       
    "small": (50, 2000, 10),
    "cfg2": (50, 20000, 10),
    "cfg3": (1000, 200000, 5),
    "cfg4": (10000, 2000000, 5),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(kernel: str, nobs: int, n_lm: int, n_cam: int) -> float:
    """Bytes one launch must move (DESIGN.md 'Kernels'; SURVEY.md section 8d per-unit figures)."""
    if kernel == "k_linearize":  # read cam idx + u + v, write W; per landmark read P, write V + gp; U/gc per camera
        return 168.0 * nobs + 120.0 * n_lm + 392.0 * n_cam
    if kernel == "k_schur":  # read W + cam idx; per landmark V + gp; S written once
        return 148.0 * nobs + 96.0 * n_lm
    if kernel == "k_backsub":  # read W + cam idx; per landmark V, gp, P read, P written
        return 148.0 * nobs + 144.0 * n_lm
    if kernel == "k_cost":
        return 24.0 * nobs + 24.0 * n_lm
    raise KeyError(kernel)


def cpu_baseline(scene, budget_s: float):
    """The oracle's restatement of the reference path (scipy TRF + LSMR + sparse FD), 1 core."""
    from oracle import ref_scipy_path as orc
    n_poses = int(scene["num_poses"])
    nobs = len(scene["obs_cam"])
    t0 = time.perf_counter()
    res, t_lsq, t_mask = orc.solve_scipy_path(scene["x0"], scene["pose0"], n_poses, scene["obs_cam"],
                                              scene["obs_pt"], scene["obs_uv"], scene["intr"])
    wall = time.perf_counter() - t0
    return {
        "value": nobs * res.nfev / wall, "unit": "observations*iterations/s", "cores": 1, "kind": "port",
        "sample": f"full solve of the same scene by oracle/ref_scipy_path.solve_scipy_path "
                  f"(scipy {__import__('scipy').__version__} least_squares trf+lsmr, jac_sparsity, tolerances 1e-6): "
                  f"{res.nfev} nfev in {wall:.1f} s, status {res.status}",
        "seconds": wall, "cost": float(res.cost), "nfev": int(res.nfev), "status": int(res.status),
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lm-check-every", type=int, default=None, help="override ba_options.lm_check_every")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        # control plane only (id exchange, barriers, max over ranks of the elapsed time): gloo on the CPU.
        # The data path is the solver's own RCCL communicator (ba_comm_init), one rank per GPU.
        import torch.distributed as dist
        dist.init_process_group("gloo")
    import scenes
    import sharding
    from src import _native

    # Weak scaling: N GPUs solve ONE scene N times the size (same cameras, N x the landmarks and
    # observations), sharded by landmark track; one RCCL all-reduce of the reduced system per solve.
    C, L1, d = WORKLOADS[args.workload]
    L = L1 * world
    scene = scenes.ring_scene(C, L, d, seed=0)
    extra = {} if args.lm_check_every is None else {"lm_check_every": args.lm_check_every}
    # BA_BENCH_DEVICE: rehearsal knob only (several ranks on the one-GPU test box, with BA_RCCL_LIBRARY naming
    # tests/fake_rccl/librccl.so); the driver's runs use one GPU per rank
    device = int(os.environ.get("BA_BENCH_DEVICE", local_rank))
    solver = _native.Solver(device=device, profile_kernels=0, **extra)
    if dist is not None:
        uid = [_native.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        solver.comm_init(rank, world, uid[0])
        sh, _ = sharding.shard_arrays(scene["obs_cam"], scene["obs_pt"], scene["obs_uv"], scene["points_guess"],
                                      rank, world)
        solver.set_problem(C, len(sh["points"]), sh["obs_cam"], sh["obs_pt"], sh["obs_uv"], scene["intr"])
        solver.upload_guess(scene["poses_guess"], sh["points"])
        nobs = len(sh["obs_cam"])
        L_local = len(sh["points"])
    else:
        solver.set_problem(C, L, scene["obs_cam"], scene["obs_pt"], scene["obs_uv"], scene["intr"])
        solver.upload_guess(scene["poses_guess"], scene["points_guess"])
        nobs = len(scene["obs_cam"])
        L_local = L

    def barrier():
        # solve_resident() returns after the solver's stream is idle, so the device is quiet here; the
        # torch.cuda.synchronize() of the bench contract is kept for form (torch may not own a context at all)
        if dist is not None:
            try:
                import torch
                if torch.cuda.is_initialized():
                    torch.cuda.synchronize()
            except Exception:
                pass
            dist.barrier()

    for _ in range(args.warmup):
        rep = solver.solve_resident()
    barrier()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.steps):
        rep = solver.solve_resident()  # synchronous: returns after the stream is idle
        iters += rep["iterations"]
    barrier()
    elapsed = time.perf_counter() - t0
    # second pass, untimed: the same K solves with a start/stop event pair attached to every kernel launch
    # (hipExtLaunchKernelGGL, on the solver's stream) for the per-kernel durations behind `roofline`
    k0 = solver.kernel_times()
    if not args.no_kernel_events:
        solver.set_kernel_timing(True)
        for _ in range(args.steps):
            solver.solve_resident()
        solver.set_kernel_timing(False)
    k1 = solver.kernel_times()

    units = float(nobs) * iters
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([units], dtype=torch.float64)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        units = float(u.item())

    if rank == 0:
        ktimes = {k: ((k1[k][0] - k0[k][0]), (k1[k][1] - k0[k][1])) for k in k1}
        big = {k: v for k, v in ktimes.items() if k in ("k_linearize", "k_schur", "k_backsub", "k_cost") and v[1] > 0}
        roofline = None
        traffic_db = {}
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and args.workload == "cfg2" and world == 1:
            with open(tpath) as fh:  # HBM bytes per launch from rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py)
                traffic_db = json.load(fh).get("kernels", {})
        if big:
            dom = max(big, key=lambda k: big[k][0])
            avg_ms = big[dom][0] / big[dom][1]
            bytes_launch = algorithmic_bytes(dom, nobs, L_local, C)
            ach = bytes_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic_db.get(dom, {}).get("traffic_bytes"),
                        "avg_kernel_ms": avg_ms,
                        "algorithmic_bytes_per_launch": bytes_launch,
                        "note": "dominant of the four HBM-streaming kernels; 'pcg' (the reduced camera solve, one "
                                "workgroup, LDS/latency-bound, no HBM traffic in its loop) and 'other' (S gather, "
                                "preconditioner, LM control) are listed in per_kernel without a bandwidth figure",
                        "per_kernel": {k: {"total_ms": v[0], "launches": v[1],
                                           "avg_us": 1e3 * v[0] / max(v[1], 1),
                                           "GBps": (algorithmic_bytes(k, nobs, L_local, C) / (v[0] / v[1] * 1e-3) / 1e9)
                                           if k in big else None}
                                       for k, v in ktimes.items()}}
        out = {
            "metric": "observations/sec per LM iteration", "value": units / elapsed,
            "unit": "observations*iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: ring scene {C} cameras x {L1} landmarks x {L1 * d} observations "
                                   f"(track length {d}), sigma_px 0.5, per GPU",
                       "cameras": C, "landmarks_total": L, "observations_total": L * d,
                       "parallelism": f"landmark-sharded x{world}, RCCL all-reduce of the reduced camera system"
                       if world > 1 else "single GPU"},
            "time_to_converge_ms": rep["solve_ms"], "lm_iterations": rep["iterations"],
            "pcg_iterations": rep["pcg_iters"], "final_cost": rep["cost"], "status": rep["status_name"],
            "rms_reproj_px": rep["rms_reproj"], "setup_ms": rep["setup_ms"],
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(scene, 30.0)
            out["cpu_baseline"] = cb
            out["cost_vs_cpu_baseline"] = rep["cost"] / cb["cost"] - 1.0
        print(json.dumps(out))
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
