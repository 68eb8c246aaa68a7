"""One rank of tests/test_gpu_two_ranks.py (not collected by pytest).  Started with the test stand-in for
librccl.so named in BA_RCCL_LIBRARY, so that several ranks can share GPU 0; does exactly what bench.py does at
N > 1: join the communicator, take its landmark shard, solve, and write its part of the estimate."""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "visual-odometry_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, workdir, scene_name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    options = {k: (float(v) if "." in v or "e" in v else int(v)) for k, v in (a.split("=") for a in sys.argv[5:])}
    import scenes
    import sharding
    from src import _native

    uid_path = os.path.join(workdir, "uid.bin")
    if rank == 0:
        uid = _native.comm_unique_id()
        with open(uid_path + ".tmp", "wb") as fh:
            fh.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        deadline = time.time() + 60.0
        while not os.path.exists(uid_path):
            if time.time() > deadline:
                raise SystemExit("rank 0 never published the unique id")
            time.sleep(0.01)
        with open(uid_path, "rb") as fh:
            uid = fh.read()

    fault = None
    if ":" in scene_name:  # "<scene>:<fault>": rank 1 hands the library something invalid (tests of the status agreement)
        scene_name, fault = scene_name.split(":")
    if scene_name == "ring":
        sc = scenes.ring_scene(50, 2000, 10)
    elif scene_name == "ring_long":
        sc = scenes.ring_scene(300, 3000, 6)
    else:
        sc = scenes.noisy_scene(8, 400)
    C = len(sc["poses_guess"])
    sh, _ = sharding.shard_arrays(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["points_guess"], rank, world)
    from src.validation_error import ValidationError
    with _native.Solver(device=0, **options) as s:
        s.comm_init(rank, world, uid)
        if fault is not None:
            stage, err = "none", ""
            try:
                if fault == "empty" and rank == 1:  # a shard without observations
                    s.set_problem(C, len(sh["points"]), sh["obs_cam"][:0], sh["obs_pt"][:0], sh["obs_uv"][:0], sc["intr"])
                else:
                    s.set_problem(C, len(sh["points"]), sh["obs_cam"], sh["obs_pt"], sh["obs_uv"], sc["intr"])
                stage = "set_problem"
                pts = sh["points"].copy()
                if fault == "nan" and rank == 1:
                    pts[0, 1] = np.nan
                s.solve(sc["poses_guess"], pts)
                stage = "solve"
            except ValidationError as e:
                err = str(e)
            # after the refused call every rank can still go through a good set_problem + solve together
            s.set_problem(C, len(sh["points"]), sh["obs_cam"], sh["obs_pt"], sh["obs_uv"], sc["intr"])
            _, _, rep = s.solve(sc["poses_guess"], sh["points"])
            np.savez(os.path.join(workdir, f"rank{rank}.npz"), stage=stage, err=err, cost=rep["cost"])
            return
        s.set_problem(C, len(sh["points"]), sh["obs_cam"], sh["obs_pt"], sh["obs_uv"], sc["intr"])
        P, X, rep = s.solve(sc["poses_guess"], sh["points"])
        kind = s.reduced_solver()
        info = s.problem_info()
    np.savez(os.path.join(workdir, f"rank{rank}.npz"), poses=P, points=X, lo=sh["lm_lo"], hi=sh["lm_hi"],
             cost=rep["cost"], cost0=rep["cost0"], iterations=rep["iterations"], status=rep["status"],
             nobs=rep["nobs"], solver=kind, sharded_solve=(info["linearize_flat"] >> 5) & 1, segments=info["band_segments"],
             bad_pivots=rep["bad_pivots"])


if __name__ == "__main__":
    main()
