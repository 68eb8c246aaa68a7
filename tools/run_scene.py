#!/usr/bin/env python3
"""Run one synthetic scene through the GPU solver and print the report (development helper)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import numpy as np
import scenes
from src import _native

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=1000); ap.add_argument("--L", type=int, default=200000)
ap.add_argument("--d", type=int, default=5); ap.add_argument("--verbose", type=int, default=1)
ap.add_argument("--reps", type=int, default=1); ap.add_argument("--visibility", default="ring")
ap.add_argument("--opt", nargs="*", default=[])
a = ap.parse_args()
t0 = time.time(); sc = scenes.ring_scene(a.C, a.L, a.d, visibility=a.visibility); print("scene built in %.1fs" % (time.time() - t0), flush=True)
opts = {}
for kv in a.opt:
    k, v = kv.split("="); opts[k] = float(v) if "." in v or "e" in v else int(v)
s = _native.Solver(verbose=a.verbose, profile_kernels=1, **opts)
t0 = time.time(); s.set_problem(a.C, a.L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"]); print("set_problem %.2fs, S upper blocks %d" % (time.time() - t0, s.schur_block_count()), flush=True)
s.upload_guess(sc["poses_guess"], sc["points_guess"])
prev = s.kernel_times()
for _ in range(a.reps):
    rep = s.solve_resident()
    cur = s.kernel_times()
    print("rep kernels (avg us, launches):", {k: (round((cur[k][0] - prev[k][0]) / max(cur[k][1] - prev[k][1], 1) * 1e3, 1), cur[k][1] - prev[k][1]) for k in cur}, flush=True)
    prev = cur
    print(json.dumps({k: v for k, v in rep.items() if not k.endswith("history")}), flush=True)
    print("pcg history", rep["pcg_history"])
print({k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in s.kernel_times().items()})
