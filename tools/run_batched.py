#!/usr/bin/env python3
"""Development helper: time a batch of local-BA windows (BASELINE config 5) as one block-diagonal problem."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import numpy as np, scenes
from src import _native
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--K", type=int, default=10); ap.add_argument("--L", type=int, default=200)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--verbose", type=int, default=0)
a = ap.parse_args()
t0 = time.time()
cam, pt, uv, poses, pts, fixed = [], [], [], [], [], []
co = lo = 0
for b in range(a.B):
    sc = scenes.local_window_scene(b, K=a.K, L=a.L)
    cam.append(sc["obs_cam"] + co); pt.append(sc["obs_pt"] + lo); uv.append(sc["obs_uv"])
    poses.append(sc["poses_guess"]); pts.append(sc["points_guess"])
    f = np.zeros(a.K, np.uint8); f[0] = 1; fixed.append(f)
    co += a.K; lo += a.L
cam = np.concatenate(cam); pt = np.concatenate(pt); uv = np.concatenate(uv); poses = np.concatenate(poses); pts = np.concatenate(pts)
fixed = np.concatenate(fixed)
print("batch built in %.1fs: %d cameras, %d landmarks, %d observations" % (time.time() - t0, co, lo, len(cam)), flush=True)
s = _native.Solver(verbose=a.verbose, profile_kernels=1)
t0 = time.time(); s.set_problem(co, lo, cam, pt, uv, sc["intr"], fixed); print("set_problem %.2fs" % (time.time() - t0), flush=True)
s.upload_guess(poses, pts)
for _ in range(a.reps):
    rep = s.solve_resident()
    print({k: rep[k] for k in ("cost0", "cost", "iterations", "accepted", "pcg_iters", "status_name", "solve_ms")}, flush=True)
print("per window: %.2f us; obs*iter/s: %.3e" % (1e3 * rep["solve_ms"] / a.B, len(cam) * rep["iterations"] / (rep["solve_ms"] * 1e-3)))
print({k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in s.kernel_times().items()})
