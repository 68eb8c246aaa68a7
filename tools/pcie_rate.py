#!/usr/bin/env python3
"""Development helper: PCIe-inclusive solve (ba_solve: upload + solve + download) vs the resident solve, config 2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import scenes
from src import _native
sc = scenes.ring_scene(50, 20000, 10)
s = _native.Solver()
s.set_problem(50, 20000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
for _ in range(3):
    s.solve(sc["poses_guess"], sc["points_guess"])
t0 = time.perf_counter()
for _ in range(20):
    P, X, rep = s.solve(sc["poses_guess"], sc["points_guess"])
t1 = time.perf_counter()
s.upload_guess(sc["poses_guess"], sc["points_guess"])
for _ in range(20):
    rep2 = s.solve_resident()
t2 = time.perf_counter()
n = len(sc["obs_cam"])
print("ba_solve (host buffers in/out): %.3f ms per solve, %.3e obs*it/s; resident: %.3f ms, %.3e obs*it/s" % (
    (t1 - t0) / 20 * 1e3, n * rep["iterations"] * 20 / (t1 - t0), (t2 - t1) / 20 * 1e3, n * rep2["iterations"] * 20 / (t2 - t1)))
