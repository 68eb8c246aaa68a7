#!/usr/bin/env python3
"""BASELINE configs[0] (the reference's demo scene: 8 poses x 100 landmarks) through the drop-in entry point:
wall time of BundleAdjustment.optimize and of the resident solve."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import scenes
from src import _native

sc = scenes.demo_scene() if hasattr(scenes, "demo_scene") else None
if sc is None:
    raise SystemExit("scenes.demo_scene missing")
C, L = len(sc["poses_gt"]), len(sc["points_gt"])
with _native.Solver() as s:
    s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    s.upload_guess(sc["poses_guess"], sc["points_guess"])
    s.solve_resident()
    t0 = time.perf_counter()
    for _ in range(20):
        rep = s.solve_resident()
    dt = (time.perf_counter() - t0) / 20
    print(f"demo scene {C} x {L} x {len(sc['obs_cam'])}: {dt * 1e3:.3f} ms per solve, {rep['iterations']} LM iterations, "
          f"cost {rep['cost0']:.6g} -> {rep['cost']:.3g}, solver {s.reduced_solver()}")
