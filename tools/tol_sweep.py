#!/usr/bin/env python3
"""Development helper: LM / PCG iteration counts and final cost for a few option sets on the fixture scenes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import numpy as np, scenes
from src import _native
cases = [("demo", scenes.demo_scene()), ("noisy_10x300", scenes.noisy_scene(10, 300)), ("ring_50x2000", scenes.ring_scene(50, 2000, 10)),
         ("cfg2", scenes.ring_scene(50, 20000, 10)), ("window", scenes.local_window_scene(3))]
opts = [dict(), dict(ftol=1e-8), dict(ftol=1e-8, pcg_tol=0.03), dict(ftol=1e-8, pcg_tol=0.1)]
for name, sc in cases:
    C = len(sc["poses_guess"]); L = len(sc["points_guess"])
    for o in opts:
        s = _native.Solver(**o)
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        s.upload_guess(sc["poses_guess"], sc["points_guess"])
        s.solve_resident(); rep = s.solve_resident()
        print(f"{name:14s} {str(o):36s} it {rep['iterations']:3d} acc {rep['accepted']:3d} pcg {rep['pcg_iters']:5d} cost {rep['cost']:.10e} {rep['status_name']:9s} {rep['solve_ms']:.3f} ms", flush=True)
        s.close()
