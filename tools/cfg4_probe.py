#!/usr/bin/env python3
"""Development helper: config 4 at the reference's tolerances; prints where the cost history is not monotone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import numpy as np
import scenes
from src import _native
sc = scenes.ring_scene(10000, 2000000, 5)
with _native.Solver(ftol=1e-6, xtol=1e-6, gtol=1e-6) as s:
    s.set_problem(10000, 2000000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
    P, X, rep = s.solve(sc["poses_guess"], sc["points_guess"])
h = rep["cost_history"]
print({k: v for k, v in rep.items() if not k.endswith("history")})
for i, (a, b) in enumerate(zip(h, h[1:])):
    if b > a:
        print("not monotone at", i, a, b, rep["accepted_history"][i:i + 2], rep["lambda_history"][i:i + 2])
print(h[:5], h[-5:])
