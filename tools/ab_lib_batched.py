#!/usr/bin/env python3
"""Batched solve (1024 windows) with the library named in BA_AMD_LIB (a build of an older commit for A / B runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd"))
import numpy as np
import scenes
from src import _native
for k in ("ba_comm_info", "ba_debug_table_digests"):
    if os.environ.get("BA_OLD_ABI"):
        _native._SIGNATURES.pop(k, None)
B, L = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 200
probs = []
for b in range(B):
    sc = scenes.local_window_scene(b, K=10, L=L)
    probs.append(dict(obs_cam=sc["obs_cam"], obs_pt=sc["obs_pt"], obs_uv=sc["obs_uv"], poses=sc["poses_guess"], points=sc["points_guess"]))
intr = sc["intr"]
with _native.Solver() as s:
    for rep in range(4):
        t0 = time.perf_counter()
        P, X, c0, c, r = s.solve_batched(probs, intr)
        print(f"L={L}: solve_ms {r['solve_ms']:.3f} setup_ms {r['setup_ms']:.1f} iterations {r['iterations']} cost {r['cost']:.6f} wall {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
