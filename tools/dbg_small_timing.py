import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odometry_amd")); sys.path.insert(0, ROOT)
import numpy as np, scenes
from src import _native
sc = scenes.ring_scene(50, 20000, 10)
s = _native.Solver(max_iters=1)
s.set_problem(50, 20000, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
s.upload_guess(sc["poses_guess"], sc["points_guess"])
rep = s.solve_resident()
print(rep["iterations"], rep["pcg_iters"])
