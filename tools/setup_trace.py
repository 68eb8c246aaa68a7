import sys, os, time
sys.path.insert(0, "visual-odometry_amd")
import scenes
from src import _native
for C, L, d in ((50, 20000, 10), (1000, 200000, 5), (10000, 2000000, 5)):
    sc = scenes.ring_scene(C, L, d, seed=0)
    with _native.Solver() as s:
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        print("---- warm", C, L, file=sys.stderr)
        t = time.perf_counter()
        s.set_problem(C, L, sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["intr"])
        print(f"---- total {1e3*(time.perf_counter()-t):.2f} ms", file=sys.stderr)
