import sys
sys.path.insert(0, "visual-odometry_amd"); sys.path.insert(0, ".")
import numpy as np, scenes
from tests.test_gpu_prior import _keyframes, K_MAT
from src.local_ba import SlidingWindowBundleAdjustment
sc = scenes.ring_scene(14, 420, 5)
kfs = _keyframes(sc, sc["poses_guess"])
for ftol in (1e-13, 1e-12, 1e-10):
    sw = SlidingWindowBundleAdjustment(window=8, stride=2, min_views=2, linear_solver=2, ftol=ftol, gtol=1e-9, max_iters=80)
    sw.optimize(kfs, sc["points_guess"], sc["landmark_ids"], K_MAT)
    for b, r in enumerate(sw.last_reports):
        print(ftol, b, r["status_name"], r["iterations"], r["accepted"], r["cost0"], r["cost"], r.get("lam"), r.get("grad_inf"))
        if r["status"] == 0:
            print("  tail costs", r["cost_history"][-6:], "acc", r["accepted_history"][-10:], "lam", r["lambda_history"][-4:])
