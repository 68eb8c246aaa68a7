import sys, time, cProfile, pstats
sys.path.insert(0, "visual-odometry_amd")
import scenes
from src.bundle_adjustment import BundleAdjustment
sc = scenes.ring_scene(50, 20000, 10)
args = scenes.to_containers(sc)
ba = BundleAdjustment()
ba.optimize(*args)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): ba.optimize(*args)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
