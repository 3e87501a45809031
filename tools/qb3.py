import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for F, S in ((32, 3), (64, 3), (64, 2), (128, 2)):
    run(1080, 1920, F, S, max(4, 640 // F))
for rps in (30, 36, 54, 60):
    run(1080, 1920, 64, 3, 10, rps=rps)
