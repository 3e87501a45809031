import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for rps in (6, 8, 12, 16, 24):
    run(2160, 3840, 1, 1, 100, rps=rps)
for rps in (8, 16, 24, 32):
    run(2160, 3840, 2, 1, 60, rps=rps)
for rps in (16, 24, 32, 45):
    run(2160, 3840, 4, 1, 40, rps=rps)
