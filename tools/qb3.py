import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for rps in (8, 12, 16, 26, 0):
    run(2160, 3840, 1, 1, 100, rps=rps, mask=1)
for rps in (8, 12, 16, 26, 0):
    run(2160, 3840, 1, 1, 100, rps=rps, mask=0)
