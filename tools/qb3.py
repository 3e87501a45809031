import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for rps in (24, 32, 40, 48, 64):
    run(2160, 3840, 16, 3, 20, rps=rps)
run(2160, 3840, 24, 3, 14)
run(2160, 3840, 16, 4, 20)
run(2160, 3840, 32, 2, 10)
