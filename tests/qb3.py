import sys, torch
sys.path.insert(0, "tests")
from quick_bench import run
for F, S in ((16, 2), (16, 3), (16, 4), (24, 2), (32, 2), (12, 4), (8, 4), (8, 6)):
    run(2160, 3840, F, S, max(6, 300 // (F * S)))
