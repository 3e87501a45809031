import sys, torch
sys.path.insert(0, "tests")
from quick_bench import run
for F, S in ((1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (2, 2), (2, 3), (2, 4), (3, 2), (4, 2), (4, 3), (8, 2), (8, 3), (16, 1), (16, 2)):
    run(2160, 3840, F, S, max(10, 200 // (F * S)))
