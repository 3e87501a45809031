import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for F, S in ((1, 3), (2, 3), (2, 4), (4, 3), (4, 4), (8, 3), (16, 3), (32, 3)):
    run(2160, 3840, F, S, max(6, 300 // (F * S)))
