import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for F, S in ((16, 2), (16, 3), (16, 4), (32, 2), (32, 3), (8, 4), (12, 4)):
    run(2160, 3840, F, S, max(6, 320 // F))
