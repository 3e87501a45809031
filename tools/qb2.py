import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for F in (1, 2, 4, 8, 16):
    run(2160, 3840, F, 1, max(10, 160 // F))
