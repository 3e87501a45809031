import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
for p in (3, 5, 7, 9):
    run(2160, 3840, 8, 3, 10, mask=1, p=p)
