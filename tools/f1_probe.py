import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from quick_bench import run
for F in (1, 2, 4):
    run(2160, 3840, F, 1, 60)
run(4320, 7680, 1, 1, 20)
run(1080, 1920, 1, 1, 60)
