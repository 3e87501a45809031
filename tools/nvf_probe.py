import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from quick_bench import run
run(2160, 3840, 16, 1, 30, mask=1)
run(2160, 3840, 16, 3, 30, mask=1)
run(2160, 3840, 16, 1, 30, mask=1, dtype=torch.uint8)
