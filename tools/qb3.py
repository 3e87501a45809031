import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
run(2160, 3838, 16, 3, 10)
run(2160, 3838, 16, 1, 10)
run(2160, 3838, 16, 3, 10, dtype=torch.uint8)
