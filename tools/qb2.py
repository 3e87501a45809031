import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
run(2160, 3840, 16, 1, 20)
run(2160, 3840, 16, 3, 20)
run(2160, 3840, 32, 3, 10)
run(2160, 3840, 8, 3, 30)
run(2160, 3840, 16, 3, 20, dtype=torch.uint8)
