import sys, torch
sys.path.insert(0, "tests")
from quick_bench import run
run(2160, 3840, 8, 2, 40)
run(2160, 3840, 16, 1, 20)
run(2160, 3840, 1, 1, 100)
run(2160, 3840, 8, 2, 40, dtype=torch.uint8)
run(2160, 3840, 8, 2, 40, mask=1)
