import sys, torch
sys.path.insert(0, "tools")
from quick_bench import run
run(1080, 1920, 32, 3, 20)
run(1080, 1920, 32, 3, 20, mask=1)
run(2160, 3840, 16, 3, 20, mask=1)
run(4320, 7680, 4, 3, 10)
run(4320, 7680, 4, 3, 10, mask=1)
run(2160, 3840, 16, 3, 20, mask=1, dtype=torch.uint8)
