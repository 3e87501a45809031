"""dev helper (GPU box): the serialised 16-frame 4K pipeline (one slot) -- frames/s and per-kernel event times; run under
different runtime settings to see what makes rocprofv3 runs faster than plain ones"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import torch
from quick_bench import run
run(2160, 3840, 16, 1, 40)
