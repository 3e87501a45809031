"""dev helper (GPU box): frames per call 1..16 on the batched sweeps and (F = 1) the fused kernels: us per frame of embed + detect"""
import subprocess
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, torch; sys.path.insert(0, 'tools'); from quick_bench import run\n"
        "for F, S in ((1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (4, 3), (16, 3)):\n"
        "    run(2160, 3840, F, S, max(20, 120 // F))\n"
        "run(4320, 7680, 1, 1, 20)\nrun(4320, 7680, 4, 2, 10)\n")
print(subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True).stdout)
