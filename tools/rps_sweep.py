"""dev helper (GPU box): does co-residency of different sweeps (fewer, longer segments per launch; several slots) raise
the batched rate?  4K f32 ME, 16 frames per launch."""
import sys
sys.path.insert(0, "tools")
from quick_bench import run
for ns in (3, 4, 6):
    for rps in (0, 96, 180, 270, 540, 1080):
        run(2160, 3840, 16, ns, 12, rps=rps)
