# dev helper (GPU box): the bench loop with and without the Gram hand-over (tools/quick_bench.py, WM_QB_HANDOVER), optionally on
# another build of the library (WM_AB_LIB)
for ho in "" 1; do
WM_QB_HANDOVER=$ho python -c "
import sys, torch; sys.path.insert(0,'tools'); from quick_bench import run
run(2160,3840,16,1,20); run(2160,3840,16,3,20)
"
done
