# dev helper (GPU box): the bench loop with and without the Gram hand-over (tools/quick_bench.py, WM_QB_HANDOVER), optionally on
# another build of the library (WM_AB_LIB).  usage: bash tools/ho_ab.sh ["F,slots F,slots ..."]
CASES=${1:-"16,1 16,3"}
for ho in "" 1; do
echo "== hand-over: ${ho:-0}"
WM_QB_HANDOVER=$ho WM_CASES="$CASES" python -c "
import os, sys, torch; sys.path.insert(0,'tools'); from quick_bench import run
for c in os.environ['WM_CASES'].split():
    F, S = map(int, c.split(','))
    run(2160, 3840, F, S, max(20, 200 // F))
" 2>&1 | grep -v amdgpu.ids
done
