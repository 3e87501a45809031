#!/bin/bash
# dev helper (GPU box): extra counter passes for one question -- what do k_detect's waves wait for? (instruction fetch, issue, memory)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ARGS="bench.py --steps 3 --warmup 1 --slots 1 --frames-per-slot 16 --no-cpu-baseline --no-stream --no-single-call --sustain-seconds 0"
OUT=pmc_extra
mkdir -p gpurun_out/$OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -o $name -- python3 $ARGS > gpurun_out/${OUT}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/${OUT}_$name.log; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
run sq3 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run sq4 SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES_EQ_64
python3 tools/pmc_summary.py gpurun_out/$OUT gpurun_out/$OUT/summary.json > /dev/null
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_extra/summary.json"))
for k in ("k_detect", "k_me_stats", "k_embed", "k_gram"):
    if k in d:
        print(k, json.dumps({a: round(b, 1) for a, b in d[k].items()}))
PY
