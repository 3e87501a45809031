#!/bin/bash
# dev helper: only the two SQ passes (see run_pmc.sh)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ARGS="bench.py --steps 3 --warmup 1 --slots 1 --frames-per-slot 16 --no-cpu-baseline $EXTRA"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcq/$name -o $name -- python3 $ARGS > gpurun_out/pmcq_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmcq_$name.log; exit 1; }; }
rm -rf gpurun_out/pmcq; mkdir -p gpurun_out/pmcq
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
