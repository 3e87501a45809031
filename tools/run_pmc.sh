#!/bin/bash
# dev helper (GPU box): rocprofv3 counter passes, one pass per counter group
# (PMC passes are kept separate from every trace domain except --kernel-trace, as gpurun requires)
#   PMC_CMD : the python command line to profile (default: the serialised bench)     PMC_OUT : output directory under gpurun_out/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ARGS=${PMC_CMD:-"bench.py --steps 3 --warmup 1 --slots 1 --frames-per-slot 16 --no-cpu-baseline --no-stream --no-single-call --no-membench $EXTRA"}
OUT=${PMC_OUT:-pmc}
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -o $name -- python3 $ARGS > gpurun_out/${OUT}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/${OUT}_$name.log; exit 1; }; }
mkdir -p gpurun_out/$OUT
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
if [ -z "$PMC_SHORT" ]; then
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE
fi
ls gpurun_out/$OUT
