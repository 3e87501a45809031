#!/bin/bash
# dev helper (GPU box): rocprofv3 counter passes over bench.py, one pass per counter group
# (PMC passes are kept separate from every trace domain except --kernel-trace, as gpurun requires)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ARGS="bench.py --steps 3 --warmup 1 --slots 1 --frames-per-slot 16 --no-cpu-baseline $EXTRA"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python3 $ARGS > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$name.log; exit 1; }; }
mkdir -p gpurun_out/pmc
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE
ls -R gpurun_out/pmc | head -40
