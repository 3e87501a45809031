#!/bin/bash
# dev helper (GPU box): the NVF legs of the round's profiles -- BASELINE.json configs[1] / [4] name the NVF mask.
# Per size (1080p F=32, 4K F=16, 8K F=4; one slot, serialised): the bench line with --mask NVF, rocprofv3 --kernel-trace
# --stats, and the PMC passes (each in a run of its own, only --kernel-trace beside --pmc).  The program follows `--` directly.
# usage: gpurun --timeout 1200 -- 'bash tools/run_nvf_profiles.sh r04'   then, here: python tools/make_nvf_profiles.py r04
RND=${1:-r04}
SIZES=${NVF_SIZES:-"1080x1920x32 2160x3840x16 4320x7680x4"}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/nvf
for sz in $SIZES; do
    rows=${sz%%x*}; rest=${sz#*x}; cols=${rest%%x*}; F=${rest#*x}
    tag=${rows}x${cols}
    COMMON="--mask NVF --rows $rows --cols $cols --frames-per-slot $F --no-cpu-baseline --no-stream --no-single-call"
    python3 bench.py $COMMON --slots 3 > gpurun_out/nvf/bench_${tag}.json 2> gpurun_out/nvf/bench_${tag}.err || { tail -5 gpurun_out/nvf/bench_${tag}.err; exit 1; }
    SER="$COMMON --steps 5 --warmup 2 --slots 1 --no-slot-out --no-membench --sustain-seconds 0"
    rm -rf gpurun_out/nvf/stats_${tag}
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nvf/stats_${tag} -o ${RND} -- python3 bench.py $SER > gpurun_out/nvf/stats_${tag}.log 2>&1 || { tail -5 gpurun_out/nvf/stats_${tag}.log; exit 1; }
    run() { name=$1; shift; rm -rf gpurun_out/nvf/pmc_${tag}/$name; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/nvf/pmc_${tag}/$name -o $name -- python3 bench.py $SER > gpurun_out/nvf/pmc_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/nvf/pmc_${tag}_$name.log; exit 1; }; }
    run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
    run fetch FETCH_SIZE
    run write WRITE_SIZE
    run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS
    echo "done $tag"
done
find gpurun_out/nvf -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.otf2" -o -name "*_results.json" \) -delete
for f in $(find gpurun_out/nvf -type f -name "*.csv" -size +256k); do
    { head -1 "$f"; grep "wmk::" "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
du -sh gpurun_out/nvf | cut -f1
