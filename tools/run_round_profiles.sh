#!/bin/bash
# dev helper (GPU box): everything profiles/ is regenerated from, in one gpurun call:
#   bench lines (f32, u8), rocprofv3 --kernel-trace --stats of the serialized bench, then the PMC passes (run_pmc.sh)
# usage: gpurun --timeout 1200 -- 'bash tools/run_round_profiles.sh r01'
RND=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_${RND}.json 2> gpurun_out/bench_${RND}.err || { tail -5 gpurun_out/bench_${RND}.err; exit 1; }
python3 bench.py --dtype u8 --no-cpu-baseline > gpurun_out/bench_${RND}_u8.json 2> gpurun_out/bench_${RND}_u8.err || { tail -5 gpurun_out/bench_${RND}_u8.err; exit 1; }
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o ${RND} -- python3 bench.py --steps 5 --warmup 2 --slots 1 --frames-per-slot 16 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1 || { tail -5 gpurun_out/prof_stats.log; exit 1; }
rm -rf gpurun_out/pmc
bash tools/run_pmc.sh > gpurun_out/run_pmc.log 2>&1 || { tail -5 gpurun_out/run_pmc.log; exit 1; }
cat gpurun_out/bench_${RND}.json
