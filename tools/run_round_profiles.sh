#!/bin/bash
# dev helper (GPU box): everything profiles/ is regenerated from, in one gpurun call:
#   bench lines (f32 with all legs, u8), rocprofv3 --kernel-trace --stats of the serialised bench and of the one-image-per-call
#   loop (fused kernels), then the PMC passes of both (run_pmc.sh)
# usage: gpurun --timeout 1200 -- 'bash tools/run_round_profiles.sh r03'   then, here: python tools/make_profiles.py r03
RND=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_${RND}.json 2> gpurun_out/bench_${RND}.err || { tail -5 gpurun_out/bench_${RND}.err; exit 1; }
python3 bench.py --dtype u8 --no-cpu-baseline --no-stream > gpurun_out/bench_${RND}_u8.json 2> gpurun_out/bench_${RND}_u8.err || { tail -5 gpurun_out/bench_${RND}_u8.err; exit 1; }
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_single
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o ${RND} -- python3 bench.py --steps 5 --warmup 2 --slots 1 --frames-per-slot 16 --no-cpu-baseline --no-stream --no-single-call --no-membench > gpurun_out/prof_stats.log 2>&1 || { tail -5 gpurun_out/prof_stats.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_single -o ${RND}_single -- python3 tools/f1_trace.py 200 > gpurun_out/prof_stats_single.log 2>&1 || { tail -5 gpurun_out/prof_stats_single.log; exit 1; }
rm -rf gpurun_out/pmc gpurun_out/pmc_single
bash tools/run_pmc.sh > gpurun_out/run_pmc.log 2>&1 || { tail -5 gpurun_out/run_pmc.log; exit 1; }
PMC_SHORT=1 PMC_OUT=pmc_single PMC_CMD="tools/f1_trace.py 40" bash tools/run_pmc.sh > gpurun_out/run_pmc_single.log 2>&1 || { tail -5 gpurun_out/run_pmc_single.log; exit 1; }
# only 64 MiB of gpurun_out/ travel back: keep the rows of this engine's kernels, drop everything else the profiler wrote
find gpurun_out -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.otf2" -o -name "*_results.json" \) -delete
for f in $(find gpurun_out -type f -name "*.csv" -size +256k); do
    { head -1 "$f"; grep "wmk::" "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
du -sh gpurun_out | cut -f1
cat gpurun_out/bench_${RND}.json
