#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats of bench.py, then FETCH_SIZE / WRITE_SIZE passes (own runs).
# usage (on the GPU box): bash tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>_*
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile "$@" > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile "$@" > $R/gpurun_out/prof_${TAG}_write.log 2>&1
cd $R
python3 tools/hbm_traffic.py gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write gpurun_out/prof_${TAG}_hbm_traffic.json
find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${TAG}_kernel_stats.csv \;
# the raw per-dispatch traces are large: keep only the summaries
find gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write -name "*kernel_trace.csv" -delete
find gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write -name "*counter_collection.csv" -delete
tail -2 gpurun_out/prof_${TAG}_stats.log
