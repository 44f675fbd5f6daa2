#!/bin/bash
# rocprofv3 kernel stats of a short bench run -> gpurun_out/<tag>_summary.txt (+ kernel_stats.csv, timeline)
# usage (GPU box): [BENCH_ARGS="--curve bls12_381 --log-n 22"] bash tools/prof_stats.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py $BENCH_ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/${TAG}_profiled_run.json 2> $R/gpurun_out/${TAG}_profiled_run.err
cd $R
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py $BENCH_ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-latency  ($TAG; includes setup kernels)"
  python3 tools/prof_reconcile.py gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_profiled_run.json
  python3 tools/prof_summ.py gpurun_out/prof_$TAG; } > gpurun_out/${TAG}_summary.txt
# 11 proofs in the trace: 1 warm-up, 5 timed, 5 in the profiling pass -> the 7th from the end is a timed one
python3 tools/timeline.py gpurun_out/prof_$TAG 7 > gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf gpurun_out/prof_$TAG
