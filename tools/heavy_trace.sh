#!/bin/bash
# which MSMs of a proof meet crowded buckets: durations of k_msm_heavy in launch order.  usage: BENCH_ARGS=... bash tools/heavy_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_heavy
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_heavy -- python3 $R/bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_heavy/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
k = 0
for r in rows:
    n = r["Kernel_Name"]
    if "k_msm_heavy" in n or "k_msm_bucket_sum" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "heavy" in n:
            print("msm %3d  heavy %9.1f us   bucket_sum %9.1f us" % (k, d, last))
            k += 1
        else:
            last = d
PY
rm -rf gpurun_out/prof_heavy
