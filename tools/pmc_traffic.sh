#!/bin/bash
# HBM traffic of the hot kernels from the TCC counters (MI355X_MICROARCH.md "HBM": separate passes for
# FETCH_SIZE and WRITE_SIZE; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x).
# usage (on the GPU box): [BENCH_ARGS=...] bash tools/pmc_traffic.sh  -> stdout (redirect into gpurun_out/pmc_traffic_<tag>.txt)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$ctr
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$ctr -- python3 $R/bench.py $BENCH_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-latency > /dev/null 2>&1
done
cd $R
python3 tools/prof_header.py $BENCH_ARGS
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % ctr):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkt::", "")[-44:]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-46s %8s %14s %14s   (KiB per launch, raw counter values)" % ("kernel", "launches", "FETCH_SIZE", "WRITE_SIZE"))
for k, v in sorted(out.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    f = v.get("FETCH_SIZE", [0]); w = v.get("WRITE_SIZE", [0])
    if "k_msm_accumulate" in k:   # the dense commitments only: the Lagrange-basis ones run the kernel over a few thousand pairs
        f = [x for x in f if x >= 0.25 * max(f)]; w = [x for x in w if x >= 0.25 * max(w)]
    if sum(f) + sum(w) < 1e5: continue
    print("%-46s %8d %14.0f %14.0f" % (k, len(f), sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)))
PY
