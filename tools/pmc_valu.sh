#!/bin/bash
# VALU-busy / stall-reason counters of the hot kernels (one rocprofv3 --pmc pass per counter group, kernel trace only;
# MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots per pass, SQ_* count quad-cycles).
# usage (on the GPU box): bash tools/pmc_valu.sh [tag]  -> gpurun_out/pmc_valu_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
export BENCH_ARGS
ARGS="$BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
rocprofv3 -L > $R/gpurun_out/counters_avail.txt 2>&1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
G3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU"
rm -rf $R/gpurun_out/pmc_valu_1 $R/gpurun_out/pmc_valu_2 $R/gpurun_out/pmc_valu_3
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $R/gpurun_out/pmc_valu_$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_valu_$i.log 2>&1 || echo "pass $i failed (see gpurun_out/pmc_valu_$i.log)"
done
cd $R
python3 - $TAG <<'PY'
import csv, glob, collections, sys, os
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_valu_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkt::", "")[-44:]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in out.items():   # the accumulation: dense commitments only (the Lagrange-basis ones run it over a few thousand pairs)
    if "k_msm_accumulate" in k:
        for cn in list(v):
            mx = max(v[cn])
            v[cn] = [x for x in v[cn] if x >= 0.25 * mx] if mx > 0 else v[cn]
names = sorted({c for v in out.values() for c in v})
import subprocess
with open("gpurun_out/pmc_valu_%s.txt" % sys.argv[1], "w") as fo:
    fo.write(subprocess.run([sys.executable, "tools/prof_header.py"] + os.environ.get("BENCH_ARGS", "").split(), capture_output=True, text=True).stdout)
    fo.write("per-launch averages, raw counter values (SQ_* cycle counters are quad-cycles summed over all SQs)\n")
    for k, v in sorted(out.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        if sum(v.get("SQ_WAVE_CYCLES", [0])) < 1e6 and "accumulate" not in k and "ntt_pass" not in k: continue
        fo.write("%s  (launches %d)\n" % (k, max(len(x) for x in v.values())))
        for c in names:
            if c in v: fo.write("    %-26s %16.0f\n" % (c, sum(v[c]) / len(v[c])))
        g = lambda c: (sum(v[c]) / len(v[c])) if c in v else None
        wc, act, valu, wany, winst = g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY"), g("SQ_ACTIVE_INST_VALU"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY")
        if wc:
            fo.write("    -> of wave lifetime: issuing %.3f (VALU %.3f), parked on waitcnt/barrier %.3f, issue-stalled %.3f\n" % (
                (act or 0) / wc, (valu or 0) / wc, (wany or 0) / wc, (winst or 0) / wc))
        busy = g("SQ_BUSY_CYCLES")
        if busy and valu:
            fo.write("    -> SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.3f\n" % (valu / busy))
PY
