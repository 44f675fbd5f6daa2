#!/bin/bash
# LDS bank-conflict counters for the NTT pass kernels (separate rocprofv3 --pmc pass, kernel trace only).
# usage (on the GPU box): bash tools/pmc_lds.sh -> gpurun_out/pmc_lds.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_lds -- python3 $R/tools/ntt_prof.py > /dev/null 2>&1 || true
cd $R
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_lds/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkt::", "")[-44:]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in out.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
