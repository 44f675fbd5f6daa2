"""rocprofv3's per-kernel average against the bench line's roofline.avg_launch_ms: k_msm_accumulate is launched for the dense
commitments (what the roofline prices) AND for the Lagrange-basis ones (a few thousand pairs, ~10 us), so rocprof's average over
all launches is lower than the dense average.  Splits the launches from the two averages the profiled run itself reports.
usage: prof_reconcile.py <kernel_stats.csv> <profiled_run.json>   (prints the lines appended to the summary)"""
import csv
import json
import sys

stats, run = sys.argv[1], sys.argv[2]
row = next(r for r in csv.DictReader(open(stats)) if "k_msm_accumulate" in r["Name"])
calls, total_ms = int(row["Calls"]), int(row["TotalDurationNs"]) / 1e6
d = json.loads([l for l in open(run).read().splitlines() if l.startswith("{")][-1])
dense = d["roofline"]["avg_launch_ms"]
sparse = (d.get("commitments") or {}).get("lagrange_accumulate_avg_ms") or 0.0
n_dense = (total_ms - sparse * calls) / (dense - sparse) if dense > sparse else float(calls)
print("k_msm_accumulate: %d launches, %.1f us on average = ~%.0f dense MSMs at %.1f us (roofline.avg_launch_ms of this run, HIP events) "
      "+ ~%.0f Lagrange-basis MSMs at %.1f us (commitments.lagrange_accumulate_avg_ms); min %.1f us, max %.1f us"
      % (calls, 1e3 * total_ms / calls, n_dense, 1e3 * dense, calls - n_dense, 1e3 * sparse, int(row["MinNs"]) / 1e3, int(row["MaxNs"]) / 1e3))
