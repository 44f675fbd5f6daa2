"""Per-proof GPU timeline from a rocprofv3 kernel trace: busy time per stream and the idle gaps.
usage: timeline.py <dir with *_kernel_trace.csv> [proof_index_from_end]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a proof starts with the first k_ntt/k_msm after the witness copy; use k_quotient as the anchor
anchors = [i for i, r in enumerate(rows) if "k_quotient" in r["Kernel_Name"]]
if len(anchors) < 3:
    sys.exit("need >= 3 proofs in the trace")
# which pair of consecutive proofs: k-th from the end (bench.py: warm-up, the timed proofs, then a profiling pass with every
# event scope on -- the timed region is what the timeline should show, so prof_stats.sh passes the index of a timed proof)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(anchors) < k + 1:
    sys.exit("not enough proofs in the trace")
a0, a1 = anchors[-k - 1], anchors[-k]
t0, t1 = int(rows[a0]["Start_Timestamp"]), int(rows[a1]["Start_Timestamp"])
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
print("window (quotient to quotient): %.3f ms, %d dispatches" % ((t1 - t0) / 1e6, len(sel)))
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ev in byq.items():
    busy = sum(e - s for s, e, _ in ev)
    print("queue %s: %d kernels, busy %.3f ms" % (q, len(ev), busy / 1e6))
# union of all intervals -> idle time
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
cur_s, cur_e = iv[0]
idle = 0
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        idle += s - cur_e
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
print("GPU idle inside the window: %.3f ms in %d gaps" % (idle / 1e6, len(gaps)))
gaps.sort(reverse=True)
for g, at in gaps[:12]:
    prev = max((r for r in sel if int(r["End_Timestamp"]) <= at + 1), key=lambda r: int(r["End_Timestamp"]))
    print("  gap %.1f us after %s" % (g / 1e3, prev["Kernel_Name"].split("(")[0][-50:]))
tot = collections.Counter()
for r in sel:
    tot[r["Kernel_Name"].split("(")[0].replace("void zkt::", "")[:60]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in tot.most_common(25):
    print("%9.3f ms  %s" % (v / 1e6, k))
