"""First line of a committed counter summary: the bench configuration and the digest of the sources it was taken on
(bench.py refuses a summary whose digest is not the current one).  usage: prof_header.py [bench.py arguments]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="bn254")
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--workload", default="withdraw")
a, _ = ap.parse_known_args()
wl = a.workload if (a.workload == "synthetic" or a.log_n in bench.WITHDRAW_SHAPES) else "synthetic"
print("# curve=%s log_n=%d workload=%s csrc_digest=%s" % (a.curve, a.log_n, wl, bench.csrc_digest()))
