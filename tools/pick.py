"""Prints selected (dotted) keys of the last JSON line on stdin.  usage: python bench.py ... | python tools/pick.py value roofline.avg_launch_ms"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
out = []
for key in sys.argv[1:]:
    v = d
    for part in key.split("."):
        v = v.get(part) if isinstance(v, dict) else None
    out.append("%s=%s" % (key, v))
print(" ".join(out))
