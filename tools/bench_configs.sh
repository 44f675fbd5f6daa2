#!/bin/bash
# the other BASELINE.json configs on one GPU -> gpurun_out/bench_<tag>_*.json   usage: bash tools/bench_configs.sh <tag>
TAG=${1:-r03}
python bench.py --curve bls12_381 --log-n 22 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_bls12_381_2_22.json 2> gpurun_out/bls22.err || tail -c 300 gpurun_out/bls22.err
python bench.py --curve bls12_381 --log-n 20 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_bls12_381_2_20.json 2> gpurun_out/bls20.err || tail -c 300 gpurun_out/bls20.err
python bench.py --log-n 14 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_bn254_2_14.json 2> gpurun_out/bn14.err || tail -c 300 gpurun_out/bn14.err
for f in bls12_381_2_22 bls12_381_2_20 bn254_2_14; do
python - gpurun_out/bench_${TAG}_$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("latency"), d.get("verify_ms"), d.get("verify",{}).get("batch16_ms_per_proof"))
PY
done
