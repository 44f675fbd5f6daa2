#!/bin/bash
# the other BASELINE.json configs and the reference's real circuit sizes on one GPU -> gpurun_out/bench_<tag>_*.json
# usage: bash tools/bench_configs.sh <tag>
#   bn254 2^14 : configs[0]  WithdrawCircuit x4, 1 note, HEIGHT 7
#   bn254 2^18 : the CLI's default features (bin/Cargo.toml:25: height-48, notes-3, x4; 200 793 gates)
#   bn254 2^19 : its largest shipped feature set (height-64, notes-4, x5; 511 702 gates)
#   bls12_381 2^20, 2^22 : configs[1]'s field / configs[4] on one GPU (withdraw-shaped with generated Poseidon constants)
TAG=${1:-r05}
run() { name=$1; shift; python bench.py "$@" > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err || { tail -c 400 gpurun_out/bench_${TAG}_$name.err; return 1; }
python - gpurun_out/bench_${TAG}_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["rounds_ms"], (d.get("witness") or {}).get("device_ms"), d.get("latency"), d.get("verify_ms"))
PY
}
run bn254_2_14 --log-n 14 --steps 60 --warmup 3 --inflight 2 &&
run bn254_2_18 --log-n 18 --steps 40 --warmup 3 --inflight 3 &&
run bn254_2_19 --log-n 19 --steps 20 --warmup 3 &&
run bls12_381_2_20 --curve bls12_381 --log-n 20 --steps 5 --warmup 1 --no-cpu-baseline &&
run bls12_381_2_22 --curve bls12_381 --log-n 22 --steps 3 --warmup 1 --no-cpu-baseline
