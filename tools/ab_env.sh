# usage: [BENCH_ARGS="--curve bls12_381 --log-n 22"] bash tools/ab_env.sh <tag> "VAR=val" "VAR2=val2 VAR3=val3" ...
# each argument = one bench run with that environment ("X=1" = default); knobs are honoured by _ab/libzkt_exp.so only
TAG=$1; shift
B="${BENCH_ARGS:-} --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-latency"
i=0
for e in "$@"; do
  i=$((i+1))
  env ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so $e python bench.py $B > gpurun_out/${TAG}_e_$i.json 2> gpurun_out/${TAG}_e_$i.err || { echo "run $i failed"; tail -3 gpurun_out/${TAG}_e_$i.err; exit 1; }
  python - "$e" gpurun_out/${TAG}_e_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], "acc", d["roofline"]["avg_launch_ms"], "main", d["int_alu"]["msm_main_stream_avg_ms"], "tail", d["int_alu"]["msm_tail_avg_ms"], {a: b["avg_ms"] for a, b in k.items()}, d["rounds_ms"])
PY
done
