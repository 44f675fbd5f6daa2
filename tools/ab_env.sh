# usage: bash tools/ab_env.sh "VAR=val" "VAR2=val2" ...   (each argument = one bench run with that environment; "X=1" = default)
B="--steps 20 --warmup 3 --no-cpu-baseline"
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py $B > gpurun_out/r03_e_$i.json 2> gpurun_out/r03_e_$i.err || { echo "run $i failed"; tail -3 gpurun_out/r03_e_$i.err; exit 1; }
  python - "$e" gpurun_out/r03_e_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
l=d["latency"]; k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["int_alu"]["msm_main_stream_avg_ms"], d["int_alu"]["msm_tail_avg_ms"], k["ntt_2^20"]["avg_ms"], k["ntt_2^22"]["avg_ms"], k["quotient"]["avg_ms"], l["cold_single_proof_ms"], l["unchained_single_proof_ms"])
PY
done
