# A/B of one change on one box: _ab/libzkt_exp.so (built from the previous commit, "old") against the main library (new), alternating
set -e
B="--steps 30 --warmup 3 --no-cpu-baseline --no-latency"
for a in "--log-n 14" "--log-n 18" "--log-n 20" "--curve bls12_381 --log-n 18"; do
  for i in 1 2; do
    ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so python bench.py $B $a > gpurun_out/ow_old.json 2>/dev/null
    echo "$a old $(python tools/pick.py value int_alu.msm_main_stream_avg_ms < gpurun_out/ow_old.json)"
    python bench.py $B $a > gpurun_out/ow_new.json 2>/dev/null
    echo "$a new $(python tools/pick.py value int_alu.msm_main_stream_avg_ms < gpurun_out/ow_new.json)"
  done
done
