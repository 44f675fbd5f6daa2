# A/B of one source change on ONE box: _ab/libzkt_exp.so (built from the previous commit: "old") against the main library ("new"),
# alternating.  usage (GPU box): bash tools/ab_ow.sh ["<bench args>" ...]
set -e
B="--steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-latency"
[ $# -eq 0 ] && set -- "--log-n 20"
for a in "$@"; do
  for i in 1 2 3; do
    ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so python bench.py $B $a > gpurun_out/ow_old.json 2>/dev/null
    echo "$a old $(python tools/pick.py value int_alu.msm_main_stream_avg_ms rounds_ms < gpurun_out/ow_old.json)"
    python bench.py $B $a > gpurun_out/ow_new.json 2>/dev/null
    echo "$a new $(python tools/pick.py value int_alu.msm_main_stream_avg_ms rounds_ms < gpurun_out/ow_new.json)"
  done
done
