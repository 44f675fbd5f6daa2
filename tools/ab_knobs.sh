# Experiment knobs of the A/B build on ONE box: one bench run per setting, same bench arguments.
# usage (GPU box): bash tools/ab_knobs.sh "<bench args>" "X=1" "ZKT_MSM_CBITS=17" "ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18" ...
# ("X=1" = defaults; the knobs are honoured by _ab/libzkt_exp.so only: python -c 'import build; build.build_experiments()')
set -e
B="--steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-latency"
a=$1; shift
for e in "$@"; do
  env ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so $e python bench.py $B $a > gpurun_out/knob.json 2>/dev/null
  echo "$a $e $(python tools/pick.py value roofline.avg_launch_ms int_alu.msm_main_stream_avg_ms int_alu.msm_tail_avg_ms gpu_active.main_stream_idle_ms_per_proof < gpurun_out/knob.json)"
done
