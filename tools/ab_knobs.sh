# regime knobs at the mid sizes on one box (exp library)
set -e
B="--steps 30 --warmup 3 --no-cpu-baseline --no-latency"
sweep() { # bench args, settings...
  a=$1; shift
  for e in "$@"; do
    env ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so $e python bench.py $B $a > gpurun_out/knob.json 2>/dev/null
    echo "$a $e $(python tools/pick.py value roofline.avg_launch_ms int_alu.msm_main_stream_avg_ms int_alu.msm_tail_avg_ms gpu_active.main_stream_idle_ms_per_proof < gpurun_out/knob.json)"
  done
}
sweep "--log-n 17" "ZKT_MSM_TAIL_INL=1" "ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18" "ZKT_MSM_TAIL_INL=1" "ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18"
sweep "--curve bls12_381 --log-n 17" "ZKT_MSM_CBITS=16 ZKT_MSM_TAIL_INL=1" "ZKT_MSM_CBITS=16 ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18" "ZKT_MSM_CBITS=18 ZKT_MSM_TAIL_INL=1" "ZKT_MSM_CBITS=16 ZKT_MSM_TAIL_INL=1" "ZKT_MSM_CBITS=16 ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18"
sweep "--curve bls12_381 --log-n 18" "ZKT_MSM_TAIL_INL=1" "ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18" "ZKT_MSM_TAIL_INL=1" "ZKT_MSM_TAIL_INL=1 ZKT_MSM_BATCH_MAX_LOG=18"
