# NTT radix splits on one box (exp library, ZKT_NTT_SPLIT); X=1 = the built-in split
set -e
B="--steps 20 --warmup 3 --no-cpu-baseline --no-latency"
sweep() { # bench args, settings...
  a=$1; shift
  for e in "$@"; do
    env ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so "$e" python bench.py $B $a > gpurun_out/knob.json 2>/dev/null
    echo "$a $e $(python tools/pick.py value verify.accepted kernels < gpurun_out/knob.json | sed 's/launches.: [0-9]*, //g; s/.GB.s.: [0-9.]*, .frac_hbm.: [0-9.]*, .frac_of_mad_issue_ceiling_upper.: [0-9.]*//g')"
  done
}
sweep "--log-n 20" "X=1" "ZKT_NTT_SPLIT=22:7,8,7" "ZKT_NTT_SPLIT=22:7,7,8" "ZKT_NTT_SPLIT=20:7,6,7" "ZKT_NTT_SPLIT=20:6,7,7" "ZKT_NTT_SPLIT=20:8,6,6" "ZKT_NTT_SPLIT=20:6,6,8" "X=1"
sweep "--log-n 18" "X=1" "ZKT_NTT_SPLIT=18:9,9" "ZKT_NTT_SPLIT=18:7,6,5" "ZKT_NTT_SPLIT=18:5,6,7" "ZKT_NTT_SPLIT=20:6,7,7" "ZKT_NTT_SPLIT=20:8,6,6" "X=1"
