# A/B of one change on one box: main library (old) against _ab/libzkt_exp.so (new), alternating
set -e
B="--steps 20 --warmup 3 --no-cpu-baseline --no-latency"
for i in 1 2 3; do
  python bench.py $B > gpurun_out/ow_old_$i.json 2>/dev/null
  python tools/pick.py value rounds_ms.round5 < gpurun_out/ow_old_$i.json
  ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so python bench.py $B > gpurun_out/ow_new_$i.json 2>/dev/null
  python tools/pick.py value rounds_ms.round5 < gpurun_out/ow_new_$i.json
done
for ln in 14 18; do
  python bench.py $B --log-n $ln > gpurun_out/ow_old_l$ln.json 2>/dev/null
  python tools/pick.py value rounds_ms.round5 < gpurun_out/ow_old_l$ln.json
  ZKT_LIB_PATH=$PWD/_ab/libzkt_exp.so python bench.py $B --log-n $ln > gpurun_out/ow_new_l$ln.json 2>/dev/null
  python tools/pick.py value rounds_ms.round5 < gpurun_out/ow_new_l$ln.json
done
