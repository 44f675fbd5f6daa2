timeout -k 10 300 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ntt.py tests/test_gpu_poseidon.py tests/test_gpu_quotient.py -x -q > gpurun_out/r03_t18.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t18.log; tail -3 gpurun_out/r03_t18.log; grep -q "rc=0" gpurun_out/r03_t18.log && { B="--steps 20 --warmup 3 --no-cpu-baseline"; run() { tag=$1; shift; env "$@" python bench.py $B > gpurun_out/r03_q_$tag.json 2> gpurun_out/r03_q_$tag.err; python -c "
import json,sys
d=json.loads(open(\"gpurun_out/r03_q_$tag.json\").read().strip().splitlines()[-1])
l=d[\"latency\"]; k=d[\"kernels\"]
print(\"$tag\", d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"avg_launch_ms\"], d[\"int_alu\"][\"msm_main_stream_avg_ms\"], d[\"int_alu\"][\"msm_tail_avg_ms\"], k[\"ntt_2^20\"][\"avg_ms\"], k[\"ntt_2^22\"][\"avg_ms\"], k[\"quotient\"][\"avg_ms\"], l[\"cold_single_proof_ms\"], l[\"unchained_single_proof_ms\"])
"; }; for i in 1 2; do run new$i X=1 && run prev$i ZKT_LIB_PATH=$PWD/_ab/libzkt_r03h.so; done; }
