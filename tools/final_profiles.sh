#!/bin/bash
# Everything profiles/ holds for a round, taken on ONE box from the current sources.  usage (GPU box): bash tools/final_profiles.sh r04
# -> gpurun_out/final_<tag>/ (copy into profiles/ afterwards: tools/collect_profiles.sh)
TAG=${1:-r05}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
step() { echo "[$(date +%T)] $*" | tee -a $OUT/steps.log; }
step "bench (default: BN254 2^20 withdraw, with the CPU baseline)"
python bench.py --steps 20 --warmup 3 > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err || { tail -5 $OUT/bench_${TAG}.err; exit 1; }
step "rocprof kernel stats, BN254 2^20"
bash tools/prof_stats.sh ${TAG}_bn254_2_20 && mv gpurun_out/${TAG}_bn254_2_20_* $OUT/ || exit 1
step "PMC traffic, BN254 2^20"
bash tools/pmc_traffic.sh > $OUT/pmc_traffic_${TAG}_bn254_2_20.txt || exit 1
step "PMC VALU, BN254 2^20"
bash tools/pmc_valu.sh ${TAG}_bn254_2_20 && mv gpurun_out/pmc_valu_${TAG}_bn254_2_20.txt $OUT/ || exit 1
export BENCH_ARGS="--curve bls12_381 --log-n 22"
step "rocprof kernel stats, BLS12-381 2^22"
bash tools/prof_stats.sh ${TAG}_bls12_381_2_22 && mv gpurun_out/${TAG}_bls12_381_2_22_* $OUT/ || exit 1
step "PMC traffic, BLS12-381 2^22"
bash tools/pmc_traffic.sh > $OUT/pmc_traffic_${TAG}_bls12_381_2_22.txt || exit 1
step "PMC VALU, BLS12-381 2^22"
bash tools/pmc_valu.sh ${TAG}_bls12_381_2_22 && mv gpurun_out/pmc_valu_${TAG}_bls12_381_2_22.txt $OUT/ || exit 1
unset BENCH_ARGS
step "the other configs"
bash tools/bench_configs.sh $TAG > $OUT/bench_configs_${TAG}.txt 2>&1 && mv gpurun_out/bench_${TAG}_*.json $OUT/ || { tail -5 $OUT/bench_configs_${TAG}.txt; exit 1; }
step "Poseidon kernels"
python tools/poseidon_bench.py > $OUT/poseidon_${TAG}.txt 2>/dev/null || exit 1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_valu_1 gpurun_out/pmc_valu_2 gpurun_out/pmc_valu_3
step "done"
