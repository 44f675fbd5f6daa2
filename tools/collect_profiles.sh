#!/bin/bash
# gpurun_out/final_<tag>/ -> profiles/ (the files the judge reads).  usage: bash tools/collect_profiles.sh r04 [suffix]
TAG=${1:-r05}; SFX=${2:-}
S=gpurun_out/final_$TAG
cp $S/bench_$TAG.json profiles/bench_${TAG}${SFX}.json
for c in bn254_2_20 bls12_381_2_22; do
  cp $S/${TAG}_${c}_summary.txt profiles/bench_${TAG}${SFX}_${c}_summary.txt
  cp $S/${TAG}_${c}_kernel_stats.csv profiles/bench_${TAG}${SFX}_${c}_kernel_stats.csv
  cp $S/${TAG}_${c}_timeline.txt profiles/bench_${TAG}${SFX}_${c}_timeline.txt
  cp $S/${TAG}_${c}_profiled_run.json profiles/bench_${TAG}${SFX}_${c}_profiled_run.json
  cp $S/pmc_traffic_${TAG}_${c}.txt profiles/pmc_traffic_${TAG}${SFX}_${c}.txt
  cp $S/pmc_valu_${TAG}_${c}.txt profiles/pmc_valu_${TAG}${SFX}_${c}.txt
done
for c in bn254_2_14 bn254_2_18 bn254_2_19 bls12_381_2_20 bls12_381_2_22; do cp $S/bench_${TAG}_$c.json profiles/bench_${TAG}${SFX}_$c.json; done
cp $S/poseidon_$TAG.txt profiles/poseidon_${TAG}${SFX}.txt
ls profiles | grep "_${TAG}${SFX}" | wc -l
