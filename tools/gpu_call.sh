#!/bin/bash
# one gpurun call = a list of steps, each logged to gpurun_out/<tag>_<step>.log with its exit code; stops at the first failure
# usage (GPU box): bash tools/gpu_call.sh <tag> "<step name>::<command>" ...
TAG=$1; shift
mkdir -p gpurun_out
for item in "$@"; do
  name=${item%%::*}; cmd=${item#*::}
  echo "[$(date +%T)] $name: $cmd" | tee -a gpurun_out/${TAG}_steps.log
  bash -o pipefail -c "$cmd" > gpurun_out/${TAG}_${name}.log 2>&1
  rc=$?
  echo "[$(date +%T)] $name rc=$rc" | tee -a gpurun_out/${TAG}_steps.log
  tail -n 6 gpurun_out/${TAG}_${name}.log
  if [ $rc -ne 0 ]; then exit $rc; fi
done
