#!/bin/bash
# Round 6: plan build time against recorder / compile thread counts on the GPU box's host (16-core quota), then the dual build.
mkdir -p gpurun_out/r06_e2e
out=gpurun_out/r06_e2e/plan_build_threads.log
: > $out
nproc >> $out; cat /sys/fs/cgroup/cpu.max >> $out 2>/dev/null
for cfg in "4 16" "8 16" "12 16" "8 24"; do
  set -- $cfg
  GSV_PLAN_WARMUP_THREADS=$1 GSV_COMPILE_THREADS=$2 timeout 400 python tools/plan_build_threads.py single >> $out 2>&1
done
GSV_PLAN_WARMUP_THREADS=8 timeout 600 python tools/plan_build_threads.py pair >> $out 2>&1
cat $out
