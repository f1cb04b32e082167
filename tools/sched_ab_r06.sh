#!/bin/bash
# Round 6: A/B of the COMPILER's step schedule on the kernel_ab3 shapes (one box, one library): "base" = ASAP levels, else
# "<key>:<and_cap>:<xor_cap>:<bucket>" = width-capped list scheduling with GSV_SCHED_KEY (0 longest path first, 1 stream order,
# 2 longest path in buckets of <bucket> levels then stream order).  A parity subset against the oracle runs under every setting.
# GSV_SCHED_KEY exists only with profiles/r06_kernel/sched_key_experiment.patch applied to csrc/engine/program.hpp (the experiment was not
# adopted); without it every <key> behaves like 0.
# usage: sched_ab_r06.sh <tag> <config> [<config> ...]
TAG=$1; shift
mkdir -p gpurun_out/r06_kernel
out=gpurun_out/r06_kernel/sched_ab_$TAG.log
: > $out
for cfg in "$@"; do
  unset GSV_SCHED_KEY GSV_AND_CAP GSV_XOR_CAP GSV_SCHED_BUCKET
  if [ "$cfg" != base ]; then
    IFS=: read k a x b <<< "$cfg"
    export GSV_SCHED_KEY=$k GSV_AND_CAP=$a GSV_XOR_CAP=$x GSV_SCHED_BUCKET=$b
  fi
  echo "== $cfg" >> $out
  timeout 1200 python tools/kernel_ab3.py ${AB_INSTANCES:-1024} >> $out 2>&1
  if [ -z "$AB_NO_PARITY" ]; then
    timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "both_and_record_forms or fq_mul_config2 or random_circuits_differential or two_instances_per_workgroup or dataflow_between_calls" 2>&1 | tail -2 >> $out
  fi
done
cat $out
