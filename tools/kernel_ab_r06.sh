#!/bin/bash
# Round-6 A/B of engine builds on the three shapes of tools/kernel_ab3.py (+ a parity subset against the oracle for every variant).
# usage: kernel_ab_r05.sh <tag> <variant> [variant ...]   ("base" = libgsv_engine.so, else libgsv_engine_<variant>.so)
TAG=$1; shift
mkdir -p gpurun_out/r06_kernel
out=gpurun_out/r06_kernel/kernel_ab_$TAG.log
: > $out
for v in "$@"; do
  if [ "$v" = base ]; then unset GSV_ENGINE_SO; else export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_$v.so; fi
  echo "== $v" >> $out
  timeout 900 python tools/kernel_ab3.py ${AB_INSTANCES:-1024} >> $out 2>&1
  if [ -z "$AB_NO_PARITY" ]; then
    timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "both_and_record_forms or fq_mul_config2 or random_circuits_differential or two_instances_per_workgroup or dataflow_between_calls" 2>&1 | tail -2 >> $out
  fi
done
cat $out
