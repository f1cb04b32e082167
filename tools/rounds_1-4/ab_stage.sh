#!/bin/bash
# usage: ab_stage.sh <tag> [parity]  — kernel_ab3 for the staged-narrow-loop build against the production build (+ an optional parity subset)
TAG=${1:-stage}
mkdir -p gpurun_out/r04_kernel
out=gpurun_out/r04_kernel/kernel_ab_$TAG.log
: > $out
export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_stage.so
if [ -n "$2" ]; then
  echo "== stage: parity" >> $out
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "both_and_record_forms or fq_mul_config2 or two_instances_per_workgroup or dataflow_between_calls or driver_mix" 2>&1 | tail -15 >> $out
fi
echo "== stage" >> $out
timeout 600 python tools/kernel_ab3.py ${AB_INSTANCES:-1024} >> $out 2>&1
unset GSV_ENGINE_SO
echo "== base" >> $out
timeout 600 python tools/kernel_ab3.py ${AB_INSTANCES:-1024} >> $out 2>&1
cat $out
