#!/bin/bash
# Narrow-step ablation on fq_inverse (diag library: build.py --diag).  Bits: 1 no AES, 4 no label loads, 8 no stores; the step skeleton:
# 32 no barrier, 64 no record prefetch, 128 no second-half load.  Results are wrong by construction, only the step clocks count.
mkdir -p gpurun_out/diag
export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_diag.so
out=gpurun_out/diag/inv_diag2.log
: > $out
for d in ${@:-0 13 45 77 141 237 32 64 128}; do
  echo "== GSV_DIAG=$d" >> $out
  GSV_DIAG=$d python tools/step_profile.py fq_inverse 1 2>&1 | grep "^== \|\[1,8)\|\[32,64)" >> $out
done
cat $out
