#!/bin/bash
# Round 6: timing ablations of the CURRENT kernel on the wide shape (fq12_mix, 1 024 instances, four per workgroup) with the diag library
# (build.py --diag; outputs are wrong when GSV_DIAG != 0: only the times count).  Bits: 1 no AES, 4 no operand loads, 8 no stores,
# 32 no step barrier, 64 no record refill, 128 no second record half.
mkdir -p gpurun_out/r06_kernel
out=gpurun_out/r06_kernel/diag_ablation_wide.log
: > $out
export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_diag.so AB_SHAPES=wide
for d in 0 1 4 8 12 5 9 13 32 45 109 237; do
  echo "== GSV_DIAG=$d" >> $out
  GSV_DIAG=$d timeout 300 python tools/kernel_ab3.py 1024 2>&1 | grep "B=1024" >> $out
done
cat $out
