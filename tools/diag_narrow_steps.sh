mkdir -p gpurun_out/diag
export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_diag.so
for d in 0 1 4 8 5 13; do
  echo "== GSV_DIAG=$d" >> gpurun_out/diag/inv_diag.log
  GSV_DIAG=$d python tools/step_profile.py fq_inverse 1 1024 2>&1 | grep -v "^class\|fit\|Traceback" >> gpurun_out/diag/inv_diag.log
done
tail -60 gpurun_out/diag/inv_diag.log
