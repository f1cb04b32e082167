#!/bin/bash
# Narrow-step phase clock (tools/narrow_phase_clock.py) with the diagnostic library; optional tag = output file suffix.
export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_diag.so
mkdir -p gpurun_out/r04_kernel
out=gpurun_out/r04_kernel/narrow_phase_clock_${1:-baseline}.log
python tools/narrow_phase_clock.py fq_inverse > $out 2>&1
PC_INSTANCES=1 python tools/narrow_phase_clock.py fq_sqrt >> $out 2>&1
cat $out
