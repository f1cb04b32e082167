#!/bin/bash
# Round 6: one whole verifier pass at 1 024 instances with the final library and with the same library WITHOUT the per-group step barrier
# (libgsv_engine_nogbar.so: kernels.hip with `group_barrier = false`, built outside the tree), same box; then the opt-in GPU test set.
mkdir -p gpurun_out/r06_final
FLAGS="--no-cpu-baseline --no-e2e --no-rate-by-instances --no-mode-rates --no-cc16 --no-headline-ct-check"
out=gpurun_out/r06_final/pass_ab_group_barrier.log
: > $out
for lib in nogbar final nogbar final; do
  if [ $lib = final ]; then unset GSV_ENGINE_SO; else export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_$lib.so; fi
  echo "== $lib" >> $out
  python3 bench.py --steps 10 --warmup 10 $FLAGS 2>> $out | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g gates/s, %.1f ms per step, slices %s' % (d['value'], d['ms_per_step'], d['config']['step_device_ms']))" >> $out
done
unset GSV_ENGINE_SO
cat $out | grep -v "^bench.py\|amdgpu.ids"
( time timeout 1200 python -m pytest tests/ -q -m "gpu and slow" --durations=10 ) > gpurun_out/r06_final/pytest_gpu_slow_set.log 2>&1
tail -15 gpurun_out/r06_final/pytest_gpu_slow_set.log
