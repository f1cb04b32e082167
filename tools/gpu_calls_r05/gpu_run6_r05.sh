# round 5, last GPU call: the opt-in sub-circuit plans ('gpu and slow') with the final library, then the ring test in a loop (the
# watchdog status that fired once this round: any event leaves its diagnosis in gpurun_out/ring_watchdog_event.txt)
mkdir -p gpurun_out/r05_debug
( time timeout 900 python -m pytest tests -m "gpu and slow" -q --durations=5 ) > gpurun_out/r05_debug/gpu_slow_set.log 2>&1
: > gpurun_out/r05_debug/ring_loop_final.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ciphertext_ring_whole_pass" 2>&1 | tail -1 >> gpurun_out/r05_debug/ring_loop_final.log
done
tail -6 gpurun_out/r05_debug/gpu_slow_set.log; cat gpurun_out/r05_debug/ring_loop_final.log; cat gpurun_out/ring_watchdog_event.txt 2>/dev/null
