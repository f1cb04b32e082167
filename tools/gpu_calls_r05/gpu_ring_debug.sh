mkdir -p gpurun_out/r05_debug
: > gpurun_out/r05_debug/ring_loop.log
for i in 1 2 3 4 5 6 7 8; do
  ( GSV_DRAIN_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generic_ciphertext_sink or ciphertext_ring_whole_pass" 2>&1 | tail -30 ) > gpurun_out/r05_debug/ring_loop_$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/r05_debug/ring_loop_$i.log)" >> gpurun_out/r05_debug/ring_loop.log
done
cat gpurun_out/r05_debug/ring_loop.log
grep -l "failed\|drain debug" gpurun_out/r05_debug/ring_loop_*.log | head
