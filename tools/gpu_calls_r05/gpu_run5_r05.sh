# round 5, run 17: the four-lanes-per-gate (two interleaved blocks) quad form against the build without it, + the new ring watchdog test
bash tools/kernel_ab_r05.sh dual base nodual > /dev/null 2>&1
mkdir -p gpurun_out/r05_debug
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring_watchdog or ciphertext_ring or evaluate or cut_and_choose" ) > gpurun_out/r05_debug/ring_watchdog.log 2>&1
cat gpurun_out/r05_kernel/kernel_ab_dual.log; tail -15 gpurun_out/r05_debug/ring_watchdog.log
