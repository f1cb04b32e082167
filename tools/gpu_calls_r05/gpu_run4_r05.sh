mkdir -p gpurun_out/r05_e2e
timeout 900 python tools/pair_overlap.py 3 > gpurun_out/r05_e2e/pair_overlap_masked.log 2>&1
GSV_PAIR_CU_MASK=0 timeout 900 python tools/pair_overlap.py 3 > gpurun_out/r05_e2e/pair_overlap_unmasked.log 2>&1
timeout 900 python tools/pair_overlap.py 2 >> gpurun_out/r05_e2e/pair_overlap_masked.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "garble_evaluate or generic_ciphertext_sink" 2>&1 | tail -4 > gpurun_out/r05_e2e/pair_tests.log
rm -f /dev/shm/gsv_pair_overlap_*.gsvplan
cat gpurun_out/r05_e2e/pair_overlap_masked.log gpurun_out/r05_e2e/pair_overlap_unmasked.log gpurun_out/r05_e2e/pair_tests.log
