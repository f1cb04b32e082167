mkdir -p gpurun_out/r05_kernel
AB_NO_PARITY=1 bash tools/kernel_ab_r05.sh inv_grp base > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_ext_host.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_kernel/ext_host_gpu.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "drain_instances or compressed_verifier_as_a_plan or final_exponentiation or plan_recorder or lockstep" 2>&1 | tail -5 > gpurun_out/r05_kernel/parity_subset.log
cat gpurun_out/r05_kernel/kernel_ab_inv_grp.log gpurun_out/r05_kernel/ext_host_gpu.log gpurun_out/r05_kernel/parity_subset.log; cat gpurun_out/ext_host_verifier.json
