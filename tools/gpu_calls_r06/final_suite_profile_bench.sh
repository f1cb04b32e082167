#!/bin/bash
# Round 6, last commit: the GPU suite, the rocprofv3 stats + FETCH / WRITE set, then the driver's bench command quoting that traffic
tools/gpu_calls_r06/gpu_suite.sh
bash tools/profile_r05.sh r06_final > gpurun_out/profile_r06_final.log 2>&1
tail -12 gpurun_out/profile_r06_final.log
cp gpurun_out/prof_r06_final/traffic.json profiles/r06_final/traffic.json   # (on the box: so that the bench run below quotes it; the same file is committed afterwards)
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final/bench_driver_command.json 2> gpurun_out/r06_final/bench_driver_command.err
tail -c 800 gpurun_out/r06_final/bench_driver_command.json; tail -20 gpurun_out/r06_final/bench_driver_command.err
