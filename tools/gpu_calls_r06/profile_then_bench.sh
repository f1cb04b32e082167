#!/bin/bash
# Round 6, final library: the rocprofv3 set (stats, FETCH / WRITE, pipe counters), then the driver's bench command on the same box
tools/profile_r06.sh
mkdir -p gpurun_out/r06_final
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final/bench_driver_command.json 2> gpurun_out/r06_final/bench_driver_command.err
tail -c 1500 gpurun_out/r06_final/bench_driver_command.json; tail -22 gpurun_out/r06_final/bench_driver_command.err
