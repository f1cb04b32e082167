#!/bin/bash
# Round 6: the driver's two GPU commands back to back on one box: `pytest -m gpu` (with durations), then `bench.py --gpus 1 --steps 20 --warmup 5`.
tools/gpu_calls_r06/gpu_suite.sh
mkdir -p gpurun_out/r06_final
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final/bench_driver_command.json 2> gpurun_out/r06_final/bench_driver_command.err
tail -c 3000 gpurun_out/r06_final/bench_driver_command.json; tail -25 gpurun_out/r06_final/bench_driver_command.err
