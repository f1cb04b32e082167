# the driver's round-end sequence as the builder runs it: the whole GPU suite, then smoke, then the driver's bench command
mkdir -p gpurun_out/r05_final
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r05_final/pytest_gpu_full_suite.log 2>&1
python __graft_entry__.py --smoke > gpurun_out/r05_final/smoke.log 2>&1
( time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_final/bench_driver_command.json 2> gpurun_out/r05_final/bench_driver_command.err
tail -5 gpurun_out/r05_final/pytest_gpu_full_suite.log; tail -2 gpurun_out/r05_final/smoke.log; tail -12 gpurun_out/r05_final/bench_driver_command.err; head -c 1500 gpurun_out/r05_final/bench_driver_command.json
