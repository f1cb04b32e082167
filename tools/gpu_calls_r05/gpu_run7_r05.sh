# round 5: the driver's bench command a second time on the final library (run-to-run spread of `value`)
mkdir -p gpurun_out/r05_final
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_final/bench_driver_command_run2.json 2> gpurun_out/r05_final/bench_driver_command_run2.err
tail -4 gpurun_out/r05_final/bench_driver_command_run2.err; head -c 600 gpurun_out/r05_final/bench_driver_command_run2.json
