R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "two_instances or half_window or lockstep or in_slices or plan_file" ) > $O/pytest_subset.log 2>&1
tail -6 $O/pytest_subset.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.out 2> $O/bench_driver.err
tail -1 $O/bench_driver.out > $O/bench_driver_command.json
grep -E "real|bench.py:" $O/bench_driver.err | tail -12
bash tools/profile_r02.sh r02_final_ni4 > $O/profile.log 2>&1
tail -3 $O/profile.log
cat $O/bench_driver_command.json
