R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O; cd $R
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.out 2> $O/bench_driver.err
tail -1 $O/bench_driver.out > $O/bench_driver_command.json
grep real $O/bench_driver.err
bash tools/profile_r02.sh r02_final > $O/profile.log 2>&1
( time python3 -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
for l in 4 16; do echo "== GSV_LDS_LIFETIME=$l" >> $O/knobs.txt; GSV_LDS_LIFETIME=$l KAB_NOCHECK=1 timeout 300 python3 tools/kernel_ab.py >> $O/knobs.txt 2>&1; done
cat $O/knobs.txt; cat $O/bench_driver_command.json
