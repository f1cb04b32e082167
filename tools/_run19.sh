R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O; cd $R
( time python3 -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1
tail -14 $O/pytest_gpu.log
( time python3 __graft_entry__.py --smoke ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
