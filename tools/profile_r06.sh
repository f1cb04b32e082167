#!/bin/bash
# Round 6: the rocprofv3 set of bench.py's headline workload with this round's tags (the scripts are round 5's, parameterised by tag):
# kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes -> kernel_stats.csv, traffic.json; then the pipe-utilisation counter groups.
bash tools/profile_r05.sh r06_final > gpurun_out/profile_r06_final.log 2>&1
bash tools/profile_r05_pipe.sh r06_pipe > gpurun_out/profile_r06_pipe.log 2>&1
tail -30 gpurun_out/profile_r06_final.log; tail -45 gpurun_out/profile_r06_pipe.log
