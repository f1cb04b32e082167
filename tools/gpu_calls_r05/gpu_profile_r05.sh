# round-5 profile set with the final library: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (-> kernel_stats.csv, traffic.json),
# then the pipe-utilisation counter groups (-> pipe_util.json); one gpurun call, the plan file is built once
bash tools/profile_r05.sh r05_final > gpurun_out/profile_r05_final.log 2>&1
bash tools/profile_r05_pipe.sh r05_pipe_final > gpurun_out/profile_r05_pipe_final.log 2>&1
tail -30 gpurun_out/profile_r05_final.log; tail -45 gpurun_out/profile_r05_pipe_final.log
