R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_final_exp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/rounds_1-4/final_exp_rate.py 512 > $OUT/run.log 2>&1
cd $R
grep -v "^W2026\|^E2026" $OUT/run.log | tail -3
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cat $f | cut -c1-220; done
find $OUT -name "*kernel_trace.csv" -size +3M -delete
