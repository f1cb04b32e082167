R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02j; mkdir -p $O; cd $R
export KAB_NOCHECK=1
for d in 0 1 4 8 5 12 13 16; do
  echo "=== GSV_DIAG=$d" >> $O/diag.txt
  GSV_DIAG=$d timeout 300 python3 tools/kernel_ab.py >> $O/diag.txt 2>&1
done
timeout 300 python3 tools/step_profile.py fq12_mul 512 > $O/step_profile_fq12_mul.txt 2>&1
timeout 300 python3 tools/step_profile.py fq12_sqmul 512 > $O/step_profile_fq12_sqmul.txt 2>&1
cat $O/diag.txt; cat $O/step_profile_fq12_mul.txt
