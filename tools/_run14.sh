R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
export GSV_PLAN_FILE=/dev/shm/gsv_ab.gsvplan
F="--steps 10 --warmup 0 --no-check --no-cpu-baseline --no-e2e"
i=0
for v in _e7 _e8 _e9 "" _e7 _e8 _e9 ""; do
  i=$((i+1))
  if [ $i -le 4 ]; then timeout 600 env GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine$v.so python3 tools/kernel_ab.py > $O/ab_$i.txt 2>&1; fi
  GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine$v.so timeout 900 python3 bench.py $F > $O/bench_$i.json 2> $O/bench_$i.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('bench$v', '%.4e'%d['value'])"
done
cat $O/ab_1.txt $O/ab_2.txt $O/ab_3.txt $O/ab_4.txt
