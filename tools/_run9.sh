R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
export GSV_PLAN_FILE=/dev/shm/gsv_ab.gsvplan
F="--steps 10 --warmup 0 --no-check --no-cpu-baseline --no-e2e"
for v in b e3 e4; do
  GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine_$v.so timeout 600 python3 tools/kernel_ab.py > $O/ab_$v.txt 2>&1
done
for v in b e3 e4 e1 b e3 e4; do
  n=$(ls $O | grep -c "bench_$v")
  GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine_$v.so timeout 900 python3 bench.py $F > $O/bench_${v}_$n.json 2> $O/bench_${v}_$n.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench_${v}_$n.json').read().strip().splitlines()[-1]); print('bench_$v', '%.4e'%d['value'], d.get('step_device_ms'))"
done
cat $O/ab_*.txt
