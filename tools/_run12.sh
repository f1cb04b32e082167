R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
export GSV_PLAN_FILE=/dev/shm/gsv_ab.gsvplan
F="--steps 10 --warmup 0 --no-check --no-cpu-baseline --no-e2e"
for v in _e5 "" _e5 ""; do
  n=$(ls $O | grep -c "bench${v}_")
  if [ $n = 0 ]; then timeout 600 env GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine$v.so python3 tools/kernel_ab.py > $O/ab$v.txt 2>&1; fi
  GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine$v.so timeout 900 python3 bench.py $F > $O/bench${v}_$n.json 2> $O/bench${v}_$n.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench${v}_$n.json').read().strip().splitlines()[-1]); print('bench$v', '%.4e'%d['value'])"
done
cat $O/ab_e5.txt $O/ab.txt
for c in "2048 2048" "1024 2048"; do set -- $c
echo "== caps $1 $2"; GSV_AND_CAP=$1 GSV_XOR_CAP=$2 KAB_NOCHECK=1 timeout 600 python3 tools/kernel_ab.py 2>&1 | tee -a $O/ab_caps.txt
done
