R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O; cd $R
GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine_pc.so timeout 300 python3 tools/_phase_clock.py fq12_sqmul 512 > $O/phase_clock.txt 2>&1
GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine_pc.so timeout 300 python3 tools/_phase_clock.py fq12_sqmul 256 >> $O/phase_clock.txt 2>&1
cat $O/phase_clock.txt
export GSV_PLAN_FILE=/dev/shm/gsv_ab.gsvplan
F="--steps 10 --warmup 0 --no-check --no-cpu-baseline --no-e2e"
for v in _e1 "" _e1 ""; do
  timeout 600 env KAB_NOCHECK=0 GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine$v.so python3 tools/kernel_ab.py > $O/ab$v.txt 2>&1
  n=$(ls $O | grep -c "bench${v}_")
  GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine$v.so timeout 900 python3 bench.py $F > $O/bench${v}_$n.json 2> $O/bench${v}_$n.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench${v}_$n.json').read().strip().splitlines()[-1]); print('bench$v', '%.4e'%d['value'])"
done
cat $O/ab_e1.txt $O/ab.txt
