for cfg in "A=1" "GSV_LDS_LIFETIME=4" "GSV_LDS_LIFETIME=16" "GSV_LDS_LIFETIME=64" "GSV_HBM_ARENA=2" "GSV_HBM_ARENA=1" "GSV_FUSE_DUP=3" "GSV_ORDER_BY_READER=0"; do
  echo "== $cfg"
  env $cfg python3 bench.py --replays 8 --cpu-baseline-chain 0 --no-check 2>&1 | tail -1 | python3 -c "
import sys, json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done
