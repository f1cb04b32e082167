timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gate_type or fq_mul_config2 or golden" 2>&1 | tail -1
for n in 512 256; do python3 bench.py --instances $n --replays 6 --cpu-baseline-chain 0 2>&1 | tail -1 | python3 -c "
import sys, json
j=json.loads(sys.stdin.read()); print($n, '%.4e'%j['value'], j.get('ciphertext_hash_match'))"; done
