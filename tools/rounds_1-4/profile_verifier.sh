#!/bin/bash
# Profile set of the default `python3 bench.py` (the real groth16_verify_compressed circuit as a plan; run on the GPU box through
# gpurun; outputs under gpurun_out/prof_<tag>/):
#   1. rocprofv3 --kernel-trace --stats   (kernel durations of the full default run: warmup step + timed step + hash check)
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over ONE step (no warmup, no check)      [only with PMC=1]
TAG=${1:-verifier}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py > $OUT/bench_stats.log 2>&1
tail -1 $OUT/bench_stats.log > $OUT/bench.json
if [ -n "$PMC" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-check --cpu-baseline-chain 0 > $OUT/bench_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-check --cpu-baseline-chain 0 > $OUT/bench_write.log 2>&1
fi
cd $R
python3 - <<PY
import csv, glob, collections, json, os
out = "$OUT"
res = {}
for d in ("fetch", "write"):
    tot = collections.Counter()
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "run_program" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    if tot:
        res[d] = dict(tot)
json.dump(res, open(os.path.join(out, "pmc_counters.json"), "w"), indent=1)
print(json.dumps(res))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read()[:1500])
    os.replace(f, os.path.join(out, "kernel_stats.csv"))
PY
# keep the merge-back small: only summaries travel
find $OUT -name "*.csv" -size +4M -delete
tail -c 600 $OUT/bench.json
