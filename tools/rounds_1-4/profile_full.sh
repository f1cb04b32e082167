#!/bin/bash
# Full-size profile set of `python3 bench.py --workload synthetic` (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/).
#   1. rocprofv3 --kernel-trace --stats   (kernel durations)
#   2. rocprofv3 --pmc FETCH_SIZE         (separate pass, one full launch)
#   3. rocprofv3 --pmc WRITE_SIZE         (separate pass)
#   4. rocprofv3 --pmc <SQ / TCC counters> on a short run
TAG=${1:-fused}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload synthetic > $OUT/bench_stats.log 2>&1
tail -1 $OUT/bench_stats.log > $OUT/bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --workload synthetic --steps 1 --warmup 0 --no-check --cpu-baseline-chain 0 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --workload synthetic --steps 1 --warmup 0 --no-check --cpu-baseline-chain 0 > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/misc -- python3 $R/bench.py --workload synthetic --steps 1 --warmup 0 --replays 8 --no-check --cpu-baseline-chain 0 > $OUT/bench_misc.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, json, os
out = "$OUT"
res = {}
for d in ("fetch", "write", "misc"):
    tot = collections.Counter()
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "run_program" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    res[d] = dict(tot)
json.dump(res, open(os.path.join(out, "pmc_counters.json"), "w"), indent=1)
print(json.dumps(res))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read()[:1500])
PY
# keep the merge-back small: only summaries travel
find $OUT -name "*.csv" -size +2M -delete
