#!/bin/bash
# Round-2 profile set of bench.py's workload (512 instances of the verifier plan, sliced steps).  Run on the GPU box through gpurun
# AFTER a bench.py run of the same call has left the plan file in /dev/shm (the profiled processes then load the plan in seconds
# instead of building it with 16 compile threads under the profiler).  Outputs under gpurun_out/prof_<tag>/:
#   1. rocprofv3 --kernel-trace --stats over one full pass (10 slices);
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, each over one full pass (10 slices).
TAG=${1:-r02_verifier}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--no-check --no-cpu-baseline --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_stats.log 2> $OUT/bench_stats.err
tail -1 $OUT/bench_stats.log > $OUT/bench_profiled.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_fetch.log 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_write.log 2> $OUT/bench_write.err
cd $R
python3 - <<PY
import csv, glob, collections, json, os
out = "$OUT"
res = {}
for d in ("fetch", "write"):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "run_program" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
    if tot:
        res[d] = {"sum": dict(tot), "dispatches": dict(n)}
json.dump(res, open(os.path.join(out, "pmc_counters.json"), "w"), indent=1)
print(json.dumps(res))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read()[:1500])
    os.replace(f, os.path.join(out, "kernel_stats.csv"))
PY
find $OUT -name "*.csv" -size +4M -delete
tail -c 400 $OUT/bench_profiled.json; tail -3 $OUT/bench_fetch.err $OUT/bench_write.err
