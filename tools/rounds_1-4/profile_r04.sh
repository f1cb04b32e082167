#!/bin/bash
# Round-4 profile set of bench.py's headline workload (1 024 instances of the one-public-input verifier plan, sliced steps; a kernel
# dispatch = one WINDOW of the session's schedule).  Run on the GPU box through gpurun; a first bench.py run in the same call leaves the
# plan file in /dev/shm so that the profiled processes load the plan in seconds instead of building it under the profiler.
# Outputs under gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/<tag>/):
#   1. rocprofv3 --kernel-trace --stats over one full pass (10 slices)            -> kernel_stats.csv, bench_profiled.json
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (never combined with a trace domain), each over one full pass
#                                                                                  -> pmc_counters.json
#   3. traffic.json: HBM bytes per launch with the guide's gfx950 correction, stamped with the sha256 of the libgsv_engine.so that was
#      profiled — bench.py quotes `roofline.traffic` from it only when the running library has the same hash.
TAG=${1:-r04_final}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-e2e --no-rate-by-instances --no-mode-rates --no-cc16"
python3 $R/bench.py --steps 1 --warmup 0 $FLAGS > $OUT/bench_plan_build.log 2> $OUT/bench_plan_build.err   # builds + saves the plan file
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_stats.log 2> $OUT/bench_stats.err
tail -1 $OUT/bench_stats.log > $OUT/bench_profiled.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_fetch.log 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_write.log 2> $OUT/bench_write.err
cd $R
python3 - <<PY
import csv, glob, collections, hashlib, json, os, subprocess, sys
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
try:
    prof = json.load(open(os.path.join(out, "bench_profiled.json")))
    rf, cfg = prof["roofline"], prof["config"]
    subprocess.check_call([sys.executable, os.path.join("$R", "tools", "traffic_from_pmc.py"), out, str(rf["algorithmic_bytes_per_launch"]), str(cfg["instances_per_gpu"]), str(cfg["gates_per_instance"])])
except Exception as e:
    print("traffic.json not written:", repr(e))
PY
find $OUT -name "*.csv" -size +4M -delete
tail -c 600 $OUT/bench_profiled.json; tail -n 3 $OUT/bench_fetch.err; tail -n 3 $OUT/bench_write.err
