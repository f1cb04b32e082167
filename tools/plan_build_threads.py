#!/usr/bin/env python3
"""Round 6: how the verifier plan's build time depends on the number of warm-up recorders (GSV_PLAN_WARMUP_THREADS) and compile workers
(GSV_COMPILE_THREADS) — recording is the critical path of a build.  Also times the dual build (both of bench.py's plan files from one
build).  usage: plan_build_threads.py [single|pair] (environment selects the threads); prints one JSON line."""
import json, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import garbled_snark_verifier_amd as gsv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
case = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'groth16_verify_compressed_1pub_golden.json')))
units = bench.VERIFIER_UNITS + ["fp254::exp_chunk"]
small = bench.SMALL_BATCH_UNITS + ["fp254::exp_chunk"]
mode = sys.argv[1] if len(sys.argv) > 1 else "single"
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
a, b = os.path.join(d, "gsv_pbt_%d_a.gsvplan" % os.getpid()), os.path.join(d, "gsv_pbt_%d_b.gsvplan" % os.getpid())
t = time.time()
try:
    if mode == "pair":
        gsv.Plan.build_file_pair(case["circuit"], units, a, 4, b, 1, units_b=small)
    else:
        gsv.Plan.build_file(case["circuit"], units, a, window_div=4)
    dt = time.time() - t
    out = {"mode": mode, "seconds": round(dt, 1), "maxrss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1),
           "warmup_threads": os.environ.get("GSV_PLAN_WARMUP_THREADS", "default"), "compile_threads": os.environ.get("GSV_COMPILE_THREADS", "default"),
           "file_gb": [round(os.path.getsize(p) / 1e9, 1) for p in (a, b) if os.path.exists(p)]}
    if mode == "pair" and os.environ.get("PBT_DIGEST"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import plan_digest
        out["digests"] = [plan_digest.digest(p, threads=16)["digest"] for p in (a, b)]
    print(json.dumps(out), flush=True)
finally:
    for p in (a, b):
        if os.path.exists(p):
            os.remove(p)
