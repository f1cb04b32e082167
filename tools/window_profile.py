#!/usr/bin/env python3
"""Where does a verifier pass spend its device time?  One pass at `instances` (default 1024) with every window of the session's
schedule launched and timed on its own (ciphertexts discarded), then bucketed by the window's mean step width (fused gates per
device step): the latency-bound calls (inversions, ladders, multiplexer trees) against the wide ones.  Diagnostic tool.
usage: window_profile.py [instances] [out.npz]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import garbled_snark_verifier_amd as gsv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
case = json.load(open(os.path.join(ROOT, "tests", "golden", bench.FIXTURE["verifier_compressed"])))
units = bench.VERIFIER_UNITS + ["fp254::exp_chunk"]
eng = gsv.Engine(0)
path = os.environ.get("GSV_PLAN_FILE")
if path and os.path.exists(path):
    plan = gsv.Plan.load(path, eng)
else:
    plan = gsv.Plan.from_circuit(case["circuit"], units, window_div=4)
    if path:
        plan.save(path)
ci = plan.call_info().astype(np.int64)
w = bench.VerifierWork(gsv, eng, plan, B, list(range(100, 100 + B)), concurrent_calls=1)
wins = w.sess.windows()
print("instances %d (%d per workgroup), %d calls, %d windows" % (B, w.sess.instances_per_workgroup, len(ci), len(wins)), flush=True)
rows = []
w.new_pass()
for first, n, _ in wins:
    ms = w.run_slice(first, n)
    g = ci[first:first + n]
    rows.append((first, n, g[:, 1].sum(), g[:, 3].sum(), g[:, 4].sum(), ms))
r = np.array(rows, dtype=np.float64)
gates, cts, steps, ms = r[:, 2], r[:, 3], r[:, 4], r[:, 5]
width = gates / np.maximum(steps, 1)
print("pass: %.1f s device, %.3e gates/s" % (ms.sum() / 1e3, B * gates.sum() / ms.sum() * 1e3))
print("%-22s %8s %12s %9s %7s %12s %9s" % ("gates per step", "windows", "gates", "time s", "% time", "gates/s", "us/step"))
edges = [0, 25, 50, 100, 200, 400, 800, 1600, 3200, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (width >= lo) & (width < hi)
    if m.any():
        print("[%5d, %10d) %8d %12.4e %9.2f %7.1f %12.3e %9.2f" % (lo, hi, m.sum(), gates[m].sum(), ms[m].sum() / 1e3, 100 * ms[m].sum() / ms.sum(),
                                                                 B * gates[m].sum() / ms[m].sum() * 1e3, ms[m].sum() * 1e3 / steps[m].sum()))
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], rows=r)
w.close()
