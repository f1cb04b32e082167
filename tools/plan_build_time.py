#!/usr/bin/env python3
"""Time and peak RSS of building the verifier plan of the 1-public-input fixture (what bench.py does on a fresh machine).  GSV_PLAN_DEBUG=1 prints the phases."""
import json, time, sys, resource
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import garbled_snark_verifier_amd as gsv
case = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'groth16_verify_compressed_1pub_golden.json')))
units = bench.VERIFIER_UNITS + ["fp254::exp_chunk"]
t=time.time()
plan = gsv.Plan.from_circuit(case["circuit"], units, window_div=4)
print("built in %.1f s, maxrss %.1f GB" % (time.time()-t, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1e6))
