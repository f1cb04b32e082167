#!/usr/bin/env python3
"""Width-capped list scheduling (compile_program step 1b: GSV_AND_CAP / GSV_XOR_CAP) against ASAP levels on the wide shape
(fq12_mix, Fq12-level units) at 1024 instances, four per workgroup.  usage: and_cap_ab.py "and:xor" ...   (0 = no cap)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

B = 1024
eng = gsv.Engine(0)
ref = None
for cfg in sys.argv[1:] or ["0:0"]:
    a, x = cfg.split(":")
    os.environ["GSV_AND_CAP"], os.environ["GSV_XOR_CAP"] = a, x
    t0 = time.time()
    plan = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], window_div=4)
    tb = time.time() - t0
    d, f, t, inp = gsv.labels_from_seed(3, plan.info["n_inputs"])
    sess = gsv.Session(eng, plan, B, retain_stream=False, concurrent_calls=1)
    best = 1e9
    for _ in range(3):
        sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
        t0 = time.perf_counter()
        sess.garble_streaming(discard=True)
        best = min(best, time.perf_counter() - t0)
    out = sess.read_outputs()
    ref = out if ref is None else ref
    steps = int(plan.call_info()[:, 4].sum())
    print("and_cap %5s xor_cap %5s: %8.1f ms -> %.3e gates/s  (%d steps, built in %.1f s)  outputs equal to the first config's: %s" % (
        a, x, best * 1e3, B * plan.info["n_gates"] / best, steps, tb, bool((out == ref).all())), flush=True)
    sess.close()
    plan.close()
