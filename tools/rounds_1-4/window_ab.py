#!/usr/bin/env python3
"""Small-batch rate of the three kernel_ab3 shapes as a function of the LDS label window the programs were compiled for (a plan built
with window_div = 4 serves 1, 2 and 4 instances per workgroup from one image but gives ONE instance a quarter of the window) and of
GSV_LDS_LIFETIME.  usage: window_ab.py [instances csv]   (GSV_ENGINE_SO selects the library)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

insts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,16").split(",")]
eng = gsv.Engine(0)
print("library:", os.environ.get("GSV_ENGINE_SO", "libgsv_engine.so"), "GSV_LDS_LIFETIME", os.environ.get("GSV_LDS_LIFETIME"), flush=True)
SHAPES = (("wide   ", "fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"]), ("ladder ", "fq_sqrt", ["fp254::exp_chunk"]),
          ("inverse", "fq_inverse", ["inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"]))
for name, spec, units in SHAPES:
    for div in (4, None):
        plan = gsv.Plan.from_circuit(spec, units, window_div=div)
        for b in insts:
            d, f, t, inp = gsv.labels_from_seed(3, plan.info["n_inputs"])
            sess = gsv.Session(eng, plan, b, retain_stream=False, concurrent_calls=1)
            best = 1e9
            for _ in range(3):
                sess.set_garble_inputs(np.tile(d, (b, 1)), np.tile(np.stack([f, t]), (b, 1, 1)), np.tile(inp, (b, 1, 1)))
                t0 = time.perf_counter()
                sess.garble_streaming(discard=True)
                best = min(best, time.perf_counter() - t0)
            print("%s %-10s window/%s B=%4d (ni %d): %8.1f ms -> %.3e gates/s" % (name, spec, div or 1, b, sess.instances_per_workgroup, best * 1e3, b * plan.info["n_gates"] / best), flush=True)
            sess.close()
        plan.close()
