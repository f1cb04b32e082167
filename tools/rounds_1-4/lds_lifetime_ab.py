#!/usr/bin/env python3
"""GSV_LDS_LIFETIME (compile_program: a wire gets a window slot only if it dies within this many steps) on the narrow shapes,
through plan sessions at 1024 instances (four per workgroup) and at ONE instance.  usage: lds_lifetime_ab.py lifetime ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
SHAPES = (("ladder ", "fq_sqrt", ["fp254::exp_chunk"]),
          ("inverse", "fq_inverse", ["inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"]),
          ("wide   ", "fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"]))
ref = {}
for life in sys.argv[1:] or ["8"]:
    os.environ["GSV_LDS_LIFETIME"] = life
    for name, spec, units in SHAPES:
        plan = gsv.Plan.from_circuit(spec, units, window_div=4)
        for b in (1024, 1):
            d, f, t, inp = gsv.labels_from_seed(3, plan.info["n_inputs"])
            sess = gsv.Session(eng, plan, b, retain_stream=False, concurrent_calls=1)
            best = 1e9
            for _ in range(3):
                sess.set_garble_inputs(np.tile(d, (b, 1)), np.tile(np.stack([f, t]), (b, 1, 1)), np.tile(inp, (b, 1, 1)))
                t0 = time.perf_counter()
                sess.garble_streaming(discard=True)
                best = min(best, time.perf_counter() - t0)
            out = sess.read_outputs()[0]
            ref.setdefault(spec, out)
            print("lifetime %5s %s %-10s B=%4d: %8.1f ms -> %.3e gates/s  outputs as first config: %s" % (life, name, spec, b, best * 1e3, b * plan.info["n_gates"] / best, bool((out == ref[spec]).all())), flush=True)
            sess.close()
        plan.close()
