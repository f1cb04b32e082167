#!/usr/bin/env python3
"""Why is a plan of inversion chunks slower per step than the flat inversion program?  One instance; flat program with the full and
with a quarter LDS window, the chunked plan with both windows; prints ms, steps, us/step.  (GSV_ENGINE_SO selects the library.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
spec = sys.argv[1] if len(sys.argv) > 1 else "fq_inverse"
units = {"fq_inverse": ["inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"], "fq_sqrt": ["fp254::exp_chunk"]}[spec]
print("library:", os.environ.get("GSV_ENGINE_SO", "libgsv_engine.so"))
for slots in (None, "1440"):
    if slots:
        os.environ["GSV_LDS_SLOTS"] = slots
    prog = gsv.Program.from_circuit(spec)
    os.environ.pop("GSV_LDS_SLOTS", None)
    d, f, t, inp = gsv.labels_from_seed(3, prog.info["n_inputs"])
    sess = gsv.Session(eng, prog, 1, 1, 1)
    best = 1e9
    for _ in range(3):
        sess.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
        sess.garble(0); sess.sync()
        best = min(best, sess.last_kernel_ms())
    print("flat  window %-5s: %8.1f ms, %7d steps, %.2f us/step, and_terms %d, hbm reads %.0f %%" % (slots or "5760", best, prog.info["n_steps"], best * 1e3 / prog.info["n_steps"], prog.info["and_terms"],
                                                                                          100.0 * prog.info["reads_hbm"] / (prog.info["reads_hbm"] + prog.info["reads_lds"])))
    sess.close(); prog.close()
for div in (None, 4):
    plan = gsv.Plan.from_circuit(spec, units, window_div=div)
    d, f, t, inp = gsv.labels_from_seed(3, plan.info["n_inputs"])
    for conc in (1, 0):
        sess = gsv.Session(eng, plan, 1, retain_stream=False, concurrent_calls=conc)
        si = sess.schedule_info()
        best = 1e9
        for _ in range(3):
            sess.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
            t0 = time.perf_counter()
            sess.garble_streaming(discard=True)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print("plan  window/%s conc %d: %8.1f ms, %d calls, %7d steps (critical %d), %.2f us per critical step" % (div or 1, conc, best, si["n_calls"], si["total_steps"], si["critical_steps"], best * 1e3 / si["critical_steps"]))
        sess.close()
    plan.close()
