#!/usr/bin/env python3
"""Round-3 kernel A/B on three shapes of work, each through a plan session (window launches), ciphertexts discarded:
  wide    fq12_mix with Fq12-level units (Miller-loop / final-exponentiation shape)
  ladder  fq_sqrt as exp_chunk units (the decompression ladders: narrow steps)
  inverse fq_inverse (binary extended Euclid: the narrowest steps)
at `instances` (default 1024: four per workgroup) and at ONE instance.  GSV_ENGINE_SO selects the library.
usage: kernel_ab3.py [instances]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = gsv.Engine(0)
print("library:", os.environ.get("GSV_ENGINE_SO", "libgsv_engine.so"), flush=True)
SHAPES = (("wide   ", "fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"]), ("ladder ", "fq_sqrt", ["fp254::exp_chunk"]),
          ("inverse", "fq_inverse", ["inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"]),
          ("inv_grp", "fq_inverse", ["inverse::iteration_group", "inverse::divide_chains"]))  # round 5: the inversion as 3 long calls
MILLER_UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::mul_by_034_montgomery", "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery",
                "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery"]
if os.environ.get("AB_SHAPES"):  # e.g. AB_SHAPES=wide,miller — `miller`: the whole Miller loop with the verifier's units (doubling / addition steps and line evaluations: the MID-width two-wire programs)
    want = os.environ["AB_SHAPES"].split(",")
    SHAPES = tuple(x for x in SHAPES + (("miller ", "miller_loop", MILLER_UNITS),) if x[0].strip() in want)
for name, spec, units in SHAPES:
    plan = gsv.Plan.from_circuit(spec, units, window_div=4)
    ref = None
    for b in (B, 1):
        d, f, t, inp = gsv.labels_from_seed(3, plan.info["n_inputs"])
        sess = gsv.Session(eng, plan, b, retain_stream=False, concurrent_calls=1)
        best = 1e9
        for _ in range(3):
            sess.set_garble_inputs(np.tile(d, (b, 1)), np.tile(np.stack([f, t]), (b, 1, 1)), np.tile(inp, (b, 1, 1)))
            t0 = time.perf_counter()
            sess.garble_streaming(discard=True)
            best = min(best, time.perf_counter() - t0)
        out = sess.read_outputs()
        ref = out[0] if ref is None else ref
        ok = bool((out == ref[None]).all())
        print("%s %-10s B=%4d (ni %d): %8.1f ms -> %.3e gates/s  outputs consistent: %s" % (name, spec, b, sess.instances_per_workgroup, best * 1e3, b * plan.info["n_gates"] / best, ok), flush=True)
        sess.close()
    plan.close()
