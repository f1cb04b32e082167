#!/usr/bin/env python3
"""Does destroying a device object from a sink callback stall a ring pass?  (The mechanism suspected behind round 5's one-off ring watchdog
failure: Python's cyclic collector finalizing a forgotten Session on the callback thread.)

A ring session garbles fq12_mix (fine units: > 20 drain segments) with GSV_DEP_WAIT_SECONDS=3; the sink handler
  case "destroy":  closes a throw-away Session on its 3rd call        (gsv_session_destroy -> hipFree: synchronises the device)
  case "gc":       runs gc.collect() on its 3rd call while a Session sits in an unreachable reference cycle (what the collector does by itself)
  case "none":     does nothing                                         (control)
and the script reports whether the pass failed with the watchdog status and what the diagnosis said.   usage: ring_gc_repro.py"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import garbled_snark_verifier_amd as gsv

FINE_UNITS = ["fq2::mul_montgomery", "fq2::square_montgomery", "fp254::mul_by_constant_montgomery", "bigint::mul_karatsuba", "fp254::montgomery_reduce"]  # tests/test_gpu_parity.py
os.environ["GSV_CT_RING_RECORDS"] = "1000000"
os.environ["GSV_DEP_WAIT_SECONDS"] = "3"
eng = gsv.Engine(0)
plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
small = gsv.Program.from_circuit("fq_add")
d, f, t, inp = gsv.labels_from_seed(101, plan.info["n_inputs"])


class Holder:  # a Session kept alive only by a reference cycle
    pass


for case in ("none", "destroy", "gc", "none"):
    st = gsv.Session(eng, plan, 1, retain_stream="ring", concurrent_calls=16, drain_segment_records=300_000)
    st.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    victim = gsv.Session(eng, small, 1, 1, 1)
    if case == "gc":
        h = Holder(); h.me = h; h.sess = victim
        del h, victim
        victim = None
    calls = {"n": 0}

    def handler(inst, first, recs):
        calls["n"] += 1
        if calls["n"] == 3:
            if case == "destroy":
                victim.close()
            elif case == "gc":
                gc.enable()      # (the wrapper pauses the collector during the call: undo that here, this is what used to happen)
                gc.collect()

    t0 = time.time()
    try:
        st.garble_to_sink(handler, threads=1, with_hashes=True)
        print("case %-8s pass ok in %.1f s" % (case, time.time() - t0), flush=True)
    except gsv.GsvError as e:
        print("case %-8s pass FAILED after %.1f s: %s" % (case, time.time() - t0, str(e)[:600]), flush=True)
    if victim is not None:
        victim.close()
    st.close()
