#!/usr/bin/env python3
"""Where a narrow step spends its time, from inside the kernel (diag library only: `python garbled_snark_verifier_amd/build.py --diag`,
GSV_ENGINE_SO=.../libgsv_engine_diag.so, GSV_DIAG=256).  Wave 0 of workgroup 0 stamps the 100 MHz wall clock at seven points of every
narrow step in which it garbles an AND gate (kernels.hip, GSV_PC_STAMP) and accumulates the intervals; 10 ns resolution, unbiased over
the 10^5 steps of a program.  usage: narrow_phase_clock.py [circuit ...]   (default: fq_inverse fq_sqrt)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import garbled_snark_verifier_amd as gsv

NAMES = ["behind the previous barrier -> record decoded (loop top, descriptor, branch, second-half load, decode)", "-> operands arrived", "-> AES done",
         "-> label + ciphertext stores issued, label store completed", "-> refill issued (rest of the step body)", "-> behind the barrier"]
os.environ["GSV_DIAG"] = str(int(os.environ.get("GSV_DIAG", "0")) | 256)
eng = gsv.Engine(0)
for spec in (sys.argv[1:] or ["fq_inverse", "fq_sqrt"]):
    prog = gsv.Program.from_circuit(spec)
    n_in = prog.info["n_inputs"]
    for B in [int(x) for x in os.environ.get("PC_INSTANCES", "1,1024").split(",")]:
        d, f, t, inp = gsv.labels_from_seed(1, n_in)
        sess = gsv.Session(eng, prog, B, 1, 1)
        sess.enable_step_clock()
        sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
        for _ in range(2):
            sess.garble(0)
            sess.sync()
        ms = sess.last_kernel_ms()
        acc = sess.read_step_clock()[:8].astype(np.float64)
        n = max(1.0, acc[7])
        print("== %s B=%d (%d per workgroup), GSV_DIAG=%s: %.2f ms, %d steps (%.3f us/step overall); %d narrow steps with an AND in wave 0:" % (
            spec, B, sess.instances_per_workgroup, os.environ["GSV_DIAG"], ms, prog.info["n_steps"], ms * 1e3 / prog.info["n_steps"], int(acc[7])))
        tot = 0.0
        for i in range(6):
            ns = acc[i] * 10.0 / n
            tot += ns
            print("   %7.1f ns  %s" % (ns, NAMES[i]))
        print("   %7.1f ns  sum" % tot)
        sess.close()
