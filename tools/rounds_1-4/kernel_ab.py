#!/usr/bin/env python3
"""Device rates of one engine build on three shapes of work, for A/B comparisons of kernel changes:
  wide    fq12_sqmul chain replayed (the Miller-loop / final-exponentiation shape), 512 instances
  narrow  fq_sqrt as a plan of exp_chunk units (the decompression ladders: thousands of narrow steps), 512 instances
  inverse fq12_inverse (one Fq inversion inside: binary extended Euclid), 512 instances
plus a hash check of each against the CPU oracle on one instance (KAB_NOCHECK=1 skips it: timing experiments under GSV_DIAG —
which only the diagnostic library honours, `build.py --diag` -> libgsv_engine_diag.so — give wrong outputs by design).  GSV_ENGINE_SO selects the library (garbled_snark_verifier_amd/build.py).
usage: kernel_ab.py [instances]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import garbled_snark_verifier_amd as gsv
import oracle_lib as o

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CHECK = os.environ.get("KAB_NOCHECK") != "1"
eng = gsv.Engine(0)
print("library:", os.environ.get("GSV_ENGINE_SO", "libgsv_engine.so"), " instances:", B, flush=True)


def tiled(n_in, seed=1):
    d, f, t, inp = gsv.labels_from_seed(seed, n_in)
    return np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1))


# wide
prog = gsv.Program.from_circuit("fq12_sqmul", chain_feedback=True)
R = 6
sess = gsv.Session(eng, prog, B, R, int(os.environ.get("KAB_CT_CAP", "2")))
D, K, I = tiled(prog.info["n_inputs"])
best = 1e9
for _ in range(3):
    sess.set_garble_inputs(D, K, I)
    sess.garble(0); sess.sync()
    best = min(best, sess.last_kernel_ms())
print("wide    fq12_sqmul x%d : %8.1f ms -> %.3e gates/s" % (R, best, B * prog.info["n_gates"] * R / best * 1e3), flush=True)
sess.close()
if CHECK:
    chk = gsv.CircuitBuilder.streaming_garbling("fq12_sqmul", [5, 5, 5], engine=eng, program=prog, replays=2, keep_ciphertexts=False)
    ref = o.garble("fq12_sqmul_chain:2", 5, capture_ct=False)
    print("        hash == oracle:", all(h == ref.ct_hash.tobytes() for h in chk.ciphertext_hash), flush=True)

for name, spec, units in (("narrow ", "fq_sqrt", ["fp254::exp_chunk"]), ("inverse", "fq12_inverse", ["inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"])):
    plan = gsv.Plan.from_circuit(spec, units, window_div=4)  # as bench.py builds its plan: one image, good for 1, 2 and 4 instances per workgroup
    sess = gsv.Session(eng, plan, B, retain_stream=False)
    D, K, I = tiled(plan.info["n_inputs"])
    best = 1e9
    for _ in range(3):
        sess.set_garble_inputs(D, K, I)
        t0 = time.perf_counter()
        sess.garble_streaming(discard=True)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print("%s %-12s   : %8.1f ms -> %.3e gates/s (%d calls)" % (name, spec, best, B * plan.info["n_gates"] / best * 1e3, plan.info["n_calls"]), flush=True)
    sess.close()
    if not CHECK:
        plan.close()
        continue
    one = gsv.Session(eng, plan, 2, retain_stream=False)
    d, f, t, inp = gsv.labels_from_seed(9, plan.info["n_inputs"])
    one.set_garble_inputs(np.tile(d, (2, 1)), np.tile(np.stack([f, t]), (2, 1, 1)), np.tile(inp, (2, 1, 1)))
    h = one.garble_streaming()
    ref = o.garble(spec, 9, capture_ct=False)
    print("        hash == oracle:", h[0] == ref.ct_hash.tobytes() and h[1] == h[0] and bool((one.read_outputs()[0] == ref.output_label0).all()), flush=True)
    one.close()
    plan.close()
