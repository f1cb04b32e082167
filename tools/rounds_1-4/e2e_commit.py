#!/usr/bin/env python3
"""PCIe-inclusive figure for DESIGN.md: garble B instances, then drain every instance's ciphertext stream to the host
and compute its CBC-MAC commitment (AESAccumulatingHash) on one host thread per instance, in parallel.
Also times evaluate on the same streams.  Diagnostic tool (bench.py's `value` is the HBM-resident rate)."""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import garbled_snark_verifier_amd as gsv

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=32)
ap.add_argument("--replays", type=int, default=8)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()

eng = gsv.Engine(0)
prog = gsv.Program.from_circuit("fq12_mul", chain_feedback=True)
n_in, B, R = prog.info["n_inputs"], a.instances, a.replays
seeds = list(range(500, 500 + B))
delta = np.zeros((B, 16), np.uint8); consts = np.zeros((B, 2, 16), np.uint8); inputs = np.zeros((B, n_in, 16), np.uint8)
for i, s in enumerate(seeds):
    delta[i], consts[i, 0], consts[i, 1], inputs[i] = gsv.labels_from_seed(s, n_in)
sess = gsv.Session(eng, prog, B, R, R)  # keep the whole stream
gates = prog.info["n_gates"] * R * B
for it in range(2):
    t0 = time.perf_counter()
    sess.set_garble_inputs(delta, consts, inputs)
    sess.garble(0)
    sess.sync()
    t1 = time.perf_counter()
    with ThreadPoolExecutor(a.threads) as ex:
        hashes = list(ex.map(sess.ciphertext_hash, range(B)))
    t2 = time.perf_counter()
print("garble: %.3f s (%.3e gates/s, kernel %.1f ms); D2H + host CBC-MAC on %d threads: %.3f s; end-to-end %.3e gates/s; commitment stage alone %.3e gates/s" % (
    t1 - t0, gates / (t1 - t0), sess.last_kernel_ms(), a.threads, t2 - t1, gates / (t2 - t0), gates / (t2 - t1)))
out0 = sess.read_outputs()
# evaluate the same streams in place
rng = np.random.default_rng(0)
bits = rng.integers(0, 2, size=(B, n_in)).astype(np.uint8)
active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)
sess.set_evaluate_inputs(ca, active, bits)
t0 = time.perf_counter(); sess.evaluate(0); sess.sync(); t1 = time.perf_counter()
oa, ob = sess.read_outputs(with_bits=True)
ok = bool((oa == np.where(ob[:, :, None] == 1, out0 ^ delta[:, None, :], out0)).all())
print("evaluate: %.3f s (%.3e gates/s, kernel %.1f ms); select(value)==active for all %d instances: %s" % (t1 - t0, gates / (t1 - t0), sess.last_kernel_ms(), B, ok))
if a.check:
    import oracle_lib as o
    ref = o.garble("fq12_mul_chain:%d" % R, seeds[0], capture_ct=False)
    print("hash match vs oracle:", ref.ct_hash.tobytes() == hashes[0])
