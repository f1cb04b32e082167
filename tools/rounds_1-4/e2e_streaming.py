#!/usr/bin/env python3
"""PCIe-inclusive rate of the full path: gsv_session_garble_streaming = garble B instances x R Fq12-mul replays with a
2-replay device ring while host threads drain each finished segment (D2H, gate-order permutation, per-instance CBC-MAC).
Diagnostic tool (bench.py's `value` is the HBM-resident rate and never this)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import garbled_snark_verifier_amd as gsv

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=64)
ap.add_argument("--replays", type=int, default=16)
ap.add_argument("--ring", type=int, default=2)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()

eng = gsv.Engine(0)
prog = gsv.Program.from_circuit("fq12_mul", chain_feedback=True)
n_in, B, R = prog.info["n_inputs"], a.instances, a.replays
seeds = list(range(500, 500 + B))
delta = np.zeros((B, 16), np.uint8); consts = np.zeros((B, 2, 16), np.uint8); inputs = np.zeros((B, n_in, 16), np.uint8)
for i, s in enumerate(seeds):
    delta[i], consts[i, 0], consts[i, 1], inputs[i] = gsv.labels_from_seed(s, n_in)
sess = gsv.Session(eng, prog, B, R, a.ring)
gates = prog.info["n_gates"] * R * B
ct_bytes = prog.info["n_ciphertexts"] * 16 * R * B
for it in range(2):
    sess.set_garble_inputs(delta, consts, inputs)
    t0 = time.perf_counter()
    hashes = sess.garble_streaming(threads=a.threads)
    t1 = time.perf_counter()
print("garble + drain + CBC-MAC: %d instances x %d replays, ring %d, %s host threads: %.2f s -> %.3e gates/s end to end, %.1f GB/s of ciphertexts over PCIe" % (
    B, R, a.ring, a.threads or "auto", t1 - t0, gates / (t1 - t0), ct_bytes / (t1 - t0) / 1e9))
if a.check:
    import oracle_lib as o
    ref = o.garble("fq12_mul_chain:%d" % R, seeds[0], capture_ct=False)
    print("hash match vs oracle:", ref.ct_hash.tobytes() == hashes[0])
