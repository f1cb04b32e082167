#!/usr/bin/env python3
"""Where does a replay's time go?  Instance 0's workgroup stamps the 100 MHz wall clock at every step of the last
replay; the steps are then bucketed by shape (narrow / wide by AND passes) and by whether they touch HBM labels.
Diagnostic tool, not a benchmark.   usage: step_profile.py [circuit] [instances ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import garbled_snark_verifier_amd as gsv

spec = sys.argv[1] if len(sys.argv) > 1 else "fq12_mul"
batches = [int(x) for x in sys.argv[2:]] or [1, 256]
eng = gsv.Engine(0)
prog = gsv.Program.from_circuit(spec, chain_feedback=True)
st = prog.step_stats().astype(np.int64)
n_in = prog.info["n_inputs"]
and_cnt, xor_cnt, rl, rh, wl, wh = (st[:, i] for i in range(6))
names = ["narrow (multi-lane AES)", "wide <=1024 gates", "wide <=4096 gates", "wide >4096 gates"]
for B in batches:
    d, f, t, inp = gsv.labels_from_seed(1, n_in)
    sess = gsv.Session(eng, prog, B, 3, 1)
    BT = 1024 // sess.instances_per_workgroup  # threads per instance (the HBM columns are the one-instance-per-workgroup image's)
    narrow = (and_cnt > 0) & (and_cnt * 8 + xor_cnt <= BT)
    cls = np.where(narrow, 0, np.where(and_cnt + xor_cnt <= 1024, 1, np.where(and_cnt + xor_cnt <= 4096, 2, 3)))
    sess.enable_step_clock()
    sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
    for _ in range(2):
        sess.garble(0)
        sess.sync()
    ms = sess.last_kernel_ms() / 3
    clk = sess.read_step_clock().astype(np.int64)
    us = np.diff(clk) / 100.0
    print("== %s B=%d (%d per workgroup): %.2f ms/replay (events), %.2f ms (step clocks), %d steps" % (spec, B, sess.instances_per_workgroup, ms, us.sum() / 1e3, len(us)))
    print("%-26s %7s %10s %9s %9s %10s %10s" % ("class", "steps", "gates", "time ms", "us/step", "hbm rd/st", "hbm wr/st"))
    for k, nm in enumerate(names):
        for hb, tag in ((0, "no HBM labels"), (1, "HBM labels")):
            m = (cls == k) & (((rh + wh) > 0) == bool(hb))
            if not m.any():
                continue
            print("%-26s %7d %10d %9.2f %9.2f %10.1f %10.1f   %s" % (nm, m.sum(), (and_cnt + xor_cnt)[m].sum(), us[m].sum() / 1e3, us[m].mean(),
                                                                rh[m].mean(), wh[m].mean(), tag))
    # cost model hints: per-step time vs passes for the widest class
    m = cls == 3
    if m.any():
        passes_and = -(-and_cnt[m] // 1024)
        passes_xor = -(-xor_cnt[m] // 1024)
        A = np.stack([np.ones(m.sum()), passes_and, passes_xor, rh[m] / 1024.0, wh[m] / 1024.0], 1)
        coef, *_ = np.linalg.lstsq(A, us[m], rcond=None)
        print("fit wide>4096: us = %.2f + %.2f*and_passes + %.2f*xor_passes + %.2f*hbm_reads/1024 + %.2f*hbm_writes/1024" % tuple(coef))
    m = cls == 0
    A = np.stack([np.ones(m.sum()), and_cnt[m], xor_cnt[m], rh[m], wh[m]], 1)
    coef, *_ = np.linalg.lstsq(A, us[m], rcond=None)
    print("fit narrow: us = %.3f + %.4f*and + %.4f*xor + %.4f*hbm_reads + %.4f*hbm_writes" % tuple(coef))
    for lo, hi in ((1, 8), (8, 32), (32, 64), (64, 96), (96, 129)):
        mm = m & (and_cnt >= lo) & (and_cnt < hi)
        if mm.any():
            print("  narrow and_cnt in [%d,%d): %d steps, %.2f us/step (no-HBM steps: %s)" % (
                lo, hi, mm.sum(), us[mm].mean(), ("%.2f" % us[mm & (rh + wh == 0)].mean()) if (mm & (rh + wh == 0)).any() else "-"))
    out = os.environ.get("GSV_STEP_DUMP")
    if out:
        np.savez_compressed("%s_B%d.npz" % (out, B), us=us, stats=st)
    sess.close()
