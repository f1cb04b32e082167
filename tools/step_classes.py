#!/usr/bin/env python3
"""Step classes of a plan at NI instances per workgroup (no GPU): how many device steps (weighted by the calls of their program), AND gates and whole
one-gate-per-lane passes fall into each class of the kernel's pass rule (kernels.hip, run_step) — narrow (one eight-lane pass), remainder 0,
remainder <= BT/8, <= BT/4, <= BT/2, larger — split by the record form of the program (two-wire = throughput-bound, four-wire = latency-bound).
What DESIGN.md §8 items 3 and 4 are sized from.   usage: step_classes.py <plan file (.gsvplan)> [NI = 4]"""
import mmap
import struct
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import lane_util as lu


def main():
    path = sys.argv[1]
    ni = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    bt = 1024 // ni
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    h = lu.HDR.unpack_from(mm, 0)
    n_prog, n_calls, calls_off, table_off = h[1], h[2], h[10], h[11]
    table = struct.unpack_from("<%dQ" % n_prog, mm, table_off)
    calls = np.zeros(n_prog, np.int64)
    pos = calls_off
    for _ in range(n_calls):
        prog, n_in, n_out, _ = struct.unpack_from("<4I", mm, pos)
        calls[prog] += 1
        pos += 16 + lu.pad16(4 * n_in) + lu.pad16(4 * n_out)
    names = ["no AND gate", "narrow (8 lanes/gate, one pass)", "remainder 0", "remainder <= BT/8", "remainder BT/8..BT/4", "remainder BT/4..BT/2", "remainder > BT/2"]
    res = {}
    for k in range(n_prog):
        if calls[k] == 0:
            continue
        fl = lu.PROG.unpack_from(mm, table[k])
        n_steps, terms = fl[0], int(fl[-1])
        a = np.frombuffer(mm, dtype=np.uint32, count=n_steps * 4, offset=table[k] + lu.pad16(lu.PROG.size)).reshape(n_steps, 4)
        ac, xc = a[:, 1].astype(np.int64), a[:, 3].astype(np.int64)
        narrow = (ac > 0) & ((((ac * 8 + 63) // 64) * 64 + xc) <= bt)
        rem, full = ac % bt, ac // bt
        cls = np.where(ac == 0, 0, np.where(narrow, 1, np.where(rem == 0, 2, np.where(rem <= bt // 8, 3, np.where(rem <= bt // 4, 4, np.where(rem <= bt // 2, 5, 6))))))
        for c in range(7):
            m = cls == c
            r = res.setdefault((terms, c), [0, 0, 0, 0])
            r[0] += int(m.sum()) * calls[k]; r[1] += int(ac[m].sum()) * calls[k]; r[2] += int(full[m].sum()) * calls[k]; r[3] += int(xc[m].sum()) * calls[k]
    ts, ta = sum(v[0] for v in res.values()), sum(v[1] for v in res.values())
    print("%s: %d programs, %d calls; NI = %d (BT = %d lanes per instance); %d device steps, %d AND records per instance" % (path, n_prog, n_calls, ni, bt, ts, ta))
    print("%-9s %-34s %10s %7s %12s %7s %12s %12s" % ("records", "class", "steps", "share", "AND gates", "share", "whole passes", "free gates"))
    for (terms, c) in sorted(res):
        v = res[(terms, c)]
        if v[0]:
            print("%-9s %-34s %10d %7.3f %12d %7.3f %12d %12d" % ("%d-wire" % terms, names[c], v[0], v[0] / ts, v[1], v[1] / ta, v[2], v[3]))


if __name__ == "__main__":
    main()
