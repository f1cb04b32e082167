#!/usr/bin/env python3
"""Pipe utilisation of run_program_kernel from the rocprofv3 --pmc passes of tools/profile_r05_pipe.sh.

usage: pipe_util.py <dir with one sub-directory per pass>   -> <dir>/pipe_util.json (+ a summary on stdout)

Every pass wrote one counter_collection.csv (one row per dispatch and counter).  The rows of run_program_kernel are summed per kernel
instantiation: `<false, 4, 0, false>` dispatches are the windows whose programs are all in the two-wire record form — the WIDE windows
(Miller loop, final exponentiation: Fq12-level units) — `<false, 4, 0, true>` the windows that hold a four-wire (latency-bound) program.
Units (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_BUSY_CU_CYCLES quad-cycles summed over CUs; SQ_INSTS_* wave-instructions.  What the ratios mean:
  lds_pipe_busy   = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES'          LDS-array cycles per CU-busy cycle (the T-table lookups' pipe)
  lds_issue_busy  = SQ_ACTIVE_INST_LDS / (SQ_BUSY_CU_CYCLES / ...)  share of the CU's time some wave has an LDS instruction in issue
  valu_busy       = SQ_ACTIVE_INST_VALU x 4 SIMDs ...               likewise for the VALU (one issue port per SIMD)
The raw sums are kept so that any other ratio can be formed later."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    n_disp = collections.defaultdict(lambda: collections.defaultdict(int))
    per_dispatch = collections.defaultdict(dict)  # (pass, dispatch id) -> {counter: value}
    kernels_of = {}
    for d in sorted(os.listdir(out)):
        for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "run_program_kernel" not in k:
                    continue
                short = k[k.index("run_program_kernel"):].split("(")[0]
                c, v = row["Counter_Name"], float(row["Counter_Value"])
                per_kernel[short][c] += v
                n_disp[short][c] += 1
                did = (d, int(row.get("Dispatch_Id", 0)))
                per_dispatch[did][c] = per_dispatch[did].get(c, 0.0) + v
                kernels_of[did] = short
    res = {"kernels": {}, "note": __doc__.split("\n\n")[2]}
    for k, cs in per_kernel.items():
        e = {"counters": dict(cs), "rows": {c: n_disp[k][c] for c in cs}}
        g = cs.get
        r = {}
        busy_cu = g("SQ_BUSY_CU_CYCLES")
        if busy_cu and g("SQ_LDS_IDX_ACTIVE"):
            r["lds_array_active_per_cu_busy_cycle"] = g("SQ_LDS_IDX_ACTIVE") / busy_cu
        if busy_cu and g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            r["lds_bank_conflict_share_of_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        wc = g("SQ_WAVE_CYCLES")
        for name in ("SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if wc and g(name) is not None and name in cs:
                r[name.lower() + "_per_wave_cycle"] = g(name) / wc
        if g("SQ_INSTS_LDS") and g("SQ_INSTS_VALU"):
            r["valu_per_lds_instruction"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_LDS")
        e["ratios"] = r
        res["kernels"][k] = e
    # per-dispatch rows (672 windows x a few counters): small enough to keep, lets a later reader bucket the windows by width
    rows = []
    for (p, did), cs in sorted(per_dispatch.items()):
        rows.append({"pass": p, "dispatch": did, "kernel": kernels_of[(p, did)], **cs})
    res["dispatches"] = rows
    json.dump(res, open(os.path.join(out, "pipe_util.json"), "w"), indent=1)
    for k, e in res["kernels"].items():
        print(k)
        for c, v in sorted(e["counters"].items()):
            print("   %-28s %.6g  (%d rows)" % (c, v, e["rows"][c]))
        for c, v in sorted(e["ratios"].items()):
            print("   -> %-44s %.4f" % (c, v))


if __name__ == "__main__":
    main()
