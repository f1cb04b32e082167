#!/usr/bin/env python3
"""Where do the SGPR spills of run_program_kernel live?  Compiles kernels.hip with -save-temps, finds in the compiler's assembly of every
instantiation the loop that contains the two step barriers (the unrolled step loop: everything a replay executes per step) and counts the
spill traffic inside it (v_readlane / v_writelane = SGPR spills to VGPR lanes, scratch_* = spills to memory) against the whole kernel.
usage: spill_audit.py  (needs hipcc; no GPU)"""
import glob
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENG = os.path.join(ROOT, "garbled_snark_verifier_amd", "csrc", "engine")


def audit():
    """-> [{"eval", "ni", "hash", "fw", "vgprs", "sgpr_spills", "scratch_bytes", "loop": {...}, "kernel": {...}}] with instruction counts
    {"ins", "readlane", "writelane", "scratch"} for the step loop (the smallest loop holding both step barriers) and the whole kernel."""
    d = tempfile.mkdtemp(prefix="gsv_spill_audit_")
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-promote-alloca-to-lds", "-c",
                               os.path.join(ENG, "kernels.hip"), "-o", os.path.join(d, "k.o"), "-save-temps=obj"], stderr=subprocess.DEVNULL)
        src = open(glob.glob(os.path.join(d, "*gfx950*.s"))[0]).read().split("\n")
    finally:
        shutil.rmtree(d, ignore_errors=True)
    notes = {}
    for i, l in enumerate(src):
        m = re.match(r"\s+\.name:\s+(_ZN3gsv3dev18run_program_kernel\S+)", l)
        if m:
            blk = "\n".join(src[i:i + 40])
            notes[m.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")
                                if re.search(r"\.%s:\s+(\d+)" % k, blk)}
    rows = []
    for i, l in enumerate(src):
        m = re.match(r"^(_ZN3gsv3dev18run_program_kernelILb([01])ELi(\d)ELi(\d)ELb([01])EEEvNS0_10KernelArgsE):", l)
        if not m:
            continue
        labpos, ins = {}, []
        for t in src[i + 1:]:
            t = t.strip()
            if t.startswith(".Lfunc_end"):
                break
            t = t.split(";")[0].strip()
            if not t:
                continue
            lm = re.match(r"^(\.LBB\d+_\d+):", t)
            if lm:
                labpos[lm.group(1)] = len(ins)
            elif not t.startswith(".") and not t.endswith(":"):
                ins.append(t)
        loops = []
        for k, t in enumerate(ins):
            bm = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
            if bm and bm.group(1) in labpos and labpos[bm.group(1)] <= k:
                loops.append((labpos[bm.group(1)], k))
        bars = [k for k, t in enumerate(ins) if t.startswith(("s_barrier", "ds_add_u32"))]  # (ds_add_u32: the arrive of the per-group step barrier, FW instantiations at NI > 1)
        cand = sorted([(a, b) for a, b in loops if sum(1 for x in bars if a <= x <= b) >= 2], key=lambda ab: ab[1] - ab[0])  # the smallest such loop: the step loop (the replay loop around it is larger)
        a, b = cand[0] if cand else (0, -1)
        cnt = lambda seg: {"ins": len(seg), "readlane": sum(1 for t in seg if t.startswith("v_readlane")), "writelane": sum(1 for t in seg if t.startswith("v_writelane")),
                           "scratch": sum(1 for t in seg if t.startswith("scratch_"))}
        n = notes.get(m.group(1), {})
        rows.append({"eval": m.group(2) == "1", "ni": int(m.group(3)), "hash": int(m.group(4)), "fw": m.group(5) == "1", "vgprs": n.get("vgpr_count"), "sgpr_spills": n.get("sgpr_spill_count"),
                     "vgpr_spills": n.get("vgpr_spill_count"), "scratch_bytes": n.get("private_segment_fixed_size"), "loop": cnt(ins[a:b + 1]), "kernel": cnt(ins)})
    return rows


if __name__ == "__main__":
    print("%-44s %6s %6s %7s | %-28s | %s" % ("instantiation <EVAL, NI, HASH, FW>", "VGPRs", "spills", "scratch", "step loop (both barriers)", "whole kernel"))
    for r in audit():
        print("%-44s %6s %6s %7s | %5d ins: readlane %d writelane %d scratch %d | %d ins: readlane %d writelane %d scratch %d" % (
            "<%s, %d, %d, %s>" % ("true" if r["eval"] else "false", r["ni"], r["hash"], "true" if r["fw"] else "false"), r["vgprs"], r["sgpr_spills"], r["scratch_bytes"],
            r["loop"]["ins"], r["loop"]["readlane"], r["loop"]["writelane"], r["loop"]["scratch"], r["kernel"]["ins"], r["kernel"]["readlane"], r["kernel"]["writelane"], r["kernel"]["scratch"]))
