#!/usr/bin/env python3
"""Canonical digest of a plan file (.gsvplan, engine.cpp "plan files"): equal digests = the same plan, byte for byte, whatever order
the compile workers appended the program blocks in (gsv_plan_build_file and a plan recorder with a plan file write a block the moment
its program exists; the offset table at the end is in the order of the programs' first calls).

  sha256( header fields | per program, in table order: sha256(block bytes) | calls + outputs )

usage: plan_digest.py <file> [<file> ...]      prints one line per file: digest, programs, calls, gates, bytes"""
import hashlib
import mmap
import struct
import sys
from concurrent.futures import ThreadPoolExecutor

HDR = struct.Struct("<8s6I3Q2Q")          # magic, n_programs n_calls n_globals n_inputs n_outputs lds_window_slots, n_gates n_ct rec_sizes, calls_off table_off
PROG = struct.Struct("<6Q8Q11Q10I")       # PlanFileProgram: n_steps n_ands n_xors n_ct_pos n_inputs n_outputs | 8 counts | gate_count[11] | 10 u32


def pad16(n):
    return (n + 15) & ~15


def block_len(mm, off, rec_sizes):
    f = PROG.unpack_from(mm, off)
    n_steps, n_ands, n_xors, n_ct_pos, n_in, n_out = f[:6]
    sd, ar, xr = rec_sizes & 0xFFFF, (rec_sizes >> 16) & 0xFFFF, (rec_sizes >> 32) & 0xFFFF
    return sum(pad16(x) for x in (PROG.size, n_steps * sd, n_ands * ar, n_xors * xr, n_ct_pos * 4, n_in * 4, n_out * 4))


def digest(path, threads=8):
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    h = HDR.unpack_from(mm, 0)
    magic, n_prog, n_calls = h[0], h[1], h[2]
    n_gates, n_ct, rec_sizes, calls_off, table_off = h[7], h[8], h[9], h[10], h[11]
    if not magic.startswith(b"GSVPLAN"):
        raise ValueError("%s: not a plan file" % path)
    table = struct.unpack_from("<%dQ" % n_prog, mm, table_off)

    def one(off):
        n = block_len(mm, off, rec_sizes)
        return hashlib.sha256(mm[off:off + n]).digest(), n

    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(one, table))
    top = hashlib.sha256()
    top.update(magic)
    top.update(struct.pack("<6I3Q", *h[1:10]))
    for d, _ in parts:
        top.update(d)
    top.update(mm[calls_off:table_off])
    size = len(mm)
    # every byte of the file belongs to the header, a program block, the calls or the table
    accounted = pad16(HDR.size) + sum(n for _, n in parts) + (table_off - calls_off) + pad16(8 * n_prog)
    mm.close()
    return {"digest": top.hexdigest(), "programs": n_prog, "calls": n_calls, "gates": n_gates, "ciphertexts": n_ct, "bytes": size, "unreferenced_bytes": size - accounted}


if __name__ == "__main__":
    for p in sys.argv[1:]:
        d = digest(p)
        print("%s  %s  programs %d  calls %d  gates %d  bytes %d  unreferenced %d" % (d["digest"], p, d["programs"], d["calls"], d["gates"], d["bytes"], d["unreferenced_bytes"]))
