#!/usr/bin/env python3
"""profiles/<dir>/pmc_counters.json (tools/rounds_1-4/profile_r02.sh: FETCH_SIZE and WRITE_SIZE of the run_program_kernel dispatches, separate
rocprofv3 --pmc passes) -> profiles/<dir>/traffic.json with HBM bytes per launch as MI355X_MICROARCH.md prescribes: both
counters are in KiB; on gfx950 FETCH_SIZE reads half of the bytes of wide coalesced 16-byte-per-lane reads, so it is doubled
before it is compared with a byte count (the kernel's reads are record streams and 16-byte label reads: the doubled figure is an
upper bound, the raw one a lower bound; both are kept).  usage: traffic_from_pmc.py profiles/r03_final [algorithmic bytes per launch [instances per GPU [gates per instance of the circuit]]]
(bench.py quotes the figure only for a run of the same batch AND the same circuit; a launch = one dispatch of run_program_kernel = one
window of the plan session's schedule since round 3)"""
import json
import os
import sys

d = sys.argv[1]
p = json.load(open(os.path.join(d, "pmc_counters.json")))
f, w = p["fetch"]["sum"]["FETCH_SIZE"] * 1024.0, p["write"]["sum"]["WRITE_SIZE"] * 1024.0
nf, nw = p["fetch"]["dispatches"]["FETCH_SIZE"], p["write"]["dispatches"]["WRITE_SIZE"]
out = {"kernel": "run_program_kernel (garble)", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_raw_per_launch": f / nf, "write_bytes_per_launch": w / nw,
       "hbm_bytes_per_launch_raw": f / nf + w / nw, "hbm_bytes_per_launch": 2 * f / nf + w / nw,
       "note": "FETCH_SIZE / WRITE_SIZE (KiB) summed over the run_program_kernel dispatches of separate rocprofv3 --pmc passes of `bench.py`, divided by the "
               "dispatch count; hbm_bytes_per_launch applies the guide's gfx950 correction (FETCH_SIZE x 2)"}
if len(sys.argv) > 3:
    out["instances_per_gpu"] = int(sys.argv[3])  # bench.py quotes the figure only for a run of the same configuration
if len(sys.argv) > 4:
    out["circuit_gates"] = int(sys.argv[4])
if len(sys.argv) > 2:
    out["algorithmic_bytes_per_launch"] = float(sys.argv[2])
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / float(sys.argv[2])
# the engine build these counters belong to: bench.py refuses a traffic.json whose hash differs from the library it is running
try:
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from garbled_snark_verifier_amd import build as _b
    with open(_b.OUT, "rb") as fh:
        out["engine_library_sha256"] = hashlib.sha256(fh.read()).hexdigest()
    out["engine_source_sha256"] = _b.source_sha256()  # reproducible across machines and directories, unlike the library's own hash
except Exception as e:  # noqa: BLE001
    out["engine_library_sha256"] = None
    out["engine_library_sha256_error"] = repr(e)
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
