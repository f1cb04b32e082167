"""The top-level loops of the verifier compared as SEQUENCES (groth16.rs:57-110,250-268, pairing.rs:507-547,945-1007,
final_exponentiation.rs:65-135, g1.rs:309-400): the C++ gadget headers (tools/call_sequence.cpp) and the independent Python restatement
(tests/ref_call_sequence.py) each print one hash per call of a unit gadget — name, arity and the provenance of every input wire, chained
through the glue gates between the units — and the two lists must be equal line by line.  Counts and Execute-mode values (the other
tests) cannot tell two orders of the same calls apart; this can, and it can tell a swapped operand from the right one."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.setrecursionlimit(10000)


@pytest.fixture(scope="module")
def cpp_tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("callseq") / "call_sequence")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "garbled_snark_verifier_amd", "csrc"), os.path.join(ROOT, "tools", "call_sequence.cpp"), "-o", exe])
    return exe


def _cpp_events(exe, units, fixture="groth16_verify_compressed_1pub_golden.json"):
    import json
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", fixture)))["circuit"]
    out = subprocess.run([exe, spec] + list(units), check=True, capture_output=True, text=True).stdout
    return [(ln.rsplit(" ", 1)[0], int(ln.rsplit(" ", 1)[1], 16)) for ln in out.splitlines()]


def _first_difference(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None if len(a) == len(b) else min(len(a), len(b))


def test_verifier_call_sequence_equals_the_independent_restatement(cpp_tool, monkeypatch):
    import ref_call_sequence as S
    cpp = _cpp_events(cpp_tool, S.UNITS)
    py, glue = S.walk()
    assert len(cpp) == 8502 and glue == 964  # 8 501 unit calls + the output bundle; 964 gates outside every unit
    d = _first_difference(cpp, py)
    assert d is None, "first difference at event %d: C++ %s, Python %s" % (d, cpp[d - 2:d + 2], py[d - 2:d + 2])
    # the check has teeth: one swapped pair of operands in the restatement (the two line-function points of the last Miller-loop
    # additions, pairing.rs:540-545) moves every hash from that call on — and nothing before it
    V = S.V
    orig = V.mul_by_char

    def swapped(c, r):
        out = orig(c, r)
        return [out[1], out[0], out[2]]
    monkeypatch.setattr(V, "mul_by_char", swapped)
    bad, _ = S.walk()
    d = _first_difference(cpp, bad)
    assert d is not None and 0 < d < len(cpp) - 1 and cpp[:d] == bad[:d] and bad[-1] != cpp[-1]


@pytest.mark.slow
def test_uncompressed_verifier_call_sequence(cpp_tool):
    """`groth16_verify` (groth16.rs:57-110; two public inputs, the proof's points as projective wire points, no decompression): 4 949
    unit calls, the same in both walks."""
    import ref_call_sequence as S
    cpp = _cpp_events(cpp_tool, S.UNITS, "groth16_verify_golden.json")
    py, glue = S.walk(n_pub=2, compressed=False)
    assert len(cpp) == 4950 and glue == 703
    assert _first_difference(cpp, py) is None


@pytest.mark.slow
def test_verifier_call_sequence_inside_the_inversions(cpp_tool):
    """The same with the six Fq inversions opened down to their bigint components: 212 717 events."""
    import ref_call_sequence as S
    units = {k: v for k, v in S.UNITS.items() if k != "fp254::inverse"}
    cpp = _cpp_events(cpp_tool, units)
    py, glue = S.walk(units)
    assert len(cpp) == 212718 and glue == 19240
    assert _first_difference(cpp, py) is None
