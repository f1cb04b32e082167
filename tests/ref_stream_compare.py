"""Gate-by-gate comparison of LARGE gate streams (round 6): the product's recorder trace against the independent Python restatement, in
numpy form — what tests/test_ref_gadgets.py does with lists of tuples (a few M gates) done with flat arrays, so that the verifier's
building blocks ABOVE the Fq6 level fit: Fq2 / Fq12 inversion, the Frobenius maps, `mul_by_char`, projective -> affine, the Fq square
root ladder and `Fq2::sqrt_general` (471 M gates).  Canonical form of a gate, as in ref_gadgets.canonical: its type, both operands
named by their DEFINITION (-1 / -2 the constants, -(3 + k) circuit input k, j >= 0 the output of gate j) and whether it is dead; the
Python side DERIVES deadness (output read by no gate, in no component call's input list, not a circuit output), the product's trace
carries the recorder's decision.

The restated gadgets above the primitives live in tests/ref_verifier_count.py (written from the Rust: final_exponentiation.rs, fq2.rs,
fq6.rs, fq12.rs, pairing.rs, groth16.rs) where they were, until this round, only COUNTED."""
from array import array

import numpy as np

import ref_gadgets as R


class ArrayCtx(R.Ctx):
    """ref_gadgets.Ctx with the gate list in flat typed arrays (13 bytes per gate instead of ~200)."""

    def __init__(self, n_inputs):
        self.next = 2
        self.t, self.a, self.b, self.c = array("B"), array("I"), array("I"), array("I")
        self.calls = array("I")
        self.inputs = [self.issue() for _ in range(n_inputs)]

    def gate(self, t, a, b, c):
        self.t.append(t); self.a.append(a); self.b.append(b); self.c.append(c)

    def _call(self, *wire_lists):
        for ws in wire_lists:
            if isinstance(ws, int):
                self.calls.append(ws)
            else:
                self.calls.extend(ws)

    def arrays(self):
        return (np.frombuffer(self.t, np.uint8), np.frombuffer(self.a, np.uint32), np.frombuffer(self.b, np.uint32), np.frombuffer(self.c, np.uint32),
                np.frombuffer(self.calls, np.uint32) if len(self.calls) else np.zeros(0, np.uint32))


def canonical_np(t, a, b, c, inputs, outputs, dead_marker=None, extra_reads=None):
    """-> (type[n], ref_a[n], ref_b[n], dead[n], output refs).  Wire ids must be SSA (every live gate defines a fresh id)."""
    inputs = np.asarray(inputs, np.int64); outputs = np.asarray(outputs, np.int64)
    n = len(t)
    live_c = c if dead_marker is None else c[c != dead_marker]
    nw = int(max(a.max(initial=1), b.max(initial=1), live_c.max(initial=1), inputs.max(initial=1), outputs.max(initial=1))) + 1
    if dead_marker is None:
        read = np.zeros(nw, bool)
        read[a] = True; read[b] = True; read[outputs] = True
        if extra_reads is not None and len(extra_reads):
            read[extra_reads] = True
        dead = ~read[c]
    else:
        dead = c == dead_marker
    INVALID = np.int64(-(1 << 62))
    ref = np.full(nw, INVALID, np.int64)
    ref[0], ref[1] = -1, -2
    ref[inputs] = -(3 + np.arange(len(inputs), dtype=np.int64))
    live = np.nonzero(~dead)[0]
    defs = c[live].astype(np.int64)
    if len(np.unique(defs)) != len(defs) or (ref[defs] != INVALID).any():
        raise AssertionError("the stream is not in SSA form")
    ref[defs] = live
    ra, rb = ref[a], ref[b]
    if (ra == INVALID).any() or (rb == INVALID).any():
        raise AssertionError("a gate reads a wire nobody defined")
    idx = np.arange(n, dtype=np.int64)
    if (ra >= idx).any() or (rb >= idx).any():
        raise AssertionError("a gate reads a later gate's output")
    return t, ra, rb, dead, ref[outputs]


def first_difference(x, y):
    """Index of the first gate at which two canonical streams differ, or None; lengths may differ."""
    n = min(len(x[0]), len(y[0]))
    bad = np.zeros(n, bool)
    for k in range(4):
        bad |= x[k][:n] != y[k][:n]
    nz = np.nonzero(bad)[0]
    if len(nz):
        return int(nz[0])
    return None if len(x[0]) == len(y[0]) else n


def _fq(ws, k): return ws[254 * k:254 * (k + 1)]
def _fq2(ws, k): return [_fq(ws, 2 * k), _fq(ws, 2 * k + 1)]
def _fq12(ws): return [[_fq2(ws, 3 * h + k) for k in range(3)] for h in range(2)]
def _flat(x): return [x] if isinstance(x, int) else [w for y in x for w in _flat(y)]


# ---- g1.rs:309-368 scalar_mul_by_constant_base_montgomery with its REAL constant tables: the table entries are arkworks' Jacobian coordinates
# after `p += base` / `b + b` (ark-ec 0.5.0, Cargo.lock:145-147, short Weierstrass `Projective`), so the restatement needs arkworks' formulas.
# They are restated here from the published algorithms arkworks implements (EFD "add-2007-bl" and, for a = 0, "dbl-2009-l") with arkworks'
# special cases (zero = (1, 1, 0); adding to zero copies the other operand; adding a point to itself doubles) — not from csrc/gadgets.
def _jac_double(p):
    P_ = R.P
    x, y, z = p
    if z == 0:
        return p
    a = x * x % P_; b = y * y % P_; c = b * b % P_
    d = 2 * ((x + b) * (x + b) - a - c) % P_
    e = 3 * a % P_; f = e * e % P_
    x3 = (f - 2 * d) % P_
    return (x3, (e * (d - x3) - 8 * c) % P_, 2 * y * z % P_)


def _jac_add(p, q):
    P_ = R.P
    if p[2] == 0:
        return q
    if q[2] == 0:
        return p
    x1, y1, z1 = p; x2, y2, z2 = q
    z1z1 = z1 * z1 % P_; z2z2 = z2 * z2 % P_
    u1 = x1 * z2z2 % P_; u2 = x2 * z1z1 % P_
    s1 = y1 * z2 * z2z2 % P_; s2 = y2 * z1 * z1z1 % P_
    if u1 == u2 and s1 == s2:
        return _jac_double(p)
    h = (u2 - u1) % P_; i = 4 * h * h % P_; j = -h * i % P_
    r = 2 * (s2 - s1) % P_; v = u1 * i % P_
    x3 = (r * r + j - 2 * v) % P_
    return (x3, (r * (v - x3) + 2 * s1 * j) % P_, 2 * z1 * z2 * h % P_)


def g1_scalar_mul_const_base(c, s, base_affine, W):  # g1.rs:309-368 (#[component(offcircuit_args = "base")])
    mont = lambda v: v * R.R_MOD_P % R.P  # noqa: E731 - G1Projective::as_montgomery: every coordinate times R
    c._call(s)
    n = 1 << W
    base = (base_affine[0], base_affine[1], 1)
    bases, p = [], (1, 1, 0)  # ark_bn254::G1Projective::default() = zero
    for _ in range(n):
        bases.append(p)
        p = _jac_add(p, base)
    to_add, index = [], 0
    while index < R.N_BITS:
        w = min(W, R.N_BITS - index)
        sel = s[index:index + w]
        table = [[R.const_wires(mont(q[k])) for q in bases[:1 << w]] for k in range(3)]
        to_add.append([R.bigint_multiplexer(c, table[k], sel) for k in range(3)])  # g1::multiplexer: x, y, z (g1.rs:276-306)
        index += W
        nb = []
        for q in bases:
            for _ in range(w):
                q = _jac_add(q, q)
            nb.append(q)
        bases = nb
    acc = to_add[0]
    for a in to_add[1:]:
        acc = R.g1_add(c, acc, a)
    return acc


def _circuits():
    import ref_verifier_count as V  # (wraps ref_gadgets' functions with a memo that is a pass-through for any context but its own CountCtx)
    p2a = getattr(V.projective_to_affine, "__wrapped__", V.projective_to_affine)
    return {
        "fq2_inverse": (508, lambda c, i: V.fq2_inverse(c, _fq2(i, 0))),                                  # fq2.rs:356-372
        "fq12_inverse": (3048, lambda c, i: V.fq12_inverse(c, _fq12(i))),                                 # fq12.rs:413-428 (Fq6 inverse, Fq2 inverse, Fq inverse inside)
        "fq12_frobenius:1": (3048, lambda c, i: V.fq12_frobenius(c, _fq12(i), 1)),                        # fq12.rs:430-442
        "fq12_frobenius:2": (3048, lambda c, i: V.fq12_frobenius(c, _fq12(i), 2)),
        "fq12_frobenius:3": (3048, lambda c, i: V.fq12_frobenius(c, _fq12(i), 3)),
        "fq12_conjugate": (3048, lambda c, i: V.fq12_conjugate(c, _fq12(i))),                             # fq12.rs:444-447
        "g2_mul_by_char": (1524, lambda c, i: V.mul_by_char(c, [_fq2(i, 0), _fq2(i, 1), _fq2(i, 2)])),    # pairing.rs:475-501
        "g1_to_affine": (762, lambda c, i: p2a(c, [_fq(i, 0), _fq(i, 1), _fq(i, 2)])),                    # groth16.rs:26-48
        "g1_scalar_mul:10": (254, lambda c, i: g1_scalar_mul_const_base(c, i, (1, 2), 10)),                # g1.rs:309-368 with the generator as base: the MSM's window tables, 225 M gates
        "fq_sqrt": (254, lambda c, i: V.fq_sqrt(c, i)),                                                   # fq.rs:290-299 -> fp254impl.rs:691-725, 149 M gates
        "fq2_sqrt": (508, lambda c, i: V.fq2_sqrt_general(c, _fq2(i, 0))),                                # fq2.rs:425-446, 471 M gates
    }


def _lookup(name):
    ext = _circuits()
    return ext[name] if name in ext else R.CIRCUITS[name]


def restated_stream(name):
    n_in, fn = _lookup(name)
    c = ArrayCtx(n_in)
    outs = _flat(fn(c, list(c.inputs)))
    t, a, b, cc, calls = c.arrays()
    return canonical_np(t, a, b, cc, c.inputs, outs, extra_reads=calls)


def product_stream(name, cap):
    """The product recorder's trace of the named circuit (tests/hostsim: RecordMode under the two-pass driver), canonical form."""
    import ctypes as C
    import hostsim_lib as h
    n_in = _lookup(name)[0]
    t = np.zeros(cap, np.uint8)
    a, b, c = (np.zeros(cap, np.uint32) for _ in range(3))
    n, nw = C.c_uint64(), C.c_uint32()
    ins, outs = np.zeros(n_in, np.uint32), np.zeros(8192, np.uint32)
    u32p = C.POINTER(C.c_uint32)
    rc = h.lib().hostsim_trace(name.encode(), C.c_uint64(cap), t.ctypes.data_as(C.POINTER(C.c_uint8)), a.ctypes.data_as(u32p), b.ctypes.data_as(u32p), c.ctypes.data_as(u32p), C.byref(n), C.byref(nw),
                               ins.ctypes.data_as(u32p), outs.ctypes.data_as(u32p))
    if rc:
        raise RuntimeError("trace of %s: %s" % (name, "capacity too small" if rc == 2 else h.lib().hostsim_last_error().decode()))
    return t[:n.value], a[:n.value], b[:n.value], c[:n.value], ins, outs
