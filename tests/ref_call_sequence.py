"""The Python half of the call-SEQUENCE check (tests/test_call_sequence.py; the C++ half is tools/call_sequence.cpp, which states the
scheme): the independent restatement of `groth16_verify_compressed` (tests/ref_verifier_count.py over tests/ref_gadgets.py, written from
the Rust source, no C++ involved) walked with a provenance hash on every wire.  A call of a UNIT gadget is an event — hash(name, arity,
provenance of every input wire in order) — whose outputs carry hash(event, j); gates outside the units hash (operands in order, type)
into their output; constants share one value, primary input i has hash(IN, i).  The event list of this walk must equal, line by line,
the one the C++ gadget headers print: same unit calls, same order, same wiring, same glue — groth16.rs:57-110,250-268,
pairing.rs:945-1007, final_exponentiation.rs:99-135 compared as sequences.

UNITS maps the C++ component name to the function of ref_gadgets that restates the same reference function.  A unit's body is not run
in either walk (its gates are pinned one by one by tests/test_ref_gadgets.py); what is NOT a unit is walked down to its gates."""
import sys

import ref_verifier_count as V

R = V.R
M64 = (1 << 64) - 1
K_CONST, K_IN, K_GATE, K_OUT = 0x1111111111111111, 0x2222222222222222, 0x100, 0x3000

UNITS = {
    "fq12::mul_montgomery": "fq12_mul", "fq12::square_montgomery": "fq12_square", "fq12::cyclotomic_square_montgomery": "fq12_cyclotomic_square",
    "fq12::mul_by_034_montgomery": "fq12_mul_by_034", "pairing::ell_by_constant_montgomery": "ell_by_constant",
    "pairing::double_in_place_circuit_montgomery": "g2_double_in_place", "pairing::add_in_place_montgomery": "g2_add_in_place", "g1::add_montgomery": "g1_add",
    "fq6::mul_montgomery": "fq6_mul", "fq2::mul_montgomery": "fq2_mul", "fq2::square_montgomery": "fq2_square", "bigint::multiplexer": "bigint_multiplexer",
    "bigint::mul_karatsuba": "mul_karatsuba", "fp254::montgomery_reduce": "montgomery_reduce", "bigint::add": "add", "bigint::sub": "sub",
    "bigint::add_constant": "add_constant", "bigint::select": "select", "fp254::neg": "fq_neg", "bigint::less_than_constant": "less_than_constant",
    "bigint::greater_than": "greater_than", "bigint::self_or_zero": "self_or_zero", "bigint::self_or_zero_inv": "self_or_zero_inv",
    "bigint::equal_constant": "equal_constant", "bigint::equal_zero": "equal_zero", "fp254::div6": "fq_div6", "bigint::mul_naive": "mul_naive",
    "bigint::mul_by_constant": "mul_by_constant", "bigint::mul_by_constant_modulo_power_two": "mul_by_constant_modulo_power_two",
    "bigint::double_without_overflow": "double_without_overflow", "fp254::inverse": "fq_inverse",
}


def mix(x, y):
    z = (x * 0x9E3779B97F4A7C15 + y) & M64
    z ^= z >> 32
    z = (z * 0xD6E8FEB86659FD93) & M64
    return z ^ (z >> 32)


def name_hash(s):  # FNV-1a 64
    h = 0xcbf29ce484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001b3) & M64
    return h


class ProvCtx(R.Ctx):
    def __init__(self, n_inputs):
        self.prov = [K_CONST, K_CONST]
        self.n_inputs = n_inputs
        self.events = []
        self.glue_gates = 0
        self.counts = [0] * 11   # (ref_verifier_count's component decorator reads these)
        self.top = {}

    def issue(self):
        i = len(self.prov)
        self.prov.append(mix(K_IN, i - 2) if i - 2 < self.n_inputs else 0)
        return i

    def gate(self, t, a, b, c):
        p = self.prov
        p[c] = mix(mix(p[a], p[b]), K_GATE + t)
        self.glue_gates += 1

    def _call(self, *wire_lists):
        pass

    def fresh(self, shape):
        if shape is None:
            return self.issue()
        return [self.fresh(s) for s in shape]


def _flatten(x, out):
    if isinstance(x, int):
        out.append(x)
    else:
        for y in x:
            _flatten(y, out)


def _sig(x):
    """Structure of a wire argument, cheap to build and hashable: a flat list of n wires is n, a nested one the tuple of its parts."""
    if isinstance(x, int):
        return None
    if all(type(y) is int for y in x):
        return len(x)
    return tuple(_sig(y) for y in x)


def _const_class(x):
    """What the SHAPE of a unit's outputs may depend on in an off-circuit argument: its structure and the bit lengths in it, not the
    values (182 line functions and 1 304 multiplications by constants would otherwise each cost a counting walk of their body).  A unit
    whose output shape did depend on a value would end up with the wrong arity — which is hashed, so the comparison would fail."""
    if isinstance(x, int):
        return x.bit_length()
    return tuple(_const_class(y) for y in x)


_SHAPES = {}  # (unit, structure of the wire arguments, class of the off-circuit arguments) -> shape of the outputs


def _install(units):
    """Wrap the unit functions of ref_gadgets (module globals: calls between gadgets go through them) for walks under a ProvCtx."""
    for cpp, name in units.items():
        inner = getattr(R, name)  # ref_verifier_count's memoising wrapper (shapes of the outputs without running the body twice)
        consts = V._CONST_ARGS.get(name, ())
        nh = name_hash(cpp)

        def wrapper(c, *args, _inner=inner, _consts=consts, _nh=nh, _cpp=cpp):
            if not isinstance(c, ProvCtx):
                return _inner(c, *args)
            key = (_cpp,) + tuple(("k", _const_class(a)) if i in _consts else _sig(a) for i, a in enumerate(args))
            shape = _SHAPES.get(key)
            if shape is None:  # the unit's output shape, from a counting walk of its body (memoised there) — not part of THIS walk
                scratch = V.CountCtx()
                shape = _SHAPES[key] = V._shape(_inner(scratch, *[a if i in _consts else scratch.fresh(V._shape(a)) for i, a in enumerate(args)]))
            ins = []
            for i, a in enumerate(args):
                if i not in _consts:
                    _flatten(a, ins)
            outs = c.fresh(shape)
            flat = []
            _flatten(outs, flat)
            e = mix(_nh, len(flat))
            p = c.prov
            for w in ins:
                e = mix(e, p[w])
            c.events.append((_cpp, e))
            for j, w in enumerate(flat):
                p[w] = mix(e, K_OUT + j)
            return outs
        setattr(R, name, wrapper)


def walk(units=None, n_pub=1, seed=6, compressed=True):
    units = dict(UNITS) if units is None else units
    saved = {name: getattr(R, name) for name in units.values()}
    _install(units)
    try:
        inst = V.G.make_instance(n_pub=n_pub, seed=seed)
        c = ProvCtx(n_inputs=254 * (n_pub + 4) + 3 if compressed else 254 * n_pub + 762 + 1524 + 762)
        out = (V.groth16_verify_compressed if compressed else V.groth16_verify)(c, inst)
    finally:
        for name, fn in saved.items():
            setattr(R, name, fn)
    flat = []
    _flatten(out, flat)
    e = mix(name_hash("<outputs>"), len(flat))
    for w in flat:
        e = mix(e, c.prov[w])
    return c.events + [("<outputs>", e)], c.glue_gates


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    ev, glue = walk()
    for name, e in ev:
        print("%s %016x" % (name, e))
    print("ref_call_sequence: %d events, %d gates outside the units" % (len(ev) - 1, glue), file=sys.stderr)
