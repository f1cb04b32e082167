"""An INDEPENDENT restatement of the reference's primitive gadget layer, in Python, written from the Rust source
(/root/reference/src/gadgets/{basic.rs, bigint/{add,cmp,mul}.rs, bn254/{fp254impl,fq,fq2,fq6}.rs}) and NOT from csrc/gadgets/*.hpp.

Why: the CPU oracle and the product share the C++ gate-stream producers (csrc/gadgets), so a wrong gate order, operand order, gate
type or dead-gate decision there would be common-mode and invisible to every GPU parity test.  This module emits the same gadgets'
gate lists a second time — straight-line Python over integer wire ids, no driver, no credits — and tests/test_ref_gadgets.py
compares them gate by gate (type, operands by DEFINITION, deadness) with the trace the C++ recorder produces for the same circuit.
These primitives (ripple adders / subtracters, constant adders, comparators, selectors, Karatsuba, constant multiplication,
Montgomery reduction, the Fq / Fq2 / Fq6 operations built from them) are > 99 % of the verifier's 11.46 B gates.

Model.  A gate is (type, a, b, c) with wire ids a, b, c; FALSE = 0, TRUE = 1; fresh wires come from `issue()`.  Every function that
is a `#[component]` / `#[bn_component]` in the reference records its input-wire list (`_call`): in the reference's two-pass credit
scheme a wire handed to a component counts as read at the call site (streaming_mode.rs:162, component_meta.rs:284-301), so a gate is
DEAD (output UNREACHABLE, no ciphertext, gate id still consumed) iff its output is read by no gate, appears in no component call's
input list and is not a circuit output (storage.rs:119-133, garble_mode.rs:192-197)."""
AND, NAND, NIMP, IMP, NCIMP, CIMP, NOR, OR, XOR, XNOR, NOT = range(11)  # GateType discriminants, src/core/gate_type.rs:3-15
FALSE, TRUE = 0, 1

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583          # fq.rs:57-58 MODULUS
M_INVERSE = 4759646384140481320982610724935209484903937857060724391493050186936685796471   # fq.rs:59-60 MONTGOMERY_M_INVERSE
N_BITS = 254


def and_variant_type(f):  # src/core/gate.rs:180-196
    return {(0, 0, 0): AND, (0, 0, 1): NAND, (0, 1, 0): NIMP, (0, 1, 1): IMP, (1, 0, 0): NCIMP, (1, 0, 1): CIMP, (1, 1, 0): NOR, (1, 1, 1): OR}[tuple(int(x) for x in f)]


class Ctx:
    def __init__(self, n_inputs):
        self.next = 2
        self.gates = []        # (type, a, b, c)
        self.call_inputs = []  # wires that appeared in a component call's input list
        self.inputs = [self.issue() for _ in range(n_inputs)]

    def issue(self):
        w = self.next
        self.next += 1
        return w

    def gate(self, t, a, b, c):
        self.gates.append((t, a, b, c))

    def _call(self, *wire_lists):
        for ws in wire_lists:
            self.call_inputs.extend([ws] if isinstance(ws, int) else ws)


def bits_with_len(v, n):  # bigint/mod.rs bits_from_biguint_with_len: LSB first
    assert v >> n == 0
    return [(v >> i) & 1 for i in range(n)]


# ------------------------------------------------------------------------------------------------ basic.rs
def half_adder(c, a, b):  # :7-15
    result, carry = c.issue(), c.issue()
    c.gate(XOR, a, b, result)
    c.gate(AND, a, b, carry)
    return result, carry


def full_adder(c, a, b, cin):  # :17-32
    axc, bxc, result, t, carry = (c.issue() for _ in range(5))
    c.gate(XOR, a, cin, axc)
    c.gate(XOR, b, cin, bxc)
    c.gate(XOR, a, bxc, result)
    c.gate(AND, axc, bxc, t)
    c.gate(XOR, cin, t, carry)
    return result, carry


def half_subtracter(c, a, b):  # :34-46
    result, borrow = c.issue(), c.issue()
    c.gate(XOR, a, b, result)
    c.gate(and_variant_type([1, 0, 0]), a, b, borrow)
    return result, borrow


def full_subtracter(c, a, b, cin):  # :48-63
    bxa, bxc, result, t, carry = (c.issue() for _ in range(5))
    c.gate(XOR, a, b, bxa)
    c.gate(XOR, b, cin, bxc)
    c.gate(XOR, bxa, cin, result)
    c.gate(AND, bxa, bxc, t)
    c.gate(XOR, cin, t, carry)
    return result, carry


def selector(c, a, b, s):  # :65-72  (s ? a : b)
    d, f, g = c.issue(), c.issue(), c.issue()
    c.gate(NAND, a, s, d)
    c.gate(and_variant_type([1, 0, 1]), s, b, f)
    c.gate(NAND, d, f, g)
    return g


# ------------------------------------------------------------------------------------------------ bigint/add.rs
def add(c, a, b):  # :8-26 (#[bn_component])
    assert len(a) == len(b)
    c._call(a, b)
    bits = []
    r, carry = half_adder(c, a[0], b[0])
    bits.append(r)
    for i in range(1, len(a)):
        r, carry = full_adder(c, a[i], b[i], carry)
        bits.append(r)
    bits.append(carry)
    return bits


def add_without_carry(c, a, b):  # :28-36
    return add(c, a, b)[:-1]


def add_constant(c, a, b):  # :38-84 (#[bn_component], b off-circuit)
    assert b != 0
    c._call(a)
    bb = bits_with_len(b, len(a))
    first_one = bb.index(1)
    bits, carry = [], None
    for i, a_i in enumerate(a):
        if i < first_one:
            bits.append(a_i)
        elif i == first_one:
            w = c.issue()
            c.gate(XOR, a_i, TRUE, w)  # Gate::not_with_xor, gate.rs:147-154
            bits.append(w)
            carry = a_i
        elif bb[i]:
            w1, w2 = c.issue(), c.issue()
            c.gate(XNOR, a_i, carry, w1)
            c.gate(OR, a_i, carry, w2)
            bits.append(w1)
            carry = w2
        else:
            w1, w2 = c.issue(), c.issue()
            c.gate(XOR, a_i, carry, w1)
            c.gate(AND, a_i, carry, w2)
            bits.append(w1)
            carry = w2
    bits.append(carry)
    return bits


def add_constant_without_carry(c, a, b):  # :86-94
    return add_constant(c, a, b)[:-1]


def sub(c, a, b):  # :96-115 (#[bn_component])
    assert len(a) == len(b)
    c._call(a, b)
    bits = []
    r, borrow = half_subtracter(c, a[0], b[0])
    bits.append(r)
    for i in range(1, len(a)):
        r, borrow = full_subtracter(c, a[i], b[i], borrow)
        bits.append(r)
    bits.append(borrow)
    return bits


def sub_without_borrow(c, a, b):  # :117-125 (#[bn_component])
    c._call(a, b)
    return sub(c, a, b)[:-1]


def half(a):  # :146-156 (no gates)
    return a[1:] + [FALSE]


# ------------------------------------------------------------------------------------------------ bigint/cmp.rs
def self_or_zero(c, a, s):  # :9-22 (#[bn_component])
    c._call(a, s)
    out = []
    for a_i in a:
        w = c.issue()
        c.gate(AND, a_i, s, w)
        out.append(w)
    return out


def greater_than(c, a, b):  # :109-129 (#[component])
    c._call(a, b)
    not_b = []
    for b_i in b:
        w = c.issue()
        c.gate(XOR, b_i, TRUE, w)
        not_b.append(w)
    return add(c, a, not_b)[-1]


def less_than_constant(c, a, b):  # :131-152 (#[component], b off-circuit)
    c._call(a)
    not_a = []
    for a_i in a:
        w = c.issue()
        c.gate(XOR, a_i, TRUE, w)
        not_a.append(w)
    return add_constant(c, not_a, b)[-1]


def select(c, a, b, s):  # :154-171 (#[bn_component])
    assert len(a) == len(b)
    c._call(a, b, s)
    return [selector(c, a_i, b_i, s) for a_i, b_i in zip(a, b)]


# ------------------------------------------------------------------------------------------------ bigint/mul.rs
def is_use_karatsuba(n):  # :8-13
    return False if n == 21 else n > 19


def mul_naive(c, a, b):  # :19-56 (#[bn_component])
    assert len(a) == len(b)
    c._call(a, b)
    n = len(a)
    result = [FALSE] * (2 * n)
    for i, cur in enumerate(b):
        add0 = result[i:i + n]
        add1 = []
        for a_bit in a:
            w = c.issue()
            c.gate(AND, a_bit, cur, w)
            add1.append(w)
        r = add(c, add0, add1)
        result[i:i + n + 1] = r
    return result


def mul_karatsuba(c, a, b):  # :58-183 (#[bn_component])
    assert len(a) == len(b)
    c._call(a, b)
    n = len(a)
    if n < 5:
        return mul_naive(c, a, b)
    result = [FALSE] * (2 * n)
    n0, n1 = n // 2, (n + 1) // 2
    a0, a1, b0, b1 = a[:n0], a[n0:], b[:n0], b[n0:]
    sq0 = mul_karatsuba(c, a0, b0) if is_use_karatsuba(n0) else mul_naive(c, a0, b0)
    sq1 = mul_karatsuba(c, a1, b1) if is_use_karatsuba(n1) else mul_naive(c, a1, b1)
    ea0, eb0, esq0 = list(a0), list(b0), list(sq0)
    if n0 < n1:
        ea0.append(FALSE); eb0.append(FALSE); esq0.append(FALSE); esq0.append(FALSE)
    sum_a = add(c, ea0, a1)
    sum_b = add(c, eb0, b1)
    sq_sum = add(c, esq0, sq1)
    sq_sum.append(FALSE)
    sum_mul = mul_karatsuba(c, sum_a, sum_b) if is_use_karatsuba(len(sum_a)) else mul_naive(c, sum_a, sum_b)
    cross = sub_without_borrow(c, sum_mul, sq_sum)[:n + 1]
    result[:2 * n0] = sq0
    seg = result[n0:n0 + n + 1]
    new_seg = add(c, seg, cross)
    result[n0:n0 + n + 2] = new_seg
    seg2 = result[2 * n0:]
    new_seg2 = add(c, seg2, sq1)
    result[2 * n0:] = new_seg2[:2 * n1]
    return result


def mul(c, a, b):  # :185-206
    n = len(a)
    if n < 5:
        return mul_naive(c, a, b)
    return mul_karatsuba(c, a, b) if is_use_karatsuba(n) else mul_naive(c, a, b)


def mul_by_constant(c, a, k):  # :208-239 (#[bn_component], k off-circuit)
    c._call(a)
    n = len(a)
    acc = [FALSE] * (2 * n)
    for i, bit in enumerate(bits_with_len(k, n)):
        if not bit:
            continue
        nb = add(c, a, acc[i:i + n])
        acc[i:i + n + 1] = nb
    return acc


def mul_by_constant_modulo_power_two(c, a, k, power):  # :241-329 (#[bn_component]; 8 one-bits per child component)
    c._call(a)
    n = len(a)
    assert power < 2 * n
    ones = [i for i, bit in enumerate(bits_with_len(k, n)) if bit and i < power]
    res = [FALSE] * power
    if not ones:
        return res
    for ch in range(0, len(ones), 8):
        c._call(a, res)  # with_named_child("mul_by_const_mod_2p", (a.bits, prev))
        res = list(res)
        for i in ones[ch:ch + 8]:
            nb = min(power - i, n)
            if nb == 0:
                continue
            new_bits = add(c, a[:nb], res[i:i + nb])
            if i + nb < power:
                res[i:i + nb + 1] = new_bits
            else:
                res[i:i + nb] = new_bits[:nb]
    return res


# ------------------------------------------------------------------------------------------------ bn254/fp254impl.rs (Fq)
NOT_MODULUS = (1 << N_BITS) - P                      # :58-62
HALF_MODULUS = pow(2, -1, P)                         # fq.rs:64-66
ONE_THIRD = pow(3, -1, P)                            # fq.rs:68-70
TWO_THIRD = 2 * pow(3, -1, P) % P                    # fq.rs:72-74
NEG_ADDEND = (1 - NOT_MODULUS) % P                   # fp254impl.rs:164: Fq(1) - Fq(2^254 - p)


def _reduce_once(c, w1, u):  # tail of add / add_constant / double: fp254impl.rs:104-114
    w2 = add_constant(c, w1, NOT_MODULUS)[:-1]
    v = less_than_constant(c, w1, P)
    s = c.issue()
    c.gate(and_variant_type([1, 0, 0]), u, v, s)
    return select(c, w1, w2, s)


def fq_add(c, a, b):  # :95-114 (#[bn_component])
    c._call(a, b)
    w1 = add(c, a, b)
    u = w1.pop()
    return _reduce_once(c, w1, u)


def fq_add_constant(c, a, k):  # :116-141 (#[bn_component])
    c._call(a)
    if k == 0:
        return list(a)
    w1 = add_constant(c, a, k)
    u = w1.pop()
    return _reduce_once(c, w1, u)


def fq_neg(c, a):  # :152-167 (#[bn_component])
    c._call(a)
    not_a = [c.issue() for _ in a]  # BigIntWires::from_ctx first, gates after
    for w, a_i in zip(not_a, a):
        c.gate(XOR, a_i, TRUE, w)
    return fq_add_constant(c, not_a, NEG_ADDEND)


def fq_sub(c, a, b):  # :142-150 (#[bn_component])
    c._call(a, b)
    return fq_add(c, a, fq_neg(c, b))


def fq_double(c, a):  # :169-190 (#[bn_component])
    c._call(a)
    shifted = list(a)
    u = shifted.pop()
    shifted.insert(0, FALSE)
    return _reduce_once(c, shifted, u)


def fq_half(c, a):  # :192-203 (#[bn_component])
    c._call(a)
    sel = a[0]
    w1 = half(a)
    w2 = add_constant_without_carry(c, w1, HALF_MODULUS)
    return select(c, w2, w1, sel)


def fq_triple(c, a):  # :727-732 (#[bn_component])
    c._call(a)
    a2 = fq_double(c, a)
    return fq_add(c, a2, a)


def fq_div6(c, a):  # :734-792 (#[bn_component])
    c._call(a)
    h = fq_half(c, a)
    result = [c.issue() for _ in a]  # pre-issued, every entry overwritten below
    r1 = r2 = FALSE
    for i in range(N_BITS):
        j = N_BITS - 1 - i
        r2_and_hj = c.issue()
        c.gate(AND, r2, h[j], r2_and_hj)
        result_wire = c.issue()
        c.gate(OR, r1, r2_and_hj, result_wire)
        result[j] = result_wire
        new_r1 = c.issue()
        c.gate(XOR, r2, result_wire, new_r1)
        r1 = new_r1
        new_r2 = c.issue()
        c.gate(XOR, h[j], result_wire, new_r2)
        r2 = new_r2
        edge = c.issue()
        c.gate(NIMP, result_wire, h[j], edge)
        new_r1 = c.issue()
        c.gate(XOR, r1, edge, new_r1)
        r1 = new_r1
    plus_third = add_constant_without_carry(c, result, ONE_THIRD)
    result = select(c, plus_third, result, r2)
    plus_two_third = add_constant_without_carry(c, result, TWO_THIRD)
    return select(c, plus_two_third, result, r1)


def montgomery_reduce(c, x):  # :303-331 (#[bn_component])
    assert len(x) == 2 * N_BITS
    c._call(x)
    x_low, x_high = x[:254], x[254:]
    q = mul_by_constant_modulo_power_two(c, x_low, M_INVERSE, 254)
    sub_ = mul_by_constant(c, q, P)[254:508]
    bound = greater_than(c, sub_, x_high)
    modulus_wires = [TRUE if b else FALSE for b in bits_with_len(P, len(x_high))]
    subtract_if_too_much = self_or_zero(c, modulus_wires, bound)
    new_sub = sub_without_borrow(c, sub_, subtract_if_too_much)
    return sub_without_borrow(c, x_high, new_sub)


def fq_mul(c, a, b):  # :216-229 (not a component)
    return montgomery_reduce(c, mul(c, a, b))


def fq_square(c, a):  # :283-285
    return fq_mul(c, a, a)


# ------------------------------------------------------------------------------------------------ fq2.rs / fq6.rs (none are components)
def fq2_add(c, a, b): return [fq_add(c, a[0], b[0]), fq_add(c, a[1], b[1])]          # fq2.rs:160-168
def fq2_sub(c, a, b): return [fq_sub(c, a[0], b[0]), fq_sub(c, a[1], b[1])]          # :188-199
def fq2_double(c, a): return [fq_double(c, a[0]), fq_double(c, a[1])]                # :201-209
def fq2_div6(c, a): return [fq_div6(c, a[0]), fq_div6(c, a[1])]                      # :386-394


def fq2_triple(c, a):  # :221-228
    a2 = fq2_double(c, a)
    return fq2_add(c, a, a2)


def fq2_mul(c, a, b):  # :230-255
    a_sum = fq_add(c, a[0], a[1])
    b_sum = fq_add(c, b[0], b[1])
    a0b0 = fq_mul(c, a[0], b[0])
    a1b1 = fq_mul(c, a[1], b[1])
    sum_prod = fq_mul(c, a_sum, b_sum)
    c0 = fq_sub(c, a0b0, a1b1)
    s = fq_add(c, a0b0, a1b1)
    c1 = fq_sub(c, sum_prod, s)
    return [c0, c1]


def fq2_mul_by_nonresidue(c, a):  # :324-339
    a0_3 = fq_triple(c, a[0])
    a0_9 = fq_triple(c, a0_3)
    a1_3 = fq_triple(c, a[1])
    a1_9 = fq_triple(c, a1_3)
    c0 = fq_sub(c, a0_9, a[1])
    c1 = fq_add(c, a1_9, a[0])
    return [c0, c1]


def fq6_mul(c, a, b):  # fq6.rs:194-260
    v0 = fq2_mul(c, a[0], b[0])
    w2 = fq2_add(c, a[0], a[2]); w3 = fq2_add(c, w2, a[1]); w4 = fq2_sub(c, w2, a[1])
    w5 = fq2_double(c, a[1]); w6 = fq2_double(c, a[2]); w7 = fq2_double(c, w6)
    w8 = fq2_add(c, a[0], w5); w9 = fq2_add(c, w8, w7)
    w10 = fq2_add(c, b[0], b[2]); w11 = fq2_add(c, w10, b[1]); w12 = fq2_sub(c, w10, b[1])
    w13 = fq2_double(c, b[1]); w14 = fq2_double(c, b[2]); w15 = fq2_double(c, w14)
    w16 = fq2_add(c, b[0], w13); w17 = fq2_add(c, w16, w15)
    v1 = fq2_mul(c, w3, w11); v2 = fq2_mul(c, w4, w12); v3 = fq2_mul(c, w9, w17); v4 = fq2_mul(c, a[2], b[2])
    v2_2 = fq2_double(c, v2)
    v0_3 = fq2_triple(c, v0); v1_3 = fq2_triple(c, v1); v2_3 = fq2_triple(c, v2); v4_3 = fq2_triple(c, v4)
    v0_6 = fq2_double(c, v0_3); v1_6 = fq2_double(c, v1_3); v4_6 = fq2_double(c, v4_3)
    v4_12 = fq2_double(c, v4_6)
    w18 = fq2_sub(c, v0_3, v1_3); w19 = fq2_sub(c, w18, v2); w20 = fq2_add(c, w19, v3); w21 = fq2_sub(c, w20, v4_12)
    w22 = fq2_mul_by_nonresidue(c, w21)
    c0 = fq2_add(c, w22, v0_6)
    w23 = fq2_sub(c, v1_6, v0_3); w24 = fq2_sub(c, w23, v2_2); w25 = fq2_sub(c, w24, v3); w26 = fq2_add(c, w25, v4_12)
    w27 = fq2_mul_by_nonresidue(c, v4_6)
    c1 = fq2_add(c, w26, w27)
    w28 = fq2_sub(c, v1_3, v0_6); w29 = fq2_add(c, w28, v2_3)
    c2 = fq2_sub(c, w29, v4_6)
    return [fq2_div6(c, c0), fq2_div6(c, c1), fq2_div6(c, c2)]


def fq6_add(c, a, b): return [fq2_add(c, a[k], b[k]) for k in range(3)]   # fq6.rs:154-160
def fq6_sub(c, a, b): return [fq2_sub(c, a[k], b[k]) for k in range(3)]   # :170-176
def fq6_double(c, a): return [fq2_double(c, a[k]) for k in range(3)]      # :178-184


def fq6_mul_by_nonresidue(c, a):  # fq6.rs:346-349
    u = fq2_mul_by_nonresidue(c, a[2])
    return [u, a[0], a[1]]


def fq12_mul(c, a, b):  # fq12.rs:198-221 (#[component])
    c._call(_flat12(a), _flat12(b))
    a_sum = fq6_add(c, a[0], a[1])
    b_sum = fq6_add(c, b[0], b[1])
    a0b0 = fq6_mul(c, a[0], b[0])
    a1b1 = fq6_mul(c, a[1], b[1])
    s = fq6_add(c, a0b0, a1b1)
    sum_prod = fq6_mul(c, a_sum, b_sum)
    nonres = fq6_mul_by_nonresidue(c, a1b1)
    c0 = fq6_add(c, a0b0, nonres)
    c1 = fq6_sub(c, sum_prod, s)
    return [c0, c1]


def fq12_square(c, a):  # fq12.rs:311-324 (#[component])
    c._call(_flat12(a))
    w1 = fq6_add(c, a[0], a[1])
    w2 = fq6_mul_by_nonresidue(c, a[1])
    w3 = fq6_add(c, a[0], w2)
    w4 = fq6_mul(c, a[0], a[1])
    w5 = fq6_mul(c, w1, w3)
    w6 = fq6_mul_by_nonresidue(c, w4)
    w7 = fq6_add(c, w4, w6)
    c0 = fq6_sub(c, w5, w7)
    c1 = fq6_double(c, w4)
    return [c0, c1]


def _flat12(x): return [w for f6 in x for f2 in f6 for fq in f2 for w in fq]
def _fq12(i, base): return [[[_fq(i, base + 6 * h + 2 * k), _fq(i, base + 6 * h + 2 * k + 1)] for k in range(3)] for h in range(2)]


# ------------------------------------------------------------------------------------------------ named circuits (as csrc/gadgets/circuits.hpp names them)
def _fq(ws, k): return ws[254 * k:254 * (k + 1)]


CIRCUITS = {
    "u254_add": (508, lambda c, i: add(c, i[:254], i[254:])),
    "bigint_mul:22": (44, lambda c, i: mul(c, i[:22], i[22:])),
    "bigint_mul:40": (80, lambda c, i: mul(c, i[:40], i[40:])),
    "fq_add": (508, lambda c, i: fq_add(c, _fq(i, 0), _fq(i, 1))),
    "fq_sub": (508, lambda c, i: fq_sub(c, _fq(i, 0), _fq(i, 1))),
    "fq_neg": (254, lambda c, i: fq_neg(c, i)),
    "fq_double": (254, lambda c, i: fq_double(c, i)),
    "fq_half": (254, lambda c, i: fq_half(c, i)),
    "fq_triple": (254, lambda c, i: fq_triple(c, i)),
    "fq_div6": (254, lambda c, i: fq_div6(c, i)),
    "fq_mul": (508, lambda c, i: fq_mul(c, _fq(i, 0), _fq(i, 1))),
    "fq2_mul": (1016, lambda c, i: sum(fq2_mul(c, [_fq(i, 0), _fq(i, 1)], [_fq(i, 2), _fq(i, 3)]), [])),
    "fq12_mul": (6096, lambda c, i: _flat12(fq12_mul(c, _fq12(i, 0), _fq12(i, 12)))),       # BASELINE config 3 (tests/fq12_mul_e2e.rs)
    "fq12_square": (3048, lambda c, i: _flat12(fq12_square(c, _fq12(i, 0)))),
    "fq6_mul": (3048, lambda c, i: sum(sum(fq6_mul(c, [[_fq(i, 2 * k), _fq(i, 2 * k + 1)] for k in range(3)], [[_fq(i, 6 + 2 * k), _fq(i, 6 + 2 * k + 1)] for k in range(3)]), []), [])),
}


def emit(name):
    """Canonical gate stream of a named circuit: a list of (type, ref_a, ref_b, dead) with ref = ('c', 0|1) for the constants,
    ('i', k) for circuit input k, ('g', j) for the output of gate j; plus the refs of the circuit outputs."""
    n_in, fn = CIRCUITS[name]
    c = Ctx(n_in)
    outs = fn(c, list(c.inputs))
    return canonical(c.gates, c.inputs, outs, extra_reads=c.call_inputs)


def canonical(gates, inputs, outputs, extra_reads=(), dead_marker=None):
    """`gates`: (type, a, b, c) over arbitrary wire ids.  dead_marker None: deadness is DERIVED (output read by no gate, in no component
    call's input list, not a circuit output); else gates whose c == dead_marker are the dead ones (the C++ recorder's trace)."""
    ref = {FALSE: ("c", 0), TRUE: ("c", 1)}
    for k, w in enumerate(inputs):
        ref[w] = ("i", k)
    read = set(extra_reads) | set(outputs)
    if dead_marker is None:
        for t, a, b, cc in gates:
            read.add(a)
            read.add(b)
    out = []
    for j, (t, a, b, cc) in enumerate(gates):
        dead = (cc == dead_marker) if dead_marker is not None else (cc not in read)
        out.append((t, ref[a], ref[b], bool(dead)))
        if not dead:
            ref[cc] = ("g", j)
    return out, [ref[w] for w in outputs]
