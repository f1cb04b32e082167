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


# ------------------------------------------------------------------------------------------------ more of bigint / fp254impl: the Fq inversion
def double_without_overflow(c, a):  # add.rs:134-141 (#[bn_component]: no gates, but the call reads all of `a`)
    c._call(a)
    return [FALSE] + a[:-1]


def self_or_zero_inv(c, a, s):  # cmp.rs:24-40 (#[bn_component])
    c._call(a, s)
    out = []
    for a_i in a:
        w = c.issue()
        c.gate(and_variant_type([0, 1, 0]), a_i, s, w)
        out.append(w)
    return out


def equal_zero(c, a):  # cmp.rs:87-107 (#[component])
    c._call(a)
    if len(a) == 1:
        z = c.issue()
        c.gate(XOR, a[0], TRUE, z)
        return z
    res = c.issue()
    c.gate(XNOR, a[0], a[1], res)
    for a_i in a[1:]:
        nxt = c.issue()
        c.gate(and_variant_type([1, 0, 0]), a_i, res, nxt)
        res = nxt
    return res


def equal_constant(c, a, b):  # cmp.rs:60-85 (#[component], b off-circuit)
    c._call(a)
    if b == 0:
        return equal_zero(c, a)
    bb = bits_with_len(b, len(a))
    one_ind = bb.index(1)
    res = a[one_ind]
    for i, a_i in enumerate(a):
        if i == one_ind:
            continue
        nr = c.issue()
        c.gate(and_variant_type([not bb[i], 0, 0]), a_i, res, nr)
        res = nr
    return res


def odd_part(c, a):  # add.rs:155-195 (not a component)
    n = len(a)
    select_bn = [a[0]] + [c.issue() for _ in range(n - 1)]  # from_ctx(n-1) then insert(0, a[0])
    for i in range(1, n):
        c.gate(OR, select_bn[i - 1], a[i], select_bn[i])
    k = [a[0]] + [c.issue() for _ in range(n - 1)]
    for i in range(1, n):
        c.gate(and_variant_type([1, 0, 0]), select_bn[i - 1], a[i], k[i])
    odd_acc = list(a)
    for i in range(n):
        half_res = half(odd_acc)
        odd_acc = select(c, odd_acc, half_res, select_bn[i])
    return odd_acc, k


def fq_equal_constant(c, a, b):  # fp254impl.rs:87-93 (not a component)
    return equal_constant(c, a, b)


def const_wires(v, n=N_BITS):  # BigIntWires::new_constant
    return [TRUE if b else FALSE for b in bits_with_len(v, n)]


def fq_inverse(c, a):  # fp254impl.rs:333-663 (#[bn_component]); binary extended Euclid, 4 iterations per child component
    c._call(a)
    odd, even_part = odd_part(c, a)
    neg_odd = fq_neg(c, odd)
    u, v = half(neg_odd), odd
    k, r, s = const_wires(1), const_wires(1), const_wires(2)
    for it0 in range(0, 2 * N_BITS, 4):
        c._call(u, v, r, s, k)  # with_named_child("inverse_iteration", IterationContext)
        for _ in range(min(4, 2 * N_BITS - it0)):
            not_x1, not_x2 = u[0], v[0]
            x3 = greater_than(c, u, v)
            p2 = c.issue()
            c.gate(and_variant_type([0, 1, 0]), not_x1, not_x2, p2)
            p3 = c.issue()
            wires_2 = c.issue()
            c.gate(AND, not_x1, not_x2, wires_2)
            c.gate(AND, wires_2, x3, p3)
            p4 = c.issue()
            c.gate(NIMP, wires_2, x3, p4)
            u1, v1, r1 = half(u), v, r
            s1 = double_without_overflow(c, s)
            k1 = add_constant_without_carry(c, k, 1)
            u2, v2 = u, half(v)
            r2 = double_without_overflow(c, r)
            s2 = s
            k2 = add_constant_without_carry(c, k, 1)
            u3 = sub_without_borrow(c, u1, v2)
            v3 = v
            r3 = add_without_carry(c, r, s)
            s3 = double_without_overflow(c, s)
            k3 = add_constant_without_carry(c, k, 1)
            u4 = u
            v4 = sub_without_borrow(c, v2, u1)
            r4 = double_without_overflow(c, r)
            s4 = add_without_carry(c, r, s)
            k4 = add_constant_without_carry(c, k, 1)

            def mix(w1, w2, w3, w4):
                t1 = self_or_zero_inv(c, w1, not_x1)
                t2 = self_or_zero(c, w2, p2)
                t3 = self_or_zero(c, w3, p3)
                t4 = self_or_zero(c, w4, p4)
                a1 = add_without_carry(c, t1, t2)
                a2 = add_without_carry(c, a1, t3)
                return add_without_carry(c, a2, t4)
            new_u = mix(u1, u2, u3, u4)
            new_v = mix(v1, v2, v3, v4)
            new_r = mix(r1, r2, r3, r4)
            new_s = mix(s1, s2, s3, s4)
            new_k = mix(k1, k2, k3, k4)
            v_eq_1 = equal_constant(c, v, 1)
            u = select(c, u, new_u, v_eq_1)
            v = select(c, v, new_v, v_eq_1)
            r = select(c, r, new_r, v_eq_1)
            s = select(c, s, new_s, v_eq_1)
            k = select(c, k, new_k, v_eq_1)
    # divide the result by the even part of the input
    c._call(s, even_part)  # "inverse::divide_result_by_even_part"
    for it0 in range(0, N_BITS, 4):
        c._call(s, even_part)  # "...::chunk"
        for _ in range(min(4, N_BITS - it0)):
            upd_s = fq_half(c, s)
            upd_e = fq_half(c, even_part)
            sel = equal_constant(c, even_part, 1)
            s = select(c, s, upd_s, sel)
            even_part = select(c, even_part, upd_e, sel)
    # divide the result by 2^k
    c._call(s, k)  # "inverse::divide_result_by_2^k"
    for it0 in range(0, 2 * N_BITS, 4):
        c._call(s, k)  # "...::chunk"
        for _ in range(min(4, 2 * N_BITS - it0)):
            upd_s = fq_half(c, s)
            upd_k = fq_add_constant(c, k, P - 1)
            sel = fq_equal_constant(c, k, 0)
            s = select(c, s, upd_s, sel)
            k = select(c, k, upd_k, sel)
    return s


R_MOD_P = (1 << N_BITS) % P


def fq_mul_by_constant(c, a, b):  # fp254impl.rs:252-275 (#[bn_component], b off-circuit: its integer value)
    c._call(a)
    if b == 0:
        return const_wires(0)
    if b == R_MOD_P:
        return list(a)
    return montgomery_reduce(c, mul_by_constant(c, a, b))


def fq_inverse_montgomery(c, a):  # fp254impl.rs:665-678
    return fq_mul_by_constant(c, fq_inverse(c, a), pow(R_MOD_P, 3, P))


# ------------------------------------------------------------------------------------------------ fq2 / fq12 / g1 / pairing steps
def fq2_square(c, a):  # fq2.rs:341-354
    p_ = fq_add(c, a[0], a[1])
    m_ = fq_sub(c, a[0], a[1])
    a0a1 = fq_mul(c, a[0], a[1])
    c0 = fq_mul(c, p_, m_)
    c1 = fq_double(c, a0a1)
    return [c0, c1]


def fq2_neg(c, a): return [fq_neg(c, a[0]), fq_neg(c, a[1])]    # fq2.rs:179-186
def fq2_half(c, a): return [fq_half(c, a[0]), fq_half(c, a[1])]  # fq2.rs:211-219
def fq2_mul_by_fq(c, a, b): return [fq_mul(c, a[0], b), fq_mul(c, a[1], b)]  # fq2.rs:282-291


def fq2_mul_by_constant(c, a, b):  # fq2.rs:257-280; b = (b0, b1) exactly as handed over
    if b == (1, 0):
        return [list(a[0]), list(a[1])]
    a_sum = fq_add(c, a[0], a[1])
    a0b0 = fq_mul_by_constant(c, a[0], b[0])
    a1b1 = fq_mul_by_constant(c, a[1], b[1])
    sm = fq_mul_by_constant(c, a_sum, (b[0] + b[1]) % P)
    c0 = fq_sub(c, a0b0, a1b1)
    s_ = fq_add(c, a0b0, a1b1)
    c1 = fq_sub(c, sm, s_)
    return [c0, c1]


def fq12_cyclotomic_square(c, a):  # fq12.rs:326-392 (not a component in the reference)
    c0, c1, c2 = a[0]
    c3, c4, c5 = a[1]

    def fp4(x, y, beta_of, added_to):
        xy = fq2_mul(c, x, y)
        x_plus_y = fq2_add(c, x, y)
        y_beta = fq2_mul_by_nonresidue(c, beta_of)
        x_plus_y_beta = fq2_add(c, added_to, y_beta)
        xy_beta = fq2_mul_by_nonresidue(c, xy)
        w1 = fq2_mul(c, x_plus_y, x_plus_y_beta)
        w2 = fq2_add(c, xy, xy_beta)
        return fq2_sub(c, w1, w2), fq2_double(c, xy)
    t0, t1 = fp4(c0, c4, c4, c0)
    t2, t3 = fp4(c2, c3, c2, c3)
    t4, t5 = fp4(c1, c5, c5, c1)

    def three_minus(t, cc):
        w1 = fq2_sub(c, t, cc)
        w2 = fq2_double(c, w1)
        return fq2_add(c, w2, t)

    def three_plus(t, cc):
        w1 = fq2_add(c, t, cc)
        w2 = fq2_double(c, w1)
        return fq2_add(c, w2, t)
    z0 = three_minus(t0, c0)
    z4 = three_minus(t2, c1)
    z3 = three_minus(t4, c2)
    t5_beta = fq2_mul_by_nonresidue(c, t5)
    z2 = three_plus(t5_beta, c3)
    z1 = three_plus(t1, c4)
    z5 = three_plus(t3, c5)
    return [[z0, z4, z3], [z2, z1, z5]]


def multiplexer_bit(c, a, s):  # basic.rs:74-105 (#[component]): in-place pairwise reduction by selector bits, LSB first
    c._call(a, s)
    cur = list(a)
    for sel in s:
        cur = [selector(c, cur[i + 1], cur[i], sel) for i in range(0, len(cur), 2)]
    return cur[0]


def bigint_multiplexer(c, a, s):  # cmp.rs:173-197 (#[bn_component]); a = list of 2^w bigints
    c._call(*a, s)
    return [multiplexer_bit(c, [a_i[bit] for a_i in a], s) for bit in range(len(a[0]))]


def g1_add(c, p, q):  # g1.rs:159-235 (#[component])
    c._call(*p, *q)
    x1, y1, z1 = p
    x2, y2, z2 = q
    z1s = fq_square(c, z1); z2s = fq_square(c, z2)
    z1c = fq_mul(c, z1s, z1); z2c = fq_mul(c, z2s, z2)
    u1 = fq_mul(c, x1, z2s); u2 = fq_mul(c, x2, z1s)
    s1 = fq_mul(c, y1, z2c); s2 = fq_mul(c, y2, z1c)
    r = fq_sub(c, s1, s2); h = fq_sub(c, u1, u2)
    h2 = fq_square(c, h)
    g = fq_mul(c, h, h2); v = fq_mul(c, u1, h2)
    r2 = fq_square(c, r)
    r2g = fq_add(c, r2, g)
    vd = fq_double(c, v)
    x3 = fq_sub(c, r2g, vd)
    vx3 = fq_sub(c, v, x3)
    w = fq_mul(c, r, vx3)
    s1g = fq_mul(c, s1, g)
    y3 = fq_sub(c, w, s1g)
    z1z2 = fq_mul(c, z1, z2)
    z3 = fq_mul(c, z1z2, h)
    z1_0 = fq_equal_constant(c, z1, 0)
    z2_0 = fq_equal_constant(c, z2, 0)
    zero = const_wires(0)
    sel = [z1_0, z2_0]
    x = bigint_multiplexer(c, [x3, x2, x1, zero], sel)
    y = bigint_multiplexer(c, [y3, y2, y1, zero], sel)
    z = bigint_multiplexer(c, [z3, z2, z1, zero], sel)
    return [x, y, z]


G2_COEFF_B = (19485874751759354771024239261021720505790618469301721065564631296452457478373, 266929791119991161246907387137283842545076965332900288569378510910307636690)  # ark_bn254::g2::Config::COEFF_B


def _mont2(b): return (b[0] * R_MOD_P % P, b[1] * R_MOD_P % P)


def g2_double_in_place(c, r):  # pairing.rs:359-407 (#[component]); returns (new r, line coefficients)
    c._call(*[w for f2 in r for w in f2])
    rx, ry, rz = r
    a = fq2_mul(c, rx, ry)
    a = fq2_half(c, a)
    b = fq2_square(c, ry)
    cc = fq2_square(c, rz)
    c_triple = fq2_triple(c, cc)
    e = fq2_mul_by_constant(c, c_triple, _mont2(G2_COEFF_B))
    f = fq2_triple(c, e)
    g = fq2_add(c, b, f)
    g = fq2_half(c, g)
    ryrz = fq2_add(c, ry, rz)
    ryrzs = fq2_square(c, ryrz)
    bc = fq2_add(c, b, cc)
    h = fq2_sub(c, ryrzs, bc)
    i = fq2_sub(c, e, b)
    j = fq2_square(c, rx)
    es = fq2_square(c, e)
    j_triple = fq2_triple(c, j)
    bf = fq2_sub(c, b, f)
    new_x = fq2_mul(c, a, bf)
    es_triple = fq2_triple(c, es)
    gs = fq2_square(c, g)
    new_y = fq2_sub(c, gs, es_triple)
    new_z = fq2_mul(c, b, h)
    hn = fq2_neg(c, h)
    return [new_x, new_y, new_z], [hn, j_triple, i]


def g2_add_in_place(c, r, q):  # pairing.rs:409-464 (#[component])
    c._call(*[w for f2 in r for w in f2], *[w for f2 in q for w in f2])
    rx, ry, rz = r
    qx, qy = q[0], q[1]
    w1 = fq2_mul(c, qy, rz)
    theta = fq2_sub(c, ry, w1)
    w2 = fq2_mul(c, qx, rz)
    lam = fq2_sub(c, rx, w2)
    cc = fq2_square(c, theta)
    d = fq2_square(c, lam)
    e = fq2_mul(c, lam, d)
    f = fq2_mul(c, rz, cc)
    g = fq2_mul(c, rx, d)
    w3 = fq2_add(c, e, f)
    w4 = fq2_double(c, g)
    h = fq2_sub(c, w3, w4)
    neg_theta = fq2_neg(c, theta)
    w5 = fq2_mul(c, theta, qx)
    w6 = fq2_mul(c, lam, qy)
    j = fq2_sub(c, w5, w6)
    new_x = fq2_mul(c, lam, h)
    w7 = fq2_sub(c, g, h)
    w8 = fq2_mul(c, theta, w7)
    w9 = fq2_mul(c, e, ry)
    new_y = fq2_sub(c, w8, w9)
    new_z = fq2_mul(c, rz, e)
    return [new_x, new_y, new_z], [lam, neg_theta, j]


def fq6_mul_by_fq2(c, a, b): return [fq2_mul(c, a[k], b) for k in range(3)]  # fq6.rs:326-332


def fq6_mul_by_01(c, a, c0, c1):  # fq6.rs:351-379
    w1 = fq2_mul(c, a[0], c0)
    w2 = fq2_mul(c, a[1], c1)
    w3 = fq2_add(c, a[1], a[2])
    w4 = fq2_mul(c, w3, c1)
    w5 = fq2_sub(c, w4, w2)
    w6 = fq2_mul_by_nonresidue(c, w5)
    w7 = fq2_add(c, w6, w1)
    w8 = fq2_add(c, a[0], a[1])
    w9 = fq2_add(c, c0, c1)
    w10 = fq2_mul(c, w8, w9)
    w11 = fq2_sub(c, w10, w1)
    w12 = fq2_sub(c, w11, w2)
    w13 = fq2_add(c, a[0], a[2])
    w14 = fq2_mul(c, w13, c0)
    w15 = fq2_sub(c, w14, w1)
    w16 = fq2_add(c, w15, w2)
    return [w7, w12, w16]


def fq12_mul_by_034(c, a, c0, c3, c4):  # fq12.rs:266-285 (#[component])
    c._call(_flat12(a), c0[0], c0[1], c3[0], c3[1], c4[0], c4[1])
    w1 = fq6_mul_by_01(c, a[1], c3, c4)
    w2 = fq6_mul_by_nonresidue(c, w1)
    w3 = fq6_mul_by_fq2(c, a[0], c0)
    new_c0 = fq6_add(c, w2, w3)
    w4 = fq6_add(c, a[0], a[1])
    w5 = fq2_add(c, c3, c0)
    w6 = fq6_mul_by_01(c, w4, w5, c4)
    w7 = fq6_add(c, w1, w3)
    new_c1 = fq6_sub(c, w6, w7)
    return [new_c0, new_c1]


def ell(c, f, coeffs, px, py):  # pairing.rs:160-171 (not a component)
    c0 = fq2_mul_by_fq(c, coeffs[0], py)
    c3 = fq2_mul_by_fq(c, coeffs[1], px)
    return fq12_mul_by_034(c, f, c0, c3, coeffs[2])


def fq2_add_constant(c, a, b): return [fq_add_constant(c, a[0], b[0]), fq_add_constant(c, a[1], b[1])]  # fq2.rs:170-177


def fq2_mul_constant_by_fq(c, a_std, b):  # fq2.rs:307-322 (#[component(offcircuit_args = "a")]): a a STANDARD-form constant, b a wire
    c._call(b)
    return [fq_mul_by_constant(c, b, a_std[0] * R_MOD_P % P), fq_mul_by_constant(c, b, a_std[1] * R_MOD_P % P)]


def fq6_mul_by_01_constant1(c, a, c0, c1):  # fq6.rs:381-410; c1 a constant in Montgomery form
    w1 = fq2_mul(c, a[0], c0)
    w2 = fq2_mul_by_constant(c, a[1], c1)
    w3 = fq2_add(c, a[1], a[2])
    w4 = fq2_mul_by_constant(c, w3, c1)
    w5 = fq2_sub(c, w4, w2)
    w6 = fq2_mul_by_nonresidue(c, w5)
    w7 = fq2_add(c, w6, w1)
    w8 = fq2_add(c, a[0], a[1])
    w9 = fq2_add_constant(c, c0, c1)
    w10 = fq2_mul(c, w8, w9)
    w11 = fq2_sub(c, w10, w1)
    w12 = fq2_sub(c, w11, w2)
    w13 = fq2_add(c, a[0], a[2])
    w14 = fq2_mul(c, w13, c0)
    w15 = fq2_sub(c, w14, w1)
    w16 = fq2_add(c, w15, w2)
    return [w7, w12, w16]


def fq12_mul_by_034_constant4(c, a, c0, c3, c4):  # fq12.rs:287-310 (#[component(offcircuit_args = "c4")])
    c._call(_flat12(a), c0[0], c0[1], c3[0], c3[1])
    w1 = fq6_mul_by_01_constant1(c, a[1], c3, c4)
    w2 = fq6_mul_by_nonresidue(c, w1)
    w3 = fq6_mul_by_fq2(c, a[0], c0)
    new_c0 = fq6_add(c, w2, w3)
    w4 = fq6_add(c, a[0], a[1])
    w5 = fq2_add(c, c3, c0)
    w6 = fq6_mul_by_01_constant1(c, w4, w5, c4)
    w7 = fq6_add(c, w1, w3)
    new_c1 = fq6_sub(c, w6, w7)
    return [new_c0, new_c1]


def ell_by_constant(c, f, coeffs_std, p):  # pairing.rs:923-942 (#[component(offcircuit_args = "coeffs")]); p = (x, y, z) wires
    c._call(_flat12(f), p[0], p[1], p[2])
    new_c0 = fq2_mul_constant_by_fq(c, coeffs_std[0], p[1])
    new_c1 = fq2_mul_constant_by_fq(c, coeffs_std[1], p[0])
    return fq12_mul_by_034_constant4(c, f, new_c0, new_c1, _mont2(coeffs_std[2]))


def _ell_const_circuit(k):
    import bn254_ref as T
    coeffs = T.ell_coeffs(T.G2_GEN)[k]  # the reference's native ell_coeffs(q) (pairing.rs:88-126) for the G2 generator: host constants
    return lambda c, i: _flat12(ell_by_constant(c, _fq12(i, 0), coeffs, [_fq(i, 12), _fq(i, 13), _fq(i, 14)]))


def _g2(i, base): return [[_fq(i, base + 2 * k), _fq(i, base + 2 * k + 1)] for k in range(3)]
def _g2_step_out(rc): return [w for part in rc for f2 in part for fq in f2 for w in fq]
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
    "fq12_cyclotomic_square": (3048, lambda c, i: _flat12(fq12_cyclotomic_square(c, _fq12(i, 0)))),
    "fq_inverse": (254, lambda c, i: fq_inverse_montgomery(c, i)),
    "g1_add": (1524, lambda c, i: sum(g1_add(c, [_fq(i, 0), _fq(i, 1), _fq(i, 2)], [_fq(i, 3), _fq(i, 4), _fq(i, 5)]), [])),
    "g2_double": (1524, lambda c, i: _g2_step_out(g2_double_in_place(c, _g2(i, 0)))),
    "g2_add": (3048, lambda c, i: _g2_step_out(g2_add_in_place(c, _g2(i, 0), _g2(i, 6)))),
    "ell_eval": (3048 + 1524 + 508, lambda c, i: _flat12(ell(c, _fq12(i, 0), [[_fq(i, 12 + 2 * k), _fq(i, 13 + 2 * k)] for k in range(3)], _fq(i, 18), _fq(i, 19)))),
    "ell_const:0": (3048 + 762, _ell_const_circuit(0)),   # a doubling step's line ...
    "ell_const:3": (3048 + 762, _ell_const_circuit(3)),   # ... and the first addition step's (ATE_LOOP_COUNT: 64th digit 0, 63rd 1)
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
