"""Execute-mode results of the restated gadget producers against plain Python integer arithmetic, and the
gate counts the survey established independently.  Mirrors the reference's gadget
tests (src/gadgets/bn254/fq.rs:564-595, bigint/add.rs:333-480, bigint/mul.rs:547-900)."""
import random
import re
import os

import numpy as np
import pytest

import bn254_ref as T
import oracle_lib as o

P = T.P


def bits_of(vals, n=254):
    return np.concatenate([o.int_to_bits(v, n) for v in vals])


def ints_of(bits, n=254):
    return [o.bits_to_int(bits[i * n:(i + 1) * n]) for i in range(len(bits) // n)]


def test_u254_add_config1():
    random.seed(1)
    for a, b in [((1 << 254) - 1, 1), (0, 0), (random.getrandbits(254), random.getrandbits(254))]:
        ob, gc, _ = o.execute("u254_add", bits_of([a, b]), capacity=10_000)
        assert o.bits_to_int(ob) == a + b
    assert int(gc.sum()) == 1267 and int(gc[:8].sum()) == 254
    g = o.garble("u254_add", 0, capacity=10_000)
    assert g.n_ciphertexts == 254  # last carry is a root output: nothing is dead


@pytest.mark.parametrize("n", [1, 2, 4, 5, 19, 20, 21, 22, 40])
def test_bigint_mul_naive_and_karatsuba(n):
    random.seed(n)
    for _ in range(3):
        a, b = random.getrandbits(n), random.getrandbits(n)
        ob, _, _ = o.execute("bigint_mul:%d" % n, np.concatenate([o.int_to_bits(a, n), o.int_to_bits(b, n)]))
        assert o.bits_to_int(ob) == a * b


@pytest.mark.parametrize("n", [1, 4, 33])
def test_bigint_add_sub(n):
    random.seed(n)
    for _ in range(4):
        a, b = random.getrandbits(n), random.getrandbits(n)
        ob, _, _ = o.execute("bigint_add:%d" % n, np.concatenate([o.int_to_bits(a, n), o.int_to_bits(b, n)]))
        assert o.bits_to_int(ob) == a + b
        ob, _, _ = o.execute("bigint_sub:%d" % n, np.concatenate([o.int_to_bits(a, n), o.int_to_bits(b, n)]))
        assert o.bits_to_int(ob) == (a - b) % (1 << (n + 1))  # n+1 bits: difference with borrow on top


def test_fq_ops_and_counts():
    random.seed(7)
    cases = [(P - 1, P - 1), (0, 0), (1, P - 1)] + [(random.randrange(P), random.randrange(P)) for _ in range(3)]
    for a, b in cases:
        assert ints_of(o.execute("fq_add", bits_of([a, b]))[0]) == [(a + b) % P]
        # neg(0) yields the non-canonical representative p (one conditional subtraction only,
        # fp254impl.rs:153-168 + :117-141), so negation / subtraction are compared modulo p.
        assert ints_of(o.execute("fq_sub", bits_of([a, b]))[0])[0] % P == (a - b) % P
        assert ints_of(o.execute("fq_neg", bits_of([a]))[0])[0] % P == (-a) % P
        assert ints_of(o.execute("fq_double", bits_of([a]))[0]) == [(2 * a) % P]
        assert ints_of(o.execute("fq_half", bits_of([a]))[0]) == [(a * pow(2, -1, P)) % P]
        assert ints_of(o.execute("fq_triple", bits_of([a]))[0]) == [(3 * a) % P]
        assert ints_of(o.execute("fq_div6", bits_of([a]))[0]) == [(a * pow(6, -1, P)) % P]
        r, gc, _ = o.execute("fq_mul", bits_of([a, b]), capacity=100_000)
        assert ints_of(r) == [(a * b * T.RINV) % P]
        r, _, _ = o.execute("fq_complex", bits_of([a, b]), capacity=100_000)
        assert ints_of(r) == [(((a * a * T.RINV) % P) * b * T.RINV + a) % P]
    assert int(gc.sum()) == 414_284 and int(gc[:8].sum()) == 102_093  # SURVEY Appendix C (two independent tallies)
    for spec, total in [("fq_add", 3298), ("fq_sub", 6090), ("fq_double", 2031), ("fq_div6", 5327)]:
        n_in, _ = o.circuit_info(spec)
        _, gc, _ = o.execute(spec, np.zeros(n_in, np.uint8))
        assert int(gc.sum()) == total


def test_fq2_fq6_fq12_mul():
    random.seed(11)
    a = [random.randrange(P) for _ in range(2)]
    b = [random.randrange(P) for _ in range(2)]
    ob, gc, _ = o.execute("fq2_mul", bits_of(a + b))
    assert ints_of(ob) == [(x * T.RINV) % P for x in T.f2_mul(tuple(a), tuple(b))]
    assert int(gc.sum()) == 1_264_926
    a = [random.randrange(P) for _ in range(6)]
    b = [random.randrange(P) for _ in range(6)]
    ob, gc, _ = o.execute("fq6_mul", bits_of(a + b))
    assert ints_of(ob) == [(x * T.RINV) % P for x in T.f6_flatten(T.f6_mul(T.f6_unflatten(a), T.f6_unflatten(b)))]
    a = [random.randrange(P) for _ in range(12)]
    b = [random.randrange(P) for _ in range(12)]
    ob, gc, peak = o.execute("fq12_mul", bits_of(a + b), capacity=15_000)
    assert ints_of(ob) == [(x * T.RINV) % P for x in T.f12_flatten(T.f12_mul(T.f12_unflatten(a), T.f12_unflatten(b)))]
    assert int(gc.sum()) == 20_284_982 and peak <= 15_000  # capacity the reference's own test uses (tests/fq12_mul_e2e.rs:190)


def test_fq12_square_and_cyclotomic_square():
    """Fq12::square_montgomery (fq12.rs:311-324) on a random element and cyclotomic_square_montgomery (fq12.rs:326-392) on
    an element of the cyclotomic subgroup f^((p^6-1)(p^2+1)), both against Python field arithmetic; the Granger-Scott formula
    must NOT square a generic element (the test would be blind otherwise)."""
    random.seed(5)
    a = [random.randrange(P) for _ in range(12)]
    ob, gc, _ = o.execute("fq12_square", bits_of(a), capacity=100_000)
    A = T.f12_unflatten(a)
    assert ints_of(ob) == [(x * T.RINV) % P for x in T.f12_flatten(T.f12_mul(A, A))]
    assert int(gc.sum()) == 13_595_222 and int(gc[:8].sum()) == 3_664_258
    one = T.f12_unflatten([1] + [0] * 11)

    def f12_pow(x, e):
        r = one
        while e:
            if e & 1:
                r = T.f12_mul(r, x)
            x = T.f12_mul(x, x)
            e >>= 1
        return r

    f = T.f12_unflatten([random.randrange(P) for _ in range(12)])
    c = f12_pow(f, (P ** 6 - 1) * (P ** 2 + 1))
    R = o.FQ_R % P
    ob, gc, _ = o.execute("fq12_cyclotomic_square", bits_of([(x * R) % P for x in T.f12_flatten(c)]), capacity=100_000)
    assert ints_of(ob) == [(x * R) % P for x in T.f12_flatten(T.f12_mul(c, c))]
    assert int(gc.sum()) == 8_032_850
    ob, _, _ = o.execute("fq12_cyclotomic_square", bits_of(a), capacity=100_000)
    assert ints_of(ob) != [(x * T.RINV) % P for x in T.f12_flatten(T.f12_mul(A, A))]


def _to_m(v):
    return [(x * (o.FQ_R % P)) % P for x in v]


def _from_m(v):
    return [(x * T.RINV) % P for x in v]


_ONE12 = T.f12_unflatten([1] + [0] * 11)


def _f12_pow(x, e):
    r = _ONE12
    while e:
        if e & 1:
            r = T.f12_mul(r, x)
        x = T.f12_mul(x, x)
        e >>= 1
    return r


def test_inverses_frobenius_conjugate():
    """bn254_ext.hpp against Python field arithmetic: Fq::inverse_montgomery (binary extended Euclid, fp254impl.rs:333-678;
    edge values 1, 2, p-1, a multiple of 2^40), Fq2 / Fq12 inverses (a * a^-1 == 1), Frobenius maps x -> x^(p^i) for i = 1..3
    (which also pins the tabulated ark_bn254 coefficients), conjugation = x^(p^6)."""
    random.seed(1)
    for a in [random.randrange(1, P), 1, 2, P - 1, 12 * (1 << 40)]:
        ob, gc, _ = o.execute("fq_inverse", bits_of(_to_m([a])), capacity=200_000)
        assert ints_of(ob) == _to_m([pow(a, -1, P)])
    assert int(gc.sum()) == 23_200_543
    a = [random.randrange(P) for _ in range(2)]
    ob, _, _ = o.execute("fq2_inverse", bits_of(_to_m(a)), capacity=200_000)
    assert T.f2_mul(tuple(a), tuple(_from_m(ints_of(ob)))) == (1, 0)
    a = [random.randrange(P) for _ in range(12)]
    A = T.f12_unflatten(a)
    ob, gc, _ = o.execute("fq12_inverse", bits_of(_to_m(a)), capacity=200_000)
    assert T.f12_flatten(T.f12_mul(A, T.f12_unflatten(_from_m(ints_of(ob))))) == T.f12_flatten(_ONE12)
    assert int(gc.sum()) == 61_993_136
    for i in (1, 2, 3):
        ob, _, _ = o.execute("fq12_frobenius:%d" % i, bits_of(_to_m(a)), capacity=200_000)
        assert _from_m(ints_of(ob)) == T.f12_flatten(_f12_pow(A, P ** i))
    ob, _, _ = o.execute("fq12_conjugate", bits_of(_to_m(a)), capacity=200_000)
    assert _from_m(ints_of(ob)) == T.f12_flatten(_f12_pow(A, P ** 6))


@pytest.mark.skipif(not os.environ.get("GSV_SLOW"), reason="3.5 B gates through the CPU oracle take ~95 s; set GSV_SLOW=1")
def test_final_exponentiation_matches_native_formula():
    """final_exponentiation_montgomery (final_exponentiation.rs:99-135) in Execute mode == the reference's own native formula
    (final_exponentiation.rs:37-63) evaluated with Python field arithmetic, and the result is an r-th root of unity."""
    X = 4965661367192848881
    conj = lambda x: _f12_pow(x, P ** 6)
    inv = lambda x: _f12_pow(x, P ** 12 - 2)
    frob = lambda x, i: _f12_pow(x, P ** i)
    nx = lambda f: conj(_f12_pow(f, X))
    m = T.f12_mul

    def native(f):
        u = m(inv(f), conj(f)); r = m(frob(u, 2), u)
        y0 = nx(r); y1 = m(y0, y0); y2 = m(y1, y1); y3 = m(y2, y1)
        y4 = nx(y3); y5 = m(y4, y4); y6 = nx(y5); y7 = conj(y3); y8 = conj(y6)
        y9 = m(y8, y4); y10 = m(y9, y7); y11 = m(y10, y1); y12 = m(y10, y4); y13 = m(y12, r)
        y14 = frob(y11, 1); y15 = m(y14, y13); y16 = frob(y10, 2); y17 = m(y16, y15); y18 = m(conj(r), y11)
        return m(frob(y18, 3), y17)

    random.seed(3)
    a = [random.randrange(P) for _ in range(12)]
    ob, gc, peak = o.execute("final_exp", bits_of(_to_m(a)), capacity=400_000)
    got = _from_m(ints_of(ob))
    assert got == T.f12_flatten(native(T.f12_unflatten(a)))
    assert int(gc.sum()) == 3_519_328_217 and int(gc[:8].sum()) == 955_048_646 and peak < 30_000
    r_order = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert T.f12_flatten(_f12_pow(T.f12_unflatten(got), r_order)) == T.f12_flatten(_ONE12)


def test_host_constants_match_python():
    """Off-circuit constants embedded in csrc/gadgets/bn254.hpp (fq.rs:56-76, fp254impl.rs:21-66)."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "garbled_snark_verifier_amd", "csrc", "gadgets", "bn254.hpp")).read()
    c = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"static const BigU& (\w+)\(\) \{ static BigU v = BigU::from_hex\(\"([0-9a-f]+)\"\)", src)}
    R = 1 << 254
    assert c["modulus"] == P
    assert (c["m_inverse"] * P) % R == 1
    assert c["m_inverse"] == 4759646384140481320982610724935209484903937857060724391493050186936685796471  # fq.rs:59-60
    assert c["not_modulus"] == R - P
    assert c["half_modulus"] == pow(2, -1, P) and c["one_third_modulus"] == pow(3, -1, P) and c["two_third_modulus"] == (2 * pow(3, -1, P)) % P
    assert c["neg_addend"] == (1 - (R - P)) % P
