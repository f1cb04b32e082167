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
