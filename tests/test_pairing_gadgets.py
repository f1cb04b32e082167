"""Execute-mode checks of the Miller-loop gadgets (csrc/gadgets/bn254_pairing.hpp) against tests/bn254_ref.py, a plain-Python
mirror of the reference's NATIVE helpers (src/gadgets/bn254/pairing.rs:30-133) which is itself checked to be a pairing
(bilinear, non-degenerate, of order r) below."""
import os
import random

import numpy as np
import pytest

import bn254_ref as T
import oracle_lib as o

P = T.P
RM = o.FQ_R % P


def bits_of(vals):
    return np.concatenate([o.int_to_bits(v % P, 254) for v in vals])


def ints_of(bits):
    return [o.bits_to_int(bits[i * 254:(i + 1) * 254]) for i in range(len(bits) // 254)]


def to_m(v):
    return [(x * RM) % P for x in v]


def from_m(v):
    return [(x * T.RINV) % P for x in v]


def flat2(*f2s):
    return [x for a in f2s for x in a]


def test_python_mirror_is_a_pairing():
    P1, Q1 = T.g1_mul(5), T.g2_mul(7)
    e1 = T.final_exponentiation(T.multi_miller_loop([(P1, Q1)]))
    assert e1 == T.final_exponentiation(T.multi_miller_loop([(T.g1_mul(35), T.G2_GEN)])) == T.final_exponentiation(T.multi_miller_loop([(T.g1_mul(1), T.g2_mul(35))]))
    assert e1 != T.F12_ONE and T.f12_pow(e1, T.R_ORDER) == T.F12_ONE
    negP = (P1[0], (-P1[1]) % P)
    assert T.final_exponentiation(T.multi_miller_loop([(P1, Q1), (negP, Q1)])) == T.F12_ONE


def test_g2_steps_and_line_evaluations():
    """double_in_place_circuit / add_in_place / mul_by_char (pairing.rs:359-501), ell_montgomery (:160-171) and
    ell_by_constant_montgomery (:923-942, constant = a line coefficient of the G2 generator) == the mirror."""
    random.seed(21)
    q = T.g2_mul(11)
    r = (q[0], q[1], (1, 0))
    for _ in range(2):  # a generic projective point: r after two doublings
        r, _c = T.g2_double_in_place(r)
    ob, gc, _ = o.execute("g2_double", bits_of(to_m(flat2(*r))), capacity=200_000)
    nr, cf = T.g2_double_in_place(r)
    assert from_m(ints_of(ob)) == flat2(*nr) + flat2(*cf)
    q2 = T.g2_mul(3)
    ob, _, _ = o.execute("g2_add", bits_of(to_m(flat2(*r) + flat2(q2[0], q2[1], (1, 0)))), capacity=200_000)
    nr, cf = T.g2_add_in_place(r, q2)
    assert from_m(ints_of(ob)) == flat2(*nr) + flat2(*cf)
    ob, _, _ = o.execute("g2_mul_by_char", bits_of(to_m(flat2(q[0], q[1], (1, 0)))), capacity=200_000)
    s = T.g2_mul_by_char(q)
    assert from_m(ints_of(ob)) == flat2(s[0], s[1], (1, 0))
    f = T.f12_unflatten([random.randrange(P) for _ in range(12)])
    p = T.g1_mul(9)
    ob, _, _ = o.execute("ell_eval", bits_of(to_m(T.f12_flatten(f) + flat2(*cf) + [p[0], p[1]])), capacity=200_000)
    assert from_m(ints_of(ob)) == T.f12_flatten(T.ell(f, cf, p))
    ells = T.ell_coeffs(T.G2_GEN)
    assert len(ells) == 64 + sum(1 for b in T.ATE_LOOP_COUNT[:64] if b) + 2
    for k in (0, 5, len(ells) - 1):
        ob, _, _ = o.execute("ell_const:%d" % k, bits_of(to_m(T.f12_flatten(f) + [p[0], p[1], 1])), capacity=200_000)
        assert from_m(ints_of(ob)) == T.f12_flatten(T.ell(f, ells[k], p))


@pytest.mark.skipif(not os.environ.get("GSV_SLOW"), reason="the Miller loop is ~3 B gates (~100 s through the CPU oracle); set GSV_SLOW=1")
def test_miller_loop_matches_mirror_and_is_bilinear():
    """multi_miller_loop_groth16_evaluate_montgomery_fast (pairing.rs:944-1007) with q1 = G, q2 = -G constant and q3 on wires ==
    the mirror's multi Miller loop, and final_exp of it equals e(p1, G) e(p2, -G) e(p3, q3)."""
    p1, p2, p3, q3 = T.g1_mul(3), T.g1_mul(8), T.g1_mul(5), T.g2_mul(4)
    ins = [p1[0], p1[1], 1, p2[0], p2[1], 1, p3[0], p3[1], 1] + flat2(q3[0], q3[1], (1, 0))
    ob, gc, peak = o.execute("miller_loop", bits_of(to_m(ins)), capacity=400_000)
    negG = (T.G2_GEN[0], T.f2_neg(T.G2_GEN[1]))
    want = T.multi_miller_loop([(p1, T.G2_GEN), (p2, negG), (p3, q3)])
    got = T.f12_unflatten(from_m(ints_of(ob)))
    assert T.f12_flatten(got) == T.f12_flatten(want)
    # e(3G1, G2) e(8G1, -G2) e(5G1, 4G2) = e(G1, G2)^(3 - 8 + 20)
    e = T.final_exponentiation(got)
    assert e == T.f12_pow(T.final_exponentiation(T.multi_miller_loop([(T.g1_mul(1), T.G2_GEN)])), 15)
    print("miller_loop gates", int(gc.sum()), "non-free", int(gc[:8].sum()), "peak live", peak)
