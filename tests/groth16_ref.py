"""Synthetic Groth16 instances for the verifier circuit (test helper).

A verifying key / proof / public-input triple that satisfies the verification equation is built from known discrete logs:
alpha = a1 G1, beta = b2 G2, gamma = g G2, delta = d G2, gamma_abc_i = k_i G1, A = a G1, B = b G2 and
C = c G1 with c = (a b - alpha beta - msm gamma) / delta mod r, msm = k_0 + sum x_i k_i.  No SNARK setup is involved; the
circuit under test (reference: src/gadgets/groth16.rs:58-110) only sees group elements."""
import random

import numpy as np

import bn254_ref as T
import oracle_lib as o

P, R = T.P, T.R_ORDER
RM = o.FQ_R % P


def f12_inv(x):
    return T.f12_pow(x, P ** 12 - 2)


def make_instance(n_pub=2, seed=1):
    rnd = random.Random(seed)
    s = lambda: rnd.randrange(1, R)
    a1, b2, g, d, a, b = s(), s(), s(), s(), s(), s()
    ks = [s() for _ in range(n_pub + 1)]
    xs = [s() for _ in range(n_pub)]
    msm = (ks[0] + sum(x * k for x, k in zip(xs, ks[1:]))) % R
    c = (a * b - a1 * b2 - msm * g) * pow(d, -1, R) % R
    beta = T.g2_mul(b2)
    neg_beta = (beta[0], T.f2_neg(beta[1]))
    alpha_beta = f12_inv(T.final_exponentiation(T.multi_miller_loop([(T.g1_mul(a1), neg_beta)])))
    return {"n_pub": n_pub, "public": xs, "gamma_abc": [T.g1_mul(k) for k in ks], "gamma": T.g2_mul(g), "delta": T.g2_mul(d),
            "alpha": T.g1_mul(a1), "beta": beta, "alpha_beta": T.f12_flatten(alpha_beta), "A": T.g1_mul(a), "B": T.g2_mul(b), "C": T.g1_mul(c)}


def check_instance(inst):
    """The verifier's equation as the reference's circuit states it: FE(ML(msm,-gamma; C,-delta; A,B)) == alpha_beta."""
    msm = inst["gamma_abc"][0]
    acc_scalar = None
    pts = [msm]
    for x, base in zip(inst["public"], inst["gamma_abc"][1:]):
        pts.append(_g1_mul_point(x, base))
    msm = pts[0]
    for q in pts[1:]:
        msm = _g1_add(msm, q)
    neg = lambda q: (q[0], T.f2_neg(q[1]))
    f = T.final_exponentiation(T.multi_miller_loop([(msm, neg(inst["gamma"])), (inst["C"], neg(inst["delta"])), (inst["A"], inst["B"])]))
    return T.f12_flatten(f) == list(inst["alpha_beta"])


def _g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0: return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def _g1_mul_point(k, p):
    return T.g1_mul(k % R, p)


def vk_blob(inst):
    fe = lambda v: int(v).to_bytes(32, "big")
    out = bytes([inst["n_pub"]])
    for x, y in inst["gamma_abc"]:
        out += fe(x) + fe(y)
    for q in (inst["gamma"], inst["delta"]):
        out += fe(q[0][0]) + fe(q[0][1]) + fe(q[1][0]) + fe(q[1][1])
    for v in inst["alpha_beta"]:
        out += fe(v)
    return out


def circuit_name(inst):
    return "groth16_verify:" + vk_blob(inst).hex()


def input_bits(inst):
    """CircuitInput order (groth16.rs:290-318): public scalars (plain bits), A, B, C (Montgomery form, z = 1)."""
    m = lambda v: (v * RM) % P
    bits = [o.int_to_bits(x % R, 254) for x in inst["public"]]
    A, B, C = inst["A"], inst["B"], inst["C"]
    for v in [A[0], A[1], 1, B[0][0], B[0][1], B[1][0], B[1][1], 1, 0, C[0], C[1], 1]:
        bits.append(o.int_to_bits(m(v), 254))
    return np.concatenate(bits)


# ---- point compression as the reference's circuit undoes it (groth16.rs:116-182) ----
def fq_sqrt_circuit(v):  # fq.rs:290-299: v^((p+1)/4)
    return pow(v, (P + 1) // 4, P)


def fq2_sqrt_circuit(a):  # fq2.rs:425-446 (complex method, general case)
    c0, c1 = a
    alpha = (c0 * c0 + c1 * c1) % P
    alpha_sqrt = fq_sqrt_circuit(alpha)
    delta = (alpha_sqrt + c0) * T.HALF % P
    if pow(delta, (P - 1) // 2, P) == P - 1:
        delta = (delta - alpha_sqrt) % P
    r0 = fq_sqrt_circuit(delta)
    r1 = c1 * T.HALF % P * pow(r0, -1, P) % P
    return (r0, r1)


def compressed_input_bits(inst):
    """Groth16VerifyCompressedInput order (groth16.rs:410-421): public scalars, (A.x, flag), (B.x, flag), (C.x, flag); a flag is 1
    when the circuit's own square root IS the point's y (0: its negative)."""
    m = lambda v: (v * RM) % P
    bits = [o.int_to_bits(x % R, 254) for x in inst["public"]]
    A, B, C = inst["A"], inst["B"], inst["C"]

    def g1(pt):
        sy = fq_sqrt_circuit((pt[0] ** 3 + 3) % P)
        assert sy in (pt[1], P - pt[1])
        return [o.int_to_bits(m(pt[0]), 254), np.array([int(sy == pt[1])], np.uint8)]

    bits += g1(A)
    y2 = T.f2_add(T.f2_mul(T.f2_sq(B[0]), B[0]), T.COEFF_B_G2)
    sy = fq2_sqrt_circuit(y2)
    assert sy in (B[1], T.f2_neg(B[1]))
    bits += [o.int_to_bits(m(B[0][0]), 254), o.int_to_bits(m(B[0][1]), 254), np.array([int(sy == B[1])], np.uint8)]
    bits += g1(C)
    return np.concatenate(bits)


def compressed_circuit_name(inst):
    return "groth16_verify_compressed:" + vk_blob(inst).hex()
