"""Execute-mode checks of the remaining groth16_verify gadgets (csrc/gadgets/bn254_groth16.hpp): G1 addition, the constant-base
window scalar multiplication, projective -> affine and — slow — the whole verifier on a valid and on a tampered proof.

Points are compared as GROUP ELEMENTS (affine) against tests/bn254_ref.py: the Jacobian representative a gadget returns is
the gadget's business.  The proofs are synthesized from known discrete logs (groth16_ref.py): no SNARK setup is needed to
satisfy e(A, B) = e(alpha, beta) e(msm, gamma) e(C, delta)."""
import os
import random

import numpy as np
import pytest

import bn254_ref as T
import groth16_ref as G
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


def jac(pt, z):  # affine -> a Jacobian representative with the given z
    return [pt[0] * z * z % P, pt[1] * z * z * z % P, z]


def affine(xyz):
    x, y, z = xyz
    if z == 0:
        return None
    zi = pow(z, -1, P)
    return (x * zi * zi % P, y * zi * zi * zi % P)


def test_g1_add_montgomery():
    """g1.rs:159-235: generic addition, either operand at infinity (z = 0) and — as in the reference — no doubling case."""
    random.seed(5)
    a, b = T.g1_mul(1234567), T.g1_mul(7654321)
    ja, jb = jac(a, random.randrange(1, P)), jac(b, random.randrange(1, P))
    ob, gc, _ = o.execute("g1_add", bits_of(to_m(ja + jb)))
    assert affine(from_m(ints_of(ob))) == T.g1_mul(1234567 + 7654321)
    inf = [random.randrange(P), random.randrange(P), 0]
    ob, _, _ = o.execute("g1_add", bits_of(to_m(inf + jb)))
    assert from_m(ints_of(ob)) == jb
    ob, _, _ = o.execute("g1_add", bits_of(to_m(ja + inf)))
    assert from_m(ints_of(ob)) == ja
    ob, _, _ = o.execute("g1_add", bits_of(to_m(inf + inf)))
    assert from_m(ints_of(ob)) == [0, 0, 0]
    ob, _, _ = o.execute("g1_add", bits_of(to_m(ja + jac(a, 77))))  # P + P: h = r = 0 -> z3 = 0
    assert from_m(ints_of(ob))[2] == 0
    assert int(gc.sum()) > 2_000_000


@pytest.mark.parametrize("w", [4, 10])
def test_g1_scalar_mul_by_constant_base(w):
    """g1.rs:309-368 with the generator as base: window tables (host Jacobian arithmetic), multiplexers, addition chain."""
    random.seed(w)
    for k in [random.randrange(1, T.R_ORDER)] + ([1, (1 << 200) + 5] if w == 10 else []):
        ob, gc, _ = o.execute("g1_scalar_mul:%d" % w, o.int_to_bits(k, 254))
        assert affine(from_m(ints_of(ob))) == T.g1_mul(k), k


def test_projective_to_affine():
    random.seed(9)
    a = T.g1_mul(424242)
    ob, _, _ = o.execute("g1_to_affine", bits_of(to_m(jac(a, random.randrange(1, P)))))
    assert from_m(ints_of(ob)) == [a[0], a[1], 1]


def test_square_roots():
    """Fq::sqrt_montgomery (fq.rs:290-299) and Fq2::sqrt_general_montgomery (fq2.rs:425-446): a root of the input, the one the
    mirror of the circuit's algorithm picks — both branches of the quadratic-non-residue test."""
    random.seed(12)
    v = random.randrange(1, P)
    ob, gc, _ = o.execute("fq_sqrt", bits_of(to_m([v * v % P])))
    r = from_m(ints_of(ob))[0]
    assert r in (v, P - v) and r == G.fq_sqrt_circuit(v * v % P)
    seen = set()
    for _ in range(6):
        w = (random.randrange(1, P), random.randrange(1, P))
        sq = T.f2_sq(w)
        alpha_sqrt = G.fq_sqrt_circuit((sq[0] * sq[0] + sq[1] * sq[1]) % P)
        branch = pow((alpha_sqrt + sq[0]) * T.HALF % P, (P - 1) // 2, P) == P - 1
        if branch in seen:
            continue
        seen.add(branch)
        ob, _, _ = o.execute("fq2_sqrt", bits_of(to_m(list(sq))))
        r = tuple(from_m(ints_of(ob)))
        assert r in (w, T.f2_neg(w)) and r == G.fq2_sqrt_circuit(sq)
    assert seen == {True, False}


def test_vk_blob_round_trip_and_instance_is_valid():
    inst = G.make_instance(n_pub=2, seed=3)
    assert G.check_instance(inst)  # the pairing equation holds in the Python mirror
    name = G.circuit_name(inst)
    assert o.circuit_info(name) == (2 * 254 + 762 + 1524 + 762, 1)
    bad = dict(inst, public=[inst["public"][0] + 1, inst["public"][1]])
    assert not G.check_instance(bad)


@pytest.mark.skipif(not os.environ.get("GSV_SLOW"), reason="11 B gates in execute mode: minutes; set GSV_SLOW=1")
def test_groth16_verify_accepts_valid_and_rejects_tampered():
    inst = G.make_instance(n_pub=2, seed=3)
    name = G.circuit_name(inst)
    ob, gc, peak = o.execute(name, G.input_bits(inst))
    assert ob.tolist() == [1]
    print("groth16_verify gates:", int(gc.sum()), "peak live wires:", peak)
    bad = dict(inst, public=[inst["public"][0] + 1, inst["public"][1]])
    ob, _, _ = o.execute(name, G.input_bits(bad))
    assert ob.tolist() == [0]


@pytest.mark.skipif(not os.environ.get("GSV_SLOW"), reason="11 B gates in execute mode: minutes; set GSV_SLOW=1")
def test_groth16_verify_compressed_accepts_valid_and_rejects_tampered():
    """groth16_verify_compressed (groth16.rs:250-268): decompression of A, B, C in front of the verifier — the reference's
    headline circuit (README.md:12 quotes 11,174,708,821 gates for its key; the count depends on the key's constants)."""
    inst = G.make_instance(n_pub=2, seed=3)
    name = G.compressed_circuit_name(inst)
    bits = G.compressed_input_bits(inst)
    assert o.circuit_info(name) == (bits.size, 1)
    ob, gc, peak = o.execute(name, bits)
    assert ob.tolist() == [1]
    print("groth16_verify_compressed gates:", int(gc.sum()), "peak live wires:", peak)
    bad = bits.copy()
    bad[2 * 254 + 254] ^= 1  # A's sign flag: the other root, another point
    ob, _, _ = o.execute(name, bad)
    assert ob.tolist() == [0]


@pytest.mark.skipif(not os.environ.get("GSV_SLOW"), reason="11 B gates in execute mode: minutes; set GSV_SLOW=1")
@pytest.mark.parametrize("n_pub,gates", [(0, 10_457_717_451), (1, 10_684_151_254)])
def test_groth16_verify_with_fewer_public_inputs(n_pub, gates):
    """No public input: the MSM is the constant identity (g1.rs:376-383) and only gamma_abc_g1[0] enters; one: a single window
    scalar multiplication (225 M gates)."""
    inst = G.make_instance(n_pub=n_pub, seed=5 + n_pub)
    assert G.check_instance(inst)
    ob, gc, _ = o.execute(G.circuit_name(inst), G.input_bits(inst))
    assert ob.tolist() == [1] and int(gc.sum()) == gates


def test_malformed_verifying_keys_are_rejected():
    """The key travels inside the circuit name: wrong length for its public-input count, non-hex text,
    an unreduced field element and a missing key are errors of the circuit factory (oracle and engine share it), not crashes."""
    import garbled_snark_verifier_amd as gsv
    good = G.vk_blob(G.make_instance(n_pub=1, seed=2))
    bad_len = good[:-32].hex()
    unreduced = (good[:1] + (P + 1).to_bytes(32, "big") + good[33:]).hex()
    for name in ["groth16_verify", "groth16_verify:" + bad_len, "groth16_verify:zz" + good.hex()[2:], "groth16_verify_compressed:" + unreduced]:
        with pytest.raises(Exception):
            o.circuit_info(name)
        with pytest.raises(gsv.GsvError):
            gsv.Plan.from_circuit(name, ["g1::add_montgomery"])
    assert o.circuit_info("groth16_verify_compressed:" + good.hex()) == (254 + 255 + 509 + 255, 1)
