"""Blake3Hasher path (reference: src/hashers/mod.rs:22-51 — the crate's DefaultHasher and the PRF of most of its own tests).

blake3 1.8.2 is a Cargo.lock dependency that is not vendored under /root/reference; both restatements (oracle, byte oriented;
engine, unrolled word form in gate_math.hpp) are pinned by the official BLAKE3 test vectors (test_vectors.json of the BLAKE3
repository: input byte i = i mod 251) for the single-block lengths, and against each other."""
import os

import numpy as np
import pytest

import hostsim_lib as h
import oracle_lib as o

OFFICIAL = {
    0: "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
    1: "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
    2: "7b7015bb92cf0b318037702a6cdd81dee41224f734684c2c122cd6359cb1ee63",
    3: "e1be4d7a8ab5560aa4199eea339849ba8e293d55ca0a81006726d184519e647f",
    4: "f30f5ab28fe047904037f77b6da4fea1e27241c5d132638d8bedce9d40494f32",
    5: "b40b44dfd97e7a84a996a91af8b85188c66c126940ba7aad2e7ae6b385402aa2",
    6: "06c4e8ffb6872fad96f9aaca5eee1553eb62aed0ad7198cef42e87f6a616c844",
    7: "3f8770f387faad08faa9d8414e9f449ac68e6ff0417f673f602a646a891419fe",
    8: "2351207d04fc16ade43ccab08600939c7c1fa70a5c0aaca76063d04c3228eaeb",
    63: "e9bc37a594daad83be9470df7f7b3798297c3d834ce80ba85d6e207627b7db7b",
    64: "4eed7141ea4a5cd4b788606bd23f46e212af9cacebacdc7d1f4c6dc7f2511b98",
}


@pytest.fixture(autouse=True)
def _restore_hashers():
    yield
    o.set_hasher("aes")
    h.set_hasher("aes")


def test_official_single_block_vectors():
    for n, hx in OFFICIAL.items():
        assert o.blake3_short(bytes(i % 251 for i in range(n))).hex() == hx, n


def test_hash_with_gate_is_blake3_of_label_and_le_gate_id_and_matches_engine_math():
    for _ in range(32):
        label = os.urandom(16)
        gid = int.from_bytes(os.urandom(8), "little")
        ref = o.blake3_short(label + gid.to_bytes(8, "little"))[:16]
        assert o.blake3_hash_with_gate(label, gid) == ref
        assert h.blake3_hash(label, gid) == ref


@pytest.mark.parametrize("t", range(8))
def test_halfgates_with_blake3(t):
    """halfgates_garbling.rs:81-157 runs every AND-variant with Blake3Hasher too."""
    o.set_hasher("blake3")
    rng = np.random.default_rng(t)
    delta, a0, b0 = (rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for _ in range(3))
    x = lambda p, q: bytes(i ^ j for i, j in zip(p, q))
    c0, ct = o.garble_gate(t, a0, b0, delta, 0)
    aa, ab, ac = (t >> 2) & 1, (t >> 1) & 1, t & 1
    for va in (0, 1):
        for vb in (0, 1):
            got = o.degarble_gate(t, ct, x(a0, delta) if va else a0, va, x(b0, delta) if vb else b0, 0)
            assert got == (x(c0, delta) if ((va ^ aa) & (vb ^ ab)) ^ ac else c0)


@pytest.mark.parametrize("spec,seed", [("driver_mix", 5), ("fq_mul", 0), ("random_circuit:3", 3)])
def test_compiled_schedule_with_blake3(spec, seed):
    o.set_hasher("blake3")
    h.set_hasher("blake3")
    sp = h.SimProgram(spec)
    n_in = sp.info["n_inputs"]
    labs = h.labels_from_seed(seed, 3 + n_in)
    out, cts = sp.garble(labs[0], labs[1:3], labs[3:])
    ref = o.garble(spec, seed)
    assert (ref.ciphertexts == cts).all() and (ref.output_label0 == out).all()
    o.set_hasher("aes")
    assert (o.garble(spec, seed).ct_hash != ref.ct_hash).any()  # the PRF really changed the stream


@pytest.mark.gpu
@pytest.mark.parametrize("spec,seeds", [("gate:0", [42]), ("gate:7", [1]), ("driver_mix", [5, 6]), ("fq_mul", [0, 1]), ("random_circuit:2", [2, 3, 4])])
def test_gpu_blake3_garble_evaluate(engine, spec, seeds):
    import garbled_snark_verifier_amd as gsv
    o.set_hasher("blake3")
    prog = gsv.Program.from_circuit(spec)
    g = gsv.CircuitBuilder.streaming_garbling(spec, seeds, engine=engine, program=prog, hasher="blake3")
    n_in = prog.info["n_inputs"]
    bits = np.random.default_rng(1).integers(0, 2, size=(len(seeds), n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, g.input_label0 ^ g.delta[:, None, :], g.input_label0)
    e = gsv.CircuitBuilder.streaming_evaluation(spec, g.true_label0 ^ g.delta, g.false_label0, active, bits, g.ciphertexts, engine=engine, program=prog,
                                                hasher="blake3")
    for i, s in enumerate(seeds):
        ref = o.garble(spec, s)
        assert (ref.ciphertexts == g.ciphertexts[i]).all() and (ref.output_label0 == g.output_label0[i]).all()
        assert ref.ct_hash.tobytes() == g.ciphertext_hash[i]
        ob, _, _ = o.execute(spec, bits[i])
        assert (ob == e.output_bits[i]).all()
        assert (e.output_active[i] == np.where(ob[:, None] == 1, g.output_label0[i] ^ g.delta[i][None, :], g.output_label0[i])).all()


@pytest.mark.gpu
def test_gpu_fq12_mul_e2e_blake3(engine):
    """tests/fq12_mul_e2e.rs:175-236 literally: Blake3Hasher, SEED = 0, garble -> evaluate, gw.select(value) == active_label."""
    import garbled_snark_verifier_amd as gsv
    o.set_hasher("blake3")
    prog = gsv.Program.from_circuit("fq12_mul")
    g = gsv.CircuitBuilder.streaming_garbling("fq12_mul", [0], engine=engine, program=prog, hasher="blake3")
    ref = o.garble("fq12_mul", 0, capture_ct=False)
    assert ref.ct_hash.tobytes() == g.ciphertext_hash[0] and (ref.output_label0 == g.output_label0[0]).all()
    bits = np.random.default_rng(0).integers(0, 2, size=(1, prog.info["n_inputs"])).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, g.input_label0 ^ g.delta[:, None, :], g.input_label0)
    e = gsv.CircuitBuilder.streaming_evaluation("fq12_mul", g.true_label0 ^ g.delta, g.false_label0, active, bits, g.ciphertexts, engine=engine,
                                                program=prog, hasher="blake3")
    sel = np.where(e.output_bits[0][:, None] == 1, g.output_label0[0] ^ g.delta[0][None, :], g.output_label0[0])
    assert (sel == e.output_active[0]).all()
