"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Bar: bit-exact ciphertext stream, ciphertext CBC-MAC, output labels (garble) and active labels +
plaintext bits (evaluate).  Shapes mirror the reference's own tests:
  tests/streaming_evaluate.rs:136-213 (every gate type x all input combinations)
  tests/streaming_evaluate.rs:401-448 (Fq a^2*b + a)
  tests/fq12_mul_e2e.rs:175-236       (Fq12 mul garble -> evaluate, select(value) == active label)
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "garble_golden.json")


def _garble_and_check(gsv, engine, spec, seeds, replays=1, oracle_spec=None, program=None):
    program = program or gsv.Program.from_circuit(spec, chain_feedback=replays > 1)
    r = gsv.CircuitBuilder.streaming_garbling(spec, seeds, engine=engine, program=program, replays=replays)
    for i, seed in enumerate(seeds):
        ref = o.garble(oracle_spec or spec, seed)
        assert (ref.delta == r.delta[i]).all() and (ref.input_label0 == r.input_label0[i]).all()
        assert ref.n_ciphertexts == r.n_ciphertexts
        assert (ref.ciphertexts == r.ciphertexts[i]).all(), "ciphertext stream differs (seed %d)" % seed
        assert ref.ct_hash.tobytes() == r.ciphertext_hash[i]
        assert (ref.output_label0 == r.output_label0[i]).all(), "output labels differ (seed %d)" % seed
        assert [int(x) for x in ref.gate_counts] == r.gate_count
    return r, program


def _evaluate_and_check(gsv, engine, spec, g, program, seeds, replays=1, oracle_spec=None, bit_seed=123):
    n_in = program.info["n_inputs"]
    B = len(seeds)
    rng = np.random.default_rng(bit_seed)
    bits = rng.integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, g.input_label0 ^ g.delta[:, None, :], g.input_label0)
    true_active = g.true_label0 ^ g.delta   # evaluator holds true.label1 / false.label0 (fq12_mul_e2e.rs:196-197)
    false_active = g.false_label0
    e = gsv.CircuitBuilder.streaming_evaluation(spec, true_active, false_active, active, bits, g.ciphertexts, engine=engine, program=program, replays=replays)
    for i in range(B):
        ob, _, _ = o.execute(oracle_spec or spec, bits[i])
        assert (ob == e.output_bits[i]).all(), "plaintext bits differ"
        sel = np.where(e.output_bits[i][:, None] == 1, g.output_label0[i] ^ g.delta[i][None, :], g.output_label0[i])
        assert (sel == e.output_active[i]).all(), "gw.select(value) != active_label"
        assert e.ciphertext_hash[i] == g.ciphertext_hash[i]
        ref = o.evaluate(oracle_spec or spec, true_active[i].tobytes(), false_active[i].tobytes(), active[i], bits[i], g.ciphertexts[i])
        assert (ref.output_active == e.output_active[i]).all() and (ref.output_bits == e.output_bits[i]).all()
    return e


@pytest.mark.parametrize("t", range(11))
def test_every_gate_type_all_inputs(engine, t):
    import garbled_snark_verifier_amd as gsv
    spec = "gate:%d" % t
    g, prog = _garble_and_check(gsv, engine, spec, [42])
    assert g.n_ciphertexts == (1 if t < 8 else 0)  # garble_test.rs:82,129-133
    for a in (0, 1):
        for b in (0, 1):
            bits = np.array([[a, b]], np.uint8)
            active = np.where(bits[:, :, None] == 1, g.input_label0 ^ g.delta[:, None, :], g.input_label0)
            e = gsv.CircuitBuilder.streaming_evaluation(spec, g.true_label0 ^ g.delta, g.false_label0, active, bits, g.ciphertexts, engine=engine, program=prog)
            ob, _, _ = o.execute(spec, bits[0])
            assert (e.output_bits[0] == ob).all()
            sel = np.where(ob[:, None] == 1, g.output_label0[0] ^ g.delta[0][None, :], g.output_label0[0])
            assert (sel == e.output_active[0]).all()


@pytest.mark.parametrize("terms", ["2", "4"])
def test_both_and_record_forms_on_the_device(engine, terms, monkeypatch):
    """program.hpp pack_and / pack_and4: the same circuits compiled with at most two and with up to four wires per AND input (forced
    with GSV_AND_TERMS; left alone the compiler picks four for latency-bound programs) garble and evaluate to the oracle's stream —
    narrow steps (multi-lane AES), wide steps (one gate per lane: random circuits, Fq12 multiplication), chained replays, and with two
    and four instances per workgroup; the two forms run different instantiations of the kernel (FW = false / true)."""
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_AND_TERMS", terms)
    for spec, seeds in (("fq_mul", [1, 2]), ("random_circuit:5", [3]), ("driver_mix", [4, 5, 6]), ("fq12_mul", [7])):
        g, prog = _garble_and_check(gsv, engine, spec, seeds)
        _evaluate_and_check(gsv, engine, spec, g, prog, seeds)
    for ni in ("2", "4"):
        monkeypatch.setenv("GSV_INSTANCES_PER_WG", ni)
        seeds = list(range(20, 20 + 2 * int(ni) + 1))  # a ragged last workgroup
        g, prog = _garble_and_check(gsv, engine, "fq_mul", seeds)
        _evaluate_and_check(gsv, engine, "fq_mul", g, prog, seeds)
    monkeypatch.delenv("GSV_INSTANCES_PER_WG")


def test_driver_mix_dead_gates_constants(engine):
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "driver_mix", [5, 6, 7])
    assert prog.info["n_dead"] == 3
    _evaluate_and_check(gsv, engine, "driver_mix", g, prog, [5, 6, 7])


def test_u254_add_config1_shape(engine):
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "u254_add", [0])
    assert prog.info["n_gates"] == 1267 and g.n_ciphertexts == 254
    _evaluate_and_check(gsv, engine, "u254_add", g, prog, [0])


def test_fq_mul_config2(engine):
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "fq_mul", [0, 1, 2, 3])
    assert prog.info["n_gates"] == 414284
    _evaluate_and_check(gsv, engine, "fq_mul", g, prog, [0, 1, 2, 3])


def test_fq_complex_chain(engine):
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "fq_complex", [99])
    _evaluate_and_check(gsv, engine, "fq_complex", g, prog, [99])


def test_fq12_mul_config3_garble_evaluate(engine):
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "fq12_mul", [0, 7])
    assert prog.info["n_gates"] == 20284982
    _evaluate_and_check(gsv, engine, "fq12_mul", g, prog, [0, 7])


def test_fq12_chain_replay_matches_streamed_chain(engine):
    """K replays of the Fq12-mul program with output->input feedback == the reference-style streamed
    circuit r <- Fq12::mul(r, b) K times (gate ids and ciphertext stream continue across replays)."""
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "fq12_mul", [3], replays=2, oracle_spec="fq12_mul_chain:2")
    _evaluate_and_check(gsv, engine, "fq12_mul", g, prog, [3], replays=2, oracle_spec="fq12_mul_chain:2")


def test_fq12_square_and_square_multiply_chain(engine):
    """Fq12::square_montgomery (13.6 M gates) and the cyclotomic squaring garbled + evaluated on the GPU against the oracle, and
    the square-and-multiply link r <- mul(square(r), b) replayed twice with feedback == the streamed 2-link chain."""
    import garbled_snark_verifier_amd as gsv
    for spec, seeds in (("fq12_square", [2, 9]), ("fq12_cyclotomic_square", [3])):
        g, prog = _garble_and_check(gsv, engine, spec, seeds)
        _evaluate_and_check(gsv, engine, spec, g, prog, seeds)
    g, prog = _garble_and_check(gsv, engine, "fq12_sqmul", [4], replays=2, oracle_spec="fq12_sqmul_chain:2")
    _evaluate_and_check(gsv, engine, "fq12_sqmul", g, prog, [4], replays=2, oracle_spec="fq12_sqmul_chain:2")


def test_golden_fixtures(engine):
    """Committed fixtures (tests/golden/make_golden.py): ciphertext hash + digest of output labels."""
    import hashlib
    import garbled_snark_verifier_amd as gsv
    with open(GOLDEN) as f:
        golden = json.load(f)
    for case in golden["cases"]:
        if case["gates"] > 70_000_000:
            continue  # final_exp: test_final_exponentiation_as_a_plan
        prog = gsv.Program.from_circuit(case["program"], chain_feedback=case["replays"] > 1)
        r = gsv.CircuitBuilder.streaming_garbling(case["program"], [case["seed"]], engine=engine, program=prog, replays=case["replays"], keep_ciphertexts=False)
        assert r.ciphertext_hash[0].hex() == case["ct_hash"], case
        assert hashlib.sha256(r.output_label0[0].tobytes()).hexdigest() == case["output_label0_sha256"], case
        assert r.n_ciphertexts == case["n_ciphertexts"]


def test_properties_at_full_batch(engine):
    """Size-independent properties on a 16-instance batch (the cut-and-choose shape): distinct seeds give
    distinct streams; the same seed twice gives identical streams; garble -> evaluate round trip holds."""
    import garbled_snark_verifier_amd as gsv
    seeds = list(range(100, 115)) + [100]
    prog = gsv.Program.from_circuit("fq_mul")
    g = gsv.CircuitBuilder.streaming_garbling("fq_mul", seeds, engine=engine, program=prog)
    assert g.ciphertext_hash[0] == g.ciphertext_hash[15]
    assert len(set(g.ciphertext_hash[:15])) == 15
    _evaluate_and_check(gsv, engine, "fq_mul", g, prog, seeds)


def test_evaluate_with_short_ciphertext_stream_fails_like_reference(engine):
    """evaluate_mode.rs:139-142 panics "Ciphertext source exhausted at gate .."; the engine refuses the launch."""
    import garbled_snark_verifier_amd as gsv
    prog = gsv.Program.from_circuit("fq_add")
    g = gsv.CircuitBuilder.streaming_garbling("fq_add", [1], engine=engine, program=prog)
    bits = np.zeros((1, prog.info["n_inputs"]), np.uint8)
    with pytest.raises(gsv.GsvError, match="exhausted"):
        gsv.CircuitBuilder.streaming_evaluation("fq_add", g.true_label0 ^ g.delta, g.false_label0, g.input_label0, bits,
                                                [g.ciphertexts[0][:-1]], engine=engine, program=prog)


def test_empty_and_free_only_programs(engine):
    """Edge cases: a circuit of free gates only emits no ciphertext; hash of the empty stream is S::ZERO bytes."""
    import garbled_snark_verifier_amd as gsv
    g, prog = _garble_and_check(gsv, engine, "gate:8", [9])
    assert g.n_ciphertexts == 0 and g.ciphertext_hash[0] == bytes(16)
    prog2 = gsv.Program.from_gates(2, [], [2, 3, 0, 1])  # outputs = inputs and constants, no gates at all
    r = gsv.CircuitBuilder.streaming_garbling("", [4], engine=engine, program=prog2)
    d, f, t, inp = gsv.labels_from_seed(4, 2)
    assert (r.output_label0[0] == np.stack([inp[0], inp[1], f, t])).all()


def test_lockstep_determinism_stress(engine):
    """Race detector for the hand-counted step barrier: 128 instances with the SAME seed must produce 128 identical
    ciphertext hashes / output labels, equal to the oracle's, on every one of several launches (all CUs busy)."""
    import garbled_snark_verifier_amd as gsv
    prog = gsv.Program.from_circuit("fq12_mul", chain_feedback=True)
    B, R, seed = 128, 2, 11
    n_in = prog.info["n_inputs"]
    d, f, t, inp = gsv.labels_from_seed(seed, n_in)
    sess = gsv.Session(engine, prog, B, R, R)
    ref = o.garble("fq12_mul_chain:2", seed, capture_ct=False)
    for _ in range(3):
        sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
        sess.garble(0)
        sess.sync()
        out = sess.read_outputs()
        assert (out == ref.output_label0[None]).all(), "output labels differ between identical instances / from the oracle"
        for i in (0, 1, 63, 64, 127):
            assert sess.ciphertext_hash(i) == ref.ct_hash.tobytes()
    sess.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_circuits_differential(engine, seed):
    """Pseudo-random DAGs (all gate types, dead gates, constants, nested components) x 3 instances: garble and
    evaluate on the GPU against the oracle."""
    import garbled_snark_verifier_amd as gsv
    spec = "random_circuit:%d" % seed
    seeds = [seed, seed + 100, seed + 200]
    g, prog = _garble_and_check(gsv, engine, spec, seeds)
    assert prog.info["n_dead"] > 500
    _evaluate_and_check(gsv, engine, spec, g, prog, seeds, bit_seed=seed)


def test_cut_and_choose_fanout_and_commit_records(engine):
    """Garbler::create -> commit (cut_and_choose/garbler.rs:191-257) on the GPU: 16 instances of one compiled circuit in a
    single launch, commit record per instance = ciphertext CBC-MAC + AES_K(label) of constants and outputs
    (GarbledInstanceCommit, garbler.rs:63-99; AesLabelCommitHasher, cut_and_choose/mod.rs:41-48), compared with records built
    from the oracle's garbling of the same seeds."""
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding
    total = 16
    seeds = [int(x) for x in sharding.instance_seeds(1234, total)]
    prog = gsv.Program.from_circuit("fq_mul")
    g = gsv.CircuitBuilder.streaming_garbling("fq_mul", seeds, engine=engine, program=prog, keep_ciphertexts=False)
    recs = np.stack([sharding.commit_record(i, g.ciphertext_hash[i], g.output_label0[i], g.delta[i], g.false_label0[i], g.true_label0[i], g.input_label0[i]) for i in range(total)])
    import torch
    table = sharding.all_gather_records(torch.from_numpy(recs), total, 0, 1)
    assert table.shape == (total, sharding.record_len(254, 508))
    for i in (0, 7, 15):
        ref = o.garble("fq_mul", seeds[i], capture_ct=False)
        exp = sharding.commit_record(i, ref.ct_hash.tobytes(), ref.output_label0, ref.delta, ref.false_label0, ref.true_label0, ref.input_label0)
        assert (table[i].numpy() == exp).all()
    # AES_K(label): one-block CBC-MAC from the zero state; the record ends with commit(true.label1), commit(false.label0)
    assert bytes(table[0].numpy()[-16:]) == o.cbcmac(g.false_label0[0].tobytes())
    assert bytes(table[0].numpy()[-32:-16]) == o.cbcmac((g.true_label0[0] ^ g.delta[0]).tobytes())
    assert bytes(table[0].numpy()[24:40]) == o.cbcmac(g.input_label0[0][0].tobytes())


def test_cc16_cut_and_choose_commit_on_one_gpu(engine):
    """BASELINE config 5 at N = 1 on a shortened circuit: sharding.cut_and_choose_commit (what bench.py --workload cc16 runs) draws 16
    seeds from the master seed, garbles all 16 instances of the Fq12 multiplication on the GPU WITH their ciphertext commitments
    (stream drained + CBC-MAC'ed), builds the GarbledInstanceCommit records and "gathers" them: every record must equal the one
    the committed fixture holds (built from the CPU oracle's garblings, tests/golden/make_cc16_golden.py), and three of them are
    rebuilt from the oracle here."""
    import hashlib
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding
    gold = json.load(open(os.path.join(os.path.dirname(GOLDEN), "cc16_golden.json")))
    prog = gsv.Program.from_circuit(gold["circuit"])
    table, seeds = sharding.cut_and_choose_commit(gold["circuit"], gold["master_seed"], gold["total"], 0, 1, engine=engine, program=prog)
    assert [int(x) for x in seeds] == gold["seeds"] and table.shape == (gold["total"], gold["record_len"])
    assert [hashlib.sha256(r.tobytes()).hexdigest() for r in table] == gold["record_sha256"]
    assert hashlib.sha256(table.tobytes()).hexdigest() == gold["table_sha256"]
    for i in (0, 7, 15):
        g = o.garble(gold["circuit"], gold["seeds"][i], capture_ct=False)
        exp = sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0)
        assert (table[i] == exp).all()
    prog.close()


def test_two_instances_per_workgroup(engine, monkeypatch):
    """More instances than CUs: sessions switch to two instances per workgroup (each with half of the LDS label window,
    program variant compiled on demand).  Forced here on small batches, odd batch size included (idle second half),
    garble and evaluate, against the oracle; then a >256-instance batch where the engine picks it by itself, checked
    through lock-step determinism (identical seeds -> identical streams) and spot instances against the oracle."""
    import garbled_snark_verifier_amd as gsv
    for ni in ("2", "4"):  # four per workgroup: a quarter of the window each (run_program_kernel<*, 4, 0>); batches that leave groups idle
        monkeypatch.setenv("GSV_INSTANCES_PER_WG", ni)
        for spec, seeds in (("fq_mul", [5, 6, 7]), ("driver_mix", [1, 2]), ("fq_complex", [9, 10, 11, 12, 13])):
            g, prog = _garble_and_check(gsv, engine, spec, seeds)
            _evaluate_and_check(gsv, engine, spec, g, prog, seeds)
        g, prog = _garble_and_check(gsv, engine, "fq12_mul", [21, 22, 23], replays=2, oracle_spec="fq12_mul_chain:2")
        _evaluate_and_check(gsv, engine, "fq12_mul", g, prog, [21, 22, 23], replays=2, oracle_spec="fq12_mul_chain:2")
    monkeypatch.delenv("GSV_INSTANCES_PER_WG")
    total = 300  # > 256 CUs
    seeds = [1000 + (i % 50) for i in range(total)]
    prog = gsv.Program.from_circuit("fq_mul")
    r = gsv.CircuitBuilder.streaming_garbling("fq_mul", seeds, engine=engine, program=prog, keep_ciphertexts=False)
    for i in range(50, total):
        assert r.ciphertext_hash[i] == r.ciphertext_hash[i % 50] and (r.output_label0[i] == r.output_label0[i % 50]).all()
    for i in (0, 49, 299):
        ref = o.garble("fq_mul", seeds[i], capture_ct=False)
        assert ref.ct_hash.tobytes() == r.ciphertext_hash[i] and (ref.output_label0 == r.output_label0[i]).all()


def test_unfused_compilation_on_gpu(engine, monkeypatch):
    """The same kernels run a program compiled WITHOUT gate fusion (GSV_FUSE=0: every reference gate is its own record,
    absent operand fields name the zero slot): results must not depend on the compiler's folding decisions."""
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_FUSE", "0")
    for spec, seeds in (("fq_mul", [3, 4]), ("driver_mix", [8]), ("random_circuit:3", [5])):
        g, prog = _garble_and_check(gsv, engine, spec, seeds)
        assert prog.info["n_fused_free"] == sum(prog.info["gate_count"][8:]) - (prog.info["n_dead"] - (sum(prog.info["gate_count"][:8]) - prog.info["n_ciphertexts"]))
        _evaluate_and_check(gsv, engine, spec, g, prog, seeds)


def test_streaming_drain_hash_and_gc_files(engine, tmp_path):
    """gsv_session_garble_streaming: a 6-link Fq2-mul chain garbled with a device ring of only 2 replays, drained segment by
    segment into per-instance CBC-MACs and gc_<i>.bin files while the next segment is garbled.  The hashes and file bytes
    must equal those of a session that retains the whole stream (same engine), the output labels must agree, and a
    2-link Fq12 chain (ring 2 = two one-replay segments) must reproduce the oracle's hash."""
    import garbled_snark_verifier_amd as gsv
    prog = gsv.Program.from_circuit("fq2_mul", chain_feedback=True)
    seeds, K = [11, 12, 13], 6
    full = gsv.CircuitBuilder.streaming_garbling("fq2_mul", seeds, engine=engine, program=prog, replays=K)
    n_in = prog.info["n_inputs"]
    B = len(seeds)
    sess = gsv.Session(engine, prog, B, K, 2)
    sess.set_garble_inputs(full.delta, np.stack([full.false_label0, full.true_label0], 1), full.input_label0)
    hashes = sess.garble_streaming(directory=str(tmp_path), first_index=40, threads=2)
    assert hashes == list(full.ciphertext_hash)
    assert (sess.read_outputs() == full.output_label0).all()
    for i in range(B):
        data, file_hash = gsv.read_gc_file(os.path.join(str(tmp_path), gsv.gc_file_name(40 + i)))
        assert data.shape == (K * prog.info["n_ciphertexts"], 16) and (data == full.ciphertexts[i]).all() and file_hash == hashes[i]
    sess.close()
    # hash only, one thread per instance, against the oracle
    prog12 = gsv.Program.from_circuit("fq12_mul", chain_feedback=True)
    d, f, t, inp = gsv.labels_from_seed(77, prog12.info["n_inputs"])
    s2 = gsv.Session(engine, prog12, 1, 2, 2)
    s2.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    h2 = s2.garble_streaming()
    ref = o.garble("fq12_mul_chain:2", 77, capture_ct=False)
    assert h2[0] == ref.ct_hash.tobytes() and (s2.read_outputs()[0] == ref.output_label0).all()
    s2.close()


def test_cut_and_choose_regarbling_check(engine, tmp_path):
    """The consumer side of cut-and-choose (cut_and_choose/evaluator.rs:83-181): 8 instances are garbled and committed with
    their streams written to gc_<i>.bin; the evaluator keeps 3 for evaluation (file hash must match the commit) and regarbles
    the 5 opened ones from their seeds in one launch (whole commit record must match).  Honest run passes; a flipped byte in
    a kept file, a wrong seed and a missing seed are each caught with the reference's error."""
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding
    total = 8
    seeds = [int(x) for x in sharding.instance_seeds(99, total)]
    prog = gsv.Program.from_circuit("fq_mul")
    commits = sharding.garble_and_commit("fq_mul", seeds, list(range(total)), engine=engine, program=prog, gc_dir=str(tmp_path))
    # the commit records are the ones the oracle's garbling gives
    for i in (0, 5):
        ref = o.garble("fq_mul", seeds[i], capture_ct=False)
        assert (commits[i] == sharding.commit_record(i, ref.ct_hash.tobytes(), ref.output_label0, ref.delta, ref.false_label0, ref.true_label0, ref.input_label0)).all()
    keep = [1, 4, 6]
    opened = {i: seeds[i] for i in range(total) if i not in keep}
    ok, errors = sharding.run_regarbling(commits, keep, opened, "fq_mul", str(tmp_path), engine=engine, program=prog)
    assert ok and not errors
    # corrupted ciphertext file of a kept instance
    path = os.path.join(str(tmp_path), gsv.gc_file_name(4))
    raw = bytearray(open(path, "rb").read())
    raw[12345] ^= 1
    open(path, "wb").write(bytes(raw))
    bad = dict(opened)
    bad[2] = seeds[2] ^ 1   # garbler revealed a seed that does not reproduce its commit
    del bad[7]              # and withheld another
    ok, errors = sharding.run_regarbling(commits, keep, bad, "fq_mul", str(tmp_path), engine=engine, program=prog)
    assert not ok and errors == {4: "ciphertext corrupted", 2: "regarbling failed", 7: "failed to find seed"}


def test_plan_of_component_programs(engine):
    """Component-level programs: the plan [Fq12::square(r) -> t ; Fq12::mul(t, b) -> r'] — two separately compiled programs over
    one wire file, one kernel launch per call — must produce exactly the stream of the flat circuit mul(square(r), b):
    ciphertexts (gate order, read across the call boundary), CBC-MAC, output labels; evaluation of the same plan must
    decode to the oracle's bits."""
    import garbled_snark_verifier_amd as gsv
    # ---- Fq12 square then mul == fq12_sqmul
    sq, mul = gsv.Program.from_circuit("fq12_square"), gsv.Program.from_circuit("fq12_mul")
    N = 3048
    r_, b_, t_, o_ = np.arange(0, N), np.arange(N, 2 * N), np.arange(2 * N, 3 * N), np.arange(3 * N, 4 * N)
    plan = gsv.Plan()
    plan.add_call(sq, r_, t_)
    plan.add_call(mul, np.concatenate([t_, b_]), o_)
    plan.finish(2 * N, o_)
    assert plan.info["n_gates"] == sq.info["n_gates"] + mul.info["n_gates"] == 33_880_204
    seeds = [31, 32]
    B = len(seeds)
    labs = [gsv.labels_from_seed(s, 2 * N) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    sess = gsv.Session(engine, plan, B)
    sess.set_garble_inputs(delta, consts, inputs)
    sess.garble(0)
    out = sess.read_outputs()
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, size=(B, 2 * N)).astype(np.uint8)
    for i, seed in enumerate(seeds):
        ref = o.garble("fq12_sqmul", seed)
        assert ref.n_ciphertexts == plan.info["n_ciphertexts"]
        assert sess.ciphertext_hash(i) == ref.ct_hash.tobytes()
        assert (out[i] == ref.output_label0).all()
        lo = sq.info["n_ciphertexts"] - 1000  # a range that straddles the call boundary
        assert (sess.read_ciphertexts(i, lo, 2000) == ref.ciphertexts[lo:lo + 2000]).all()
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)
    sess.set_evaluate_inputs(ca, active, bits)
    sess.evaluate(0)
    oa, ob = sess.read_outputs(with_bits=True)
    for i in range(B):
        eb, _, _ = o.execute("fq12_sqmul", bits[i])
        assert (ob[i] == eb).all() and (oa[i] == np.where(ob[i][:, None] == 1, out[i] ^ delta[i][None, :], out[i])).all()
    sess.close()


def test_plan_built_from_circuit_with_units(engine):
    """Plan.from_circuit: the two-pass driver records `fq12_mix` (square; Fq6 add/sub glue; mul; mul with the square's output
    reused; only c0 of the last product is returned, so that call runs a second variant of Fq12::mul with half of its outputs
    dead) with the Fq12 components as calls.  Three unit programs + one glue program serve five calls; the stream, hash,
    output labels and evaluated bits must be those of the flat circuit (oracle)."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"])
    assert plan.info["n_calls"] == 4 and plan.info["n_inputs"] == 6096 and plan.info["n_outputs"] == 1524
    seeds = [41, 42, 43]
    B = len(seeds)
    n_in = plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    sess = gsv.Session(engine, plan, B)
    sess.set_garble_inputs(delta, consts, inputs)
    sess.garble(0)
    out = sess.read_outputs()
    for i, seed in enumerate(seeds):
        ref = o.garble("fq12_mix", seed, capture_ct=False)
        assert plan.info["n_gates"] == int(ref.gate_counts.sum()) and plan.info["n_ciphertexts"] == ref.n_ciphertexts
        assert sess.ciphertext_hash(i) == ref.ct_hash.tobytes()
        assert (out[i] == ref.output_label0).all()
    bits = np.random.default_rng(3).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    sess.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    sess.evaluate(0)
    oa, ob = sess.read_outputs(with_bits=True)
    for i in range(B):
        eb, _, _ = o.execute("fq12_mix", bits[i])
        assert (ob[i] == eb).all() and (oa[i] == np.where(ob[i][:, None] == 1, out[i] ^ delta[i][None, :], out[i])).all()
    sess.close()
    # the same plan without retaining the stream: one call block on the device, drained call by call while the next call runs
    st = gsv.Session(engine, plan, B, retain_stream=False)
    st.set_garble_inputs(delta, consts, inputs)
    hashes = st.garble_streaming(threads=2)
    assert (st.read_outputs() == out).all()
    for i, seed in enumerate(seeds):
        assert hashes[i] == o.garble("fq12_mix", seed, capture_ct=False).ct_hash.tobytes()
    with pytest.raises(gsv.GsvError):
        st.garble(0)  # a streaming-only session cannot keep the whole stream
    st.close()
    # small circuits with many calls, dead / constant / pass-through unit outputs (same cases as the host-interpreter test)
    for spec, units, seed in (("driver_mix", ["test::inner", "bigint::add"], 6), ("random_circuit:3", ["test::random_block"], 1)):
        p2 = gsv.Plan.from_circuit(spec, units)
        ref = o.garble(spec, seed)
        d, f, t, inp = gsv.labels_from_seed(seed, ref.n_in)
        s2 = gsv.Session(engine, p2, 1)
        s2.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
        s2.garble(0)
        assert s2.ciphertext_hash(0) == ref.ct_hash.tobytes() and (s2.read_outputs()[0] == ref.output_label0).all()
        assert (s2.read_ciphertexts(0, 0, ref.n_ciphertexts) == ref.ciphertexts).all()
        s2.close()


FINE_UNITS = ["fq2::mul_montgomery", "fq2::square_montgomery", "fp254::mul_by_constant_montgomery", "bigint::mul_karatsuba", "fp254::montgomery_reduce"]


@pytest.mark.parametrize("units", [["fq12::mul_montgomery", "fq12::square_montgomery"], FINE_UNITS])
def test_concurrent_calls_of_a_plan_match_the_flat_stream(engine, tmp_path, units):
    """Intra-instance width (schedule.hpp): independent calls of a plan run side by side in ONE launch (grid = instance groups x
    calls), each in a scratch region of its own; gate ids and ciphertext positions stay those of the stream order.  `fq12_mix` with the
    Fq12 components as units (a chain: nothing to overlap, the schedule degenerates to the stream order) and with Fq2-level units (15 independent Fq2 multiplications per Fq12
    multiplication): for every concurrency / window setting the stream's CBC-MAC, the output labels, the gc files and the evaluation
    must be those of the oracle's flat, sequential stream.  The calls of a window run as a dataflow inside one launch (every call
    waits for the completion flags of the calls it depends on: kernels.hip)."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq12_mix", units)
    seeds = [61, 62, 63]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s, capture_ct=False) for s in seeds]
    bits = np.random.default_rng(5).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    exp_bits = [o.execute("fq12_mix", bits[i])[0] for i in range(B)]
    seq_depth = None
    for conc in (1, 4, 64):
        sess = gsv.Session(engine, plan, B, concurrent_calls=conc)  # whole stream retained
        info = sess.schedule_info()
        assert info["n_calls"] == plan.info["n_calls"]
        if conc == 1:
            seq_depth = info["critical_steps"]
            assert info["max_width"] == 1 and info["critical_steps"] == info["total_steps"]
        elif units is FINE_UNITS:
            assert info["max_width"] >= 2 and info["critical_steps"] < seq_depth
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble(0)
        out = sess.read_outputs()
        for i in range(B):
            assert sess.ciphertext_hash(i) == refs[i].ct_hash.tobytes() and (out[i] == refs[i].output_label0).all()
        sess.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
        sess.evaluate(0)
        oa, ob = sess.read_outputs(with_bits=True)
        for i in range(B):
            assert (ob[i] == exp_bits[i]).all() and (oa[i] == np.where(ob[i][:, None] == 1, out[i] ^ delta[i][None, :], out[i])).all()
        sess.close()
    # the stream leaves the device window by window (gate order restored per call), here with windows of about two Fq2 multiplications
    # ... and, second setting, segment by segment of ONE running window (the host follows the completion flags of the launch)
    for conc, win, seg in ((16, 700_000, 0), (64, 0, 500_000)):
        st = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=conc, window_ct_records=win, drain_segment_records=seg)
        info = st.schedule_info()
        gc = os.path.join(str(tmp_path), "gc_%d_%d" % (conc, win))
        os.makedirs(gc)
        assert seg == 0 or (info["n_windows"] == 1 and info["n_segments"] >= 3)  # (Fq12-level units: a segment is at least one call)
        st.set_garble_inputs(delta, consts, inputs)
        hashes = st.garble_streaming(directory=gc, first_index=3, threads=2)
        out = st.read_outputs()
        for i in range(B):
            assert hashes[i] == refs[i].ct_hash.tobytes() and (out[i] == refs[i].output_label0).all()
        # window-aligned slices chain the MACs; a slice off a window boundary or out of sequence is refused
        wins = st.windows()
        assert len(wins) == info["n_windows"] and sum(w[1] for w in wins) == plan.info["n_calls"]
        if len(wins) >= 2:
            st.set_garble_inputs(delta, consts, inputs)
            for first, n, _ in wins:
                hashes = st.garble_calls(first, n)
            assert [h for h in hashes] == [r.ct_hash.tobytes() for r in refs]
            with pytest.raises(gsv.GsvError):
                st.garble_calls(wins[1][0], wins[1][1])  # does not continue the previous slice
            if wins[0][1] > 1:
                with pytest.raises(gsv.GsvError):
                    st.garble_calls(0, wins[0][1] - 1)  # not a window boundary
        # the evaluator reads the gc files back window by window
        es = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=conc, window_ct_records=win, drain_segment_records=seg)
        es.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
        fh = es.evaluate_streaming(gc, first_index=3)
        oa, ob = es.read_outputs(with_bits=True)
        for i in range(B):
            assert fh[i] == refs[i].ct_hash.tobytes() and (ob[i] == exp_bits[i]).all()
            assert (oa[i] == np.where(ob[i][:, None] == 1, out[i] ^ delta[i][None, :], out[i])).all()
        es.close()
        st.close()
    plan.close()


def test_dataflow_between_calls_with_several_instances_per_workgroup(engine, monkeypatch):
    """The completion flags of the dataflow are per INSTANCE GROUP (workgroup): with two and four instances per workgroup (lockstep groups
    sharing the step barrier) and several calls in flight per group the plan must still give the oracle's stream for every instance —
    including a batch that does not fill its last workgroup."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS, window_div=4)
    seeds = [71, 72, 73, 74, 75, 76, 77]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s, capture_ct=False) for s in seeds]
    for ni in ("2", "4"):
        monkeypatch.setenv("GSV_INSTANCES_PER_WG", ni)
        sess = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=8, window_ct_records=3_000_000)
        monkeypatch.delenv("GSV_INSTANCES_PER_WG")
        assert sess.instances_per_workgroup == int(ni) and sess.schedule_info()["max_width"] >= 2 and sess.schedule_info()["n_windows"] >= 2
        sess.set_garble_inputs(delta, consts, inputs)
        hashes = sess.garble_streaming(threads=2)
        out = sess.read_outputs()
        for i in range(B):
            assert hashes[i] == refs[i].ct_hash.tobytes() and (out[i] == refs[i].output_label0).all()
        sess.close()
    plan.close()


def test_final_exponentiation_as_a_plan(engine):
    """final_exponentiation_montgomery (final_exponentiation.rs:99-135): 3,519,328,217 gates, 31 % of the Groth16 verifier, far
    beyond a flat recording.  Recorded as a plan — Fq12 mul / square / cyclotomic square / inverse as units (each recorded
    once per output-liveness pattern), Frobenius maps, conjugations and the constant ONE as glue — and garbled for two seeds in
    one session without retaining the stream (each call's ciphertexts are drained and hashed while the next call runs).  The
    hash and the output labels must equal the committed fixture, which the CPU oracle produced from the FLAT stream
    (tests/golden/make_golden.py, 2.5 minutes of oracle time; execute-mode correctness of the same circuit against the
    reference's native formula: tests/test_gadgets_execute.py, slow)."""
    import hashlib
    import garbled_snark_verifier_amd as gsv
    with open(GOLDEN) as f:
        case = [c for c in json.load(f)["cases"] if c["circuit"] == "final_exp"][0]
    plan = gsv.Plan.from_circuit("final_exp", ["fq12::mul_montgomery", "fq12::square_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::inverse_montgomery"])
    assert plan.info["n_gates"] == case["gates"] == 3_519_328_217 and plan.info["n_ciphertexts"] == case["n_ciphertexts"]
    assert plan.info["n_calls"] > 250
    seeds = [case["seed"], case["seed"] + 1]
    n_in = plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    assert delta[0].tobytes().hex() == case["delta"]
    sess = gsv.Session(engine, plan, len(seeds), retain_stream=False)
    sess.set_garble_inputs(delta, consts, inputs)
    hashes = sess.garble_streaming(threads=2)
    out = sess.read_outputs()
    assert hashes[0].hex() == case["ct_hash"]
    assert hashlib.sha256(out[0].tobytes()).hexdigest() == case["output_label0_sha256"] and out[0][0].tobytes().hex() == case["first_output_label0"]
    assert hashes[1] != hashes[0]
    sess.close()


def test_evaluate_streaming_from_gc_files(engine, tmp_path):
    """The evaluator at scale (EvaluateMode over a FileSource, ciphertext_source.rs:36-107): ciphertexts are never resident as a
    whole — the garbler drains them to gc_<i>.bin (gsv_session_garble_streaming), the evaluator reads them back ring by ring /
    call by call (gsv_session_evaluate_streaming), hashing while reading.  Chain of 6 Fq2 muls with a 2-replay ring, and the
    fq12_mix plan with one call block on the device: decoded bits == the oracle's, active labels == select(label0, bit), file
    hashes == the garbler's commitments; a truncated file fails like an exhausted source."""
    import garbled_snark_verifier_amd as gsv
    # ---- program session: chain with a ring
    prog = gsv.Program.from_circuit("fq2_mul", chain_feedback=True)
    K, seeds = 6, [71, 72, 73]
    B, n_in = len(seeds), prog.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    d1 = tmp_path / "chain"; d1.mkdir()
    gs = gsv.Session(engine, prog, B, K, 2)
    gs.set_garble_inputs(delta, consts, inputs)
    commits = gs.garble_streaming(directory=str(d1), first_index=0, threads=2)
    out0 = gs.read_outputs()
    gs.close()
    bits = np.random.default_rng(4).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)
    es = gsv.Session(engine, prog, B, K, 2)
    es.set_evaluate_inputs(ca, active, bits)
    assert es.evaluate_streaming(str(d1)) == commits
    oa, ob = es.read_outputs(with_bits=True)
    assert (oa == np.where(ob[:, :, None] == 1, out0 ^ delta[:, None, :], out0)).all()
    # plaintext check against integer arithmetic: r <- r * b (Fq2, Montgomery) six times is what the chain computes
    full = gsv.CircuitBuilder.streaming_garbling("fq2_mul", seeds, engine=engine, program=prog, replays=K)
    assert (full.output_label0 == out0).all() and list(full.ciphertext_hash) == commits
    # a truncated file: the source runs dry
    path = os.path.join(str(d1), gsv.gc_file_name(1))
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[: len(raw) - 16 * 1000])
    es.set_evaluate_inputs(ca, active, bits)
    with pytest.raises(gsv.GsvError, match="exhausted"):
        es.evaluate_streaming(str(d1))
    es.close()
    # ---- plan session, one call block on the device
    plan = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"])
    seeds = [81, 82]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    d2 = tmp_path / "plan"; d2.mkdir()
    gs = gsv.Session(engine, plan, B, retain_stream=False)
    gs.set_garble_inputs(delta, consts, inputs)
    commits = gs.garble_streaming(directory=str(d2), first_index=10, threads=2)
    out0 = gs.read_outputs()
    gs.close()
    bits = np.random.default_rng(5).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    es = gsv.Session(engine, plan, B, retain_stream=False)
    es.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    assert es.evaluate_streaming(str(d2), first_index=10) == commits
    oa, ob = es.read_outputs(with_bits=True)
    for i, seed in enumerate(seeds):
        eb, _, _ = o.execute("fq12_mix", bits[i])
        assert (ob[i] == eb).all() and (oa[i] == np.where(ob[i][:, None] == 1, out0[i] ^ delta[i][None, :], out0[i])).all()
        assert commits[i] == o.garble("fq12_mix", seed, capture_ct=False).ct_hash.tobytes()
    es.close()


def test_generic_ciphertext_sink_and_source(engine):
    """The third CiphertextHandler / CiphertextSource of the reference — the channel Sender<S> / Receiver<S> pair (circuit/mod.rs:160-170,
    ciphertext_source.rs:14-34) — as host callbacks (gsv_session_garble_streaming_sink / gsv_session_evaluate_streaming_source): a
    Python handler receives every ciphertext of every instance in gate order, window by window (no stream retained on the device), and
    must end up with exactly the oracle's stream; a source feeds the evaluator from that memory; a source that runs dry fails like the
    reference's exhausted receiver; an exception in the handler aborts the pass and is re-raised.  Plan sessions (several windows,
    calls side by side) and a program session with a two-replay ring."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    seeds = [91, 92, 93]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s) for s in seeds]
    n_ct = plan.info["n_ciphertexts"]
    for threads in (1, 2):
        # threads = 1: windows of 900 k records, each drained whole; threads = 2: ONE window, drained in segments while it runs
        kw = dict(window_ct_records=900_000) if threads == 1 else dict(drain_segment_records=700_000)
        st = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=16, **kw)
        si = st.schedule_info()
        assert (si["n_windows"] > 4) if threads == 1 else (si["n_windows"] == 1 and si["n_segments"] > 8)
        st.set_garble_inputs(delta, consts, inputs)
        got = [np.zeros((n_ct, 16), np.uint8) for _ in range(B)]
        nxt = [0] * B

        def handler(inst, first, recs):
            assert first == nxt[inst], "runs of one instance must arrive in stream order, back to back"
            got[inst][first:first + recs.shape[0]] = recs
            nxt[inst] = first + recs.shape[0]

        hashes = st.garble_to_sink(handler, threads=threads, with_hashes=True)
        out = st.read_outputs()
        for i in range(B):
            assert nxt[i] == n_ct == refs[i].n_ciphertexts and (got[i] == refs[i].ciphertexts).all()
            assert hashes[i] == refs[i].ct_hash.tobytes() and (out[i] == refs[i].output_label0).all()
        st.close()
    # a failing handler aborts the pass with its own exception
    st = gsv.Session(engine, plan, B, retain_stream=False, window_ct_records=900_000)
    st.set_garble_inputs(delta, consts, inputs)

    def bad(inst, first, recs):
        raise KeyError("handler gave up")

    with pytest.raises(KeyError):
        st.garble_to_sink(bad, threads=1)
    st.close()
    # the evaluator pulls the same stream back through a source
    bits = np.random.default_rng(6).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    es = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=16, window_ct_records=900_000)
    ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)
    es.set_evaluate_inputs(ca, active, bits)
    fh = es.evaluate_from_source(lambda inst, first, n: got[inst][first:first + n])
    oa, ob = es.read_outputs(with_bits=True)
    for i in range(B):
        eb, _, _ = o.execute("fq12_mix", bits[i])
        assert fh[i] == refs[i].ct_hash.tobytes() and (ob[i] == eb).all()
        assert (oa[i] == np.where(ob[i][:, None] == 1, refs[i].output_label0 ^ delta[i][None, :], refs[i].output_label0)).all()
    es.set_evaluate_inputs(ca, active, bits)
    with pytest.raises(gsv.GsvError, match="exhausted"):
        es.evaluate_from_source(lambda inst, first, n: None if (inst == 1 and first + n > n_ct // 2) else got[inst][first:first + n])
    es.close()
    plan.close()
    # program session: a chain of 5 replays through a ring of two
    prog = gsv.Program.from_circuit("fq2_mul", chain_feedback=True)
    K, seed = 5, 94
    d, f, t, inp = gsv.labels_from_seed(seed, prog.info["n_inputs"])
    ps = gsv.Session(engine, prog, 1, K, 2)
    ps.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    chunks = []
    ps.garble_to_sink(lambda inst, first, recs: chunks.append((first, recs.copy())))
    full = gsv.CircuitBuilder.streaming_garbling("fq2_mul", [seed], engine=engine, program=prog, replays=K)
    stream = np.concatenate([c for _, c in chunks])
    assert [c[0] for c in chunks] == list(np.cumsum([0] + [c[1].shape[0] for c in chunks[:-1]])) and (stream == full.ciphertexts[0]).all()
    ps.close()


def test_ciphertext_ring_whole_pass_in_one_window(engine, tmp_path, monkeypatch):
    """retain_stream = GSV_STREAM_RING (or GSV_CT_RING=1): sessions that do not retain the stream run the WHOLE pass as one launch — the scope in which an instance's independent call
    chains overlap — over a ciphertext RING of a few drain segments (schedule.hpp, SchedParams::ring_ct): a garbling call waits on the
    device until what its block of the ring held on the previous lap has been gathered off the device, an evaluating call until its
    segment has been uploaded; the host publishes its stream position in mapped host memory.  fq12_mix with Fq2-level units (14 M
    ciphertexts) through a ring of ~1.6 M records (nine laps; sized by GSV_CT_RING_RECORDS here, by the free memory in production): the
    stream through a sink, the CBC-MACs, gc files, the discarding form, evaluation from the files and from a source — everything equal
    to the oracle's flat stream, twice over the same sessions."""
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_CT_RING_RECORDS", "1000000")
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    seeds = [101, 102, 103]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s) for s in seeds]
    n_ct = plan.info["n_ciphertexts"]
    kw = dict(retain_stream="ring", concurrent_calls=16, drain_segment_records=300_000)  # gsv_plan_session_opts.retain_stream = GSV_STREAM_RING
    st = gsv.Session(engine, plan, B, **kw)
    si = st.schedule_info()
    assert si["n_windows"] == 1 and si["ct_ring_records"] >= 1_000_000 and si["ct_ring_records"] * 5 < n_ct and si["n_segments"] > 20
    gc = str(tmp_path)
    for rep in range(2):
        st.set_garble_inputs(delta, consts, inputs)
        got = [np.zeros((n_ct, 16), np.uint8) for _ in range(B)]

        def handler(inst, first, recs):
            got[inst][first:first + recs.shape[0]] = recs

        try:
            hashes = st.garble_to_sink(handler, threads=2, with_hashes=True)
        except gsv.GsvError as e:
            # The ring's progress watchdog fired ONCE in round 5's runs of this test (profiles/r05_debug/): a forgotten Session finalized on the
            # sink thread, hipFree waiting for the pass that waited for that callback.  Since round 6 the engine defers such releases
            # (test_destroy_inside_a_sink_callback_is_deferred), so the status is a regression again: keep the diagnosis and FAIL.
            out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            if os.path.isdir(out_dir):
                open(os.path.join(out_dir, "ring_watchdog_event.txt"), "a").write(str(e) + "\n")
            pytest.fail("the ring pass failed: %s" % e)
        out = st.read_outputs()
        for i in range(B):
            assert (got[i] == refs[i].ciphertexts).all() and hashes[i] == refs[i].ct_hash.tobytes() and (out[i] == refs[i].output_label0).all()
    st.set_garble_inputs(delta, consts, inputs)
    assert st.garble_streaming(directory=gc, first_index=5, threads=2) == [r.ct_hash.tobytes() for r in refs]
    st.set_garble_inputs(delta, consts, inputs)
    st.garble_streaming(discard=True)  # nothing leaves the device: every block of the ring is free at once
    assert (st.read_outputs() == np.stack([r.output_label0 for r in refs])).all()
    # the evaluator: the window is launched first, its calls wait for their segments' uploads
    bits = np.random.default_rng(11).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)
    monkeypatch.setenv("GSV_CT_RING", "1")  # the other way to ask for it: the environment, for sessions created with retain_stream = 0
    es = gsv.Session(engine, plan, B, **dict(kw, retain_stream=False))
    monkeypatch.delenv("GSV_CT_RING")
    assert es.schedule_info()["ct_ring_records"] == si["ct_ring_records"]
    for how in ("files", "source", "files"):
        es.set_evaluate_inputs(ca, active, bits)
        fh = es.evaluate_streaming(gc, first_index=5) if how == "files" else es.evaluate_from_source(lambda inst, first, n: got[inst][first:first + n])
        oa, ob = es.read_outputs(with_bits=True)
        for i in range(B):
            eb, _, _ = o.execute("fq12_mix", bits[i])
            assert fh[i] == refs[i].ct_hash.tobytes() and (ob[i] == eb).all()
            assert (oa[i] == np.where(ob[i][:, None] == 1, refs[i].output_label0 ^ delta[i][None, :], refs[i].output_label0)).all()
    # a source that runs dry in the middle of the pass: the error comes back and the launch does not hang
    es.set_evaluate_inputs(ca, active, bits)
    with pytest.raises(gsv.GsvError, match="exhausted"):
        es.evaluate_from_source(lambda inst, first, n: None if first + n > n_ct // 2 else got[inst][first:first + n])
    # a pair needs explicit windows
    with pytest.raises(gsv.GsvError, match="window"):
        st.garble_evaluate(es)
    es.close(); st.close()
    plan.close()


def test_ring_watchdog_names_a_stalled_host(engine, monkeypatch):
    """The ring's wait is bounded by a PROGRESS watchdog (kernels.hip: the host's position unchanged for GSV_DEP_WAIT_SECONDS): a sink that
    stops consuming stalls the host's drain, the device gives up instead of hanging, the pass fails with status GSV_ERR_DEVICE and the
    message says who waited for what and where the host's time went (engine.cpp, check_plan_error) — and the same session garbles the
    oracle's stream again afterwards."""
    import time
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_CT_RING_RECORDS", "1000000")
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    n_in, n_ct = plan.info["n_inputs"], plan.info["n_ciphertexts"]
    d, f, t, inp = gsv.labels_from_seed(101, n_in)
    ref = o.garble("fq12_mix", 101)
    st = gsv.Session(engine, plan, 1, retain_stream="ring", concurrent_calls=16, drain_segment_records=300_000)
    assert st.schedule_info()["n_segments"] > 20
    calls = {"n": 0}

    def sleepy(inst, first, recs):
        calls["n"] += 1
        if calls["n"] == 3:
            time.sleep(8.0)

    monkeypatch.setenv("GSV_DEP_WAIT_SECONDS", "2")
    st.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    with pytest.raises(gsv.GsvError, match=r"stand still.*gave up.*wanted position \d+.*host: longest interval between two positions [0-9.]+ s.*gate-order buffer"):
        st.garble_to_sink(sleepy, threads=1, with_hashes=True)
    monkeypatch.delenv("GSV_DEP_WAIT_SECONDS")
    st.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    got = np.zeros((n_ct, 16), np.uint8)

    def handler(inst, first, recs):
        got[first:first + recs.shape[0]] = recs

    hashes = st.garble_to_sink(handler, threads=1, with_hashes=True)
    assert hashes[0] == ref.ct_hash.tobytes() and (got == ref.ciphertexts).all() and (st.read_outputs()[0] == ref.output_label0).all()
    st.close(); plan.close()


def test_collector_is_paused_during_a_ring_pass(engine, monkeypatch):
    """The cause of round 5's one-off ring watchdog failure (tools/ring_gc_repro.py reproduces it at will): a Session finalized on a sink
    callback thread — by Python's cyclic collector, in the middle of a pass — runs gsv_session_destroy -> hipFree, which synchronises the
    device: it waits for the ring window, which waits for the host's position, which waits for that callback.  The Python layer pauses the
    collector for the duration of every streaming call: with a forgotten Session sitting in a reference cycle and a handler that allocates
    enough containers to trigger generation-0 collections, the pass runs through, and the garbage goes afterwards."""
    import gc
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_CT_RING_RECORDS", "1000000")
    monkeypatch.setenv("GSV_DEP_WAIT_SECONDS", "2")
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    d, f, t, inp = gsv.labels_from_seed(101, plan.info["n_inputs"])
    ref = o.garble("fq12_mix", 101)
    st = gsv.Session(engine, plan, 1, retain_stream="ring", concurrent_calls=16, drain_segment_records=300_000)
    small = gsv.Program.from_circuit("fq_add")

    class Holder:
        pass

    h = Holder(); h.me = h; h.session = gsv.Session(engine, small, 1, 1, 1)
    del h  # unreachable, alive until the collector runs
    seen = []

    def handler(inst, first, recs):
        seen.append(gc.isenabled())
        junk = [[i] for i in range(3000)]  # container allocations: what makes the collector run
        del junk

    st.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    assert gc.isenabled()
    hashes = st.garble_to_sink(handler, threads=1, with_hashes=True)
    assert len(seen) > 20 and not any(seen) and gc.isenabled()
    assert hashes[0] == ref.ct_hash.tobytes() and (st.read_outputs()[0] == ref.output_label0).all()
    gc.collect()  # outside any pass: the forgotten session is destroyed now
    st.close(); plan.close()


def test_destroy_inside_a_sink_callback_is_deferred(engine, monkeypatch):
    """The C ABI itself is safe now, not only the Python wrapper: gsv_session_destroy / gsv_plan_destroy /
    gsv_program_destroy called from a sink callback in the middle of a ring pass — explicitly (`close()`), by a refcount-triggered
    `__del__`, and by `gc.collect()` on a forgotten Session in a reference cycle, none of which the collector pause covers — return at
    once: the engine queues the release until the pass has ended (gsv_deferred_release_count).  tools/ring_gc_repro.py, which produced
    round 5's 60-second stall at will, is the same sequence.  The pass garbles the oracle's stream in its normal time."""
    import gc
    import time
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_CT_RING_RECORDS", "1000000")
    monkeypatch.setenv("GSV_DEP_WAIT_SECONDS", "20")
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    d, f, t, inp = gsv.labels_from_seed(101, plan.info["n_inputs"])
    ref = o.garble("fq12_mix", 101)
    st = gsv.Session(engine, plan, 1, retain_stream="ring", concurrent_calls=16, drain_segment_records=300_000)
    small = gsv.Program.from_circuit("fq_add")
    small2 = gsv.Program.from_circuit("fq_add")
    plan2 = gsv.Plan.from_circuit("fq_mul", ["bigint::mul_karatsuba"])

    class Holder:
        pass

    victims = {"explicit": gsv.Session(engine, small, 1, 1, 1), "refcount": gsv.Session(engine, small, 1, 1, 1), "plan_session": gsv.Session(engine, plan2, 1)}
    h = Holder(); h.me = h; h.session = gsv.Session(engine, small2, 1, 1, 1)
    del h
    before = gsv.lib().gsv_deferred_release_count()
    calls = {"n": 0, "max_s": 0.0}

    def handler(inst, first, recs):
        calls["n"] += 1
        t0 = time.perf_counter()
        if calls["n"] == 3:
            victims.pop("explicit").close()
        elif calls["n"] == 5:
            victims.pop("refcount")  # last reference: Session.__del__ -> gsv_session_destroy on this thread
        elif calls["n"] == 7:
            gc.collect()             # the forgotten session in the cycle (the collector pause only keeps AUTOMATIC collections away)
        elif calls["n"] == 9:
            victims.pop("plan_session").close(); plan2.close()
        calls["max_s"] = max(calls["max_s"], time.perf_counter() - t0)

    st.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    t0 = time.perf_counter()
    hashes = st.garble_to_sink(handler, threads=1, with_hashes=True)
    took = time.perf_counter() - t0
    assert calls["n"] > 20 and calls["max_s"] < 1.0 and took < 15.0, (calls, took)
    assert gsv.lib().gsv_deferred_release_count() - before >= 5  # three sessions + the cycle's + the plan (its programs ride along)
    assert hashes[0] == ref.ct_hash.tobytes() and (st.read_outputs()[0] == ref.output_label0).all()
    # the queue has run: the same engine garbles again, and new sessions get their memory
    st.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    assert st.garble_to_sink(lambda *a: None, threads=1, with_hashes=True)[0] == ref.ct_hash.tobytes()
    again = gsv.Session(engine, small, 1, 1, 1)
    again.close(); small.close(); small2.close(); st.close(); plan.close()


def test_forced_dependency_fault_falls_back_to_the_safe_schedule(engine, tmp_path, monkeypatch):
    """The schedule's one assumption (a workgroup only waits for workgroups the hardware dispatched before it) is not an architectural
    guarantee.  GSV_FAULT_WITHHOLD_DEP=1 points one dependency of the window at a completion flag nobody writes: on the device exactly
    what a violated assumption looks like.  The waiting call's progress watchdog gives up (GSV_DEP_WAIT_SECONDS), the pass ends with
    status 1 — and instead of failing (rounds 1-5) the engine installs the SAFE schedule into the same session (one call per launch, stream
    order, no device-side wait), re-stages the host's inputs and repeats the pass: MACs, gc files and output labels are the oracle's.
    A pass through a host callback cannot be repeated behind the host's back (it has seen an invalid prefix): it fails, names the
    remedy, and the host's repeat succeeds on the safe schedule."""
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_FAULT_WITHHOLD_DEP", "1")
    monkeypatch.setenv("GSV_DEP_WAIT_SECONDS", "0.5")  # (progress based: no call of the instance group completed for that long; fq12_mix's calls take milliseconds)
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    seeds = [201, 202]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s) for s in seeds]
    gc = str(tmp_path)
    for kw in (dict(retain_stream=False, concurrent_calls=16), dict(retain_stream="ring", concurrent_calls=16, drain_segment_records=300_000)):
        if kw["retain_stream"] == "ring":
            monkeypatch.setenv("GSV_CT_RING_RECORDS", "1000000")
        sess = gsv.Session(engine, plan, B, **kw)
        before = sess.schedule_info()
        assert before["n_windows"] < before["n_calls"] and sess.fallback_count() == 0
        sess.set_garble_inputs(delta, consts, inputs)
        hashes = sess.garble_streaming(directory=gc, first_index=40)
        after = sess.schedule_info()
        assert sess.fallback_count() == 1 and after["n_windows"] == after["n_calls"] and after["ct_ring_records"] == 0 and after["n_dependencies"] == 0
        assert hashes == [r.ct_hash.tobytes() for r in refs] and (sess.read_outputs() == np.stack([r.output_label0 for r in refs])).all()
        for i in range(B):
            assert open(os.path.join(gc, gsv.gc_file_name(40 + i)), "rb").read() == refs[i].ciphertexts.tobytes()
        # the safe schedule stays: further passes (also the discarding form) run on it without another fault
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble_streaming(discard=True)
        assert sess.fallback_count() == 1 and (sess.read_outputs() == np.stack([r.output_label0 for r in refs])).all()
        sess.close()
    # through a sink: the failing pass is reported, the host repeats it
    sess = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=16)
    sess.set_garble_inputs(delta, consts, inputs)
    with pytest.raises(gsv.GsvError, match="safe schedule"):
        sess.garble_to_sink(lambda *a: None, with_hashes=True)
    assert sess.fallback_count() == 1
    sess.set_garble_inputs(delta, consts, inputs)
    assert sess.garble_to_sink(lambda *a: None, with_hashes=True) == [r.ct_hash.tobytes() for r in refs]
    sess.close()
    # the evaluator: from the gc files the engine repeats the pass by itself, from a host source the host does
    bits = np.random.default_rng(3).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)

    def check_eval(es, fh):
        oa, ob = es.read_outputs(with_bits=True)
        for i in range(B):
            eb, _, _ = o.execute("fq12_mix", bits[i])
            assert fh[i] == refs[i].ct_hash.tobytes() and (ob[i] == eb).all()
            assert (oa[i] == np.where(ob[i][:, None] == 1, refs[i].output_label0 ^ delta[i][None, :], refs[i].output_label0)).all()

    es = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=16)
    es.set_evaluate_inputs(ca, active, bits)
    check_eval(es, es.evaluate_streaming(gc, first_index=40))
    assert es.fallback_count() == 1
    es.close()
    es = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=16)
    es.set_evaluate_inputs(ca, active, bits)
    src = lambda inst, first, n: refs[inst].ciphertexts[first:first + n]  # noqa: E731
    with pytest.raises(gsv.GsvError, match="safe schedule"):
        es.evaluate_from_source(src)
    es.set_evaluate_inputs(ca, active, bits)
    check_eval(es, es.evaluate_from_source(src))
    es.close()
    # a garble || evaluate pair: both sessions are switched, the host repeats the pair
    kw = dict(retain_stream=False, concurrent_calls=16, window_ct_records=4_000_000)
    gs, es = gsv.Session(engine, plan, B, **kw), gsv.Session(engine, plan, B, **kw)
    gs.set_garble_inputs(delta, consts, inputs); es.set_evaluate_inputs(ca, active, bits)
    with pytest.raises(gsv.GsvError, match="safe schedule"):
        gs.garble_evaluate(es, with_hashes=True)
    assert gs.fallback_count() == 1 and es.fallback_count() == 1
    gs.set_garble_inputs(delta, consts, inputs); es.set_evaluate_inputs(ca, active, bits)
    check_eval(es, gs.garble_evaluate(es, with_hashes=True))
    gs.close(); es.close()
    plan.close()


def test_garble_and_evaluate_side_by_side_on_the_device(engine):
    """examples/groth16_garble.rs:171-230 / tests/garbler_evaluator_connection.rs:64-172: the garbler feeds the evaluator while it
    garbles.  gsv_session_garble_evaluate: window k of the garbler's device block is evaluated on a second stream while window k+1 is
    garbled into the other of two blocks; nothing is retained, nothing crosses PCIe unless the commitment is asked for.  Outputs,
    decoded bits and (when asked for) the CBC-MACs must be the oracle's; sessions with different schedules are refused."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq12_mix", FINE_UNITS)
    seeds = [95, 96, 97, 98, 99]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s, capture_ct=False) for s in seeds]
    bits = np.random.default_rng(7).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    ca = np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1)
    for conc, win, with_hashes in ((16, 600_000, True), (1, 0, False), (64, 2_000_000, True)):
        kw = dict(retain_stream=False, concurrent_calls=conc, window_ct_records=win, drain_segment_records=300_000)
        gs, es = gsv.Session(engine, plan, B, **kw), gsv.Session(engine, plan, B, **kw)
        si = gs.schedule_info()
        assert si["n_windows"] >= 3 and (si["n_segments"] > si["n_windows"] or conc == 1)  # (sequential sessions: one call per window, one segment each)
        for _ in range(2):  # a second pass over the same sessions
            gs.set_garble_inputs(delta, consts, inputs)
            es.set_evaluate_inputs(ca, active, bits)
            hashes = gs.garble_evaluate(es, with_hashes=with_hashes)
            out0 = gs.read_outputs()
            oa, ob = es.read_outputs(with_bits=True)
            for i in range(B):
                eb, _, _ = o.execute("fq12_mix", bits[i])
                assert (out0[i] == refs[i].output_label0).all() and (ob[i] == eb).all()
                assert (oa[i] == np.where(ob[i][:, None] == 1, out0[i] ^ delta[i][None, :], out0[i])).all()
                assert not with_hashes or hashes[i] == refs[i].ct_hash.tobytes()
        other = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=conc, window_ct_records=win + 1_000_000)
        with pytest.raises(gsv.GsvError, match="schedule"):
            gs.garble_evaluate(other)
        for x in (gs, es, other):
            x.close()
    plan.close()


def test_batched_evaluate_from_finalized_instances(engine, tmp_path):
    """Evaluator::evaluate_from is `into_par_iter` over the finalized cases (cut_and_choose/evaluator.rs:354-475): here ALL of them are
    evaluated in ONE session — instance k of the batch reads gc_<index_k>.bin (gsv_session_evaluate_streaming_indexed) — and every
    consistency check still fires with the reference's variant, in the order a case-by-case run would report them.  16 instances
    garbled and committed, 8 finalized (an arbitrary subset), a program session and a plan session."""
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding
    total, keep = 16, [1, 2, 5, 7, 8, 11, 14, 15]
    for circuit, prog in (("fq_mul", gsv.Program.from_circuit("fq_mul")), ("fq12_mix", gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"]))):
        gc = os.path.join(str(tmp_path), circuit); os.makedirs(gc)
        seeds = [int(x) for x in sharding.instance_seeds(77, total)]
        n_in, n_out = prog.info["n_inputs"], prog.info["n_outputs"]
        commits = sharding.garble_and_commit(circuit, seeds, list(range(total)), engine=engine, program=prog, gc_dir=gc)
        rng = np.random.default_rng(9)

        def case(i, **over):
            d, f, t, inp = gsv.labels_from_seed(seeds[i], n_in)
            bits = rng.integers(0, 2, n_in).astype(np.uint8)
            c = {"index": i, "true_constant_wire": t ^ d, "false_constant_wire": f, "input_active": np.where(bits[:, None] == 1, inp ^ d[None, :], inp), "input_bits": bits}
            c.update(over)
            return c

        cases = [case(i) for i in keep]
        res = sharding.evaluate_from(commits, cases, circuit, gc, n_out, engine=engine, program=prog)
        assert [r[0] for r in res] == keep
        for c, (i, act, ob) in zip(cases, res):
            eb, _, _ = o.execute(circuit, c["input_bits"])
            assert (ob == eb).all()
        # tampering: the first failing case in list order decides, whatever comes later
        d5 = gsv.labels_from_seed(seeds[5], n_in)
        bad_in = case(5); bad_in["input_active"] = bad_in["input_active"].copy(); bad_in["input_active"][3] ^= d5[0]
        with pytest.raises(sharding.ConsistencyError) as ei:
            sharding.evaluate_from(commits, [case(1), bad_in, case(7, true_constant_wire=np.zeros(16, np.uint8))], circuit, gc, n_out, engine=engine, program=prog)
        assert (ei.value.kind, ei.value.index) == ("InputLabelsMismatch", 5)
        with pytest.raises(sharding.ConsistencyError) as ei:
            sharding.evaluate_from(commits, [case(1), case(7, true_constant_wire=np.zeros(16, np.uint8)), bad_in], circuit, gc, n_out, engine=engine, program=prog)
        assert (ei.value.kind, ei.value.index) == ("TrueConstantMismatch", 7)
        path = os.path.join(gc, gsv.gc_file_name(8))
        raw = bytearray(open(path, "rb").read()); raw[4321] ^= 4; open(path, "wb").write(bytes(raw))
        with pytest.raises(sharding.ConsistencyError) as ei:
            sharding.evaluate_from(commits, [case(i) for i in keep], circuit, gc, n_out, engine=engine, program=prog)
        assert (ei.value.kind, ei.value.index) in (("CiphertextMismatch", 8), ("OutputLabelMismatch", 8))
        prog.close()


def test_drain_instances_checks_a_sample_of_a_batch(engine, tmp_path, monkeypatch):
    """gsv_session_set_drain_instances: a streaming pass garbles EVERY instance but only the first n instances' ciphertext streams leave the
    device — how bench.py checks the ciphertexts of its timed 1 024-instance, four-per-workgroup configuration (8 x 47.7 GB over PCIe
    instead of 1 024 x).  Five instances of fq12_mix, four per workgroup, two drained: their MACs and gc files equal the oracle's flat
    streams, the other three report no MAC and write no file, every instance's output labels equal the oracle's; the sink form sees
    instances 0 and 1 only.  Raising the sample after the gate-order buffers exist re-allocates them (round 6; round 5 refused it and
    — ADVICE r05 — let a later evaluate_streaming, which uploads EVERY instance's stream, write past the sample-sized buffers): four
    instances drained after two, then all five streams evaluated from their gc files on the same kind of session."""
    import garbled_snark_verifier_amd as gsv
    monkeypatch.setenv("GSV_INSTANCES_PER_WG", "4")
    plan = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], window_div=4)
    seeds = [71, 72, 73, 74, 75]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s, capture_ct=False) for s in seeds]
    sess = gsv.Session(engine, plan, B, retain_stream=False, window_ct_records=6_000_000)
    assert sess.instances_per_workgroup == 4
    sess.set_drain_instances(2)
    sess.set_garble_inputs(delta, consts, inputs)
    gc = str(tmp_path)
    hashes = sess.garble_streaming(directory=gc)
    out = sess.read_outputs()
    for i in range(B):
        assert (out[i] == refs[i].output_label0).all()
        if i < 2:
            assert hashes[i] == refs[i].ct_hash.tobytes() and os.path.getsize(os.path.join(gc, gsv.gc_file_name(i))) == 16 * refs[i].n_ciphertexts
        else:
            assert hashes[i] == bytes(16) and not os.path.exists(os.path.join(gc, gsv.gc_file_name(i)))
    seen = {}

    def handler(inst, first, recs):
        assert first == seen.get(inst, 0)
        seen[inst] = first + len(recs)

    sess.set_garble_inputs(delta, consts, inputs)
    h2 = sess.garble_to_sink(handler, with_hashes=True)
    assert seen == {0: refs[0].n_ciphertexts, 1: refs[1].n_ciphertexts} and h2[:2] == hashes[:2]
    sess.set_drain_instances(4)  # the gate-order buffers were sized for two: the next streaming call re-allocates them
    sess.set_garble_inputs(delta, consts, inputs)
    h4 = sess.garble_streaming()
    assert h4[:4] == [r.ct_hash.tobytes() for r in refs[:4]] and h4[4] == bytes(16)
    sess.set_drain_instances(1)
    sess.set_garble_inputs(delta, consts, inputs)
    assert sess.garble_streaming()[0] == refs[0].ct_hash.tobytes()
    # a sample drain, then an evaluation on the SAME session: the evaluator's uploads cover all five instances
    sess.set_drain_instances(0)
    sess.set_garble_inputs(delta, consts, inputs)
    assert sess.garble_streaming(directory=gc) == [r.ct_hash.tobytes() for r in refs]
    sess.set_drain_instances(2)
    sess.set_garble_inputs(delta, consts, inputs)
    assert sess.garble_streaming()[:2] == hashes[:2]  # the gate-order buffers are sized for two again ...
    sess.close()
    es = gsv.Session(engine, plan, B, retain_stream=False, window_ct_records=6_000_000)
    es.set_drain_instances(2)
    es.set_garble_inputs(delta, consts, inputs)
    assert es.garble_streaming()[:2] == hashes[:2]  # ... and on this session too, right before it evaluates
    bits = np.random.default_rng(5).integers(0, 2, size=(B, n_in)).astype(np.uint8)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    es.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    fh = es.evaluate_streaming(gc)
    oa, ob = es.read_outputs(with_bits=True)
    for i in range(B):
        eb, _, _ = o.execute("fq12_mix", bits[i])
        assert fh[i] == refs[i].ct_hash.tobytes() and (ob[i] == eb).all()
        assert (oa[i] == np.where(ob[i][:, None] == 1, refs[i].output_label0 ^ delta[i][None, :], refs[i].output_label0)).all()
    es.close()
    plan.close()


def test_plan_recorder_through_the_c_abi(engine):
    """gsv_plan_recorder_*: the plan builder driven the way a host with its own two-pass driver would.
    The gates of Fq::add are pushed one by one as glue (taken from a recording of the component), Fq::mul_montgomery is a
    call of a program the 'host' compiled itself, and the finished plan must garble to the stream of the flat circuit
    (a + b) * b.  Then the Fq12 pair square -> mul without any glue."""
    import ctypes as C
    import garbled_snark_verifier_amd as gsv
    import hostsim_lib as h
    L = h.lib()
    L.hostsim_trace.argtypes = [C.c_char_p, C.c_uint64] + [C.c_void_p] * 4 + [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
    cap = 10_000
    ty = np.zeros(cap, np.uint8); ta = np.zeros(cap, np.uint32); tb = np.zeros(cap, np.uint32); tc = np.zeros(cap, np.uint32)
    n, nw = C.c_uint64(), C.c_uint32()
    in_ssa, out_ssa = np.zeros(508, np.uint32), np.zeros(254, np.uint32)
    assert L.hostsim_trace(b"fq_add", cap, ty.ctypes.data, ta.ctypes.data, tb.ctypes.data, tc.ctypes.data, C.byref(n), C.byref(nw), in_ssa.ctypes.data, out_ssa.ctypes.data) == 0
    rec = gsv.PlanRecorder()
    a_w = [rec.input_wire() for _ in range(254)]
    b_w = [rec.input_wire() for _ in range(254)]
    wire = {0: 0, 1: 1}
    for i, s in enumerate(in_ssa):
        wire[int(s)] = (a_w + b_w)[i]
    gates = []
    for i in range(n.value):
        c = None
        if tc[i] != 0xFFFFFFFF:
            c = rec.allocate_wire(1)
        gates.append((int(ty[i]), wire[int(ta[i])], wire[int(tb[i])], c))
        if c is not None:
            wire[int(tc[i])] = c
    rec.push_gates(gates)
    mul = gsv.Program.from_circuit("fq_mul")
    m_w = rec.call(mul, [wire[int(s)] for s in out_ssa] + b_w)
    plan = rec.finish(m_w)
    assert plan.info["n_calls"] == 2
    seed = 17
    ref = o.garble("fq_addmul", seed)
    assert plan.info["n_gates"] == int(ref.gate_counts.sum())
    d, f, t, inp = gsv.labels_from_seed(seed, 508)
    s = gsv.Session(engine, plan, 1)
    s.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    s.garble(0)
    assert s.ciphertext_hash(0) == ref.ct_hash.tobytes() and (s.read_outputs()[0] == ref.output_label0).all()
    assert (s.read_ciphertexts(0, 0, ref.n_ciphertexts) == ref.ciphertexts).all()
    s.close()
    # two external units, no glue
    sq, mul12 = gsv.Program.from_circuit("fq12_square"), gsv.Program.from_circuit("fq12_mul")
    rec = gsv.PlanRecorder()
    r_w = [rec.input_wire() for _ in range(3048)]
    b_w = [rec.input_wire() for _ in range(3048)]
    plan = rec.finish(rec.call(mul12, rec.call(sq, r_w) + b_w))
    ref = o.garble("fq12_sqmul", 18, capture_ct=False)
    d, f, t, inp = gsv.labels_from_seed(18, 6096)
    s = gsv.Session(engine, plan, 1)
    s.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    s.garble(0)
    assert s.ciphertext_hash(0) == ref.ct_hash.tobytes() and (s.read_outputs()[0] == ref.output_label0).all()
    s.close()


@pytest.mark.slow
def test_miller_loop_as_a_plan(engine):
    """multi_miller_loop_groth16_evaluate_montgomery_fast (pairing.rs:944-1007): 6,909,061,143 gates, 62 % of the verifier, as a
    plan; hash + output labels == the oracle's flat-stream fixture (tests/golden/miller_loop_golden.json, 8 minutes of oracle
    time).  The plan is built for the half LDS window (one compilation per program, traces dropped: ~1-2 minutes of host time for
    its 178 constant-specialised line programs).  tools/miller_plan.py also measures the device rate."""
    import hashlib
    import garbled_snark_verifier_amd as gsv
    case = json.load(open(os.path.join(os.path.dirname(GOLDEN), "miller_loop_golden.json")))
    plan = gsv.Plan.from_circuit("miller_loop", ["fq12::square_montgomery", "fq12::mul_by_034_montgomery", "pairing::ell_by_constant_montgomery",
                                                 "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery"], half_window=True)
    assert plan.info["n_gates"] == case["gates"] == 6_909_061_143 and plan.info["n_ciphertexts"] == case["n_ciphertexts"]
    d, f, t, inp = gsv.labels_from_seed(case["seed"], plan.info["n_inputs"])
    sess = gsv.Session(engine, plan, 1, retain_stream=False)
    sess.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    hashes = sess.garble_streaming()
    assert hashes[0].hex() == case["ct_hash"] and hashlib.sha256(sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
    sess.close()
    plan.close()


def test_half_window_plan_serves_both_layouts(engine, monkeypatch):
    """GSV_PLAN_WINDOW_DIV / Plan.from_circuit(window_div=4): every program of the plan is compiled once, for a quarter of the LDS
    window, and the same image runs with one, two and four instances per workgroup; hashes and labels == the oracle's.  (A
    half-window plan, window_div=2, cannot serve four: the session falls back to two.)"""
    import garbled_snark_verifier_amd as gsv
    plan2 = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], half_window=True)
    monkeypatch.setenv("GSV_INSTANCES_PER_WG", "4")
    sess = gsv.Session(engine, plan2, 5)
    assert sess.instances_per_workgroup == 2
    sess.close(); plan2.close()
    plan = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], window_div=4)
    seeds = [91, 92, 93, 94, 95]
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s, capture_ct=False) for s in seeds]
    for ni in ("1", "2", "4"):
        monkeypatch.setenv("GSV_INSTANCES_PER_WG", ni)
        sess = gsv.Session(engine, plan, B)
        assert sess.instances_per_workgroup == int(ni)
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble(0)
        out = sess.read_outputs()
        for i, ref in enumerate(refs):
            assert sess.ciphertext_hash(i) == ref.ct_hash.tobytes() and (out[i] == ref.output_label0).all()
        sess.close()


def test_msm_plan_with_constant_tables(engine):
    """The verifier's window scalar multiplication (g1.rs:309-368) as a plan: the multiplexer units take the MSM's constant
    tables as inputs (call operands that are the constant wires), G1 additions are the other unit.  Window 10 as in the verifier:
    225,290,965 gates in 103 calls of 3 programs; hash / labels == the CPU oracle's flat stream; the evaluator's bits == k * G."""
    import garbled_snark_verifier_amd as gsv
    spec = "g1_scalar_mul:10"
    plan = gsv.Plan.from_circuit(spec, ["bigint::multiplexer", "g1::add_montgomery"])
    seeds = [31, 32]
    ref = o.garble(spec, seeds[0], capture_ct=False)
    assert plan.info["n_gates"] == int(ref.gate_counts.sum()) == 225_290_965 and plan.info["n_ciphertexts"] == ref.n_ciphertexts
    B, n_in = len(seeds), plan.info["n_inputs"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    sess = gsv.Session(engine, plan, B)
    sess.set_garble_inputs(delta, consts, inputs)
    sess.garble(0)
    out = sess.read_outputs()
    assert sess.ciphertext_hash(0) == ref.ct_hash.tobytes() and (out[0] == ref.output_label0).all()
    k = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF
    bits = np.stack([o.int_to_bits(k, 254), o.int_to_bits(k + 1, 254)])
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    sess.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    sess.evaluate(0)
    oa, ob = sess.read_outputs(with_bits=True)
    assert (oa == np.where(ob[:, :, None] == 1, out ^ delta[:, None, :], out)).all()
    import bn254_ref as T
    for i, kk in enumerate([k, k + 1]):
        x, y, z = [o.bits_to_int(ob[i][j * 254:(j + 1) * 254]) * T.RINV % T.P for j in range(3)]
        zi = pow(z, -1, T.P)
        assert (x * zi * zi % T.P, y * zi * zi * zi % T.P) == T.g1_mul(kk)
    sess.close()


VERIFIER_UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::mul_by_034_montgomery",
                  "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery",
                  "bigint::multiplexer", "g1::add_montgomery",
                  # the Fq inversions (binary extended Euclid, fp254impl.rs:333-690) enter as their own 4-iteration components: as ONE unit an
                  # inversion (11 M ciphertexts) — or the Fq12 inversion around it (21 M) — would set the size of every instance's device
                  # ciphertext block (340 MB x 512 instances); their chunks keep the largest block at an Fq12 multiplication's 5.4 M records
                  "inverse::iteration_group", "inverse::divide_chains"]


@pytest.fixture(scope="module")
def compressed_verifier_plan(engine, verifier_plan_file):
    """The plan of the reference's headline circuit, groth16_verify_compressed (groth16.rs:250-268), for the tests below: the session's
    plan file (conftest.verifier_plan_file: gsv_plan_build_file — warm-up recorders beside the driver, every program written to the file by
    the worker that compiled it, one image per program for a quarter of the LDS window — the way bench.py gets it on a fresh machine),
    streamed into the GPU's memory by gsv_plan_load (41.8 GB of records, a few seconds)."""
    import garbled_snark_verifier_amd as gsv
    from conftest import VERIFIER_PLAN_UNITS
    assert VERIFIER_UNITS + ["fp254::exp_chunk"] == VERIFIER_PLAN_UNITS
    # ONE public input: the reference's own benchmark configuration (examples/groth16_garble.rs:107-110, groth16_cut_and_choose.rs:116-119)
    plan = gsv.Plan.load(verifier_plan_file["path"], engine)
    yield verifier_plan_file["case"], plan
    plan.close()


def test_compressed_verifier_as_a_plan_in_slices(engine, compressed_verifier_plan):
    """BASELINE config 4.  groth16_verify_compressed (groth16.rs:250-268: decompression of A, B, C; MSM; projective-to-affine;
    Miller loop; final exponentiation; comparison) for the synthetic instance of tests/groth16_ref.py as one plan, garbled the way
    bench.py steps through it — in slices of consecutive calls, the stream drained and CBC-MAC'ed slice by slice: the final MAC
    and the output label == the fixture the CPU oracle produced from the FLAT 11,456,865,898-gate stream (one public input)
    (tests/golden/make_big_golden.py).  tools/groth16_plan.py --compressed also evaluates a valid and a tampered proof."""
    import hashlib
    import bench
    import garbled_snark_verifier_amd as gsv
    case, plan = compressed_verifier_plan
    assert plan.info["n_gates"] == case["gates"] == 11_456_865_898 and plan.info["n_ciphertexts"] == case["n_ciphertexts"] == 2_980_165_547
    ci = plan.call_info()
    d, f, t, inp = gsv.labels_from_seed(case["seed"], plan.info["n_inputs"])
    # one instance: independent calls run side by side (schedule.hpp), windows of at most 64 M ciphertexts (1 GB) leave the device
    sess = gsv.Session(engine, plan, 1, retain_stream=False, window_ct_records=64 << 20)
    slices = bench.session_slices(sess.windows(), ci[:, 1], 10)
    assert sum(s[2] for s in slices) == case["gates"] and sum(s[1] for s in slices) == plan.info["n_calls"] and max(s[2] for s in slices) < 1.2 * case["gates"] / 10
    sess.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    for first, n, _ in slices:
        hashes = sess.garble_calls(first, n)
    assert hashes[0].hex() == case["ct_hash"] and hashlib.sha256(sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
    sess.close()


def test_verifier_lockstep_two_instances_per_workgroup(engine, compressed_verifier_plan):
    """The bench's configuration on real verifier programs: 1024 instances, four per workgroup (the kernel instantiation
    run_program_kernel<false, 4, 0>, groups in lockstep on the LDS-only step barrier), then 512 instances, two per workgroup
    (<false, 2, 0>), all with the SAME seed, through the first slice of the plan with the ciphertexts drained: every instance
    must produce the same CBC-MAC state, a second run the same again, and both layouts the same as one instance alone."""
    import bench
    import garbled_snark_verifier_amd as gsv
    case, plan = compressed_verifier_plan
    ci = plan.call_info()
    d, f, t, inp = gsv.labels_from_seed(7, plan.info["n_inputs"])
    seen = []
    first = n = None
    for B, ni in ((1024, 4), (512, 2)):
        sess = gsv.Session(engine, plan, B, retain_stream=False, concurrent_calls=1)  # sequential schedule: the same windows for every batch size
        assert sess.instances_per_workgroup == ni
        sl = bench.session_slices(sess.windows(), ci[:, 1], 400)[0]  # decompression ladders: thousands of narrow steps (~7.5 M ciphertexts per instance: 1 024 x 120 MB over PCIe per run)
        assert (first, n) in ((None, None), sl[:2])
        first, n = sl[:2]
        for _ in range(2):
            sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
            hashes = sess.garble_calls(first, n)
            assert len(set(hashes)) == 1, "identical instances produced different ciphertext streams"
            seen.append(hashes[0])
        sess.close()
    assert len(set(seen)) == 1
    # the same slice for ONE instance (one per workgroup, run_program_kernel<false, 1, 0>) gives the same stream
    one = gsv.Session(engine, plan, 1, retain_stream=False, concurrent_calls=1)
    one.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    assert one.garble_calls(first, n)[0] == seen[0]
    one.close()
    # ... and so does one instance with the independent calls of the slice side by side (the three ladders of B's square root, A's and
    # C's ladders: whatever the window holds), in a window that is exactly this slice
    par = gsv.Session(engine, plan, 1, retain_stream=False, concurrent_calls=64, window_ct_records=int(ci[first:first + n, 3].sum()), max_window_calls=n)
    assert par.windows()[0][:2] == (first, n)
    par.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    assert par.garble_calls(first, n)[0] == seen[0]
    par.close()


@pytest.mark.slow  # (round 6: opt-in, `-m "gpu and slow"` — a whole pass at four instances per workgroup takes ~95 s however few workgroups run it (a workgroup's time is
                   #  the pass); bench.py checks exactly this on its timed 1 024-instance session in every run (`headline_ciphertext_check`: 8 of 8 MACs), the default set keeps
                   #  four-per-workgroup sessions on the verifier's first slice (test_verifier_lockstep_two_instances_per_workgroup) and on fq12_mix (test_drain_instances_...))
def test_verifier_whole_pass_four_per_workgroup_ciphertexts(engine, compressed_verifier_plan, monkeypatch):
    """The bench's TIMED kernel configuration over a WHOLE pass with its ciphertexts checked: the verifier at four instances per workgroup
    (run_program_kernel<false, 4, 0, *>, the default schedule's windows), every instance garbled, the streams of instances 0..7 — all four
    workgroup positions, two workgroups — drained and CBC-MAC'ed (gsv_session_set_drain_instances).  Instance 0 carries the
    single-instance fixture's seed, instances 1..7 seeds of the cut-and-choose fixture: MAC over all 2 980 165 547 ciphertexts and the
    output label of each == the CPU oracle's flat stream (examples/groth16_garble.rs:255-263 compares exactly this hash).  The other
    instances carry the fixture's seed as well: their output labels must equal instance 0's.  262 instances (66 workgroups, the last one
    RAGGED: two of its four groups idle through every barrier) since round 6 — bench.py itself checks the 1 024-instance batch after its
    timed steps in every run (`headline_ciphertext_check`), and the suite has to fit the driver's time limit (113 s -> ~40 s)."""
    import garbled_snark_verifier_amd as gsv
    case, plan = compressed_verifier_plan
    gold = json.load(open(os.path.join(os.path.dirname(GOLDEN), "cc16_verifier_golden.json")))
    assert gold["gates"] == case["gates"] == plan.info["n_gates"]
    monkeypatch.setenv("GSV_INSTANCES_PER_WG", "4")
    B, n_in = 262, plan.info["n_inputs"]
    seeds = [case["seed"]] + [int(x) for x in gold["seeds"][:7]] + [case["seed"]] * (B - 8)
    labs = {sd: gsv.labels_from_seed(sd, n_in) for sd in set(seeds)}
    delta = np.stack([labs[sd][0] for sd in seeds]); consts = np.stack([np.stack([labs[sd][1], labs[sd][2]]) for sd in seeds]); inputs = np.stack([labs[sd][3] for sd in seeds])
    sess = gsv.Session(engine, plan, B, retain_stream=False)
    assert sess.instances_per_workgroup == 4
    sess.set_drain_instances(8)
    sess.set_garble_inputs(delta, consts, inputs)
    hashes = sess.garble_streaming()
    out = sess.read_outputs()
    assert hashes[0].hex() == case["ct_hash"] and out[0][0].tobytes().hex() == case["first_output_label0"]
    for k in range(7):
        assert hashes[1 + k].hex() == gold["ct_hashes"][k] and out[1 + k][0].tobytes().hex() == gold["first_output_label0"][k], "instance %d (workgroup position %d)" % (1 + k, (1 + k) % 4)
    assert (out[8:] == out[0][None]).all()
    sess.close()


@pytest.mark.slow  # (round 6: opt-in, `-m "gpu and slow"` — the valid / tampered evaluation at full size stays in the default set as test_verifier_garble_evaluate_at_full_size,
                   #  which evaluates the same two proofs WHILE they are garbled; gsv_session_evaluate over a retained stream is covered at component size by every _evaluate_and_check)
def test_compressed_verifier_evaluates_valid_and_tampered_proof(engine, compressed_verifier_plan):
    """BASELINE config 4, evaluator side (EvaluateMode over the whole circuit, evaluate_mode.rs:123-158): two instances of the
    verifier are garbled with their 49 GB ciphertext streams retained in HBM and evaluated — one with the valid proof's input bits,
    one with A's sign flag flipped (another point: the proof must no longer verify).  The plaintext output bits are (1, 0), every
    active output label is select(label0, bit), and the retained streams' CBC-MAC equals the oracle's flat-stream fixture."""
    import garbled_snark_verifier_amd as gsv
    case, plan = compressed_verifier_plan
    n_in = plan.info["n_inputs"]
    seeds = [case["seed"], case["seed"] + 1]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    bits_ok = np.unpackbits(np.frombuffer(bytes.fromhex(case["input_bits_hex"]), np.uint8), bitorder="little")[:n_in].astype(np.uint8)
    bits_bad = bits_ok.copy(); bits_bad[case["tamper_bit"]] ^= 1
    bits = np.stack([bits_ok, bits_bad])
    sess = gsv.Session(engine, plan, 2)  # whole stream retained
    sess.set_garble_inputs(delta, consts, inputs)
    sess.garble(0); sess.sync()
    out0 = sess.read_outputs()
    assert out0[0][0].tobytes().hex() == case["first_output_label0"]  # (the whole stream's CBC-MAC is checked by the slices test and by the garble || evaluate test below)
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    sess.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    sess.evaluate(0); sess.sync()
    oa, ob = sess.read_outputs(with_bits=True)
    assert case["expected_output"] == 1 and ob[:, 0].tolist() == [1, 0]
    assert (oa == np.where(ob[:, :, None] == 1, out0 ^ delta[:, None, :], out0)).all()
    sess.close()


def test_cc16_verifier_full_size_on_one_gpu(engine, compressed_verifier_plan):
    """BASELINE config 5 at its REAL size on the hardware that exists: sharding.cut_and_choose_commit — what `bench.py --workload cc16`
    and the 8-GPU run execute per rank — garbles the 16 instances of master seed 1234 on the FULL one-public-input verifier
    (11 456 865 898 gates each, 183 B gates) on one GPU, every instance WITH its ciphertext commitment, and gathers the
    GarbledInstanceCommit records: all 16 must equal the records the CPU oracle built from 16 flat garblings of the same seeds
    (tests/golden/cc16_verifier_golden.json, tests/golden/make_cc16_verifier_golden.py)."""
    import hashlib
    from garbled_snark_verifier_amd import sharding
    case, plan = compressed_verifier_plan
    gold = json.load(open(os.path.join(os.path.dirname(GOLDEN), "cc16_verifier_golden.json")))
    assert gold["gates"] == case["gates"] and gold["n_ciphertexts"] == case["n_ciphertexts"]
    table, seeds = sharding.cut_and_choose_commit(case["circuit"], gold["master_seed"], gold["total"], 0, 1, engine=engine, program=plan)
    assert [int(x) for x in seeds] == gold["seeds"] and table.shape == (gold["total"], gold["record_len"])
    assert [bytes(r[8:24]).hex() for r in table] == gold["ct_hashes"]
    assert [hashlib.sha256(r.tobytes()).hexdigest() for r in table] == gold["record_sha256"]
    assert hashlib.sha256(table.tobytes()).hexdigest() == gold["table_sha256"]


def test_verifier_garble_evaluate_at_full_size(engine, compressed_verifier_plan):
    """The second phase of the reference's benchmark at verifier size (examples/groth16_garble.rs:171-230): two instances are garbled
    and — window by window, from the garbler's device block, nothing retained (retain_stream = 0) — evaluated at the same time, one with
    the valid proof's bits, one with A's sign flag flipped: decoded outputs (1, 0), active output labels = select(label0, bit), and
    the garbler's commitment of instance 0 == the oracle's flat-stream fixture.  (The generic sink at this size — every record of the
    48 GB stream through a host callback in gate order — is tests/test_ext_host.py's external host since round 5; at component size:
    test_generic_ciphertext_sink_and_source.)"""
    import garbled_snark_verifier_amd as gsv
    case, plan = compressed_verifier_plan
    n_in = plan.info["n_inputs"]
    seeds = [case["seed"], case["seed"] + 1]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    bits_ok = np.unpackbits(np.frombuffer(bytes.fromhex(case["input_bits_hex"]), np.uint8), bitorder="little")[:n_in].astype(np.uint8)
    bits_bad = bits_ok.copy(); bits_bad[case["tamper_bit"]] ^= 1
    bits = np.stack([bits_ok, bits_bad])
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    kw = dict(retain_stream=False, window_ct_records=1 << 28)  # 4 GB windows: the evaluation of window k runs beside the garbling of window k+1
    gs, es = gsv.Session(engine, plan, 2, **kw), gsv.Session(engine, plan, 2, **kw)
    si = gs.schedule_info()
    assert 8 <= si["n_windows"] <= 16 and si["n_segments"] >= 40 and si["segment_ct_records"] <= 1 << 26  # launches of <= 4 GB, drained in segments of <= 1 GB per instance
    gs.set_garble_inputs(delta, consts, inputs)
    es.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    hashes = gs.garble_evaluate(es, with_hashes=True)
    out0 = gs.read_outputs()
    oa, ob = es.read_outputs(with_bits=True)
    assert hashes[0].hex() == case["ct_hash"] and hashes[1] != hashes[0]
    assert out0[0][0].tobytes().hex() == case["first_output_label0"]
    assert case["expected_output"] == 1 and ob[:, 0].tolist() == [1, 0]
    assert (oa == np.where(ob[:, :, None] == 1, out0 ^ delta[:, None, :], out0)).all()
    es.close()
    gs.close()


def test_plan_slices_and_plan_file_on_the_device(engine, tmp_path):
    """gsv_session_garble_streaming_calls + gsv_plan_save / gsv_plan_load: a plan garbled slice by slice (MACs chained, gc files
    appended) equals the whole pass and the oracle; the same plan loaded from its file straight into device memory (no host
    records) gives the same stream again, with one and with two instances per workgroup."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], half_window=True)
    seeds = [91, 92, 93]
    B, n_in, n_calls = len(seeds), plan.info["n_inputs"], plan.info["n_calls"]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    refs = [o.garble("fq12_mix", s, capture_ct=True) for s in seeds]
    path = os.path.join(str(tmp_path), "mix.gsvplan")
    plan.save(path)
    loaded = gsv.Plan.load(path, engine)
    assert loaded.info == plan.info and loaded.image_bytes() == plan.image_bytes()
    with pytest.raises(gsv.GsvError):
        loaded.save(os.path.join(str(tmp_path), "again.gsvplan"))  # no host records to write
    for pl, ni in ((plan, "1"), (loaded, "1"), (loaded, "2")):
        os.environ["GSV_INSTANCES_PER_WG"] = ni
        try:
            sess = gsv.Session(engine, pl, B, retain_stream=False, concurrent_calls=1)
        finally:
            del os.environ["GSV_INSTANCES_PER_WG"]
        assert sess.instances_per_workgroup == int(ni)
        gc = os.path.join(str(tmp_path), "gc_%s_%s" % (ni, pl is loaded))
        os.makedirs(gc)
        sess.set_garble_inputs(delta, consts, inputs)
        wins = sess.windows()
        assert len(wins) >= 2
        for first, n, _ in wins:  # one slice per window of the schedule
            hashes = sess.garble_calls(first, n, directory=gc, first_index=5)
        out = sess.read_outputs()
        for i, ref in enumerate(refs):
            assert hashes[i] == ref.ct_hash.tobytes() and (out[i] == ref.output_label0).all()
            cts, h = gsv.read_gc_file(os.path.join(gc, gsv.gc_file_name(5 + i)))
            assert h == ref.ct_hash.tobytes() and (cts == ref.ciphertexts).all()
        # discarding slices leave the same output labels
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble_calls(0, wins[0][1], discard=True)
        sess.garble_calls(wins[1][0], n_calls - wins[1][0], discard=True)
        assert (sess.read_outputs() == out).all()
        with pytest.raises(gsv.GsvError):
            sess.garble_calls(n_calls, 1, discard=True)
        sess.close()
    loaded.close()
    plan.close()


@pytest.mark.slow
@pytest.mark.parametrize("fixture,units,gates", [
    ("groth16_verify_golden.json", VERIFIER_UNITS, 10_914_485_653),
])
def test_groth16_verifier_as_a_plan(engine, fixture, units, gates):
    """groth16_verify (groth16.rs:58-110: MSM, projective-to-affine, Miller loop, final exponentiation, comparison; uncompressed
    A, B, C) for the synthetic instance of tests/groth16_ref.py as a plan of its own: hash + output label == the oracle's
    flat-stream fixture (tests/golden/make_big_golden.py).  (The compressed circuit — the reference's headline — is covered by the
    tests above; tools/groth16_plan.py runs both end to end and measures the device rate.)"""
    import hashlib
    import garbled_snark_verifier_amd as gsv
    case = json.load(open(os.path.join(os.path.dirname(GOLDEN), fixture)))
    plan = gsv.Plan.from_circuit(case["circuit"], units, half_window=True)
    assert plan.info["n_gates"] == case["gates"] == gates and plan.info["n_ciphertexts"] == case["n_ciphertexts"]
    d, f, t, inp = gsv.labels_from_seed(case["seed"], plan.info["n_inputs"])
    sess = gsv.Session(engine, plan, 1, retain_stream=False)
    sess.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
    hashes = sess.garble_streaming()
    assert hashes[0].hex() == case["ct_hash"] and hashlib.sha256(sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
    sess.close()
    plan.close()
