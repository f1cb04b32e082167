"""world_size-2 gloo test of the N>1 path: cut-and-choose instances are sharded across ranks with no
data-path collective; the one exchange is an all-gather of the per-instance commit records
(ciphertext hash + output label commits), mirroring Garbler::create -> commit
(src/cut_and_choose/garbler.rs:191-257, cut_and_choose/mod.rs:41-48).  Runs on CPU: the per-rank
"garbling" is done by the CPU oracle here, because this test is about the sharding / gather logic of
garbled_snark_verifier_amd.sharding, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garbled_snark_verifier_amd import sharding
    import garbled_snark_verifier_amd as gsv
    seeds = sharding.instance_seeds(1234, total)
    mine = sharding.shard_instances(total, rank, world)
    recs = []
    for i in mine:
        g = o.garble("fq_add", int(seeds[i]))
        recs.append(sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0))
    local = torch.from_numpy(np.stack(recs)) if recs else torch.zeros((0, sharding.record_len(254, 508)), dtype=torch.uint8)
    allrec = sharding.all_gather_records(local, total, rank, world, device="cpu")
    q.put((rank, allrec.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_commits_world2():
    world, total = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (res[0] == res[1]).all()
    sys.path.insert(0, ROOT)
    from garbled_snark_verifier_amd import sharding
    seeds = sharding.instance_seeds(1234, total)
    assert len(set(int(s) for s in seeds)) == total
    for i in range(total):
        g = o.garble("fq_add", int(seeds[i]))
        exp = sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0)
        assert (res[0][i] == exp).all()


def _cc16_worker(rank, world, port, total, master_seed, circuit, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garbled_snark_verifier_amd import sharding
    import hostsim_lib as h
    sp = h.SimProgram(circuit)

    def garble(c, seeds, indexes):  # the GPU garbler's stand-in: the host interpreter of the compiled device schedule
        recs = []
        for sd, idx in zip(seeds, indexes):
            labs = h.labels_from_seed(sd, 3 + sp.info["n_inputs"])
            delta, consts, inputs = labs[0], labs[1:3], labs[3:]
            out, cts = sp.garble(delta, consts, inputs)
            recs.append(sharding.commit_record(idx, h.cbcmac(cts), out, delta, consts[0], consts[1], inputs))
        return np.stack(recs)

    table, seeds = sharding.cut_and_choose_commit(circuit, master_seed, total, rank, world, garble=garble, device="cpu")
    q.put((rank, table.copy(), [int(x) for x in seeds]))
    dist.barrier()
    dist.destroy_process_group()


def test_cc16_cut_and_choose_commit_world2():
    """BASELINE config 5 as sharding.cut_and_choose_commit runs it (bench.py --workload cc16): 16 seeds from one master seed,
    instance i -> rank i mod 2, every instance garbled WITH its ciphertext commitment, one all-gather of the GarbledInstanceCommit
    records.  Here on gloo with the host interpreter as the garbler and a shortened circuit (Fq multiplication): both ranks end up
    with the same table, and all 16 records equal the ones built from the CPU oracle's garbling."""
    world, total, master, circuit = 2, 16, 1234, "fq_mul"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cc16_worker, args=(r, world, port, total, master, circuit, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, table, seeds = q.get(timeout=300)
        res[r] = (table, seeds)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (res[0][0] == res[1][0]).all() and res[0][1] == res[1][1] and len(set(res[0][1])) == total
    sys.path.insert(0, ROOT)
    from garbled_snark_verifier_amd import sharding
    assert res[0][0].shape == (total, sharding.record_len(254, 508))
    for i in range(total):
        g = o.garble(circuit, res[0][1][i])
        exp = sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0)
        assert (res[0][0][i] == exp).all(), "record %d differs from the oracle's" % i


def test_cc16_cut_and_choose_commit_world8():
    """BASELINE config 5 at its stated rank count: 16 instances over EIGHT ranks (two per rank, instance i -> rank i mod 8), gloo on
    CPU with the host interpreter as the garbler (u254 adder: eight processes share this container's cores): every rank ends up with
    the same 16-record table, ordered by instance index, each record equal to the one built from the CPU oracle's garbling — the
    rank logic the driver's 8-GPU run executes, exercised before it ever meets eight GPUs."""
    world, total, master, circuit = 8, 16, 1234, "u254_add"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cc16_worker, args=(r, world, port, total, master, circuit, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, table, seeds = q.get(timeout=600)
        res[r] = (table, seeds)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from garbled_snark_verifier_amd import sharding
    assert all((res[r][0] == res[0][0]).all() and res[r][1] == res[0][1] for r in range(world))
    assert res[0][0].shape == (total, sharding.record_len(255, 508)) and len(set(res[0][1])) == total
    assert [int.from_bytes(bytes(rec[:8]), "little") for rec in res[0][0]] == list(range(total))
    for i in range(total):
        g = o.garble(circuit, res[0][1][i])
        exp = sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0)
        assert (res[0][0][i] == exp).all(), "record %d differs from the oracle's" % i


def _plan_failure_worker(rank, world, port, bad_rank, tmpdir, q):
    """bench.get_plan with a stand-in engine module: local rank 0 "builds" the plan file, every other rank loads it; `bad_rank`'s load
    raises.  Every rank must come back with the same RuntimeError instead of waiting in a barrier."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    import argparse
    import bench

    class FakePlan:
        @staticmethod
        def build_file(circuit, units, path, window_div=4):
            open(path, "wb").write(b"plan")

        @staticmethod
        def load(path, engine):
            if rank == bad_rank:
                raise OSError("rank %d: cannot map the plan file" % rank)
            assert open(path, "rb").read() == b"plan"
            return "plan@%d" % rank

    class FakeGsv:
        Plan = FakePlan

    d = bench.Dist(world, "gloo", "cpu")
    args = argparse.Namespace(no_plan_cache=False, plan_cache=tmpdir)
    try:
        plan, info, _ = bench.get_plan(FakeGsv, None, args, "circuit", ["u"], rank, rank, world, d, lambda m: None)
        q.put((rank, "ok", plan))
    except RuntimeError as e:
        q.put((rank, "error", str(e)))
    d.barrier()
    d.close()


@pytest.mark.parametrize("bad_rank", [0, 5])
def test_cc16_ranks_agree_on_a_failed_plan_load_world8(bad_rank, tmp_path):
    """One plan per node: local rank 0 builds the plan file, every rank loads it (bench.get_plan, used by `--workload cc16` and the
    headline).  A rank whose load fails — rank 0 itself, or one of the others — must take ALL EIGHT ranks out with an error they agree
    on (min over ranks) before anyone waits in a barrier; a hang here would cost the 8-GPU run its whole time limit."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_failure_worker, args=(r, world, port, bad_rank, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (st, msg)) for r, st, msg in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(res[r][0] == "error" for r in range(world)), res
    assert "plan" in res[0][1] and "failed on a rank" in res[0][1]


def _plan_sharing_worker(rank, world, port, tmpdir, q):
    """bench.get_plan with the REAL plan builder and loader on a small circuit: local rank 0 builds the plan file (gsv_plan_build_file),
    every rank loads it (gsv_plan_load; without a device the loader makes a host copy instead of streaming into a GPU — the file path,
    the barriers and the agreement logic are the ones the 8-GPU run takes)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    os.environ["GSV_COMPILE_THREADS"] = "1"
    import argparse
    import bench
    import garbled_snark_verifier_amd as gsv
    builds = []

    class CountingPlan:
        @staticmethod
        def build_file(circuit, units, path, window_div=4):
            builds.append(rank)
            gsv.Plan.build_file(circuit, units, path, window_div=window_div)

        @staticmethod
        def load(path, engine):
            return gsv.Plan.load(path, None)

    class Gsv:
        Plan = CountingPlan

    d = bench.Dist(world, "gloo", "cpu")
    args = argparse.Namespace(no_plan_cache=False, plan_cache=tmpdir)
    plan, info, _ = bench.get_plan(Gsv, None, args, "fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], rank, rank, world, d, lambda m: None)
    forms = plan.call_record_forms()
    q.put((rank, len(builds), info["how"], info["cache_file"], dict(plan.info), forms, bench.mac_threads_for_rank(0, world, quota=16.0)))
    plan.close()
    d.barrier()
    d.close()


def test_plan_file_is_built_once_and_shared_by_eight_ranks(tmp_path):
    """What `bench.py --gpus 8` does before its first launch: ONE plan file per node — local rank 0 builds it, the other seven load the same
    file (page cache) — with the real builder and loader.  Every rank ends up with the same plan; exactly one build happened; every rank
    takes an eighth of a 16-core quota for its MAC workers."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_sharing_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: rest for r, *rest in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sum(res[r][0] for r in range(world)) == 1 and res[0][0] == 1
    assert res[0][1] == "built to file, loaded" and all(res[r][1] == "loaded" for r in range(1, world))
    assert len({res[r][2] for r in range(world)}) == 1 and os.path.exists(res[0][2])
    assert all(res[r][3] == res[0][3] and res[r][4] == res[0][4] for r in range(world)) and res[0][3]["n_calls"] > 1
    assert all(res[r][5] == 2 for r in range(world))


def test_instance_seeds_are_drawn_as_the_reference_draws_them():
    """Garbler::create draws `rng.gen::<u64>()` per instance (cut_and_choose/garbler.rs:201-203) on the caller's RNG — the reference's
    own test: ChaCha20Rng::seed_from_u64(1234) (cut_and_choose/tests.rs:102).  sharding.instance_seeds (the product's ChaCha stream,
    gsv_labels_from_seed) equals the oracle's independent ChaCha restatement value for value, the committed cut-and-choose fixtures were
    built from exactly these seeds, an odd count takes the low half of the last u128, and the pre-round-5 numpy draw is still there
    behind an explicit argument."""
    import json
    import oracle_lib as o
    from garbled_snark_verifier_amd import sharding
    ref = sharding.u64_stream_from_labels(o.chacha_labels(1234, 8), 16)
    got = sharding.instance_seeds(1234, 16)
    assert got.dtype == np.uint64 and (got == ref).all() and len(set(int(x) for x in got)) == 16
    # next_u64 pairs: u128 = lo | hi << 64 with the FIRST call the low half; delta is the first u128 GarbleMode::new draws from the same stream
    lab = o.chacha_labels(1234, 1)[0]
    assert int(got[0]) == int.from_bytes(bytes(lab[8:16]), "big") and int(got[1]) == int.from_bytes(bytes(lab[0:8]), "big")
    assert (sharding.instance_seeds(1234, 5) == got[:5]).all() and len(sharding.instance_seeds(1234, 0)) == 0
    for name in ("cc16_golden.json", "cc16_verifier_golden.json"):
        g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))
        assert g["master_seed"] == 1234 and g["seeds"] == [int(x) for x in sharding.instance_seeds(g["master_seed"], g["total"])]
    old = sharding.instance_seeds(2024, 16, rng="numpy")
    assert (old == np.random.Generator(np.random.PCG64(2024)).integers(0, 2**63, size=16, dtype=np.uint64)).all()
    with pytest.raises(ValueError):
        sharding.instance_seeds(1, 1, rng="mt19937")


def test_shard_instances_partition():
    sys.path.insert(0, ROOT)
    from garbled_snark_verifier_amd import sharding
    for total in (1, 5, 16, 17):
        for world in (1, 2, 4, 8):
            parts = [sharding.shard_instances(total, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert all(i % world == r for r, p in enumerate(parts) for i in p)  # instance i -> GPU i mod n_gpu


def test_regarbling_check_file_side_on_cpu(tmp_path):
    """Evaluator::run_regarbling for instances KEPT for evaluation needs no GPU: gc_<i>.bin is streamed through the CBC-MAC
    and compared with the committed ciphertext hash (evaluator.rs:105-138).  Files and commits come from the oracle here."""
    import os
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding
    recs = []
    for i, seed in enumerate([5, 6, 7]):
        g = o.garble("u254_add", seed)
        h = gsv.write_gc_file(os.path.join(str(tmp_path), gsv.gc_file_name(i)), g.ciphertexts)
        assert h == g.ct_hash.tobytes()
        recs.append(sharding.commit_record(i, h, g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0))
    commits = np.stack(recs)
    assert sharding.run_regarbling(commits, [0, 1, 2], {}, "u254_add", str(tmp_path)) == (True, {})
    with open(os.path.join(str(tmp_path), gsv.gc_file_name(1)), "r+b") as f:
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 0x80]))
    os.remove(os.path.join(str(tmp_path), gsv.gc_file_name(2)))
    ok, errors = sharding.run_regarbling(commits, [0, 1, 2], {}, "u254_add", str(tmp_path))
    assert not ok and errors[1] == "ciphertext corrupted" and errors[2].startswith("failed to get ciphertext source") and 0 not in errors


def test_evaluate_from_consistency_checks(tmp_path):
    """Evaluator::evaluate_from (cut_and_choose/evaluator.rs:338-476): the evaluator checks the constants, the active input labels
    and the ciphertext file it was handed against the garbler's commit record, and the output label it derives against the output
    commits.  Honest case passes; every tampering is reported with the reference's ConsistencyError variant.  The evaluation
    itself is the CPU oracle's here (the GPU path of the same function is covered by test_gpu_parity)."""
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding
    circuit, n_in, n_out = "u254_add", 508, 255
    gs, recs = [], []
    for i, seed in enumerate([21, 22]):
        g = o.garble(circuit, seed)
        gs.append(g)
        h = gsv.write_gc_file(os.path.join(str(tmp_path), gsv.gc_file_name(i)), g.ciphertexts)
        recs.append(sharding.commit_record(i, h, g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0))
    commits = np.stack(recs)
    assert commits.shape[1] == sharding.record_len(n_out, n_in)
    # the record never holds both labels of a constant: publishing AES_K(l) and AES_K(l ^ delta) under a public key reveals delta
    idx, cth, inc, outc, tc, fc = sharding.record_fields(commits[0], n_out, n_in)
    assert idx == 0 and bytes(tc) == o.cbcmac((gs[0].true_label0 ^ gs[0].delta).tobytes()) and bytes(fc) == o.cbcmac(gs[0].false_label0.tobytes())
    assert bytes(inc[3, 1]) == o.cbcmac((gs[0].input_label0[3] ^ gs[0].delta).tobytes()) and bytes(outc[0, 1]) == o.cbcmac(gs[0].output_label0[0].tobytes())
    rng = np.random.default_rng(5)

    def oracle_eval(index, t_act, f_act, in_act, in_bits):
        cts, h = gsv.read_gc_file(os.path.join(str(tmp_path), gsv.gc_file_name(index)))
        e = o.evaluate(circuit, t_act, f_act, in_act, in_bits, cts)
        return e.output_active, e.output_bits, e.ct_hash.tobytes()

    def case(i, **over):
        g = gs[i]
        bits = rng.integers(0, 2, n_in).astype(np.uint8)
        c = {"index": i, "true_constant_wire": g.true_label0 ^ g.delta, "false_constant_wire": g.false_label0,
             "input_active": np.where(bits[:, None] == 1, g.input_label0 ^ g.delta[None, :], g.input_label0), "input_bits": bits}
        c.update(over)
        return c

    res = sharding.evaluate_from(commits, [case(0), case(1)], circuit, str(tmp_path), n_out, evaluate=oracle_eval)
    assert [r[0] for r in res] == [0, 1]
    for (i, act, bits) in res:
        assert (act == np.where(bits[:, None] == 1, gs[i].output_label0 ^ gs[i].delta[None, :], gs[i].output_label0)).all()

    def kind(c, ev=oracle_eval):
        with pytest.raises(sharding.ConsistencyError) as ei:
            sharding.evaluate_from(commits, [c], circuit, str(tmp_path), n_out, evaluate=ev)
        return ei.value.kind, ei.value.index

    assert kind(case(0, true_constant_wire=gs[0].true_label0)) == ("TrueConstantMismatch", 0)   # label0 of TRUE instead of its label1
    assert kind(case(1, false_constant_wire=gs[1].false_label0 ^ gs[1].delta)) == ("FalseConstantMismatch", 1)
    c = case(0)
    c["input_active"] = c["input_active"].copy(); c["input_active"][7] ^= gs[0].delta          # the garbler's OTHER label for input 7
    assert kind(c) == ("InputLabelsMismatch", 0)
    c = case(0)
    c["input_active"] = c["input_active"][:-1]; c["input_bits"] = c["input_bits"][:-1]
    assert kind(c, ev=lambda *a: (np.zeros((n_out, 16), np.uint8), np.zeros(n_out, np.uint8), bytes(16))) == ("InputLabelsCountMismatch", 0)
    path = os.path.join(str(tmp_path), gsv.gc_file_name(1))
    raw = bytearray(open(path, "rb").read()); raw[-1] ^= 1; open(path, "wb").write(bytes(raw))
    assert kind(case(1))[0] in ("CiphertextMismatch",)
    os.remove(path)
    assert kind(case(1)) == ("MissingCiphertextHash", 1)
    # a wrong output label (evaluation backend returning garbage) is caught by the output commits
    assert kind(case(0), ev=lambda i, t, f, a, b: (np.zeros((n_out, 16), np.uint8), np.zeros(n_out, np.uint8), bytes(commits[0, 8:24]))) == ("OutputLabelMismatch", 0)
