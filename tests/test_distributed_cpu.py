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
        recs.append(sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0))
    local = torch.from_numpy(np.stack(recs)) if recs else torch.zeros((0, sharding.record_len(254)), dtype=torch.uint8)
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
        exp = sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0)
        assert (res[0][i] == exp).all()


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
        recs.append(sharding.commit_record(i, h, g.output_label0, g.delta, g.false_label0, g.true_label0))
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
