"""Cut-and-choose instance sharding across GPUs (one process per GPU, torch.distributed).

The path partitions into independent units: instance i — own seed, own delta, labels and
ciphertext stream — goes to rank i mod world (reference: `seeds.par_iter()` over a pinned rayon pool,
src/cut_and_choose/garbler.rs:206-234).  Nothing is exchanged while garbling; the single collective is an
all-gather of fixed-size commit records at the end (GarbledInstanceCommit, garbler.rs:63-99), which is
latency-bound (tens of KB) so ring-vs-direct over xGMI is irrelevant.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import lib as _lib  # noqa: F401  (host AES for label commits lives behind the C ABI)


def instance_seeds(master_seed, total):
    """`total` u64 seeds drawn up front by the garbler (garbler.rs:201-203 draws rng.gen::<u64>() per instance).
    The stand-alone harness uses numpy's PCG64 here; a Rust host passes its own seeds."""
    rng = np.random.Generator(np.random.PCG64(master_seed))
    return rng.integers(0, 2**63, size=total, dtype=np.uint64)


def shard_instances(total, rank, world):
    return list(range(rank, total, world))


def record_len(n_outputs):
    # index (8 B) | ciphertext hash (16) | commit(false.label0) commit(false.label1) commit(true.label0) commit(true.label1) (64)
    # | per output: commit(label0), commit(label1) (32 each)
    return 8 + 16 + 64 + 32 * n_outputs


def commit_record(index, ct_hash, output_label0, delta, false_label0, true_label0):
    """GarbledInstanceCommit::new (garbler.rs:85-99) with AesLabelCommitHasher = AES_K(label) (cut_and_choose/mod.rs:41-48)."""
    from . import _chk, _p, lib
    out0 = np.ascontiguousarray(output_label0, np.uint8).reshape(-1, 16)
    delta = np.ascontiguousarray(delta, np.uint8).reshape(16)
    labels = np.concatenate([
        np.stack([false_label0, false_label0 ^ delta, true_label0, true_label0 ^ delta]).astype(np.uint8),
        np.stack([out0, out0 ^ delta[None, :]], axis=1).reshape(-1, 16),
    ])
    labels = np.ascontiguousarray(labels, np.uint8)
    commits = np.zeros_like(labels)
    _chk(lib().gsv_commit_labels(_p(labels), labels.shape[0], _p(commits)))
    rec = np.zeros(record_len(out0.shape[0]), np.uint8)
    rec[:8] = np.frombuffer(int(index).to_bytes(8, "little"), np.uint8)
    rec[8:24] = np.frombuffer(bytes(ct_hash), np.uint8)
    rec[24:] = commits.reshape(-1)
    return rec


def all_gather_records(local, total, rank, world, device=None):
    """All ranks end up with the [total, record_len] table ordered by instance index.  One all_gather
    (RCCL over xGMI when the backend is nccl; gloo in the CPU tests)."""
    if world == 1:
        out = local
    else:
        per = -(-total // world)
        rec_len = local.shape[1]
        dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
        buf = torch.zeros((per, rec_len), dtype=torch.uint8, device=dev)
        buf[: local.shape[0]] = local.to(dev)
        gathered = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        rows = []
        for r in range(world):
            n_r = len(shard_instances(total, r, world))
            rows.append(gathered[r][:n_r].cpu())
        out = torch.cat(rows)
    # order by the instance index stored in the first 8 bytes
    idx = np.array([int.from_bytes(bytes(row[:8].tolist()), "little") for row in out.cpu().numpy()])
    return out.cpu()[torch.from_numpy(np.argsort(idx))]


def garble_and_commit(circuit, seeds, indexes, engine=None, program=None, replays=1, gc_dir=None):
    """Garbler::create + commit (garbler.rs:191-257) for the given (index, seed) pairs in ONE launch: returns the
    [len(seeds), record_len] commit records; with gc_dir the ciphertext streams go to gc_<index>.bin
    (ciphertext_repository.rs:94-127).  Indexes must be consecutive when gc_dir is used."""
    from . import CircuitBuilder, Engine, Program, Session, labels_from_seed
    import numpy as _np
    engine = engine or Engine(0)
    program = program or Program.from_circuit(circuit, chain_feedback=replays > 1)
    n_in = program.info["n_inputs"]
    B = len(seeds)
    delta = _np.zeros((B, 16), _np.uint8); consts = _np.zeros((B, 2, 16), _np.uint8); inputs = _np.zeros((B, n_in, 16), _np.uint8)
    for i, s in enumerate(seeds):
        delta[i], consts[i, 0], consts[i, 1], inputs[i] = labels_from_seed(int(s), n_in)
    sess = Session(engine, program, B, replays, min(replays, 2) if replays > 1 else 1)
    sess.set_garble_inputs(delta, consts, inputs)
    if gc_dir is not None:
        assert list(indexes) == list(range(indexes[0], indexes[0] + B)), "gc files are numbered first_index + i"
    hashes = sess.garble_streaming(directory=gc_dir, first_index=int(indexes[0]) if B else 0)
    outs = sess.read_outputs()
    recs = _np.stack([commit_record(indexes[i], hashes[i], outs[i], delta[i], consts[i, 0], consts[i, 1]) for i in range(B)])
    sess.close()
    return recs


def run_regarbling(commits, to_finalize, seeds, circuit, gc_dir, engine=None, program=None, replays=1):
    """Evaluator::run_regarbling (cut_and_choose/evaluator.rs:83-181).  `commits`: [total, record_len] table of the
    garbler's commit records; `to_finalize`: indexes kept for evaluation; `seeds`: {index: seed} of the OPENED instances.
      * finalized index: its gc_<index>.bin is streamed through the CBC-MAC and compared with the committed ciphertext
        hash ("ciphertext corrupted" otherwise);
      * opened index: the circuit is garbled again from the revealed seed — all opened instances in one GPU launch —
        and the whole commit record must match ("regarbling failed"); a missing seed is an error ("failed to find seed").
    Returns (ok, errors) with errors = {index: message}; the reference returns Err(()) as soon as any instance fails."""
    import os
    from . import gc_file_name, read_gc_file
    commits = np.asarray(commits, np.uint8)
    errors = {}
    fin = set(int(i) for i in to_finalize)
    for index in sorted(fin):
        path = os.path.join(gc_dir, gc_file_name(index))
        try:
            _, h = read_gc_file(path)
        except Exception as e:  # FileSource::from_path failing (ciphertext_source.rs:36-60)
            errors[index] = "failed to get ciphertext source: %s" % e
            continue
        if bytes(h) != bytes(commits[index, 8:24]):
            errors[index] = "ciphertext corrupted"
    opened = [i for i in range(commits.shape[0]) if i not in fin]
    missing = [i for i in opened if i not in seeds]
    for i in missing:
        errors[i] = "failed to find seed"
    todo = [i for i in opened if i in seeds]
    if todo:
        recs = garble_and_commit(circuit, [seeds[i] for i in todo], todo, engine=engine, program=program, replays=replays)
        for k, i in enumerate(todo):
            if not (recs[k] == commits[i]).all():
                errors[i] = "regarbling failed"
    return (not errors), errors
