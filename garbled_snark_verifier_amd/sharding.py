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
