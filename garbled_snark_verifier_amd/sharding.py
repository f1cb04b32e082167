"""Cut-and-choose instance sharding across GPUs (one process per GPU, torch.distributed).

The path partitions into independent units: instance i — own seed, own delta, labels and
ciphertext stream — goes to rank i mod world (reference: `seeds.par_iter()` over a pinned rayon pool,
src/cut_and_choose/garbler.rs:206-234).  Nothing is exchanged while garbling; the single collective is an
all-gather of fixed-size commit records at the end (GarbledInstanceCommit, garbler.rs:61-99), which is
latency-bound (tens of KB per instance) so ring-vs-direct over xGMI is irrelevant.

Commit record (bytes), one per instance — the fields of GarbledInstanceCommit in declaration order behind the
instance index:
    index u64 LE | ciphertext_commit 16 | input_labels_commit: n_in x {commit(label0), commit(label1)} |
    per output wire {commit(label1), commit(label0)} (output_label1_commit, output_label0_commit) |
    true_constant_commit = commit(true.label1) | false_constant_commit = commit(false.label0)
with commit = AesLabelCommitHasher = AES_K(label) (cut_and_choose/mod.rs:41-48).  Only the constants' SEMANTIC labels are
committed (true.select(true), false.select(false), garbler.rs:94-97): AES_K has a public key and is invertible, so publishing
both labels of one wire would reveal delta.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import lib as _lib  # noqa: F401  (host AES for label commits lives behind the C ABI)


def u64_stream_from_labels(labels, total):
    """The first `total` values of `rng.next_u64()` given the first ceil(total / 2) values of `rng.gen::<u128>()` as 16-byte big-endian
    labels (S::to_bytes): rand 0.8.5 draws a u128 as next_u64() | next_u64() << 64 — the first call is the LOW half."""
    out = []
    for lab in np.asarray(labels, np.uint8).reshape(-1, 16):
        b = bytes(lab)
        out += [int.from_bytes(b[8:16], "big"), int.from_bytes(b[0:8], "big")]
    return np.array(out[:total], dtype=np.uint64)


def instance_seeds(master_seed, total, rng="chacha"):
    """`total` u64 seeds drawn up front by the garbler, as the reference draws them: `rng.gen::<u64>()` per instance on the caller's RNG
    (cut_and_choose/garbler.rs:201-203), which the reference's own test seeds as `ChaCha20Rng::seed_from_u64(1234)`
    (cut_and_choose/tests.rs:102).  The ChaCha stream is the product's own (gsv_labels_from_seed: the generator behind
    GarbleMode::new); tests compare it with the oracle's independent restatement.  rng="numpy" keeps the pre-round-5 PCG64 draw for
    callers that stored such seeds."""
    if rng == "numpy":
        g = np.random.Generator(np.random.PCG64(master_seed))
        return g.integers(0, 2**63, size=total, dtype=np.uint64)
    if rng != "chacha":
        raise ValueError("rng must be \"chacha\" or \"numpy\"")
    from . import labels_from_seed
    n128 = (int(total) + 1) // 2
    d, f, t, inp = labels_from_seed(int(master_seed), max(0, n128 - 3))
    labels = np.concatenate([np.stack([d, f, t]), np.asarray(inp, np.uint8).reshape(-1, 16)])[:n128]
    return u64_stream_from_labels(labels, int(total))


def shard_instances(total, rank, world):
    return list(range(rank, total, world))


def record_len(n_outputs, n_inputs=0):
    return 8 + 16 + 32 * n_inputs + 32 * n_outputs + 32


def record_fields(rec, n_outputs, n_inputs):
    """Views into one record: (index, ct_hash, input_commits[n_in,2,16], output_commits[n_out,2,16] = (label1, label0),
    true_constant_commit, false_constant_commit)."""
    rec = np.asarray(rec, np.uint8)
    assert rec.shape[-1] == record_len(n_outputs, n_inputs)
    o = 24
    inp = rec[o:o + 32 * n_inputs].reshape(n_inputs, 2, 16); o += 32 * n_inputs
    out = rec[o:o + 32 * n_outputs].reshape(n_outputs, 2, 16); o += 32 * n_outputs
    return int.from_bytes(bytes(rec[:8]), "little"), rec[8:24], inp, out, rec[o:o + 16], rec[o + 16:o + 32]


def commit_labels(labels):
    """AesLabelCommitHasher over an [n,16] array of labels."""
    from . import _chk, _p, lib
    labels = np.ascontiguousarray(labels, np.uint8).reshape(-1, 16)
    out = np.zeros_like(labels)
    _chk(lib().gsv_commit_labels(_p(labels), labels.shape[0], _p(out)))
    return out


def commit_record(index, ct_hash, output_label0, delta, false_label0, true_label0, input_label0=None):
    """GarbledInstanceCommit::new (garbler.rs:85-99).  `input_label0`: [n_in,16] label0 of the circuit's input wires in
    EncodeInput order (GarbledInstance.input_wire_values); label1 = label0 ^ delta."""
    out0 = np.ascontiguousarray(output_label0, np.uint8).reshape(-1, 16)
    delta = np.ascontiguousarray(delta, np.uint8).reshape(16)
    in0 = np.zeros((0, 16), np.uint8) if input_label0 is None else np.ascontiguousarray(input_label0, np.uint8).reshape(-1, 16)
    labels = np.concatenate([
        np.stack([in0, in0 ^ delta[None, :]], axis=1).reshape(-1, 16),    # per input: label0, label1
        np.stack([out0 ^ delta[None, :], out0], axis=1).reshape(-1, 16),  # per output: label1, label0
        (np.asarray(true_label0, np.uint8) ^ delta)[None, :],             # true.select(true)
        np.asarray(false_label0, np.uint8)[None, :],                      # false.select(false)
    ])
    rec = np.zeros(record_len(out0.shape[0], in0.shape[0]), np.uint8)
    rec[:8] = np.frombuffer(int(index).to_bytes(8, "little"), np.uint8)
    rec[8:24] = np.frombuffer(bytes(ct_hash), np.uint8)
    rec[24:] = commit_labels(labels).reshape(-1)
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
    a = out.cpu().numpy()
    idx = a[:, :8].copy().view("<u8").reshape(-1)
    return out.cpu()[torch.from_numpy(np.argsort(idx, kind="stable"))]


def garble_and_commit(circuit, seeds, indexes, engine=None, program=None, replays=1, gc_dir=None, session_kw=None, threads=0):
    """Garbler::create + commit (garbler.rs:191-257) for the given (index, seed) pairs in ONE session: returns the
    [len(seeds), record_len] commit records; with gc_dir the ciphertext streams go to gc_<index>.bin
    (ciphertext_repository.rs:94-127).  Indexes must be consecutive when gc_dir is used.  `program` may be a Plan (the
    verifier): the stream is then drained window by window of the session's schedule."""
    from . import Engine, Plan, Program, Session, labels_from_seed
    engine = engine or Engine(0)
    program = program or Program.from_circuit(circuit, chain_feedback=replays > 1)
    n_in = program.info["n_inputs"]
    B = len(seeds)
    delta = np.zeros((B, 16), np.uint8); consts = np.zeros((B, 2, 16), np.uint8); inputs = np.zeros((B, n_in, 16), np.uint8)
    for i, s in enumerate(seeds):
        delta[i], consts[i, 0], consts[i, 1], inputs[i] = labels_from_seed(int(s), n_in)
    if isinstance(program, Plan):
        sess = Session(engine, program, B, retain_stream=False, **(session_kw or {}))
    else:
        sess = Session(engine, program, B, replays, min(replays, 2) if replays > 1 else 1)
    sess.set_garble_inputs(delta, consts, inputs)
    if gc_dir is not None:
        assert list(indexes) == list(range(indexes[0], indexes[0] + B)), "gc files are numbered first_index + i"
    hashes = sess.garble_streaming(directory=gc_dir, first_index=int(indexes[0]) if B else 0, threads=threads)  # threads: host MAC workers (0 = the engine's default)
    outs = sess.read_outputs()
    recs = np.stack([commit_record(indexes[i], hashes[i], outs[i], delta[i], consts[i, 0], consts[i, 1], inputs[i]) for i in range(B)])
    sess.close()
    return recs


def cut_and_choose_commit(circuit, master_seed, total, rank, world, engine=None, program=None, garble=None, device=None, session_kw=None, threads=0):
    """BASELINE config 5 / `Garbler::create` -> `commit` (garbler.rs:191-257) across ranks: `total` seeds are drawn from one master
    seed (:201-203), instance i goes to rank i mod world (the reference: one instance per pinned core, mod.rs:131-186), every rank
    garbles its instances WITH the ciphertext commitment (AESAccumulatingHash over the whole stream, :219-222) and builds their
    GarbledInstanceCommit records, and ONE all-gather leaves every rank with the [total, record_len] table ordered by instance
    index.  Nothing else is exchanged.  `garble(circuit, seeds, indexes) -> records` replaces the GPU garbler in the CPU tests.
    `threads`: host MAC workers of this rank's drain — the ranks of a node share the host's cores (bench.mac_threads_for_rank).
    Returns (table as a uint8 numpy array, seeds)."""
    seeds = instance_seeds(master_seed, total)
    mine = shard_instances(total, rank, world)
    if garble is None:
        garble = lambda c, sd, idx: garble_and_commit(c, sd, idx, engine=engine, program=program, session_kw=session_kw, threads=threads)  # noqa: E731
    n_out, n_in = (program.info["n_outputs"], program.info["n_inputs"]) if program is not None else (None, None)
    if mine:
        local = np.ascontiguousarray(garble(circuit, [int(seeds[i]) for i in mine], mine), np.uint8)
    else:
        if n_out is None:
            raise ValueError("a rank without instances needs `program` to size its (empty) share of the gather")
        local = np.zeros((0, record_len(n_out, n_in)), np.uint8)
    table = all_gather_records(torch.from_numpy(local), total, rank, world, device=device)
    return table.numpy(), seeds


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


class ConsistencyError(Exception):
    """cut_and_choose/evaluator.rs ConsistencyError: `kind` is the variant name, `index` the instance."""

    def __init__(self, kind, index, **details):
        super().__init__("%s (instance %d)%s" % (kind, index, (" " + repr(details)) if details else ""))
        self.kind, self.index, self.details = kind, index, details


def _gpu_evaluate_batch(circuit, engine, program, gc_dir):
    """Default evaluation backend of evaluate_from: EvaluateMode over FileSources on the GPU, ALL finalized instances in one session —
    one launch per window of the stream for the whole batch (gsv_session_evaluate_streaming_indexed).  The reference evaluates the
    cases side by side on a rayon pool (`into_par_iter`, cut_and_choose/evaluator.rs:354-475); a one-instance session per case would
    use one CU of 256 and pay a session per case.  The files' CBC-MACs (the CiphertextMismatch check) are folded while reading, instance i on
    worker i mod T of the engine's pool, beside the uploads (engine.cpp, evaluate_streaming_impl)."""
    from . import Engine, Plan, Program, Session
    eng = engine or Engine(0)
    prog = program or Program.from_circuit(circuit)

    def run(indexes, true_active, false_active, input_active, input_bits):
        B = len(indexes)
        sess = Session(eng, prog, B, retain_stream=False) if isinstance(prog, Plan) else Session(eng, prog, B)
        try:
            consts = np.stack([np.asarray(false_active, np.uint8).reshape(B, 16), np.asarray(true_active, np.uint8).reshape(B, 16)], axis=1)
            sess.set_evaluate_inputs(consts, np.asarray(input_active, np.uint8).reshape(B, -1, 16), np.asarray(input_bits, np.uint8).reshape(B, -1))
            hashes = sess.evaluate_streaming_indexed(gc_dir, [int(i) for i in indexes])
            labels, bits = sess.read_outputs(with_bits=True)
            return [(labels[k], bits[k], hashes[k]) for k in range(B)]
        finally:
            sess.close()
    return run


def evaluate_from(commits, cases, circuit, gc_dir, n_outputs, engine=None, program=None, evaluate=None, evaluate_batch=None):
    """Evaluator::evaluate_from (cut_and_choose/evaluator.rs:338-476): evaluate the finalized instances from their gc_<i>.bin and
    check everything the evaluator was handed against the garbler's commit record BEFORE trusting the result.
    `cases`: list of dicts {index, true_constant_wire[16], false_constant_wire[16], input_active[n_in,16], input_bits[n_in]}
    (EvaluatorCaseInput).  Raises ConsistencyError with the reference's variants — TrueConstantMismatch,
    FalseConstantMismatch, MissingCiphertextHash, InputLabelsCountMismatch, InputLabelsMismatch, CiphertextMismatch,
    OutputLabelMismatch — and returns [(index, output_active[n_out,16], output_bits[n_out])].

    The cases are evaluated TOGETHER (the reference: `into_par_iter`): the checks that need no evaluation run for every case first,
    then one batch evaluates every case in front of the first failing one, then the remaining checks run case by case — so the error
    raised is the one a case-by-case run in list order would raise (per case the reference's order of checks, evaluator.rs:371-465).
    `evaluate_batch(indexes, true[B,16], false[B,16], input_active[B,n_in,16], input_bits[B,n_in]) -> [(output_active, output_bits,
    ciphertext_hash)]` defaults to the GPU evaluator; `evaluate(index, true, false, input_active, input_bits) -> (..)` is the
    case-by-case form tests without a GPU pass (the CPU oracle's)."""
    import os
    from . import gc_file_name
    commits = np.asarray(commits, np.uint8)
    if evaluate_batch is None:
        if evaluate is not None:
            evaluate_batch = lambda idx, t, f, a, b: [evaluate(idx[k], t[k], f[k], a[k], b[k]) for k in range(len(idx))]  # noqa: E731
        else:
            evaluate_batch = _gpu_evaluate_batch(circuit, engine, program, gc_dir)
    n_in_committed = (commits.shape[1] - record_len(n_outputs, 0)) // 32
    parsed, first_error = [], None
    for case in cases:
        index = int(case["index"])
        t_act = np.asarray(case["true_constant_wire"], np.uint8).reshape(16)
        f_act = np.asarray(case["false_constant_wire"], np.uint8).reshape(16)
        in_act = np.asarray(case["input_active"], np.uint8).reshape(-1, 16)
        in_bits = np.asarray(case["input_bits"], np.uint8).reshape(-1)
        _, ct_commit, in_commits, out_commits, true_commit, false_commit = record_fields(commits[index], n_outputs, n_in_committed)
        got = commit_labels(np.stack([t_act, f_act]))
        if bytes(got[0]) != bytes(true_commit):
            first_error = ConsistencyError("TrueConstantMismatch", index, expected=bytes(true_commit), actual=bytes(got[0]))
        elif bytes(got[1]) != bytes(false_commit):
            first_error = ConsistencyError("FalseConstantMismatch", index, expected=bytes(false_commit), actual=bytes(got[1]))
        elif not os.path.exists(os.path.join(gc_dir, gc_file_name(index))):
            first_error = ConsistencyError("MissingCiphertextHash", index)
        elif in_act.shape[0] != n_in_committed or in_bits.shape[0] != n_in_committed:
            # (the reference counts after evaluating; a batch cannot hold a ragged case, and no other check can fire first)
            first_error = ConsistencyError("InputLabelsCountMismatch", index, expected=n_in_committed, actual=in_act.shape[0])
        if first_error is not None:
            break
        parsed.append((index, t_act, f_act, in_act, in_bits, ct_commit, in_commits, out_commits))
    results = []
    if parsed:
        evaluated = evaluate_batch([c[0] for c in parsed], np.stack([c[1] for c in parsed]), np.stack([c[2] for c in parsed]),
                                   np.stack([c[3] for c in parsed]), np.stack([c[4] for c in parsed]))
        for (index, _t, _f, in_act, in_bits, ct_commit, in_commits, out_commits), (out_act, out_bits, file_hash) in zip(parsed, evaluated):
            actual = commit_labels(in_act)
            expected = in_commits[np.arange(n_in_committed), in_bits.astype(np.int64) & 1]  # commit_for_value(value)
            bad = np.nonzero((actual != expected).any(axis=1))[0]
            if bad.size:
                k = int(bad[0])
                raise ConsistencyError("InputLabelsMismatch", index, label_index=k, expected=bytes(expected[k]), actual=bytes(actual[k]))
            if bytes(file_hash) != bytes(ct_commit):
                raise ConsistencyError("CiphertextMismatch", index, expected=bytes(ct_commit), actual=bytes(file_hash))
            out_act = np.asarray(out_act, np.uint8).reshape(-1, 16)
            out_bits = np.asarray(out_bits, np.uint8).reshape(-1)
            oh = commit_labels(out_act)
            exp_out = out_commits[np.arange(n_outputs), 1 - (out_bits.astype(np.int64) & 1)]  # value 1 -> label1 commit (stored first)
            if (oh != exp_out).any():
                raise ConsistencyError("OutputLabelMismatch", index)
            results.append((index, out_act, out_bits))
    if first_error is not None:
        raise first_error
    return results
