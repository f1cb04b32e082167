"""bench.py's rank logic on CPU: slice partition of a plan, seed partition between ranks, the sliced timed loop with its
barriers, the per-pass all-gather of commit records and the max-over-ranks elapsed time — at world_size 2 on gloo.  The GPU
session is replaced by the test-only host interpreter (tests/hostsim) running a small plan, so the records that travel are
real: they are compared with records built from the CPU oracle's garbling of the same seeds.  Also: plan files
(gsv_plan_save / gsv_plan_load) round-trip without a device, and bench.py refuses a WORLD_SIZE that contradicts --gpus."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPEC, UNITS, N_IN = "random_circuit:3", ["test::random_block"], 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plan_slices_partition():
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(1)
    for n, S in [(1340, 10), (1340, 8), (7, 10), (1, 3), (50, 50), (100, 1)]:
        g = rng.integers(1, 20_000_000, n)
        sl = bench.plan_slices(g, S)
        assert len(sl) == min(S, n) and sl[0][0] == 0 and sum(s[1] for s in sl) == n and all(s[1] >= 1 for s in sl)
        assert all(sl[k][0] + sl[k][1] == sl[k + 1][0] for k in range(len(sl) - 1))
        assert [s[2] for s in sl] == [int(g[s[0]:s[0] + s[1]].sum()) for s in sl]
        if n >= 20 * S:  # balanced to within the largest call
            assert max(s[2] for s in sl) - min(s[2] for s in sl) <= 2 * g.max()
    a, b = bench.instance_seeds(0, 512), bench.instance_seeds(1, 512)
    assert len(set(a) | set(b)) == 1024


class _StubWork:
    """What bench.VerifierWork is to timed_steps, on the host interpreter: B instances of a small plan; a slice only counts its
    calls and the pass is interpreted when its last slice runs (the interpreter has no call-range form)."""

    def __init__(self, rank, B, n_calls):
        import hostsim_lib as h
        import bench
        self.h, self.B, self.n_calls = h, B, n_calls
        self.plan = h.SimPlan(SPEC, UNITS)
        self.seeds = bench.instance_seeds(rank, B)
        self.labs = [h.labels_from_seed(s, 3 + N_IN) for s in self.seeds]
        self.done, self.passes, self.out, self.hashes = 0, 0, None, None

    def new_pass(self):
        assert self.done in (0, self.n_calls)  # a pass restarts only after the previous one ran all its calls
        self.done = 0

    def run_slice(self, first, n):
        assert first == self.done
        self.done += n
        if self.done == self.n_calls:
            res = [self.plan.garble(l[0], l[1:3], l[3:]) for l in self.labs]
            self.out = [r[0] for r in res]
            self.hashes = [self.h.cbcmac(r[1]) for r in res]
            self.passes += 1
        return 1.0

    def commit_records(self):
        from garbled_snark_verifier_amd import sharding
        return np.stack([sharding.commit_record(self.seeds[i], self.hashes[i], self.out[i], self.labs[i][0], self.labs[i][1], self.labs[i][2], self.labs[i][3:]) for i in range(self.B)])


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    import time
    import bench
    import hostsim_lib as h
    dist = bench.Dist(world, "gloo", "cpu")
    n_calls = h.SimPlan(SPEC, UNITS).info["n_calls"]
    work = _StubWork(rank, 2, n_calls)
    slices = bench.plan_slices([1000 + 7 * k for k in range(n_calls)], 4)
    if rank == 1:
        time.sleep(0.5)  # a slow rank: every rank must report the slowest rank's elapsed time
    r = bench.timed_steps(work, slices, warmup=2, steps=6, dist=dist, sync=lambda: None)
    # the budget guard cuts the step count identically on every rank (min over ranks)
    r2 = bench.timed_steps(_StubWork(rank, 1, n_calls), slices, warmup=4, steps=8, dist=dist, sync=lambda: None, time_budget_s=1e9 if rank == 0 else 0.0, t_start=time.time())
    q.put((rank, r["elapsed"], r["steps_run"], r["gates_per_instance"], r["calls"], work.passes, r["commit_table"].numpy().copy(), r2["steps_run"]))
    dist.barrier()
    dist.close()


def test_sliced_timed_loop_world2_gloo():
    sys.path.insert(0, ROOT)
    import bench
    from garbled_snark_verifier_amd import sharding
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (e0, k0, g0, c0, passes0, t0, b0), (e1, k1, g1, c1, passes1, t1, b1) = res[0], res[1]
    assert e0 == e1 and k0 == k1 == 6 and g0 == g1 and c0 == c1 and b0 == b1 == 1  # one elapsed time for the job (max over ranks); rank 1 was out of budget: both cut to one step
    # warm-up 2 + 6 timed steps over 4 slices = exactly two passes; slices 2,3,0,1,2,3 are the timed ones
    assert passes0 == passes1 == 2
    # every rank holds the same table: both ranks' records, rank-major, equal to records built from the oracle's garbling
    assert (t0 == t1).all() and t0.shape == (4, sharding.record_len(16, N_IN))
    for rank in range(world):
        for i, seed in enumerate(bench.instance_seeds(rank, 2)):
            g = o.garble(SPEC, seed)
            exp = sharding.commit_record(seed, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0)
            assert (t0[rank * 2 + i] == exp).all()


def test_session_slices_follow_window_boundaries():
    """A plan session executes whole windows of its schedule: slices are groups of windows, nearly equal in gates, covering every call."""
    sys.path.insert(0, ROOT)
    import bench
    gates = [100 + (7 * k) % 50 for k in range(40)]
    windows = [(0, 3, 1), (3, 1, 1), (4, 10, 2), (14, 6, 1), (20, 20, 3)]
    sl = bench.session_slices(windows, gates, 3)
    assert [x[0] for x in sl] == sorted(x[0] for x in sl) and sl[0][0] == 0 and sum(x[1] for x in sl) == 40 and sum(x[2] for x in sl) == sum(gates)
    starts = {w[0] for w in windows}
    assert all(x[0] in starts for x in sl) and len(sl) == 3
    assert len(bench.session_slices(windows, gates, 50)) == len(windows)  # never more slices than windows


def test_bench_refuses_mismatched_world_size():
    """`--gpus 8` under a launcher that started a different number of ranks must fail loudly, not run on fewer GPUs."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()


def test_plan_file_roundtrip_without_device(tmp_path):
    """gsv_plan_save / gsv_plan_load: a built plan survives the file byte for byte (save -> load -> save), with the same calls,
    counts and program images; truncated or foreign files are rejected; a plan loaded without an engine is a host copy."""
    import garbled_snark_verifier_amd as gsv
    plan = gsv.Plan.from_circuit("fq_complex", ["fp254::montgomery_reduce", "bigint::mul_karatsuba"], window_div=4)  # as bench.py builds its plan
    a, b = os.path.join(str(tmp_path), "a.gsvplan"), os.path.join(str(tmp_path), "b.gsvplan")
    plan.save(a)
    p2 = gsv.Plan.load(a)
    assert p2.info == plan.info and p2.image_bytes() == plan.image_bytes() and (p2.call_info() == plan.call_info()).all()
    ci = plan.call_info()
    assert ci[:, 1].sum() == plan.info["n_gates"] and ci[:, 3].sum() == plan.info["n_ciphertexts"] and (ci[1:, 0] == np.cumsum(ci[:-1, 1])).all()
    p2.save(b)
    raw = open(a, "rb").read()
    assert raw == open(b, "rb").read() and not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
    open(b, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(gsv.GsvError):
        gsv.Plan.load(b)
    open(b, "wb").write(b"not a plan" * 100)
    with pytest.raises(gsv.GsvError):
        gsv.Plan.load(b)
    with pytest.raises(gsv.GsvError):
        gsv.Plan.load(os.path.join(str(tmp_path), "missing.gsvplan"))
    # an offset table that points outside the file / into the header is refused
    import struct
    tab = struct.unpack_from("<Q", raw, 8 + 6 * 4 + 3 * 8 + 8)[0]
    for evil in (len(raw) + 16, 0):
        open(b, "wb").write(raw[:tab] + struct.pack("<Q", evil) + raw[tab + 8:])
        with pytest.raises(gsv.GsvError):
            gsv.Plan.load(b)


def test_plan_built_straight_to_a_file(tmp_path):
    """gsv_plan_build_file: the plan is never held — each program is appended to the file by the worker that compiled it (warm-up
    recorders and compile pool running) and dropped.  Loaded back it is the plan gsv_plan_from_circuit builds in memory: same calls,
    same counts, same program images; the in-memory plan saved by gsv_plan_save loads to the same thing; no temp file is left."""
    import garbled_snark_verifier_amd as gsv
    units = ["fq12::mul_montgomery", "fq12::square_montgomery"]
    a, b = os.path.join(str(tmp_path), "a.gsvplan"), os.path.join(str(tmp_path), "b.gsvplan")
    gsv.Plan.build_file("fq12_mix", units, a, window_div=4)
    ref = gsv.Plan.from_circuit("fq12_mix", units, window_div=4)
    ref.save(b)
    pa, pb = gsv.Plan.load(a), gsv.Plan.load(b)
    for q in (pa, pb):
        assert q.info == ref.info and q.image_bytes() == ref.image_bytes() and (q.call_info() == ref.call_info()).all() and q.wire_file() == ref.wire_file()
    assert os.path.getsize(a) == os.path.getsize(b) and not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
    # odd shapes: no unit at all (one glue program), a single gate, units with dead / constant / passed-through outputs, 145 calls of 72 programs
    for spec, us in (("fq_mul", ["no::such_unit"]), ("gate:0", ["x::y"]), ("driver_mix", ["test::inner", "bigint::add"]), ("random_circuit:3", ["test::random_block"])):
        gsv.Plan.build_file(spec, us, a, window_div=4)
        pa, pr = gsv.Plan.load(a), gsv.Plan.from_circuit(spec, us, window_div=4)
        assert pa.info == pr.info and pa.image_bytes() == pr.image_bytes() and (pa.call_info() == pr.call_info()).all()
    # window_div = 1: one image with the FULL LDS window (sessions with one instance per workgroup: small batches) — the program images
    # of the plan built in memory for the full window
    gsv.Plan.build_file("fq12_mix", units, a, window_div=1)
    pa, pr = gsv.Plan.load(a), gsv.Plan.from_circuit("fq12_mix", units)
    assert pa.info == pr.info and pa.image_bytes() == pr.image_bytes() and (pa.call_info() == pr.call_info()).all() and pa.wire_file() == pr.wire_file()
    with pytest.raises(ValueError):
        gsv.Plan.build_file("fq12_mix", units, a, window_div=3)
    with pytest.raises(gsv.GsvError):
        gsv.Plan.build_file("fq12_mix", units, os.path.join(str(tmp_path), "no_such_dir", "x.gsvplan"))


def test_traffic_counters_are_bound_to_the_engine_sources():
    """profiles/*_final/traffic.json is quoted by bench.py only for the engine it was measured on: by the library's sha256 (the same file)
    or by the sha256 of the engine's sources (the same tree built on another machine: build paths are compiled into the library).  The
    source hash depends on relative paths and contents only; the committed file of this round carries both."""
    import json
    from garbled_snark_verifier_amd import build as b
    h = b.source_sha256()
    assert len(h) == 64 and h == b.source_sha256()
    finals = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.endswith("_final") and os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json")))
    tj = json.load(open(os.path.join(ROOT, "profiles", finals[-1], "traffic.json")))
    assert len(tj["engine_library_sha256"]) == 64 and len(tj["engine_source_sha256"]) == 64 and tj["hbm_bytes_per_launch"] > 0


def test_mac_workers_are_shared_among_the_ranks_of_a_node():
    """A lone rank leaves the drain's worker count to the engine; the ranks of a node split the container's CPU quota (or the visible CPUs)."""
    import bench
    assert bench.mac_threads_for_rank(0, 1, quota=16.0) == 0
    assert bench.mac_threads_for_rank(5, 8, quota=16.0) == 5        # an explicit --mac-threads wins
    assert bench.mac_threads_for_rank(0, 8, quota=16.0) == 2
    assert bench.mac_threads_for_rank(0, 8, quota=4.0) == 1         # never zero workers
    assert bench.mac_threads_for_rank(0, 4, quota=None, visible=96) == 24


def test_dry_run_prints_the_per_rank_budget():
    """`bench.py --gpus 8 --workload cc16 --dry-run` (no GPU, no process group): two instances per rank, device memory well inside 288 GB,
    one all-gather, and the honest expectation — config 5 as stated is flat in N; the weak-scaling headline is linear and its
    with-commitment form becomes host-bound once N GPUs' PCIe streams outrun the quota's CBC-MAC capacity."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "cc16", "--dry-run"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["dry_run"] and j["instances_per_rank"] == [2] * 8 and j["per_rank"]["device_memory_gb"] < 200 and "all-gather of 16 commit records" in j["exchange"]
    assert 25 < j["expected"]["seconds_per_job"] < 35 and "flat in N" in j["expected"]["why"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True, text=True)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["instances_total"] == 8192 and j["per_rank"]["device_memory_gb"] < 288 and j["expected"]["value_gates_per_s"] > 8e11
    assert set(j["expected"]["host_bound_from_n_gpus"]) == {"aes-ni, 4 chains", "vaes, 16 chains"}


def test_bench_reexports_its_support_module():
    """bench.py keeps the driver's contract (the workloads, `main`); what they share lives in bench_support.py and stays reachable as
    `bench.<name>` (tools/ and the tests use it that way)."""
    import bench
    import bench_support
    for name in ("plan_slices", "session_slices", "instance_seeds", "Dist", "timed_steps", "VerifierWork", "get_plan", "mac_threads_for_rank", "dry_run", "cpu_baseline",
                 "VERIFIER_UNITS", "SMALL_BATCH_UNITS", "FIXTURE", "garble_then_evaluate", "cc16_one_gpu", "measure_aes_ceiling", "_plan_cache_path", "T_START"):
        assert getattr(bench, name) is getattr(bench_support, name), name
    assert all(hasattr(bench, f) for f in ("run_verifier", "run_cc16", "run_synthetic", "main"))
