"""Support code of bench.py (the driver's contract keeps bench.py at the repository root; this module holds what its three workloads share):
constants and unit lists, the rank logic that runs without a GPU (slices, seeds, the process group wrapper, the timed loop), the plan
file cache shared by the ranks of a node, host CPU discovery and the MAC worker split, `--dry-run`, the CPU baseline and the small
measurement legs.  Everything here is importable as `bench.<name>` too (bench.py re-exports it): tests/test_bench_logic.py and
tests/test_distributed_cpu.py exercise it at world sizes 2 and 8 on gloo."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

T_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VERIFIER_GATES = 11_174_708_821  # README.md:12 of the reference (its own 1-public-input key)
FIXTURE = {"verifier_compressed": "groth16_verify_compressed_1pub_golden.json",  # ONE public input: the reference's benchmark configuration
           "verifier_compressed_2pub": "groth16_verify_compressed_golden.json", "verifier": "groth16_verify_golden.json"}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s
AES_CEILING_AND_PER_S_R02 = 4.82e10  # round 2's measurement (profiles/r02_final/aes_forms.txt): only quoted when the micro-benchmark cannot run


def measure_aes_ceiling(log):
    """The T-table AES ceiling of THIS box, measured in this run, outside the timed region: tools/ubench/aes_forms (built by
    __graft_entry__.build()) runs the production cipher form alone with every CU full and prints blocks/s; a garbled AND is two blocks.
    Runs as a child process (it owns its HIP context).  Returns (ANDs per second, source string)."""
    exe = os.path.join(ROOT, "tools", "ubench", "aes_forms")
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
        for ln in out.splitlines():
            if ln.startswith("T-table"):
                blocks = float(ln.split(":")[1].split("blocks/s")[0])
                if "matches host AES" not in ln:
                    raise ValueError("micro-benchmark output did not validate: %s" % ln)
                return blocks / 2.0, "tools/ubench/aes_forms in this run: %.4g T-table AES blocks/s with every CU full" % blocks
        raise ValueError("no T-table line in %r" % out[-200:])
    except Exception as e:  # noqa: BLE001 - the ceiling is context, not the result
        log("bench.py: AES micro-benchmark did not run (%r): quoting round 2's ceiling" % (e,))
        return AES_CEILING_AND_PER_S_R02, "round-2 constant (the micro-benchmark did not run here: %r)" % (e,)


VERIFIER_UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::mul_by_034_montgomery",
                  "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery",
                  "bigint::multiplexer", "g1::add_montgomery",
                  # the Fq inversions (binary extended Euclid, fp254impl.rs:333-690).  As ONE unit an inversion would be 11.2 M ciphertexts — twice an Fq12
                  # multiplication's 5.4 M, which sets every instance's device ciphertext block — so it enters as three calls: two groups of 64
                  # `inverse_iteration` components (3.9 M ciphertexts each) and the two division chains together (3.2 M); these wrappers are
                  # component boundaries the reference does not have (stream-neutral, bn254_ext.hpp).  Round 4 entered the 318 four-iteration
                  # components themselves: 36 % more device steps (a call boundary ends the overlap of chained adders)
                  "inverse::iteration_group", "inverse::divide_chains"]


# The same circuit with the Fq12 multiplications and squarings entered one level finer, as their three fq6::mul_montgomery units (a component
# boundary the reference does not have: stream-neutral): 1 861 calls instead of 1 257 (round 4, before the inversions became three long calls: 3 751 / 3 147), more width for the call-level dataflow
# and 19 % more device steps.  ONE instance garbles 7.6 % faster, sixteen 8.7 % (profiles/r04_e2e/verifier_mixed_units.log); a full GPU pays
# for the extra steps.  bench.py builds this plan for its small-batch legs (--small-batch-units fq6, the default) beside the Fq12-level one.
SMALL_BATCH_UNITS = ["fq6::mul_montgomery"] + [u for u in VERIFIER_UNITS if u not in ("fq12::square_montgomery", "fq12::mul_montgomery")]


# ---------------------------------------------------------------------------------------------------------- rank logic (no GPU)
def plan_slices(call_gates, n_slices):
    """Cut calls 0..n-1 into `n_slices` consecutive groups of nearly equal gate count: [(first_call, n_calls, gates)]."""
    n = len(call_gates)
    n_slices = max(1, min(int(n_slices), n))
    cum = [0]
    for g in call_gates:
        cum.append(cum[-1] + int(g))
    total, bounds = cum[-1], [0]
    for k in range(1, n_slices):
        target = total * k / n_slices
        lo = bounds[-1] + 1
        hi = n - (n_slices - k)
        j = min(range(lo, hi + 1), key=lambda c: abs(cum[c] - target))
        bounds.append(j)
    bounds.append(n)
    return [(bounds[k], bounds[k + 1] - bounds[k], cum[bounds[k + 1]] - cum[bounds[k]]) for k in range(n_slices)]


def session_slices(windows, call_gates, n_slices):
    """The same for a plan SESSION: its schedule executes whole windows of consecutive calls (Session.windows(): [(first_call, n_calls,
    n_batches)]), so a slice starts and ends on window boundaries — `n_slices` groups of windows of nearly equal gate count."""
    wg = [sum(int(g) for g in call_gates[f:f + n]) for f, n, _ in windows]
    out = []
    for w0, nw, gates in plan_slices(wg, n_slices):
        out.append((windows[w0][0], sum(windows[w][1] for w in range(w0, w0 + nw)), gates))
    return out


def instance_seeds(rank, n):
    """Seeds of this rank's instances: disjoint between ranks (instance i of the job -> rank i mod world in a real run)."""
    return [1_000_003 * (rank + 1) + i for i in range(n)]


class Dist:
    """torch.distributed behind the three things the bench needs; world == 1 needs no process group."""

    def __init__(self, world, backend, device):
        self.world, self.device = world, device
        if world > 1:
            import datetime
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {"device_id": torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))} if backend == "nccl" else {}
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=30), **kw)
            self.dist = dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_float(self, v):
        if self.world == 1:
            return float(v)
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_int(self, v):
        if self.world == 1:
            return int(v)
        import torch
        t = torch.tensor([v], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def all_gather_records(self, local):
        """The path's one exchange: all-gather of the ranks' commit records.  Returns [world * B, rec_len]."""
        import torch
        t = torch.from_numpy(local)
        if self.world == 1:
            return t
        t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return torch.cat(out).cpu()

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed_steps(work, slices, warmup, steps, dist, sync, time_budget_s=None, t_start=None):
    """The contract's loop over a sliced pass: `warmup` untimed steps, then `steps` timed ones bracketed by sync + barrier,
    max over ranks.  work.new_pass() restarts the pass, work.run_slice(first, n) garbles one slice and returns its device
    milliseconds, work.commit_records() is called at the end of every pass and its records are all-gathered.
    Returns a dict with the elapsed time, the gates garbled per instance in the timed steps and per-step device times."""
    S = len(slices)
    gathered = {"table": None}

    def step(i):
        k = i % S
        if k == 0:
            work.new_pass()
        ms = work.run_slice(slices[k][0], slices[k][1])
        if k == S - 1:
            gathered["table"] = dist.all_gather_records(work.commit_records())
        return ms

    t0 = time.perf_counter()
    for i in range(warmup):
        step(i)
    sync(); dist.barrier()
    warm_s = time.perf_counter() - t0
    steps_run = steps
    if time_budget_s is not None:
        # projected duration of the timed steps from the warm-up's gate rate (no projection when there was no warm-up)
        wg = sum(slices[i % S][2] for i in range(warmup))
        rate = wg / warm_s if warmup and warm_s > 0 else None
        left = time_budget_s - (time.time() - (t_start or T_START))
        if rate:
            acc, fit = 0.0, 0
            for j in range(steps):
                acc += slices[(warmup + j) % S][2] / rate
                if acc > left:
                    break
                fit = j + 1
            steps_run = max(1, fit)
        steps_run = dist.min_int(steps_run)
    sync(); dist.barrier()
    t0 = time.perf_counter()
    ms = [step(warmup + j) for j in range(steps_run)]
    sync(); dist.barrier()
    elapsed = dist.max_float(time.perf_counter() - t0)
    gates = sum(slices[(warmup + j) % S][2] for j in range(steps_run))
    calls = sum(slices[(warmup + j) % S][1] for j in range(steps_run))
    return {"elapsed": elapsed, "steps_run": steps_run, "gates_per_instance": gates, "calls": calls, "step_ms": ms, "commit_table": gathered["table"]}


# ---------------------------------------------------------------------------------------------------------- GPU workload
class VerifierWork:
    """`B` instances of the verifier plan on one GPU (the object timed_steps drives).  seeds[i] seeds instance i."""

    def __init__(self, gsv, engine, plan, B, seeds, retain_stream=False, **session_kw):
        import numpy as np
        self.np, self.gsv, self.plan, self.B = np, gsv, plan, B
        n_in = plan.info["n_inputs"]
        self.delta = np.zeros((B, 16), np.uint8); self.consts = np.zeros((B, 2, 16), np.uint8); self.inputs = np.zeros((B, n_in, 16), np.uint8)
        for i, sd in enumerate(seeds):
            self.delta[i], self.consts[i, 0], self.consts[i, 1], self.inputs[i] = gsv.labels_from_seed(sd, n_in)
        self.sess = gsv.Session(engine, plan, B, retain_stream=retain_stream, **session_kw)  # False: windows; "ring": one launch over a ciphertext ring
        self.seeds = seeds
        self.ct_hashes = None  # set by a pass that drained and MAC'ed the stream

    def slices(self, call_gates, n_slices):
        """[(first_call, n_calls, gates)]: groups of whole windows of the session's schedule, nearly equal in gates."""
        return session_slices(self.sess.windows(), call_gates, n_slices)

    def new_pass(self):
        self.sess.set_garble_inputs(self.delta, self.consts, self.inputs)  # fresh labels resident in HBM before the first slice starts

    def run_slice(self, first, n):
        self.sess.garble_calls(first, n, discard=True)  # returns when the slice's last call has finished
        return self.sess.last_kernel_ms()

    def run_pass(self, commit=False, threads=0):
        """One whole pass; commit=True: every ciphertext drained over PCIe and folded into its instance's CBC-MAC.  Returns seconds."""
        self.new_pass()
        t0 = time.perf_counter()
        if commit:
            self.ct_hashes = self.sess.garble_streaming(threads=threads)
        else:
            self.sess.garble_streaming(discard=True)
        return time.perf_counter() - t0

    def commit_records(self):
        """GarbledInstanceCommit per instance (cut_and_choose/garbler.rs:61-99): label commits of inputs, outputs and constants, and
        the ciphertext commitment when the pass drained the stream (zero when the ciphertexts stayed in HBM: the timed headline)."""
        from garbled_snark_verifier_amd import sharding
        out = self.sess.read_outputs()
        ch = self.ct_hashes or [bytes(16)] * self.B
        return self.np.stack([sharding.commit_record(self.seeds[i], ch[i], out[i], self.delta[i], self.consts[i, 0], self.consts[i, 1], self.inputs[i]) for i in range(self.B)])

    def close(self):
        self.sess.close()


def _plan_cache_path(args, circuit, units, window_div=4):
    if args.no_plan_cache:
        return None
    if os.environ.get("GSV_PLAN_FILE"):  # experiments: one plan file for several engine builds (the key below includes the library)
        return os.environ["GSV_PLAN_FILE"]
    import garbled_snark_verifier_amd.build as b
    h = hashlib.sha256()
    with open(b.build(), "rb") as f:
        h.update(f.read())  # the file format and the compiler live in the library: any rebuild invalidates the cache
    h.update(("|".join([circuit, ",".join(units), "window/%d" % window_div])).encode())
    name = "plan_%s.gsvplan" % h.hexdigest()[:24]
    cands = [args.plan_cache] if args.plan_cache else [os.environ.get("GSV_PLAN_CACHE"), "/dev/shm", "/tmp"]
    for d in cands:
        if not d or not os.path.isdir(d):
            continue
        sub = os.path.join(d, "gsv_plan_cache_%d" % os.getuid()) if d in ("/dev/shm", "/tmp") else d
        if os.path.exists(os.path.join(sub, name)) and os.stat(sub).st_uid == os.getuid() and not (os.stat(sub).st_mode & 0o022):
            return os.path.join(sub, name)
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < 60e9:  # the verifier plan's images are ~40 GB
                continue
            os.makedirs(sub, mode=0o700, exist_ok=True)
            ds = os.stat(sub)
            if ds.st_uid != os.getuid() or (ds.st_mode & 0o022):  # a plan file is trusted input of the loader: only our own directory
                continue
            return os.path.join(sub, name)
        except OSError:
            continue
    return None


def get_plan(gsv, engine, args, circuit, units, rank, local_rank, local_world, dist, log, window_div=4):
    """Local rank 0 loads the node's plan file or builds the plan straight into that file; every rank of the node then streams the
    file into its GPU's memory (without a cache directory a single rank builds the plan in memory).  Returns (plan, {how, seconds, ...}, save_later).  A failure on any rank is
    agreed on by all (min over ranks) before anyone waits in a barrier: every rank exits non-zero together."""
    path = _plan_cache_path(args, circuit, units, window_div)
    t0 = time.time()
    info = {"cache_file": path, "window_div": window_div}
    plan, save_later, err = None, None, None
    if local_rank == 0:
        try:
            if path and os.path.exists(path):
                plan = gsv.Plan.load(path, engine)
                info["how"] = "loaded"
            else:
                if path:
                    # straight into the node's plan file: each program is written by the worker that compiled it and dropped, so the
                    # host never holds the 41 GB of records (gsv_plan_build_file: ~25 GB peak instead of ~54 GB); then every rank —
                    # this one too — streams the file into its GPU's memory
                    t1 = time.time()
                    gsv.Plan.build_file(circuit, units, path, window_div=window_div)  # one image per program: 4 = good for 1, 2 and 4 instances per workgroup
                    info["build_s"] = time.time() - t1
                    plan = gsv.Plan.load(path, engine)
                    info["how"] = "built to file, loaded"
                else:
                    if local_world > 1:
                        raise RuntimeError("no directory with room for the plan file the other ranks load (set --plan-cache)")
                    plan = gsv.Plan.from_circuit(circuit, units, window_div=window_div)  # no directory for a plan file: ~54 GB of host memory
                    info["how"] = "built"
        except Exception as e:  # noqa: BLE001
            err = e
    if dist.min_int(0 if err else 1) == 0:
        raise RuntimeError("plan build / load failed on a rank: %r" % (err,))
    dist.barrier()
    if local_rank != 0:
        try:
            plan = gsv.Plan.load(path, engine)
            info["how"] = "loaded"
        except Exception as e:  # noqa: BLE001
            err = e
    if dist.min_int(0 if err else 1) == 0:
        raise RuntimeError("plan load failed on a rank: %r" % (err,))
    info["seconds"] = time.time() - t0
    return plan, info, save_later


def physical_cores():
    """One logical CPU per physical core of this process's affinity mask (the reference pins one garbling task per physical core,
    cut_and_choose/mod.rs:131-186)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    seen, picks = set(), []
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            picks.append(c)
    return picks


def cpu_quota_cores():
    """CPU bandwidth limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def mac_threads_for_rank(requested, local_world, quota=None, visible=None):
    """Host MAC workers of ONE rank's drain.  A lone rank leaves it to the engine (0: up to 32 workers, one per chain group).  The ranks of a
    node share the host: each takes its share of the container's CPU quota (or of the visible CPUs), at least one — eight ranks with 32
    workers each on a 16-core quota would spend the quota on context switches instead of CBC-MAC chains."""
    if requested:
        return int(requested)
    if local_world <= 1:
        return 0
    cores = quota if quota is not None else cpu_quota_cores()
    if cores is None:
        cores = float(visible if visible is not None else (os.cpu_count() or 1))
    return max(1, int(cores // local_world))


def dry_run(args):
    """`--dry-run`: what `--gpus N` of this workload will ask of every rank — device memory, host memory, host cores — and what it is expected
    to deliver, from the plan's known sizes and this host's CPU quota.  No GPU, no process group: run it on the login shell of an 8-GPU node
    before the job.  The numbers marked `measured` come from one-GPU runs (profiles/r05_final/bench_driver_command.json)."""
    world = max(1, args.gpus)
    quota = cpu_quota_cores()
    visible = os.cpu_count() or 1
    cores = quota if quota is not None else float(visible)
    gates, n_ct = 11_456_865_898, 2_980_165_547           # the restated circuit, one public input (tests/golden/groth16_verify_compressed_1pub_golden.json)
    f_nf = n_ct / gates
    image_gb = {"headline (Fq12-level units, quarter LDS window)": 41.8, "small-batch (Fq6-level units, full window)": 41.6}
    hbm_gb, pcie_gbs = 288.0, 54.4                         # MI355X; measured D2H rate of the drain (54-57 GB/s in 16 MiB chunks)
    mac_core = {"aes-ni, 4 chains": 4.5e8, "vaes, 16 chains": 1.7e9}  # measured CBC-MAC blocks/s of one host core
    threads = mac_threads_for_rank(args.mac_threads, world, quota, visible)
    cc16 = args.workload == "cc16"
    total = 16 if cc16 else args.instances * world
    per_rank = [len(range(r, total, world)) for r in range(world)] if cc16 else [args.instances] * world
    B = max(per_rank)
    wire_mb, ring_gb_one = (19.1, 3.2) if not cc16 else (23.3, 3.2)
    if cc16:
        ct_block_gb = min(0.4 * (hbm_gb - 41.6), 48.0)    # the window block of a session that does not retain the stream (engine.cpp make_schedule)
        dev_gb = 41.6 + B * wire_mb / 1e3 + ct_block_gb + 3 * B * 1.07  # + three gate-order buffers of <= 1 GB per instance
        one_instance_s, sixteen_s = 29.8, 32.1             # measured: one / sixteen instances with commitments on one GPU
        job_s = one_instance_s + (sixteen_s - one_instance_s) * (B - 1) / 15.0
        expected = {"seconds_per_job": round(job_s, 1), "gates_per_s": gates * total / job_s,
                    "why": "an instance is ~7.3 M dependent device steps (~27 s) however many GPUs there are: sixteen instances on ONE GPU take %.1f s, %d per GPU ~%.1f s — config 5 as stated is flat in N" % (sixteen_s, B, job_s)}
    else:
        dev_gb = 41.8 + B * wire_mb / 1e3 + B * 88.3 / 1e3
        per_gpu = 1.05e11                                  # measured: 1 024 instances per GPU, ciphertexts into HBM
        pcie_bound = pcie_gbs * 1e9 / 16 / f_nf
        mac_bound = {k: cores * v / f_nf for k, v in mac_core.items()}
        expected = {"value_gates_per_s": per_gpu * world, "scaling": "weak: no collective and no host work in the timed region (one barrier per step): linear in N",
                    "with_commitment_gates_per_s": {k: min(world * pcie_bound, v) for k, v in mac_bound.items()},
                    "with_commitment_bound": {k: ("pcie (%.3g per GPU)" % pcie_bound if world * pcie_bound <= v else "host MAC: %.0f cores x %.2g blocks/s" % (cores, mac_core[k])) for k, v in mac_bound.items()},
                    "host_bound_from_n_gpus": {k: int(v // pcie_bound) + 1 for k, v in mac_bound.items()}}
    out = {"dry_run": True, "workload": args.workload, "n_gpus": world, "instances_total": total, "instances_per_rank": per_rank,
           "host": {"cpu_quota_cores": quota, "visible_cpus": visible, "mac_workers_per_rank": threads if threads else "engine default (<= 32)", "mac_workers_total": (threads or 32) * world},
           "per_rank": {"device_memory_gb": round(dev_gb, 1), "of_hbm_gb": hbm_gb, "plan_image_gb": image_gb, "wire_file_mb_per_instance": wire_mb,
                        "host_rss_gb": {"rank that builds the plan file (local rank 0, once per machine)": 17.0, "ranks that load it": 2.0},
                        "plan_file": "one file per node in /dev/shm (41.8 GB, page cache shared): local rank 0 builds (~50 s), the others gsv_plan_load it into their GPU (5-11 s)",
                        "pinned_host_buffers_gb": round((threads or 32) * 2 * (16 if B >= 128 else 4 if B > cores else 1) * 16 / 1024.0, 2)},
           "exchange": "one all-gather of %d commit records x 48 952 B (RCCL over xGMI), nothing else" % total,
           "expected": expected}
    print(json.dumps(out), flush=True)


_CPU_WORKER = """
import json, os, sys
sys.path.insert(0, %r)
cpu = int(sys.argv[1])
try:
    os.sched_setaffinity(0, {cpu})
except (AttributeError, OSError):
    pass
import oracle_lib as o
sec, gates, h = o.bench_garble_prefix(%r, %d, seed=%d)
print(json.dumps({"seconds": sec, "gates": gates, "hash": h.hex()}))
"""


def cpu_baseline(o, circuit, seed, log, prefix_gates=400_000_000, budget_s=75.0):
    """The restated CPU path (C++ oracle: AES-NI gate hash + inline CBC-MAC, the reference's per-gate loop, garble_mode.rs:160-222) on
    this host, timed on a PREFIX of the very stream the GPU garbles (the first `prefix_gates` gates of the verifier: decompression
    ladders, i.e. Fq multiplications like the rest of the circuit): one core, then one instance per physical core, each in its own
    process pinned to its core (reference: one garbling task per physical core, cut_and_choose/mod.rs:131-186)."""
    t0 = time.time()
    one_s, one_g, one_h = o.bench_garble_prefix(circuit, prefix_gates, seed=seed)
    out = {"value": one_g / one_s, "unit": "gates/s", "cores": 1, "kind": "port",
           "sample": "the first %d gates of the benchmarked circuit's own stream (seed %d) garbled by the C++ restatement of the reference's loop (AES-NI hash, inline CBC-MAC)" % (one_g, seed),
           "cpu_1core": {"value": one_g / one_s, "unit": "gates/s", "cores": 1, "seconds": one_s, "gates": one_g},
           "reference_published": {"cpu_1core": 32e6, "cpu_8cores": 249e6, "source": "README.md:12-13 of the reference (developer laptop, whole circuit incl. gadget code); this port: same loop, this host's core"}}
    cpus = physical_cores()
    quota = cpu_quota_cores()
    out["host"] = {"physical_cores_in_affinity_mask": len(cpus), "cgroup_cpu_quota_cores": quota}
    if quota is not None and quota < len(cpus):  # more processes than the container may run at once would only time-slice
        cpus = cpus[: max(1, int(quota))]
    code = _CPU_WORKER % (os.path.join(ROOT, "tests"), circuit, prefix_gates, seed)
    t1 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(c)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in cpus]
    done, ok = [], True
    for p in procs:
        try:
            so, _ = p.communicate(timeout=max(1.0, budget_s - (time.perf_counter() - t1)))
            r = json.loads(so.strip().splitlines()[-1])
            ok = ok and r["hash"] == one_h.hex()  # same seed -> same MAC state after the prefix on every core
            done.append(r)
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            p.kill()
            ok = False
    wall = time.perf_counter() - t1
    if done:
        g = sum(r["gates"] for r in done)
        out["cpu_allcores"] = {"value": g / wall, "unit": "gates/s", "cores": len(done), "seconds": wall, "gates": g, "hashes_equal_single_core": bool(ok),
                               "per_core_rate_mean": sum(r["gates"] / r["seconds"] for r in done) / len(done)}
        out.update({"value": g / wall, "cores": len(done)})
        out["sample"] += "; all-cores leg: one process per physical core on %d cores, %.1f s wall (process start included)" % (len(done), wall)
    out["seconds_total"] = time.time() - t0
    return out


def mode_rates(gsv, engine, np, instances=256, replays=4):
    """Rows a9 / f4 of SURVEY.md §8 on a component chain (fq12_sqmul replayed): device rates of EvaluateMode (one AES per AND, the
    ciphertexts read back where the garbler left them: evaluate_mode.rs:123-158) and of the Blake3Hasher PRF (hashers/mod.rs:22-51)
    beside the AES garble rate of the same session shape.  The evaluated output labels must be select(label0, bit)."""
    prog = gsv.Program.from_circuit("fq12_sqmul", chain_feedback=True)
    n_in, gates = prog.info["n_inputs"], prog.info["n_gates"] * replays
    labs = [gsv.labels_from_seed(7000 + i, n_in) for i in range(instances)]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    out = {"circuit": "fq12_sqmul chain x%d (%d gates per instance), %d instances" % (replays, gates, instances)}
    for name in ("aes", "blake3"):
        sess = gsv.Session(engine, prog, instances, replays, replays)
        sess.set_hasher(name)
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble(0); sess.sync()
        out["garble_%s" % name] = instances * gates / (sess.last_kernel_ms() / 1e3)
        label0 = sess.read_outputs()
        bits = np.random.default_rng(1).integers(0, 2, size=(instances, n_in)).astype(np.uint8)
        active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
        sess.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
        sess.evaluate(0); sess.sync()
        out["evaluate_%s" % name] = instances * gates / (sess.last_kernel_ms() / 1e3)
        oa, ob = sess.read_outputs(with_bits=True)
        out["evaluate_%s_consistent" % name] = bool((oa == np.where(ob[:, :, None] == 1, label0 ^ delta[:, None, :], label0)).all())
        sess.close()
    out["unit"] = "gates/s"
    return out


def cc16_verifier_fixture(case):
    """tests/golden/cc16_verifier_golden.json (the 16 instances of master seed 1234 on the full verifier, garbled by the CPU oracle), or
    None when it does not belong to this circuit."""
    path = os.path.join(ROOT, "tests", "golden", "cc16_verifier_golden.json")
    if not os.path.exists(path):
        return None
    g = json.load(open(path))
    return g if g.get("gates") == case["gates"] and g.get("n_ciphertexts") == case["n_ciphertexts"] else None


def cc16_one_gpu(gsv, engine, plan, case, gold, log):
    """BASELINE config 5 with all 16 instances on THIS GPU: sharding.cut_and_choose_commit (Garbler::create -> commit,
    cut_and_choose/garbler.rs:191-257) on the full verifier — 16 seeds from master seed 1234, every instance garbled WITH its ciphertext
    commitment (stream drained over PCIe, sixteen serial CBC-MAC chains on the host), GarbledInstanceCommit records built — and every
    record compared with the one built from the CPU oracle's flat garbling of the same seed."""
    from garbled_snark_verifier_amd import sharding
    t0 = time.perf_counter()
    table, seeds = sharding.cut_and_choose_commit(case["circuit"], gold["master_seed"], gold["total"], 0, 1, engine=engine, program=plan)
    dt = time.perf_counter() - t0
    rec_ok = [hashlib.sha256(r.tobytes()).hexdigest() == gold["record_sha256"][i] for i, r in enumerate(table)]
    out = {"seconds": dt, "gates_per_s": case["gates"] * gold["total"] / dt, "instances": gold["total"], "master_seed": gold["master_seed"],
           "all_16_match": bool(all(rec_ok) and [int(x) for x in seeds] == gold["seeds"]), "records_matching": int(sum(rec_ok)),
           "table_sha256_match": hashlib.sha256(table.tobytes()).hexdigest() == gold["table_sha256"],
           "reference_published": "16 instances on 8 physical cores: ~11 m 58 s (README.md:13)",
           "sample": "the whole job: 16 x %d gates garbled, 16 x %d ciphertexts drained and CBC-MAC'ed, 16 commit records; fixture tests/golden/cc16_verifier_golden.json" % (case["gates"], case["n_ciphertexts"])}
    log("bench.py: cc16 on one GPU: %.1f s, %.3g gates/s, %d of 16 records match the oracle's" % (dt, out["gates_per_s"], sum(rec_ok)))
    return out


def garble_then_evaluate(gsv, engine, plan, case, np):
    """gsv_session_garble_evaluate on the whole verifier, one instance: window k of the garbler's device block is evaluated on a second
    stream while window k+1 is garbled.  The evaluator holds the valid proof's input bits: the decoded output must be 1 and every active
    output label select(label0, bit)."""
    n_in = plan.info["n_inputs"]
    d, f, t, inp = gsv.labels_from_seed(case["seed"], n_in)
    bits = np.unpackbits(np.frombuffer(bytes.fromhex(case["input_bits_hex"]), np.uint8), bitorder="little")[:n_in].astype(np.uint8)
    active = np.where(bits[:, None] == 1, inp ^ d[None, :], inp)
    # windows of 4 GB: the evaluation of window k overlaps the garbling of window k+1, so the pair wants MORE windows than a garbler alone
    # (whose default is two: the scope in which its call chains overlap) — twelve here, the last one's evaluation is the tail
    kw = dict(retain_stream=False, window_ct_records=1 << 28)
    gs, es = gsv.Session(engine, plan, 1, **kw), gsv.Session(engine, plan, 1, **kw)
    try:
        gs.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
        es.set_evaluate_inputs(np.stack([f, t ^ d])[None], active[None], bits[None])
        t0 = time.perf_counter()
        gs.garble_evaluate(es)
        dt = time.perf_counter() - t0
        out0 = gs.read_outputs()[0]
        oa, ob = es.read_outputs(with_bits=True)
        ok = bool(ob[0][0] == case.get("expected_output", 1) and (oa[0] == np.where(ob[0][:, None] == 1, out0 ^ d[None, :], out0)).all()
                  and out0[0].tobytes().hex() == case["first_output_label0"])
        g = plan.info["n_gates"]
        return {"seconds": dt, "gates_per_s_garbled": g / dt, "gates_per_s_garbled_plus_evaluated": 2 * g / dt, "instances": 1, "decoded_output": int(ob[0][0]),
                "labels_consistent_and_output_label_matches_fixture": ok, "windows": gs.schedule_info()["n_windows"],
                "sample": "one whole pass: %d gates garbled and the same %d gates evaluated side by side on the device, retain_stream = 0" % (g, g)}
    finally:
        gs.close(); es.close()
