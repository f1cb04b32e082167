#!/usr/bin/env python3
"""Headline benchmark: gates/s (garble) on the Groth16/BN254 verifier circuit, one rank per GPU.

Workload (BASELINE.json configs[3]): the restated `groth16_verify_compressed` circuit (reference: src/gadgets/groth16.rs:250-268,
the circuit src/garbled_groth16.rs garbles and examples/groth16_garble.rs:116-129 times): point decompression, window-10 MSM,
Miller loop, final exponentiation, comparison — 11,687,200,297 gates per instance for the synthetic 2-public-input verifying
key of tests/groth16_ref.py (the reference quotes 11,174,708,821 for its 1-public-input key; DESIGN.md §2 has the
component-by-component reconciliation), recorded as a plan of component programs and garbled exactly as the reference streams
it.  Every GPU garbles `--instances` (default 1024 = four per workgroup on 256 CUs) independent cut-and-choose instances (own seed => own
delta / labels / ciphertext stream).

A "step" is one SLICE of that pass: the plan's calls are cut into `--slices` consecutive groups of (nearly) equal gate count and
step i garbles slice i mod slices for all instances of the rank — wires, gate ids and the ciphertext stream continue from step
to step, so `--slices` consecutive steps are exactly one full verifier pass per instance (a whole pass is ~130 s at 1024
instances, which no driver budget fits 25 times).  `value` = gates garbled by all ranks in the timed steps / elapsed, with the
ciphertexts produced into HBM (one call block per instance, overwritten by the next call: inputs and outputs resident in HBM).
The PCIe-inclusive rate — every ciphertext copied out and folded into the per-instance CBC-MAC commitment
(src/ciphertext_hasher.rs:23-29), as the reference's timed garble does — is measured in the same run and reported beside it as
`e2e_with_commitment`; it is never `value`.

Before the timed loop rank 0 garbles ONE instance of the whole circuit through the streaming path and checks the CBC-MAC of
the full ciphertext stream and the output label against the fixture the CPU oracle produced from the flat stream
(`ciphertext_hash_match`).

`--gpus N` without a launcher spawns the N ranks itself (python -m torch.distributed.run) before anything touches a GPU.
"""
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
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s
AES_CEILING_AND_PER_S = 4.82e10  # tools/ubench/aes_forms.hip on MI355X (profiles/r02_final/aes_forms.txt): 9.65e10 T-table AES blocks/s with every CU full, two blocks per garbled AND

VERIFIER_UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::mul_by_034_montgomery",
                  "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery",
                  "bigint::multiplexer", "g1::add_montgomery",
                  # the Fq inversions (binary extended Euclid, fp254impl.rs:333-690) enter as their own 4-iteration components: as ONE unit an
                  # inversion (11 M ciphertexts) — or the Fq12 inversion around it (21 M) — would set the size of every instance's device
                  # ciphertext block (340 MB x 512 instances); their chunks keep the largest block at an Fq12 multiplication's 5.4 M records
                  "inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"]


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
    """`B` instances of the verifier plan on one GPU (the object timed_steps drives)."""

    def __init__(self, gsv, engine, plan, B, seeds):
        import numpy as np
        self.np, self.gsv, self.plan, self.B = np, gsv, plan, B
        n_in = plan.info["n_inputs"]
        self.delta = np.zeros((B, 16), np.uint8); self.consts = np.zeros((B, 2, 16), np.uint8); self.inputs = np.zeros((B, n_in, 16), np.uint8)
        for i, sd in enumerate(seeds):
            self.delta[i], self.consts[i, 0], self.consts[i, 1], self.inputs[i] = gsv.labels_from_seed(sd, n_in)
        self.sess = gsv.Session(engine, plan, B, retain_stream=False)
        self.seeds = seeds

    def new_pass(self):
        self.sess.set_garble_inputs(self.delta, self.consts, self.inputs)  # fresh labels resident in HBM before the first slice starts

    def run_slice(self, first, n):
        self.sess.garble_calls(first, n, discard=True)  # returns when the slice's last call has finished
        return self.sess.last_kernel_ms()

    def commit_records(self):
        """GarbledInstanceCommit per instance (cut_and_choose/garbler.rs:61-99): label commits of inputs, outputs and constants.  The
        ciphertext-commit field is zero here: the timed step keeps the ciphertexts in HBM (the streamed + hashed path fills it)."""
        from garbled_snark_verifier_amd import sharding
        out = self.sess.read_outputs()
        return self.np.stack([sharding.commit_record(self.seeds[i], bytes(16), out[i], self.delta[i], self.consts[i, 0], self.consts[i, 1], self.inputs[i]) for i in range(self.B)])

    def close(self):
        self.sess.close()


def _plan_cache_path(args, circuit, units):
    if args.no_plan_cache:
        return None
    if os.environ.get("GSV_PLAN_FILE"):  # experiments: one plan file for several engine builds (the key below includes the library)
        return os.environ["GSV_PLAN_FILE"]
    import garbled_snark_verifier_amd.build as b
    h = hashlib.sha256()
    with open(b.build(), "rb") as f:
        h.update(f.read())  # the file format and the compiler live in the library: any rebuild invalidates the cache
    h.update(("|".join([circuit, ",".join(units), "window/4"])).encode())
    name = "plan_%s.gsvplan" % h.hexdigest()[:24]
    cands = [args.plan_cache] if args.plan_cache else [os.environ.get("GSV_PLAN_CACHE"), "/dev/shm", "/tmp"]
    for d in cands:
        if not d or not os.path.isdir(d):
            continue
        sub = os.path.join(d, "gsv_plan_cache_%d" % os.getuid()) if d in ("/dev/shm", "/tmp") else d
        if os.path.exists(os.path.join(sub, name)):
            return os.path.join(sub, name)
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < 60e9:  # the verifier plan's images are ~40 GB
                continue
            os.makedirs(sub, exist_ok=True)
            return os.path.join(sub, name)
        except OSError:
            continue
    return None


def get_plan(gsv, engine, args, circuit, units, rank, local_rank, local_world, dist, log):
    """Local rank 0 loads the node's plan file or builds the plan (and saves it when other ranks need it); the other ranks of the
    node load the file straight into their GPU's memory.  Returns (plan, {how, seconds, ...}, save_later)."""
    path = _plan_cache_path(args, circuit, units)
    t0 = time.time()
    info = {"cache_file": path}
    plan, save_later = None, None
    if local_rank == 0:
        if path and os.path.exists(path):
            plan = gsv.Plan.load(path, engine)
            info["how"] = "loaded"
        else:
            try:  # ~50 GB of host memory while the plan is built
                avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
                if avail_gb < 60:
                    log("bench.py: %.0f GB of host memory available, the plan build needs ~50 GB" % avail_gb)
            except (OSError, IndexError, ValueError):
                pass
            plan = gsv.Plan.from_circuit(circuit, units, window_div=4)  # one image per program, good for 1, 2 and 4 instances per workgroup
            info["how"] = "built"
            if local_world > 1:
                if not path:
                    raise RuntimeError("no directory with room for the plan file the other ranks load (set --plan-cache)")
                plan.save(path)
                info["saved_s"] = time.time() - t0
            elif path:
                save_later = path  # single rank: written after the result line, for the next process on this machine
    dist.barrier()
    if local_rank != 0:
        plan = gsv.Plan.load(path, engine)
        info["how"] = "loaded"
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


_CPU_WORKER = """
import json, os, sys
sys.path.insert(0, %r)
cpu = int(sys.argv[1])
try:
    os.sched_setaffinity(0, {cpu})
except (AttributeError, OSError):
    pass
import oracle_lib as o
res = [o.bench_garble(s, seed=0) for s in %r]
print(json.dumps({"seconds": sum(r[0] for r in res), "gates": sum(r[1] for r in res), "hashes": [r[2].hex() for r in res]}))
"""


def cpu_baseline(np, o, log, budget_s=75.0):
    """The restated CPU path (C++ oracle: AES-NI gate hash + inline CBC-MAC, the reference's per-gate loop) on this host: one core,
    then one instance per physical core, each in its own process pinned to its core (reference: one garbling task per physical
    core, cut_and_choose/mod.rs:131-186).  Sample per core: verifier components, ~0.36 B gates (~15 s)."""
    specs = ["g1_scalar_mul:10", "fq12_sqmul_chain:4"]  # the MSM's window scalar multiplication + 4 square-and-multiply links of the pairing core
    t0 = time.time()
    ref = [o.bench_garble(s, seed=0) for s in specs]
    one_s, one_g = sum(r[0] for r in ref), sum(r[1] for r in ref)
    out = {"value": one_g / one_s, "unit": "gates/s", "cores": 1, "kind": "port",
           "sample": "%s garbled back to back by the C++ restatement of the reference's loop (AES-NI hash, inline CBC-MAC): %d gates per core" % (" + ".join(specs), one_g),
           "cpu_1core": {"value": one_g / one_s, "unit": "gates/s", "cores": 1, "seconds": one_s, "gates": one_g},
           "reference_published": {"cpu_1core": 32e6, "cpu_8cores": 249e6, "source": "README.md:12-13 of the reference (developer laptop)"}}
    cpus = physical_cores()
    quota = cpu_quota_cores()
    out["host"] = {"physical_cores_in_affinity_mask": len(cpus), "cgroup_cpu_quota_cores": quota}
    if quota is not None and quota < len(cpus):  # more processes than the container may run at once would only time-slice
        cpus = cpus[: max(1, int(quota))]
    code = _CPU_WORKER % (os.path.join(ROOT, "tests"), specs)
    t1 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(c)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in cpus]
    done, ok = [], True
    for p in procs:
        try:
            so, _ = p.communicate(timeout=max(1.0, budget_s - (time.perf_counter() - t1)))
            r = json.loads(so.strip().splitlines()[-1])
            ok = ok and r["hashes"] == [x[2].hex() for x in ref]  # same seed -> same ciphertext hashes on every core
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


def run_verifier(args):
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if world > 1:
        torch.cuda.set_device(local_rank)
    dist = Dist(world, "nccl", "cuda")
    import garbled_snark_verifier_amd as gsv

    def log(msg):
        print(msg, file=sys.stderr, flush=True)

    compressed = args.workload == "verifier_compressed"
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16_verify_compressed_golden.json" if compressed else "groth16_verify_golden.json")))
    units = VERIFIER_UNITS + (["fp254::exp_chunk"] if compressed else [])
    engine = gsv.Engine(local_rank)  # raises without a HIP device: no CPU fallback
    plan, plan_info, save_later = get_plan(gsv, engine, args, case["circuit"], units, rank, local_rank, local_world, dist, log)
    t_first_launch = time.time() - T_START
    B, n_in, gates = args.instances, plan.info["n_inputs"], plan.info["n_gates"]
    n_calls = plan.info["n_calls"]
    f_nf = plan.info["n_ciphertexts"] / gates
    bytes_per_gate = 64.0 + 16.0 * f_nf  # 16 B record + 2x16 B label reads + 16 B write + 16 B*f_nf ciphertext
    ci = plan.call_info()
    slices = plan_slices(ci[:, 1], args.slices)
    image_bytes, n_programs = plan.image_bytes()
    n_glob, n_prog_slots = plan.wire_file()
    max_block = int(ci[:, 3].max())
    if rank == 0:
        log("bench.py: plan %s in %.1f s (%d calls of %d programs, %.1f GB of program records), %d slices; per instance: wire file %.1f MB (%d + %d slots), ciphertext block %.1f MB"
            % (plan_info["how"], plan_info["seconds"], n_calls, n_programs, image_bytes / 1e9, len(slices), (n_glob + n_prog_slots) * 16 / 1e6, n_prog_slots, n_glob, max_block * 16 / 1e6))

    result = {}
    # ---- whole-stream check on the fixture's seed, BEFORE the timed loop: one instance, stream drained and hashed on the host
    if rank == 0 and not args.no_check:
        t0 = time.time()
        d, f, t, inp = gsv.labels_from_seed(case["seed"], n_in)
        chk = gsv.Session(engine, plan, 1, retain_stream=False)
        chk.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
        h = chk.garble_streaming()[0].hex()
        ok = h == case["ct_hash"] and hashlib.sha256(chk.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
        chk.close()
        result["ciphertext_hash_match"] = bool(ok)
        result["hash_check"] = {"circuit": "the whole circuit (%d gates, %d ciphertexts), seed %d, one instance through gsv_session_garble_streaming" % (gates, plan.info["n_ciphertexts"], case["seed"]),
                                "gpu": h, "oracle": case["ct_hash"], "seconds": time.time() - t0}
        log("bench.py: whole-stream hash check %s in %.1f s" % ("ok" if ok else "MISMATCH", time.time() - t0))
    # ---- CPU baseline (rank 0 at N = 1 only), also before the timed loop so that the result line follows the timing directly
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            import oracle_lib as o
            result["cpu_baseline"] = cpu_baseline(np, o, log)
            log("bench.py: cpu baseline %.3g gates/s on 1 core, %.3g on %d cores" % (result["cpu_baseline"]["cpu_1core"]["value"], result["cpu_baseline"]["value"], result["cpu_baseline"]["cores"]))
        except Exception as e:  # the baseline must not cost the run its result line
            result["cpu_baseline"] = {"error": repr(e)}

    # ---- PCIe-inclusive rate with the commitment: calls from the middle of the plan, every ciphertext copied out and folded into its
    # instance's CBC-MAC while the next call is garbled (correctness of this path: the check above).  In a session of its own, closed
    # before the timed one is created: the drain keeps a gate-order copy of the ciphertext blocks (twice their HBM).
    if rank == 0 and world == 1 and not args.no_e2e:
        try:
            Be = max(1, min(B, args.e2e_instances))
            e2e = VerifierWork(gsv, engine, plan, Be, instance_seeds(rank, Be))
            try:
                first = slices[len(slices) // 2][0]
                n, g_acc = 0, 0
                while first + n < n_calls and g_acc * Be < args.e2e_gates:
                    g_acc += int(ci[first + n, 1]); n += 1
                e2e.new_pass()
                t0 = time.perf_counter()
                e2e.sess.garble_calls(first, n, discard=False)
                dt = time.perf_counter() - t0
                ct = int(ci[first:first + n, 3].sum())
                result["e2e_with_commitment"] = {"value": g_acc * Be / dt, "unit": "gates/s", "instances": Be, "instances_per_workgroup": e2e.sess.instances_per_workgroup, "seconds": dt,
                                                 "ciphertext_gb_per_s": ct * Be * 16 / dt / 1e9,
                                                 "sample": "calls %d..%d of the plan (%d gates, %d ciphertexts per instance) for %d instances: garbled, drained over PCIe and CBC-MAC'ed per instance on the host (gsv_session_garble_streaming_calls)"
                                                           % (first, first + n - 1, g_acc, ct, Be)}
                log("bench.py: e2e with commitment %.3g gates/s (%.1f GB/s of ciphertexts)" % (g_acc * Be / dt, ct * Be * 16 / dt / 1e9))
            finally:
                e2e.close()
        except Exception as e:
            result["e2e_with_commitment"] = {"error": repr(e)}

    work = VerifierWork(gsv, engine, plan, B, instance_seeds(rank, B))
    ni = work.sess.instances_per_workgroup

    def sync():
        torch.cuda.synchronize()
        work.sess.sync()

    r = timed_steps(work, slices, args.warmup, args.steps, dist, sync, args.time_budget, T_START)
    work.close()
    if rank == 0:
        el, K = r["elapsed"], r["steps_run"]
        stream_s = sum(r["step_ms"]) / 1e3  # device time of the timed steps: HIP events on the engine's stream around every slice
        n_launch = r["calls"]
        g_rank = r["gates_per_instance"] * B
        achieved = g_rank * bytes_per_gate / stream_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_final", "traffic.json")
        if os.path.exists(tpath) and compressed:
            try:
                tj = json.load(open(tpath))
                if int(tj.get("instances_per_gpu", 512)) == B:  # PMC passes of this very configuration (tools/profile_r02.sh)
                    traffic = float(tj["hbm_bytes_per_launch"])
            except (KeyError, ValueError):
                pass
        result.update({
            "metric": "gates/sec (garble) on Groth16/BN254 verifier at 1/2/4/8 GPUs; ciphertext-hash match", "value": g_rank * world / el, "unit": "gates/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "restated %s circuit (synthetic 2-public-input verifying key / proof of tests/groth16_ref.py; %d gates per instance, the reference quotes 11,174,708,821 "
                                   "for its 1-public-input key: DESIGN.md §2), %d cut-and-choose instances per GPU; one step = one of %d slices of the plan's %d calls, %d consecutive steps = one "
                                   "full verifier pass per instance" % ("groth16_verify_compressed" if compressed else "groth16_verify", gates, B, len(slices), n_calls, len(slices)),
                       "instances_per_gpu": B, "instances_per_workgroup": ni, "gates_per_instance": gates, "nonfree_fraction": f_nf, "plan_calls": n_calls, "plan_programs": n_programs,
                       "slices_per_pass": len(slices), "gates_per_step_per_instance": [s[2] for s in slices], "steps_requested": args.steps,
                       "passes_timed": r["gates_per_instance"] / gates, "step_device_ms": [round(x, 1) for x in r["step_ms"]], "plan": plan_info, "plan_image_gb": image_bytes / 1e9, "seconds_to_first_launch": t_first_launch,
                       "host_peak_rss_gb": __import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6},
            "commit_records_gathered": None if r["commit_table"] is None else list(r["commit_table"].shape),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # one step = the launches of one slice, all of the same kernel over different component programs: averages per launch
                         "kernel": "run_program_kernel<false, %d, 0>" % ni, "launches_timed": n_launch, "kernel_ms_avg": stream_s * 1e3 / max(1, n_launch),
                         "algorithmic_bytes_per_launch": g_rank * bytes_per_gate / max(1, n_launch), "bytes_per_gate": bytes_per_gate,
                         "note": "algorithmic-bytes accounting of SURVEY.md §8(d); fusion and the LDS label window keep most of those bytes off HBM, the limit that binds is T-table AES issue (DESIGN.md §3)",
                         "binding_limit": "aes-issue", "aes_ceiling_gates_per_s": AES_CEILING_AND_PER_S / f_nf, "aes_ceiling_frac": (g_rank / stream_s) / (AES_CEILING_AND_PER_S / f_nf)},
        })
        print(json.dumps(result), flush=True)
        if save_later:  # after the result line: the next process on this machine starts from the file
            try:
                t0 = time.time()
                plan.save(save_later)
                log("bench.py: plan saved to %s in %.1f s" % (save_later, time.time() - t0))
            except Exception as e:
                log("bench.py: plan not saved: %r" % (e,))
    dist.barrier()
    dist.close()
    plan.close()
    engine.close()


def run_synthetic(args):
    """--workload synthetic: the Groth16-SHAPED chain of SURVEY.md §8(d) the engine was tuned on in round 1 (330 Fq12 square-and-multiply
    links = 11.18 B gates per instance, one compiled program replayed with a ciphertext ring); a step is the whole chain."""
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(local_rank)
    dist = Dist(world, "nccl", "cuda")
    import garbled_snark_verifier_amd as gsv
    engine = gsv.Engine(local_rank)
    prog = gsv.Program.from_circuit(args.component, chain_feedback=True)
    info = prog.info
    gpr = info["n_gates"]
    replays = args.replays or -(-VERIFIER_GATES // gpr)
    B, n_in = args.instances, info["n_inputs"]
    f_nf = info["n_ciphertexts"] / gpr
    bytes_per_gate = 64.0 + 16.0 * f_nf
    seeds = instance_seeds(rank, B)
    delta = np.zeros((B, 16), np.uint8); consts = np.zeros((B, 2, 16), np.uint8); inputs = np.zeros((B, n_in, 16), np.uint8)
    for i, s in enumerate(seeds):
        delta[i], consts[i, 0], consts[i, 1], inputs[i] = gsv.labels_from_seed(s, n_in)
    sess = gsv.Session(engine, prog, B, replays, min(args.ct_ring, replays))
    ni = sess.instances_per_workgroup

    class Work:
        def new_pass(self):
            sess.set_garble_inputs(delta, consts, inputs)

        def run_slice(self, first, n):
            sess.garble(0); sess.sync()
            return sess.last_kernel_ms()

        def commit_records(self):
            from garbled_snark_verifier_amd import sharding
            out = sess.read_outputs()
            return np.stack([sharding.commit_record(seeds[i], bytes(16), out[i], delta[i], consts[i, 0], consts[i, 1], inputs[i]) for i in range(B)])

    def sync():
        torch.cuda.synchronize(); sess.sync()
    r = timed_steps(Work(), [(0, 1, gpr * replays)], args.warmup, args.steps, dist, sync, args.time_budget, T_START)
    if rank == 0:
        el, K = r["elapsed"], r["steps_run"]
        stream_s = sum(r["step_ms"]) / 1e3
        g_rank = r["gates_per_instance"] * B
        achieved = g_rank * bytes_per_gate / stream_s / 1e9
        result = {"metric": "gates/sec (garble) on Groth16/BN254 verifier at 1/2/4/8 GPUs; ciphertext-hash match", "value": g_rank * world / el, "unit": "gates/s", "n_gpus": world,
                  "steps": K, "warmup": args.warmup, "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                  "config": {"workload": "Groth16-shaped SYNTHETIC: chain of %d %s links = %d gates per instance; %d cut-and-choose instances per GPU" % (replays, args.component, gpr * replays, B),
                             "instances_per_gpu": B, "instances_per_workgroup": ni, "replays": replays, "nonfree_fraction": f_nf, "program_steps": info["n_steps"]},
                  "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "run_program_kernel<false, %d, 0>" % ni, "kernel_ms_avg": stream_s * 1e3 / K, "bytes_per_gate": bytes_per_gate,
                               "algorithmic_bytes_per_launch": g_rank / K * bytes_per_gate, "binding_limit": "aes-issue",
                               "aes_ceiling_gates_per_s": AES_CEILING_AND_PER_S / f_nf, "aes_ceiling_frac": (g_rank / stream_s) / (AES_CEILING_AND_PER_S / f_nf)}}
        if not args.no_check:
            import oracle_lib as o
            os.environ["GSV_INSTANCES_PER_WG"] = str(ni)
            chk = gsv.CircuitBuilder.streaming_garbling(args.component, [seeds[1], seeds[0], seeds[2]], engine=engine, program=prog, replays=2, keep_ciphertexts=False)
            del os.environ["GSV_INSTANCES_PER_WG"]
            ref = o.garble(args.component + "_chain:2", seeds[0], capture_ct=False)
            result["ciphertext_hash_match"] = bool(chk.ciphertext_hash[1] == ref.ct_hash.tobytes() and (chk.output_label0[1] == ref.output_label0).all())
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as o
            result["cpu_baseline"] = cpu_baseline(np, o, lambda m: None)
        print(json.dumps(result), flush=True)
    dist.barrier()
    dist.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=1024, help="cut-and-choose instances per GPU (4 x the 256 CUs: four per workgroup; 257..512: two per workgroup)")
    ap.add_argument("--e2e-instances", type=int, default=512, help="instances of the e2e_with_commitment measurement (its gate-order copy of the ciphertext blocks doubles their HBM)")
    ap.add_argument("--slices", type=int, default=10, help="steps per full verifier pass: the plan's calls are cut into this many slices of equal gate count")
    ap.add_argument("--workload", default="verifier_compressed", choices=["synthetic", "verifier", "verifier_compressed"],
                    help="verifier_compressed (default) / verifier: the restated groth16_verify_compressed / groth16_verify circuit of the committed fixture as a plan of "
                         "component programs; synthetic: the Groth16-shaped chain")
    ap.add_argument("--time-budget", type=float, default=840.0, help="seconds from process start within which the timed steps must end; steps are reduced (and reported) if they would not fit")
    ap.add_argument("--plan-cache", default=None, help="directory of the plan file shared by the ranks of a node (default: $GSV_PLAN_CACHE, /dev/shm, /tmp)")
    ap.add_argument("--no-plan-cache", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-gates", type=float, default=1.5e11, help="gates (all instances) garbled by the e2e_with_commitment measurement")
    ap.add_argument("--replays", type=int, default=0, help="synthetic: chain links per instance (0 = enough for 11.17 B gates)")
    ap.add_argument("--ct-ring", type=int, default=2, help="synthetic: replays of ciphertexts kept per instance in HBM")
    ap.add_argument("--component", default="fq12_sqmul", choices=["fq12_sqmul", "fq12_mul"], help="synthetic: link of the chain")
    args = ap.parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # No launcher: start the ranks ourselves, as fresh processes, before anything here has touched a GPU (no exec from a GPU process).
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if int(world_env or "1") != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node == --gpus" % (args.gpus, world_env), file=sys.stderr)
        sys.exit(2)
    if args.workload == "synthetic":
        run_synthetic(args)
    else:
        run_verifier(args)
    # sessions, plan and engine are closed in order by now (and the package's atexit hook closes whatever is left before the
    # interpreter tears down); a normal exit also lets a profiler's own exit handlers write their output
    sys.stdout.flush(); sys.stderr.flush()


if __name__ == "__main__":
    main()
