#!/usr/bin/env python3
"""Headline benchmark: gates/s (garble) on the Groth16/BN254 verifier circuit, one rank per GPU.

Default workload (BASELINE.json configs[3]): the REAL verifier — `groth16_verify_compressed` (reference:
src/gadgets/groth16.rs:250-268, the circuit src/garbled_groth16.rs garbles): point decompression, window-10 MSM,
Miller loop, final exponentiation, comparison — 11,687,200,297 gates per instance for the synthetic 2-public-input
verifying key of tests/groth16_ref.py (the reference quotes 11,174,708,821 for its own key), recorded as a plan of
component programs and garbled exactly as the reference streams it; the whole stream's CBC-MAC and the
output label are checked against the fixture the CPU oracle produced from the FLAT stream.  Every GPU garbles
`--instances` independent cut-and-choose instances (own seed => own delta / labels / ciphertext stream) per step.

One "step" = one pass of the hot path: garble all instances of this rank's batch — every call of the plan for all
instances, ciphertexts produced in HBM (one call block per instance, overwritten by the next call), output labels
gathered.  `value` = total gates garbled by all ranks per second (weak scaling: per-GPU work fixed).

`--workload synthetic` is the Groth16-SHAPED chain of SURVEY.md §8(d) the engine was tuned on (330 Fq12
square-and-multiply links = 11.18 B gates, one compiled program replayed with a ciphertext ring; profiles/r01_sqmul);
`--workload verifier` the uncompressed `groth16_verify` (10.91 B gates).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VERIFIER_GATES = 11_174_708_821  # README.md:12 of the reference
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec


VERIFIER_UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::inverse_montgomery", "fq12::mul_by_034_montgomery",
                  "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery",
                  "bigint::multiplexer", "g1::add_montgomery", "fp254::inverse"]


def real_verifier(args):
    """--workload verifier[_compressed]: the real Groth16 verifier circuit instead of the synthetic chain.  Same
    contract: a step garbles `--instances` instances per rank (ciphertexts produced on the device and discarded, as in the
    synthetic step where the ring overwrites them), barrier + synchronize around exactly K steps, max over ranks, one JSON line."""
    import hashlib
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    import garbled_snark_verifier_amd as gsv
    compressed = args.workload == "verifier_compressed"
    try:  # every rank builds its own plan: ~50 GB of host memory each while it is built
        avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if rank == 0 and avail_gb < 55 * local_world:
            print("bench.py: %.0f GB of host memory available for %d ranks, the plan build needs ~50 GB per rank" % (avail_gb, local_world), file=sys.stderr)
    except (OSError, IndexError, ValueError):
        pass
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16_verify_compressed_golden.json" if compressed else "groth16_verify_golden.json")))
    engine = gsv.Engine(local_rank)
    t0 = time.time()
    plan = gsv.Plan.from_circuit(case["circuit"], VERIFIER_UNITS + (["fp254::exp_chunk"] if compressed else []), half_window=True)
    build_s = time.time() - t0
    B, n_in, gates = args.instances, plan.info["n_inputs"], plan.info["n_gates"]
    f_nf = plan.info["n_ciphertexts"] / gates
    bytes_per_gate = 64.0 + 16.0 * f_nf
    seeds = [1_000_003 * (rank + 1) + i for i in range(B)]
    delta = np.zeros((B, 16), np.uint8); consts = np.zeros((B, 2, 16), np.uint8); inputs = np.zeros((B, n_in, 16), np.uint8)
    for i, sd in enumerate(seeds):
        delta[i], consts[i, 0], consts[i, 1], inputs[i] = gsv.labels_from_seed(sd, n_in)
    sess = gsv.Session(engine, plan, B, retain_stream=False)
    ni = sess.instances_per_workgroup

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def step():
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble_streaming(discard=True)  # returns when the last call has finished
        if world > 1:
            out = torch.from_numpy(sess.read_outputs()).to("cuda")
            dist.all_gather([torch.empty_like(out) for _ in range(world)], out)
        return sess.last_kernel_ms()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    gpu_ms = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sess.close()
    result = None
    if rank == 0:
        avg_s = sum(gpu_ms) / len(gpu_ms) / 1e3  # stream time of one step: all launches of the plan back to back
        n_launch = plan.info["n_calls"]
        achieved = gates * B * bytes_per_gate / avg_s / 1e9
        result = {"metric": "gates/sec (garble) on Groth16/BN254 verifier; ciphertext-hash match", "value": gates * B * world * args.steps / elapsed, "unit": "gates/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                  "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                  "config": {"workload": "the real %s circuit (synthetic 2-public-input verifying key / proof of tests/groth16_ref.py): %d gates per instance in %d calls of "
                                         "component programs; %d cut-and-choose instances per GPU" % ("groth16_verify_compressed" if compressed else "groth16_verify", gates, plan.info["n_calls"], B),
                             "instances_per_gpu": B, "gates_per_instance": gates, "nonfree_fraction": f_nf, "plan_calls": plan.info["n_calls"], "plan_build_s": build_s,
                             "host_peak_rss_gb": __import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6},
                  "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               # one step = n_launch launches of the same kernel over different component programs: averages per launch
                               "kernel": "run_program_kernel<false, %d, 0>" % ni, "launches_per_step": n_launch, "kernel_ms_avg": avg_s * 1e3 / n_launch,
                               "step_stream_ms": avg_s * 1e3, "bytes_per_gate": bytes_per_gate, "algorithmic_bytes_per_launch": gates * B * bytes_per_gate / n_launch,
                               "aes_ceiling_gates_per_s": 4.9e10 / f_nf, "aes_ceiling_frac": (gates * B / avg_s) / (4.9e10 / f_nf)}}
        if not args.no_check:  # the fixture's seed through the streaming path (stream drained and hashed on the host): oracle's flat-stream hash
            d, f, t, inp = gsv.labels_from_seed(case["seed"], n_in)
            chk = gsv.Session(engine, plan, 1, retain_stream=False)
            chk.set_garble_inputs(d[None], np.stack([f, t])[None], inp[None])
            h = chk.garble_streaming()[0].hex()
            ok = h == case["ct_hash"] and hashlib.sha256(chk.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
            chk.close()
            result["ciphertext_hash_match"] = bool(ok)
            result["hash_check"] = {"circuit": "the whole circuit, seed %d" % case["seed"], "gpu": h, "oracle": case["ct_hash"]}
        if args.cpu_baseline_chain and world == 1:
            import oracle_lib as o
            sec, g, _ = o.bench_garble("g1_scalar_mul:10", seed=0)
            result["cpu_baseline"] = {"value": g / sec, "unit": "gates/s", "cores": 1, "kind": "port",
                                      "sample": "g1_scalar_mul:10 (one window-10 scalar multiplication of the verifier's MSM, %d gates) garbled once by the C++ oracle, AES-NI hash + inline CBC-MAC, 1 thread, %.1f s" % (g, sec)}
    barrier()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=512, help="cut-and-choose instances per GPU per step (more than the 256 CUs: two per workgroup)")
    ap.add_argument("--replays", type=int, default=0, help="chain links per instance (0 = enough for 11.17 B gates)")
    ap.add_argument("--ct-ring", type=int, default=2, help="replays of ciphertexts kept per instance in HBM")
    ap.add_argument("--cpu-baseline-chain", type=int, default=8, help="chain links garbled by the CPU oracle for cpu_baseline (0 = skip)")
    ap.add_argument("--component", default="fq12_sqmul", choices=["fq12_sqmul", "fq12_mul"],
                    help="link of the chain: fq12_sqmul = r <- Fq12::mul(Fq12::square(r), b) (33.9 M gates), fq12_mul = r <- Fq12::mul(r, b) (20.3 M)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--workload", default="verifier_compressed", choices=["synthetic", "verifier", "verifier_compressed"],
                    help="verifier_compressed (default) / verifier: the REAL groth16_verify_compressed / groth16_verify circuit of the committed fixture as a plan "
                         "of component programs (~95 s of plan build and ~48 GB of host memory per rank); synthetic: the Groth16-shaped chain")
    args = ap.parse_args()
    if args.workload != "synthetic":
        return real_verifier(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    import garbled_snark_verifier_amd as gsv

    engine = gsv.Engine(local_rank)  # raises without a HIP device: no CPU fallback
    t0 = time.time()
    prog = gsv.Program.from_circuit(args.component, chain_feedback=True)
    compile_s = time.time() - t0
    info = prog.info
    gates_per_replay = info["n_gates"]
    replays = args.replays or -(-VERIFIER_GATES // gates_per_replay)
    B = args.instances
    n_in = info["n_inputs"]
    f_nf = info["n_ciphertexts"] / gates_per_replay
    bytes_per_gate = 64.0 + 16.0 * f_nf  # 16 B record + 2x16 B label reads + 16 B write + 16 B*f_nf ciphertext

    seeds = [1_000_003 * (rank + 1) + i for i in range(B)]
    delta = np.zeros((B, 16), np.uint8)
    consts = np.zeros((B, 2, 16), np.uint8)
    inputs = np.zeros((B, n_in, 16), np.uint8)
    for i, s in enumerate(seeds):
        delta[i], consts[i, 0], consts[i, 1], inputs[i] = gsv.labels_from_seed(s, n_in)
    sess = gsv.Session(engine, prog, B, replays, min(args.ct_ring, replays))
    ni = sess.instances_per_workgroup

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    out_dev = torch.empty((B, info["n_outputs"], 16), dtype=torch.uint8, device="cuda")
    gathered = [torch.empty_like(out_dev) for _ in range(world)] if world > 1 else None

    def step():
        sess.set_garble_inputs(delta, consts, inputs)  # fresh labels resident in HBM before the kernel starts
        sess.garble(0)
        sess.sync()
        if world > 1:  # the one exchange of the path: all-gather of the instances' output labels (RCCL over xGMI)
            out_dev.copy_(torch.from_numpy(sess.read_outputs()).to("cuda"))
            dist.all_gather(gathered, out_dev)
        return sess.last_kernel_ms()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gates_per_step_rank = gates_per_replay * replays * B
    total_gates = gates_per_step_rank * world * args.steps
    value = total_gates / elapsed

    # HBM bytes per launch from the PMC counters: collected in separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    # passes of this same command and committed under profiles/ (a run cannot read its own counters); only reported
    # when the committed measurement was taken on the configuration being run.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_sqmul", "traffic.json")
    if os.path.exists(tpath) and args.component == "fq12_sqmul" and B == 512 and ni == 2 and replays == -(-VERIFIER_GATES // gates_per_replay) and world == 1:
        with open(tpath) as f:
            traffic = float(json.load(f)["hbm_bytes_raw"])
    result = None
    if rank == 0:
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        achieved_gbs = gates_per_step_rank * bytes_per_gate / avg_kernel_s / 1e9
        result = {
            "metric": "gates/sec (garble) on Groth16/BN254 verifier; ciphertext-hash match",
            "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "Groth16-shaped synthetic: chain of %d %s links = %d gates per instance (the verifier has 11,174,708,821); "
                                   "%d cut-and-choose instances per GPU" % (replays, {"fq12_sqmul": "Fq12 square-and-multiply (square_montgomery + mul_montgomery)",
                                                                             "fq12_mul": "Fq12::mul_montgomery"}[args.component], gates_per_replay * replays, B),
                       "component": args.component,
                       "instances_per_gpu": B, "instances_per_workgroup": ni, "replays": replays, "gates_per_instance": gates_per_replay * replays,
                       "nonfree_fraction": f_nf, "program_steps": info["n_steps"], "and_depth": info["and_depth"],
                       "wire_slots": info["n_slots"], "program_image_bytes": info["device_bytes"], "compile_s": compile_s},
            "per_instance_gates_per_s": gates_per_replay * replays / avg_kernel_s,
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "run_program_kernel<false, %d, 0>" % ni, "kernel_ms_avg": avg_kernel_s * 1e3,
                         "bytes_per_gate": bytes_per_gate, "algorithmic_bytes_per_launch": gates_per_step_rank * bytes_per_gate,
                         # what actually binds (DESIGN.md "Roofline model"): T-table AES issue, 741 VALU + 364 LDS wave-instructions
                         # per 64 garbled ANDs -> ~5 us per 1024 ANDs per CU -> 4.9e10 AND/s per GPU
                         "aes_ceiling_gates_per_s": 4.9e10 / f_nf, "aes_ceiling_frac": (gates_per_step_rank / avg_kernel_s) / (4.9e10 / f_nf),
                         "device_records_per_replay": info["n_ciphertexts"] + info.get("n_fused_free", 0),
                         "lds_label_reads_frac": info["reads_lds"] / max(1, info["reads_lds"] + info["reads_hbm"]),
                         "lds_label_writes_frac": info["writes_lds"] / max(1, info["writes_lds"] + info["writes_hbm"])},
        }
        if not args.no_check:
            import oracle_lib as o
            # bit-exactness of this very program against the CPU oracle on the chain's first two components
            # (same program variant / kernel instantiation as the timed launches: three instances, the middle one checked)
            os.environ["GSV_INSTANCES_PER_WG"] = str(ni)
            chk = gsv.CircuitBuilder.streaming_garbling(args.component, [seeds[1], seeds[0], seeds[2]], engine=engine, program=prog, replays=2, keep_ciphertexts=False)
            del os.environ["GSV_INSTANCES_PER_WG"]
            ref = o.garble(args.component + "_chain:2", seeds[0], capture_ct=False)
            result["ciphertext_hash_match"] = bool(chk.ciphertext_hash[1] == ref.ct_hash.tobytes() and (chk.output_label0[1] == ref.output_label0).all())
            result["hash_check"] = {"circuit": args.component + "_chain:2", "seed": seeds[0], "gpu": chk.ciphertext_hash[1].hex(), "oracle": ref.ct_hash.tobytes().hex()}
        if args.cpu_baseline_chain and world == 1:
            import oracle_lib as o
            spec = "%s_chain:%d" % (args.component, args.cpu_baseline_chain)
            sec, gates, _ = o.bench_garble(spec, seed=0)
            result["cpu_baseline"] = {"value": gates / sec, "unit": "gates/s", "cores": 1, "kind": "port",
                                      "sample": "%s (%d gates) garbled once by the C++ oracle, AES-NI hash + inline CBC-MAC, 1 thread, %.1f s" % (spec, gates, sec)}
    barrier()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
