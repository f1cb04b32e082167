#!/usr/bin/env python3
"""Headline benchmark: gates/s (garble) on the Groth16/BN254 verifier circuit, one rank per GPU.

Workload (BASELINE.json configs[3]): the restated `groth16_verify_compressed` circuit (reference: src/gadgets/groth16.rs:250-268,
the circuit src/garbled_groth16.rs garbles and examples/groth16_garble.rs:116-129 times) for a verifying key with ONE public input
— the reference's own benchmark configuration (`GarblerInput { public_params_len: 1, .. }`, examples/groth16_garble.rs:107-110,
groth16_cut_and_choose.rs:116-119): point decompression, window-10 MSM, Miller loop, final exponentiation, comparison —
11,456,865,898 gates per instance as this tree's gadgets emit them (the reference's README quotes 11,174,708,821; DESIGN.md §2:
component-by-component reconciliation, tools/gate_counts --json for a key-by-key diff against the reference's own counter),
recorded as a plan of component programs and garbled exactly as the reference streams it.  Every GPU garbles `--instances`
(default 1024 = four per workgroup on 256 CUs) independent cut-and-choose instances (own seed => own delta / labels / ciphertext
stream).

A "step" is one SLICE of that pass: the windows of the session's schedule are cut into `--slices` consecutive groups of (nearly)
equal gate count and step i garbles slice i mod slices for all instances of the rank — wires, gate ids and the ciphertext stream
continue from step to step, so `--slices` consecutive steps are exactly one full verifier pass per instance (a whole pass is ~125 s
at 1024 instances, which no driver budget fits 25 times).  `value` = gates garbled by all ranks in the timed steps / elapsed, with
the ciphertexts produced into HBM (one window's block per instance, overwritten by the next window: inputs and outputs resident in
HBM).  Instance 0 of rank 0 carries the fixture's seed: once a whole pass has run its output label is compared with the fixture
(`headline_output_label_match`: the very kernel instantiation that is timed).

Beside it, measured in the same run on rank 0 (N = 1):
  e2e_with_commitment   ONE WHOLE PASS of `--e2e-instances` instances with every ciphertext copied out over PCIe and folded into its
                        instance's CBC-MAC (src/ciphertext_hasher.rs:23-29) — what the reference's timed garble does
                        (examples/groth16_garble.rs:111-129).  Instance 0 carries the fixture's seed: its final MAC and output label
                        are the `ciphertext_hash_match` against the fixture the CPU oracle produced from the flat stream.  Never `value`.
  rate_by_instances     HBM-resident rate of whole passes at 1 and 16 instances (BASELINE configs 4 / 5 at their stated sizes) and
                        of a sample at 256, next to the headline's 1024; the single instance also WITH its commitment (one serial chain).
  cc16_one_gpu          BASELINE config 5 with all 16 instances on this GPU: sharding.cut_and_choose_commit on the full verifier, every
                        one of the 16 commit records compared with the CPU oracle's (tests/golden/cc16_verifier_golden.json).
  cpu_baseline          the C++ restatement of the reference's per-gate loop on a PREFIX of the same stream, one core and all cores.
  mode_rates            evaluate-mode and Blake3Hasher garble rates on a component chain (rows a9 / f4 of SURVEY.md §8).

`--workload cc16` is BASELINE config 5: 16 cut-and-choose instances from one master seed, instance i -> rank i mod N, garbled with
the ciphertext commitments, one all-gather of the GarbledInstanceCommit records.

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

from bench_support import *  # noqa: F401,F403 - constants, rank logic, plan cache, CPU baseline, measurement legs (bench_support.py)
from bench_support import T_START, _plan_cache_path  # noqa: F401 - (underscore / start-time names are not covered by the star import)


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

    compressed = args.workload.startswith("verifier_compressed")
    case = json.load(open(os.path.join(ROOT, "tests", "golden", FIXTURE[args.workload])))
    units = (SMALL_BATCH_UNITS if args.units == "fq6" else VERIFIER_UNITS) + (["fp254::exp_chunk"] if compressed else [])
    engine = gsv.Engine(local_rank)  # raises without a HIP device: no CPU fallback
    extras = rank == 0 and world == 1
    want_small = extras and compressed and args.small_batch_units == "fq6" and args.units == "fq12" and not (args.no_rate_by_instances and args.no_cc16 and args.no_mode_rates)
    # The small-batch plan's FILE is built in the background, BEHIND the timed steps (round 6; round 5 built it beside the headline plan: two
    # CPU-bound builds on a 16-core quota, 86 s each instead of 50 s); it is loaded when the legs that follow the timed steps need it.  Only the
    # file: nothing in that thread touches the device.
    small_build = {"thread": None, "seconds": None, "error": None, "pair_build_s": None}
    if want_small:
        sp = _plan_cache_path(args, case["circuit"], SMALL_BATCH_UNITS + ["fp254::exp_chunk"], 1)
        hp = _plan_cache_path(args, case["circuit"], units, 4)
        if os.environ.get("GSV_BENCH_PAIR_BUILD") == "1" and sp and hp and not os.path.exists(sp) and not os.path.exists(hp):
            # Opt-in: BOTH plan files from one build (gsv_plan_build_file_pair: the 182 constant line functions recorded once, compiled for both
            # shares of the LDS window).  Measured on the 16-core quota (profiles/r06_e2e/plan_build_threads.log): 87.6 s for the pair against
            # 50-55 s for one plan — a build is bound by compile CPU time, not by the recording, so the pair saves CPU (and 7 % of it at that),
            # not time to the first launch.  The default builds the headline plan alone and the other one behind the timed steps.
            t1 = time.time()
            try:
                gsv.Plan.build_file_pair(case["circuit"], units, hp, 4, sp, 1, units_b=SMALL_BATCH_UNITS + ["fp254::exp_chunk"])
                small_build["pair_build_s"] = small_build["seconds"] = time.time() - t1
            except Exception as e:  # noqa: BLE001 - the separate builds below take over
                small_build["error"] = repr(e)
                for q in (sp, hp):
                    if os.path.exists(q):
                        os.remove(q)
        if sp and not os.path.exists(sp):
            def _build_small():
                t1 = time.time()
                try:
                    gsv.Plan.build_file(case["circuit"], SMALL_BATCH_UNITS + ["fp254::exp_chunk"], sp, window_div=1)
                    small_build["seconds"] = time.time() - t1
                except Exception as e:  # noqa: BLE001 - get_plan below builds it again (and reports) if the file is not there
                    small_build["error"] = repr(e)
            small_build["thread"] = threading.Thread(target=_build_small, daemon=True)  # started once the headline plan is in HBM (below)
    plan, plan_info, save_later = get_plan(gsv, engine, args, case["circuit"], units, rank, local_rank, local_world, dist, log)
    if small_build["pair_build_s"] is not None:
        plan_info["how"] = "built to file together with the small-batch plan (gsv_plan_build_file_pair), loaded"
        plan_info["build_s"] = small_build["pair_build_s"]
        plan_info["seconds"] += small_build["pair_build_s"]
    if small_build["thread"] is not None:
        # the builder's threads (compile pool + warm-up recorders) stay below the CPU quota: a cgroup that runs out of quota throttles ALL its
        # threads — the one that launches the timed windows too — for the rest of the period
        cores = cpu_quota_cores() or float(os.cpu_count() or 4)
        os.environ.setdefault("GSV_COMPILE_THREADS", str(max(2, int(cores) - 5)))
        os.environ.setdefault("GSV_PLAN_WARMUP_THREADS", "3")
        small_build["thread"].start()
    B, n_in, gates = args.instances, plan.info["n_inputs"], plan.info["n_gates"]
    n_calls = plan.info["n_calls"]
    n_ct = plan.info["n_ciphertexts"]
    f_nf = n_ct / gates
    bytes_per_gate = 64.0 + 16.0 * f_nf  # 16 B record + 2x16 B label reads + 16 B write + 16 B*f_nf ciphertext
    ci = plan.call_info()
    image_bytes, n_programs = plan.image_bytes()
    n_pub = (n_in - 254 - 255 - 509 - 255) // 254 + 1 if compressed else None
    if rank == 0:
        log("bench.py: plan %s in %.1f s (%d calls of %d programs, %.1f GB of program records)" % (plan_info["how"], plan_info["seconds"], n_calls, n_programs, image_bytes / 1e9))

    def fixture_ok(hash_bytes, out_labels):
        return bool(hash_bytes.hex() == case["ct_hash"] and hashlib.sha256(out_labels.tobytes()).hexdigest() == case["output_label0_sha256"])

    result = {}
    # Round 6 order: the TIMED steps come first — the headline plan is built ALONE (round 5 built the small-batch plan beside it and paid 86 s
    # for each instead of 50 s; a build is bound by the host's CPU quota, so nothing is gained by sharing it) — and the small-batch plan's
    # file is built in the background while the timed steps run: they need no host work beyond ~70 launches per slice, and the builder is
    # held to fewer threads than the quota has cores so that the launching thread is never throttled with it.
    seeds = instance_seeds(rank, B)
    head_gold = None
    if rank == 0:
        seeds[0] = case["seed"]  # the fixture's seed: the timed kernel's output label is checked once a whole pass has run
        # instances 1..7 — workgroup positions 1, 2, 3 of the first workgroup and all four of the second — carry seeds of the cut-and-choose
        # fixture, whose whole-stream MACs the CPU oracle computed: the timed configuration's CIPHERTEXTS are checked after the timed steps
        head_gold = cc16_verifier_fixture(case) if compressed and args.workload == "verifier_compressed" else None
        if head_gold is not None:
            for k in range(min(7, B - 1, len(head_gold["seeds"]))):
                seeds[1 + k] = head_gold["seeds"][k]
    work = VerifierWork(gsv, engine, plan, B, seeds)
    ni = work.sess.instances_per_workgroup
    slices = work.slices(ci[:, 1], args.slices)
    work_windows = work.sess.windows()
    sched = work.sess.schedule_info()
    forms = plan.call_record_forms()
    n_fw_windows = sum(1 for (c0, nc, _w) in work.sess.windows() if any(f == 4 for f in forms[c0:c0 + nc]))
    if rank == 0:
        log("bench.py: %d instances, %d per workgroup; %d windows in %d slices; per instance: wire file %.1f MB, ciphertext window %.1f MB"
            % (B, ni, sched["n_windows"], len(slices), sched["wire_file_slots"] * 16 / 1e6, sched["window_ct_records"] * 16 / 1e6))

    def sync():
        torch.cuda.synchronize()
        work.sess.sync()

    t_first_launch = time.time() - T_START
    r = timed_steps(work, slices, args.warmup, args.steps, dist, sync, args.time_budget, T_START)
    # (the T-table ceiling of this box: a child process of a few seconds, behind the timed steps since round 6)
    aes_and_per_s, aes_src = measure_aes_ceiling(log) if rank == 0 else (AES_CEILING_AND_PER_S_R02, "not measured on this rank")
    label_match = None
    if rank == 0 and (args.warmup + r["steps_run"]) >= len(slices) and r["commit_table"] is not None:
        # the last completed pass's output label of instance 0 (fixture seed), as gathered in its commit record: commit(label0) of output 0
        from garbled_snark_verifier_amd import sharding
        rec = r["commit_table"][0].numpy() if hasattr(r["commit_table"][0], "numpy") else np.asarray(r["commit_table"][0])
        _, _, _, outc, _, _ = sharding.record_fields(rec, plan.info["n_outputs"], n_in)
        exp = sharding.commit_labels(np.frombuffer(bytes.fromhex(case["first_output_label0"]), np.uint8)[None, :])[0]
        label_match = bool((outc[0, 1] == exp).all())
    # ---- the timed configuration's ciphertexts (examples/groth16_garble.rs:255-263 compares the garbler's and the evaluator's ciphertext
    # hash): ONE MORE whole pass of the very session that was timed — same instances, same kernel instantiation, same windows — with the
    # streams of its first instances drained and CBC-MAC'ed (gsv_session_set_drain_instances: every instance is garbled, 8 x 47.7 GB cross
    # PCIe instead of 1 024 x), against the oracle's flat-stream fixtures.  Outside the timed region.
    head_ct = None
    if rank == 0 and world == 1 and not args.no_headline_ct_check and time.time() - T_START < args.time_budget + 60:
        try:
            n_chk = min(8, B) if head_gold is not None else 1
            work.sess.set_drain_instances(n_chk)
            work.sess.set_unchecked_slices(False)
            t0 = time.perf_counter()
            dt = work.run_pass(commit=True, threads=mac_threads_for_rank(args.mac_threads, local_world))
            out = work.sess.read_outputs()
            ok0 = fixture_ok(work.ct_hashes[0], out[0])
            oks = [ok0] + [work.ct_hashes[1 + k].hex() == head_gold["ct_hashes"][k] and out[1 + k][0].tobytes().hex() == head_gold["first_output_label0"][k] for k in range(n_chk - 1)]
            head_ct = {"match": bool(all(oks)), "instances_checked": n_chk, "instances_matching": int(sum(oks)), "workgroup_positions_checked": sorted({i % ni for i in range(n_chk)}),
                       "ciphertexts_checked": n_chk * n_ct, "seconds": dt, "instances_garbled": B, "instances_per_workgroup": ni,
                       "sample": "one more whole pass of the timed session (%d instances, %d per workgroup, the timed windows); the streams of instances 0..%d drained over PCIe and CBC-MAC'ed: "
                                 "instance 0 = the single-instance fixture's seed, the others = seeds of tests/golden/cc16_verifier_golden.json" % (B, ni, n_chk - 1)}
            log("bench.py: headline ciphertext check: %d of %d instances match the oracle's MACs (%.1f s)" % (sum(oks), n_chk, dt))
        except Exception as e:  # noqa: BLE001
            head_ct = {"error": repr(e)}
    work.close()

    # the small-batch plan: the same circuit with Fq6-level units (SMALL_BATCH_UNITS) for the legs with 1 and 16 instances — more width for
    # the call-level dataflow; the stream is the same stream (every leg below checks its output label / MAC / records against the fixtures)
    plan_small, plan_small_info = None, None
    if want_small:
        try:
            if small_build["thread"] is not None:
                small_build["thread"].join()
            # (its programs keep the FULL LDS label window — window_div 1, one instance per workgroup, which is what 1 and 16 instances run:
            # 3 % faster steps than the quarter-window image the full GPU's four instances per workgroup need)
            plan_small, plan_small_info, _ = get_plan(gsv, engine, args, case["circuit"], SMALL_BATCH_UNITS + ["fp254::exp_chunk"], rank, local_rank, local_world, dist, log, window_div=1)
            plan_small_info["built_behind_the_timed_steps_s"] = small_build["seconds"]
            plan_small_info["built_with_the_headline_plan_in_one_build_s"] = small_build["pair_build_s"]
            plan_small_info["seconds_from_process_start_to_ready"] = time.time() - T_START
            log("bench.py: small-batch plan (Fq6-level units) %s in %.1f s (%d calls); its file was built behind the timed steps in %s s" % (plan_small_info["how"], plan_small_info["seconds"], plan_small.info["n_calls"], small_build["seconds"]))
        except Exception as e:  # noqa: BLE001 - the legs fall back to the headline's plan
            plan_small, plan_small_info = None, {"error": repr(e)}
    plan_sb = plan_small or plan
    # ---- CPU baseline: the restated per-gate loop on a prefix of the very stream the GPU garbles (rank 0 at N = 1 only)
    if extras and not args.no_cpu_baseline:
        try:
            import oracle_lib as o
            result["cpu_baseline"] = cpu_baseline(o, case["circuit"], case["seed"], log)
            log("bench.py: cpu baseline %.3g gates/s on 1 core, %.3g on %d cores" % (result["cpu_baseline"]["cpu_1core"]["value"], result["cpu_baseline"]["value"], result["cpu_baseline"]["cores"]))
        except Exception as e:  # the baseline must not cost the run its result line
            result["cpu_baseline"] = {"error": repr(e)}

    # ---- ONE WHOLE PASS with the commitment: every ciphertext of every instance drained over PCIe and folded into its CBC-MAC while the
    # GPU garbles the next window (the like-for-like of the reference's timed garble, examples/groth16_garble.rs:111-129).  Instance 0
    # carries the fixture's seed: its MAC over the whole stream and its output label are this run's ciphertext-hash check.
    if extras and not args.no_e2e:
        try:
            Be = max(1, args.e2e_instances)
            seeds = [case["seed"]] + instance_seeds(rank, Be)[1:]
            # instances 1..16 carry the 16 seeds of the cut-and-choose fixture (master seed 1234): their commitments are checked against
            # the CPU oracle's flat garblings too (tests/golden/cc16_verifier_golden.json), so 17 of the pass's MACs are verified
            cc_gold = cc16_verifier_fixture(case) if compressed and args.workload == "verifier_compressed" else None
            n_cc = 0
            if cc_gold is not None:
                n_cc = min(len(cc_gold["seeds"]), Be - 1)
                seeds[1:1 + n_cc] = cc_gold["seeds"][:n_cc]
            e2e = VerifierWork(gsv, engine, plan, Be, seeds)
            try:
                si = e2e.sess.schedule_info()
                quota = cpu_quota_cores()
                dt = e2e.run_pass(commit=True, threads=mac_threads_for_rank(args.mac_threads, local_world))
                out = e2e.sess.read_outputs()
                ok = fixture_ok(e2e.ct_hashes[0], out[0])
                cc_ok = [e2e.ct_hashes[1 + k].hex() == cc_gold["ct_hashes"][k] and out[1 + k][0].tobytes().hex() == cc_gold["first_output_label0"][k] for k in range(n_cc)]
                ok = bool(ok and all(cc_ok))
                result["ciphertext_hash_match"] = ok
                gbs = n_ct * Be * 16 / dt / 1e9
                # the drain's grouping (engine.cpp gsv_drain::group_for): sixteen chains per worker on VAES hosts from 128 instances up, else four
                chains = 16 if gsv.cbcmac_chains_per_step() == 16 and Be >= 128 else 4
                groups = (Be + chains - 1) // chains
                workers = args.mac_threads or max(1, min(groups, 8 if chains == 16 else 32, os.cpu_count() or 1))
                # what ONE host core MACs (the drain's inner loop) with four chains interleaved and with as many as this host's widest form takes
                mac_rate = {}
                for nch in sorted({4, gsv.cbcmac_chains_per_step()}):
                    bufs = [np.full(1 << 20, (17 * k + 1) & 255, np.uint8) for k in range(nch)]
                    for _ in range(5):
                        t1 = time.perf_counter()
                        gsv.cbcmac_many(bufs)
                        mac_rate[nch] = max(mac_rate.get(nch, 0.0), nch * (1 << 20) / 16 / (time.perf_counter() - t1))
                mac_core = max(mac_rate.values())
                cores = quota or float(os.cpu_count() or 16)
                result["e2e_with_commitment"] = {
                    "value": gates * Be / dt, "unit": "gates/s", "instances": Be, "instances_per_workgroup": e2e.sess.instances_per_workgroup, "seconds": dt, "passes": 1,
                    "ciphertext_gb_per_s": gbs, "ciphertext_gb_total": n_ct * Be * 16 / 1e9, "mac_workers": workers, "mac_chains_per_worker": chains, "host_cores_quota": quota,
                    "windows": si["n_windows"], "window_ct_records": si["window_ct_records"], "distinct_macs": len(set(e2e.ct_hashes)),
                    "instance0": {"seed": case["seed"], "ct_hash": e2e.ct_hashes[0].hex(), "fixture_ct_hash": case["ct_hash"], "match": fixture_ok(e2e.ct_hashes[0], out[0])},
                    "macs_checked_against_oracle_fixtures": 1 + n_cc, "macs_matching": int(fixture_ok(e2e.ct_hashes[0], out[0])) + sum(cc_ok),
                    "per_node_ceiling_8gpus": {"gates_per_s": min(8 * gates * Be / dt, cores * mac_core / f_nf), "pcie_bound_gates_per_s": 8 * gates * Be / dt,
                                               "mac_bound_gates_per_s": cores * mac_core / f_nf, "mac_blocks_per_s_per_core": mac_core,
                                               "mac_blocks_per_s_per_core_by_chains": {str(k): v for k, v in mac_rate.items()},
                                               "note": "8 x this GPU's PCIe-bound rate, capped by the host's MAC capacity: one core advances mac_blocks_per_s_per_core CBC-MAC blocks/s with "
                                                       "its chains interleaved (4 per step with AES-NI, 16 with VAES + AVX-512; measured in this run on cache-resident buffers; the drain "
                                                       "groups sixteen from 128 instances up) and the stream holds f_nf = %.4f blocks per gate; cores = this container's cpu.max quota (%s) "
                                                       "or the visible CPUs" % (f_nf, quota)},
                    "sample": "one WHOLE pass of the circuit (%d gates, %d ciphertexts per instance) for %d instances: garbled, every ciphertext drained over PCIe and CBC-MAC'ed per instance "
                              "on the host (gsv_session_garble_streaming); instance 0 = the fixture's seed, hash + output label compared with the oracle's flat-stream fixture" % (gates, n_ct, Be)}
                log("bench.py: e2e with commitment %.3g gates/s (%.1f GB/s of ciphertexts, %.1f s), fixture hash %s" % (gates * Be / dt, gbs, dt, "ok" if ok else "MISMATCH"))
            finally:
                e2e.close()
        except Exception as e:  # noqa: BLE001
            result["e2e_with_commitment"] = {"error": repr(e)}

    # ---- the configurations BASELINE names at their stated sizes: ONE instance (config 4) and 16 (config 5's total), whole passes with
    # the ciphertexts produced into HBM; independent calls of the plan run side by side (schedule.hpp).  A sample of the pass at 256.
    if extras and not args.no_rate_by_instances:
        rbi = {}
        try:
            for Bi, whole in ((1, True), (16, True), (256, False)):
                if time.time() - T_START > args.extras_budget - 500:
                    rbi[str(Bi)] = {"skipped": "time budget"}
                    continue
                # sixteen instances: the session as a ciphertext RING (gsv_plan_session_opts.retain_stream = GSV_STREAM_RING) — the whole pass one
                # launch, so the instances' independent call chains overlap as far as one instance's do; sixteen default windows cannot each
                # be as large as one instance's (17 launches, +12 % depth): that figure follows as "default_windows"
                w = VerifierWork(gsv, engine, plan_sb if Bi <= 16 else plan, Bi, [case["seed"]] + instance_seeds(rank, Bi)[1:], retain_stream="ring" if Bi == 16 else False)
                try:
                    si = w.sess.schedule_info()
                    if whole:
                        dt = w.run_pass()
                        g = gates
                        okl = hashlib.sha256(w.sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
                    else:
                        sl = w.slices(ci[:, 1], 10)
                        w.sess.set_unchecked_slices(True)  # a sample from the middle of the pass: timing only
                        w.new_pass()
                        pick = [sl[0], sl[len(sl) // 2], sl[-1]]
                        t0 = time.perf_counter()
                        for first, n, _ in pick:
                            w.run_slice(first, n)
                        dt = time.perf_counter() - t0
                        g = sum(x[2] for x in pick)
                        okl = None
                    rbi[str(Bi)] = {"gates_per_s": g * Bi / dt, "seconds": dt, "gates_per_instance": g, "whole_pass": whole, "output_label_match": okl, "instances_per_workgroup": w.sess.instances_per_workgroup,
                                    "plan_units": "fq6 (small-batch plan, %d calls)" % plan_small.info["n_calls"] if (plan_small is not None and Bi <= 16) else "fq12 (%d calls)" % n_calls,
                                    "max_width": si["max_width"], "windows": si["n_windows"], "depth_steps": si["critical_steps"], "total_steps": si["total_steps"]}
                    log("bench.py: %d instance(s): %.3g gates/s" % (Bi, g * Bi / dt))
                    if Bi == 16:
                        rbi["16"]["ciphertext_ring_records"] = si["ct_ring_records"]
                        if time.time() - T_START < args.extras_budget - 450:
                            w.close()
                            w = VerifierWork(gsv, engine, plan_sb, Bi, [case["seed"]] + instance_seeds(rank, Bi)[1:])
                            dtw, siw = w.run_pass(), w.sess.schedule_info()
                            rbi["16"]["default_windows"] = {"gates_per_s": gates * Bi / dtw, "seconds": dtw, "windows": siw["n_windows"], "depth_steps": siw["critical_steps"],
                                                            "output_label_match": hashlib.sha256(w.sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]}
                            log("bench.py: 16 instances in default windows: %.3g gates/s" % (gates * Bi / dtw))
                    if Bi == 1 and time.time() - T_START < args.extras_budget - 400:
                        # BASELINE's single-instance target is stated WITH the ciphertext hash: the same pass again, the stream drained and
                        # folded into ONE serial CBC-MAC chain on one host core (the chain, ~1.1e8 blocks/s = 27 s, is nearly as long as the
                        # garbling).  The session's default: two launch windows (the scope in which the instance's call chains overlap), the
                        # stream taken off the device in 1 GB segments of the RUNNING window, eight gate-order buffers between the device and
                        # the chain (engine.cpp, garble_streaming_range; tools/rounds_1-4/small_batch_commit.py: 30.4 s against round 3's 36.9 s)
                        wc = VerifierWork(gsv, engine, plan_sb, 1, [case["seed"]])
                        try:
                            dtc = wc.run_pass(commit=True)
                            okc = fixture_ok(wc.ct_hashes[0], wc.sess.read_outputs()[0])
                            sic = wc.sess.schedule_info()
                        finally:
                            wc.close()
                        rbi["1"]["with_commitment"] = {"gates_per_s": g / dtc, "seconds": dtc, "ciphertext_hash_match": okc, "vs_reference_published_32M": g / dtc / 32e6, "windows": sic["n_windows"],
                                                       "drain_segments": sic["n_segments"], "segment_ct_records": sic["segment_ct_records"]}
                        rbi["1"]["vs_reference_published_32M"] = g / dt / 32e6
                        log("bench.py: 1 instance with the commitment: %.3g gates/s, hash %s" % (g / dtc, "ok" if okc else "MISMATCH"))
                finally:
                    w.close()
        except Exception as e:  # noqa: BLE001
            rbi["error"] = repr(e)
        result["rate_by_instances"] = rbi
    # ---- BASELINE config 5 at its real size on this one GPU, all 16 commit records against the oracle-built fixture
    if extras and not args.no_cc16 and compressed and time.time() - T_START < args.extras_budget - 150:
        try:
            gold16 = cc16_verifier_fixture(case)
            result["cc16_one_gpu"] = cc16_one_gpu(gsv, engine, plan_sb, case, gold16, log) if gold16 is not None else {"skipped": "no cc16 fixture for this circuit"}
            if gold16 is not None:
                result["cc16_one_gpu"]["plan_units"] = "fq6 (small-batch plan)" if plan_small is not None else "fq12"
        except Exception as e:  # noqa: BLE001
            result["cc16_one_gpu"] = {"error": repr(e)}
    if extras and not args.no_mode_rates:
        try:
            result["mode_rates"] = mode_rates(gsv, engine, np)
        except Exception as e:  # noqa: BLE001
            result["mode_rates"] = {"error": repr(e)}
        # the second phase of the reference's benchmark (examples/groth16_garble.rs:171-230): ONE instance garbled and — window by window, from
        # the garbler's device block, nothing retained, nothing over PCIe — evaluated at the same time with the valid proof's input bits
        if compressed and "input_bits_hex" in case and time.time() - T_START < args.extras_budget - 60:
            try:
                result["mode_rates"]["garble_then_evaluate"] = garble_then_evaluate(gsv, engine, plan_sb, case, np)
                log("bench.py: garble + evaluate side by side, one instance: %.1f s, decoded output %s" % (result["mode_rates"]["garble_then_evaluate"]["seconds"], result["mode_rates"]["garble_then_evaluate"]["decoded_output"]))
            except Exception as e:  # noqa: BLE001
                result["mode_rates"]["garble_then_evaluate"] = {"error": repr(e)}

    if plan_small is not None:
        plan_small.close()  # its 41 GB of program records make room for the headline's session
        plan_small = None
    if rank == 0:
        el, K = r["elapsed"], r["steps_run"]
        stream_s = sum(r["step_ms"]) / 1e3  # device time of the timed steps: HIP events on the engine's stream around every slice
        n_calls_timed = r["calls"]
        # kernel dispatches of the timed steps: ONE per window of the schedule (its calls are rows of the grid); what rocprofv3 counts
        wins = work_windows
        win_per_slice = [sum(1 for w in wins if sl[0] <= w[0] < sl[0] + sl[1]) for sl in slices]
        n_launch = sum(win_per_slice[(args.warmup + j) % len(slices)] for j in range(K))
        g_rank = r["gates_per_instance"] * B
        achieved = g_rank * bytes_per_gate / stream_s / 1e9
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (tools/profile_r05.sh; counters cannot be read from inside the
        # process).  It is quoted only when it belongs to THIS build and THIS configuration: the traffic.json records the sha256 of the
        # libgsv_engine.so that was profiled, the batch and the circuit; anything else leaves `traffic` null.
        traffic, traffic_source = None, None
        try:
            import garbled_snark_verifier_amd.build as _b
            with open(_b.build(), "rb") as fh:
                lib_sha = hashlib.sha256(fh.read()).hexdigest()
        except Exception:  # noqa: BLE001
            lib_sha = None
        try:
            src_sha = _b.source_sha256()  # the same sources built elsewhere give another library hash (paths are compiled in): either match counts
        except Exception:  # noqa: BLE001
            src_sha = None
        for cand in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.endswith("_final")), reverse=True):
            tpath = os.path.join(ROOT, "profiles", cand, "traffic.json")
            if os.path.exists(tpath) and compressed:
                try:
                    tj = json.load(open(tpath))
                    if int(tj.get("instances_per_gpu", 512)) != B or tj.get("circuit_gates") != gates:  # PMC passes of this very configuration (circuit and batch)
                        continue
                    same_lib = bool(lib_sha) and tj.get("engine_library_sha256") == lib_sha
                    same_src = bool(src_sha) and tj.get("engine_source_sha256") == src_sha
                    if not (same_lib or same_src):
                        traffic_source = "profiles/%s/traffic.json belongs to another engine build (library and source sha256 differ): not quoted; re-run tools/profile_r05.sh" % cand
                        continue
                    traffic = float(tj["hbm_bytes_per_launch"])
                    traffic_source = ("profiles/%s/traffic.json (separate rocprofv3 --pmc passes of this workload with %s; NOT measured in this run)"
                                      % (cand, "this very libgsv_engine.so, sha256 %s..." % lib_sha[:12] if same_lib else "a library built from these very engine sources, source sha256 %s..." % src_sha[:12]))
                    break
                except (KeyError, ValueError):
                    pass
        # LDS / VALU pipe utilisation of the wide windows: separate rocprofv3 --pmc passes (tools/profile_r05_pipe.sh), quoted as they were
        # measured (profiles/<round>_final/pipe_util_summary.json names the build they belong to)
        pipe_util = None
        for cand in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.endswith("_final")), reverse=True):
            pp = os.path.join(ROOT, "profiles", cand, "pipe_util_summary.json")
            if os.path.exists(pp):
                try:
                    pipe_util = json.load(open(pp))
                    pipe_util["source"] = "profiles/%s/pipe_util_summary.json (separate rocprofv3 --pmc passes; NOT measured in this run)" % cand
                except ValueError:
                    pipe_util = None
                break
        # cold start: process start -> one (sixteen) instance(s) garbled, as this run paid it (the small-batch plan's file was built BESIDE the
        # headline's, which slows both builds; alone it is the `build_s` of a run with --no-... legs)
        cold_start = None
        try:
            rbi_ = result.get("rate_by_instances") or {}
            if plan_small_info and "seconds" in plan_small_info:
                ready = (small_build["seconds"] or 0.0) + plan_small_info["seconds"]
                cold_start = {"plan_file_build_s": small_build["seconds"], "plan_load_s": plan_small_info["seconds"],
                              "1": None if "1" not in rbi_ or "seconds" not in rbi_["1"] else ready + rbi_["1"]["seconds"],
                              "1_with_commitment": None if "with_commitment" not in rbi_.get("1", {}) else ready + rbi_["1"]["with_commitment"]["seconds"],
                              "16": None if "16" not in rbi_ or "seconds" not in rbi_["16"] else ready + rbi_["16"]["seconds"],
                              "note": "plan file build (behind the timed steps, with fewer threads than the CPU quota has cores, when the file did not exist) + load into HBM + one whole pass; reference: ~350 s for one instance on one core"}
        except Exception as e:  # noqa: BLE001
            cold_start = {"error": repr(e)}
        result.update({
            "metric": "gates/sec (garble) on Groth16/BN254 verifier at 1/2/4/8 GPUs; ciphertext-hash match", "value": g_rank * world / el, "unit": "gates/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "headline_output_label_match": label_match,
            "headline_ciphertext_hash_match": None if head_ct is None else head_ct.get("match"), "headline_ciphertext_check": head_ct,
            # the like-for-like of the reference's timed garble (hash included, garbler.rs:219-222) next to `value`, which is the commitment-free device rate
            "value_with_commitment": (result.get("e2e_with_commitment") or {}).get("value"),
            "single_instance_with_commitment": ((result.get("rate_by_instances") or {}).get("1") or {}).get("with_commitment", {}).get("gates_per_s"),
            "cold_start_s": cold_start,
            "config": {"workload": "restated %s circuit, %s public input(s) (synthetic verifying key / proof of tests/groth16_ref.py; %d gates per instance as this tree's gadgets emit them, the reference "
                                   "quotes 11,174,708,821 for the same configuration: DESIGN.md §2, tools/gate_counts --json), %d cut-and-choose instances per GPU, ciphertexts produced into HBM; one step = "
                                   "one of %d slices of the plan's %d calls, %d consecutive steps = one full verifier pass per instance"
                                   % ("groth16_verify_compressed" if compressed else "groth16_verify", n_pub if compressed else "2", gates, B, len(slices), n_calls, len(slices)),
                       "public_inputs": n_pub, "instances_per_gpu": B, "instances_per_workgroup": ni, "gates_per_instance": gates, "reference_published_gates": VERIFIER_GATES, "nonfree_fraction": f_nf,
                       "plan_calls": n_calls, "plan_programs": n_programs, "plan_windows": sched["n_windows"],
                       "slices_per_pass": len(slices), "gates_per_step_per_instance": [s[2] for s in slices], "steps_requested": args.steps,
                       "small_batch_plan": plan_small_info,
                       "passes_timed": r["gates_per_instance"] / gates, "step_device_ms": [round(x, 1) for x in r["step_ms"]], "plan": plan_info, "plan_image_gb": image_bytes / 1e9, "seconds_to_first_launch": t_first_launch,
                       "wire_file_mb_per_instance": sched["wire_file_slots"] * 16 / 1e6, "ciphertext_window_mb_per_instance": sched["window_ct_records"] * 16 / 1e6,
                       "host_peak_rss_gb": __import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6},
            "commit_records_gathered": None if r["commit_table"] is None else list(r["commit_table"].shape),
            # `bound`: what binds is the issue of the T-table AES (LDS lookups + the VALU work around them), not HBM: the §8(d) algorithmic-bytes
            # figure (achieved / peak / frac) is kept as the contract asks, the bytes that really cross the HBM interface are hbm_measured_*
            "roofline": {"bound": "lds-aes-issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "hbm_measured_gbps": None if traffic is None else traffic / (stream_s / max(1, n_launch)) / 1e9,
                         "hbm_measured_frac": None if traffic is None else traffic / (stream_s / max(1, n_launch)) / 1e9 / HBM_PEAK_GBS,
                         "pipe_utilisation": pipe_util,
                         # one step = the window launches of one slice, all of the same kernel over different component programs; a launch = one WINDOW of the schedule
                         # (grid.y = its calls) for all instances of the GPU: `launches_timed` dispatches, what rocprofv3 --kernel-trace counts
                         "kernel": "run_program_kernel<false, %d, 0, FW>" % ni, "launches_timed": n_launch, "kernel_ms_avg": stream_s * 1e3 / max(1, n_launch),
                         "kernel_note": "one kernel, two instantiations: FW = true for the windows that hold a program in the four-wire record form (%d of a pass's %d), false for the others; "
                                        "kernel_ms_avg is over all window launches = the dispatch-weighted mean of the two rows of profiles/<round>_final/kernel_stats.csv" % (n_fw_windows, sched["n_windows"]),
                         # what `frac` is a fraction OF: the contract's algorithmic bytes, which fusion and the LDS window mostly keep off HBM — not HBM traffic
                         "frac_of": "algorithmic bytes (SURVEY.md §8d: 64 + 16 f_nf per reference gate) / HBM peak — NOT HBM traffic: see hbm_measured_frac (counters) and aes_ceiling_frac (what binds)",
                         "algorithmic_bytes_per_launch": g_rank * bytes_per_gate / max(1, n_launch), "bytes_per_gate": bytes_per_gate, "calls_timed": n_calls_timed,
                         "note": "algorithmic-bytes accounting of SURVEY.md §8(d); fusion and the LDS label window keep most of those bytes off HBM, the limit that binds is T-table AES issue (DESIGN.md §3)",
                         "binding_limit": "aes-issue", "aes_ceiling_gates_per_s": aes_and_per_s / f_nf, "aes_ceiling_frac": (g_rank / stream_s) / (aes_and_per_s / f_nf), "aes_ceiling_source": aes_src},
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


def run_cc16(args):
    """BASELINE config 5: 16 cut-and-choose instances of the verifier from ONE master seed, instance i -> rank i mod N, each garbled
    with its ciphertext commitment (stream drained over PCIe and CBC-MAC'ed), ONE all-gather of the GarbledInstanceCommit records
    (cut_and_choose/garbler.rs:191-257).  The same path runs first on a shortened circuit (Fq12 multiplication) whose gathered table
    must equal the committed fixture built from the CPU oracle's garblings (tests/golden/cc16_golden.json).  A step = the whole job."""
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if world > 1:
        torch.cuda.set_device(local_rank)
    dist = Dist(world, "nccl", "cuda")
    import garbled_snark_verifier_amd as gsv
    from garbled_snark_verifier_amd import sharding

    def log(msg):
        print(msg, file=sys.stderr, flush=True)

    engine = gsv.Engine(local_rank)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cc16_golden.json")))
    total = gold["total"]
    # shortened circuit: every record against the fixture
    prog = gsv.Program.from_circuit(gold["circuit"])
    table, seeds = sharding.cut_and_choose_commit(gold["circuit"], gold["master_seed"], total, rank, world, engine=engine, program=prog, device="cuda" if world > 1 else None)
    short_ok = bool(hashlib.sha256(table.tobytes()).hexdigest() == gold["table_sha256"] and [int(x) for x in seeds] == gold["seeds"])
    prog.close()
    case = json.load(open(os.path.join(ROOT, "tests", "golden", FIXTURE["verifier_compressed"])))
    # at most 16 instances per GPU = one per workgroup: the small-batch plan (Fq6-level units, full LDS window), as in cc16_one_gpu
    units = (SMALL_BATCH_UNITS if args.small_batch_units == "fq6" else VERIFIER_UNITS) + ["fp254::exp_chunk"]
    plan, plan_info, save_later = get_plan(gsv, engine, args, case["circuit"], units, rank, local_rank, local_world, dist, log, window_div=1 if args.small_batch_units == "fq6" else 4)
    gates = plan.info["n_gates"]
    times = []
    tab = None
    for it in range(args.warmup + args.steps):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        tab, _ = sharding.cut_and_choose_commit(case["circuit"], gold["master_seed"], total, rank, world, engine=engine, program=plan, device="cuda" if world > 1 else None,
                                                threads=mac_threads_for_rank(args.mac_threads, local_world))
        torch.cuda.synchronize(); dist.barrier()
        dt = dist.max_float(time.perf_counter() - t0)
        if it >= args.warmup:
            times.append(dt)
    if rank == 0:
        el = sum(times)
        ct = [bytes(tab[i][8:24]).hex() for i in range(total)]
        # the gathered table of the FULL verifier against the oracle's 16 flat garblings (tests/golden/cc16_verifier_golden.json)
        full = None
        try:
            vg = json.load(open(os.path.join(ROOT, "tests", "golden", "cc16_verifier_golden.json")))
            if vg["master_seed"] == gold["master_seed"] and vg["total"] == total:
                t_np = tab.cpu().numpy() if hasattr(tab, "cpu") else np.asarray(tab)
                rec_ok = [hashlib.sha256(np.ascontiguousarray(t_np[i]).tobytes()).hexdigest() == vg["record_sha256"][i] for i in range(total)]
                full = {"records_matching": int(sum(rec_ok)), "all_16_match": bool(all(rec_ok)), "fixture": "tests/golden/cc16_verifier_golden.json"}
        except Exception as e:  # noqa: BLE001 - the check must not cost the run its result line
            full = {"error": repr(e)}
        result = {"metric": "gates/sec (garble) on Groth16/BN254 verifier at 1/2/4/8 GPUs; ciphertext-hash match", "value": gates * total * len(times) / el, "unit": "gates/s", "n_gpus": world,
                  "steps": len(times), "warmup": args.warmup, "ms_per_step": el / len(times) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                  "ciphertext_hash_match": short_ok,
                  "config": {"workload": "cc16: %d cut-and-choose instances of the restated groth16_verify_compressed circuit (1 public input, %d gates each) from master seed %d, instance i -> rank i mod %d, "
                                         "every ciphertext drained over PCIe and CBC-MAC'ed (the commitment the reference's Garbler::create computes, garbler.rs:219-222), one all-gather of the "
                                         "GarbledInstanceCommit records; a step = the whole job" % (total, gates, gold["master_seed"], world),
                             "instances_total": total, "instances_per_gpu": len(sharding.shard_instances(total, rank, world)), "gates_per_instance": gates, "plan": plan_info,
                             "reference_published": "16 instances on 8 physical cores: ~11 m 58 s, ~249 M gates/s (README.md:13)"},
                  "commit_records_gathered": list(tab.shape), "distinct_ciphertext_commitments": len(set(ct)), "full_verifier_records_vs_oracle": full,
                  "shortened_circuit_check": {"circuit": gold["circuit"], "table_sha256": hashlib.sha256(table.tobytes()).hexdigest(), "fixture": gold["table_sha256"], "match": short_ok}}
        print(json.dumps(result), flush=True)
        if save_later:
            try:
                plan.save(save_later)
            except Exception as e:  # noqa: BLE001
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
        aes_and_per_s, aes_src = measure_aes_ceiling(lambda m: print(m, file=sys.stderr))
        result = {"metric": "gates/sec (garble) on Groth16/BN254 verifier at 1/2/4/8 GPUs; ciphertext-hash match", "value": g_rank * world / el, "unit": "gates/s", "n_gpus": world,
                  "steps": K, "warmup": args.warmup, "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                  "config": {"workload": "Groth16-shaped SYNTHETIC: chain of %d %s links = %d gates per instance; %d cut-and-choose instances per GPU" % (replays, args.component, gpr * replays, B),
                             "instances_per_gpu": B, "instances_per_workgroup": ni, "replays": replays, "nonfree_fraction": f_nf, "program_steps": info["n_steps"]},
                  "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "run_program_kernel<false, %d, 0, %s>" % (ni, "true" if prog.info.get("and_terms", 2) == 4 else "false"), "kernel_ms_avg": stream_s * 1e3 / K, "bytes_per_gate": bytes_per_gate,
                               "algorithmic_bytes_per_launch": g_rank / K * bytes_per_gate, "binding_limit": "aes-issue",
                               "aes_ceiling_gates_per_s": aes_and_per_s / f_nf, "aes_ceiling_frac": (g_rank / stream_s) / (aes_and_per_s / f_nf), "aes_ceiling_source": aes_src}}
        if not args.no_check:
            import oracle_lib as o
            os.environ["GSV_INSTANCES_PER_WG"] = str(ni)
            chk = gsv.CircuitBuilder.streaming_garbling(args.component, [seeds[1], seeds[0], seeds[2]], engine=engine, program=prog, replays=2, keep_ciphertexts=False)
            del os.environ["GSV_INSTANCES_PER_WG"]
            ref = o.garble(args.component + "_chain:2", seeds[0], capture_ct=False)
            result["ciphertext_hash_match"] = bool(chk.ciphertext_hash[1] == ref.ct_hash.tobytes() and (chk.output_label0[1] == ref.output_label0).all())
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as o
            result["cpu_baseline"] = cpu_baseline(o, args.component + "_chain:16", seeds[0], lambda m: None)
        print(json.dumps(result), flush=True)
    dist.barrier()
    dist.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=1024, help="cut-and-choose instances per GPU (4 x the 256 CUs: four per workgroup; 257..512: two per workgroup)")
    ap.add_argument("--e2e-instances", type=int, default=64, help="instances of the e2e_with_commitment measurement: ONE WHOLE PASS, every ciphertext drained over PCIe (47.7 GB per "
                    "instance at ~50 GB/s) and CBC-MAC'ed")
    ap.add_argument("--mac-threads", type=int, default=0, help="host threads of the drain (0 = up to 32, one per four instances)")
    ap.add_argument("--slices", type=int, default=10, help="steps per full verifier pass: the plan's calls are cut into this many slices of equal gate count")
    ap.add_argument("--workload", default="verifier_compressed", choices=["synthetic", "verifier", "verifier_compressed", "verifier_compressed_2pub", "cc16"],
                    help="verifier_compressed (default): the restated groth16_verify_compressed circuit with ONE public input (the reference's benchmark configuration) as a plan of component "
                         "programs; verifier_compressed_2pub / verifier: round-2 fixtures (two public inputs, with / without point decompression); cc16: BASELINE config 5 (16 instances from one "
                         "master seed over the ranks, commitments, one all-gather); synthetic: the Groth16-shaped chain")
    ap.add_argument("--time-budget", type=float, default=840.0, help="seconds from process start within which the timed steps must end; steps are reduced (and reported) if they would not fit")
    ap.add_argument("--extras-budget", type=float, default=1350.0, help="seconds from process start within which the legs BEHIND the timed steps (CPU baseline, whole passes with commitments at 64 / 1 / 16 "
                    "instances, cc16 on one GPU, garble || evaluate) must end: a leg that would not fit is skipped and reported as such (the driver stops the run at 1 800 s)")
    ap.add_argument("--plan-cache", default=None, help="directory of the plan file shared by the ranks of a node (default: $GSV_PLAN_CACHE, /dev/shm, /tmp)")
    ap.add_argument("--no-plan-cache", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-rate-by-instances", action="store_true")
    ap.add_argument("--no-mode-rates", action="store_true")
    ap.add_argument("--small-batch-units", default="fq6", choices=["fq6", "fq12"], help="unit granularity of the plan the legs with 1 and 16 instances run (rate_by_instances, cc16_one_gpu, "
                    "garble_then_evaluate): fq6 (default) builds a second plan with the Fq12 multiplications and squarings entered as fq6::mul_montgomery units, fq12 uses the headline's plan")
    ap.add_argument("--units", default="fq12", choices=["fq12", "fq6"], help="unit granularity of the headline's plan: fq12 (default) or fq6 (experiments: the small-batch plan at full occupancy)")
    ap.add_argument("--no-headline-ct-check", action="store_true", help="skip the extra whole pass after the timed steps that checks the timed configuration's ciphertexts (~2 min)")
    ap.add_argument("--no-cc16", action="store_true", help="skip the cc16_one_gpu leg (BASELINE config 5 with all 16 instances on this GPU, ~40 s)")
    ap.add_argument("--replays", type=int, default=0, help="synthetic: chain links per instance (0 = enough for 11.17 B gates)")
    ap.add_argument("--ct-ring", type=int, default=2, help="synthetic: replays of ciphertexts kept per instance in HBM")
    ap.add_argument("--component", default="fq12_sqmul", choices=["fq12_sqmul", "fq12_mul"], help="synthetic: link of the chain")
    ap.add_argument("--dry-run", action="store_true", help="print the per-rank device / host memory and CPU budget of `--gpus N` of this workload and what it is expected to deliver; no GPU needed")
    args = ap.parse_args()
    if args.dry_run:
        dry_run(args)
        return
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
    elif args.workload == "cc16":
        run_cc16(args)
    else:
        run_verifier(args)
    # sessions, plan and engine are closed in order by now (and the package's atexit hook closes whatever is left before the
    # interpreter tears down); a normal exit also lets a profiler's own exit handlers write their output
    sys.stdout.flush(); sys.stderr.flush()


if __name__ == "__main__":
    main()
