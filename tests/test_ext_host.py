"""The public C ABI driven by an EXTERNAL host (tests/ext_host/ext_host.cpp): a program that links only libgsv_engine.so, includes only
include/gsv_engine.h from the engine and implements `CircuitMode` (src/circuit/modes.rs:26-51) plus the `with_named_child` unit hook
(src/circuit/streaming_mode.rs:189-241) over gsv_recorder_* / gsv_program_compile_opts / gsv_plan_recorder_* — what
bindings/rust/src/gpu_garble_mode.rs does on the Rust side.  The circuit itself comes from the shared, mode-generic driver + gadget
headers (the restated reference layers above the seam).

CPU (`not gpu`): the plan the external host builds through the ABI is, byte for byte, the plan the engine's built-in builder
(gsv_plan_build_file) writes for the same circuit, units and window share — tools/plan_digest.py: same header, same program blocks,
same calls, whatever order the compile workers appended the blocks in.
GPU: BASELINE config 4 — the whole groth16_verify_compressed circuit (1 public input, 11 456 865 898 gates, 1 257 calls of 212 unit programs)
recorded through the ABI, loaded, ONE instance garbled with nothing retained on the device through gsv_session_garble_streaming_sink
into the host program's own CBC-MAC: MAC and output label == the oracle's flat-stream fixture."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
EXT_DIR = os.path.join(ROOT, "tests", "ext_host")
EXT = os.path.join(EXT_DIR, "ext_host")
GOLDEN = os.path.join(ROOT, "tests", "golden")

from conftest import VERIFIER_PLAN_UNITS as VERIFIER_UNITS  # noqa: E402 - the headline circuit's units, shared with tests/test_gpu_parity.py


@pytest.fixture(scope="module")
def ext_host():
    import garbled_snark_verifier_amd  # noqa: F401 - builds libgsv_engine.so if needed
    subprocess.check_call(["make", "-C", EXT_DIR], stdout=subprocess.DEVNULL)
    return EXT


def run_ext(exe, spec, units, path, *more, env=None):
    r = subprocess.run([exe, spec, ",".join(units), path] + [str(x) for x in more], capture_output=True, text=True, env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_ext_host_links_only_the_public_library(ext_host):
    """The external host's only engine dependency is libgsv_engine.so (the HIP runtime comes in through that library), and its source
    includes nothing of csrc/engine."""
    needed = subprocess.check_output(["readelf", "-d", ext_host], text=True)
    libs = [ln.split("[")[1].split("]")[0] for ln in needed.splitlines() if "(NEEDED)" in ln]
    assert "libgsv_engine.so" in libs and not [x for x in libs if "hostsim" in x or "oracle" in x or "amdhip" in x]
    src = open(os.path.join(EXT_DIR, "ext_host.cpp")).read()
    incs = [ln for ln in src.splitlines() if ln.startswith("#include \"")]
    assert all("csrc/engine" not in ln for ln in incs) and any("include/gsv_engine.h" in ln for ln in incs)


@pytest.mark.parametrize("window_div", [4, 1])
def test_abi_route_plan_equals_the_builtin_builders(ext_host, tmp_path, window_div):
    """Units with glue between them, a second liveness pattern of the same component (fq12_mix); no unit at all; one gate; the in-place
    NOT; units whose outputs are dead / constants / passed-through inputs, nested components (driver_mix); 145 calls of 72 programs
    (random_circuit); unit inputs that are the constant wires (the MSM's tables behind multiplexers, g1_mux_add)."""
    import garbled_snark_verifier_amd as gsv
    import plan_digest
    a, b = os.path.join(str(tmp_path), "abi.gsvplan"), os.path.join(str(tmp_path), "builtin.gsvplan")
    cases = [("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"]), ("fq_mul", ["no::such_unit"]), ("gate:0", ["x::y"]), ("gate:10", ["x::y"]),
             ("driver_mix", ["test::inner", "bigint::add"]), ("driver_mix", ["test::mixed_outputs"]), ("random_circuit:3", ["test::random_block"]),
             ("g1_mux_add", ["bigint::multiplexer", "g1::add_montgomery"]), ("fq_complex", ["fp254::montgomery_reduce", "bigint::mul_karatsuba"])]
    for spec, units in cases:
        j = run_ext(ext_host, spec, units, a, "--window-div", window_div)
        gsv.Plan.build_file(spec, units, b, window_div=window_div)
        da, db = plan_digest.digest(a), plan_digest.digest(b)
        assert da == db and da["unreferenced_bytes"] == 0, (spec, units, da, db)
        ref = gsv.Plan.load(b)
        assert (j["n_gates"], j["n_ciphertexts"], j["n_calls"]) == (ref.info["n_gates"], ref.info["n_ciphertexts"], ref.info["n_calls"])
        ref.close()
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]


@pytest.mark.parametrize("divs", [(4, 1)])
def test_plan_file_pair_equals_separately_built_files(tmp_path, divs):
    """gsv_plan_build_file_pair: ONE build feeds TWO plans — the units they share are recorded once and compiled for both shares of the LDS
    window — and writes two plan files, each, byte for byte (program blocks, calls, header; tools/plan_digest.py), what
    gsv_plan_build_file writes for its units and window_div alone.  bench.py builds its headline plan (Fq12-level units, window_div 4)
    and its small-batch plan (Fq6-level units, window_div 1) this way."""
    import garbled_snark_verifier_amd as gsv
    import plan_digest
    d = str(tmp_path)
    for spec, units, variants in [("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"], [["fq6::mul_montgomery", "fq2::square_montgomery"]]), ("fq_mul", ["no::such_unit"], [None])]:
        pa, pb = os.path.join(d, "pair_a.gsvplan"), os.path.join(d, "pair_b.gsvplan")
        # fq12_mix: plan B cut at Fq6 level (two recorders over one unit cache: what bench.py's headline / small-batch pair would be);
        # fq_mul: the same (empty) unit set for both plans (one recorder, every program compiled twice)
        for units_b in variants:
            gsv.Plan.build_file_pair(spec, units, pa, divs[0], pb, divs[1], units_b=units_b)
            for path, div, un in ((pa, divs[0], units), (pb, divs[1], units_b or units)):
                ref = os.path.join(d, "single.gsvplan")
                gsv.Plan.build_file(spec, un, ref, window_div=div)
                da, dr = plan_digest.digest(path), plan_digest.digest(ref)
                assert da == dr and da["unreferenced_bytes"] == 0, (spec, div, un, da, dr)
    with pytest.raises(gsv.GsvError):
        gsv.Plan.build_file_pair("fq_mul", ["x::y"], os.path.join(d, "same"), 4, os.path.join(d, "same"), 1)
    with pytest.raises(gsv.GsvError):
        gsv.Plan.build_file_pair("fq_mul", ["x::y"], os.path.join(d, "p"), 3, os.path.join(d, "q"), 1)
    assert not [f for f in os.listdir(d) if ".tmp." in f]


def test_abi_misuse_is_refused(tmp_path):
    """struct_size guards both option structs; a spilled program cannot be run; a finished plan recorder takes no more units."""
    import ctypes as C
    import garbled_snark_verifier_amd as gsv
    L = gsv.lib()

    class RecOpts(C.Structure):
        _fields_ = [("struct_size", C.c_uint32), ("window_div", C.c_uint32), ("plan_file", C.c_char_p)]

    class CompOpts(C.Structure):
        _fields_ = [("struct_size", C.c_uint32), ("window_div", C.c_uint32), ("keep_trace", C.c_uint32), ("background", C.c_uint32), ("consume_recorder", C.c_uint32),
                    ("reserved", C.c_uint32), ("for_plan", C.c_void_p)]

    pr = C.c_void_p()
    assert L.gsv_plan_recorder_create_opts(C.byref(RecOpts(4, 4, None)), C.byref(pr)) == 1  # wrong struct_size
    assert L.gsv_plan_recorder_create_opts(C.byref(RecOpts(C.sizeof(RecOpts), 3, None)), C.byref(pr)) == 1  # window_div 3
    path = os.path.join(str(tmp_path), "p.gsvplan").encode()
    assert L.gsv_plan_recorder_create_opts(C.byref(RecOpts(C.sizeof(RecOpts), 4, path)), C.byref(pr)) == 0
    rec = C.c_void_p()
    assert L.gsv_recorder_create(C.byref(rec)) == 0
    assert L.gsv_recorder_record_circuit(rec, b"fq_add") == 0
    prog = C.c_void_p()
    assert L.gsv_program_compile_opts(rec, C.byref(CompOpts(8, 0, 0, 0, 0, 0, None)), C.byref(prog)) == 1  # wrong struct_size
    assert L.gsv_program_compile_opts(rec, C.byref(CompOpts(C.sizeof(CompOpts), 2, 0, 1, 0, 0, pr)), C.byref(prog)) == 1  # window_div differs from the recorder's
    assert L.gsv_program_compile_opts(rec, C.byref(CompOpts(C.sizeof(CompOpts), 0, 0, 1, 0, 0, pr)), C.byref(prog)) == 0  # background, spilled into the plan file
    assert L.gsv_program_wait(prog) == 0
    # a plan of one call of it: two inputs of 254 wires, one output
    first = C.c_uint64()
    assert L.gsv_plan_recorder_allocate_wires(pr, 508, C.byref(first)) == 0
    for i in range(508):
        assert L.gsv_plan_recorder_declare_input(pr, first.value + i) == 0
    ins = (C.c_uint64 * 508)(*[first.value + i for i in range(508)])
    outs = (C.c_uint64 * 254)()
    assert L.gsv_plan_recorder_call(pr, prog, ins, outs) == 0
    plan = C.c_void_p()
    assert L.gsv_plan_recorder_finish(pr, outs, 254, C.byref(plan)) == 0
    assert L.gsv_plan_recorder_finish(pr, outs, 254, C.byref(plan)) == 1  # already finished
    prog2 = C.c_void_p()
    assert L.gsv_program_compile_opts(rec, C.byref(CompOpts(C.sizeof(CompOpts), 0, 0, 0, 0, 0, pr)), C.byref(prog2)) == 1  # recorder finished
    loaded = gsv.Plan.load(path.decode())
    ref = gsv.Plan.from_circuit("fq_add", ["no::unit"], window_div=4)
    assert loaded.info == ref.info
    loaded.close(); ref.close()
    L.gsv_plan_destroy(plan); L.gsv_plan_recorder_destroy(pr); L.gsv_program_destroy(prog); L.gsv_recorder_destroy(rec)


@pytest.mark.gpu
def test_full_verifier_through_the_public_abi(ext_host, request, tmp_path):
    """BASELINE config 4 from an external host: record -> plan file through gsv_plan_recorder_* (1 257 calls, 212 unit programs, 11.46 B
    gates), gsv_plan_load, one instance garbled with retain_stream = 0 through gsv_session_garble_streaming_sink into the host's own
    CBC-MAC.  MAC and output label == the CPU oracle's flat-stream fixture; the plan file == the built-in builder's (the session's shared
    file, conftest.verifier_plan_file); build time and host RSS reported (profiles/r05_e2e/ext_host_verifier.json keeps one run)."""
    import plan_digest
    from conftest import HERE
    case = json.load(open(os.path.join(HERE, "golden", "groth16_verify_compressed_1pub_golden.json")))
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > 100e9 else str(tmp_path)
    a = os.path.join(d, "gsv_ext_host_%d_abi.gsvplan" % os.getpid())
    try:
        # the host process, its stderr followed: once its plan file is complete both files are digested here (CPU) while the process loads the
        # file and garbles (GPU + one serial CBC-MAC chain) — the suite has a time limit
        import threading
        proc = subprocess.Popen([ext_host, case["circuit"], ",".join(VERIFIER_UNITS), a, "--window-div", "4", "--garble", str(case["seed"]), "--no-engine-mac"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        digests, err_lines = {}, []

        def follow():
            for ln in proc.stderr:
                err_lines.append(ln)
                if ln.startswith("PLAN_FILE_READY") and "a" not in digests:
                    digests["a"] = plan_digest.digest(a, threads=8)
        th = threading.Thread(target=follow)
        th.start()
        # the session's built-in plan file (conftest.verifier_plan_file) is built NOW, beside the host process's own build through the ABI
        # (both CPU-bound: ~85 s side by side instead of 50 + 52 s one after the other)
        verifier_plan_file = request.getfixturevalue("verifier_plan_file")
        b = verifier_plan_file["path"]
        digests["b"] = plan_digest.digest(b, threads=8)
        out_text = proc.stdout.read()
        th.join()
        assert proc.wait() == 0, "".join(err_lines)
        j = json.loads(out_text.strip().splitlines()[-1])
        assert (j["n_gates"], j["n_ciphertexts"]) == (case["gates"], case["n_ciphertexts"]) == (11_456_865_898, 2_980_165_547)
        assert j["ct_hash"] == case["ct_hash"] == j["engine_ct_hash"] and j["sink_in_order"] and j["sink_records"] == case["n_ciphertexts"]
        out = bytes.fromhex(j["output_label0"])
        assert hashlib.sha256(out).hexdigest() == case["output_label0_sha256"] and out[:16].hex() == case["first_output_label0"]
        j["builtin_build_s"] = verifier_plan_file["build_s"]
        da, db = digests["a"], digests["b"]
        j["plan_digest"], j["builtin_plan_digest"] = da["digest"], db["digest"]
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            json.dump(j, open(os.path.join(out_dir, "ext_host_verifier.json"), "w"), indent=1)
        assert da == db
        assert j["build_s"] <= 2.0 * j["builtin_build_s"] + 10.0, j  # (both built side by side since round 6)
    finally:
        if os.path.exists(a):
            os.remove(a)


@pytest.mark.gpu
def test_destroy_from_the_sink_callback_mid_pass(ext_host, tmp_path):
    """A mode is dropped wherever the host drops it (src/circuit/modes.rs:26-51): the external host destroys a SECOND session and its plan
    from inside its `CiphertextHandler` callback, a third of the way through a pass that runs as ONE launch over a ciphertext ring.  Until
    round 5 that was a deadlock by construction (hipFree waits for the ring pass, the pass for the host's stream position, the position for
    this callback) that the device's watchdog ended after GSV_DEP_WAIT_SECONDS with a failed pass.  Now the destroy calls return at once
    (two deferred releases), the pass finishes in its normal time and MAC and output labels equal the run without the destroy and the CPU
    oracle's flat stream."""
    import oracle_lib as o
    units = ["fq2::mul_montgomery", "fq2::square_montgomery", "fp254::mul_by_constant_montgomery", "bigint::mul_karatsuba", "fp254::montgomery_reduce"]
    path = os.path.join(str(tmp_path), "mix.gsvplan")
    env = {"GSV_CT_RING_RECORDS": "1000000", "GSV_DEP_WAIT_SECONDS": "20"}
    plain = run_ext(ext_host, "fq12_mix", units, path, "--window-div", 1, "--garble", 77, "--ring", env=env)
    j = run_ext(ext_host, "fq12_mix", units, path, "--window-div", 1, "--garble", 77, "--ring", "--destroy-in-sink", env=env)
    ref = o.garble("fq12_mix", 77)
    assert j["destroyed_in_sink"] is True and j["deferred_releases"] == 2 and j["destroy_call_s"] < 0.5, j
    assert j["deferred_total_after_pass"] == 2, j
    assert j["ct_hash"] == j["engine_ct_hash"] == plain["ct_hash"] == ref.ct_hash.tobytes().hex() and j["sink_in_order"]
    assert bytes.fromhex(j["output_label0"]) == ref.output_label0.tobytes() == bytes.fromhex(plain["output_label0"])
    assert j["garble_s"] < 3.0 * plain["garble_s"] + 2.0, (j["garble_s"], plain["garble_s"])  # (the stall it replaces: GSV_DEP_WAIT_SECONDS, then a failed pass)
