"""bindings/rust/ cannot be compiled in this image (no Rust toolchain); what CAN be checked without one is checked here:

  * every `extern "C"` declaration of gpu_ffi.rs names a function of include/gsv_engine.h with the same parameters (count and type,
    C type by C type), and the `#[repr(C)]` structs have the C structs' fields in the C order;
  * every `ffi::gsv_*` the mode files call is declared in gpu_ffi.rs;
  * streaming_mode_unit_hook.patch applies, hunk by hunk, to the reference's own src/circuit/{component_key,modes,streaming_mode,mod}.rs
    (copies in a scratch directory; skipped where /root/reference does not exist) and the patched files hold the names the shim uses.
The call sequence itself is compiled and run as tests/ext_host/ext_host.cpp (tests/test_ext_host.py)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "bindings", "rust")
REF = "/root/reference"

# C parameter type (whitespace-normalised, parameter name stripped) -> Rust type in gpu_ffi.rs
C2RUST = {
    "int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint16_t": "u16",
    "const char*": "*const c_char", "const uint8_t*": "*const u8", "uint8_t*": "*mut u8", "const uint64_t*": "*const u64", "uint64_t*": "*mut u64",
    "const uint32_t*": "*const u32", "uint32_t*": "*mut u32", "void*": "*mut std::ffi::c_void",
    "gsv_ct_sink_fn": "GsvCtSinkFn", "gsv_ct_source_fn": "GsvCtSourceFn",
    "const gsv_gate*": "*const GsvGate", "const gsv_compile_opts*": "*const GsvCompileOpts", "const gsv_plan_recorder_opts*": "*const GsvPlanRecorderOpts",
    "const gsv_plan_session_opts*": "*const GsvPlanSessionOpts", "gsv_plan_schedule_info*": "*mut GsvPlanScheduleInfo", "int*": "*mut c_int",
}
for c, r in (("gsv_recorder", "GsvRecorder"), ("gsv_program", "GsvProgram"), ("gsv_engine", "GsvEngine"), ("gsv_session", "GsvSession"), ("gsv_plan", "GsvPlan"),
             ("gsv_plan_recorder", "GsvPlanRecorder")):
    C2RUST[c + "*"] = "*mut " + r
    C2RUST["const " + c + "*"] = "*const " + r
    C2RUST[c + "**"] = "*mut *mut " + r


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_functions():
    hdr = strip_comments(open(os.path.join(ROOT, "include", "gsv_engine.h")).read())
    out = {}
    for m in re.finditer(r"\b(?:int|void|uint64_t|const char\s*\*)\s*(gsv_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        name, params = m.group(1), " ".join(m.group(2).split())
        types = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.match(r"(.*?)\s*\b[A-Za-z_][A-Za-z_0-9]*\s*\[\d*\]$", p)  # `uint8_t hash[16]` is a pointer parameter
                t = arr.group(1) + "*" if arr else re.sub(r"\s*\b[A-Za-z_][A-Za-z_0-9]*$", "", p) if not p.endswith("*") else p  # drop the parameter's name
                t = re.sub(r"\s*\*", "*", t).strip()
                types.append(t)
        out[name] = types
    return out


def rust_functions():
    src = strip_comments(open(os.path.join(RUST, "src", "gpu_ffi.rs")).read())
    block = src[src.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (gsv_[a-z_0-9]+)\s*\(([^)]*)\)", block):
        params = [p.strip() for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = [p.split(":", 1)[1].strip() for p in params]
    return out


def test_ffi_declarations_match_the_header():
    c, r = c_functions(), rust_functions()
    assert len(r) >= 55 and len(c) >= 70
    for name, rtypes in r.items():
        assert name in c, "gpu_ffi.rs declares %s, which include/gsv_engine.h does not" % name
        ctypes_ = c[name]
        assert len(ctypes_) == len(rtypes), (name, ctypes_, rtypes)
        for ct, rt in zip(ctypes_, rtypes):
            assert ct in C2RUST, "no Rust mapping for C type %r (%s)" % (ct, name)
            assert C2RUST[ct] == rt, (name, ct, rt)


def c_struct_fields(name):
    hdr = strip_comments(open(os.path.join(ROOT, "include", "gsv_engine.h")).read())
    body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), hdr, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*(?:\[\d+\])?(?:\s*,\s*[A-Za-z_][A-Za-z_0-9]*)*)$", decl)
        ctype = re.sub(r"\s*\*", "*", m.group(1).strip())
        for f in m.group(2).split(","):
            fields.append((f.strip(), ctype))
    return fields


def rust_struct_fields(name):
    src = strip_comments(open(os.path.join(RUST, "src", "gpu_ffi.rs")).read())
    m = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct %s\s*\{(.*?)\}" % name, src, flags=re.S)
    return [(f.split(":")[0].replace("pub", "").strip(), f.split(":", 1)[1].strip()) for f in m.group(1).split(",") if ":" in f]


@pytest.mark.parametrize("cname,rname", [("gsv_gate", "GsvGate"), ("gsv_plan_session_opts", "GsvPlanSessionOpts"), ("gsv_compile_opts", "GsvCompileOpts"),
                                         ("gsv_plan_recorder_opts", "GsvPlanRecorderOpts"), ("gsv_plan_schedule_info", "GsvPlanScheduleInfo")])
def test_repr_c_structs_match_the_header(cname, rname):
    scalar = {"int": "c_int", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "const char*": "*const c_char", "gsv_plan_recorder*": "*mut GsvPlanRecorder"}
    cf, rf = c_struct_fields(cname), rust_struct_fields(rname)
    assert len(cf) == len(rf), (cf, rf)
    for (cn, ct), (rn, rt) in zip(cf, rf):
        arr = re.match(r"([a-z_0-9]+)\[(\d+)\]$", cn)
        if arr:
            assert rn == arr.group(1) and rt == "[%s; %s]" % (scalar[ct], arr.group(2)), (cn, ct, rn, rt)
        else:
            assert rn == cn and rt == scalar[ct], (cn, ct, rn, rt)


def test_mode_files_call_only_declared_functions():
    declared = set(rust_functions())
    for f in ("gpu_recorder.rs", "gpu_garble_mode.rs", "gpu_evaluate_mode.rs"):
        src = strip_comments(open(os.path.join(RUST, "src", f)).read())
        used = set(re.findall(r"\b(gsv_[a-z_0-9]+)\s*\(", src))
        assert used and used <= declared, (f, used - declared)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "circuit")) or shutil.which("patch") is None, reason="the reference tree (or patch) is not on this machine")
def test_unit_hook_patch_applies_to_the_reference_sources(tmp_path):
    """Copies of the four files the patch touches, patched in a scratch directory: every hunk applies, nothing is rejected."""
    dst = tmp_path / "src" / "circuit"
    dst.mkdir(parents=True)
    for f in ("component_key.rs", "modes.rs", "streaming_mode.rs", "mod.rs"):
        shutil.copy(os.path.join(REF, "src", "circuit", f), dst / f)
    with open(os.path.join(RUST, "streaming_mode_unit_hook.patch")) as fh:
        r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-d", str(tmp_path)], stdin=fh, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    assert not list(tmp_path.rglob("*.rej"))
    modes = (dst / "modes.rs").read_text()
    for name in ("fn execution_finished", "fn unit_begin", "fn unit_end", "fn unit_call", "pub mod gpu_garble_mode", "pub mod gpu_evaluate_mode", "pub use gpu_recorder::UnitAction"):
        assert name in modes, name
    sm = (dst / "streaming_mode.rs").read_text()
    assert "ctx.mode.unit_begin(key, &component_name(&key), &liveness)" in sm and "ctx.mode.unit_call(key, &liveness, &input_wires)" in sm
    assert "pub fn component_name" in (dst / "component_key.rs").read_text()
    assert "execution_finished(&output_wires)" in (dst / "mod.rs").read_text()
    # the trait methods the shim implements are the ones the patched trait declares
    for f in ("gpu_garble_mode.rs", "gpu_evaluate_mode.rs"):
        src = open(os.path.join(RUST, "src", f)).read()
        for meth in ("fn evaluate_gate", "fn allocate_wire", "fn lookup_wire", "fn feed_wire", "fn add_credits", "fn execution_finished"):
            assert meth in src, (f, meth)
