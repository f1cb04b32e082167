"""Builds libgsv_engine.so (HIP kernels for gfx950 + host runtime + C ABI) in-tree.

  kernels.hip  -> hipcc --offload-arch=gfx950          (device code; cross-compiles without a GPU)
  engine.cpp   -> g++ -maes (host runtime, AES-NI CBC-MAC) against the HIP runtime headers
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ENG = os.path.join(CSRC, "engine")
OUT = os.path.join(HERE, "libgsv_engine.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _all_sources():
    srcs = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hpp", ".h", ".hip", ".cpp", ".ipp")):
                srcs.append(os.path.join(root, f))
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "gsv_engine.h"))
    return srcs


def source_sha256():
    """sha256 over the engine's sources (path relative to the package + contents, in sorted order).  The same tree gives the same value
    on any machine and in any directory; the library's own hash does not (build paths are compiled in).  profiles/*/traffic.json carries
    both, bench.py accepts either as "these counters belong to this build"."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.dirname(HERE)
    for path in sorted(_all_sources(), key=lambda q: os.path.relpath(q, base)):
        h.update(os.path.relpath(path, base).replace(os.sep, "/").encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def disassemble_kernels(obj=None):
    """gfx950 ISA of the kernels object as text (llvm-objdump of the code object inside the offload bundle)."""
    import shutil
    import tempfile
    obj = obj or os.path.join(ENG, "kernels.o")
    objdump = os.path.join(ROCM, "lib", "llvm", "bin", "llvm-objdump")
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "kernels.o")
        shutil.copy(obj, tmp)
        subprocess.check_call([objdump, "--offloading", tmp], stdout=subprocess.DEVNULL, cwd=td)  # writes <tmp>.0.hipv4-amdgcn-amd-amdhsa--gfx950
        co = [f for f in os.listdir(td) if "gfx950" in f]
        if len(co) != 1:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        return subprocess.check_output([objdump, "-d", os.path.join(td, co[0])], text=True)


def _functions(asm):
    import re
    funcs, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None and line.startswith("\t"):
            funcs[cur].append(line.split("//")[0].strip())
    return funcs


def check_step_barrier_isa(asm=None):
    """The step barrier of run_program_kernel is `s_waitcnt lgkmcnt(0); s_barrier` (kernels.hip): it waits for the wave's LDS
    label stores and scalar loads but for NO vector-memory operation — label stores to the HBM wire file are visible to the
    other waves of the workgroup in issue order (workgroup-scope release/acquire of the AMDGPU memory model outside
    threadgroup-split mode; check_workgroup_release_model pins that on the compiler).  Checked on the ISA of every
    instantiation, so that a compiler bump cannot silently put the store-acknowledgement wait back:
      * each kernel holds exactly two `s_waitcnt lgkmcnt(0)` + `s_barrier` pairs (the step loop is unrolled by two: ping-pong
        record registers);
      * in each of them the instruction in front of the wait — looking through scalar bookkeeping (SGPR spill reloads from VGPR lanes,
        scalar ALU, s_setprio) — is the 16-byte record prefetch (global_load_dwordx4): the prefetch crosses the barrier in flight and
        nothing waits for vector memory in between.  (Other lgkmcnt(0) + s_barrier
        pairs are the compiler's own __syncthreads of the prologue / replay epilogue.)
    Returns {kernel symbol: number of step barriers}; raises RuntimeError on a violation."""
    asm = asm if asm is not None else disassemble_kernels()
    out = {}
    kernels = {k: v for k, v in _functions(asm).items() if "run_program_kernel" in k}
    if len(kernels) != 16:
        raise RuntimeError("expected 16 instantiations of run_program_kernel, found %d" % len(kernels))
    for name, ins in kernels.items():
        def prefetch_in_front(i):
            # the instruction in front of the wait, looking through scalar bookkeeping the compiler may put there (reloads of spilled
            # SGPRs from VGPR lanes, scalar ALU, and — since the FW instantiations choose between two barrier forms — scalar branches and
            # the other form's own scalar instructions): it must be the record prefetch, and nothing in between may wait for vector memory
            j = i - 1
            while j > 0 and (ins[j].startswith(("v_readlane_b32", "v_writelane_b32", "s_nop", "s_setprio")) or (ins[j].startswith("s_") and "vmcnt" not in ins[j] and not ins[j].startswith("s_endpgm"))):
                j -= 1
            return ins[j].startswith("global_load_dwordx4")
        grouped = "ELb1EEE" in name and ("ELi2ELi0" in name or "ELi4ELi0" in name)  # FW instantiations with several instances per workgroup: per-group barrier
        n = sum(1 for i, t in enumerate(ins) if t == "s_waitcnt lgkmcnt(0)" and 0 < i < len(ins) - 1 and ins[i + 1] == "s_barrier" and prefetch_in_front(i))
        if n != (0 if grouped else 2):
            raise RuntimeError("%s: found %d step barriers with the record prefetch issued right in front of them, expected %d" % (name, n, 0 if grouped else 2))
        # (the one legitimate vmcnt wait in front of a barrier is the dataflow epilogue's agent-scope release: buffer_wbl2 ; s_waitcnt vmcnt(0))
        if any(t.startswith("s_waitcnt vmcnt") and i + 1 < len(ins) and ins[i + 1] == "s_barrier" and not (i and ins[i - 1].startswith("buffer_wbl2")) for i, t in enumerate(ins)):
            raise RuntimeError("%s: a barrier waits for vector memory" % name)
        # Per-group step barrier (round 6: FW instantiations with several instances per workgroup): the arrive is `s_waitcnt lgkmcnt(0)`,
        # lane 0's `ds_add_u32`, then a poll loop over `ds_read_b32` — all on VGPRs of its own.  Exactly two of them (the step loop is
        # unrolled by two), each reached from the record prefetch without a vector-memory wait (a barrier that waited for the step's store
        # acknowledgements would cost ~1 us per step), and the poll loop holds no vector-memory wait either.
        arrives = [i for i, t in enumerate(ins) if t.startswith("ds_add_u32")]
        if grouped:
            if len(arrives) != 2:
                raise RuntimeError("%s: found %d per-group barrier arrivals (ds_add_u32), expected 2" % (name, len(arrives)))
            for i in arrives:
                if not (ins[i - 1] == "s_mov_b64 exec, 1" and prefetch_in_front(i)):
                    raise RuntimeError("%s: a per-group barrier is not reached from the record prefetch without a vector-memory wait" % name)
                loop = ins[i + 1:i + 12]
                if not any(t.startswith("ds_read_b32") for t in loop) or any("vmcnt" in t for t in loop):
                    raise RuntimeError("%s: unexpected per-group barrier poll loop: %r" % (name, loop))
        elif arrives:
            raise RuntimeError("%s: a per-group barrier in an instantiation that should not have one" % name)
        out[name] = n
    return out


_RELEASE_PROBE = r"""
#include <hip/hip_runtime.h>
__global__ void gsv_probe(unsigned* __restrict__ a, unsigned* __restrict__ out) {
  a[threadIdx.x] = threadIdx.x * 3u;
  __syncthreads();
  out[threadIdx.x] = a[(threadIdx.x + 64) & 1023];
}
"""


def check_workgroup_release_model():
    """What the step barrier relies on, pinned on the compiler: for `global store; __syncthreads(); global load of another
    wave's element` hipcc emits, on gfx950, no `s_waitcnt vmcnt` between the store and s_barrier and no cache invalidate
    (buffer_inv / buffer_wbl2) around it — i.e. the toolchain's own workgroup-scope release/acquire does not wait for
    store acknowledgements.  Returns the instructions of the probe kernel; raises RuntimeError if the toolchain stops doing so
    (then the step barrier of kernels.hip must wait for vmcnt as well)."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src, obj = os.path.join(td, "probe.hip"), os.path.join(td, "kernels.o")
        open(src, "w").write(_RELEASE_PROBE)
        subprocess.check_call([os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-c", src, "-o", obj], stderr=subprocess.DEVNULL)
        ins = [v for k, v in _functions(disassemble_kernels(obj)).items() if "gsv_probe" in k][0]
        # the model holds outside threadgroup-split mode only: the default, which build() never changes (no -mtgsplit)
        asm_text = subprocess.check_output([os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", src, "-o", "-"], stderr=subprocess.DEVNULL, text=True)
        if ".amdhsa_tg_split 0" not in asm_text or ".amdhsa_tg_split 1" in asm_text:
            raise RuntimeError("hipcc builds gfx950 kernels in threadgroup-split mode by default now: the step barrier must wait for vmcnt")
    st = [i for i, t in enumerate(ins) if t.startswith("global_store")][0]
    ba = ins.index("s_barrier")
    ld = [i for i, t in enumerate(ins) if t.startswith("global_load")][0]
    if not st < ba < ld:
        raise RuntimeError("probe kernel: unexpected order of store / barrier / load")
    for t in ins[st:ld]:
        if re.match(r"s_waitcnt .*vmcnt", t) or t.startswith(("buffer_inv", "buffer_wbl2")):
            raise RuntimeError("hipcc now emits `%s` inside a workgroup release/acquire on gfx950: the step barrier's memory model no longer holds" % t)
    return ins[: ins.index("s_endpgm") + 1]


def build(force=False, verbose=False, diag=False, variant=None, defines=()):
    """diag=True builds libgsv_engine_diag.so instead: the same library with the kernel's timing ablations compiled in
    (-DGSV_DIAG_BUILD; GSV_DIAG=<bits> then takes effect, see kernel_api.h) — load it with GSV_ENGINE_SO for experiments."""
    if os.environ.get("GSV_ENGINE_SO") and not diag:  # experiments: load a differently built library
        return os.environ["GSV_ENGINE_SO"]
    out = OUT.replace(".so", "_diag.so") if diag else OUT
    if variant:  # kernel A/B experiments: libgsv_engine_<variant>.so built with extra -D flags, loaded through GSV_ENGINE_SO
        out = OUT.replace(".so", "_%s.so" % variant)
    if not force and not _newer(out, _all_sources()):
        return out
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    k_o = os.path.join(ENG, "kernels_diag.o" if diag else ("kernels_%s.o" % variant if variant else "kernels.o"))
    e_o = os.path.join(ENG, "engine.o")
    cmds = [
        [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-promote-alloca-to-lds"] + (["-DGSV_DIAG_BUILD"] if diag else []) + list(defines) +
        ["-c", os.path.join(ENG, "kernels.hip"), "-o", k_o],
        ["g++", "-O2", "-std=c++17", "-fPIC", "-maes", "-msse2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
         "-Wall", "-Wno-unused-parameter", "-c", os.path.join(ENG, "engine.cpp"), "-o", e_o],
        [hipcc, "-shared", "-o", out, k_o, e_o],
    ]
    for i, c in enumerate(cmds):
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
        if i == 0:
            _check_kernels_object(k_o, production=not diag and not variant)
    return out


def _check_kernels_object(k_o, production):
    """ISA checks on the freshly compiled kernels object, before it is linked.
      * threadgroup-split mode is a CORRECTNESS precondition of the vmcnt-free step barrier: a detection raises.  A missing / renamed
        llvm tool only warns (the import must not depend on binutils being installed; tests/test_engine_host.py asserts the property).
      * the step-barrier shape (record prefetch right in front of `s_waitcnt lgkmcnt(0); s_barrier`, no vmcnt wait) is what keeps a
        compiler bump from silently shipping a barrier that waits for store acknowledgements: fatal for the production library
        (GSV_ALLOW_BARRIER_ISA_DRIFT=1 turns it into a warning for experiments), a warning for --diag / --variant builds."""
    try:
        asm = disassemble_kernels(k_o)
        tools_ok = True
    except (OSError, subprocess.CalledProcessError) as e:
        print("garbled_snark_verifier_amd.build: warning: cannot disassemble the kernels object (%s): ISA checks skipped" % e, file=sys.stderr)
        asm, tools_ok = None, False
    if tools_ok:
        try:
            check_not_tgsplit(k_o)
        except (OSError, subprocess.CalledProcessError) as e:
            print("garbled_snark_verifier_amd.build: warning: threadgroup-split check skipped (%s)" % e, file=sys.stderr)
        try:
            check_step_barrier_isa(asm)
        except RuntimeError as e:
            if production and os.environ.get("GSV_ALLOW_BARRIER_ISA_DRIFT") != "1":
                raise RuntimeError("step-barrier ISA check failed, library not linked (GSV_ALLOW_BARRIER_ISA_DRIFT=1 to override): %s" % e)
            print("garbled_snark_verifier_amd.build: warning: step-barrier ISA check: %s" % e, file=sys.stderr)


def check_not_tgsplit(obj=None):
    """The step barrier (`s_waitcnt lgkmcnt(0); s_barrier`, no vmcnt) relies on the waves of a workgroup sharing their CU's vector L1,
    i.e. on the code object NOT being built for threadgroup-split mode (AMDGPU memory model, workgroup-scope release/acquire).
    Asserted on the real kernels object's metadata / target features: raises RuntimeError if any kernel has tg_split set (e.g.
    HIPCC_COMPILE_FLAGS_APPEND=-mtgsplit)."""
    import shutil
    import tempfile
    obj = obj or os.path.join(ENG, "kernels.o")
    bin_ = os.path.join(ROCM, "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "kernels.o")
        shutil.copy(obj, tmp)
        subprocess.check_call([os.path.join(bin_, "llvm-objdump"), "--offloading", tmp], stdout=subprocess.DEVNULL, cwd=td)
        co = [f for f in os.listdir(td) if "gfx950" in f]
        if len(co) != 1:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        notes = subprocess.check_output([os.path.join(bin_, "llvm-readelf"), "--notes", os.path.join(td, co[0])], text=True)
    if "tgsplit" in notes.replace("-", "").replace("_", "").lower() and ("tgsplit+" in notes or ".uses_tg_split: true" in notes or "tg_split: 1" in notes):
        raise RuntimeError("kernels.o is built for threadgroup-split mode: the vmcnt-free step barrier is not valid there")
    # amdhsa.target carries the target id (features such as sramecc / xnack / tgsplit show up there when set)
    for line in notes.splitlines():
        if "amdhsa.target" in line and "tgsplit" in line:
            raise RuntimeError("kernels.o target has tgsplit: %s" % line.strip())
    return True


if __name__ == "__main__":
    _var = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")]
    _defs = [a for a in sys.argv if a.startswith("-D")]
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv, variant=_var[0] if _var else None, defines=_defs))
