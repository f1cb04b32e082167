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
            if f.endswith((".hpp", ".h", ".hip", ".cpp")):
                srcs.append(os.path.join(root, f))
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "gsv_engine.h"))
    return srcs


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


def check_step_barrier_isa(asm=None):
    """Every barrier of run_program_kernel must be a FULL one: the step barrier is a hand-written
    `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` (kernels.hip) — all label stores of the step, LDS and HBM, have completed before any
    wave reads them in the next step.  (Round 1 used counted waits, vmcnt(1|2), that relied on the compiler's instruction order;
    they are gone, and this check keeps them gone.)  On the ISA of every instantiation: the step loop, unrolled by two, shows at
    least two such pairs, and no `s_barrier` is directly preceded by a partial `s_waitcnt vmcnt(N > 0)`.
    Returns {kernel symbol: number of full step barriers}; raises RuntimeError on a violation."""
    import re
    asm = asm if asm is not None else disassemble_kernels()
    funcs, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None and line.startswith("\t"):
            funcs[cur].append(line.split("//")[0].strip())
    out = {}
    kernels = {k: v for k, v in funcs.items() if "run_program_kernel" in k}
    if len(kernels) != 6:
        raise RuntimeError("expected 6 instantiations of run_program_kernel, found %d" % len(kernels))
    for name, ins in kernels.items():
        full = 0
        for i, t in enumerate(ins):
            if t != "s_barrier" or i == 0:
                continue
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ins[i - 1])
            if m and int(m.group(1)) != 0:
                raise RuntimeError("%s: a barrier behind a partial wait `%s`" % (name, ins[i - 1]))
            if ins[i - 1] == "s_waitcnt vmcnt(0) lgkmcnt(0)":
                full += 1
        if full < 2:
            raise RuntimeError("%s: found %d full step barriers, expected at least 2" % (name, full))
        out[name] = full
    return out


def build(force=False, verbose=False):
    if os.environ.get("GSV_ENGINE_SO"):  # experiments: load a differently built library
        return os.environ["GSV_ENGINE_SO"]
    if not force and not _newer(OUT, _all_sources()):
        return OUT
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    k_o = os.path.join(ENG, "kernels.o")
    e_o = os.path.join(ENG, "engine.o")
    cmds = [
        [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-promote-alloca-to-lds", "-c", os.path.join(ENG, "kernels.hip"), "-o", k_o],
        ["g++", "-O2", "-std=c++17", "-fPIC", "-maes", "-msse2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
         "-Wall", "-Wno-unused-parameter", "-c", os.path.join(ENG, "engine.cpp"), "-o", e_o],
        [hipcc, "-shared", "-o", OUT, k_o, e_o],
    ]
    for i, c in enumerate(cmds):
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
        if i == 0:
            check_step_barrier_isa()  # refuse to link a kernel whose hand-counted step barrier the compiler has rearranged
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
