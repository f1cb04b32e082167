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
    """The step barrier of run_program_kernel is a hand-counted `s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier` (kernels.hip): it
    waits for the step's label stores but not for the record prefetch issued just before it (N = 1) nor, when the wave's last
    vector-memory store was a ciphertext, for that store (N = 2).  That is only right while the compiler keeps the prefetch load
    the YOUNGEST vector-memory operation in front of the barrier.  Checked on the ISA of every instantiation, so that a compiler
    bump cannot silently break it:
      * each kernel holds exactly two `s_waitcnt vmcnt(1) lgkmcnt(0)` + `s_barrier` pairs (the step loop is unrolled by two:
        ping-pong record registers), garbling kernels also exactly two vmcnt(2) pairs, each immediately followed by its s_barrier;
      * walking back from the vmcnt(1) pair, the first vector-memory instruction is the 16-byte record prefetch
        (global_load_dwordx4) and no label store (global_store / ds_write) sits between it and the barrier.
    Returns {kernel symbol: (n_vmcnt1, n_vmcnt2)}; raises RuntimeError on a violation."""
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
        garble = "ILb0E" in name  # run_program_kernel<false, ...>
        n = {1: 0, 2: 0}
        for i, t in enumerate(ins):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)$", t)
            if not m or int(m.group(1)) not in (1, 2):
                continue
            k = int(m.group(1))
            if i + 1 >= len(ins) or ins[i + 1] != "s_barrier":
                continue  # an ordinary compiler-generated wait
            n[k] += 1
            if k == 1:
                j = i - 1
                while j >= 0 and not re.match(r"(global_|buffer_|scratch_|flat_)", ins[j]):
                    if re.match(r"ds_write|ds_store", ins[j]):
                        raise RuntimeError("%s: an LDS label store sits between the record prefetch and the step barrier" % name)
                    j -= 1
                if j < 0 or not ins[j].startswith("global_load_dwordx4"):
                    raise RuntimeError("%s: the youngest vector-memory operation before the step barrier is `%s`, not the record prefetch" % (name, ins[j] if j >= 0 else "none"))
        if n[1] != 2 or n[2] != (2 if garble else 0):
            raise RuntimeError("%s: found %d / %d counted step barriers (vmcnt 1 / 2), expected 2 / %d" % (name, n[1], n[2], 2 if garble else 0))
        out[name] = (n[1], n[2])
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
