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
    for c in cmds:
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
