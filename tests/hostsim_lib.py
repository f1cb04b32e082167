"""ctypes binding for tests/hostsim (TEST-ONLY host interpreter of compiled engine programs)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "hostsim", "libgsv_hostsim.so")
_lib = None
INFO_FIELDS = ["n_inputs", "n_outputs", "n_gates", "n_ct", "n_dead", "n_steps", "and_depth", "n_and_steps", "max_step_width", "n_slots",
               "peak_live", "component_calls", "n_lds_slots", "reads_lds", "reads_hbm", "writes_lds", "writes_hbm", "n_fused_free"]


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "hostsim")], stdout=subprocess.DEVNULL)
        L = C.CDLL(_SO)
        L.hostsim_last_error.restype = C.c_char_p
        u8p = C.POINTER(C.c_uint8)
        L.hostsim_compile.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.hostsim_free.argtypes = [C.c_void_p]
        L.hostsim_run.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
        L.hostsim_plan_build.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.hostsim_plan_free.argtypes = [C.c_void_p]
        L.hostsim_plan_run.argtypes = [C.c_void_p, C.c_int, C.c_uint64, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
        L.hostsim_labels_from_seed.argtypes = [C.c_uint64, C.c_uint64, u8p]
        L.hostsim_cbcmac.argtypes = [u8p, C.c_uint64, u8p]
        L.hostsim_aes_ttable.argtypes = [u8p, u8p]
        L.hostsim_aes_portable.argtypes = [u8p, u8p]
        L.hostsim_hash.argtypes = [u8p, C.c_uint64, u8p]
        L.hostsim_sbox.argtypes = [u8p]
        L.hostsim_set_hasher.argtypes = [C.c_int]
        L.hostsim_blake3_hash.argtypes = [u8p, C.c_uint64, u8p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


class SimProgram:
    def __init__(self, spec, chain_feedback=False):
        h = C.c_void_p()
        info = np.zeros(len(INFO_FIELDS), np.uint64)
        if lib().hostsim_compile(spec.encode(), int(chain_feedback), C.byref(h), info.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError(lib().hostsim_last_error().decode())
        self.h = h
        self.info = dict(zip(INFO_FIELDS, (int(x) for x in info)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().hostsim_free(self.h)
            self.h = None

    def garble(self, delta, consts, inputs, replays=1, gid_base=0):
        n_out, n_ct = self.info["n_outputs"], self.info["n_ct"]
        cts = np.zeros((replays * n_ct, 16), np.uint8)
        out = np.zeros((n_out, 16), np.uint8)
        inputs = np.ascontiguousarray(inputs, np.uint8)
        consts = np.ascontiguousarray(consts, np.uint8)
        delta = np.ascontiguousarray(delta, np.uint8)
        if lib().hostsim_run(self.h, 0, replays, gid_base, _p(delta), _p(consts), _p(inputs), None, _p(cts), _p(out), None):
            raise RuntimeError(lib().hostsim_last_error().decode())
        return out, cts

    def evaluate(self, consts_active, inputs_active, input_bits, cts, replays=1, gid_base=0):
        n_out = self.info["n_outputs"]
        out = np.zeros((n_out, 16), np.uint8)
        bits = np.zeros(n_out, np.uint8)
        cts = np.ascontiguousarray(cts, np.uint8).copy()
        ia = np.ascontiguousarray(inputs_active, np.uint8)
        ib = np.ascontiguousarray(input_bits, np.uint8)
        ca = np.ascontiguousarray(consts_active, np.uint8)
        z = np.zeros(16, np.uint8)
        if lib().hostsim_run(self.h, 1, replays, gid_base, _p(z), _p(ca), _p(ia), _p(ib), _p(cts), _p(out), _p(bits)):
            raise RuntimeError(lib().hostsim_last_error().decode())
        return out, bits


def labels_from_seed(seed, n):
    out = np.zeros((n, 16), np.uint8)
    lib().hostsim_labels_from_seed(seed, n, _p(out))
    return out


def cbcmac(cts):
    a = np.ascontiguousarray(cts, np.uint8).reshape(-1)
    out = np.zeros(16, np.uint8)
    lib().hostsim_cbcmac(_p(a) if a.size else None, a.size // 16, _p(out))
    return out.tobytes()


def aes_ttable(block):
    i = np.frombuffer(bytes(block), np.uint8).copy()
    o = np.zeros(16, np.uint8)
    lib().hostsim_aes_ttable(_p(i), _p(o))
    return o.tobytes()


def aes_portable(block):
    i = np.frombuffer(bytes(block), np.uint8).copy()
    o = np.zeros(16, np.uint8)
    lib().hostsim_aes_portable(_p(i), _p(o))
    return o.tobytes()


def hash_with_gate(label, gid):
    i = np.frombuffer(bytes(label), np.uint8).copy()
    o = np.zeros(16, np.uint8)
    lib().hostsim_hash(_p(i), gid, _p(o))
    return o.tobytes()


def sbox():
    o = np.zeros(256, np.uint8)
    lib().hostsim_sbox(_p(o))
    return o


def set_hasher(kind):
    lib().hostsim_set_hasher({"aes": 0, "blake3": 1}[kind])


def blake3_hash(label, gid):
    i = np.frombuffer(bytes(label), np.uint8).copy()
    o = np.zeros(16, np.uint8)
    lib().hostsim_blake3_hash(_p(i), gid, _p(o))
    return o.tobytes()


PLAN_FIELDS = ["n_inputs", "n_outputs", "n_gates", "n_ct", "n_calls", "n_programs", "n_globals", "n_unit_programs"]


class SimPlan:
    """A circuit recorded with some components as calls of separately compiled programs (plan_builder.hpp), interpreted
    call by call on the host."""

    def __init__(self, spec, units):
        h = C.c_void_p()
        info = np.zeros(len(PLAN_FIELDS), np.uint64)
        if lib().hostsim_plan_build(spec.encode(), ",".join(units).encode(), C.byref(h), info.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError(lib().hostsim_last_error().decode())
        self.h = h
        self.info = dict(zip(PLAN_FIELDS, (int(x) for x in info)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().hostsim_plan_free(self.h)

    def schedule(self, max_calls, max_slots=0, window_ct=0, window_calls=0):
        """Compute (and verify against the hazard rules) the engine's call-level schedule; garble / evaluate then execute it with the
        freedom of the device's dataflow execution (any order the dependencies allow).  Returns {n_windows, n_dependencies, max_width,
        scratch_slots, critical_steps, total_steps, max_window_ct}."""
        info = np.zeros(7, np.uint64)
        if lib().hostsim_plan_schedule(self.h, C.c_uint32(max_calls), C.c_uint64(max_slots), C.c_uint64(window_ct), C.c_uint32(window_calls), info.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError(lib().hostsim_last_error().decode())
        return dict(zip(["n_windows", "n_dependencies", "max_width", "scratch_slots", "critical_steps", "total_steps", "max_window_ct"], (int(x) for x in info)))

    def ring(self, segment_ct, ring_ct, **kw):
        """schedule(**kw) with drain segments and a ciphertext ring of `ring_ct` records (schedule.hpp, SchedParams::ring_ct; the
        schedule's own verification simulates the ring record by record).  Returns (info, segments, [(ring_off, ring_need, seg_end, ovl0, ovl1)] per call)."""
        lib().hostsim_set_ring_ct.argtypes = [C.c_uint64]
        lib().hostsim_plan_ring.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64]
        lib().hostsim_plan_ring.restype = C.c_uint64
        lib().hostsim_set_ring_ct(ring_ct)
        try:
            info, segs = self.segments(segment_ct, **kw)
        finally:
            lib().hostsim_set_ring_ct(0)
        n = int(lib().hostsim_plan_ring(self.h, None, 0))
        out = np.zeros((n, 5), np.uint64)
        lib().hostsim_plan_ring(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return info, segs, [tuple(int(x) for x in row) for row in out]

    def segments(self, segment_ct, **kw):
        """schedule(**kw) with drain segments of at most `segment_ct` ciphertext records (schedule.hpp, SchedParams::segment_ct):
        returns (schedule info, [(window, call0, call1, ct0, n_ct)])."""
        lib().hostsim_set_segment_ct.argtypes = [C.c_uint64]
        lib().hostsim_plan_segments.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64]
        lib().hostsim_plan_segments.restype = C.c_uint64
        lib().hostsim_set_segment_ct(segment_ct)
        try:
            info = self.schedule(**kw)
        finally:
            lib().hostsim_set_segment_ct(0)
        n = int(lib().hostsim_plan_segments(self.h, None, 0))
        out = np.zeros((n, 5), np.uint64)
        lib().hostsim_plan_segments(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return info, [tuple(int(x) for x in row) for row in out]

    def garble(self, delta, consts, inputs, gid_base=0):
        cts = np.zeros((self.info["n_ct"], 16), np.uint8)
        out = np.zeros((self.info["n_outputs"], 16), np.uint8)
        if lib().hostsim_plan_run(self.h, 0, gid_base, _p(np.ascontiguousarray(delta, np.uint8)), _p(np.ascontiguousarray(consts, np.uint8)),
                                  _p(np.ascontiguousarray(inputs, np.uint8)), None, _p(cts), _p(out), None):
            raise RuntimeError(lib().hostsim_last_error().decode())
        return out, cts

    def evaluate(self, consts_active, inputs_active, input_bits, cts, gid_base=0):
        out = np.zeros((self.info["n_outputs"], 16), np.uint8)
        bits = np.zeros(self.info["n_outputs"], np.uint8)
        cts = np.ascontiguousarray(cts, np.uint8).copy()
        z = np.zeros(16, np.uint8)
        if lib().hostsim_plan_run(self.h, 1, gid_base, _p(z), _p(np.ascontiguousarray(consts_active, np.uint8)), _p(np.ascontiguousarray(inputs_active, np.uint8)),
                                  _p(np.ascontiguousarray(input_bits, np.uint8)), _p(cts), _p(out), _p(bits)):
            raise RuntimeError(lib().hostsim_last_error().decode())
        return out, bits


def trace(spec, cap=40_000_000):
    """The raw gate stream the product's recorder (RecordMode under the two-pass driver, program.hpp) sees for a named circuit:
    (type[n], a[n], b[n], c[n]) as SSA ids — 0 / 1 the constants, c == 0xFFFFFFFF for a dead gate — plus the SSA ids of the circuit's
    inputs and outputs.  n_inputs / n_outputs come from compiling the circuit once (SimProgram)."""
    info = SimProgram(spec).info
    t = np.zeros(cap, np.uint8)
    a, b, c = (np.zeros(cap, np.uint32) for _ in range(3))
    n, nw = C.c_uint64(), C.c_uint32()
    ins, outs = np.zeros(info["n_inputs"], np.uint32), np.zeros(info["n_outputs"], np.uint32)
    u32p = C.POINTER(C.c_uint32)
    rc = lib().hostsim_trace(spec.encode(), C.c_uint64(cap), t.ctypes.data_as(C.POINTER(C.c_uint8)), a.ctypes.data_as(u32p), b.ctypes.data_as(u32p), c.ctypes.data_as(u32p), C.byref(n), C.byref(nw),
                             ins.ctypes.data_as(u32p), outs.ctypes.data_as(u32p))
    if rc:
        raise RuntimeError("trace of %s: %s" % (spec, "capacity too small" if rc == 2 else lib().hostsim_last_error().decode()))
    k = n.value
    return t[:k], a[:k], b[:k], c[:k], ins, outs
