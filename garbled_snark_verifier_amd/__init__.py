"""MI355X-native garbling / evaluation engine for the streaming Free-XOR / half-gates gate loop of
BitVM/garbled-snark-verifier.

This module is the thin Python host layer over the C ABI in ``include/gsv_engine.h``
(``libgsv_engine.so``: hand-written HIP kernels for gfx950 + C++ host runtime).  It mirrors the
reference's entry points for the hot path

    CircuitBuilder::streaming_garbling    src/circuit/mod.rs:180-203   ->  CircuitBuilder.streaming_garbling
    CircuitBuilder::streaming_evaluation  src/circuit/mod.rs:225-249   ->  CircuitBuilder.streaming_evaluation
    AESAccumulatingHash                   src/ciphertext_hasher.rs     ->  StreamingResult.ciphertext_hash

There is no CPU fallback: every garble/evaluate call runs on a HIP device or raises ``GsvError``.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

from . import build as _build

__all__ = ["GsvError", "lib", "Program", "Engine", "Session", "CircuitBuilder", "StreamingResult", "labels_from_seed", "GATE_NAMES", "cbcmac", "cbcmac_many",
           "write_gc_file", "read_gc_file", "gc_file_name"]

GATE_NAMES = ["And", "Nand", "Nimp", "Imp", "Ncimp", "Cimp", "Nor", "Or", "Xor", "Xnor", "Not"]
_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class GsvError(RuntimeError):
    pass


# Device objects are released in dependency order BEFORE the interpreter tears down: a session collected during interpreter
# shutdown would call into a HIP runtime that has already destroyed its own state (observed: std::bad_variant_access inside hipFree).
_live = {"session": weakref.WeakSet(), "plan": weakref.WeakSet(), "program": weakref.WeakSet(), "engine": weakref.WeakSet()}


class _gc_paused:
    """Cyclic garbage collection paused for the duration of a streaming call.  A streaming pass runs Python on OTHER threads (the sink /
    source callbacks) and, in ring mode, its device calls wait for the host's progress.  If the collector fires on such a thread and
    finalizes a forgotten Session / Plan / Program, `gsv_*_destroy` -> hipFree synchronises the WHOLE device: it waits for the running
    window, which waits for the host's stream position, which waits for the callback that is stuck in hipFree — until the device's
    watchdog ends the pass after GSV_DEP_WAIT_SECONDS with GSV_ERR_DEVICE (seen once in round 5's full GPU suite).
    Since round 6 the engine itself defers every release requested while a streaming pass is in flight (engine.cpp, ReleaseGate;
    include/gsv_engine.h "Deferred release"), so a destroy from a callback is harmless whoever issues it; the pause stays as a second
    layer (it also keeps collector pauses out of the callbacks, which sit on the drain's critical path)."""

    _lock = __import__("threading").Lock()
    _depth = 0
    _was_enabled = False

    def __enter__(self):
        import gc
        with _gc_paused._lock:
            if _gc_paused._depth == 0:
                _gc_paused._was_enabled = gc.isenabled()
                gc.disable()
            _gc_paused._depth += 1

    def __exit__(self, *exc):
        import gc
        with _gc_paused._lock:
            _gc_paused._depth -= 1
            if _gc_paused._depth == 0 and _gc_paused._was_enabled:
                gc.enable()
        return False


@atexit.register
def _close_everything():
    for kind in ("session", "plan", "program", "engine"):
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:
                pass


class _ProgramInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_inputs", "n_outputs", "n_gates", "n_ciphertexts", "n_dead")] + [("gate_count", C.c_uint64 * 11)] + [
        (n, C.c_uint64) for n in ("n_steps", "and_depth", "n_and_steps", "max_step_width", "n_slots", "peak_live", "device_bytes", "n_lds_slots",
                                 "reads_lds", "reads_hbm", "writes_lds", "writes_hbm", "n_fused_free", "and_terms")]


class _PlanSessionOpts(C.Structure):
    _fields_ = [("retain_stream", C.c_int), ("max_concurrent_calls", C.c_uint32), ("window_ct_records", C.c_uint64), ("max_scratch_slots", C.c_uint64),
                ("max_window_calls", C.c_uint32), ("drain_segment_records", C.c_uint32)]


_SCHED_FIELDS = ["n_calls", "n_windows", "n_dependencies", "max_width", "scratch_slots", "wire_file_slots", "window_ct_records", "critical_steps", "total_steps",
                 "n_segments", "segment_ct_records", "ct_ring_records"]


class _PlanScheduleInfo(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in _SCHED_FIELDS]


class _Gate(C.Structure):
    _fields_ = [("wire_a", C.c_uint64), ("wire_b", C.c_uint64), ("wire_c", C.c_uint64), ("gate_type", C.c_uint8), ("pad", C.c_uint8 * 7)]


EXPORTS = [
    "gsv_last_error", "gsv_recorder_create", "gsv_recorder_destroy", "gsv_recorder_allocate_wire", "gsv_recorder_declare_input",
    "gsv_recorder_push_gates", "gsv_recorder_declare_outputs", "gsv_recorder_record_circuit", "gsv_recorder_counts", "gsv_program_compile", "gsv_program_destroy",
    "gsv_program_get_info", "gsv_engine_create", "gsv_engine_destroy", "gsv_deferred_release_count", "gsv_session_fallback_count", "gsv_plan_build_file_pair", "gsv_plan_call_record_form", "gsv_labels_from_seed", "gsv_session_create", "gsv_session_destroy",
    "gsv_session_set_garble_inputs", "gsv_session_garble", "gsv_session_set_evaluate_inputs", "gsv_session_upload_ciphertexts",
    "gsv_session_evaluate", "gsv_session_set_hasher", "gsv_session_sync", "gsv_session_last_kernel_ms", "gsv_session_read_outputs", "gsv_session_read_ciphertexts",
    "gsv_session_ciphertext_hash", "gsv_cbcmac_update", "gsv_cbcmac_update_many", "gsv_commit_labels",
    "gsv_plan_from_circuit", "gsv_plan_io", "gsv_plan_recorder_create", "gsv_plan_recorder_destroy", "gsv_plan_recorder_allocate_wire",
    "gsv_plan_recorder_declare_input", "gsv_plan_recorder_push_gates", "gsv_plan_recorder_call", "gsv_plan_recorder_finish", "gsv_plan_create", "gsv_plan_destroy", "gsv_plan_add_call", "gsv_plan_finish", "gsv_plan_counts", "gsv_session_create_plan", "gsv_session_create_plan_ex",
    "gsv_session_garble_streaming", "gsv_session_garble_streaming_calls", "gsv_plan_call_info", "gsv_plan_image_bytes", "gsv_plan_wire_file", "gsv_plan_save", "gsv_plan_load", "gsv_plan_build_file", "gsv_cbcmac_chains_per_step", "gsv_session_evaluate_streaming", "gsv_session_instances_per_workgroup", "gsv_session_enable_step_clock", "gsv_session_read_step_clock", "gsv_program_step_stats",
    "gsv_session_create_plan_opts", "gsv_session_plan_schedule_info", "gsv_session_plan_window", "gsv_session_set_unchecked_slices",
    "gsv_session_garble_streaming_sink", "gsv_session_garble_evaluate", "gsv_session_evaluate_streaming_indexed", "gsv_session_evaluate_streaming_source",
    "gsv_recorder_allocate_wires", "gsv_plan_recorder_allocate_wires", "gsv_program_compile_opts", "gsv_program_wait", "gsv_plan_recorder_create_opts", "gsv_session_set_drain_instances",
]

# CiphertextHandler / CiphertextSource as host callbacks (include/gsv_engine.h: gsv_ct_sink_fn, gsv_ct_source_fn)
CT_SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64)
CT_SOURCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64)


def lib():
    """Loads (building in-tree if needed) libgsv_engine.so.  Raises if the library cannot be built/loaded."""
    global _lib
    if _lib is None:
        so = _build.build()
        L = C.CDLL(so)
        u8p, vp = C.POINTER(C.c_uint8), C.c_void_p
        L.gsv_last_error.restype = C.c_char_p
        L.gsv_recorder_create.argtypes = [C.POINTER(vp)]
        L.gsv_recorder_destroy.argtypes = [vp]
        L.gsv_recorder_destroy.restype = None
        L.gsv_recorder_allocate_wire.argtypes = [vp, C.c_uint16, C.POINTER(C.c_uint64)]
        L.gsv_recorder_declare_input.argtypes = [vp, C.c_uint64]
        L.gsv_recorder_push_gates.argtypes = [vp, C.POINTER(_Gate), C.c_size_t]
        L.gsv_recorder_declare_outputs.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t]
        L.gsv_recorder_record_circuit.argtypes = [vp, C.c_char_p]
        L.gsv_recorder_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsv_program_compile.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(vp)]
        L.gsv_program_destroy.argtypes = [vp]
        L.gsv_program_destroy.restype = None
        L.gsv_program_get_info.argtypes = [vp, C.POINTER(_ProgramInfo)]
        L.gsv_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.gsv_engine_destroy.argtypes = [vp]
        L.gsv_engine_destroy.restype = None
        L.gsv_session_fallback_count.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.gsv_plan_call_record_form.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32)]
        L.gsv_plan_build_file_pair.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint32]
        L.gsv_deferred_release_count.argtypes = []
        L.gsv_deferred_release_count.restype = C.c_uint64
        L.gsv_labels_from_seed.argtypes = [C.c_uint64, C.c_size_t, u8p, u8p, u8p, u8p]
        L.gsv_session_create.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.c_uint64, C.POINTER(vp)]
        L.gsv_session_destroy.argtypes = [vp]
        L.gsv_session_destroy.restype = None
        L.gsv_session_set_garble_inputs.argtypes = [vp, u8p, u8p, u8p]
        L.gsv_session_garble.argtypes = [vp, C.c_uint64]
        L.gsv_session_set_evaluate_inputs.argtypes = [vp, u8p, u8p, u8p]
        L.gsv_session_upload_ciphertexts.argtypes = [vp, C.c_size_t, u8p, C.c_uint64]
        L.gsv_session_evaluate.argtypes = [vp, C.c_uint64]
        L.gsv_session_set_hasher.argtypes = [vp, C.c_int]
        L.gsv_session_sync.argtypes = [vp]
        L.gsv_session_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_double)]
        L.gsv_session_read_outputs.argtypes = [vp, u8p, u8p]
        L.gsv_session_read_ciphertexts.argtypes = [vp, C.c_size_t, C.c_uint64, C.c_uint64, u8p]
        L.gsv_session_ciphertext_hash.argtypes = [vp, C.c_size_t, u8p]
        L.gsv_cbcmac_update.argtypes = [u8p, u8p, C.c_uint64]
        L.gsv_commit_labels.argtypes = [u8p, C.c_uint64, u8p]
        L.gsv_cbcmac_update_many.argtypes = [u8p, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint64]
        L.gsv_plan_from_circuit.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
        L.gsv_plan_io.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsv_plan_recorder_create.argtypes = [C.POINTER(vp)]
        L.gsv_plan_recorder_destroy.argtypes = [vp]
        L.gsv_plan_recorder_destroy.restype = None
        L.gsv_plan_recorder_allocate_wire.argtypes = [vp, C.c_uint16, C.POINTER(C.c_uint64)]
        L.gsv_plan_recorder_declare_input.argtypes = [vp, C.c_uint64]
        L.gsv_plan_recorder_push_gates.argtypes = [vp, C.POINTER(_Gate), C.c_size_t]
        L.gsv_plan_recorder_call.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsv_plan_recorder_finish.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(vp)]
        L.gsv_plan_create.argtypes = [C.POINTER(vp)]
        L.gsv_plan_destroy.argtypes = [vp]
        L.gsv_plan_destroy.restype = None
        L.gsv_plan_add_call.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.gsv_plan_finish.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.c_size_t]
        L.gsv_plan_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsv_session_create_plan.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
        L.gsv_session_create_plan_ex.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
        L.gsv_session_evaluate_streaming.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_uint64, u8p]
        L.gsv_session_garble_streaming.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int, u8p]
        L.gsv_session_garble_streaming_calls.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int, u8p]
        L.gsv_plan_call_info.argtypes = [vp, C.c_uint64] + [C.POINTER(C.c_uint64)] * 5
        L.gsv_plan_image_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsv_plan_wire_file.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsv_plan_save.argtypes = [vp, C.c_char_p]
        L.gsv_plan_load.argtypes = [C.c_char_p, vp, C.POINTER(vp)]
        L.gsv_plan_build_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.gsv_session_instances_per_workgroup.argtypes = [vp, C.POINTER(C.c_int)]
        L.gsv_session_enable_step_clock.argtypes = [vp]
        L.gsv_session_read_step_clock.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.gsv_program_step_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.gsv_session_create_plan_opts.argtypes = [vp, vp, C.c_size_t, C.POINTER(_PlanSessionOpts), C.POINTER(vp)]
        L.gsv_session_plan_schedule_info.argtypes = [vp, C.POINTER(_PlanScheduleInfo)]
        L.gsv_session_plan_window.argtypes = [vp, C.c_uint64] + [C.POINTER(C.c_uint64)] * 3
        L.gsv_session_set_unchecked_slices.argtypes = [vp, C.c_int]
        L.gsv_session_set_drain_instances.argtypes = [vp, C.c_size_t]
        L.gsv_session_garble_streaming_sink.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, CT_SINK_FN, vp, C.c_int, u8p]
        L.gsv_session_garble_evaluate.argtypes = [vp, vp, C.c_uint64, C.c_int, u8p]
        L.gsv_session_evaluate_streaming_indexed.argtypes = [vp, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), u8p]
        L.gsv_session_evaluate_streaming_source.argtypes = [vp, C.c_uint64, CT_SOURCE_FN, vp, u8p]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise GsvError("gsv status %d: %s" % (rc, lib().gsv_last_error().decode(errors="replace")))


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _u8(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a if shape is None else a.reshape(shape)


def labels_from_seed(seed, n_inputs):
    """(delta, false_label0, true_label0, input_label0[n_inputs,16]) as GarbleMode::new draws them
    (src/circuit/modes/garble_mode.rs:80-97,116-118)."""
    d, f, t = (np.zeros(16, np.uint8) for _ in range(3))
    inp = np.zeros((n_inputs, 16), np.uint8)
    _chk(lib().gsv_labels_from_seed(seed, n_inputs, _p(d), _p(f), _p(t), _p(inp)))
    return d, f, t, inp


def cbcmac(cts, state=None):
    """AESAccumulatingHash over 16-byte records (src/ciphertext_hasher.rs:23-29)."""
    st = np.zeros(16, np.uint8) if state is None else _u8(state).copy()
    a = _u8(cts).reshape(-1)
    _chk(lib().gsv_cbcmac_update(_p(st), _p(a) if a.size else None, a.size // 16))
    return st.tobytes()


def cbcmac_many(streams, states=None):
    """AESAccumulatingHash of several equally long streams at once (gsv_cbcmac_update_many: four chains side by side per step)."""
    arrs = [_u8(a).reshape(-1) for a in streams]
    n = arrs[0].size // 16
    assert all(a.size == n * 16 for a in arrs)
    st = np.zeros((len(arrs), 16), np.uint8) if states is None else _u8(states, (len(arrs), 16)).copy()
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    _chk(lib().gsv_cbcmac_update_many(_p(st), ptrs, len(arrs), n))
    return [bytes(st[i]) for i in range(len(arrs))]


def cbcmac_chains_per_step():
    """How many chains one host thread advances side by side (16 with VAES + AVX-512, 4 with AES-NI)."""
    return int(lib().gsv_cbcmac_chains_per_step())


class Program:
    """A recorded circuit compiled to device steps (gsv_recorder + gsv_program)."""

    def __init__(self, handle):
        self.h = handle
        _live["program"].add(self)
        info = _ProgramInfo()
        _chk(lib().gsv_program_get_info(self.h, C.byref(info)))
        self.info = {n: (list(getattr(info, n)) if n == "gate_count" else int(getattr(info, n))) for n, _ in _ProgramInfo._fields_}

    def step_stats(self):
        """Diagnostics: per step [and_cnt, xor_cnt, lds_reads, hbm_reads, lds_writes, hbm_writes]."""
        out = np.zeros((self.info["n_steps"], 6), np.uint32)
        _chk(lib().gsv_program_step_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    @classmethod
    def from_circuit(cls, spec, chain_feedback=False):
        """Record one of the built-in restated circuits through the two-pass credit driver and compile it.
        chain_feedback: output i is copied to input i after every replay (component chains)."""
        L = lib()
        r = C.c_void_p()
        _chk(L.gsv_recorder_create(C.byref(r)))
        try:
            _chk(L.gsv_recorder_record_circuit(r, spec.encode()))
            return cls._compile(r, chain_feedback)
        finally:
            L.gsv_recorder_destroy(r)

    @classmethod
    def from_gates(cls, n_inputs, gates, outputs, credits=None):
        """Record an explicit gate list through the CircuitMode-shaped recorder API.
        gates: iterable of (gate_type, a, b, c) with wire ids: 0/1 constants, inputs 2..2+n_inputs-1, further
        wires in first-write order; c == None marks a dead gate (UNREACHABLE)."""
        L = lib()
        r = C.c_void_p()
        _chk(L.gsv_recorder_create(C.byref(r)))
        try:
            known = set()
            for _ in range(n_inputs):
                w = C.c_uint64()
                _chk(L.gsv_recorder_allocate_wire(r, 1, C.byref(w)))
                _chk(L.gsv_recorder_declare_input(r, w.value))
                known.add(w.value)
            arr = (_Gate * max(1, len(gates)))()
            for i, (t, a, b, c) in enumerate(gates):
                if c is not None and c not in known and c >= 2:
                    w = C.c_uint64()
                    _chk(L.gsv_recorder_allocate_wire(r, 1, C.byref(w)))
                    if w.value != c:
                        raise GsvError("from_gates: wires must be numbered in first-write order (expected %d, got %d)" % (w.value, c))
                    known.add(c)
                arr[i].wire_a, arr[i].wire_b, arr[i].gate_type = a, b, t
                arr[i].wire_c = 0xFFFFFFFFFFFFFFFF if c is None else c
            _chk(L.gsv_recorder_push_gates(r, arr, len(gates)))
            outs = (C.c_uint64 * max(1, len(outputs)))(*outputs)
            _chk(L.gsv_recorder_declare_outputs(r, outs, len(outputs)))
            return cls._compile(r, False)
        finally:
            L.gsv_recorder_destroy(r)

    @classmethod
    def _compile(cls, r, chain_feedback):
        L = lib()
        h = C.c_void_p()
        if chain_feedback:
            no = C.c_uint64()
            _chk(L.gsv_recorder_counts(r, None, C.byref(no), None))
            n_out = no.value
            idx = (C.c_uint32 * n_out)(*range(n_out))
            _chk(L.gsv_program_compile(r, idx, idx, n_out, C.byref(h)))
        else:
            _chk(L.gsv_program_compile(r, None, None, 0, C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "h", None) is not None and _lib is not None and not sys.is_finalizing():
            _lib.gsv_program_destroy(self.h)
        self.h = None

    __del__ = close


class Engine:
    """One GPU (gsv_engine).  Creation fails loudly when no HIP device is present."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().gsv_engine_create(device, C.byref(self.h)))
        self.device = device
        _live["engine"].add(self)

    def close(self):
        # sessions (and plans loaded straight into this device) use the engine's stream and device: they go first, whoever still holds them
        # (a session kept alive by a reference cycle and collected after the engine was closed ran gsv_session_destroy on a freed engine)
        if getattr(self, "h", None) is not None and self.h:
            for kind in ("session", "plan"):
                for s in list(_live[kind]):
                    if getattr(s, "engine", None) is self:
                        s.close()
        if getattr(self, "h", None) is not None and _lib is not None and self.h and not sys.is_finalizing():
            _lib.gsv_engine_destroy(self.h)
        self.h = None

    __del__ = close


class Plan:
    """A sequence of calls to compiled programs over one wire file (gsv_plan): component-level programs."""

    def __init__(self):
        self.h = C.c_void_p()
        _chk(lib().gsv_plan_create(C.byref(self.h)))
        _live["plan"].add(self)
        self.programs = []  # keep the programs alive
        self.n_inputs = self.n_outputs = 0

    @classmethod
    def from_circuit(cls, spec, units, half_window=False, window_div=None):
        """Record a built-in circuit with the named components (list of names such as "fq12::mul_montgomery") as calls.
        window_div = 2 | 4 (half_window=True is window_div=2): compile every program once, for half / a quarter of the LDS label
        window; the one image then serves every layout of up to that many instances per workgroup and the recorded traces are not
        kept — a third less host memory and one compilation for plans with hundreds of programs."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        self.programs = []
        _live["plan"].add(self)
        div = int(window_div) if window_div else (2 if half_window else 1)
        if div not in (1, 2, 4):
            raise ValueError("window_div must be 1, 2 or 4")
        saved = {k: os.environ.get(k) for k in ("GSV_PLAN_WINDOW_DIV", "GSV_PLAN_HALF_WINDOW")}
        if div > 1:
            os.environ["GSV_PLAN_WINDOW_DIV"] = str(div)
            os.environ.pop("GSV_PLAN_HALF_WINDOW", None)
        try:
            _chk(lib().gsv_plan_from_circuit(spec.encode(), ",".join(units).encode(), C.byref(self.h)))
        finally:
            if div > 1:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        g, c, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_counts(self.h, C.byref(g), C.byref(c), C.byref(k)))
        n_in, n_out = C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_io(self.h, C.byref(n_in), C.byref(n_out)))
        self.n_inputs, self.n_outputs = n_in.value, n_out.value
        self.info = {"n_inputs": n_in.value, "n_outputs": n_out.value, "n_gates": g.value, "n_ciphertexts": c.value, "n_calls": k.value, "n_steps": 0}
        return self

    @staticmethod
    def build_file(spec, units, path, window_div=4):
        """Build the plan of a built-in circuit straight into the plan file `path` (gsv_plan_build_file): every program is written
        by the worker that compiled it and dropped from memory.  Load it with Plan.load(path, engine)."""
        if int(window_div) not in (1, 2, 4):
            raise ValueError("window_div must be 1, 2 or 4 (one image per program; 1 = full LDS window, one instance per workgroup only)")
        saved = {k: os.environ.get(k) for k in ("GSV_PLAN_WINDOW_DIV", "GSV_PLAN_HALF_WINDOW")}
        os.environ["GSV_PLAN_WINDOW_DIV"] = str(int(window_div))
        os.environ.pop("GSV_PLAN_HALF_WINDOW", None)
        try:
            _chk(lib().gsv_plan_build_file(spec.encode(), ",".join(units).encode(), os.fsencode(path)))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    @staticmethod
    def build_file_pair(spec, units_a, path_a, window_div_a, path_b, window_div_b, units_b=None):
        """Two plan files from ONE build (gsv_plan_build_file_pair): the units both plans share are recorded once and compiled for both
        shares of the LDS window (units_b None: the same units).  Each file equals what build_file writes for its units and window_div."""
        _chk(lib().gsv_plan_build_file_pair(spec.encode(), ",".join(units_a).encode(), os.fsencode(path_a), int(window_div_a),
                                            ",".join(units_b).encode() if units_b is not None else None, os.fsencode(path_b), int(window_div_b)))

    def _read_info(self):
        g, c, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_counts(self.h, C.byref(g), C.byref(c), C.byref(k)))
        n_in, n_out = C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_io(self.h, C.byref(n_in), C.byref(n_out)))
        self.n_inputs, self.n_outputs = n_in.value, n_out.value
        self.info = {"n_inputs": n_in.value, "n_outputs": n_out.value, "n_gates": g.value, "n_ciphertexts": c.value, "n_calls": k.value, "n_steps": 0}

    def save(self, path):
        """Write the finished plan (compiled programs + calls) to `path` (gsv_plan_save)."""
        _chk(lib().gsv_plan_save(self.h, os.fsencode(path)))

    @classmethod
    def load(cls, path, engine=None):
        """Read a plan file.  With `engine` the program records go straight into that GPU's memory and the host keeps
        only metadata (the ranks of a node share the file through the page cache); without, a complete host copy."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        self.programs = []
        _live["plan"].add(self)
        self.engine = engine  # a device-resident plan must not outlive its engine
        _chk(lib().gsv_plan_load(os.fsencode(path), engine.h if engine is not None else None, C.byref(self.h)))
        self._read_info()
        return self

    def image_bytes(self):
        """(bytes of compiled program records, number of distinct programs)."""
        b, n = C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_image_bytes(self.h, C.byref(b), C.byref(n)))
        return b.value, n.value

    def wire_file(self):
        """(global wire slots, largest program's own slots): a session's wire file is their sum x 16 bytes per instance."""
        g, m = C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_wire_file(self.h, C.byref(g), C.byref(m)))
        return g.value, m.value

    def call_info(self):
        """Per call: [gate offset, gates, ciphertext offset, ciphertexts, device steps] as a uint64 array [n_calls, 5]."""
        n = self.info["n_calls"]
        out = np.zeros((n, 5), np.uint64)
        v = [C.c_uint64() for _ in range(5)]
        for k in range(n):
            _chk(lib().gsv_plan_call_info(self.h, k, *[C.byref(x) for x in v]))
            out[k] = [x.value for x in v]
        return out

    def call_record_forms(self):
        """Per call: wires per AND input of its program's records (2 or 4); a window with a four-wire program runs the FW kernel."""
        v = C.c_uint32()
        out = []
        for k in range(self.info["n_calls"]):
            _chk(lib().gsv_plan_call_record_form(self.h, k, C.byref(v)))
            out.append(int(v.value))
        return out

    def add_call(self, program, in_globals, out_globals):
        a = np.ascontiguousarray(in_globals, np.uint32)
        b = np.ascontiguousarray(out_globals, np.uint32)
        assert a.size == program.info["n_inputs"] and b.size == program.info["n_outputs"]
        _chk(lib().gsv_plan_add_call(self.h, program.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), b.ctypes.data_as(C.POINTER(C.c_uint32))))
        self.programs.append(program)

    def finish(self, n_inputs, output_globals):
        o = np.ascontiguousarray(output_globals, np.uint32)
        _chk(lib().gsv_plan_finish(self.h, n_inputs, o.ctypes.data_as(C.POINTER(C.c_uint32)), o.size))
        self.n_inputs, self.n_outputs = n_inputs, int(o.size)
        g, c, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_counts(self.h, C.byref(g), C.byref(c), C.byref(k)))
        self.info = {"n_inputs": n_inputs, "n_outputs": int(o.size), "n_gates": g.value, "n_ciphertexts": c.value, "n_calls": k.value, "n_steps": 0}

    def close(self):
        if getattr(self, "h", None) and _lib is not None and not sys.is_finalizing():
            _lib.gsv_plan_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlanRecorder:
    """The plan builder driven gate by gate through the C ABI (gsv_plan_recorder_*): what a host with its own two-pass
    driver uses.  Wires: 0 / 1 constants, further ids handed out by allocate_wire."""

    def __init__(self):
        self.h = C.c_void_p()
        _chk(lib().gsv_plan_recorder_create(C.byref(self.h)))
        self.programs = []
        self.n_inputs = 0

    def allocate_wire(self, credits=1):
        w = C.c_uint64()
        _chk(lib().gsv_plan_recorder_allocate_wire(self.h, credits, C.byref(w)))
        return w.value

    def input_wire(self):
        w = self.allocate_wire(1)
        _chk(lib().gsv_plan_recorder_declare_input(self.h, w))
        self.n_inputs += 1
        return w

    def push_gates(self, gates):
        """gates: list of (gate_type, a, b, c); c == None marks a dead gate."""
        arr = (_Gate * max(1, len(gates)))()
        for i, (t, a, b, c) in enumerate(gates):
            arr[i].wire_a, arr[i].wire_b, arr[i].gate_type = a, b, t
            arr[i].wire_c = 0xFFFFFFFFFFFFFFFF if c is None else c
        _chk(lib().gsv_plan_recorder_push_gates(self.h, arr, len(gates)))

    def call(self, program, in_wires):
        self.programs.append(program)
        a = (C.c_uint64 * max(1, len(in_wires)))(*in_wires)
        n_out = program.info["n_outputs"]
        o = (C.c_uint64 * max(1, n_out))()
        assert len(in_wires) == program.info["n_inputs"]
        _chk(lib().gsv_plan_recorder_call(self.h, program.h, a, o))
        return list(o[:n_out])

    def finish(self, output_wires):
        plan = Plan.__new__(Plan)
        plan.h = C.c_void_p()
        _live["plan"].add(plan)
        plan.programs = list(self.programs)
        o = (C.c_uint64 * max(1, len(output_wires)))(*output_wires)
        _chk(lib().gsv_plan_recorder_finish(self.h, o, len(output_wires), C.byref(plan.h)))
        g, c, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().gsv_plan_counts(plan.h, C.byref(g), C.byref(c), C.byref(k)))
        plan.n_inputs, plan.n_outputs = self.n_inputs, len(output_wires)
        plan.info = {"n_inputs": self.n_inputs, "n_outputs": len(output_wires), "n_gates": g.value, "n_ciphertexts": c.value, "n_calls": k.value, "n_steps": 0}
        return plan

    def close(self):
        if self.h:
            lib().gsv_plan_recorder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """A batch of instances on one program (gsv_session) or, with a Plan, on a sequence of component programs."""

    def __init__(self, engine, program, n_instances=1, replays=1, ct_capacity_replays=None, retain_stream=True, concurrent_calls=0, window_ct_records=0,
                 max_scratch_slots=0, max_window_calls=0, drain_segment_records=0):
        """Plan sessions: retain_stream = True (the whole stream stays on the device), False (one window of it) or "ring" (GSV_STREAM_RING:
        the whole pass as one launch over a ciphertext ring); concurrent_calls = how many independent calls of the plan may run side by
        side (0: as many as give every CU a workgroup, 1: sequential); window_ct_records / max_scratch_slots / max_window_calls: see gsv_plan_session_opts (0 = automatic)."""
        self.engine, self.program = engine, program
        self.n, self.replays = n_instances, replays
        self.ct_cap = replays if ct_capacity_replays is None else ct_capacity_replays
        self.h = C.c_void_p()
        if isinstance(program, Plan):
            assert replays == 1
            if retain_stream == "ring" or (retain_stream is not True and retain_stream is not False and retain_stream == 2):
                rs = 2  # GSV_STREAM_RING
            elif retain_stream in (True, False, 0, 1):
                rs = int(bool(retain_stream))
            else:
                raise ValueError("retain_stream must be True, False, 0, 1, 2 or \"ring\"")
            o = _PlanSessionOpts(rs, int(concurrent_calls), int(window_ct_records), int(max_scratch_slots), int(max_window_calls), int(drain_segment_records))
            _chk(lib().gsv_session_create_plan_opts(engine.h, program.h, n_instances, C.byref(o), C.byref(self.h)))
        else:
            _chk(lib().gsv_session_create(engine.h, program.h, n_instances, replays, self.ct_cap, C.byref(self.h)))
        self.n_in, self.n_out = program.info["n_inputs"], program.info["n_outputs"]
        _live["session"].add(self)

    def set_garble_inputs(self, delta, const_label0, input_label0):
        d = _u8(delta, (self.n, 16))
        c = _u8(const_label0, (self.n, 32))
        i = _u8(input_label0, (self.n, self.n_in * 16))
        _chk(lib().gsv_session_set_garble_inputs(self.h, _p(d), _p(c), _p(i) if self.n_in else None))

    def garble(self, gate_id_base=0):
        _chk(lib().gsv_session_garble(self.h, gate_id_base))

    def garble_streaming(self, gate_id_base=0, directory=None, first_index=0, threads=0, discard=False):
        """Garble all replays while the host drains the stream segment by segment: returns the per-instance ciphertext
        hashes (CBC-MAC, gate order) and, with `directory`, writes gc_<first_index+i>.bin files.  discard=True: garble
        only, the ciphertexts are dropped."""
        if discard:
            with _gc_paused():
                _chk(lib().gsv_session_garble_streaming(self.h, gate_id_base, None, 0, 0, None))
            return None
        out = np.zeros((self.n, 16), np.uint8)
        with _gc_paused():
            _chk(lib().gsv_session_garble_streaming(self.h, gate_id_base, directory.encode() if directory else None, first_index, threads, _p(out)))
        return [bytes(out[i]) for i in range(self.n)]

    def garble_calls(self, first_call, n_calls, gate_id_base=0, directory=None, first_index=0, threads=0, discard=False):
        """Plan sessions: garble calls [first_call, first_call + n_calls) only (a slice of the plan).  Wires, gate ids and the
        CBC-MAC states continue from the previous slice; first_call == 0 starts a new pass.  Returns the MAC states after the
        slice (the commitments once the last slice has run), or None with discard=True."""
        if discard:
            with _gc_paused():
                _chk(lib().gsv_session_garble_streaming_calls(self.h, gate_id_base, first_call, n_calls, None, 0, 0, None))
            return None
        out = np.zeros((self.n, 16), np.uint8)
        with _gc_paused():
            _chk(lib().gsv_session_garble_streaming_calls(self.h, gate_id_base, first_call, n_calls, directory.encode() if directory else None, first_index, threads, _p(out)))
        return [bytes(out[i]) for i in range(self.n)]

    def garble_to_sink(self, handler, gate_id_base=0, first_call=0, n_calls=0, threads=0, with_hashes=False):
        """Garble with every ciphertext handed to `handler(instance, first_record, records)` — the generic CiphertextHandler
        (circuit/mod.rs:140-178): `records` is an [n,16] uint8 array in gate order, valid during the call only; the runs of one
        instance arrive in stream order.  Exceptions raised by the handler abort the pass.  Returns the CBC-MACs with
        with_hashes=True."""
        failure = []

        def _cb(_user, inst, first, ptr, n):
            try:
                handler(int(inst), int(first), np.ctypeslib.as_array(ptr, shape=(int(n), 16)))
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through the C frames
                failure.append(e)
                return 1

        cb = CT_SINK_FN(_cb)
        out = np.zeros((self.n, 16), np.uint8) if with_hashes else None
        with _gc_paused():
            rc = lib().gsv_session_garble_streaming_sink(self.h, gate_id_base, first_call, n_calls, cb, None, threads, _p(out))
        if failure:
            raise failure[0]
        _chk(rc)
        return [bytes(out[i]) for i in range(self.n)] if with_hashes else None

    def garble_evaluate(self, evaluator, gate_id_base=0, threads=0, with_hashes=False):
        """Garble this (plan, retain_stream=False) session while `evaluator` — a session of the same plan and options with its inputs
        set — evaluates every window straight from this session's device block (gsv_session_garble_evaluate)."""
        out = np.zeros((self.n, 16), np.uint8) if with_hashes else None
        with _gc_paused():
            _chk(lib().gsv_session_garble_evaluate(self.h, evaluator.h, gate_id_base, threads, _p(out)))
        return [bytes(out[i]) for i in range(self.n)] if with_hashes else None

    def evaluate_streaming_indexed(self, directory, indexes, gate_id_base=0):
        """Evaluate with instance i reading gc_<indexes[i]>.bin: the finalized instances of a cut-and-choose run in one session."""
        idx = np.ascontiguousarray(indexes, np.uint64)
        assert idx.size == self.n
        out = np.zeros((self.n, 16), np.uint8)
        with _gc_paused():
            _chk(lib().gsv_session_evaluate_streaming_indexed(self.h, gate_id_base, directory.encode(), idx.ctypes.data_as(C.POINTER(C.c_uint64)), _p(out)))
        return [bytes(out[i]) for i in range(self.n)]

    def evaluate_from_source(self, source, gate_id_base=0):
        """Evaluate with the ciphertexts pulled from `source(instance, first_record, n) -> [n,16] uint8` (None / short = exhausted):
        the generic CiphertextSource (ciphertext_source.rs:14-34).  Returns the CBC-MACs of what was read."""
        failure = []

        def _cb(_user, inst, first, ptr, n):
            try:
                a = source(int(inst), int(first), int(n))
                if a is None:
                    return 1
                a = _u8(a).reshape(-1)
                if a.size != int(n) * 16:
                    return 1
                C.memmove(ptr, a.ctypes.data, a.size)
                return 0
            except BaseException as e:  # noqa: BLE001
                failure.append(e)
                return 2

        cb = CT_SOURCE_FN(_cb)
        out = np.zeros((self.n, 16), np.uint8)
        with _gc_paused():
            rc = lib().gsv_session_evaluate_streaming_source(self.h, gate_id_base, cb, None, _p(out))
        if failure:
            raise failure[0]
        _chk(rc)
        return [bytes(out[i]) for i in range(self.n)]

    def schedule_info(self):
        """Plan sessions: the call-level schedule this session executes (windows, batches of calls side by side, wire-file layout,
        depth in device steps)."""
        i = _PlanScheduleInfo()
        _chk(lib().gsv_session_plan_schedule_info(self.h, C.byref(i)))
        return {k: int(getattr(i, k)) for k in _SCHED_FIELDS}

    def fallback_count(self):
        """How often this plan session has fallen back to the safe schedule (one call per launch) after a dependency wait gave up."""
        n = C.c_uint64()
        _chk(lib().gsv_session_fallback_count(self.h, C.byref(n)))
        return int(n.value)

    def windows(self):
        """[(first_call, n_calls, max_width)] of the schedule: slices handed to garble_calls start and end on these boundaries."""
        out = []
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        for w in range(self.schedule_info()["n_windows"]):
            _chk(lib().gsv_session_plan_window(self.h, w, C.byref(a), C.byref(b), C.byref(c)))
            out.append((a.value, b.value, c.value))
        return out

    def set_drain_instances(self, n):
        """The streaming calls garble every instance but only the first `n` instances' ciphertext streams leave the device (0 = all):
        their hashes / gc files / sink runs are what the calls then return (a checked SAMPLE of a full-GPU batch)."""
        _chk(lib().gsv_session_set_drain_instances(self.h, int(n)))
        self.n_drain = int(n) or self.n

    def set_unchecked_slices(self, on=True):
        """Timing harnesses only: allow garble_calls slices that do not continue the previous one (stale wires, meaningless MACs)."""
        _chk(lib().gsv_session_set_unchecked_slices(self.h, int(bool(on))))

    def evaluate_streaming(self, directory, first_index=0, gate_id_base=0):
        """Evaluate with the ciphertexts read from gc_<first_index+i>.bin segment by segment; returns the files' CBC-MACs."""
        out = np.zeros((self.n, 16), np.uint8)
        with _gc_paused():
            _chk(lib().gsv_session_evaluate_streaming(self.h, gate_id_base, directory.encode(), first_index, _p(out)))
        return [bytes(out[i]) for i in range(self.n)]

    def set_evaluate_inputs(self, const_active, input_active, input_bits):
        c = _u8(const_active, (self.n, 32))
        a = _u8(input_active, (self.n, self.n_in * 16))
        b = _u8(input_bits, (self.n, self.n_in))
        _chk(lib().gsv_session_set_evaluate_inputs(self.h, _p(c), _p(a) if self.n_in else None, _p(b) if self.n_in else None))

    def upload_ciphertexts(self, instance, cts):
        a = _u8(cts).reshape(-1)
        _chk(lib().gsv_session_upload_ciphertexts(self.h, instance, _p(a) if a.size else None, a.size // 16))

    def evaluate(self, gate_id_base=0):
        _chk(lib().gsv_session_evaluate(self.h, gate_id_base))

    def set_hasher(self, kind):
        """'aes' (AesNiHasher, default) or 'blake3' (Blake3Hasher)."""
        _chk(lib().gsv_session_set_hasher(self.h, {"aes": 0, "blake3": 1}[kind]))

    def sync(self):
        _chk(lib().gsv_session_sync(self.h))

    @property
    def instances_per_workgroup(self):
        n = C.c_int()
        _chk(lib().gsv_session_instances_per_workgroup(self.h, C.byref(n)))
        return n.value

    def enable_step_clock(self):
        """Diagnostics: instance 0 stamps a 100 MHz clock at every step of a launch's last replay."""
        _chk(lib().gsv_session_enable_step_clock(self.h))

    def read_step_clock(self):
        out = np.zeros(self.program.info["n_steps"] + 1, np.uint64)
        _chk(lib().gsv_session_read_step_clock(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def last_kernel_ms(self):
        ms = C.c_double()
        _chk(lib().gsv_session_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def read_outputs(self, with_bits=False):
        lab = np.zeros((self.n, self.n_out, 16), np.uint8)
        bits = np.zeros((self.n, self.n_out), np.uint8) if with_bits else None
        _chk(lib().gsv_session_read_outputs(self.h, _p(lab), _p(bits)))
        return (lab, bits) if with_bits else lab

    def read_ciphertexts(self, instance, first=0, n=None):
        if n is None:
            n = self.ct_cap * self.program.info["n_ciphertexts"] - first
        out = np.zeros((n, 16), np.uint8)
        _chk(lib().gsv_session_read_ciphertexts(self.h, instance, first, n, _p(out) if n else None))
        return out

    def ciphertext_hash(self, instance):
        h = np.zeros(16, np.uint8)
        _chk(lib().gsv_session_ciphertext_hash(self.h, instance, _p(h)))
        return h.tobytes()

    def close(self):
        if getattr(self, "h", None) is not None and _lib is not None and self.h and not sys.is_finalizing():
            _lib.gsv_session_destroy(self.h)
        self.h = None

    __del__ = close


class StreamingResult:
    """Fields of the reference's StreamingResult that exist for this path (src/circuit/mod.rs:81-107)."""
    pass


class CircuitBuilder:
    """Host-side mirror of the reference's CircuitBuilder entry points for the garble/evaluate path."""

    @staticmethod
    def streaming_garbling(circuit, seeds, engine=None, program=None, replays=1, keep_ciphertexts=True, hasher="aes"):
        """Garble `circuit` once per seed (one instance per seed, all on `engine`'s GPU).
        Mirrors CircuitBuilder::streaming_garbling(inputs, cap, seed, AESAccumulatingHash, f) per instance
        (src/circuit/mod.rs:185-203): labels from the seed's ChaCha stream, ciphertext hash = CBC-MAC."""
        seeds = [seeds] if np.isscalar(seeds) else list(seeds)
        engine = engine or Engine(0)
        program = program or Program.from_circuit(circuit)
        n_in = program.info["n_inputs"]
        B = len(seeds)
        delta = np.zeros((B, 16), np.uint8)
        consts = np.zeros((B, 2, 16), np.uint8)
        inputs = np.zeros((B, n_in, 16), np.uint8)
        for i, s in enumerate(seeds):
            delta[i], consts[i, 0], consts[i, 1], inputs[i] = labels_from_seed(s, n_in)
        sess = Session(engine, program, B, replays)
        sess.set_hasher(hasher)
        sess.set_garble_inputs(delta, consts, inputs)
        sess.garble(0)
        sess.sync()
        r = StreamingResult()
        r.session, r.program = sess, program
        r.delta, r.false_label0, r.true_label0, r.input_label0 = delta, consts[:, 0], consts[:, 1], inputs
        r.output_label0 = sess.read_outputs()
        r.kernel_ms = sess.last_kernel_ms()
        r.n_ciphertexts = program.info["n_ciphertexts"] * replays
        r.gate_count = [g * replays for g in program.info["gate_count"]]
        r.ciphertext_hash = [sess.ciphertext_hash(i) for i in range(B)]
        r.ciphertexts = [sess.read_ciphertexts(i) for i in range(B)] if keep_ciphertexts else None
        return r

    @staticmethod
    def streaming_evaluation(circuit, true_active, false_active, input_active, input_bits, ciphertexts, engine=None, program=None, replays=1,
                             hasher="aes"):
        """Evaluate instances from their ciphertext streams (src/circuit/mod.rs:225-249).  Arrays carry a
        leading instance dimension; `ciphertexts` is a list of [n,16] uint8 arrays (gc_{i}.bin bytes)."""
        engine = engine or Engine(0)
        program = program or Program.from_circuit(circuit)
        ta, fa = _u8(true_active).reshape(-1, 16), _u8(false_active).reshape(-1, 16)
        B = ta.shape[0]
        consts = np.stack([fa, ta], axis=1)
        sess = Session(engine, program, B, replays)
        sess.set_hasher(hasher)
        sess.set_evaluate_inputs(consts, input_active, input_bits)
        for i in range(B):
            sess.upload_ciphertexts(i, ciphertexts[i])
        sess.evaluate(0)
        sess.sync()
        r = StreamingResult()
        r.session, r.program = sess, program
        r.output_active, r.output_bits = sess.read_outputs(with_bits=True)
        r.kernel_ms = sess.last_kernel_ms()
        r.ciphertext_hash = [sess.ciphertext_hash(i) for i in range(B)]
        return r


# ---- ciphertext files: the reference's gc_{index}.bin format ---------------------------------------------------
# src/cut_and_choose/ciphertext_repository.rs:29,94-106,157 writes, and FileSource (src/circuit/ciphertext_source.rs:36-107)
# reads, a bare concatenation of 16-byte S::to_bytes() records in gate order, no header.  FileSource CBC-MACs what it
# reads, so evaluation re-derives the commitment; read_gc_file returns the same pair.
def gc_file_name(index):
    return "gc_%d.bin" % index


def write_gc_file(path, ciphertexts):
    """Writes a ciphertext stream ([n,16] uint8) as gc_{i}.bin bytes and returns its AESAccumulatingHash."""
    a = _u8(ciphertexts).reshape(-1, 16)
    with open(path, "wb") as f:
        f.write(a.tobytes())
    return cbcmac(a)


def read_gc_file(path):
    """Returns ([n,16] uint8 ciphertexts, AESAccumulatingHash) of a gc_{i}.bin file; rejects a trailing partial record."""
    data = np.fromfile(path, dtype=np.uint8)
    if data.size % 16:
        raise GsvError("%s: length %d is not a multiple of 16-byte ciphertext records" % (path, data.size))
    cts = data.reshape(-1, 16)
    return cts, cbcmac(cts)
