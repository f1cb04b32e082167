"""ctypes binding for the CPU oracle (oracle/libgsv_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_ROOT, "oracle", "libgsv_oracle.so")

GATE_NAMES = ["And", "Nand", "Nimp", "Imp", "Ncimp", "Cimp", "Nor", "Or", "Xor", "Xnor", "Not"]


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.gsvo_last_error.restype = C.c_char_p
        u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
        L.gsvo_circuit_info.argtypes = [C.c_char_p, u64p, u64p]
        L.gsvo_aes128_encrypt.argtypes = [u8p, u8p, u8p, C.c_int]
        L.gsvo_tweak.argtypes = [C.c_uint64, u8p]
        L.gsvo_hash.argtypes = [u8p, C.c_uint64, u8p]
        L.gsvo_garble_gate.argtypes = [C.c_uint8, u8p, u8p, u8p, C.c_uint64, u8p, u8p]
        L.gsvo_degarble_gate.argtypes = [C.c_uint8, u8p, u8p, C.c_int, u8p, C.c_uint64, u8p]
        L.gsvo_cbcmac.argtypes = [u8p, C.c_uint64, u8p]
        L.gsvo_chacha_labels.argtypes = [C.c_uint64, C.c_uint64, u8p]
        L.gsvo_chacha_words_from_key.argtypes = [u8p, C.c_uint64, C.POINTER(C.c_uint32)]
        L.gsvo_execute.argtypes = [C.c_char_p, C.c_uint64, u8p, u8p, u64p, u64p]
        L.gsvo_garble.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, u8p, u8p, u8p, u8p, u8p, u8p, u64p, u64p, u8p, C.c_uint64, u64p]
        L.gsvo_evaluate.argtypes = [C.c_char_p, C.c_uint64, u8p, u8p, u8p, u8p, u8p, C.c_uint64, u8p, u8p, u8p, u64p]
        L.gsvo_bench_garble.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), u64p, u8p]
        L.gsvo_set_use_aesni.argtypes = [C.c_int]
        L.gsvo_set_hasher.argtypes = [C.c_int]
        L.gsvo_blake3_short.argtypes = [u8p, C.c_uint64, u8p]
        L.gsvo_blake3_hash_with_gate.argtypes = [u8p, C.c_uint64, u8p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().gsvo_last_error().decode())


def _b16(x):
    a = np.frombuffer(bytes(x), dtype=np.uint8).copy()
    assert a.size == 16
    return a


def set_hasher(kind):
    """'aes' = AesNiHasher (default), 'blake3' = Blake3Hasher (src/hashers/mod.rs:22-51)."""
    lib().gsvo_set_hasher({"aes": 0, "blake3": 1}[kind])


def blake3_short(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    out = np.zeros(32, np.uint8)
    _check(lib().gsvo_blake3_short(_p(a) if a.size else None, a.size, _p(out)))
    return out.tobytes()


def blake3_hash_with_gate(label, gate_id):
    out = np.zeros(16, np.uint8)
    lib().gsvo_blake3_hash_with_gate(_p(_b16(label)), gate_id, _p(out))
    return out.tobytes()


def circuit_info(circuit):
    n_in, n_out = C.c_uint64(), C.c_uint64()
    _check(lib().gsvo_circuit_info(circuit.encode(), C.byref(n_in), C.byref(n_out)))
    return n_in.value, n_out.value


def aes128_encrypt(key, block, portable=False):
    out = np.zeros(16, np.uint8)
    lib().gsvo_aes128_encrypt(_p(_b16(key)), _p(_b16(block)), _p(out), int(portable))
    return out.tobytes()


def tweak(gate_id):
    out = np.zeros(16, np.uint8)
    lib().gsvo_tweak(gate_id, _p(out))
    return out.tobytes()


def hash_with_gate(label, gate_id):
    out = np.zeros(16, np.uint8)
    lib().gsvo_hash(_p(_b16(label)), gate_id, _p(out))
    return out.tobytes()


def garble_gate(gate_type, a0, b0, delta, gate_id):
    c0, ct = np.zeros(16, np.uint8), np.zeros(16, np.uint8)
    has = lib().gsvo_garble_gate(gate_type, _p(_b16(a0)), _p(_b16(b0)), _p(_b16(delta)), gate_id, _p(c0), _p(ct))
    return c0.tobytes(), (ct.tobytes() if has else None)


def degarble_gate(gate_type, ct, a, a_value, b, gate_id):
    out = np.zeros(16, np.uint8)
    lib().gsvo_degarble_gate(gate_type, _p(_b16(ct if ct is not None else bytes(16))), _p(_b16(a)), int(a_value), _p(_b16(b)), gate_id, _p(out))
    return out.tobytes()


def cbcmac(cts):
    a = np.ascontiguousarray(np.frombuffer(bytes(cts), dtype=np.uint8)) if not isinstance(cts, np.ndarray) else np.ascontiguousarray(cts.reshape(-1))
    assert a.size % 16 == 0
    out = np.zeros(16, np.uint8)
    lib().gsvo_cbcmac(_p(a) if a.size else None, a.size // 16, _p(out))
    return out.tobytes()


def chacha_labels(seed, n):
    out = np.zeros((n, 16), np.uint8)
    lib().gsvo_chacha_labels(seed, n, _p(out))
    return out


def chacha_words_from_key(key32, n):
    k = np.frombuffer(bytes(key32), dtype=np.uint8).copy()
    out = np.zeros(n, np.uint32)
    lib().gsvo_chacha_words_from_key(_p(k), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def execute(circuit, input_bits, capacity=200_000):
    n_in, n_out = circuit_info(circuit)
    ib = np.ascontiguousarray(np.asarray(input_bits, dtype=np.uint8))
    assert ib.size == n_in
    ob = np.zeros(n_out, np.uint8)
    gc = np.zeros(11, np.uint64)
    peak = C.c_uint64()
    _check(lib().gsvo_execute(circuit.encode(), capacity, _p(ib), _p(ob), _p64(gc), C.byref(peak)))
    return ob, gc, peak.value


class GarbleResult:
    pass


def garble(circuit, seed, capacity=200_000, capture_ct=True, ct_cap=None):
    n_in, n_out = circuit_info(circuit)
    r = GarbleResult()
    r.delta, r.false_label0, r.true_label0 = (np.zeros(16, np.uint8) for _ in range(3))
    r.input_label0 = np.zeros((n_in, 16), np.uint8)
    r.output_label0 = np.zeros((n_out, 16), np.uint8)
    r.ct_hash = np.zeros(16, np.uint8)
    r.gate_counts = np.zeros(11, np.uint64)
    n_ct, peak = C.c_uint64(), C.c_uint64()
    if capture_ct and ct_cap is None:
        # first pass without capture to learn the count would double the work; over-allocate instead.
        g = np.zeros(11, np.uint64)
        ob, g, _ = execute(circuit, np.zeros(n_in, np.uint8), capacity)
        ct_cap = int(g[:8].sum())
    cts = np.zeros((ct_cap if capture_ct else 0, 16), np.uint8)
    _check(lib().gsvo_garble(circuit.encode(), capacity, seed, _p(r.delta), _p(r.false_label0), _p(r.true_label0), _p(r.input_label0),
                             _p(r.output_label0), _p(r.ct_hash), C.byref(n_ct), _p64(r.gate_counts),
                             _p(cts) if capture_ct and ct_cap else None, ct_cap if capture_ct else 0, C.byref(peak)))
    r.n_ciphertexts = n_ct.value
    r.ciphertexts = cts[: r.n_ciphertexts] if capture_ct else None
    r.peak_live = peak.value
    r.n_in, r.n_out = n_in, n_out
    return r


class EvalResult:
    pass


def evaluate(circuit, true_active, false_active, input_active, input_bits, ciphertexts, capacity=200_000):
    n_in, n_out = circuit_info(circuit)
    ia = np.ascontiguousarray(input_active, dtype=np.uint8).reshape(n_in, 16)
    ib = np.ascontiguousarray(np.asarray(input_bits, dtype=np.uint8))
    cts = np.ascontiguousarray(ciphertexts, dtype=np.uint8).reshape(-1, 16)
    r = EvalResult()
    r.output_active = np.zeros((n_out, 16), np.uint8)
    r.output_bits = np.zeros(n_out, np.uint8)
    r.ct_hash = np.zeros(16, np.uint8)
    nc = C.c_uint64()
    _check(lib().gsvo_evaluate(circuit.encode(), capacity, _p(_b16(true_active)), _p(_b16(false_active)), _p(ia), _p(ib),
                               _p(cts) if cts.size else None, cts.shape[0], _p(r.output_active), _p(r.output_bits), _p(r.ct_hash), C.byref(nc)))
    r.n_consumed = nc.value
    return r


def bench_garble(circuit, seed=0, capacity=200_000):
    sec, gates = C.c_double(), C.c_uint64()
    h = np.zeros(16, np.uint8)
    _check(lib().gsvo_bench_garble(circuit.encode(), capacity, seed, C.byref(sec), C.byref(gates), _p(h)))
    return sec.value, gates.value, h.tobytes()


def bench_garble_prefix(circuit, max_gates, seed=0, capacity=200_000):
    """Time the first `max_gates` gates of a (long) circuit's garbling on one core: (seconds, gates, MAC state after the prefix)."""
    sec, gates = C.c_double(), C.c_uint64()
    h = np.zeros(16, np.uint8)
    _check(lib().gsvo_bench_garble_prefix(circuit.encode(), capacity, seed, C.c_uint64(int(max_gates)), C.byref(sec), C.byref(gates), _p(h)))
    return sec.value, gates.value, h.tobytes()


# ---- helpers shared by the tests -------------------------------------------------------------
FQ_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
FQ_R = 1 << 254


def int_to_bits(v, n):
    return np.array([(v >> i) & 1 for i in range(n)], dtype=np.uint8)


def bits_to_int(bits):
    return sum(int(b) << i for i, b in enumerate(bits))
