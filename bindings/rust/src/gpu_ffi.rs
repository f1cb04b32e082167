//! `extern "C"` declarations of include/gsv_engine.h (libgsv_engine.so).  Plain pointers and sizes; every function returns 0 on
//! success, a non-zero `gsv_status` otherwise (the C side never unwinds).  `chk` turns non-zero into `panic!`, the reference's own
//! error behaviour on this path (garble_mode.rs:170-177,221; evaluate_mode.rs:141).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct GsvGate {
    pub wire_a: u64,
    pub wire_b: u64,
    pub wire_c: u64, // u64::MAX = WireId::UNREACHABLE (src/core/wire.rs:8): dead gate, gate id still consumed
    pub gate_type: u8, // GateType repr(C) discriminant (src/core/gate_type.rs:3-15)
    pub pad: [u8; 7],
}
macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0] } )* } }
opaque!(GsvRecorder, GsvProgram, GsvEngine, GsvSession, GsvPlan, GsvPlanRecorder);

#[repr(C)]
pub struct GsvPlanSessionOpts {
    pub retain_stream: c_int, // 0 = windows, 1 = the whole stream stays on the device, 2 = GSV_STREAM_RING (one launch over a ciphertext ring)
    pub max_concurrent_calls: u32,
    pub window_ct_records: u64,
    pub max_scratch_slots: u64,
    pub max_window_calls: u32,
    pub drain_segment_records: u32,
}
// By hand, not derived: the C API's NULL-options default is retain_stream = 1 (a derived Default would say 0 and hand a caller of
// `..Default::default()` a streaming-only session, on which gsv_session_garble / _evaluate fail with GSV_ERR_INVALID).
impl Default for GsvPlanSessionOpts {
    fn default() -> Self {
        GsvPlanSessionOpts { retain_stream: 1, max_concurrent_calls: 0, window_ct_records: 0, max_scratch_slots: 0, max_window_calls: 0, drain_segment_records: 0 }
    }
}

/// gsv_plan_schedule_info (include/gsv_engine.h): what a plan session's schedule looks like (windows, drain segments, the ciphertext ring).
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct GsvPlanScheduleInfo {
    pub n_calls: u64,
    pub n_windows: u64,
    pub n_dependencies: u64,
    pub max_width: u64,
    pub scratch_slots: u64,
    pub wire_file_slots: u64,
    pub window_ct_records: u64,
    pub critical_steps: u64,
    pub total_steps: u64,
    pub n_segments: u64,
    pub segment_ct_records: u64,
    pub ct_ring_records: u64,
}

/// gsv_compile_opts (include/gsv_engine.h): one image for a share of the LDS window, background compilation, records into a plan file.
#[repr(C)]
pub struct GsvCompileOpts {
    pub struct_size: u32, // size_of::<GsvCompileOpts>()
    pub window_div: u32,
    pub keep_trace: u32,
    pub background: u32,
    pub consume_recorder: u32,
    pub reserved: u32,
    pub for_plan: *mut GsvPlanRecorder,
}
#[repr(C)]
pub struct GsvPlanRecorderOpts {
    pub struct_size: u32, // size_of::<GsvPlanRecorderOpts>()
    pub window_div: u32,
    pub plan_file: *const c_char,
}

/// CiphertextHandler::handle over a run of records of one instance (include/gsv_engine.h, gsv_ct_sink_fn): non-zero aborts the pass.
pub type GsvCtSinkFn = unsafe extern "C" fn(user: *mut std::ffi::c_void, instance: usize, first_record: u64, records: *const u8, n_records: u64) -> c_int;
/// CiphertextSource::recv for a run of records (gsv_ct_source_fn): non-zero = the source has run dry.
pub type GsvCtSourceFn = unsafe extern "C" fn(user: *mut std::ffi::c_void, instance: usize, first_record: u64, records: *mut u8, n_records: u64) -> c_int;

pub const GSV_STREAM_RING: c_int = 2; // GsvPlanSessionOpts::retain_stream: the whole pass as one launch over a ciphertext ring (gsv_engine.h)
pub const GSV_HASHER_AES: c_int = 0; // AesNiHasher   (src/hashers/mod.rs:54-96)
pub const GSV_HASHER_BLAKE3: c_int = 1; // Blake3Hasher  (src/hashers/mod.rs:22-51)

extern "C" {
    pub fn gsv_last_error() -> *const c_char;
    // recording: the CircuitMode seam
    pub fn gsv_recorder_create(out: *mut *mut GsvRecorder) -> c_int;
    pub fn gsv_recorder_destroy(r: *mut GsvRecorder);
    pub fn gsv_recorder_allocate_wire(r: *mut GsvRecorder, credits: u16, wire_out: *mut u64) -> c_int;
    pub fn gsv_recorder_allocate_wires(r: *mut GsvRecorder, n: usize, first_wire_out: *mut u64) -> c_int;
    pub fn gsv_recorder_declare_input(r: *mut GsvRecorder, wire: u64) -> c_int;
    pub fn gsv_recorder_push_gates(r: *mut GsvRecorder, gates: *const GsvGate, n: usize) -> c_int;
    pub fn gsv_recorder_declare_outputs(r: *mut GsvRecorder, wires: *const u64, n: usize) -> c_int;
    pub fn gsv_program_compile(r: *mut GsvRecorder, fb_out: *const u32, fb_in: *const u32, n_fb: usize, out: *mut *mut GsvProgram) -> c_int;
    pub fn gsv_program_compile_opts(r: *mut GsvRecorder, opts: *const GsvCompileOpts, out: *mut *mut GsvProgram) -> c_int;
    pub fn gsv_program_wait(p: *mut GsvProgram) -> c_int;
    pub fn gsv_program_destroy(p: *mut GsvProgram);
    // plans: component-level programs (the verifier)
    pub fn gsv_plan_recorder_create(out: *mut *mut GsvPlanRecorder) -> c_int;
    pub fn gsv_plan_recorder_create_opts(opts: *const GsvPlanRecorderOpts, out: *mut *mut GsvPlanRecorder) -> c_int;
    pub fn gsv_plan_recorder_destroy(r: *mut GsvPlanRecorder);
    pub fn gsv_plan_recorder_allocate_wires(r: *mut GsvPlanRecorder, n: usize, first_wire_out: *mut u64) -> c_int;
    pub fn gsv_plan_recorder_allocate_wire(r: *mut GsvPlanRecorder, credits: u16, wire_out: *mut u64) -> c_int;
    pub fn gsv_plan_recorder_declare_input(r: *mut GsvPlanRecorder, wire: u64) -> c_int;
    pub fn gsv_plan_recorder_push_gates(r: *mut GsvPlanRecorder, gates: *const GsvGate, n: usize) -> c_int;
    pub fn gsv_plan_recorder_call(r: *mut GsvPlanRecorder, program: *const GsvProgram, in_wires: *const u64, out_wires: *mut u64) -> c_int;
    pub fn gsv_plan_recorder_finish(r: *mut GsvPlanRecorder, output_wires: *const u64, n_outputs: usize, out: *mut *mut GsvPlan) -> c_int;
    pub fn gsv_plan_destroy(p: *mut GsvPlan);
    pub fn gsv_plan_load(path: *const c_char, e: *mut GsvEngine, out: *mut *mut GsvPlan) -> c_int;
    pub fn gsv_plan_io(p: *const GsvPlan, n_inputs: *mut u64, n_outputs: *mut u64) -> c_int;
    // engine + sessions
    pub fn gsv_engine_create(device: c_int, out: *mut *mut GsvEngine) -> c_int;
    pub fn gsv_engine_destroy(e: *mut GsvEngine);
    pub fn gsv_session_create(e: *mut GsvEngine, p: *const GsvProgram, n_instances: usize, replays: u64, ct_cap: u64, out: *mut *mut GsvSession) -> c_int;
    pub fn gsv_session_create_plan_opts(e: *mut GsvEngine, plan: *const GsvPlan, n_instances: usize, opts: *const GsvPlanSessionOpts, out: *mut *mut GsvSession) -> c_int;
    pub fn gsv_session_destroy(s: *mut GsvSession);
    pub fn gsv_session_set_hasher(s: *mut GsvSession, kind: c_int) -> c_int;
    pub fn gsv_session_set_garble_inputs(s: *mut GsvSession, delta: *const u8, const_label0: *const u8, input_label0: *const u8) -> c_int;
    pub fn gsv_session_garble(s: *mut GsvSession, gate_id_base: u64) -> c_int;
    pub fn gsv_session_garble_streaming(s: *mut GsvSession, gate_id_base: u64, dir: *const c_char, first_index: u64, n_threads: c_int, hashes: *mut u8) -> c_int;
    pub fn gsv_session_garble_streaming_sink(s: *mut GsvSession, gate_id_base: u64, first_call: u64, n_calls: u64, sink: GsvCtSinkFn, user: *mut std::ffi::c_void, n_threads: c_int, hashes: *mut u8) -> c_int;
    pub fn gsv_session_garble_evaluate(garbler: *mut GsvSession, evaluator: *mut GsvSession, gate_id_base: u64, n_threads: c_int, hashes: *mut u8) -> c_int;
    pub fn gsv_session_evaluate_streaming_indexed(s: *mut GsvSession, gate_id_base: u64, dir: *const c_char, indexes: *const u64, hashes: *mut u8) -> c_int;
    pub fn gsv_session_evaluate_streaming_source(s: *mut GsvSession, gate_id_base: u64, source: GsvCtSourceFn, user: *mut std::ffi::c_void, hashes: *mut u8) -> c_int;
    pub fn gsv_session_set_evaluate_inputs(s: *mut GsvSession, const_active: *const u8, input_active: *const u8, input_bits: *const u8) -> c_int;
    pub fn gsv_session_upload_ciphertexts(s: *mut GsvSession, instance: usize, cts: *const u8, n: u64) -> c_int;
    pub fn gsv_session_evaluate(s: *mut GsvSession, gate_id_base: u64) -> c_int;
    pub fn gsv_session_evaluate_streaming(s: *mut GsvSession, gate_id_base: u64, dir: *const c_char, first_index: u64, hashes: *mut u8) -> c_int;
    pub fn gsv_session_sync(s: *mut GsvSession) -> c_int;
    pub fn gsv_session_read_outputs(s: *mut GsvSession, labels: *mut u8, bits: *mut u8) -> c_int;
    pub fn gsv_session_read_ciphertexts(s: *mut GsvSession, instance: usize, first: u64, n: u64, out: *mut u8) -> c_int;
    pub fn gsv_session_ciphertext_hash(s: *mut GsvSession, instance: usize, hash: *mut u8) -> c_int;
    pub fn gsv_commit_labels(labels: *const u8, n: u64, out: *mut u8) -> c_int;
    // labels, commitment helper, plan files and slices of a plan, schedule introspection, sample drains
    pub fn gsv_labels_from_seed(seed: u64, n_inputs: usize, delta: *mut u8, false_label0: *mut u8, true_label0: *mut u8, input_label0: *mut u8) -> c_int;
    pub fn gsv_cbcmac_update(state: *mut u8, records: *const u8, n_records: u64) -> c_int;
    pub fn gsv_plan_save(p: *const GsvPlan, path: *const c_char) -> c_int;
    pub fn gsv_plan_counts(p: *const GsvPlan, n_gates: *mut u64, n_ciphertexts: *mut u64, n_calls: *mut u64) -> c_int;
    pub fn gsv_plan_call_info(p: *const GsvPlan, call: u64, gate_offset: *mut u64, n_gates: *mut u64, ct_offset: *mut u64, n_ciphertexts: *mut u64, n_steps: *mut u64) -> c_int;
    pub fn gsv_session_plan_schedule_info(s: *const GsvSession, info: *mut GsvPlanScheduleInfo) -> c_int;
    pub fn gsv_session_plan_window(s: *const GsvSession, window: u64, first_call: *mut u64, n_calls: *mut u64, max_width: *mut u64) -> c_int;
    pub fn gsv_session_garble_streaming_calls(s: *mut GsvSession, gate_id_base: u64, first_call: u64, n_calls: u64, dir: *const c_char, first_index: u64, n_threads: c_int, hashes: *mut u8) -> c_int;
    pub fn gsv_session_set_drain_instances(s: *mut GsvSession, n: usize) -> c_int;
    pub fn gsv_session_instances_per_workgroup(s: *const GsvSession, out: *mut c_int) -> c_int;
    // round 6: deferred release (a `Drop` inside `CiphertextHandler::handle` is safe), the safe-schedule fallback, two plans from one build
    pub fn gsv_deferred_release_count() -> u64;
    pub fn gsv_session_fallback_count(s: *const GsvSession, n: *mut u64) -> c_int;
    pub fn gsv_plan_build_file_pair(circuit_spec: *const c_char, units_csv_a: *const c_char, path_a: *const c_char, window_div_a: u32, units_csv_b: *const c_char, path_b: *const c_char, window_div_b: u32) -> c_int;
    pub fn gsv_plan_call_record_form(p: *const GsvPlan, call: u64, and_terms: *mut u32) -> c_int;
}

pub fn chk(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(gsv_last_error()) }.to_string_lossy().into_owned();
        panic!("gsv engine: status {rc}: {msg}");
    }
}
