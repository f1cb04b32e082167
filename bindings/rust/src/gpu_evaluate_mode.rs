//! `GpuEvaluateMode<SRC>`: the evaluator's `impl CircuitMode`, mirror of `EvaluateMode` (src/circuit/modes/evaluate_mode.rs:59-196),
//! over the same recording as `GpuGarbleMode` (gpu_recorder.rs).  NEVER COMPILED in this repository (bindings/rust/README.md); the
//! C-ABI calls it makes are the ones tests/test_gpu_parity.py drives through ctypes (`test_generic_ciphertext_sink_and_source`,
//! `test_compressed_verifier_evaluates_valid_and_tampered_proof`).
//!
//! Semantics kept from `EvaluateMode`:
//!  * constants: `true_wire` / `false_wire` are the ACTIVE labels the garbler handed over (`EvaluateMode::new`, :70-79);
//!  * every wire carries (active label, plaintext bit) (`EvaluatedWire`, :14-18); the bit follows the plain truth function (:150) and
//!    `degarble_gate` branches on `a.value` (halfgates_garbling.rs:62-66): on the device the bit travels beside the label;
//!  * ciphertexts are pulled from the `CiphertextSource` in gate order, one per live non-free gate (:137-143); a source that runs dry
//!    is `panic!("Ciphertext source exhausted ...")` (:141) — here GSV_ERR_EXHAUSTED from the engine, turned into the same panic by `chk`;
//!  * the gate id advances on EVERY gate, dead or not (:128-133).
use std::collections::HashMap;
use std::num::NonZero;

use super::gpu_ffi::*;
use super::gpu_recorder::{GpuRecorder, UnitAction};
use crate::{
    Gate, S, WireId,
    circuit::{CircuitMode, FALSE_WIRE, TRUE_WIRE, ciphertext_source::CiphertextSource, component_key::ComponentKey, modes::EvaluatedWire},
    storage::Credits,
};

pub struct GpuEvaluateMode<SRC: CiphertextSource> {
    rec: GpuRecorder,
    source: SRC,
    false_wire: S,
    true_wire: S,
    inputs: Vec<EvaluatedWire>,          // root inputs in feed order
    input_index: HashMap<WireId, usize>,
    output_index: HashMap<WireId, usize>,
    results: Vec<EvaluatedWire>,
    device: i32,
}

impl<SRC: CiphertextSource> std::fmt::Debug for GpuEvaluateMode<SRC> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("GpuEvaluateMode").field("inputs", &self.inputs.len()).finish()
    }
}

impl<SRC: CiphertextSource> GpuEvaluateMode<SRC> {
    /// Mirror of `EvaluateMode::new(capacity, true_wire, false_wire, source)` (evaluate_mode.rs:70-79), flat recording.
    pub fn new(_capacity: usize, true_wire: S, false_wire: S, source: SRC) -> Self { Self::over(GpuRecorder::flat(), true_wire, false_wire, source) }
    /// Over the plan recorder (the verifier): see `GpuGarbleMode::with_plan`.  A plan file written by the garbler's host for the same
    /// verifying key can be loaded instead of recorded again (gsv_plan_load): the plan does not depend on the mode.
    pub fn with_plan(true_wire: S, false_wire: S, source: SRC, units: &[&str], window_div: u32, plan_file: Option<&str>) -> Self {
        Self::over(GpuRecorder::plan(units, window_div, plan_file), true_wire, false_wire, source)
    }
    fn over(rec: GpuRecorder, true_wire: S, false_wire: S, source: SRC) -> Self {
        Self { rec, source, false_wire, true_wire, inputs: vec![], input_index: HashMap::new(), output_index: HashMap::new(), results: vec![], device: 0 }
    }

    fn run(&mut self, outputs: &[WireId]) {
        let (mut engine, mut sess, mut plan, mut prog) = (std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut());
        chk(unsafe { gsv_engine_create(self.device, &mut engine) });
        if self.rec.is_plan() {
            plan = self.rec.finish_plan(outputs);
            if let Some(path) = self.rec.plan_file.clone() {
                unsafe { gsv_plan_destroy(plan) };
                chk(unsafe { gsv_plan_load(path.as_ptr(), engine, &mut plan) });
            }
            let opts = GsvPlanSessionOpts { retain_stream: 0, ..Default::default() }; // the stream is uploaded window by window, nothing retained
            chk(unsafe { gsv_session_create_plan_opts(engine, plan, 1, &opts, &mut sess) });
        } else {
            prog = self.rec.finish_flat(outputs);
            chk(unsafe { gsv_session_create(engine, prog, 1, 1, 1, &mut sess) });
        }
        let consts: Vec<u8> = [self.false_wire.to_bytes(), self.true_wire.to_bytes()].concat(); // {false active, true active}, evaluate_mode.rs:70
        let labels: Vec<u8> = self.inputs.iter().flat_map(|w| w.active_label.to_bytes()).collect();
        let bits: Vec<u8> = self.inputs.iter().map(|w| w.value as u8).collect();
        chk(unsafe { gsv_session_set_evaluate_inputs(sess, consts.as_ptr(), labels.as_ptr(), bits.as_ptr()) });
        // CiphertextSource::recv (ciphertext_source.rs:14-34): the engine pulls the stream in gate order in bounded runs
        unsafe extern "C" fn pull<SRC: CiphertextSource>(user: *mut std::ffi::c_void, _instance: usize, _first: u64, records: *mut u8, n: u64) -> std::os::raw::c_int {
            let src = &mut *(user as *mut SRC);
            let out = std::slice::from_raw_parts_mut(records, (n as usize) * 16);
            for rec in out.chunks_exact_mut(16) {
                match src.recv() { Some(ct) => rec.copy_from_slice(&ct.to_bytes()), None => return 1 } // -> GSV_ERR_EXHAUSTED (evaluate_mode.rs:139-142)
            }
            0
        }
        chk(unsafe { gsv_session_evaluate_streaming_source(sess, 0, pull::<SRC>, &mut self.source as *mut SRC as *mut std::ffi::c_void, std::ptr::null_mut()) });
        let (mut out, mut ob) = (vec![0u8; outputs.len() * 16], vec![0u8; outputs.len()]);
        chk(unsafe { gsv_session_read_outputs(sess, out.as_mut_ptr(), ob.as_mut_ptr()) });
        self.results = out.chunks_exact(16).zip(ob.iter()).map(|(b, v)| EvaluatedWire { active_label: S::from_bytes(b.try_into().unwrap()), value: *v != 0 }).collect();
        unsafe {
            gsv_session_destroy(sess);
            if !plan.is_null() { gsv_plan_destroy(plan) }
            if !prog.is_null() { gsv_program_destroy(prog) }
            gsv_engine_destroy(engine);
        }
    }
}

impl<SRC: CiphertextSource> CircuitMode for GpuEvaluateMode<SRC> {
    type WireValue = EvaluatedWire;
    type CiphertextAcc = SRC::Result;

    fn false_value(&self) -> EvaluatedWire { EvaluatedWire { active_label: self.false_wire, value: false } }
    fn true_value(&self) -> EvaluatedWire { EvaluatedWire { active_label: self.true_wire, value: true } }
    fn allocate_wire(&mut self, credits: Credits) -> WireId { self.rec.allocate_wire(credits) }
    fn evaluate_gate(&mut self, g: &Gate) { self.rec.evaluate_gate(g) } // evaluate_mode.rs:123-158, as "enqueue"

    fn feed_wire(&mut self, wire: WireId, value: EvaluatedWire) {
        if matches!(wire, TRUE_WIRE | FALSE_WIRE | WireId::UNREACHABLE) { return; }
        let recording_unit = self.rec.unit_action_in_progress();
        self.rec.declare_input(wire);
        if !recording_unit { self.input_index.insert(wire, self.inputs.len()); self.inputs.push(value); }
    }

    fn lookup_wire(&mut self, wire: WireId) -> Option<EvaluatedWire> {
        match wire {
            TRUE_WIRE => return Some(self.true_value()),
            FALSE_WIRE => return Some(self.false_value()),
            _ => (),
        }
        if let Some(i) = self.output_index.get(&wire) { return Some(self.results[*i].clone()); }
        if let Some(i) = self.input_index.get(&wire) { return Some(self.inputs[*i].clone()); }
        Some(EvaluatedWire::default()) // an unpin during the execution pass: the value is dropped (streaming_mode.rs:223-232)
    }

    fn add_credits(&mut self, _wires: &[WireId], _credits: NonZero<Credits>) {}

    fn execution_finished(&mut self, output_wires: &[WireId]) {
        let outs: Vec<WireId> = output_wires.iter().copied().filter(|w| !matches!(*w, TRUE_WIRE | FALSE_WIRE)).collect();
        for (i, w) in outs.iter().enumerate() { self.output_index.insert(*w, i); }
        self.run(&outs);
    }
    fn unit_begin(&mut self, key: ComponentKey, name: &str, output_liveness: &[bool]) -> UnitAction { self.rec.unit_begin(key, name, output_liveness) }
    fn unit_end(&mut self, outputs: &[WireId]) { self.rec.unit_end(outputs) }
    fn unit_call(&mut self, key: ComponentKey, output_liveness: &[bool], inputs: &[WireId]) -> Vec<WireId> { self.rec.unit_call(key, output_liveness, inputs) }

    fn finalize_ciphertext_accumulator(self) -> SRC::Result { self.source.finalize() }
}
