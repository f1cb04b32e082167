//! `GpuGarbleMode<CTH>`: one more `impl CircuitMode` next to `garble_mode.rs` (src/circuit/modes.rs:26-51) whose per-gate loop body is
//! "enqueue": gates are recorded through the C ABI of libgsv_engine.so (gpu_recorder.rs), compiled into a device schedule once, and
//! garbled on the MI355X when the execution pass is over (`execution_finished`, the one-line hook the patch adds to
//! `CircuitBuilder::run_streaming`, circuit/mod.rs:283).  NEVER COMPILED in this repository (no Rust toolchain in the build image): see
//! README.md; tests/ext_host/ext_host.cpp is the compiled, GPU-tested C++ mirror of this file.
//!
//! Semantics kept from `GarbleMode` (garble_mode.rs:80-267):
//!  * randomness: `ChaChaRng::seed_from_u64(seed)`, then Delta, false.label0, true.label0, one label0 per `issue_garbled_wire`
//!    (:80-97, :116-118) — the engine never draws randomness when driven from Rust;
//!  * `allocate_wire(0)` is `WireId::UNREACHABLE` (storage.rs:119-133) and a gate with an UNREACHABLE output still consumes its gate
//!    id (:192-197): both are decided by the recorder, which sees the same calls in the same order;
//!  * ciphertexts reach the `CiphertextHandler` in gate order (circuit/mod.rs:140-178).
//!
//! `lookup_wire` during the execution pass is only ever an "unpin" whose value is dropped (streaming_mode.rs:223-232): it returns a
//! placeholder and records NOTHING (round 4's version filed every such wire as an output: millions of entries and a linear scan each).
//! The wires whose values are wanted are told to the mode once, by `execution_finished(&output_wires)`; lookups after it are O(1).
use std::collections::HashMap;
use std::num::NonZero;

use rand::SeedableRng;
use rand_chacha::ChaChaRng;

use super::gpu_ffi::*;
use super::gpu_recorder::{GpuRecorder, UnitAction};
use crate::{
    Delta, Gate, S, WireId,
    circuit::{CiphertextHandler, CircuitMode, FALSE_WIRE, TRUE_WIRE, component_key::ComponentKey, modes::GarbledWire},
    storage::Credits,
};

pub struct GpuGarbleMode<CTH: CiphertextHandler> {
    rec: GpuRecorder,
    rng: ChaChaRng,
    delta: Delta,
    false_wire: GarbledWire,
    true_wire: GarbledWire,
    inputs: Vec<S>,                      // label0 of the root inputs, in feed order
    input_index: HashMap<WireId, usize>, // wire -> position in `inputs`
    output_index: HashMap<WireId, usize>,
    results: Vec<S>,                     // label0 per declared output once the GPU has run
    handler: Option<CTH>,
    device: i32,
}

impl<CTH: CiphertextHandler> std::fmt::Debug for GpuGarbleMode<CTH> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("GpuGarbleMode").field("inputs", &self.inputs.len()).finish()
    }
}

impl<CTH: CiphertextHandler> GpuGarbleMode<CTH> {
    /// Mirror of `GarbleMode::new(capacity, seed, output_handler)` (garble_mode.rs:80-97), flat recording; `capacity` is not needed:
    /// the compiler's lifetime pass sizes the wire file.
    pub fn new(_capacity: usize, seed: u64, handler: CTH) -> Self { Self::over(GpuRecorder::flat(), seed, handler) }
    /// The same over the plan recorder: `units` are the component names taken over as calls (the verifier: bench.py VERIFIER_UNITS),
    /// `plan_file` where the plan's programs are spilled while they are compiled (None: the images stay in host memory).
    pub fn with_plan(seed: u64, handler: CTH, units: &[&str], window_div: u32, plan_file: Option<&str>) -> Self {
        Self::over(GpuRecorder::plan(units, window_div, plan_file), seed, handler)
    }
    fn over(rec: GpuRecorder, seed: u64, handler: CTH) -> Self {
        let mut rng = ChaChaRng::seed_from_u64(seed);
        let delta = Delta::generate(&mut rng);
        let [false_wire, true_wire] = std::array::from_fn(|_| GarbledWire::random(&mut rng, &delta));
        Self { rec, rng, delta, false_wire, true_wire, inputs: vec![], input_index: HashMap::new(), output_index: HashMap::new(), results: vec![], handler: Some(handler), device: 0 }
    }
    pub fn issue_garbled_wire(&mut self) -> GarbledWire { GarbledWire::random(&mut self.rng, &self.delta) } // garble_mode.rs:116-118

    /// Compile, garble on the GPU, stream the ciphertexts (gate order) through the handler, fetch the output labels.
    fn run(&mut self, outputs: &[WireId]) {
        let (mut engine, mut sess, mut plan, mut prog) = (std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut());
        chk(unsafe { gsv_engine_create(self.device, &mut engine) });
        if self.rec.is_plan() {
            plan = self.rec.finish_plan(outputs);
            if let Some(path) = self.rec.plan_file.clone() {
                unsafe { gsv_plan_destroy(plan) }; // metadata only: the records are in the file
                chk(unsafe { gsv_plan_load(path.as_ptr(), engine, &mut plan) });
            }
            // nothing retained: the ~48 GB stream of the verifier leaves the device segment by segment of the running window
            let opts = GsvPlanSessionOpts { retain_stream: 0, ..Default::default() };
            chk(unsafe { gsv_session_create_plan_opts(engine, plan, 1, &opts, &mut sess) });
        } else {
            prog = self.rec.finish_flat(outputs);
            chk(unsafe { gsv_session_create(engine, prog, 1, 1, 1, &mut sess) });
        }
        let delta = self.delta.to_bytes();
        let consts: Vec<u8> = [self.false_wire.label0.to_bytes(), self.true_wire.label0.to_bytes()].concat();
        let ins: Vec<u8> = self.inputs.iter().flat_map(|l| l.to_bytes()).collect();
        chk(unsafe { gsv_session_set_garble_inputs(sess, delta.as_ptr(), consts.as_ptr(), ins.as_ptr()) });
        // CiphertextHandler::handle (circuit/mod.rs:140-178) in gate order: the engine garbles, drains the stream beside the running
        // launch and hands every run of records to this callback — any handler, channel Sender<S> included (circuit/mod.rs:160-170).
        unsafe extern "C" fn sink<H: CiphertextHandler>(user: *mut std::ffi::c_void, _instance: usize, _first: u64, records: *const u8, n: u64) -> std::os::raw::c_int {
            let handler = &mut *(user as *mut H);
            let bytes = std::slice::from_raw_parts(records, (n as usize) * 16);
            for rec in bytes.chunks_exact(16) { handler.handle(S::from_bytes(rec.try_into().unwrap())); }
            0
        }
        let mut handler = self.handler.take().expect("already finalised");
        chk(unsafe { gsv_session_garble_streaming_sink(sess, 0, 0, 0, sink::<CTH>, &mut handler as *mut CTH as *mut std::ffi::c_void, 1, std::ptr::null_mut()) });
        self.handler = Some(handler);
        let mut out = vec![0u8; outputs.len() * 16];
        chk(unsafe { gsv_session_read_outputs(sess, out.as_mut_ptr(), std::ptr::null_mut()) });
        self.results = out.chunks_exact(16).map(|b| S::from_bytes(b.try_into().unwrap())).collect();
        // (round 6: these calls are safe from ANY thread at ANY time — also from inside a CiphertextHandler of another mode that is in the
        //  middle of its pass: the engine queues releases while a streaming pass is in flight and runs them when it has ended,
        //  include/gsv_engine.h "Deferred release"; tests/ext_host/ext_host.cpp --destroy-in-sink is the compiled twin)
        unsafe {
            gsv_session_destroy(sess);
            if !plan.is_null() { gsv_plan_destroy(plan) }
            if !prog.is_null() { gsv_program_destroy(prog) }
            gsv_engine_destroy(engine);
        }
    }
}

impl<CTH: CiphertextHandler> CircuitMode for GpuGarbleMode<CTH> {
    type WireValue = GarbledWire;
    type CiphertextAcc = CTH::Result;

    fn false_value(&self) -> GarbledWire { self.false_wire.clone() }
    fn true_value(&self) -> GarbledWire { self.true_wire.clone() }
    fn allocate_wire(&mut self, credits: Credits) -> WireId { self.rec.allocate_wire(credits) }
    fn evaluate_gate(&mut self, g: &Gate) { self.rec.evaluate_gate(g) } // garble_mode.rs:160-222, as "enqueue"

    // root inputs (EncodeInput::encode, garbled_groth16.rs:156-176): the label0 stays on the host until the run.  While a unit body is
    // recorded on its own (hook) the driver feeds the unit's inputs through here too: they only exist in the recording.
    fn feed_wire(&mut self, wire: WireId, value: GarbledWire) {
        if matches!(wire, TRUE_WIRE | FALSE_WIRE | WireId::UNREACHABLE) { return; }
        let recording_unit = self.rec.unit_action_in_progress();
        self.rec.declare_input(wire);
        if !recording_unit { self.input_index.insert(wire, self.inputs.len()); self.inputs.push(value.label0); }
    }

    fn lookup_wire(&mut self, wire: WireId) -> Option<GarbledWire> {
        match wire {
            TRUE_WIRE => return Some(self.true_value()),
            FALSE_WIRE => return Some(self.false_value()),
            _ => (),
        }
        let pair = |l0: S, d: &Delta| GarbledWire { label0: l0, label1: l0 ^ d };
        if let Some(i) = self.output_index.get(&wire) { return Some(pair(self.results[*i], &self.delta)); } // after the run: O(1)
        if let Some(i) = self.input_index.get(&wire) { return Some(pair(self.inputs[*i], &self.delta)); }   // circuit/mod.rs:270-273
        Some(self.false_value()) // an unpin during the execution pass (streaming_mode.rs:223-232): the value is dropped, nothing is recorded
    }

    fn add_credits(&mut self, _wires: &[WireId], _credits: NonZero<Credits>) {} // the recorder keeps SSA wires; credits only decide dead gates

    // ---- hooks added by the patch (defaults in the trait do nothing / say Inline)
    fn execution_finished(&mut self, output_wires: &[WireId]) {
        let outs: Vec<WireId> = output_wires.iter().copied().filter(|w| !matches!(*w, TRUE_WIRE | FALSE_WIRE)).collect();
        for (i, w) in outs.iter().enumerate() { self.output_index.insert(*w, i); }
        self.run(&outs);
    }
    fn unit_begin(&mut self, key: ComponentKey, name: &str, output_liveness: &[bool]) -> UnitAction { self.rec.unit_begin(key, name, output_liveness) }
    fn unit_end(&mut self, outputs: &[WireId]) { self.rec.unit_end(outputs) }
    fn unit_call(&mut self, key: ComponentKey, output_liveness: &[bool], inputs: &[WireId]) -> Vec<WireId> { self.rec.unit_call(key, output_liveness, inputs) }

    fn finalize_ciphertext_accumulator(mut self) -> CTH::Result { self.handler.take().unwrap().finalize() }
}
