//! `GpuGarbleMode<CTH>`: one more `impl CircuitMode` next to `garble_mode.rs` (src/circuit/modes.rs:26-51) whose per-gate loop body is
//! "enqueue": gates are recorded through the C ABI of libgsv_engine.so, compiled into a device schedule once, and garbled on the
//! MI355X when the driver first needs a value that only the run can produce (the root outputs, circuit/mod.rs:283) or when the
//! ciphertext accumulator is finalised.  NEVER COMPILED in this repository (no Rust toolchain in the build image): see README.md.
//!
//! Semantics kept from `GarbleMode` (garble_mode.rs:80-267):
//!  * randomness: `ChaChaRng::seed_from_u64(seed)`, then Delta, false.label0, true.label0, one label0 per `issue_garbled_wire`
//!    (:80-97, :116-118) — the engine never draws randomness when driven from Rust;
//!  * `allocate_wire(0)` is `WireId::UNREACHABLE` (storage.rs:119-133) and a gate with an UNREACHABLE output still consumes its gate
//!    id (:192-197): both are decided inside the recorder, which sees the same calls in the same order;
//!  * ciphertexts reach the `CiphertextHandler` in gate order (circuit/mod.rs:140-178).
//!
//! Two recorders exist on the engine side (include/gsv_engine.h):
//!  * `gsv_recorder_*`      a flat recording — components up to ~10^8 gates (13 bytes of trace per gate);
//!  * `gsv_plan_recorder_*` component-level programs — the 11 B-gate verifier.  It needs the `with_named_child` hook of
//!    `streaming_mode_unit_hook.patch`: unit components are recorded and compiled once per (ComponentKey, output liveness) and then
//!    only referenced.  `GpuGarbleMode::with_plan(units)` selects it.
use std::num::NonZero;

use rand::SeedableRng;
use rand_chacha::ChaChaRng;

use super::gpu_ffi::*;
use crate::{
    Delta, Gate, S, WireId,
    circuit::{CiphertextHandler, CircuitMode, FALSE_WIRE, TRUE_WIRE, modes::GarbledWire},
    storage::Credits,
};

const FLUSH: usize = 1 << 16;

enum Recorder {
    Flat(*mut GsvRecorder),
    Plan(*mut GsvPlanRecorder, Vec<String>), // unit component names, e.g. "fq12::mul_montgomery"
}

pub struct GpuGarbleMode<CTH: CiphertextHandler> {
    rec: Recorder,
    pending: Vec<GsvGate>,
    rng: ChaChaRng,
    delta: Delta,
    false_wire: GarbledWire,
    true_wire: GarbledWire,
    inputs: Vec<(WireId, S)>,     // root inputs in feed order: (wire, label0)
    outputs: Vec<WireId>,         // wires whose values were asked for after the execution pass, in order
    results: Option<Vec<S>>,      // label0 per `outputs` entry once the GPU has run
    handler: Option<CTH>,
    device: i32,
}

impl<CTH: CiphertextHandler> std::fmt::Debug for GpuGarbleMode<CTH> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("GpuGarbleMode").field("pending", &self.pending.len()).finish()
    }
}

impl<CTH: CiphertextHandler> GpuGarbleMode<CTH> {
    /// Mirror of `GarbleMode::new(capacity, seed, output_handler)` (garble_mode.rs:80-97); `capacity` is not needed: the compiler's
    /// lifetime pass sizes the wire file.
    pub fn new(_capacity: usize, seed: u64, handler: CTH) -> Self {
        let mut rng = ChaChaRng::seed_from_u64(seed);
        let delta = Delta::generate(&mut rng);
        let [false_wire, true_wire] = std::array::from_fn(|_| GarbledWire::random(&mut rng, &delta));
        let mut rec = std::ptr::null_mut();
        chk(unsafe { gsv_recorder_create(&mut rec) });
        Self { rec: Recorder::Flat(rec), pending: Vec::with_capacity(FLUSH), rng, delta, false_wire, true_wire, inputs: vec![], outputs: vec![], results: None, handler: Some(handler), device: 0 }
    }
    /// The same over the plan recorder: `units` are the component names `StreamingMode::with_named_child` hands over as calls.
    pub fn with_plan(seed: u64, handler: CTH, units: &[&str]) -> Self {
        let mut m = Self::new(0, seed, handler);
        if let Recorder::Flat(r) = m.rec { unsafe { gsv_recorder_destroy(r) } }
        let mut rec = std::ptr::null_mut();
        chk(unsafe { gsv_plan_recorder_create(&mut rec) });
        m.rec = Recorder::Plan(rec, units.iter().map(|s| s.to_string()).collect());
        m
    }
    pub fn issue_garbled_wire(&mut self) -> GarbledWire { GarbledWire::random(&mut self.rng, &self.delta) } // garble_mode.rs:116-118

    fn flush(&mut self) {
        if self.pending.is_empty() { return; }
        match &self.rec {
            Recorder::Flat(r) => chk(unsafe { gsv_recorder_push_gates(*r, self.pending.as_ptr(), self.pending.len()) }),
            Recorder::Plan(r, _) => chk(unsafe { gsv_plan_recorder_push_gates(*r, self.pending.as_ptr(), self.pending.len()) }),
        }
        self.pending.clear();
    }

    /// Hook target of `streaming_mode_unit_hook.patch`: is this component a unit, i.e. recorded on its own and called?
    pub fn is_unit(&self, component_name: &str) -> bool {
        matches!(&self.rec, Recorder::Plan(_, units) if units.iter().any(|u| u == component_name))
    }
    /// Hook target: call of a unit whose program the caller looked up / compiled (one per ComponentKey x output liveness; the body is
    /// recorded with `gsv_recorder_*` under a root that gives output i a credit only if the parent reads it, outputs = the wires the
    /// body produces).  Returns one fresh parent wire per program output.
    pub fn call_unit(&mut self, program: *const GsvProgram, inputs: &[WireId], n_outputs: usize) -> Vec<WireId> {
        self.flush();
        let Recorder::Plan(r, _) = &self.rec else { panic!("call_unit on a flat recorder") };
        let ins: Vec<u64> = inputs.iter().map(|w| w.0 as u64).collect();
        let mut outs = vec![0u64; n_outputs];
        chk(unsafe { gsv_plan_recorder_call(*r, program, ins.as_ptr(), outs.as_mut_ptr()) });
        outs.into_iter().map(|w| WireId(w as usize)).collect()
    }

    /// Compile, garble on the GPU, stream the ciphertexts (gate order) through the handler, fetch the requested labels.
    fn run(&mut self) {
        if self.results.is_some() { return; }
        self.flush();
        let outs: Vec<u64> = self.outputs.iter().map(|w| w.0 as u64).collect();
        let (mut engine, mut sess) = (std::ptr::null_mut(), std::ptr::null_mut());
        chk(unsafe { gsv_engine_create(self.device, &mut engine) });
        match &self.rec {
            Recorder::Flat(r) => {
                chk(unsafe { gsv_recorder_declare_outputs(*r, outs.as_ptr(), outs.len()) });
                let mut prog = std::ptr::null_mut();
                chk(unsafe { gsv_program_compile(*r, std::ptr::null(), std::ptr::null(), 0, &mut prog) });
                chk(unsafe { gsv_session_create(engine, prog, 1, 1, 1, &mut sess) });
            }
            Recorder::Plan(r, _) => {
                let mut plan = std::ptr::null_mut();
                chk(unsafe { gsv_plan_recorder_finish(*r, outs.as_ptr(), outs.len(), &mut plan) });
                // nothing retained: the ~48 GB stream of the verifier leaves the device segment by segment of the running window
                let opts = GsvPlanSessionOpts { retain_stream: 0, ..Default::default() };
                chk(unsafe { gsv_session_create_plan_opts(engine, plan, 1, &opts, &mut sess) });
            }
        }
        let label = |s: &S| s.to_bytes();
        let delta = label(&self.delta);
        let consts: Vec<u8> = [label(&self.false_wire.label0), label(&self.true_wire.label0)].concat();
        let ins: Vec<u8> = self.inputs.iter().flat_map(|(_, l)| label(l)).collect();
        chk(unsafe { gsv_session_set_garble_inputs(sess, delta.as_ptr(), consts.as_ptr(), ins.as_ptr()) });
        // CiphertextHandler::handle (circuit/mod.rs:140-178) in gate order: the engine garbles, drains the stream beside the running
        // launch and hands every run of records to this callback — any handler, channel Sender<S> included (circuit/mod.rs:160-170).
        // (AESAccumulatingHash alone could take `hashes` of gsv_session_garble_streaming, a file handler its `dir`: same bytes.)
        unsafe extern "C" fn sink<H: CiphertextHandler>(user: *mut std::ffi::c_void, _instance: usize, _first: u64, records: *const u8, n: u64) -> c_int {
            let handler = &mut *(user as *mut H);
            let bytes = std::slice::from_raw_parts(records, (n as usize) * 16);
            for rec in bytes.chunks_exact(16) { handler.handle(S::from_bytes(rec.try_into().unwrap())); }
            0
        }
        let mut handler = self.handler.take().expect("already finalised");
        chk(unsafe { gsv_session_garble_streaming_sink(sess, 0, 0, 0, sink::<CTH>, &mut handler as *mut CTH as *mut std::ffi::c_void, 1, std::ptr::null_mut()) });
        self.handler = Some(handler);
        let mut out = vec![0u8; outs.len() * 16];
        chk(unsafe { gsv_session_read_outputs(sess, out.as_mut_ptr(), std::ptr::null_mut()) });
        self.results = Some(out.chunks_exact(16).map(|b| S::from_bytes(b.try_into().unwrap())).collect());
        unsafe { gsv_session_destroy(sess); gsv_engine_destroy(engine); }
    }
}

impl<CTH: CiphertextHandler> CircuitMode for GpuGarbleMode<CTH> {
    type WireValue = GarbledWire;
    type CiphertextAcc = CTH::Result;

    fn false_value(&self) -> GarbledWire { self.false_wire.clone() }
    fn true_value(&self) -> GarbledWire { self.true_wire.clone() }

    fn allocate_wire(&mut self, credits: Credits) -> WireId {
        let mut w = 0u64;
        match &self.rec {
            Recorder::Flat(r) => chk(unsafe { gsv_recorder_allocate_wire(*r, credits, &mut w) }),
            Recorder::Plan(r, _) => chk(unsafe { gsv_plan_recorder_allocate_wire(*r, credits, &mut w) }),
        }
        if w == u64::MAX { WireId::UNREACHABLE } else { WireId(w as usize) }
    }

    // garble_mode.rs:160-222, as "enqueue"
    fn evaluate_gate(&mut self, g: &Gate) {
        let id = |w: WireId| if w == WireId::UNREACHABLE { u64::MAX } else { w.0 as u64 };
        self.pending.push(GsvGate { wire_a: id(g.wire_a), wire_b: id(g.wire_b), wire_c: id(g.wire_c), gate_type: g.gate_type as u8, pad: [0; 7] });
        if self.pending.len() == FLUSH { self.flush(); }
    }

    // root inputs (EncodeInput::encode, garbled_groth16.rs:156-176): the label0 stays on the host until the run
    fn feed_wire(&mut self, wire: WireId, value: GarbledWire) {
        if matches!(wire, TRUE_WIRE | FALSE_WIRE | WireId::UNREACHABLE) { return; }
        match &self.rec {
            Recorder::Flat(r) => chk(unsafe { gsv_recorder_declare_input(*r, wire.0 as u64) }),
            Recorder::Plan(r, _) => chk(unsafe { gsv_plan_recorder_declare_input(*r, wire.0 as u64) }),
        }
        self.inputs.push((wire, value.label0));
    }

    // Most calls discard the value (child "unpin", streaming_mode.rs:223-232).  Values exist for the constants, for root inputs
    // (circuit/mod.rs:270-273) and — after the run — for the wires asked for once the execution pass is over (outputs, :283).
    fn lookup_wire(&mut self, wire: WireId) -> Option<GarbledWire> {
        match wire {
            TRUE_WIRE => return Some(self.true_value()),
            FALSE_WIRE => return Some(self.false_value()),
            _ => (),
        }
        if let Some((_, l0)) = self.inputs.iter().find(|(w, _)| *w == wire) { return Some(GarbledWire { label0: *l0, label1: *l0 ^ &self.delta }); }
        if self.pending.is_empty() && self.results.is_none() && self.outputs.is_empty() { return Some(self.false_value()); } // unpin before any gate: value unused
        // a produced wire: remember it as an output; the first such request after the last gate triggers the GPU run
        if let Some(i) = self.outputs.iter().position(|w| *w == wire) {
            self.run();
            let l0 = self.results.as_ref().unwrap()[i];
            return Some(GarbledWire { label0: l0, label1: l0 ^ &self.delta });
        }
        self.outputs.push(wire);
        Some(self.false_value()) // placeholder for unpin-style lookups; `CircuitOutput::decode` asks again through `output_value`
    }

    fn add_credits(&mut self, _wires: &[WireId], _credits: NonZero<Credits>) {} // the recorder keeps SSA wires; credits only decide dead gates

    fn finalize_ciphertext_accumulator(mut self) -> CTH::Result {
        self.run();
        self.handler.take().unwrap().finalize()
    }
}

impl<CTH: CiphertextHandler> GpuGarbleMode<CTH> {
    /// What `CircuitOutput::decode` (circuit/mod.rs:364-409) should call instead of `lookup_wire` for a root output: declares the
    /// wire as an output of the recording and returns its garbled wire after the GPU run.
    pub fn output_value(&mut self, wire: WireId) -> GarbledWire {
        if !self.outputs.contains(&wire) { assert!(self.results.is_none(), "outputs must be declared before the run"); self.outputs.push(wire); }
        self.run();
        let i = self.outputs.iter().position(|w| *w == wire).unwrap();
        let l0 = self.results.as_ref().unwrap()[i];
        GarbledWire { label0: l0, label1: l0 ^ &self.delta }
    }
}
