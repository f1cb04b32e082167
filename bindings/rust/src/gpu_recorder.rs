//! What `GpuGarbleMode` and `GpuEvaluateMode` share: the recording side of the C ABI (include/gsv_engine.h).  NEVER COMPILED in this
//! repository (no Rust toolchain in the build image, bindings/rust/README.md); the same calls in the same order are exercised by the C++
//! external host tests/ext_host/ext_host.cpp, whose `RecorderIo` / `AbiRecordMode` / `AbiPlanMode` this file mirrors one to one.
//!
//! Two recordings exist:
//!  * FLAT  (`gsv_recorder_*`): the whole circuit as one program — components up to ~10^8 gates (13 bytes of trace per gate);
//!  * PLAN  (`gsv_plan_recorder_*`): the 11 B-gate verifier.  Components named in `units` are taken over as CALLS through the
//!    `with_named_child` hook (streaming_mode_unit_hook.patch): the first call of a (ComponentKey, output liveness) pair records the
//!    body ON ITS OWN under a flat recorder (`unit_begin` .. `unit_end`: the driver runs the body as a root), compiles it in the
//!    background for the plan recorder (`gsv_program_compile_opts`, `for_plan`) and every call — the first one included — becomes one
//!    `gsv_plan_recorder_call`.  Everything between units is pushed as glue gates.
use std::collections::HashMap;
use std::ffi::CString;

use super::gpu_ffi::*;
use crate::{
    Gate, WireId,
    circuit::{FALSE_WIRE, TRUE_WIRE, component_key::ComponentKey},
    storage::Credits,
};

const FLUSH: usize = 1 << 16; // gates per push_gates call
const WIRE_BLOCK: usize = 4096; // wires per allocate_wires call

/// What the mode answers the `with_named_child` hook (streaming_mode_unit_hook.patch).
pub enum UnitAction {
    /// not a unit (or a component inside a unit that is being recorded): the driver runs the body gate by gate
    Inline,
    /// first call of this (key, liveness): the driver runs the body as the root of a stand-alone recording, then calls `unit_end`
    Record,
    /// compiled before: the driver skips the body and calls `unit_call`
    Cached,
}

struct Unit {
    program: *mut GsvProgram,
    out_index: Vec<i32>, // per component output: index into the program's outputs, -1 dead, -2 FALSE, -3 TRUE, -(4+k) input k passed through
    n_program_outputs: usize,
}

struct Io {
    pending: Vec<GsvGate>,
    next: u64,
    end: u64,
}

pub struct GpuRecorder {
    flat: *mut GsvRecorder,      // the circuit itself (flat recording) — null in plan recordings
    plan: *mut GsvPlanRecorder,  // plan recordings
    io: Io,                      // of `flat` / `plan`
    unit: *mut GsvRecorder,      // non-null while a unit body is being recorded on its own
    unit_io: Io,
    unit_key: Option<(ComponentKey, Vec<bool>)>,
    unit_inputs: Vec<WireId>,
    units: HashMap<(ComponentKey, Vec<bool>), Unit>,
    unit_names: Vec<String>,
    window_div: u32,
    pub plan_file: Option<CString>,
}

fn gate_record(g: &Gate) -> GsvGate {
    let id = |w: WireId| if w == WireId::UNREACHABLE { u64::MAX } else { w.0 as u64 };
    GsvGate { wire_a: id(g.wire_a), wire_b: id(g.wire_b), wire_c: id(g.wire_c), gate_type: g.gate_type as u8, pad: [0; 7] }
}

impl GpuRecorder {
    pub fn flat() -> Self {
        let mut r = std::ptr::null_mut();
        chk(unsafe { gsv_recorder_create(&mut r) });
        Self::with(r, std::ptr::null_mut(), vec![], 1, None)
    }
    /// `units`: component names (`module_path!() :: fn`, the string the #[component] macro hashes into the ComponentKey) taken over as calls.
    /// `plan_file`: the plan's programs go to this file as they are compiled (the verifier: 41 GB the host never holds), or None.
    pub fn plan(units: &[&str], window_div: u32, plan_file: Option<&str>) -> Self {
        let file = plan_file.map(|p| CString::new(p).unwrap());
        let opts = GsvPlanRecorderOpts {
            struct_size: std::mem::size_of::<GsvPlanRecorderOpts>() as u32,
            window_div,
            plan_file: file.as_ref().map_or(std::ptr::null(), |c| c.as_ptr()),
        };
        let mut r = std::ptr::null_mut();
        chk(unsafe { gsv_plan_recorder_create_opts(&opts, &mut r) });
        Self::with(std::ptr::null_mut(), r, units.iter().map(|s| s.to_string()).collect(), window_div, file)
    }
    fn with(flat: *mut GsvRecorder, plan: *mut GsvPlanRecorder, unit_names: Vec<String>, window_div: u32, plan_file: Option<CString>) -> Self {
        let io = || Io { pending: Vec::with_capacity(FLUSH), next: 0, end: 0 };
        Self { flat, plan, io: io(), unit: std::ptr::null_mut(), unit_io: io(), unit_key: None, unit_inputs: vec![], units: HashMap::new(), unit_names, window_div, plan_file }
    }
    pub fn is_plan(&self) -> bool { !self.plan.is_null() }
    /// A unit body is being recorded on its own right now (between `unit_begin` -> Record and `unit_end`).
    pub fn unit_action_in_progress(&self) -> bool { !self.unit.is_null() }

    // ---- CircuitMode::allocate_wire / evaluate_gate, as "enqueue" (garble_mode.rs:160-222, evaluate_mode.rs:123-158)
    pub fn allocate_wire(&mut self, credits: Credits) -> WireId {
        if credits == 0 { return WireId::UNREACHABLE; } // storage.rs:119-133: decided before the recorder hears of the wire
        let (io, unit, flat, plan) = (if self.unit.is_null() { &mut self.io } else { &mut self.unit_io }, self.unit, self.flat, self.plan);
        if io.next == io.end {
            let mut first = 0u64;
            if !unit.is_null() { chk(unsafe { gsv_recorder_allocate_wires(unit, WIRE_BLOCK, &mut first) }) }
            else if !plan.is_null() { chk(unsafe { gsv_plan_recorder_allocate_wires(plan, WIRE_BLOCK, &mut first) }) }
            else { chk(unsafe { gsv_recorder_allocate_wires(flat, WIRE_BLOCK, &mut first) }) }
            io.next = first;
            io.end = first + WIRE_BLOCK as u64;
        }
        io.next += 1;
        WireId((io.next - 1) as usize)
    }
    pub fn evaluate_gate(&mut self, g: &Gate) {
        let io = if self.unit.is_null() { &mut self.io } else { &mut self.unit_io };
        io.pending.push(gate_record(g));
        if io.pending.len() >= FLUSH { self.flush(); }
    }
    pub fn flush(&mut self) {
        if !self.unit.is_null() {
            if !self.unit_io.pending.is_empty() { chk(unsafe { gsv_recorder_push_gates(self.unit, self.unit_io.pending.as_ptr(), self.unit_io.pending.len()) }); self.unit_io.pending.clear(); }
            return;
        }
        if self.io.pending.is_empty() { return; }
        if !self.plan.is_null() { chk(unsafe { gsv_plan_recorder_push_gates(self.plan, self.io.pending.as_ptr(), self.io.pending.len()) }) }
        else { chk(unsafe { gsv_recorder_push_gates(self.flat, self.io.pending.as_ptr(), self.io.pending.len()) }) }
        self.io.pending.clear();
    }
    /// Root input (EncodeInput::encode -> feed_wire): the next circuit input.  While a unit is being recorded: the next input of the unit.
    pub fn declare_input(&mut self, wire: WireId) {
        if !self.unit.is_null() { chk(unsafe { gsv_recorder_declare_input(self.unit, wire.0 as u64) }); self.unit_inputs.push(wire); }
        else if !self.plan.is_null() { chk(unsafe { gsv_plan_recorder_declare_input(self.plan, wire.0 as u64) }) }
        else { chk(unsafe { gsv_recorder_declare_input(self.flat, wire.0 as u64) }) }
    }

    // ---- the with_named_child hook
    /// `name`: the component's name as the #[component] macro spells it (component_key.rs keeps key -> name, see the patch).
    pub fn unit_begin(&mut self, key: ComponentKey, name: &str, output_liveness: &[bool]) -> UnitAction {
        if self.plan.is_null() || !self.unit.is_null() || !self.unit_names.iter().any(|u| u == name) { return UnitAction::Inline; } // components inside a unit are flattened into it
        let k = (key, output_liveness.to_vec());
        if self.units.contains_key(&k) { return UnitAction::Cached; }
        self.flush(); // the glue gates in front of the call belong to the plan
        let mut r = std::ptr::null_mut();
        chk(unsafe { gsv_recorder_create(&mut r) });
        self.unit = r;
        self.unit_io.next = 0; self.unit_io.end = 0;
        self.unit_key = Some(k);
        self.unit_inputs.clear();
        UnitAction::Record
    }
    /// The body has run as a root; `outputs` are its output wires (in the unit's own id space).
    pub fn unit_end(&mut self, outputs: &[WireId]) {
        self.flush();
        let mut out_index = Vec::with_capacity(outputs.len());
        let mut produced: Vec<u64> = vec![];
        for w in outputs {
            if *w == WireId::UNREACHABLE { out_index.push(-1) }
            else if *w == FALSE_WIRE { out_index.push(-2) }
            else if *w == TRUE_WIRE { out_index.push(-3) }
            else if let Some(k) = self.unit_inputs.iter().position(|i| i == w) { out_index.push(-(4 + k as i32)) }
            else { out_index.push(produced.len() as i32); produced.push(w.0 as u64) }
        }
        chk(unsafe { gsv_recorder_declare_outputs(self.unit, produced.as_ptr(), produced.len()) });
        let opts = GsvCompileOpts { struct_size: std::mem::size_of::<GsvCompileOpts>() as u32, window_div: 0, keep_trace: 0, background: 1, consume_recorder: 1, reserved: 0, for_plan: self.plan };
        let mut program = std::ptr::null_mut();
        chk(unsafe { gsv_program_compile_opts(self.unit, &opts, &mut program) });
        unsafe { gsv_recorder_destroy(self.unit) };
        self.unit = std::ptr::null_mut();
        self.units.insert(self.unit_key.take().unwrap(), Unit { program, out_index, n_program_outputs: produced.len() });
    }
    /// One call of a compiled unit: `inputs` are the parent's wires; returns the component's output wires in the parent's id space.
    pub fn unit_call(&mut self, key: ComponentKey, output_liveness: &[bool], inputs: &[WireId]) -> Vec<WireId> {
        self.flush();
        let u = &self.units[&(key, output_liveness.to_vec())];
        let ins: Vec<u64> = inputs.iter().map(|w| w.0 as u64).collect(); // FALSE_WIRE / TRUE_WIRE are 0 / 1 on both sides
        let mut produced = vec![0u64; u.n_program_outputs];
        chk(unsafe { gsv_plan_recorder_call(self.plan, u.program, ins.as_ptr(), produced.as_mut_ptr()) });
        u.out_index.iter().map(|oi| match *oi {
            -1 => WireId::UNREACHABLE,
            -2 => FALSE_WIRE,
            -3 => TRUE_WIRE,
            k if k <= -4 => inputs[(-k - 4) as usize],
            k => WireId(produced[k as usize] as usize),
        }).collect()
    }

    // ---- after the execution pass
    /// Flat recording: declare the outputs and compile.  The caller owns the program.
    pub fn finish_flat(&mut self, outputs: &[WireId]) -> *mut GsvProgram {
        self.flush();
        let outs: Vec<u64> = outputs.iter().map(|w| w.0 as u64).collect();
        chk(unsafe { gsv_recorder_declare_outputs(self.flat, outs.as_ptr(), outs.len()) });
        let mut p = std::ptr::null_mut();
        chk(unsafe { gsv_program_compile(self.flat, std::ptr::null(), std::ptr::null(), 0, &mut p) });
        p
    }
    /// Plan recording: finish the plan (with a plan file: complete the file; the returned plan then holds metadata only and the caller
    /// loads the file with gsv_plan_load).
    pub fn finish_plan(&mut self, outputs: &[WireId]) -> *mut GsvPlan {
        self.flush();
        let outs: Vec<u64> = outputs.iter().map(|w| w.0 as u64).collect();
        let mut plan = std::ptr::null_mut();
        chk(unsafe { gsv_plan_recorder_finish(self.plan, outs.as_ptr(), outs.len(), &mut plan) });
        plan
    }
    pub fn window_div(&self) -> u32 { self.window_div }
}

impl Drop for GpuRecorder {
    fn drop(&mut self) {
        unsafe {
            if !self.unit.is_null() { gsv_recorder_destroy(self.unit) }
            if !self.plan.is_null() { gsv_plan_recorder_destroy(self.plan) } // waits for the background compilations that write its plan file
            for (_, u) in self.units.drain() { gsv_program_destroy(u.program) }
            if !self.flat.is_null() { gsv_recorder_destroy(self.flat) }
        }
    }
}
