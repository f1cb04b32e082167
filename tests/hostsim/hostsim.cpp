// TEST-ONLY host interpreter of the engine's compiled gate programs.
//
// Purpose: exercise, on a machine WITHOUT a GPU, exactly the pieces the gfx950 kernel is built from —
// RecordMode + compile_program (csrc/engine/program.hpp), the T-table AES / half-gate math of
// csrc/engine/gate_math.hpp and the table/seed code of csrc/engine/host_crypto.hpp — by walking the
// device schedule step by step the way run_program_kernel does.  Gates inside a step are visited in
// REVERSE order so that any dependency between gates of one step (a scheduling bug) shows up as a
// mismatch against the oracle.  This file is linked only into tests/hostsim/libgsv_hostsim.so; the
// product library (libgsv_engine.so) has no CPU execution path.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../garbled_snark_verifier_amd/csrc/engine/gate_math.hpp"
#include "../../garbled_snark_verifier_amd/csrc/engine/host_crypto.hpp"
#include "../../garbled_snark_verifier_amd/csrc/engine/plan_builder.hpp"
#include "../../garbled_snark_verifier_amd/csrc/engine/program.hpp"
#include "../../garbled_snark_verifier_amd/csrc/engine/schedule.hpp"
#include "../../garbled_snark_verifier_amd/csrc/gadgets/circuits.hpp"

using namespace gsv;
using gsv::dev::Label;

static thread_local std::string g_err;
static int g_hasher = 0;  // 0 AES, 1 Blake3

static Label load(const uint8_t* p) { Label l; std::memcpy(l.w, p, 16); return l; }
static void store(uint8_t* p, const Label& l) { std::memcpy(p, l.w, 16); }

struct SimProgram {
  Program prog;
};

extern "C" {

const char* hostsim_last_error() { return g_err.c_str(); }

int hostsim_compile(const char* spec, int chain_feedback, SimProgram** out, uint64_t* info /* 18 */) {
  try {
    NamedCircuit nc = make_circuit(spec);
    RecordMode mode;
    StreamingRunner run(mode, nc.n_inputs, nc.fn);
    std::vector<uint32_t> in_ssa, out_ssa;
    for (WireId w : run.prepare()) in_ssa.push_back(mode.define_input(w));
    for (WireId w : run.execute()) out_ssa.push_back(mode.current(w));
    std::vector<std::pair<uint32_t, uint32_t>> fb;
    if (chain_feedback) for (uint32_t i = 0; i < out_ssa.size(); ++i) fb.push_back({i, i});  // output i -> input i
    auto sp = std::make_unique<SimProgram>();
    CompileOptions opt;
    if (const char* e = getenv("GSV_FUSE")) opt.fuse = atoi(e) != 0;  // same knob as the engine (engine.cpp)
    if (const char* e = getenv("GSV_LDS_SLOTS")) opt.lds_slots = std::min<uint32_t>(uint32_t(atoi(e)), LDS_WINDOW_SLOTS);
    sp->prog = compile_program(mode.trace(), in_ssa, out_ssa, fb, opt);
    const Program& g = sp->prog;
    if (info) {
      info[0] = g.input_slots.size(); info[1] = g.output_slots.size(); info[2] = g.n_gates; info[3] = g.n_ct; info[4] = g.n_dead;
      info[5] = g.steps.size(); info[6] = g.and_depth; info[7] = g.n_and_steps; info[8] = g.max_step_width; info[9] = g.n_slots;
      info[10] = g.peak_live; info[11] = run.ctx().component_calls;
      info[12] = g.n_lds_slots; info[13] = g.reads_lds; info[14] = g.reads_hbm; info[15] = g.writes_lds; info[16] = g.writes_hbm;
      info[17] = g.n_fused_free;
    }
    *out = sp.release();
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void hostsim_free(SimProgram* p) { delete p; }

// One pass of one compiled program over a wire file (W labels / VB plaintext bits, at least g.n_slots entries; slots 0/1/2
// = FALSE / TRUE / ZERO already set, inputs already in g.input_slots).  `ct` is this pass's ciphertext block in GATE order.
static void interpret(const Program& g, bool evaluate, uint64_t gb, const Label& d, std::vector<uint8_t>& W, std::vector<uint8_t>& VB, uint8_t* ct) {
  const AesTables& T = AesTables::fixed_key();
  dev::PlainTables aes{{T.te[0], T.te[1], T.te[2], T.te[3]}, T.rk};
  std::vector<uint8_t> LW(size_t(LDS_WINDOW_SLOTS) * 16, 0x5A), LB(LDS_WINDOW_SLOTS, 0);  // LDS window image
  std::memset(&LW[0], 0, 16);  // window entry 0: the all-zero label of absent operands
  LB[0] = 0;
  auto lab = [&](uint32_t slot) -> uint8_t* { return (slot & SLOT_LDS_FLAG) ? &LW[size_t(slot & SLOT_INDEX_MASK) * 16] : &W[size_t(slot) * 16]; };
  auto bit = [&](uint32_t slot) -> uint8_t& { return (slot & SLOT_LDS_FLAG) ? LB[slot & SLOT_INDEX_MASK] : VB[slot]; };
  // the records carry PROGRAM-order ciphertext positions; `ct` is in gate order
  std::vector<uint32_t> gate_of(g.ct_pos.size());
  for (size_t k = 0; k < g.ct_pos.size(); ++k) gate_of[g.ct_pos[k]] = uint32_t(k);
  for (const StepDesc& sd : g.steps) {
    // snapshot semantics: all reads of a step see the state before the step (as on the GPU,
    // where every lane loads its operands before anyone's store is guaranteed visible)
    std::vector<std::pair<uint32_t, Label>> wr;
    std::vector<std::pair<uint32_t, uint8_t>> wb;
    wr.reserve(sd.and_cnt + sd.xor_cnt);
    for (uint32_t k = sd.xor_cnt; k-- > 0;) {
      const XorRec& r = g.xors[sd.xor_off + k];
      const uint32_t sx[4] = {uint32_t(r.w0) & SLOT_MASK, uint32_t(r.w0 >> 21) & SLOT_MASK, uint32_t(r.w0 >> 42) & SLOT_MASK, uint32_t(r.w1) & SLOT_MASK};
      const uint32_t sc = uint32_t(r.w1 >> 21) & SLOT_MASK, par = uint32_t(r.w0 >> 63);
      Label x{{0, 0, 0, 0}};
      uint32_t vb = par;
      for (uint32_t q : sx) { x = dev::lxor(x, load(lab(q))); vb ^= bit(q); }
      if (!evaluate) wr.push_back({sc, dev::lxor_if(x, d, par)});
      else { wr.push_back({sc, x}); wb.push_back({sc, uint8_t(vb & 1u)}); }
    }
    for (uint32_t k = sd.and_cnt; k-- > 0;) {
      const AndRec& r = g.ands[sd.and_off + k];
      // two record forms (program.hpp pack_and / pack_and4): up to two or up to four wires per AND input; absent operands name a zero label
      uint32_t sa[4], sb[4], sp, sc, ty;
      uint64_t gid;
      const uint32_t s3[9] = {uint32_t(r.w0) & SLOT_MASK, uint32_t(r.w0 >> 21) & SLOT_MASK, uint32_t(r.w0 >> 42) & SLOT_MASK, uint32_t(r.w1) & SLOT_MASK, uint32_t(r.w1 >> 21) & SLOT_MASK,
                              uint32_t(r.w1 >> 42) & SLOT_MASK, uint32_t(r.w2) & SLOT_MASK, uint32_t(r.w2 >> 21) & SLOT_MASK, uint32_t(r.w2 >> 42) & SLOT_MASK};
      const uint32_t zero = g.lds_slots_limit ? SLOT_LDS_ZERO : SLOT_ZERO;
      if (g.and_terms == 4) {
        for (int q = 0; q < 4; ++q) { sa[q] = s3[q]; sb[q] = s3[4 + q]; }
        sp = s3[8]; sc = uint32_t(r.w3) & SLOT_MASK;
        ty = uint32_t(r.w0 >> 63) | (uint32_t(r.w1 >> 63) << 1) | (uint32_t(r.w2 >> 63) << 2);
        gid = gb + ((r.w3 >> 21) & 0x7FFFFFFFull);
      } else {
        sa[0] = s3[0]; sa[1] = s3[1]; sb[0] = s3[2]; sb[1] = s3[3]; sa[2] = sa[3] = sb[2] = sb[3] = zero;
        sp = s3[4]; sc = s3[5];
        ty = uint32_t(r.w0 >> 63) | (uint32_t(r.w1 >> 63) << 1) | (uint32_t((r.w2 >> 40) & 1u) << 2);
        gid = gb + (r.w2 & 0xFFFFFFFFFFull);
      }
      const uint32_t cti = gate_of[sd.and_off + k];  // the ciphertext of record k sits at position k of the device stream
      Label a{{0, 0, 0, 0}}, b{{0, 0, 0, 0}};
      uint32_t va = 0, vb = 0;
      for (int q = 0; q < 4; ++q) { a = dev::lxor(a, load(lab(sa[q]))); b = dev::lxor(b, load(lab(sb[q]))); va ^= bit(sa[q]); vb ^= bit(sb[q]); }
      va &= 1u; vb &= 1u;
      const Label pl = load(lab(sp));
      const uint32_t vp = bit(sp) & 1u;
      if (!evaluate) {
        Label c0, c;
        if (g_hasher == 1) dev::garble_and_blake3(ty, a, b, d, gid, c0, c);
        else dev::garble_and(aes, ty, a, b, d, gid, c0, c);
        wr.push_back({sc, dev::lxor(c0, pl)});
        store(ct + size_t(cti) * 16, c);
      } else {
        Label c = load(ct + size_t(cti) * 16);
        const Label h = g_hasher == 1 ? dev::degarble_and_blake3(ty, c, a, va, b, gid) : dev::degarble_and(aes, ty, c, a, va, b, gid);
        wr.push_back({sc, dev::lxor(h, pl)});
        wb.push_back({sc, uint8_t((dev::gate_eval_bit(ty, va, vb) ^ vp) & 1u)});
      }
    }
    for (auto& x : wr) store(lab(x.first), x.second);
    for (auto& x : wb) bit(x.first) = x.second;
  }
}

// mode 0 = garble, 1 = evaluate.  Buffers as in include/gsv_engine.h (16-byte records).
int hostsim_run(SimProgram* sp, int evaluate, uint32_t replays, uint64_t gid_base, const uint8_t delta[16], const uint8_t consts[32],
                const uint8_t* inputs, const uint8_t* input_bits, uint8_t* cts /* in (evaluate) / out (garble): replays*n_ct*16 */,
                uint8_t* out_labels, uint8_t* out_bits) {
  try {
    const Program& g = sp->prog;
    std::vector<uint8_t> W(size_t(g.n_slots) * 16, 0xA5);  // poison: reading a never-written slot is visible
    std::vector<uint8_t> VB(g.n_slots, 0);
    auto lab = [&](uint32_t slot) -> uint8_t* { if (slot & SLOT_LDS_FLAG) gsv_panic("pinned wire in the LDS window"); return &W[size_t(slot) * 16]; };
    std::memcpy(&W[0], consts, 32);
    std::memset(&W[size_t(SLOT_ZERO) * 16], 0, 16);
    VB[0] = 0; VB[1] = 1; VB[SLOT_ZERO] = 0;
    for (size_t i = 0; i < g.input_slots.size(); ++i) {
      std::memcpy(&W[size_t(g.input_slots[i]) * 16], inputs + 16 * i, 16);
      if (evaluate) VB[g.input_slots[i]] = input_bits[i] ? 1 : 0;
    }
    Label d = evaluate ? Label{{0, 0, 0, 0}} : load(delta);
    for (uint32_t rep = 0; rep < replays; ++rep) {
      interpret(g, evaluate != 0, gid_base + uint64_t(rep) * g.n_gates, d, W, VB, cts + size_t(rep) * g.n_ct * 16);
      if (!g.fb_src_slot.empty()) {
        std::vector<uint8_t> tmp(g.fb_src_slot.size() * 16), tb(g.fb_src_slot.size());
        for (size_t i = 0; i < g.fb_src_slot.size(); ++i) { std::memcpy(&tmp[16 * i], lab(g.fb_src_slot[i]), 16); tb[i] = VB[g.fb_src_slot[i]]; }
        for (size_t i = 0; i < g.fb_dst_slot.size(); ++i) { std::memcpy(lab(g.fb_dst_slot[i]), &tmp[16 * i], 16); VB[g.fb_dst_slot[i]] = tb[i]; }
      }
    }
    for (size_t i = 0; i < g.output_slots.size(); ++i) {
      std::memcpy(out_labels + 16 * i, lab(g.output_slots[i]), 16);
      if (out_bits) out_bits[i] = VB[g.output_slots[i]];
    }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// ---- plans (plan_builder.hpp): a circuit recorded with some components as calls of separately compiled programs,
// interpreted call by call over a global wire array — the host-side twin of engine.cpp's launch_plan.
struct SimPlan {
  BuiltPlan bp;
  uint64_t n_ct = 0;
  Schedule sched;        // hostsim_plan_schedule: the engine's call-level schedule (schedule.hpp); empty = stream order
  bool scheduled = false;
};
int hostsim_plan_build(const char* spec, const char* units_csv, SimPlan** out, uint64_t* info /* 8: n_inputs n_outputs n_gates n_ct n_calls n_programs n_globals n_unit_programs */) {
  try {
    std::vector<std::string> names;
    std::string cur;
    for (const char* q = units_csv;; ++q) {
      if (*q == ',' || *q == 0) { if (!cur.empty()) names.push_back(cur); cur.clear(); if (!*q) break; }
      else cur.push_back(*q);
    }
    NamedCircuit nc = make_circuit(spec);
    PlanRecordMode mode(names);
    CompileOptions opt;
    if (const char* e = getenv("GSV_FUSE")) opt.fuse = atoi(e) != 0;
    if (getenv("HOSTSIM_PLAN_BACKGROUND")) mode.compile_in_background(opt, false);  // the engine's way: units compiled on the pool while recording goes on
    std::vector<uint32_t> in_ssa, out_ssa;
    record_plan(mode, nc.n_inputs, nc.fn, nc.warmups, in_ssa, out_ssa);  // as gsv_plan_from_circuit: warm-up recorders beside the driver
    mode.wait_for_compilations();
    auto sp = std::make_unique<SimPlan>();
    const size_t n_units = mode.units.size();
    sp->bp = finish_plan(mode, in_ssa, out_ssa, opt);
    if (sp->bp.n_gates != mode.n_gates()) gsv_panic("plan gate count differs from the recorded stream");
    uint32_t n_globals = sp->bp.n_inputs;
    for (auto& c : sp->bp.calls) {
      sp->n_ct += sp->bp.programs[size_t(c.program)].n_ct;
      for (uint32_t w : c.in_globals) if (w < PLAN_WIRE_FALSE) n_globals = std::max(n_globals, w + 1);
      for (uint32_t w : c.out_globals) n_globals = std::max(n_globals, w + 1);
    }
    if (info) {
      info[0] = sp->bp.n_inputs; info[1] = sp->bp.outputs.size(); info[2] = sp->bp.n_gates; info[3] = sp->n_ct; info[4] = sp->bp.calls.size();
      info[5] = sp->bp.programs.size(); info[6] = n_globals; info[7] = n_units;
    }
    *out = sp.release();
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void hostsim_plan_free(SimPlan* p) { delete p; }
static uint32_t plan_n_globals(const BuiltPlan& bp) {
  uint32_t n_globals = bp.n_inputs;
  for (auto& c : bp.calls) {
    for (uint32_t w : c.in_globals) if (w < PLAN_WIRE_FALSE) n_globals = std::max(n_globals, w + 1);
    for (uint32_t w : c.out_globals) n_globals = std::max(n_globals, w + 1);
  }
  return n_globals;
}
// The engine's scheduler (schedule.hpp, the very code engine.cpp runs at session creation) over this plan; the schedule is checked by
// brute force (every hazard and every scratch overlap covered by a dependency path) and kept: hostsim_plan_run then executes it the
// way the device may — window by window, inside a window ANY order the dependencies allow: here always the ready call with the
// LARGEST stream index, all calls in one shared scratch ring.  info: n_windows n_dependencies max_width scratch_slots
// critical_steps total_steps max_window_ct.
static uint64_t g_segment_ct = 0, g_ring_ct = 0;  // hostsim_set_segment_ct / _ring_ct: SchedParams::segment_ct / ring_ct of the next hostsim_plan_schedule
void hostsim_set_segment_ct(uint64_t v) { g_segment_ct = v; }
void hostsim_set_ring_ct(uint64_t v) { g_ring_ct = v; }
// ring tables of the last schedule: per call {ring_off, ring_need, seg_end, ovl0, ovl1}
uint64_t hostsim_plan_ring(SimPlan* sp, uint64_t* out, uint64_t cap) {
  const Schedule& sc = sp->sched;
  if (!sc.ring_ct) return 0;
  for (size_t k = 0; k < sc.ring_off.size() && out && k < cap; ++k) { out[5 * k] = sc.ring_off[k]; out[5 * k + 1] = sc.ring_need[k]; out[5 * k + 2] = sc.seg_end[k]; out[5 * k + 3] = sc.ovl0[k]; out[5 * k + 4] = sc.ovl1[k]; }
  return sc.ring_off.size();
}
// The drain segments of the last schedule: per segment {window, call0, call1, ct0, n_ct}; returns their number (out may be NULL).
uint64_t hostsim_plan_segments(SimPlan* sp, uint64_t* out, uint64_t cap) {
  const Schedule& sc = sp->sched;
  uint64_t n = 0;
  for (size_t w = 0; w < sc.windows.size(); ++w)
    for (uint32_t q = sc.windows[w].seg0; q < sc.windows[w].seg1; ++q, ++n)
      if (out && n < cap) { out[5 * n] = w; out[5 * n + 1] = sc.segments[q].call0; out[5 * n + 2] = sc.segments[q].call1; out[5 * n + 3] = sc.segments[q].ct0; out[5 * n + 4] = sc.segments[q].n_ct; }
  return n;
}
int hostsim_plan_schedule(SimPlan* sp, uint32_t max_calls, uint64_t max_slots, uint64_t window_ct, uint32_t window_calls, uint64_t* info /* 7 */) {
  try {
    const BuiltPlan& bp = sp->bp;
    std::vector<SchedCall> calls(bp.calls.size());
    for (size_t k = 0; k < bp.calls.size(); ++k) {
      const BuiltPlan::Call& c = bp.calls[k];
      const Program& g = bp.programs[size_t(c.program)];
      calls[k].in = c.in_globals.data(); calls[k].n_in = c.in_globals.size();
      calls[k].out = c.out_globals.data(); calls[k].n_out = c.out_globals.size();
      calls[k].n_slots = g.n_slots; calls[k].n_ct = g.n_ct; calls[k].n_steps = g.n_steps;
    }
    SchedParams p;
    p.max_calls_in_flight = max_calls; p.max_scratch_slots = max_slots ? max_slots : ~0ull; p.max_window_ct = window_ct ? window_ct : ~0ull;
    p.max_window_calls = window_calls ? window_calls : 32768;
    p.segment_ct = g_segment_ct;
    p.ring_ct = g_ring_ct;
    const uint32_t n_ids = plan_n_globals(bp);
    sp->sched = schedule_calls(calls, n_ids, bp.outputs, p);
    const std::string err = verify_schedule(calls, n_ids, bp.outputs, sp->sched);
    if (!err.empty()) gsv_panic("schedule violates a hazard: " + err);
    sp->scheduled = true;
    if (info) {
      info[0] = sp->sched.windows.size(); info[1] = sp->sched.deps.size(); info[2] = sp->sched.max_width;
      info[3] = sp->sched.scratch_slots; info[4] = sp->sched.critical_steps; info[5] = sp->sched.total_steps; info[6] = sp->sched.max_window_ct;
    }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
static int plan_run_scheduled(SimPlan* sp, int evaluate, uint64_t gid_base, const uint8_t delta[16], const uint8_t consts[32], const uint8_t* inputs,
                              const uint8_t* input_bits, uint8_t* cts, uint8_t* out_labels, uint8_t* out_bits) {
  const BuiltPlan& bp = sp->bp;
  const Schedule& sc = sp->sched;
  const uint32_t n_globals = plan_n_globals(bp);
  std::vector<uint8_t> G(size_t(n_globals) * 16, 0xC3), GB(n_globals, 0);
  for (uint32_t i = 0; i < bp.n_inputs; ++i) { std::memcpy(&G[size_t(i) * 16], inputs + 16 * i, 16); if (evaluate) GB[i] = input_bits[i] ? 1 : 0; }
  const Label d = evaluate ? Label{{0, 0, 0, 0}} : load(delta);
  std::vector<uint64_t> gid_off(bp.calls.size()), ct_off(bp.calls.size());
  { uint64_t g = 0, c = 0; for (size_t k = 0; k < bp.calls.size(); ++k) { gid_off[k] = g; ct_off[k] = c; const Program& pr = bp.programs[size_t(bp.calls[k].program)]; g += pr.n_gates; c += pr.n_ct; } }
  std::vector<uint8_t> W(size_t(std::max<uint64_t>(sc.scratch_slots, 8)) * 16, 0xA5), VB(std::max<uint64_t>(sc.scratch_slots, 8), 0);
  for (const Schedule::Window& win : sc.windows) {
    const size_t m = win.call1 - win.call0;
    std::vector<uint8_t> done(m, 0);
    for (size_t executed = 0; executed < m; ++executed) {
      // the ready call with the largest stream index: as far from the stream order as the dependencies allow
      size_t pick = m;
      for (size_t kk = m; kk-- > 0;) {
        if (done[kk]) continue;
        bool ready = true;
        for (uint32_t q = sc.dep_off[win.call0 + kk]; q < sc.dep_off[win.call0 + kk + 1] && ready; ++q) ready = done[sc.deps[q] - win.call0] != 0;
        if (ready) { pick = kk; break; }
      }
      if (pick == m) gsv_panic("schedule deadlocks");
      const uint32_t k = uint32_t(win.call0 + pick);
      const BuiltPlan::Call& c = bp.calls[k];
      const Program& g = bp.programs[size_t(c.program)];
      const size_t base = sc.scratch_base[k];
      std::memset(&W[base * 16], 0xA5, size_t(g.n_slots) * 16);  // poison: whatever an earlier occupant of the region left behind
      std::memcpy(&W[base * 16], consts, 32);
      std::memset(&W[(base + SLOT_ZERO) * 16], 0, 16);
      VB[base + 0] = 0; VB[base + 1] = 1; VB[base + SLOT_ZERO] = 0;
      for (size_t i = 0; i < c.in_globals.size(); ++i) {
        const uint32_t w = c.in_globals[i];
        const size_t dst = base + g.input_slots[i];
        if (w == PLAN_WIRE_FALSE) { std::memcpy(&W[dst * 16], consts, 16); VB[dst] = 0; }
        else if (w == PLAN_WIRE_TRUE) { std::memcpy(&W[dst * 16], consts + 16, 16); VB[dst] = 1; }
        else { std::memcpy(&W[dst * 16], &G[size_t(w) * 16], 16); VB[dst] = GB[w]; }
      }
      std::vector<uint8_t> w(W.begin() + base * 16, W.begin() + (base + g.n_slots) * 16), vb(VB.begin() + base, VB.begin() + base + g.n_slots);
      interpret(g, evaluate != 0, gid_base + gid_off[k], d, w, vb, cts + ct_off[k] * 16);
      std::memcpy(&W[base * 16], w.data(), w.size());
      std::memcpy(&VB[base], vb.data(), vb.size());
      for (size_t i = 0; i < c.out_globals.size(); ++i) {
        const uint32_t src = g.output_slots[i];
        if (src & SLOT_LDS_FLAG) gsv_panic("program output in the LDS window");
        std::memcpy(&G[size_t(c.out_globals[i]) * 16], &W[(base + src) * 16], 16);
        GB[c.out_globals[i]] = VB[base + src];
      }
      done[pick] = 1;
    }
  }
  for (size_t i = 0; i < bp.outputs.size(); ++i) {
    const uint32_t w = bp.outputs[i];
    if (w == PLAN_WIRE_FALSE) { std::memcpy(out_labels + 16 * i, consts, 16); if (out_bits) out_bits[i] = 0; }
    else if (w == PLAN_WIRE_TRUE) { std::memcpy(out_labels + 16 * i, consts + 16, 16); if (out_bits) out_bits[i] = 1; }
    else { std::memcpy(out_labels + 16 * i, &G[size_t(w) * 16], 16); if (out_bits) out_bits[i] = GB[w]; }
  }
  return 0;
}
int hostsim_plan_run(SimPlan* sp, int evaluate, uint64_t gid_base, const uint8_t delta[16], const uint8_t consts[32], const uint8_t* inputs,
                     const uint8_t* input_bits, uint8_t* cts, uint8_t* out_labels, uint8_t* out_bits) {
  try {
    if (sp->scheduled) return plan_run_scheduled(sp, evaluate, gid_base, delta, consts, inputs, input_bits, cts, out_labels, out_bits);
    const BuiltPlan& bp = sp->bp;
    const uint32_t n_globals = plan_n_globals(bp);
    std::vector<uint8_t> G(size_t(n_globals) * 16, 0xC3), GB(n_globals, 0);
    for (uint32_t i = 0; i < bp.n_inputs; ++i) { std::memcpy(&G[size_t(i) * 16], inputs + 16 * i, 16); if (evaluate) GB[i] = input_bits[i] ? 1 : 0; }
    const Label d = evaluate ? Label{{0, 0, 0, 0}} : load(delta);
    uint64_t gid = gid_base, ct_off = 0;
    for (const BuiltPlan::Call& c : bp.calls) {
      const Program& g = bp.programs[size_t(c.program)];
      std::vector<uint8_t> W(size_t(g.n_slots) * 16, 0xA5), VB(g.n_slots, 0);
      std::memcpy(&W[0], consts, 32);
      std::memset(&W[size_t(SLOT_ZERO) * 16], 0, 16);
      VB[0] = 0; VB[1] = 1; VB[SLOT_ZERO] = 0;
      for (size_t i = 0; i < c.in_globals.size(); ++i) {
        const uint32_t w = c.in_globals[i], dst = g.input_slots[i];
        if (w == PLAN_WIRE_FALSE) { std::memcpy(&W[size_t(dst) * 16], &W[0], 16); VB[dst] = 0; }
        else if (w == PLAN_WIRE_TRUE) { std::memcpy(&W[size_t(dst) * 16], &W[16], 16); VB[dst] = 1; }
        else { std::memcpy(&W[size_t(dst) * 16], &G[size_t(w) * 16], 16); VB[dst] = GB[w]; }
      }
      interpret(g, evaluate != 0, gid, d, W, VB, cts + ct_off * 16);
      for (size_t i = 0; i < c.out_globals.size(); ++i) {
        const uint32_t src = g.output_slots[i];
        if (src & SLOT_LDS_FLAG) gsv_panic("program output in the LDS window");
        std::memcpy(&G[size_t(c.out_globals[i]) * 16], &W[size_t(src) * 16], 16);
        GB[c.out_globals[i]] = VB[src];
      }
      gid += g.n_gates;
      ct_off += g.n_ct;
    }
    for (size_t i = 0; i < bp.outputs.size(); ++i) {
      const uint32_t w = bp.outputs[i];
      if (w == PLAN_WIRE_FALSE) { std::memcpy(out_labels + 16 * i, consts, 16); if (out_bits) out_bits[i] = 0; }
      else if (w == PLAN_WIRE_TRUE) { std::memcpy(out_labels + 16 * i, consts + 16, 16); if (out_bits) out_bits[i] = 1; }
      else { std::memcpy(out_labels + 16 * i, &G[size_t(w) * 16], 16); if (out_bits) out_bits[i] = GB[w]; }
    }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// raw trace export (type, a, b, c as SSA ids; c = 0xFFFFFFFF for dead gates) for schedule experiments
int hostsim_trace(const char* spec, uint64_t cap, uint8_t* type, uint32_t* a, uint32_t* b, uint32_t* c, uint64_t* n_out, uint32_t* n_wires,
                  uint32_t* in_ssa, uint32_t* out_ssa) {
  try {
    NamedCircuit nc = make_circuit(spec);
    RecordMode mode;
    StreamingRunner run(mode, nc.n_inputs, nc.fn);
    size_t k = 0;
    for (WireId w : run.prepare()) in_ssa[k++] = mode.define_input(w);
    k = 0;
    for (WireId w : run.execute()) out_ssa[k++] = mode.current(w);
    const Trace& t = mode.trace();
    *n_out = t.size(); *n_wires = t.n_wires;
    if (t.size() > cap) return 2;
    std::memcpy(type, t.type.data(), t.size()); std::memcpy(a, t.a.data(), 4 * t.size());
    std::memcpy(b, t.b.data(), 4 * t.size()); std::memcpy(c, t.c.data(), 4 * t.size());
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

void hostsim_set_hasher(int k) { g_hasher = k == 1 ? 1 : 0; }
void hostsim_blake3_hash(const uint8_t label[16], uint64_t gid, uint8_t out[16]) { store(out, dev::blake3_hash_with_gate(load(label), gid)); }

// host crypto of the product, for known-answer tests
void hostsim_labels_from_seed(uint64_t seed, uint64_t n, uint8_t* out) {
  ChaCha20Seed r(seed);
  for (uint64_t i = 0; i < n; ++i) r.next_label(out + 16 * i);
}
void hostsim_cbcmac(const uint8_t* cts, uint64_t n, uint8_t out[16]) {
  CbcMacHost m;
  m.update(cts, n);
  m.digest(out);
}
void hostsim_aes_ttable(const uint8_t in[16], uint8_t out[16]) {
  const AesTables& T = AesTables::fixed_key();
  dev::PlainTables aes{{T.te[0], T.te[1], T.te[2], T.te[3]}, T.rk};
  Label o = dev::aes128_encrypt(aes, load(in));
  store(out, o);
}
void hostsim_aes_portable(const uint8_t in[16], uint8_t out[16]) { CbcMacHost::encrypt_portable(AesTables::fixed_key(), in, out); }
void hostsim_hash(const uint8_t label[16], uint64_t gid, uint8_t out[16]) {
  const AesTables& T = AesTables::fixed_key();
  dev::PlainTables aes{{T.te[0], T.te[1], T.te[2], T.te[3]}, T.rk};
  store(out, dev::hash_with_gate(aes, load(label), gid));
}
void hostsim_sbox(uint8_t out[256]) { std::memcpy(out, AesTables::fixed_key().sbox, 256); }

}  // extern "C"
