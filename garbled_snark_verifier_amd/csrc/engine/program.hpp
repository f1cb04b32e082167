// Gate recorder (the engine's `impl CircuitMode`) and gate-program compiler.
//
// RecordMode sits under the same driver seam as the reference's GarbleMode / EvaluateMode
// (src/circuit/modes.rs:26-51): `allocate_wire` follows Storage::allocate's rule that zero credits
// yield UNREACHABLE (src/storage.rs:119-133) and `evaluate_gate` consumes a gate_id for EVERY call,
// dead or not (src/circuit/modes/garble_mode.rs:192-197).  Instead of garbling immediately it records
// the stream in SSA form.  Labels never depend on WireIds or slot numbers, only on
// gate order, gate type, dataflow and dead-gate decisions, which is what the trace keeps.
//
// compile_program() turns a trace into the device schedule:
//   * gates are levelised ASAP over the dataflow DAG; one level = one device step holding that
//     level's AND-family gates (AES work) and free gates (XOR work) as two homogeneous record runs;
//   * wire slots are assigned by a linear scan over steps (a slot is recycled one step after its
//     last reader), which keeps the live window L2-resident instead of one slot per wire.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../circuit/circuit.hpp"

namespace gsv {

constexpr uint32_t DEAD_WIRE = 0xFFFFFFFFu;

struct Trace {
  std::vector<uint8_t> type;
  std::vector<uint32_t> a, b, c;  // SSA wire ids; c == DEAD_WIRE for a gate whose output has zero fan-out
  uint32_t n_wires = 2;           // ids 0 / 1 are the FALSE / TRUE constants
  size_t size() const { return type.size(); }
};

class RecordMode final : public CircuitMode {
 public:
  RecordMode() { ver_.assign(2, 0); ver_[1] = 1; written_.assign(2, 1); }

  WireId allocate_wire(Credits credits) override {
    if (credits == 0) return UNREACHABLE;
    WireId id = ver_.size();
    if (id >= 0xFFFFFFF0ull) gsv_panic("RecordMode: more than 2^32 wires in one program");
    ver_.push_back(0);
    written_.push_back(0);
    return id;
  }
  void evaluate_gate(const Gate& g) override {
    uint32_t ra = read(g.a), rb = read(g.b);
    trace_.type.push_back(uint8_t(g.t));
    trace_.a.push_back(ra);
    trace_.b.push_back(rb);
    if (g.c == UNREACHABLE) { trace_.c.push_back(DEAD_WIRE); return; }
    if (g.c == FALSE_WIRE || g.c == TRUE_WIRE) gsv_panic("gate output is a constant wire");
    trace_.c.push_back(define(g.c));
  }
  bool consume_wire(WireId w) override { return w < ver_.size() && (w < 2 || written_[size_t(w)]); }
  void add_credits(const WireId*, size_t, Credits) override {}

  // Root inputs: "feed" marks the wire as defined by the host (EncodeInput::encode).
  uint32_t define_input(WireId w) {
    if (w == UNREACHABLE) gsv_panic("input wire has zero fan-out and no root credit");
    return define(w);
  }
  uint32_t current(WireId w) { return read(w); }
  Trace& trace() { return trace_; }

 private:
  uint32_t read(WireId w) {
    if (w == FALSE_WIRE) return 0;
    if (w == TRUE_WIRE) return 1;
    if (w >= ver_.size()) gsv_panic("RecordMode: read of unknown wire");
    if (!written_[size_t(w)]) gsv_panic("RecordMode: wire read before it was written");
    return ver_[size_t(w)];
  }
  uint32_t define(WireId w) {  // fresh SSA id on every write (in-place gates such as Gate::not re-version the wire)
    if (w >= ver_.size() || w < 2) gsv_panic("RecordMode: write to unknown wire");
    uint32_t id = trace_.n_wires++;
    if (id >= 0xFFFFFFF0u) gsv_panic("RecordMode: SSA id overflow");
    ver_[size_t(w)] = id;
    written_[size_t(w)] = 1;
    return id;
  }
  Trace trace_;
  std::vector<uint32_t> ver_;
  std::vector<uint8_t> written_;
};

// ---- device-facing program format (all little-endian u32; uploaded verbatim) ----------------------
struct XorRec { uint32_t a, b, c, type; };                       // 16 B: slots + GateType (8, 9, 10)
struct AndRec { uint32_t a, b, c, type, gid, ct, pad0, pad1; };  // 32 B: + gate_id and ciphertext index, both
                                                                 //       relative to the replay's bases
struct StepDesc { uint32_t and_off, and_cnt, xor_off, xor_cnt; };  // one dependency level: AND-family + free gates

struct Program {
  std::vector<StepDesc> steps;
  std::vector<AndRec> ands;
  std::vector<XorRec> xors;
  std::vector<uint32_t> input_slots, output_slots;
  uint32_t n_slots = 0;        // wire-file entries per instance (incl. constants, inputs, feedback staging)
  uint32_t fb_stage_base = 0;  // first staging slot for feedback copies
  std::vector<uint32_t> fb_src_slot, fb_dst_slot;  // replay epilogue: W[dst] <- W[src]
  uint64_t n_gates = 0;        // gates in stream order INCLUDING dead ones (= gate_ids consumed per replay)
  uint64_t n_ct = 0;           // ciphertexts per replay (AND-family, live)
  uint64_t n_dead = 0;
  uint64_t gate_count[GATE_TYPE_COUNT] = {0};
  uint32_t and_depth = 0, n_and_steps = 0, max_step_width = 0;
  uint32_t peak_live = 0;
};

// inputs / outputs: SSA ids of the circuit's input and output wires.
// feedback: pairs (output index -> input index) copied at the end of every replay (chained circuits).
inline Program compile_program(const Trace& t, const std::vector<uint32_t>& inputs, const std::vector<uint32_t>& outputs,
                               const std::vector<std::pair<uint32_t, uint32_t>>& feedback = {}) {
  const size_t n = t.size();
  const uint32_t nw = t.n_wires;
  Program p;
  p.n_gates = n;
  if (n >= 0xFFFFFFFFull) gsv_panic("program too large: gate index must fit 32 bits per replay");

  // 1. ASAP dependency level per wire (inputs / constants = 0) and AND-depth (statistic).
  std::vector<uint32_t> lev(nw, 0), ad(nw, 0);
  uint32_t n_steps = 0;
  for (size_t i = 0; i < n; ++i) {
    p.gate_count[t.type[i]]++;
    uint32_t c = t.c[i];
    if (c == DEAD_WIRE) { p.n_dead++; continue; }
    uint32_t a = t.a[i], b = t.b[i];
    lev[c] = std::max(lev[a], lev[b]) + 1;
    ad[c] = std::max(ad[a], ad[b]) + (t.type[i] < 8 ? 1 : 0);
    n_steps = std::max(n_steps, lev[c]);
    p.and_depth = std::max(p.and_depth, ad[c]);
  }
  auto step_of = [&](size_t i) -> uint32_t { return lev[t.c[i]] - 1; };
  // 2. counting sort of live gates by (step, kind): AND-family first, then free gates
  std::vector<uint32_t> cnt(2 * size_t(n_steps) + 1, 0);
  for (size_t i = 0; i < n; ++i) if (t.c[i] != DEAD_WIRE) cnt[2 * size_t(step_of(i)) + (t.type[i] < 8 ? 0 : 1) + 1]++;
  for (size_t k = 0; k < 2 * size_t(n_steps); ++k) cnt[k + 1] += cnt[k];
  const size_t n_live = cnt[2 * size_t(n_steps)];
  std::vector<uint32_t> order(n_live);
  {
    std::vector<uint32_t> cursor(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < n; ++i) if (t.c[i] != DEAD_WIRE) order[cursor[2 * size_t(step_of(i)) + (t.type[i] < 8 ? 0 : 1)]++] = uint32_t(i);
  }
  // 3. last reader step per wire (live gates only).  NEVER = pinned, UNUSED = no live reader.
  constexpr uint32_t NEVER = 0xFFFFFFFFu, UNUSED = 0xFFFFFFFEu;
  std::vector<uint32_t> last_use(nw, UNUSED);
  for (size_t i = 0; i < n; ++i) {
    if (t.c[i] == DEAD_WIRE) continue;
    uint32_t s = step_of(i);
    for (uint32_t w : {t.a[i], t.b[i]}) if (last_use[w] == UNUSED || last_use[w] < s) last_use[w] = s;
  }
  last_use[0] = last_use[1] = NEVER;
  for (uint32_t w : inputs) last_use[w] = NEVER;
  for (uint32_t w : outputs) last_use[w] = NEVER;
  // 4. slots: constants 0,1; inputs 2..; then a linear scan over steps
  std::vector<uint32_t> slot(nw, DEAD_WIRE);
  slot[0] = 0; slot[1] = 1;
  uint32_t next_slot = 2;
  for (uint32_t w : inputs) { if (slot[w] == DEAD_WIRE) slot[w] = next_slot++; }
  std::vector<uint32_t> free_stack;
  auto dies_at = [&](uint32_t w, uint32_t def_step) -> uint32_t { return last_use[w] == UNUSED ? def_step : last_use[w]; };
  std::vector<uint32_t> die_cnt(size_t(n_steps) + 1, 0);
  for (size_t k = 0; k < n_live; ++k) {
    size_t i = order[k];
    uint32_t c = t.c[i];
    if (last_use[c] == NEVER) continue;
    die_cnt[dies_at(c, step_of(i)) + 1]++;
  }
  for (uint32_t s = 0; s < n_steps; ++s) die_cnt[s + 1] += die_cnt[s];
  std::vector<uint32_t> die_list(die_cnt[n_steps]);
  {
    std::vector<uint32_t> cur(die_cnt.begin(), die_cnt.end() - 1);
    for (size_t k = 0; k < n_live; ++k) {
      size_t i = order[k];
      uint32_t c = t.c[i];
      if (last_use[c] == NEVER) continue;
      die_list[cur[dies_at(c, step_of(i))]++] = c;
    }
  }
  // ciphertext index: prefix count of live AND-family gates in STREAM order
  std::vector<uint32_t> ct_index(n);
  {
    uint64_t k = 0;
    for (size_t i = 0; i < n; ++i) {
      ct_index[i] = uint32_t(k);
      if (t.c[i] != DEAD_WIRE && t.type[i] < 8) ++k;
    }
    p.n_ct = k;
  }
  uint32_t live = next_slot, peak = next_slot;
  p.steps.reserve(n_steps);
  for (uint32_t s = 0; s < n_steps; ++s) {
    StepDesc sd{uint32_t(p.ands.size()), cnt[2 * size_t(s) + 1] - cnt[2 * size_t(s)], uint32_t(p.xors.size()),
                cnt[2 * size_t(s) + 2] - cnt[2 * size_t(s) + 1]};
    for (uint32_t k = cnt[2 * size_t(s)]; k < cnt[2 * size_t(s) + 2]; ++k) {
      size_t i = order[k];
      uint32_t c = t.c[i];
      uint32_t sl;
      if (!free_stack.empty()) { sl = free_stack.back(); free_stack.pop_back(); }
      else sl = next_slot++;
      slot[c] = sl;
      ++live;
      uint32_t sa = slot[t.a[i]], sb = slot[t.b[i]];
      if (sa == DEAD_WIRE || sb == DEAD_WIRE) gsv_panic("internal: operand without slot");
      if (t.type[i] < 8) p.ands.push_back(AndRec{sa, sb, sl, t.type[i], uint32_t(i), ct_index[i], 0, 0});
      else p.xors.push_back(XorRec{sa, sb, sl, t.type[i]});
    }
    peak = std::max(peak, live);
    for (uint32_t k = die_cnt[s]; k < die_cnt[s + 1]; ++k) { free_stack.push_back(slot[die_list[k]]); --live; }
    p.steps.push_back(sd);
    p.max_step_width = std::max(p.max_step_width, sd.and_cnt + sd.xor_cnt);
    if (sd.and_cnt) p.n_and_steps++;
  }
  p.peak_live = peak;
  for (uint32_t w : inputs) p.input_slots.push_back(slot[w]);
  for (uint32_t w : outputs) {
    if (slot[w] == DEAD_WIRE) gsv_panic("output wire was never produced");
    p.output_slots.push_back(slot[w]);
  }
  p.fb_stage_base = next_slot;
  for (auto& fb : feedback) {
    if (fb.first >= outputs.size() || fb.second >= inputs.size()) gsv_panic("feedback index out of range");
    p.fb_src_slot.push_back(p.output_slots[fb.first]);
    p.fb_dst_slot.push_back(p.input_slots[fb.second]);
  }
  p.n_slots = next_slot + uint32_t(feedback.size());
  return p;
}

}  // namespace gsv
