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
//   * wire slots are assigned by a scan over steps (a slot is recycled one step after its last
//     reader) from two next-fit pools: short-lived wires (the large majority: a ripple-carry bit is
//     read one or two steps after it is produced) live in the workgroup's LDS label window, the rest
//     in the instance's HBM wire file.  This is the "active-wire window sized by the fan-out pass".
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../circuit/circuit.hpp"
#include "limits.h"

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

// ---- device-facing program format (little-endian; uploaded verbatim) -----------------------------
// A wire location ("slot") is 21 bits: bit 20 set = entry of the workgroup's LDS label window,
// clear = entry of the instance's wire file in HBM.  HBM slots 0/1/2 hold the FALSE constant, the TRUE
// constant and the all-zero label (lets the in-place NOT gate be encoded as XNOR(a, ZERO)).
constexpr uint32_t SLOT_BITS = 21;
constexpr uint32_t SLOT_LDS_FLAG = 1u << 20;
constexpr uint32_t SLOT_INDEX_MASK = SLOT_LDS_FLAG - 1;
constexpr uint32_t SLOT_MASK = (1u << SLOT_BITS) - 1;
constexpr uint32_t SLOT_FALSE = 0, SLOT_TRUE = 1, SLOT_ZERO = 2, SLOT_FIRST_INPUT = 3;
constexpr uint32_t LDS_WINDOW_SLOTS = GSV_LDS_SLOTS;  // 90 KiB of the CU's 160 KiB (64 KiB go to the banked AES tables)

// Free gate, 8 bytes:   bits 0..20 a | 21..41 b | 42..62 c | 63 xnor
struct XorRec { uint64_t v; };
inline XorRec pack_xor(uint32_t a, uint32_t b, uint32_t c, bool xnor) {
  return XorRec{uint64_t(a) | (uint64_t(b) << 21) | (uint64_t(c) << 42) | (uint64_t(xnor ? 1 : 0) << 63)};
}
// AND-family gate, 16 bytes: lo = a | b<<21 | c<<42 | (type&1)<<63 ; hi = type>>1 (2 bits) | gid<<2 (31 bits) | ct<<33 (31 bits)
// gid is relative to the replay's gate-id base; ct is the record's own index in the program (= where its ciphertext
// sits inside the replay's block of the device stream).
struct AndRec { uint64_t lo, hi; };
inline AndRec pack_and(uint32_t a, uint32_t b, uint32_t c, uint32_t type, uint32_t gid, uint32_t ct) {
  return AndRec{uint64_t(a) | (uint64_t(b) << 21) | (uint64_t(c) << 42) | (uint64_t(type & 1u) << 63),
                uint64_t(type >> 1) | (uint64_t(gid) << 2) | (uint64_t(ct) << 33)};
}
struct StepDesc { uint32_t and_off, and_cnt, xor_off, xor_cnt; };  // one dependency level: AND-family + free gates

struct Program {
  std::vector<StepDesc> steps;
  std::vector<AndRec> ands;
  std::vector<XorRec> xors;
  std::vector<uint32_t> input_slots, output_slots;
  uint32_t n_slots = 0;        // HBM wire-file entries per instance (constants, inputs, long-lived wires, feedback staging)
  uint32_t n_lds_slots = 0;    // LDS window entries used
  uint32_t lds_slots_limit = 0; // window size the program was compiled for (decides how many instances share a workgroup)
  uint32_t fb_stage_base = 0;  // first staging slot for feedback copies
  std::vector<uint32_t> fb_src_slot, fb_dst_slot;  // replay epilogue: W[dst] <- W[src]
  uint64_t n_gates = 0;        // gates in stream order INCLUDING dead ones (= gate_ids consumed per replay)
  uint64_t n_ct = 0;           // ciphertexts per replay (AND-family, live)
  std::vector<uint32_t> ct_pos;  // gate-order ciphertext index -> position inside a replay's block of the device stream
  uint64_t n_dead = 0;
  uint64_t gate_count[GATE_TYPE_COUNT] = {0};
  uint32_t and_depth = 0, n_and_steps = 0, max_step_width = 0;
  uint32_t peak_live = 0;
  uint64_t reads_lds = 0, reads_hbm = 0, writes_lds = 0, writes_hbm = 0;  // label accesses per replay by location
};

struct CompileOptions {
  uint32_t lds_slots = LDS_WINDOW_SLOTS;  // 0 = keep every wire in HBM
  uint32_t lds_max_lifetime = 8;          // a wire goes to the LDS window only if it dies within this many steps
  uint32_t schedule = 0;                  // 0 = ASAP levels, 1 = ALAP levels (experiment)
  bool order_by_reader = true;            // order the gates of a step by the position of their output's first reader
  uint32_t hbm_arena_factor = 4;          // HBM wire file = factor x peak live wires (next-fit then sweeps mostly free space)
};

// next-fit slot pool over a bitmap: consecutive allocations get ascending (mostly consecutive) slots, so the
// stores of one step coalesce and the operands of neighbouring gates sit in neighbouring lines.
class SlotPool {
 public:
  explicit SlotPool(uint32_t fixed_capacity = 0) : fixed_(fixed_capacity != 0) { if (fixed_) used_.assign(fixed_capacity, 0); }
  // returns DEAD_WIRE when a fixed pool is full
  uint32_t alloc() {
    if (n_free_ == 0) {
      if (fixed_ && high_ >= used_.size()) return DEAD_WIRE;
      if (!fixed_ && high_ >= used_.size()) used_.resize(std::max<size_t>(1024, used_.size() * 2), 0);
      used_[high_] = 1;
      cursor_ = high_ + 1;
      return high_++;
    }
    uint32_t i = cursor_ < high_ ? cursor_ : 0;
    while (used_[i]) { if (++i >= high_) i = 0; }
    used_[i] = 1; --n_free_;
    cursor_ = i + 1;
    return i;
  }
  void release(uint32_t i) { used_[i] = 0; ++n_free_; }
  void reserve_low(uint32_t n) { if (used_.size() < n) used_.resize(n, 0); for (uint32_t i = high_; i < n; ++i) used_[i] = 1; high_ = std::max(high_, n); cursor_ = high_; }
  // growable pools: make [high, n) allocatable right away (a roomy arena keeps next-fit allocations contiguous)
  void preextend(uint32_t n) { if (n > high_) { if (used_.size() < n) used_.resize(n, 0); n_free_ += n - high_; high_ = n; } }
  uint32_t high() const { return high_; }
  // slots that alloc() can still hand out right now (fixed pools only)
  uint32_t available() const { return n_free_ + (fixed_ ? uint32_t(used_.size()) - high_ : 0u); }
 private:
  std::vector<uint8_t> used_;
  uint32_t high_ = 0, cursor_ = 0, n_free_ = 0;
  bool fixed_;
};

// inputs / outputs: SSA ids of the circuit's input and output wires.
// feedback: pairs (output index -> input index) copied at the end of every replay (chained circuits).
inline Program compile_program(const Trace& t, const std::vector<uint32_t>& inputs, const std::vector<uint32_t>& outputs,
                               const std::vector<std::pair<uint32_t, uint32_t>>& feedback = {}, const CompileOptions& opt = CompileOptions()) {
  const size_t n = t.size();
  const uint32_t nw = t.n_wires;
  Program p;
  p.n_gates = n;
  if (n >= 0x7FFFFFFFull) gsv_panic("program too large: gate index must fit 31 bits per replay");
  if (opt.lds_slots > SLOT_INDEX_MASK) gsv_panic("LDS window larger than the slot encoding");

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
  if (opt.schedule == 1) {
    // ALAP: a gate runs one step before its earliest reader (circuit outputs: the last step); depth unchanged
    std::vector<uint32_t> need(nw, n_steps + 1);
    for (size_t i = n; i-- > 0;) {
      uint32_t c = t.c[i];
      if (c == DEAD_WIRE) continue;
      uint32_t al = need[c] - 1;  // level in 1..n_steps
      if (al < lev[c]) gsv_panic("internal: ALAP below ASAP");
      lev[c] = al;
      need[t.a[i]] = std::min(need[t.a[i]], al);
      if (t.type[i] != uint8_t(GateType::Not)) need[t.b[i]] = std::min(need[t.b[i]], al);
    }
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
  // 2b. Order inside a step.  Lanes of a wave take consecutive records, and a step's outputs get consecutive slots
  // in record order, so the order decides how many 128-byte lines one wave-wide label load or store touches.
  // Stream order scatters them; ordering every step's gates by the position of their output's FIRST reader
  // (steps processed last to first, so reader positions are final) makes producer order follow consumer order.
  if (opt.order_by_reader) {
    std::vector<uint32_t> minpos(nw, 0xFFFFFFFFu);
    std::vector<std::pair<uint32_t, uint32_t>> keyed;
    for (uint32_t s = n_steps; s-- > 0;) {
      for (int kind = 1; kind >= 0; --kind) {
        const uint32_t lo = cnt[2 * size_t(s) + kind], hi = cnt[2 * size_t(s) + kind + 1];
        keyed.clear();
        for (uint32_t k = lo; k < hi; ++k) keyed.push_back({minpos[t.c[order[k]]], order[k]});
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint32_t, uint32_t>& x, const std::pair<uint32_t, uint32_t>& y) { return x.first < y.first; });
        for (uint32_t k = lo; k < hi; ++k) order[k] = keyed[k - lo].second;
      }
      for (uint32_t k = cnt[2 * size_t(s)]; k < cnt[2 * size_t(s) + 2]; ++k) {
        const size_t i = order[k];
        minpos[t.a[i]] = std::min(minpos[t.a[i]], k);
        if (t.type[i] != uint8_t(GateType::Not)) minpos[t.b[i]] = std::min(minpos[t.b[i]], k);
      }
    }
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
  // 4. slots: HBM 0,1,2 = FALSE, TRUE, ZERO; inputs 3..; then a scan over steps with two next-fit pools
  std::vector<uint32_t> slot(nw, DEAD_WIRE);
  slot[0] = SLOT_FALSE; slot[1] = SLOT_TRUE;
  uint32_t next_in = SLOT_FIRST_INPUT;
  for (uint32_t w : inputs) { if (slot[w] == DEAD_WIRE) slot[w] = next_in++; }
  SlotPool hbm, lds(opt.lds_slots);
  hbm.reserve_low(next_in);
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
    p.ct_pos.assign(size_t(k), 0);
  }
  if (opt.hbm_arena_factor > 1) {
    uint32_t lv = next_in, pk = next_in;
    for (uint32_t s = 0; s < n_steps; ++s) {
      lv += cnt[2 * size_t(s) + 2] - cnt[2 * size_t(s)];
      pk = std::max(pk, lv);
      lv -= die_cnt[s + 1] - die_cnt[s];
    }
    hbm.preextend(uint32_t(std::min<uint64_t>(uint64_t(opt.hbm_arena_factor) * pk, SLOT_INDEX_MASK - feedback.size() - 1)));
  }
  uint32_t live = next_in, peak = next_in;
  p.steps.reserve(n_steps);
  std::vector<uint64_t> cand;
  std::vector<uint32_t> want_lds;
  for (uint32_t s = 0; s < n_steps; ++s) {
    StepDesc sd{uint32_t(p.ands.size()), cnt[2 * size_t(s) + 1] - cnt[2 * size_t(s)], uint32_t(p.xors.size()),
                cnt[2 * size_t(s) + 2] - cnt[2 * size_t(s) + 1]};
    // Which of this step's outputs get a window slot: the eligible ones (short-lived, not pinned) with the SHORTEST
    // lifetimes first — they free their slot soonest, so the window serves the largest number of wires.
    want_lds.clear();
    if (opt.lds_slots) {
      cand.clear();
      for (uint32_t k = cnt[2 * size_t(s)]; k < cnt[2 * size_t(s) + 2]; ++k) {
        uint32_t c = t.c[order[k]];
        if (last_use[c] == NEVER) continue;
        uint32_t life = dies_at(c, s) - s;
        if (life <= opt.lds_max_lifetime) cand.push_back((uint64_t(life) << 32) | k);
      }
      uint32_t avail = lds.available();
      if (cand.size() > avail) { std::nth_element(cand.begin(), cand.begin() + avail, cand.end()); cand.resize(avail); }
      for (uint64_t v : cand) want_lds.push_back(uint32_t(v));
      std::sort(want_lds.begin(), want_lds.end());
    }
    size_t wl = 0;
    for (uint32_t k = cnt[2 * size_t(s)]; k < cnt[2 * size_t(s) + 2]; ++k) {
      size_t i = order[k];
      uint32_t c = t.c[i];
      uint32_t sl = DEAD_WIRE;
      if (wl < want_lds.size() && want_lds[wl] == k) {
        ++wl;
        uint32_t l = lds.alloc();
        if (l != DEAD_WIRE) sl = l | SLOT_LDS_FLAG;
      }
      if (sl == DEAD_WIRE) {
        sl = hbm.alloc();
        if (sl > SLOT_INDEX_MASK) gsv_panic("program needs more than 2^20 HBM wire slots per instance");
      }
      slot[c] = sl;
      ++live;
      uint32_t sa = slot[t.a[i]], sb = slot[t.b[i]];
      if (sa == DEAD_WIRE || sb == DEAD_WIRE) gsv_panic("internal: operand without slot");
      const uint8_t ty = t.type[i];
      if (ty == uint8_t(GateType::Not)) sb = SLOT_ZERO;  // NOT(a) == XNOR(a, ZERO)
      ((sa & SLOT_LDS_FLAG) ? p.reads_lds : p.reads_hbm)++;
      ((sb & SLOT_LDS_FLAG) ? p.reads_lds : p.reads_hbm)++;
      ((sl & SLOT_LDS_FLAG) ? p.writes_lds : p.writes_hbm)++;
      if (ty < 8) {
        // The ciphertext goes to the gate's PROGRAM-order position: the lanes of a wave then write one contiguous
        // kilobyte.  At the gate-order index every store was its own 128-byte line (1.1 stores per line touched,
        // -16 % throughput); readers of the stream get gate order back through ct_pos (engine.cpp).
        p.ct_pos[ct_index[i]] = uint32_t(p.ands.size());
        p.ands.push_back(pack_and(sa, sb, sl, ty, uint32_t(i), uint32_t(p.ands.size())));
      }
      else p.xors.push_back(pack_xor(sa, sb, sl, ty != uint8_t(GateType::Xor)));
    }
    peak = std::max(peak, live);
    for (uint32_t k = die_cnt[s]; k < die_cnt[s + 1]; ++k) {
      uint32_t sl = slot[die_list[k]];
      if (sl & SLOT_LDS_FLAG) lds.release(sl & SLOT_INDEX_MASK); else hbm.release(sl);
      --live;
    }
    p.steps.push_back(sd);
    p.max_step_width = std::max(p.max_step_width, sd.and_cnt + sd.xor_cnt);
    if (sd.and_cnt) p.n_and_steps++;
  }
  p.peak_live = peak;
  p.n_lds_slots = lds.high();
  p.lds_slots_limit = opt.lds_slots;
  for (uint32_t w : inputs) p.input_slots.push_back(slot[w]);
  for (uint32_t w : outputs) {
    if (slot[w] == DEAD_WIRE) gsv_panic("output wire was never produced");
    p.output_slots.push_back(slot[w]);
  }
  p.fb_stage_base = std::max(hbm.high(), next_in);
  for (auto& fb : feedback) {
    if (fb.first >= outputs.size() || fb.second >= inputs.size()) gsv_panic("feedback index out of range");
    p.fb_src_slot.push_back(p.output_slots[fb.first]);
    p.fb_dst_slot.push_back(p.input_slots[fb.second]);
  }
  p.n_slots = p.fb_stage_base + uint32_t(feedback.size());
  if (p.n_slots > SLOT_INDEX_MASK) gsv_panic("program needs more than 2^20 HBM wire slots per instance");
  return p;
}

}  // namespace gsv
