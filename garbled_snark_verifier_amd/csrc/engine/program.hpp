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
#include <queue>
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
// constant and the all-zero label; window entry 0 is an all-zero label too (never allocated): it stands
// for the ABSENT operands of a fused gate, so that every record has the same shape.
constexpr uint32_t SLOT_BITS = 21;
constexpr uint32_t SLOT_LDS_FLAG = 1u << 20;
constexpr uint32_t SLOT_INDEX_MASK = SLOT_LDS_FLAG - 1;
constexpr uint32_t SLOT_MASK = (1u << SLOT_BITS) - 1;
constexpr uint32_t SLOT_FALSE = 0, SLOT_TRUE = 1, SLOT_ZERO = 2, SLOT_FIRST_INPUT = 3;
constexpr uint32_t SLOT_LDS_ZERO = SLOT_LDS_FLAG | 0u;
constexpr uint32_t LDS_WINDOW_SLOTS = GSV_LDS_SLOTS;  // 90 KiB of the CU's 160 KiB (64 KiB go to the banked AES tables)

// Fused gates (see fuse_trace).  Labels are XOR-linear, so a free gate whose output has one reader (or is cheap
// to recompute) is folded into that reader, and the XOR that consumes a single-reader AND output is folded into the
// AND.  A full adder (4 XOR + 1 AND, three dependent levels) becomes ONE fused AND (one level) and one 3-input XOR:
//   AND-family : out = AND_t(a1 ^ a2, b1 ^ b2) ^ p     t = the 3 alpha bits with the folded inputs' NOT-parities XORed in
//   free       : out = x1 ^ x2 ^ x3 ^ x4 (^ delta)
// Free gate, 16 bytes:        w0 = x1 | x2<<21 | x3<<42 | parity<<63 ;  w1 = x4 | out<<21
struct XorRec { uint64_t w0, w1; };
inline XorRec pack_xor(const uint32_t x[4], uint32_t c, bool parity) {
  return XorRec{uint64_t(x[0]) | (uint64_t(x[1]) << 21) | (uint64_t(x[2]) << 42) | (uint64_t(parity ? 1 : 0) << 63), uint64_t(x[3]) | (uint64_t(c) << 21)};
}
// AND-family gate, 32 bytes:  w0 = a1 | a2<<21 | b1<<42 | (t&1)<<63 ;  w1 = b2 | p<<21 | out<<42 | ((t>>1)&1)<<63 ;
//                             w2 = gate id relative to the replay's base (40 bits) | (t>>2)<<40 ;  w3 = 0
// The ciphertext of record k of the program goes to position k of the replay's block of the device stream.
struct AndRec { uint64_t w0, w1, w2, w3; };
inline AndRec pack_and(const uint32_t in[5], uint32_t c, uint32_t type, uint64_t gid) {
  return AndRec{uint64_t(in[0]) | (uint64_t(in[1]) << 21) | (uint64_t(in[2]) << 42) | (uint64_t(type & 1u) << 63),
                uint64_t(in[3]) | (uint64_t(in[4]) << 21) | (uint64_t(c) << 42) | (uint64_t((type >> 1) & 1u) << 63),
                gid | (uint64_t((type >> 2) & 1u) << 40), 0};
}
// The FOUR-WIRE form (Program::and_terms == 4; latency-bound programs, compile_program 0.):
//   out = AND_t(a1 ^ a2 ^ a3 ^ a4, b1 ^ b2 ^ b3 ^ b4) ^ p
//   w0 = a1 | a2<<21 | a3<<42 | (t&1)<<63 ;  w1 = a4 | b1<<21 | b2<<42 | ((t>>1)&1)<<63 ;
//   w2 = b3 | b4<<21 | p<<42 | ((t>>2)&1)<<63 ;  w3 = out | gate id (31 bits: an index into the program's stream) << 21
// A free gate that feeds an AND through a XOR of up to four wires then needs no step of its own: the square-root ladders lose 42 % of
// their steps, the inversions 33 %, an Fq12 multiplication 13 % (tools/depth_stats.cpp) — at the price of four more label loads per AND.
// in[9]: a1..a4, b1..b4, p.
inline AndRec pack_and4(const uint32_t in[9], uint32_t c, uint32_t type, uint64_t gid) {
  return AndRec{uint64_t(in[0]) | (uint64_t(in[1]) << 21) | (uint64_t(in[2]) << 42) | (uint64_t(type & 1u) << 63),
                uint64_t(in[3]) | (uint64_t(in[4]) << 21) | (uint64_t(in[5]) << 42) | (uint64_t((type >> 1) & 1u) << 63),
                uint64_t(in[6]) | (uint64_t(in[7]) << 21) | (uint64_t(in[8]) << 42) | (uint64_t((type >> 2) & 1u) << 63),
                uint64_t(c) | (gid << 21)};
}
struct StepDesc { uint32_t and_off, and_cnt, xor_off, xor_cnt; };  // one dependency level: AND-family + free gates

struct Program {
  std::vector<StepDesc> steps;
  uint32_t n_steps = 0;        // == steps.size(); kept separately because a plan loaded straight to the device (gsv_plan_load) has no host copy of the records
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
  uint64_t n_fused_free = 0;   // free gates left after fusion (the device executes n_ct + n_fused_free records per replay)
  uint32_t and_depth = 0, n_and_steps = 0, max_step_width = 0;
  uint32_t peak_live = 0;
  uint32_t and_terms = 2;      // wires per AND input in the records: 2 (pack_and) or 4 (pack_and4)
  uint64_t reads_lds = 0, reads_hbm = 0, writes_lds = 0, writes_hbm = 0;  // label accesses per replay by location
  // Set when the records (steps, ands, xors, ct_pos) were written to a plan file as soon as the program existed and dropped from
  // memory (gsv_plan_build_file): the offset of the program's block in that file.  The metadata above stays.
  uint64_t file_off = 0;
  bool spilled = false;
};

struct CompileOptions {
  uint32_t lds_slots = LDS_WINDOW_SLOTS;  // 0 = keep every wire in HBM
  // A wire goes to the LDS window only if it dies within this many steps; among the eligible outputs of a step the shortest-lived get
  // the free slots.  8 and 1024 are equal on wide programs (the window is always contended: 1.030 vs 1.032e11 gates/s on fq12_mix at
  // 1024 instances); on narrow ones, whose window is mostly empty, long-lived wires stop going through the HBM wire file:
  // +5 % on the decompression ladders and the inversions (profiles/r03_kernel/lds_lifetime_ab.log).
  uint32_t lds_max_lifetime = 1024;
  bool order_by_reader = true;            // order the gates of a step by the position of their output's first reader
  uint32_t hbm_arena_factor = 4;          // HBM wire file = factor x peak live wires (next-fit then sweeps mostly free space)
  bool fuse = true;                       // fold free gates into their readers / into the AND that feeds them
  uint32_t fuse_dup_fanout = 2;           // a free gate of <= 2 operands is also folded (recomputed) when it has up to this many readers
  uint32_t and_cap = 0, xor_cap = 0;      // most AND-family / free gates in one step (0 = no cap: ASAP levels); see compile_program 1b
  // Wires per AND input: 2, 4, or 0 = choose — four for a program whose two-wire schedule has fewer than `narrow_width` fused gates
  // per step on average (latency-bound: fewer steps matter), two otherwise (throughput-bound: fewer label loads matter).
  uint32_t and_terms = 0;
  uint32_t narrow_width = 600;
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

// ---- gate fusion ---------------------------------------------------------------------------------
constexpr uint32_t FUSED_AND = 0x80;  // (FusedOps::push tests this bit)
constexpr int FUSED_IN = 9;  // operand slots of a fused op: AND a1..a4 (0-3), b1..b4 (4-7), p (8) ; free x1..x4 (0-3) ; DEAD_WIRE = absent
struct FusedOps {
  std::vector<uint8_t> kind;   // FUSED_AND | 3-bit type   or   parity bit of a free op
  std::vector<uint32_t> in;    // `stride` per op: 9 in the four-wire form (layout above), 5 in the two-wire form: AND a1 a2 b1 b2 p ; free x1 x2 x3 x4 -
  std::vector<uint32_t> out;
  std::vector<uint32_t> gid;   // stream index of the AND gate (its gate id inside a replay); unused for free ops
  int stride = FUSED_IN;
  size_t size() const { return out.size(); }
  void reserve(size_t n) { kind.reserve(n); in.reserve(n * size_t(stride)); out.reserve(n); gid.reserve(n); }
  // i9: the nine-slot layout whatever the stride
  void push(uint8_t k, const uint32_t i9[FUSED_IN], uint32_t o, uint32_t g) {
    kind.push_back(k);
    if (stride == FUSED_IN) in.insert(in.end(), i9, i9 + FUSED_IN);
    else if (k & 0x80) { const uint32_t v[5] = {i9[0], i9[1], i9[4], i9[5], i9[8]}; in.insert(in.end(), v, v + 5); }
    else { const uint32_t v[5] = {i9[0], i9[1], i9[2], i9[3], 0xFFFFFFFFu}; in.insert(in.end(), v, v + 5); }
    out.push_back(o); gid.push_back(g);
  }
};

// ka: wires per AND input (2 or 4); an expression of up to ka wires (two when ka == 2) is also folded into TWO readers (recomputed)
inline FusedOps fuse_trace(const Trace& t, const std::vector<uint32_t>& inputs, const std::vector<uint32_t>& outputs, const CompileOptions& opt, uint32_t ka = 2) {
  const size_t n = t.size();
  const uint32_t nw = t.n_wires;
  constexpr uint8_t NOT = uint8_t(GateType::Not), XOR = uint8_t(GateType::Xor), XNOR = uint8_t(GateType::Xnor);
  FusedOps f;
  f.stride = ka > 2 ? FUSED_IN : 5;
  if (!opt.fuse) {
    for (size_t i = 0; i < n; ++i) {
      if (t.c[i] == DEAD_WIRE) continue;
      const uint8_t ty = t.type[i];
      constexpr uint32_t D = DEAD_WIRE;
      if (ty < 8) { const uint32_t in[FUSED_IN] = {t.a[i], D, D, D, t.b[i], D, D, D, D}; f.push(uint8_t(FUSED_AND | ty), in, t.c[i], uint32_t(i)); }
      else { const uint32_t in[FUSED_IN] = {t.a[i], ty == NOT ? D : t.b[i], D, D, D, D, D, D, D}; f.push(ty == XOR ? 0 : 1, in, t.c[i], 0); }
    }
    return f;
  }
  constexpr uint32_t KX = 4;
  struct Expr { uint32_t w[KX]; uint8_t n, par; };  // sorted wire list (duplicates cancelled) and NOT-parity
  std::vector<uint8_t> fan(nw, 0), pinned(nw, 0), isfree(nw, 0), need(nw, 0);
  std::vector<uint32_t> cons(nw, DEAD_WIRE);  // a reader (THE reader when fan == 1)
  auto use = [&](uint32_t w, size_t i) { if (fan[w] < 255) ++fan[w]; cons[w] = uint32_t(i); };
  for (size_t i = 0; i < n; ++i) {
    if (t.c[i] == DEAD_WIRE) continue;
    use(t.a[i], i);
    if (t.type[i] != NOT) use(t.b[i], i);
    if (t.type[i] >= 8) isfree[t.c[i]] = 1;
  }
  pinned[0] = pinned[1] = 1;
  for (uint32_t w : inputs) pinned[w] = 1;
  for (uint32_t w : outputs) pinned[w] = 1;
  std::vector<Expr> expr(nw);
  std::vector<uint8_t> absorbed(n, 0);
  struct AndDec { uint32_t in[FUSED_IN]; uint32_t out; uint8_t type; };
  const uint32_t dup_terms = ka > 2 ? ka : 2;
  std::vector<AndDec> decs;
  { size_t n_and = 0; for (size_t i = 0; i < n; ++i) n_and += t.c[i] != DEAD_WIRE && t.type[i] < 8; decs.reserve(n_and); f.reserve(n_and + n_and / 2 + 1024); }
  auto single = [](uint32_t x) { Expr e; e.n = 1; e.par = 0; e.w[0] = x; return e; };
  // operand list of wire x for a reader that takes at most `cap` wires from it
  auto resolve = [&](uint32_t x, uint32_t cap) -> Expr {
    if (isfree[x] && !pinned[x]) {
      const Expr& ex = expr[x];
      if (ex.n <= cap && (fan[x] == 1 || (fan[x] <= opt.fuse_dup_fanout && ex.n <= dup_terms))) return ex;
    }
    need[x] = 1;
    return single(x);
  };
  struct Wide { uint32_t w[2 * KX]; uint32_t n; };
  auto symdiff = [](const Expr& a, const Expr& b) {
    Wide r; r.n = 0;
    uint32_t i = 0, j = 0;
    while (i < a.n || j < b.n) {
      if (j == b.n || (i < a.n && a.w[i] < b.w[j])) r.w[r.n++] = a.w[i++];
      else if (i == a.n || b.w[j] < a.w[i]) r.w[r.n++] = b.w[j++];
      else { ++i; ++j; }
    }
    return r;
  };
  for (size_t i = 0; i < n; ++i) {
    const uint32_t c = t.c[i];
    if (c == DEAD_WIRE || absorbed[i]) continue;
    const uint8_t ty = t.type[i];
    const uint32_t a = t.a[i], b = t.b[i];
    if (ty >= 8) {
      Expr ea = resolve(a, KX), eb;
      if (ty == NOT) { eb.n = 0; eb.par = 0; } else eb = resolve(b, KX);
      Wide r = symdiff(ea, eb);
      if (r.n > KX) {  // too long: read the longer folded side as a wire instead, then the other one
        if (ea.n >= eb.n && ea.n > 1) { ea = single(a); need[a] = 1; } else { eb = single(b); need[b] = 1; }
        r = symdiff(ea, eb);
        if (r.n > KX) { ea = single(a); need[a] = 1; eb = single(b); need[b] = 1; r = symdiff(ea, eb); }
      }
      Expr e; e.n = uint8_t(r.n); e.par = uint8_t(ea.par ^ eb.par ^ (ty != XOR ? 1 : 0));
      for (uint32_t k = 0; k < r.n; ++k) e.w[k] = r.w[k];
      expr[c] = e;
    } else {
      const Expr ea = resolve(a, ka), eb = resolve(b, ka);
      AndDec d;
      for (uint32_t k = 0; k < 4; ++k) { d.in[k] = k < ea.n ? ea.w[k] : DEAD_WIRE; d.in[4 + k] = k < eb.n ? eb.w[k] : DEAD_WIRE; }
      d.in[8] = DEAD_WIRE;
      d.type = uint8_t(ty ^ (ea.par << 2) ^ (eb.par << 1));
      d.out = c;
      // the single reader of this AND is a XOR/XNOR with an operand that already exists: fold it into the AND's output
      if (fan[c] == 1 && !pinned[c]) {
        const size_t j = cons[c];
        const uint8_t tj = t.type[j];
        if ((tj == XOR || tj == XNOR) && t.c[j] != DEAD_WIRE) {
          const uint32_t y = t.a[j] == c ? t.b[j] : t.a[j];
          if (y != c && y < c) {  // SSA ids grow in definition order: y is defined before this gate
            const Expr ey = resolve(y, 1);
            d.in[8] = ey.n ? ey.w[0] : DEAD_WIRE;
            d.type ^= uint8_t(ey.par ^ (tj == XNOR ? 1 : 0));
            d.out = t.c[j];
            absorbed[j] = 1;
            isfree[d.out] = 0;
          }
        }
      }
      decs.push_back(d);
    }
  }
  size_t kd = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t c = t.c[i];
    if (c == DEAD_WIRE || absorbed[i]) continue;
    if (t.type[i] < 8) { const AndDec& d = decs[kd++]; f.push(uint8_t(FUSED_AND | d.type), d.in, d.out, uint32_t(i)); continue; }
    if (!need[c] && !pinned[c]) continue;  // folded into every reader
    const Expr& e = expr[c];
    uint32_t in[FUSED_IN];
    for (int k = 0; k < FUSED_IN; ++k) in[k] = DEAD_WIRE;
    for (uint32_t k = 0; k < e.n; ++k) in[k] = e.w[k];
    f.push(e.par, in, c, 0);
  }
  return f;
}

// inputs / outputs: SSA ids of the circuit's input and output wires.
// feedback: pairs (output index -> input index) copied at the end of every replay (chained circuits).
inline Program compile_program(const Trace& t, const std::vector<uint32_t>& inputs, const std::vector<uint32_t>& outputs,
                               const std::vector<std::pair<uint32_t, uint32_t>>& feedback = {}, const CompileOptions& opt_in = CompileOptions()) {
  CompileOptions opt = opt_in;
  if (const char* e = getenv("GSV_AND_CAP")) opt.and_cap = uint32_t(atoi(e));  // tuning knobs (the defaults are the measured best)
  if (const char* e = getenv("GSV_XOR_CAP")) opt.xor_cap = uint32_t(atoi(e));
  if (const char* e = getenv("GSV_LDS_SLOTS_CAP")) opt.lds_slots = std::min<uint32_t>(opt.lds_slots, uint32_t(atoi(e)));  // experiments: more instances per workgroup
  const uint32_t nw = t.n_wires;
  Program p;
  p.n_gates = t.size();
  if (t.size() >= 0x7FFFFFFFull) gsv_panic("program too large: gate index must fit 31 bits per replay");
  if (opt.lds_slots > SLOT_INDEX_MASK) gsv_panic("LDS window larger than the slot encoding");
  for (size_t i = 0; i < t.size(); ++i) { p.gate_count[t.type[i]]++; if (t.c[i] == DEAD_WIRE) p.n_dead++; }
  // 0. Fusion, and how many wires an AND input may take.  Throughput-bound programs (wide steps) keep the two-wire records: every
  // extra operand is a label load on the LDS pipe that the AES needs.  Latency-bound programs (narrow steps: ladders, inversions,
  // carry chains) take four: a free gate that only feeds ANDs then disappears as a step of its own.
  if (const char* e = getenv("GSV_AND_TERMS")) { const int v = atoi(e); if (v == 0 || v == 2 || v == 4) opt.and_terms = uint32_t(v); }
  if (!opt.fuse) opt.and_terms = 2;
  FusedOps f = fuse_trace(t, inputs, outputs, opt, opt.and_terms == 4 ? 4u : 2u);
  p.and_terms = opt.and_terms == 4 ? 4u : 2u;
  auto asap_steps = [&](const FusedOps& g) {
    std::vector<uint32_t> l(nw, 0);
    uint32_t mx = 0;
    for (size_t i = 0; i < g.size(); ++i) {
      uint32_t v = 0;
      for (int k = 0; k < g.stride; ++k) { const uint32_t w = g.in[size_t(g.stride) * i + k]; if (w != DEAD_WIRE) v = std::max(v, l[w]); }
      l[g.out[i]] = v + 1;
      mx = std::max(mx, v + 1);
    }
    return mx;
  };
  if (opt.and_terms == 0 && f.size()) {
    const uint32_t steps2 = asap_steps(f);
    if (double(f.size()) / steps2 < double(opt.narrow_width)) {
      FusedOps f4 = fuse_trace(t, inputs, outputs, opt, 4u);
      if (asap_steps(f4) < steps2) { f = std::move(f4); p.and_terms = 4; }
    }
  }
  const size_t n = f.size();
  auto is_and = [&](size_t i) { return (f.kind[i] & FUSED_AND) != 0; };
  const int fi = f.stride;  // operand slots per op: 5 (two-wire) or 9 (four-wire)
  auto ins = [&](size_t i) { return &f.in[size_t(fi) * i]; };

  // 1. ASAP dependency level per wire (inputs / constants = 0) and AND-depth (statistic).
  std::vector<uint32_t> lev(nw, 0), ad(nw, 0);
  uint32_t n_steps = 0;
  for (size_t i = 0; i < n; ++i) {
    uint32_t l = 0, d = 0;
    for (int k = 0; k < fi; ++k) { const uint32_t w = ins(i)[k]; if (w != DEAD_WIRE) { l = std::max(l, lev[w]); d = std::max(d, ad[w]); } }
    const uint32_t c = f.out[i];
    lev[c] = l + 1;
    ad[c] = d + (is_and(i) ? 1 : 0);
    n_steps = std::max(n_steps, lev[c]);
    p.and_depth = std::max(p.and_depth, ad[c]);
  }
  // 1b. Width-capped list scheduling.  ASAP levels put every independent sub-circuit's multiplier array into the same few
  // steps (an Fq12 multiplication: 740 of 8197 levels hold 59 % of the gates, 7000+ each) and leave thousands of levels
  // on the carry chains with a few dozen gates.  The wide levels overflow the LDS label window into the HBM wire file, the
  // narrow ones cost a barrier for a handful of gates.  With a cap on the AND-family and free gates of a step, the ready
  // gates with the longest path to a sink go first and the rest wait for a later step: work with slack moves under the
  // carry chains.  The schedule is at least as long as the critical path, steps become uniform, live ranges shrink.
  // Labels do not depend on the schedule (only on gate ids and dataflow), so the result is the same stream.
  std::vector<uint32_t> stp;
  if (opt.and_cap || opt.xor_cap) {
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> prod(nw, NONE), pending(n, 0), height(n, 0), succ_off(n + 1, 0);
    for (size_t i = 0; i < n; ++i) prod[f.out[i]] = uint32_t(i);
    for (size_t i = 0; i < n; ++i)
      for (int k = 0; k < fi; ++k) { const uint32_t w = ins(i)[k]; if (w != DEAD_WIRE && prod[w] != NONE) { succ_off[prod[w] + 1]++; pending[i]++; } }
    for (size_t i = 0; i < n; ++i) succ_off[i + 1] += succ_off[i];
    std::vector<uint32_t> succ(succ_off[n]);
    {
      std::vector<uint32_t> cur(succ_off.begin(), succ_off.end() - 1);
      for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < fi; ++k) { const uint32_t w = ins(i)[k]; if (w != DEAD_WIRE && prod[w] != NONE) succ[cur[prod[w]]++] = uint32_t(i); }
    }
    for (size_t i = n; i-- > 0;)  // the fused list is in stream order: producers precede their readers
      for (int k = 0; k < fi; ++k) { const uint32_t w = ins(i)[k]; if (w != DEAD_WIRE && prod[w] != NONE) height[prod[w]] = std::max(height[prod[w]], height[i] + 1); }
    auto key = [&](uint32_t i) -> uint64_t { return (uint64_t(height[i]) << 32) | (0xFFFFFFFFu - i); };  // longest path first, then stream order
    std::priority_queue<uint64_t> ready[2];
    for (size_t i = 0; i < n; ++i) if (pending[i] == 0) ready[is_and(i) ? 0 : 1].push(key(uint32_t(i)));
    const uint32_t cap[2] = {opt.and_cap ? opt.and_cap : 0xFFFFFFFFu, opt.xor_cap ? opt.xor_cap : 0xFFFFFFFFu};
    stp.assign(n, 0);
    std::vector<uint32_t> batch;
    size_t done = 0;
    uint32_t s = 0;
    while (done < n) {
      batch.clear();
      for (int kind = 0; kind < 2; ++kind)
        for (uint32_t c = 0; c < cap[kind] && !ready[kind].empty(); ++c) { batch.push_back(0xFFFFFFFFu - uint32_t(ready[kind].top())); ready[kind].pop(); }
      if (batch.empty()) gsv_panic("internal: list scheduler found no ready gate");
      for (uint32_t i : batch) stp[i] = s;
      for (uint32_t i : batch)
        for (uint32_t q = succ_off[i]; q < succ_off[i + 1]; ++q) { const uint32_t j = succ[q]; if (--pending[j] == 0) ready[is_and(j) ? 0 : 1].push(key(j)); }
      done += batch.size();
      ++s;
    }
    n_steps = s;
  }
  auto step_of = [&](size_t i) -> uint32_t { return stp.empty() ? lev[f.out[i]] - 1 : stp[i]; };
  // 2. counting sort of the ops by (step, kind): AND-family first, then free gates
  std::vector<uint32_t> cnt(2 * size_t(n_steps) + 1, 0);
  for (size_t i = 0; i < n; ++i) cnt[2 * size_t(step_of(i)) + (is_and(i) ? 0 : 1) + 1]++;
  for (size_t k = 0; k < 2 * size_t(n_steps); ++k) cnt[k + 1] += cnt[k];
  std::vector<uint32_t> order(n);
  {
    std::vector<uint32_t> cursor(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < n; ++i) order[cursor[2 * size_t(step_of(i)) + (is_and(i) ? 0 : 1)]++] = uint32_t(i);
  }
  // 2b. Order inside a step.  Lanes of a wave take consecutive records, and a step's outputs get consecutive slots
  // in record order, so the order decides how many 128-byte lines one wave-wide label load or store touches.
  // Stream order scatters them; ordering every step's gates by the position of their output's FIRST reader
  // (steps processed last to first, so reader positions are final) makes producer order follow consumer order.
  if (opt.order_by_reader) {
    std::vector<uint32_t> minpos(nw, 0xFFFFFFFFu);
    std::vector<uint64_t> keyed;  // (first reader's position << 32) | position inside the segment: a plain sort of these IS the stable sort by reader position
    std::vector<uint32_t> seg;
    for (uint32_t s = n_steps; s-- > 0;) {
      for (int kind = 1; kind >= 0; --kind) {
        const uint32_t lo = cnt[2 * size_t(s) + kind], hi = cnt[2 * size_t(s) + kind + 1];
        if (hi - lo < 2) continue;
        keyed.clear();
        for (uint32_t k = lo; k < hi; ++k) keyed.push_back((uint64_t(minpos[f.out[order[k]]]) << 32) | (k - lo));
        std::sort(keyed.begin(), keyed.end());
        seg.assign(order.begin() + lo, order.begin() + hi);
        for (uint32_t k = lo; k < hi; ++k) order[k] = seg[uint32_t(keyed[k - lo])];
      }
      for (uint32_t k = cnt[2 * size_t(s)]; k < cnt[2 * size_t(s) + 2]; ++k)
        for (int q = 0; q < fi; ++q) { const uint32_t w = ins(order[k])[q]; if (w != DEAD_WIRE) minpos[w] = std::min(minpos[w], k); }
    }
  }
  // 3. last reader step per wire.  NEVER = pinned, UNUSED = no reader.
  constexpr uint32_t NEVER = 0xFFFFFFFFu, UNUSED = 0xFFFFFFFEu;
  std::vector<uint32_t> last_use(nw, UNUSED);
  for (size_t i = 0; i < n; ++i) {
    const uint32_t s = step_of(i);
    for (int q = 0; q < fi; ++q) { const uint32_t w = ins(i)[q]; if (w != DEAD_WIRE && (last_use[w] == UNUSED || last_use[w] < s)) last_use[w] = s; }
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
  if (opt.lds_slots) lds.reserve_low(1);  // window entry 0 = the all-zero label of absent operands
  const uint32_t absent = opt.lds_slots ? SLOT_LDS_ZERO : SLOT_ZERO;
  auto dies_at = [&](uint32_t w, uint32_t def_step) -> uint32_t { return last_use[w] == UNUSED ? def_step : last_use[w]; };
  std::vector<uint32_t> die_cnt(size_t(n_steps) + 1, 0);
  for (size_t k = 0; k < n; ++k) {
    const size_t i = order[k];
    const uint32_t c = f.out[i];
    if (last_use[c] == NEVER) continue;
    die_cnt[dies_at(c, step_of(i)) + 1]++;
  }
  for (uint32_t s = 0; s < n_steps; ++s) die_cnt[s + 1] += die_cnt[s];
  std::vector<uint32_t> die_list(die_cnt[n_steps]);
  {
    std::vector<uint32_t> cur(die_cnt.begin(), die_cnt.end() - 1);
    for (size_t k = 0; k < n; ++k) {
      const size_t i = order[k];
      const uint32_t c = f.out[i];
      if (last_use[c] == NEVER) continue;
      die_list[cur[dies_at(c, step_of(i))]++] = c;
    }
  }
  // gate-order ciphertext index of an AND op = its rank among the AND ops (the fused list keeps stream order)
  std::vector<uint32_t> ct_index(n, 0);
  {
    uint64_t k = 0;
    for (size_t i = 0; i < n; ++i) if (is_and(i)) ct_index[i] = uint32_t(k++);
    p.n_ct = k;
    p.ct_pos.assign(size_t(k), 0);
    p.n_fused_free = n - k;
    p.ands.reserve(size_t(k));
    p.xors.reserve(n - size_t(k));
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
        const uint32_t c = f.out[order[k]];
        if (last_use[c] == NEVER) continue;
        const uint32_t life = dies_at(c, s) - s;
        if (life <= opt.lds_max_lifetime) cand.push_back((uint64_t(life) << 32) | k);
      }
      const uint32_t avail = lds.available();
      if (cand.size() > avail) { std::nth_element(cand.begin(), cand.begin() + avail, cand.end()); cand.resize(avail); }
      for (uint64_t v : cand) want_lds.push_back(uint32_t(v));
      std::sort(want_lds.begin(), want_lds.end());
    }
    size_t wl = 0;
    for (uint32_t k = cnt[2 * size_t(s)]; k < cnt[2 * size_t(s) + 2]; ++k) {
      const size_t i = order[k];
      const uint32_t c = f.out[i];
      uint32_t sl = DEAD_WIRE;
      if (wl < want_lds.size() && want_lds[wl] == k) {
        ++wl;
        const uint32_t l = lds.alloc();
        if (l != DEAD_WIRE) sl = l | SLOT_LDS_FLAG;
      }
      if (sl == DEAD_WIRE) {
        sl = hbm.alloc();
        if (sl > SLOT_INDEX_MASK) gsv_panic("program needs more than 2^20 HBM wire slots per instance");
      }
      slot[c] = sl;
      ++live;
      uint32_t si[FUSED_IN];
      for (int q = 0; q < fi; ++q) {
        const uint32_t w = ins(i)[q];
        if (w == DEAD_WIRE) { si[q] = absent; continue; }
        si[q] = slot[w];
        if (si[q] == DEAD_WIRE) gsv_panic("internal: operand without slot");
        ((si[q] & SLOT_LDS_FLAG) ? p.reads_lds : p.reads_hbm)++;
      }
      ((sl & SLOT_LDS_FLAG) ? p.writes_lds : p.writes_hbm)++;
      if (is_and(i)) {
        // The ciphertext goes to the gate's PROGRAM-order position: the lanes of a wave then write one contiguous
        // kilobyte.  At the gate-order index every store was its own 128-byte line (1.1 stores per line touched,
        // -16 % throughput); readers of the stream get gate order back through ct_pos (engine.cpp).
        p.ct_pos[ct_index[i]] = uint32_t(p.ands.size());
        if (p.and_terms == 4) p.ands.push_back(pack_and4(si, sl, f.kind[i] & 7u, f.gid[i]));
        else p.ands.push_back(pack_and(si, sl, f.kind[i] & 7u, f.gid[i]));
      } else {
        p.xors.push_back(pack_xor(si, sl, (f.kind[i] & 1u) != 0));
      }
    }
    peak = std::max(peak, live);
    for (uint32_t k = die_cnt[s]; k < die_cnt[s + 1]; ++k) {
      const uint32_t sl = slot[die_list[k]];
      if (sl & SLOT_LDS_FLAG) lds.release(sl & SLOT_INDEX_MASK); else hbm.release(sl);
      --live;
    }
    p.steps.push_back(sd);
    p.max_step_width = std::max(p.max_step_width, sd.and_cnt + sd.xor_cnt);
    if (sd.and_cnt) p.n_and_steps++;
  }
  if (getenv("GSV_SCHED_STATS") && p.and_terms == 2) {
    // coalescing of the wire-file accesses: distinct 128-byte lines (8 slots) per wave-wide access (64 consecutive records of a step)
    uint64_t acc[2] = {0, 0}, lines[2] = {0, 0}, lanes[2] = {0, 0};
    std::vector<uint32_t> ls;
    auto tally = [&](int rw) { if (ls.empty()) return; std::sort(ls.begin(), ls.end()); acc[rw]++; lanes[rw] += ls.size(); lines[rw] += uint64_t(std::unique(ls.begin(), ls.end()) - ls.begin()); };
    for (const StepDesc& sd : p.steps) {
      for (uint32_t w0 = 0; w0 < sd.and_cnt; w0 += 64) {
        const uint32_t w1 = std::min(sd.and_cnt, w0 + 64);
        for (int q = 0; q < 6; ++q) {
          ls.clear();
          for (uint32_t k = w0; k < w1; ++k) {
            const AndRec& r = p.ands[sd.and_off + k];
            const uint32_t sl[6] = {uint32_t(r.w0 & 0x1FFFFF), uint32_t((r.w0 >> 21) & 0x1FFFFF), uint32_t((r.w0 >> 42) & 0x1FFFFF), uint32_t(r.w1 & 0x1FFFFF), uint32_t((r.w1 >> 21) & 0x1FFFFF), uint32_t((r.w1 >> 42) & 0x1FFFFF)};
            if (!(sl[q] & SLOT_LDS_FLAG) && sl[q] != SLOT_ZERO) ls.push_back(sl[q] / 8);
          }
          tally(q == 5 ? 1 : 0);
        }
      }
      for (uint32_t w0 = 0; w0 < sd.xor_cnt; w0 += 64) {
        const uint32_t w1 = std::min(sd.xor_cnt, w0 + 64);
        for (int q = 0; q < 5; ++q) {
          ls.clear();
          for (uint32_t k = w0; k < w1; ++k) {
            const XorRec& r = p.xors[sd.xor_off + k];
            const uint32_t sl[5] = {uint32_t(r.w0 & 0x1FFFFF), uint32_t((r.w0 >> 21) & 0x1FFFFF), uint32_t((r.w0 >> 42) & 0x1FFFFF), uint32_t(r.w1 & 0x1FFFFF), uint32_t((r.w1 >> 21) & 0x1FFFFF)};
            if (!(sl[q] & SLOT_LDS_FLAG) && sl[q] != SLOT_ZERO) ls.push_back(sl[q] / 8);
          }
          tally(q == 4 ? 1 : 0);
        }
      }
    }
    fprintf(stderr, "[sched] wire-file loads: %llu wave accesses, %.1f lanes and %.1f lines each (%.2f lanes per line); stores: %llu wave accesses, %.1f lanes, %.1f lines (%.2f per line)\n",
            (unsigned long long)acc[0], double(lanes[0]) / std::max<uint64_t>(1, acc[0]), double(lines[0]) / std::max<uint64_t>(1, acc[0]), double(lanes[0]) / std::max<uint64_t>(1, lines[0]),
            (unsigned long long)acc[1], double(lanes[1]) / std::max<uint64_t>(1, acc[1]), double(lines[1]) / std::max<uint64_t>(1, acc[1]), double(lanes[1]) / std::max<uint64_t>(1, lines[1]));
  }
  p.peak_live = peak;
  p.n_steps = uint32_t(p.steps.size());
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
