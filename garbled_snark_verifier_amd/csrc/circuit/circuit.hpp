// Mode-generic streaming circuit driver (host side, header only).
//
// This is the host-side mirror of the reference's layer L2 — the code that sits
// ABOVE the `CircuitMode` seam and is identical for Execute / Garble / Evaluate
// and for the GPU recording mode:
//   * WireId / Gate / GateType            (ref: src/core/wire.rs:4-9, src/core/gate.rs:7-12,
//                                               src/core/gate_type.rs:3-15)
//   * CircuitMode                         (ref: src/circuit/modes.rs:26-51)
//   * CircuitContext                      (ref: src/circuit/circuit_context_trait.rs:12-48)
//   * ComponentMetaBuilder / Template     (ref: src/circuit/component_meta.rs:42-249,270-301)
//   * StreamingContext (execution pass)   (ref: src/circuit/streaming_mode.rs:79-118,134-271)
//   * StreamingRunner  (run_streaming)    (ref: src/circuit/mod.rs:253-301)
//
// Both the CPU oracle's modes (oracle/) and the MI355X engine's recording mode
// (csrc/engine/) plug in underneath this driver, exactly as `impl CircuitMode`
// types do in the reference.  Nothing here touches labels or AES.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace gsv {

using WireId = uint64_t;  // reference: WireId(usize)
constexpr WireId FALSE_WIRE = 0;              // circuit_context_trait.rs:2
constexpr WireId TRUE_WIRE = 1;               // circuit_context_trait.rs:3
constexpr WireId WIRE_MIN = 2;                // wire.rs:7
constexpr WireId UNREACHABLE = ~WireId(0);    // wire.rs:8 (usize::MAX)
using Credits = uint16_t;                     // storage.rs:30
using Wires = std::vector<WireId>;

[[noreturn]] inline void gsv_panic(const std::string& msg) { throw std::runtime_error(msg); }

// gate_type.rs:3-15 — discriminants are part of the C-ABI gate record.
enum class GateType : uint8_t {
  And = 0, Nand = 1, Nimp = 2, Imp = 3, Ncimp = 4, Cimp = 5, Nor = 6, Or = 7,
  Xor = 8, Xnor = 9, Not = 10,
};
constexpr int GATE_TYPE_COUNT = 11;

inline bool gate_is_free(GateType t) { return t == GateType::Xor || t == GateType::Xnor || t == GateType::Not; }

// gate_type.rs:39-61 — boolean function of each type.
inline bool gate_f(GateType t, bool a, bool b) {
  switch (t) {
    case GateType::And: return a & b;
    case GateType::Nand: return !(a & b);
    case GateType::Nimp: return a & !b;
    case GateType::Imp: return !a | b;
    case GateType::Ncimp: return !a & b;
    case GateType::Cimp: return !b | a;
    case GateType::Nor: return !(a | b);
    case GateType::Or: return a | b;
    case GateType::Xor: return a ^ b;
    case GateType::Xnor: return !(a ^ b);
    case GateType::Not: return !a;
  }
  return false;
}

struct Gate {
  WireId a, b, c;
  GateType t;
  static Gate make(GateType t, WireId a, WireId b, WireId c) { return Gate{a, b, c, t}; }
  static Gate and_(WireId a, WireId b, WireId c) { return {a, b, c, GateType::And}; }
  static Gate nand(WireId a, WireId b, WireId c) { return {a, b, c, GateType::Nand}; }
  static Gate nimp(WireId a, WireId b, WireId c) { return {a, b, c, GateType::Nimp}; }
  static Gate or_(WireId a, WireId b, WireId c) { return {a, b, c, GateType::Or}; }
  static Gate xor_(WireId a, WireId b, WireId c) { return {a, b, c, GateType::Xor}; }
  static Gate xnor(WireId a, WireId b, WireId c) { return {a, b, c, GateType::Xnor}; }
  // gate.rs:139-147 — in-place NOT: a == b == c.
  static Gate not_(WireId a) { return {a, a, a, GateType::Not}; }
  // gate.rs:149-157 — NOT expressed as XOR with the TRUE constant.
  static Gate not_with_xor(WireId a, WireId c) { return {a, TRUE_WIRE, c, GateType::Xor}; }
  // gate.rs:182-197 — f = [f0,f1,f2] selects ((a^f0)&(b^f1))^f2.
  static Gate and_variant(WireId a, WireId b, WireId c, bool f0, bool f1, bool f2) {
    return {a, b, c, static_cast<GateType>((f0 ? 4 : 0) | (f1 ? 2 : 0) | (f2 ? 1 : 0))};
  }
};

struct GateCount {  // gate_type.rs:123-148
  uint64_t n[GATE_TYPE_COUNT] = {0};
  uint64_t total() const { uint64_t s = 0; for (auto v : n) s += v; return s; }
  uint64_t nonfree() const { uint64_t s = 0; for (int i = 0; i < 8; i++) s += n[i]; return s; }
};

// ---------------------------------------------------------------------------------------------
// The seam.  modes.rs:26-51.  `lookup_wire`/`feed_wire` are typed per mode in the reference
// (WireValue); the driver itself only ever *discards* looked-up values (streaming_mode.rs:223-232),
// so the virtual interface exposes that as `consume_wire`, and the typed accessors live on the
// concrete mode classes and are called by whoever encodes inputs / decodes outputs.
struct CircuitMode {
  virtual ~CircuitMode() = default;
  virtual WireId allocate_wire(Credits credits) = 0;
  virtual void evaluate_gate(const Gate& g) = 0;
  // lookup_wire with the value dropped: consumes one credit.  Returns false if the wire is absent
  // (reference: `lookup_wire(..) -> None`, which callers `.unwrap()`).
  virtual bool consume_wire(WireId w) = 0;
  virtual void add_credits(const WireId* wires, size_t n, Credits credits) = 0;
  // Engine extension (no reference counterpart): a mode may take over a whole component call instead of letting the
  // driver run its body — the plan recorder turns chosen components into calls of separately compiled programs.
  // Returns null when the component is to be executed normally.
  virtual struct UnitHook* unit_hook() { return nullptr; }
};

struct CircuitContext;
using ChildFn = std::function<Wires(CircuitContext&, const Wires&)>;
using ComponentKey = std::string;  // reference: 8-byte SipHash of the same fields (component_key.rs:16-39)
struct ComponentMetaTemplate;
struct UnitHook {
  virtual ~UnitHook() = default;
  // Called by with_named_child after the template lookup and the input bookkeeping, before the body would run.
  // Return true after producing `out` (the component's output wires in the parent's id space); false = run the body.
  virtual bool call_unit(const ComponentKey& key, const Wires& inputs, const std::vector<Credits>& out_credits, const ComponentMetaTemplate& tpl,
                         const ChildFn& body, size_t arity, Wires& out) = 0;
};

struct CircuitContext {
  virtual ~CircuitContext() = default;
  virtual WireId issue_wire() = 0;
  virtual void add_gate(const Gate& g) = 0;
  virtual Wires with_named_child(const ComponentKey& key, const Wires& inputs, const ChildFn& f,
                                 size_t arity) = 0;
  Wires issue_wires(size_t n) {
    Wires w(n);
    for (auto& x : w) x = issue_wire();
    return w;
  }
};

// component_key.rs:16-39: name, output arity, input length, then "|name=bytes" per off-circuit param.
class KeyBuilder {
 public:
  explicit KeyBuilder(const char* name) : s_(name) {}
  KeyBuilder& param(const char* pname, const void* bytes, size_t n) {
    s_.push_back('|'); s_ += pname; s_.push_back('=');
    s_.append(static_cast<const char*>(bytes), n);
    return *this;
  }
  KeyBuilder& param_usize(const char* pname, uint64_t v) { return param(pname, &v, sizeof v); }
  ComponentKey finish(size_t arity, size_t input_len) {
    uint64_t tail[2] = {arity, input_len};
    s_.push_back('#'); s_.append(reinterpret_cast<const char*>(tail), sizeof tail);
    return s_;
  }
 private:
  std::string s_;
};

// ---------------------------------------------------------------------------------------------
// component_meta.rs:150-249
struct ComponentMetaTemplate {
  enum class Out : uint8_t { Internal, Input, Constant };
  std::vector<Credits> credits_stack;             // internal wires, issue order
  std::vector<Credits> credits_by_input_position;
  std::vector<std::pair<Out, size_t>> output_wire_types;

  // component_meta.rs:169-214. Returns the instance stack already reversed (pop from the back).
  template <class AddCredit>
  std::vector<Credits> to_instance(const std::vector<Credits>& output_credits, AddCredit&& add_credit_to_input) const {
    std::vector<Credits> st = credits_stack;
    for (size_t pos = 0; pos < credits_by_input_position.size(); ++pos)
      if (credits_by_input_position[pos] != 0) add_credit_to_input(pos, credits_by_input_position[pos]);
    if (output_wire_types.size() != output_credits.size()) gsv_panic("zip_eq: output arity mismatch");
    for (size_t i = 0; i < output_wire_types.size(); ++i) {
      const auto& ot = output_wire_types[i];
      Credits c = output_credits[i];
      switch (ot.first) {
        case Out::Constant: break;
        case Out::Input: if (c != 0) add_credit_to_input(ot.second, c); break;
        case Out::Internal: {
          uint32_t v = uint32_t(st[ot.second]) + c;
          if (v > 0xFFFF) gsv_panic("credits overflow in template instance");
          st[ot.second] = Credits(v);
        } break;
      }
    }
    std::vector<Credits> rev(st.rbegin(), st.rend());
    return rev;
  }
};

// component_meta.rs:42-148,256-301 — metadata pass: counts reads, never touches a mode.
class ComponentMetaBuilder final : public CircuitContext {
 public:
  explicit ComponentMetaBuilder(size_t input_count) : input_len_(input_count) {}
  void set_input_len_from_cursor() { input_len_ = size_t(cursor_ - WIRE_MIN); }  // new_with_input :62-71

  WireId issue_wire() override {
    WireId next = cursor_++;
    credits_.push_back(0);
    return next;
  }
  void add_gate(const Gate& g) override {
    if (g.a == UNREACHABLE || g.b == UNREACHABLE) gsv_panic("meta add_gate: UNREACHABLE input");
    bump(g.a, 1);
    bump(g.b, 1);
  }
  Wires with_named_child(const ComponentKey&, const Wires& inputs, const ChildFn&, size_t arity) override {
    for (WireId w : inputs) bump(w, 1);
    return issue_wires(arity);
  }
  ComponentMetaTemplate build(const Wires& output_wires) const {  // :112-147
    ComponentMetaTemplate t;
    using Out = ComponentMetaTemplate::Out;
    for (WireId w : output_wires) {
      if (w == TRUE_WIRE || w == FALSE_WIRE) { t.output_wire_types.push_back({Out::Constant, 0}); continue; }
      if (w == UNREACHABLE || w >= cursor_) gsv_panic("meta build: wrong output wire");
      size_t idx = size_t(w - WIRE_MIN);
      if (idx < input_len_) t.output_wire_types.push_back({Out::Input, idx});
      else t.output_wire_types.push_back({Out::Internal, idx - input_len_});
    }
    t.credits_by_input_position.assign(credits_.begin(), credits_.begin() + input_len_);
    t.credits_stack.assign(credits_.begin() + input_len_, credits_.end());
    return t;
  }
  size_t input_len() const { return input_len_; }

 private:
  void bump(WireId w, Credits c) {  // :83-109
    if (w == TRUE_WIRE || w == FALSE_WIRE || w == UNREACHABLE) return;
    if (w >= cursor_) gsv_panic("meta: external wire (not in mock range)");
    uint32_t v = uint32_t(credits_[size_t(w - WIRE_MIN)]) + c;
    if (v > 0xFFFF) gsv_panic("meta: fan-out exceeds u16 credits");
    credits_[size_t(w - WIRE_MIN)] = Credits(v);
  }
  std::vector<Credits> credits_;
  size_t input_len_;
  WireId cursor_ = WIRE_MIN;
};

// streaming_mode.rs:17-25,134-271 — execution pass.
class StreamingContext final : public CircuitContext {
 public:
  explicit StreamingContext(CircuitMode& mode) : mode_(mode) {}

  WireId issue_wire() override {  // :261-271
    auto& top = stack_.back();
    if (top.empty()) gsv_panic("No credits available");
    Credits c = top.back();
    top.pop_back();
    return mode_.allocate_wire(c);
  }
  void add_gate(const Gate& g) override {  // :134-148
    gate_count.n[int(g.t)]++;
    if (g.a == UNREACHABLE || g.b == UNREACHABLE) gsv_panic("add_gate: UNREACHABLE input");
    mode_.evaluate_gate(g);
  }
  Wires with_named_child(const ComponentKey& key, const Wires& inputs, const ChildFn& f, size_t arity) override {
    // :175-181 pop `arity` output credits from the parent frame
    std::vector<Credits> out_credits(arity);
    {
      auto& top = stack_.back();
      for (size_t i = 0; i < arity; ++i) {
        if (top.empty()) gsv_panic("with_named_child: parent frame out of credits");
        out_credits[i] = top.back();
        top.pop_back();
      }
    }
    // :189-210 template lookup / build (child's own metadata pass on mock inputs)
    auto it = templates_.find(key);
    if (it == templates_.end()) {
      ComponentMetaBuilder child(inputs.size());
      Wires mock = child.issue_wires(inputs.size());
      Wires meta_out = f(child, mock);
      it = templates_.emplace(key, child.build(meta_out)).first;
      ++templates_built;
    }
    const ComponentMetaTemplate& tpl = it->second;
    // :212-220
    std::vector<Credits> inst = tpl.to_instance(out_credits, [&](size_t idx, Credits c) {
      WireId w = inputs[idx];
      if (w != TRUE_WIRE && w != FALSE_WIRE) mode_.add_credits(&w, 1, c);
    });
    // :223-232 unpin inputs
    for (WireId w : inputs) {
      if (w == UNREACHABLE || w == TRUE_WIRE || w == FALSE_WIRE) continue;
      if (!mode_.consume_wire(w)) gsv_panic("with_named_child: input wire missing from storage");
    }
    ++component_calls;
    if (UnitHook* h = mode_.unit_hook()) {
      Wires taken;
      if (h->call_unit(key, inputs, out_credits, tpl, f, arity, taken)) {
        if (taken.size() != arity) gsv_panic("unit hook returned wrong arity");
        return taken;
      }
    }
    stack_.push_back(std::move(inst));
    Wires out = f(*this, inputs);
    if (!stack_.back().empty()) gsv_panic("component left unused credits (template/structure mismatch)");
    stack_.pop_back();
    if (out.size() != arity) gsv_panic("component returned wrong arity");
    return out;
  }

  void push_root_frame(std::vector<Credits> frame) { stack_.push_back(std::move(frame)); }
  bool root_frame_empty() const { return stack_.size() == 1 && stack_.back().empty(); }

  GateCount gate_count;
  uint64_t templates_built = 0, component_calls = 0;

 private:
  CircuitMode& mode_;
  std::vector<std::vector<Credits>> stack_;
  std::unordered_map<ComponentKey, ComponentMetaTemplate> templates_;  // reference: 5000-entry LRU
                                                                      // (eviction only re-derives the same template)
};

// circuit/mod.rs:253-301 split in the two halves around input encoding so typed feed/lookup stay
// with the concrete mode:
//   prepare()  = metadata pass over the root + to_root_ctx up to (and including) wire allocation
//   [caller encodes inputs, then looks every input up once  — mod.rs:270-273]
//   execute()  = execution pass
//   [caller looks up TRUE/FALSE and decodes outputs         — mod.rs:278-283]
using CircuitFn = std::function<Wires(CircuitContext&, const Wires&)>;

class StreamingRunner {
 public:
  StreamingRunner(CircuitMode& mode, size_t n_inputs, CircuitFn f) : ctx_(mode), n_inputs_(n_inputs), f_(std::move(f)) {}
  // Engine extension: run a component body as a root whose output i is read by nobody when live[i] == 0 (a component
  // recorded on its own must make the same dead-gate decisions as inside its parent, where some outputs had no credits).
  void set_output_liveness(std::vector<uint8_t> live) { out_live_ = std::move(live); }

  const Wires& prepare() {
    ComponentMetaBuilder meta(0);
    Wires mock_in = meta.issue_wires(n_inputs_);  // ComponentMetaBuilder::new_with_input
    meta.set_input_len_from_cursor();
    Wires meta_out = f_(meta, mock_in);
    ComponentMetaTemplate tpl = meta.build(meta_out);  // streaming_mode.rs:86
    std::vector<Credits> input_credits(n_inputs_, 1);  // :89 seed with 1
    std::vector<Credits> root_out(meta_out.size(), 1);
    if (!out_live_.empty()) {
      if (out_live_.size() != meta_out.size()) gsv_panic("output liveness mask has the wrong length");
      for (size_t i = 0; i < root_out.size(); ++i) root_out[i] = out_live_[i] ? 1 : 0;
    }
    std::vector<Credits> inst = tpl.to_instance(root_out, [&](size_t idx, Credits c) {
      size_t rev = n_inputs_ - 1 - idx;  // :93
      uint32_t v = uint32_t(input_credits[rev]) + c;
      if (v > 0xFFFF) gsv_panic("root input credits overflow");
      input_credits[rev] = Credits(v);
    });
    inst.insert(inst.end(), input_credits.begin(), input_credits.end());  // :98
    ctx_.push_root_frame(std::move(inst));
    inputs_ = ctx_.issue_wires(n_inputs_);  // :111
    return inputs_;
  }
  const Wires& execute() {
    outputs_ = f_(ctx_, inputs_);
    if (!ctx_.root_frame_empty()) gsv_panic("root frame left unused credits");
    return outputs_;
  }
  const Wires& inputs() const { return inputs_; }
  const Wires& outputs() const { return outputs_; }
  StreamingContext& ctx() { return ctx_; }

 private:
  StreamingContext ctx_;
  size_t n_inputs_;
  CircuitFn f_;
  Wires inputs_, outputs_;
  std::vector<uint8_t> out_live_;
};

}  // namespace gsv
