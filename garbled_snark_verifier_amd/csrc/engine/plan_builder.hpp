// Plan builder: records a circuit under the unchanged two-pass driver, but turns chosen components ("units") into
// CALLS of separately compiled programs instead of flattening them into the root trace.
//
// Why: the reference instantiates a few component shapes thousands of times (with_named_child,
// src/circuit/streaming_mode.rs:150-247; the verifier is ~330 Fq12-sized multiplications/squarings plus glue).  A flat
// recording costs 13 bytes of trace per gate — out of reach at 11 B gates — while a unit is recorded and compiled once
// per (component key, liveness pattern of its outputs) and then only referenced.
//
// What is kept exact: the gate stream is the reference's — every gate of a unit consumes its gate id at its position
// in the stream (the plan gives each call its gate-id and ciphertext offsets), and a unit recorded on its own makes the
// same dead-gate decisions as inside its parent: an output wire without credits in the parent (zero fan-out there and
// inside the component) is dead in the stand-alone recording too (StreamingRunner::set_output_liveness).
//
// The gates between units ("glue") are collected into segments; when the circuit is complete every wire that crosses a
// segment boundary becomes a global wire of the plan and each glue segment is compiled as a program of its own.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cstdlib>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <set>
#include <string>
#include <unordered_map>

#include "program.hpp"

namespace gsv {

constexpr uint32_t PLAN_WIRE_FALSE = 0xFFFFFFFEu, PLAN_WIRE_TRUE = 0xFFFFFFFFu;  // call operands that are the constants

struct PlanUnit {  // one compiled (component key, output liveness) pair
  Trace trace;
  std::vector<uint32_t> inputs, outputs;  // SSA ids inside the unit's own trace (outputs: only the produced ones)
  std::vector<int32_t> out_index;         // per component output: index into `outputs`, -1 = dead, -2 = FALSE, -3 = TRUE, -(4+k) = input k passed through
  uint64_t n_gates = 0;
  int external = -1;  // >= 0: the program was compiled by the caller (C ABI plan recorder); trace / inputs / outputs are empty
  size_t n_ext_outputs = 0;
  std::unique_ptr<Program> compiled;  // set by the background compiler (PlanRecordMode::compile_in_background) before finish_plan
  std::unique_ptr<Program> compiled_b;  // ... and the SECOND image of a dual build (PlanUnitCache::dual): the same trace compiled with bg_opt_b
};

inline size_t plan_compile_threads() {  // GSV_COMPILE_THREADS, default: the hardware's, at most 16
  size_t nt = std::thread::hardware_concurrency();
  if (const char* ev = getenv("GSV_COMPILE_THREADS")) nt = size_t(std::max(1, atoi(ev)));
  return std::min<size_t>(nt ? nt : 1, 16);
}

// A small bounded worker pool: unit programs are compiled while the driver keeps recording (the recording is serial, a
// compilation takes about as long as recording the unit).  submit() blocks while as many jobs as threads are pending, which bounds the
// traces and compiler temporaries in flight; the first exception is rethrown by wait().
class CompilePool {
 public:
  explicit CompilePool(size_t threads) {
    for (size_t i = 0; i < std::max<size_t>(1, threads); ++i)
      th_.emplace_back([this] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;
            job = std::move(q_.front());
            q_.pop_front();
            ++running_;
          }
          try { job(); } catch (...) { std::lock_guard<std::mutex> lk(mu_); if (!err_) err_ = std::current_exception(); }
          { std::lock_guard<std::mutex> lk(mu_); --running_; }
          cv_done_.notify_all();
        }
      });
  }
  ~CompilePool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void submit(std::function<void()> job) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this] { return q_.size() < th_.size(); });
    q_.push_back(std::move(job));
    cv_.notify_one();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this] { return q_.empty() && running_ == 0; });
    if (err_) { auto e = err_; err_ = nullptr; std::rethrow_exception(e); }
  }

 private:
  std::vector<std::thread> th_;
  std::deque<std::function<void()>> q_;
  std::mutex mu_;
  std::condition_variable cv_, cv_done_;
  size_t running_ = 0;
  bool stop_ = false;
  std::exception_ptr err_;
};

// The units of a plan under construction, shared by every recorder working on it: the driver's own PlanRecordMode and the warm-up
// recorders (gsv_plan_from_circuit runs a circuit's warm-up mini-circuits on other threads so that the constant-specialised units —
// the verifier has 178 line functions of 18 M gates each — are recorded side by side instead of one after the other by the driver).
// index: cache key -> unit, or -1 while some thread is recording it (the others wait instead of recording it again).
struct PlanUnitCache {
  std::mutex mu;
  std::condition_variable cv;
  std::unordered_map<std::string, int> index;
  std::vector<std::unique_ptr<PlanUnit>> units;
  // Called (on the compiling thread) with every program the builder compiles, right after it exists: gsv_plan_build_file writes the
  // records to the plan file there and drops them, so that the host never holds more than the programs still being compiled.
  std::function<void(Program&)> sink;
  CompileOptions bg_opt;
  bool bg_drop = false;
  // Dual build (round 6, gsv_plan_build_file_pair): ONE recording of the circuit feeds TWO compilations of every program — e.g. the image
  // for a quarter of the LDS window (four instances per workgroup) and the one for the full window (small batches).  Recording is the
  // critical path of a build (the driver's walk is serial; the compile pool keeps up), so the second plan costs compile time only.
  bool dual = false;
  CompileOptions bg_opt_b;
  std::function<void(Program&)> sink_b;
  // The two plans of a dual build may cut the circuit into DIFFERENT units (bench.py: Fq12-level units for full batches, Fq6-level for
  // small ones): two recorders with their own unit names share this cache, a unit is recorded once whoever meets it first, and it is
  // compiled for plan A / plan B only if its component is a unit of that plan (both lists empty: every unit for both).
  std::vector<std::string> names_a, names_b;
  static bool names_match(const std::vector<std::string>& names, const std::string& key) {
    for (const std::string& n : names)
      if (key.size() > n.size() && key.compare(0, n.size(), n) == 0 && (key[n.size()] == '#' || key[n.size()] == '|')) return true;
    return false;
  }
  std::unique_ptr<CompilePool> pool;  // set by compile_in_background; declared last: destroyed (its workers joined) before what the jobs use
};

// A glue segment in canonical form: operands are either a wire defined earlier in the same segment (its definition index:
// the SSA ids defined inside a segment are consecutive) or the k-th distinct outside wire in first-use order.  Segments
// with the same canonical form (a gadget that is not a component, repeated) share one compiled program.
struct GlueClass {
  std::vector<uint8_t> type;
  std::vector<uint32_t> a, b;     // bit 31 set: outside wire (low bits = ordinal, 0x7FFFFFFE / 0x7FFFFFFF = FALSE / TRUE); else definition index
  std::vector<uint8_t> live;      // 1 = the gate defines a wire (definition index = number of live gates before it)
  uint32_t n_inputs = 0, n_defs = 0;
  std::vector<uint8_t> out_mark;  // per definition index: read outside its segment by SOME instance
  uint64_t hash = 0;
  bool same(const GlueClass& o) const { return n_inputs == o.n_inputs && type == o.type && a == o.a && b == o.b && live == o.live; }
};
struct PlanSegment {
  int unit = -1;                      // >= 0: call of units[unit]; -1: glue
  std::vector<uint32_t> in_ssa;       // global SSA id (or PLAN_WIRE_*) per unit input / per outside wire of a glue segment
  std::vector<uint32_t> out_ssa;      // unit: global SSA id per produced output
  Trace glue;                         // glue being recorded: gates with GLOBAL SSA ids; emptied when the segment is closed
  int glue_class = -1;                // glue: index into PlanRecordMode::glue_classes
  uint32_t base_ssa = 0;              // glue: SSA id of its first definition
  uint64_t n_gates = 0;
};

class PlanRecordMode final : public CircuitMode, public UnitHook {
 public:
  explicit PlanRecordMode(std::vector<std::string> unit_names, std::shared_ptr<PlanUnitCache> cache = nullptr)
      : cache_(cache ? std::move(cache) : std::make_shared<PlanUnitCache>()), units(cache_->units), unit_names_(std::move(unit_names)) {
    ver_.assign(2, 0); ver_[1] = 1; written_.assign(2, 1);
  }
  const std::shared_ptr<PlanUnitCache>& cache() const { return cache_; }
  const std::vector<std::string>& unit_names() const { return unit_names_; }

  // Compile every unit as soon as it has been recorded, on a worker pool, with the options finish_plan will be given.
  // drop_traces: free a unit's trace once its program exists (no other variant of the program will be compiled).
  void compile_in_background(const CompileOptions& opt, bool drop_traces) {
    cache_->bg_opt = opt; cache_->bg_drop = drop_traces;
    cache_->pool.reset(new CompilePool(plan_compile_threads()));
  }
  void wait_for_compilations() { if (cache_->pool) cache_->pool->wait(); }

  // ---- CircuitMode (same rules as RecordMode, program.hpp)
  WireId allocate_wire(Credits credits) override {
    if (credits == 0) return UNREACHABLE;
    WireId id = ver_.size();
    ver_.push_back(0);
    written_.push_back(0);
    return id;
  }
  void evaluate_gate(const Gate& g) override {
    uint32_t ra = read(g.a), rb = read(g.b);
    Trace& t = glue();
    t.type.push_back(uint8_t(g.t));
    t.a.push_back(ra);
    t.b.push_back(rb);
    ++n_gates_;
    if (g.c == UNREACHABLE) { t.c.push_back(DEAD_WIRE); return; }
    if (g.c == FALSE_WIRE || g.c == TRUE_WIRE) gsv_panic("gate output is a constant wire");
    t.c.push_back(define(g.c));
  }
  bool consume_wire(WireId w) override { return w < ver_.size() && (w < 2 || written_[size_t(w)]); }
  void add_credits(const WireId*, size_t, Credits) override {}
  UnitHook* unit_hook() override { return this; }

  uint32_t define_input(WireId w) {
    if (w == UNREACHABLE) gsv_panic("input wire has zero fan-out and no root credit");
    return define(w);
  }
  uint32_t current(WireId w) { return read(w); }

  // ---- UnitHook
  bool call_unit(const ComponentKey& key, const Wires& inputs, const std::vector<Credits>& out_credits, const ComponentMetaTemplate& tpl, const ChildFn& body,
                 size_t arity, Wires& out) override {
    if (!is_unit(key)) return false;
    using Out = ComponentMetaTemplate::Out;
    // liveness of every output wire: credits inside the component + credits in the parent
    std::vector<uint8_t> live(arity, 0);
    std::string cache_key = key;
    cache_key.push_back('!');
    for (size_t i = 0; i < arity; ++i) {
      live[i] = out_credits[i] != 0;
      cache_key.push_back(live[i] ? '1' : '0');
    }
    PlanUnitCache& uc = *cache_;
    int unit_id = -1;
    {
      std::unique_lock<std::mutex> lk(uc.mu);
      for (;;) {
        auto it = uc.index.find(cache_key);
        if (it == uc.index.end()) { uc.index.emplace(cache_key, -1); break; }  // ours to record
        if (it->second >= 0) { unit_id = it->second; break; }
        uc.cv.wait(lk);  // another recorder is at it
      }
    }
    if (unit_id < 0) try {
      // record the component on its own (nested components are flattened into it)
      RecordMode rec;
      StreamingRunner run(rec, inputs.size(), body);
      run.set_output_liveness(live);
      auto u = std::make_unique<PlanUnit>();
      const Wires& in = run.prepare();
      for (WireId w : in) u->inputs.push_back(rec.define_input(w));
      const Wires& o = run.execute();
      if (o.size() != arity) gsv_panic("unit returned wrong arity");
      for (size_t i = 0; i < arity; ++i) {
        const WireId w = o[i];
        if (w == UNREACHABLE) { u->out_index.push_back(-1); continue; }
        if (w == FALSE_WIRE) { u->out_index.push_back(-2); continue; }
        if (w == TRUE_WIRE) { u->out_index.push_back(-3); continue; }
        bool passed = false;
        for (size_t k = 0; k < in.size(); ++k) if (in[k] == w) { u->out_index.push_back(-int32_t(4 + k)); passed = true; break; }
        if (passed) continue;
        u->out_index.push_back(int32_t(u->outputs.size()));
        u->outputs.push_back(rec.current(w));
      }
      u->trace = std::move(rec.trace());
      u->n_gates = u->trace.size();
      PlanUnit* pu = u.get();  // stable: the vector holds pointers; nothing else touches the trace before finish_plan
      {
        std::lock_guard<std::mutex> lk(uc.mu);
        unit_id = int(uc.units.size());
        uc.units.push_back(std::move(u));
        uc.index[cache_key] = unit_id;
      }
      uc.cv.notify_all();
      if (uc.pool) {
        const CompileOptions opt = uc.bg_opt;
        const bool drop = uc.bg_drop;
        PlanUnitCache* ucp = &uc;  // outlives the pool it owns
        const bool split = uc.dual && !(uc.names_a.empty() && uc.names_b.empty());
        const bool need_a = !split || PlanUnitCache::names_match(uc.names_a, key), need_b = uc.dual && (!split || PlanUnitCache::names_match(uc.names_b, key));
        uc.pool->submit([pu, opt, drop, ucp, need_a, need_b] {
          if (need_a) {
            pu->compiled.reset(new Program(compile_program(pu->trace, pu->inputs, pu->outputs, {}, opt)));
            if (ucp->sink) ucp->sink(*pu->compiled);
          }
          if (need_b) {
            pu->compiled_b.reset(new Program(compile_program(pu->trace, pu->inputs, pu->outputs, {}, ucp->bg_opt_b)));
            if (ucp->sink_b) ucp->sink_b(*pu->compiled_b);
          }
          if (drop) pu->trace = Trace();
        });
      }
    } catch (...) {
      { std::lock_guard<std::mutex> lk(uc.mu); auto it = uc.index.find(cache_key); if (it != uc.index.end() && it->second < 0) uc.index.erase(it); }
      uc.cv.notify_all();
      throw;
    }
    const PlanUnit* up;
    { std::lock_guard<std::mutex> lk(uc.mu); up = uc.units[size_t(unit_id)].get(); }  // the vector may grow under another recorder
    const PlanUnit& u = *up;
    (void)tpl; (void)Out::Internal;
    PlanSegment seg;
    seg.unit = unit_id;
    for (WireId w : inputs) seg.in_ssa.push_back(w == FALSE_WIRE ? PLAN_WIRE_FALSE : w == TRUE_WIRE ? PLAN_WIRE_TRUE : read(w));
    seg.out_ssa.assign(u.outputs.size(), DEAD_WIRE);
    out.assign(arity, UNREACHABLE);
    for (size_t i = 0; i < arity; ++i) {
      const int32_t oi = u.out_index[i];
      if (oi == -1) continue;
      if (oi == -2) { out[i] = FALSE_WIRE; continue; }
      if (oi == -3) { out[i] = TRUE_WIRE; continue; }
      if (oi <= -4) { out[i] = inputs[size_t(-oi - 4)]; continue; }
      const WireId w = allocate_wire(1);  // the unit produced it: it exists in the parent whatever its remaining credits
      seg.out_ssa[size_t(oi)] = define(w);
      out[i] = w;
    }
    n_gates_ += u.n_gates;
    close_glue();
    segments.push_back(std::move(seg));
    return true;
  }
  // Canonicalise the glue segment under construction (if any) and file it under its class.
  void close_glue() {
    if (segments.empty() || segments.back().unit >= 0 || segments.back().glue_class >= 0) return;
    PlanSegment& s = segments.back();
    const Trace& t = s.glue;
    GlueClass g;
    const size_t n = t.size();
    g.type = t.type; g.a.resize(n); g.b.resize(n); g.live.resize(n);
    uint32_t base = 0;
    bool have_base = false;
    for (size_t i = 0; i < n; ++i) if (t.c[i] != DEAD_WIRE) { base = t.c[i]; have_base = true; break; }
    std::unordered_map<uint32_t, uint32_t> ordinal;
    uint32_t defs = 0;
    uint64_t h = 1469598103934665603ull;
    auto mixh = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    auto enc = [&](uint32_t w) -> uint32_t {
      if (w == 0) return 0xFFFFFFFEu;
      if (w == 1) return 0xFFFFFFFFu;
      if (have_base && w >= base && w < base + defs) return w - base;  // defined earlier in this segment
      auto f = ordinal.find(w);
      if (f != ordinal.end()) return 0x80000000u | f->second;
      const uint32_t k = uint32_t(s.in_ssa.size());
      ordinal.emplace(w, k);
      s.in_ssa.push_back(w);
      return 0x80000000u | k;
    };
    for (size_t i = 0; i < n; ++i) {
      g.a[i] = enc(t.a[i]); g.b[i] = enc(t.b[i]);
      g.live[i] = t.c[i] != DEAD_WIRE;
      if (g.live[i]) { if (t.c[i] != base + defs) gsv_panic("internal: glue definitions are not consecutive"); ++defs; }
      mixh(g.type[i]); mixh(g.a[i]); mixh(g.b[i]); mixh(g.live[i]);
    }
    g.n_inputs = uint32_t(s.in_ssa.size()); g.n_defs = defs; g.hash = h;
    g.out_mark.assign(defs, 0);
    s.base_ssa = base; s.n_gates = n;
    auto range = class_index_.equal_range(h);
    for (auto it = range.first; it != range.second; ++it)
      if (glue_classes[size_t(it->second)]->same(g)) { s.glue_class = it->second; break; }
    if (s.glue_class < 0) {
      s.glue_class = int(glue_classes.size());
      class_index_.emplace(h, s.glue_class);
      glue_classes.push_back(std::make_unique<GlueClass>(std::move(g)));
    }
    s.glue = Trace();  // the canonical form (kept once per class) is all that is needed from here on
  }

  // C ABI route (gsv_plan_recorder_*): the host recorded and compiled the unit itself; a call hands over parent wires
  // and receives one fresh parent wire per program output.
  int add_external_unit(int external_index, uint64_t n_gates, size_t n_outputs) {
    auto u = std::make_unique<PlanUnit>();
    u->external = external_index; u->n_gates = n_gates; u->n_ext_outputs = n_outputs;
    units.push_back(std::move(u));
    return int(units.size()) - 1;
  }
  void call_external(int unit, const Wires& inputs, Wires& out) {
    const PlanUnit& u = *units[size_t(unit)];
    PlanSegment seg;
    seg.unit = unit;
    for (WireId w : inputs) seg.in_ssa.push_back(w == FALSE_WIRE ? PLAN_WIRE_FALSE : w == TRUE_WIRE ? PLAN_WIRE_TRUE : read(w));
    out.clear();
    for (size_t i = 0; i < u.n_ext_outputs; ++i) {
      const WireId w = allocate_wire(1);
      seg.out_ssa.push_back(define(w));
      out.push_back(w);
    }
    n_gates_ += u.n_gates;
    close_glue();
    segments.push_back(std::move(seg));
  }

  uint64_t n_gates() const { return n_gates_; }
  uint32_t n_ssa() const { return next_ssa_; }
 private:
  std::shared_ptr<PlanUnitCache> cache_;
 public:
  std::vector<std::unique_ptr<PlanUnit>>& units;  // the cache's (single-threaded use only: after the warm-up recorders have been joined)
  std::vector<PlanSegment> segments;
  std::vector<std::unique_ptr<GlueClass>> glue_classes;

 private:
  bool is_unit(const ComponentKey& key) const {
    for (const std::string& n : unit_names_)
      if (key.size() > n.size() && key.compare(0, n.size(), n) == 0 && (key[n.size()] == '#' || key[n.size()] == '|')) return true;
    return false;
  }
  Trace& glue() {
    if (segments.empty() || segments.back().unit >= 0 || segments.back().glue_class >= 0) segments.emplace_back();
    return segments.back().glue;
  }
  uint32_t read(WireId w) {
    if (w == FALSE_WIRE) return 0;
    if (w == TRUE_WIRE) return 1;
    if (w >= ver_.size()) gsv_panic("PlanRecordMode: read of unknown wire");
    if (!written_[size_t(w)]) gsv_panic("PlanRecordMode: wire read before it was written");
    return ver_[size_t(w)];
  }
  uint32_t define(WireId w) {
    if (w >= ver_.size() || w < 2) gsv_panic("PlanRecordMode: write to unknown wire");
    if (next_ssa_ >= 0xFFFFFFF0u) gsv_panic("PlanRecordMode: SSA id overflow");
    const uint32_t id = next_ssa_++;
    ver_[size_t(w)] = id;
    written_[size_t(w)] = 1;
    return id;
  }
  std::vector<std::string> unit_names_;
  std::unordered_multimap<uint64_t, int> class_index_;
  std::vector<uint32_t> ver_;
  std::vector<uint8_t> written_;
  uint32_t next_ssa_ = 2;  // 0 / 1 are the constants
  uint64_t n_gates_ = 0;
};

// The recording itself: the two-pass driver walks (n_inputs, fn) on this thread while up to a quarter of the compile threads
// (GSV_PLAN_WARMUP_THREADS, 0 = none) run the circuit's warm-up mini-circuits, each under a recorder of its own that shares `mode`'s
// unit cache: a unit the warm-ups reach first is recorded (and sent to the compile pool) by them, the driver waits for one that is
// still being recorded and records what the warm-ups do not cover.  The plan is the same with or without them.
// Warmup: anything with .n_inputs and .fn (NamedCircuit::Warmup).  in_ssa / out_ssa: the circuit's inputs / outputs for finish_plan.
template <class Warmup>
inline void record_plan(PlanRecordMode& mode, size_t n_inputs, const CircuitFn& fn, const std::vector<Warmup>& warmups, std::vector<uint32_t>& in_ssa,
                        std::vector<uint32_t>& out_ssa, size_t* n_recorders_out = nullptr) {
  std::atomic<size_t> next{0};
  const size_t n = warmups.size();
  struct Crew {
    std::vector<std::thread> th;
    std::atomic<size_t>& next; size_t n;
    ~Crew() { next.store(n); for (auto& t : th) if (t.joinable()) t.join(); }  // (the driver threw: stop handing out warm-ups)
  } crew{{}, next, n};
  std::mutex err_mu;
  std::exception_ptr err;
  size_t nrec = std::max<size_t>(1, plan_compile_threads() / 4);
  if (const char* e = getenv("GSV_PLAN_WARMUP_THREADS")) nrec = size_t(std::max(0, atoi(e)));
  nrec = std::min(nrec, n);
  if (n_recorders_out) *n_recorders_out = nrec;
  for (size_t t = 0; t < nrec; ++t)
    crew.th.emplace_back([&] {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n) return;
        try {
          PlanRecordMode wm(mode.unit_names(), mode.cache());
          StreamingRunner wrun(wm, warmups[i].n_inputs, warmups[i].fn);
          for (WireId w : wrun.prepare()) wm.define_input(w);
          (void)wrun.execute();
        } catch (...) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (!err) err = std::current_exception();
          next.store(n);
          return;
        }
      }
    });
  StreamingRunner run(mode, n_inputs, fn);
  for (WireId w : run.prepare()) in_ssa.push_back(mode.define_input(w));
  for (WireId w : run.execute()) out_ssa.push_back(mode.current(w));
  for (auto& t : crew.th) t.join();
  if (err) std::rethrow_exception(err);
}

// The finished plan in host form: programs (units first, then one per glue segment) and calls over global wire ids.
// Runs fn(0 .. n-1) on up to GSV_COMPILE_THREADS (default: the hardware's, at most 16) threads; the first exception is rethrown.
// Programs of a plan are compiled independently of each other (the Miller loop alone has ~190 of them).
template <class Fn>
inline void parallel_for_programs(size_t n, Fn&& fn) {
  const size_t nt = std::min<size_t>(plan_compile_threads(), n);
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<size_t> next{0};
  std::exception_ptr err;
  std::mutex mu;
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t)
    th.emplace_back([&] {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n) return;
        try { fn(i); } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!err) err = std::current_exception(); next.store(n); return; }
      }
    });
  for (auto& t : th) t.join();
  if (err) std::rethrow_exception(err);
}

struct BuiltPlan {
  struct Call { int program; std::vector<uint32_t> in_globals, out_globals; };  // program < 0: external program -1 - program
  std::vector<Program> programs;
  std::vector<Program> programs_b;  // dual build: the second image of every program (same order), else empty
  std::vector<Trace> traces;  // kept per program so that the half-window variants can be compiled later
  std::vector<std::vector<uint32_t>> prog_inputs, prog_outputs;
  std::vector<Call> calls;
  uint32_t n_inputs = 0;
  std::vector<uint32_t> outputs;  // global ids (or PLAN_WIRE_*)
  uint64_t n_gates = 0;
  uint32_t n_global_ids = 0;      // recycled pool size (without the trash ids behind it)
  uint32_t n_crossing_wires = 0;  // wires that cross calls (what the pool would be without recycling)
};

// inputs / outputs: global SSA ids of the circuit's inputs / outputs as PlanRecordMode handed them out.
// which (dual builds, PlanUnitCache::dual): 0 = the first plan's images (or a plain build), 1 = the SECOND plan's images (compiled_b,
// bg_opt_b, sink_b — they land in BuiltPlan::programs), 2 = both plans from this one recorder (programs and programs_b).
inline BuiltPlan finish_plan(PlanRecordMode& m, const std::vector<uint32_t>& inputs, const std::vector<uint32_t>& outputs, const CompileOptions& opt = CompileOptions(), int which = 0) {
  m.close_glue();
  m.wait_for_compilations();
  BuiltPlan bp;
  if (which != 0 && !m.cache()->dual) gsv_panic("internal: finish_plan asked for the second image of a build that is not dual");
  const bool dual = which == 2;
  const uint32_t nw = m.n_ssa();
  constexpr int32_t SEG_INPUT = -1, SEG_NONE = -2;
  std::vector<int32_t> def_seg(nw, SEG_NONE);
  def_seg[0] = def_seg[1] = SEG_INPUT;
  for (uint32_t w : inputs) def_seg[w] = SEG_INPUT;
  for (size_t si = 0; si < m.segments.size(); ++si) {
    const PlanSegment& s = m.segments[si];
    if (s.unit >= 0) { for (uint32_t w : s.out_ssa) if (w != DEAD_WIRE) def_seg[w] = int32_t(si); }
    else { const uint32_t nd = m.glue_classes[size_t(s.glue_class)]->n_defs; for (uint32_t d = 0; d < nd; ++d) def_seg[s.base_ssa + d] = int32_t(si); }
  }
  // wires read outside the segment that defines them (or circuit inputs / outputs) become globals; a glue segment's
  // outside wires are exactly its in_ssa list
  std::vector<uint8_t> crossing(nw, 0);
  for (uint32_t w : inputs) crossing[w] = 1;
  for (uint32_t w : outputs) if (w > 1) crossing[w] = 1;
  for (const PlanSegment& s : m.segments) for (uint32_t w : s.in_ssa) if (w < PLAN_WIRE_FALSE && w > 1) crossing[w] = 1;
  // a glue class exports a definition if ANY of its instances has that wire read elsewhere
  for (const PlanSegment& s : m.segments) {
    if (s.unit >= 0) continue;
    GlueClass& g = *m.glue_classes[size_t(s.glue_class)];
    for (uint32_t d = 0; d < g.n_defs; ++d) if (crossing[s.base_ssa + d]) g.out_mark[d] = 1;
  }
  // Global wire ids with RECYCLING: a global is live from the call that writes it to the last call that reads it (circuit inputs
  // and outputs are pinned), and its id returns to a free list after that call — the verifier's plan crosses ~2 x 10^7 wires
  // between calls, of which only a small fraction is alive at any point (the reference's Storage does the same with credits,
  // storage.rs:119-198).  A call's pre-copy (globals -> program inputs) runs before its kernel and its post-copy (program outputs
  // -> globals) after it, so an id freed by the call's own last read may be handed to one of its outputs.
  constexpr int32_t NEVER_FREED = 0x7FFFFFFF;
  std::vector<int32_t> last_use(nw, -1);
  for (size_t si = 0; si < m.segments.size(); ++si)
    for (uint32_t w : m.segments[si].in_ssa) if (w < PLAN_WIRE_FALSE && w > 1) last_use[w] = int32_t(si);
  for (uint32_t w : inputs) last_use[w] = NEVER_FREED;
  for (uint32_t w : outputs) if (w > 1) last_use[w] = NEVER_FREED;
  std::vector<uint32_t> global_of(nw, DEAD_WIRE);
  uint32_t next_global = 0;
  for (uint32_t w : inputs) global_of[w] = next_global++;
  bp.n_inputs = next_global;
  // Recycling is FIFO with slack: a freed id is handed out again only once `slack` younger ids wait behind it.  A session runs
  // independent calls side by side (schedule.hpp) and sees global ids as memory locations: an id reused right away would chain its
  // new writer behind every reader of the old value (WAR) although the two calls have nothing to do with each other; with the
  // queue the reuse distance is slack / (outputs per call) calls, beyond any scheduling window.  Cost: slack x 16 bytes per instance
  // (GSV_PLAN_ID_SLACK, default 262 144 ids = 4 MB per instance: 4.3 GB of the 205 GB a 1 024-instance session takes.  The plan is
  // built once and serves sessions of every concurrency, so the slack is not derived from one session's in-flight bound; a deployment
  // that only runs sequential sessions — max_concurrent_calls = 1, where reuse distance buys nothing — builds its plan with
  // GSV_PLAN_ID_SLACK=0 and gets the smallest wire file).
  std::deque<uint32_t> free_ids;
  size_t id_slack = 262144;
  if (const char* ev = getenv("GSV_PLAN_ID_SLACK")) {
    char* end = nullptr;
    const long long v = std::strtoll(ev, &end, 10);
    if (end == ev || *end != 0 || v < 0 || v > (1ll << 28)) gsv_panic("GSV_PLAN_ID_SLACK must be an integer in [0, 2^28]");
    id_slack = size_t(v);
  }
  auto alloc_global = [&]() -> uint32_t { if (free_ids.size() > id_slack) { uint32_t g = free_ids.front(); free_ids.pop_front(); return g; } return next_global++; };
  // released after segment si has read its inputs; each wire once even when a call names it several times
  auto release_dead_inputs = [&](size_t si) {
    for (uint32_t w : m.segments[si].in_ssa)
      if (w < PLAN_WIRE_FALSE && w > 1 && last_use[w] == int32_t(si) && global_of[w] != DEAD_WIRE) { free_ids.push_back(global_of[w]); last_use[w] = -2; }
  };
  constexpr uint32_t TRASH_FLAG = 0x80000000u;  // patched to ids behind the pool's high-water mark once that is known
  std::vector<uint8_t> done;
  auto add_program = [&](Trace&& t, std::vector<uint32_t> in, std::vector<uint32_t> out) -> int {
    bp.programs.emplace_back();  // compiled below, all programs in parallel (units may have been compiled while recording)
    if (dual) bp.programs_b.emplace_back();
    done.push_back(0);
    bp.traces.push_back(std::move(t));
    bp.prog_inputs.push_back(std::move(in));
    bp.prog_outputs.push_back(std::move(out));
    return int(bp.programs.size()) - 1;
  };
  std::vector<int> unit_program(m.units.size(), -1), class_program(m.glue_classes.size(), -1);
  std::vector<std::vector<uint32_t>> class_outputs(m.glue_classes.size());  // exported definition indices
  for (size_t si = 0; si < m.segments.size(); ++si) {
    const PlanSegment& s = m.segments[si];
    BuiltPlan::Call call;
    if (s.unit >= 0) {
      PlanUnit& u = *m.units[size_t(s.unit)];
      if (u.external >= 0) {
        call.program = -1 - u.external;  // the caller's own program
      } else {
        if (unit_program[size_t(s.unit)] < 0) {
          // later calls of the unit reuse the program index, so the trace can move (178 constant-specialised line
          // functions of the Miller loop would otherwise exist twice)
          // (a dual build's cache is shared by two recorders: the trace stays where it is — it has been dropped after the compilations anyway)
          unit_program[size_t(s.unit)] = m.cache()->dual ? add_program(Trace(u.trace), u.inputs, u.outputs) : add_program(std::move(u.trace), u.inputs, u.outputs);
          if (which == 1) {
            if (u.compiled_b) { bp.programs.back() = std::move(*u.compiled_b); u.compiled_b.reset(); done.back() = 1; }
          } else if (u.compiled && (!dual || u.compiled_b)) {
            bp.programs.back() = std::move(*u.compiled); u.compiled.reset();
            if (dual) { bp.programs_b.back() = std::move(*u.compiled_b); u.compiled_b.reset(); }
            done.back() = 1;
          }
        }
        call.program = unit_program[size_t(s.unit)];
      }
      for (uint32_t w : s.in_ssa) call.in_globals.push_back(w == PLAN_WIRE_FALSE || w == 0 ? PLAN_WIRE_FALSE : w == PLAN_WIRE_TRUE || w == 1 ? PLAN_WIRE_TRUE : global_of[w]);
      release_dead_inputs(si);
      uint32_t t_used = 0;
      for (uint32_t w : s.out_ssa) {
        if (w != DEAD_WIRE && crossing[w]) { if (global_of[w] == DEAD_WIRE) global_of[w] = alloc_global(); call.out_globals.push_back(global_of[w]); }
        else call.out_globals.push_back(TRASH_FLAG | t_used++);
      }
      bp.n_gates += u.n_gates;
    } else {
      const size_t ci = size_t(s.glue_class);
      const GlueClass& g = *m.glue_classes[ci];
      if (class_program[ci] < 0) {
        // dense local trace: 0 / 1 constants, inputs 2 .. 2+n_inputs-1 (first-use order), definitions behind them
        Trace lt;
        lt.n_wires = 2 + g.n_inputs + g.n_defs;
        auto dec = [&](uint32_t e) -> uint32_t {
          if (e == 0xFFFFFFFEu) return 0;
          if (e == 0xFFFFFFFFu) return 1;
          return (e & 0x80000000u) ? 2 + (e & 0x7FFFFFFFu) : 2 + g.n_inputs + e;
        };
        uint32_t d = 0;
        for (size_t i = 0; i < g.type.size(); ++i) {
          lt.type.push_back(g.type[i]); lt.a.push_back(dec(g.a[i])); lt.b.push_back(dec(g.b[i]));
          lt.c.push_back(g.live[i] ? 2 + g.n_inputs + d++ : DEAD_WIRE);
        }
        std::vector<uint32_t> in_local, out_local;
        for (uint32_t k = 0; k < g.n_inputs; ++k) in_local.push_back(2 + k);
        for (uint32_t k = 0; k < g.n_defs; ++k) if (g.out_mark[k]) { out_local.push_back(2 + g.n_inputs + k); class_outputs[ci].push_back(k); }
        class_program[ci] = add_program(std::move(lt), in_local, out_local);
      }
      call.program = class_program[ci];
      for (uint32_t w : s.in_ssa) call.in_globals.push_back(global_of[w]);
      release_dead_inputs(si);
      uint32_t t_used = 0;
      for (uint32_t k : class_outputs[ci]) {
        const uint32_t w = s.base_ssa + k;
        if (crossing[w]) { if (global_of[w] == DEAD_WIRE) global_of[w] = alloc_global(); call.out_globals.push_back(global_of[w]); }
        else call.out_globals.push_back(TRASH_FLAG | t_used++);
      }
      bp.n_gates += s.n_gates;
    }
    bp.calls.push_back(std::move(call));
  }
  // outputs nobody reads in their instance still have to land somewhere: a few ids behind the recycled pool
  for (BuiltPlan::Call& c : bp.calls) for (uint32_t& g : c.out_globals) if (g < PLAN_WIRE_FALSE && (g & TRASH_FLAG)) g = next_global + (g & ~TRASH_FLAG);
  bp.n_global_ids = next_global;
  for (uint32_t w = 2; w < nw; ++w) if (crossing[w]) ++bp.n_crossing_wires;
  if (getenv("GSV_PLAN_DEBUG")) std::fprintf(stderr, "plan: %zu segments, %u wires cross calls, %u global ids after recycling (%u inputs pinned)\n", m.segments.size(), bp.n_crossing_wires, next_global, bp.n_inputs);
  for (uint32_t w : outputs) bp.outputs.push_back(w == 0 ? PLAN_WIRE_FALSE : w == 1 ? PLAN_WIRE_TRUE : global_of[w]);
  const std::function<void(Program&)>& sink = m.cache()->sink;
  const std::function<void(Program&)>& sink_b = m.cache()->sink_b;
  const CompileOptions opt_b = m.cache()->bg_opt_b;
  parallel_for_programs(bp.programs.size(), [&](size_t i) {
    if (done[i]) return;
    if (which == 1) {
      bp.programs[i] = compile_program(bp.traces[i], bp.prog_inputs[i], bp.prog_outputs[i], {}, opt_b);
      if (sink_b) { bp.traces[i] = Trace(); sink_b(bp.programs[i]); }
      return;
    }
    bp.programs[i] = compile_program(bp.traces[i], bp.prog_inputs[i], bp.prog_outputs[i], {}, opt);
    if (dual) {
      bp.programs_b[i] = compile_program(bp.traces[i], bp.prog_inputs[i], bp.prog_outputs[i], {}, opt_b);
      if (sink_b) sink_b(bp.programs_b[i]);
    }
    if (sink) { bp.traces[i] = Trace(); sink(bp.programs[i]); }
  });
  return bp;
}

}  // namespace gsv
