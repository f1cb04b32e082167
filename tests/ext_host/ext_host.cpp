// TEST-ONLY: an EXTERNAL host of libgsv_engine.so, the way a maintainer's Rust host would drive it.
//
// Links ONLY libgsv_engine.so and includes ONLY include/gsv_engine.h from the engine — no engine header (program.hpp, plan_builder.hpp,
// schedule.hpp) and none of its classes.  What stands in for the reference's host here, C++ by necessity (no Rust toolchain):
//   * the two-pass driver and the gadgets are the shared, mode-generic headers csrc/circuit/circuit.hpp + csrc/gadgets/*.hpp — the
//     restatement of the reference's layers ABOVE the CircuitMode seam (src/circuit/streaming_mode.rs, src/gadgets/**), identical for
//     every mode, exactly as in the reference;
//   * `AbiRecordMode` / `AbiPlanMode` are `impl CircuitMode` (src/circuit/modes.rs:26-51) over the C ABI: the counterpart of
//     bindings/rust/src/gpu_garble_mode.rs.  allocate_wire / evaluate_gate forward to gsv_recorder_* / gsv_plan_recorder_*;
//   * `AbiPlanMode::call_unit` is the `with_named_child` unit hook (streaming_mode.rs:189-241, bindings/rust/streaming_mode_unit_hook.patch):
//     a unit component is recorded ON ITS OWN once per (ComponentKey, liveness of its outputs) through gsv_recorder_*, compiled with
//     gsv_program_compile_opts (background, for the plan recorder) and afterwards only referenced by gsv_plan_recorder_call.
// The finished plan goes to a plan file (gsv_plan_recorder_opts.plan_file), is loaded into the GPU (gsv_plan_load) and — `--garble SEED` —
// one instance is garbled with nothing retained on the device, the ciphertext stream handed to a sink of this program
// (gsv_session_garble_streaming_sink: CiphertextHandler::handle, circuit/mod.rs:140-178) which folds it into the CBC-MAC
// (AESAccumulatingHash, ciphertext_hasher.rs:23-29).  Prints one JSON line; tests/test_ext_host.py compares it with the oracle's fixtures
// and the plan file with the one gsv_plan_build_file writes.
//
// usage: ext_host <circuit spec> <unit names, comma separated> <plan file> [--window-div N] [--warmup-threads N] [--garble SEED]
#include <sys/resource.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/gsv_engine.h"
#include "../../garbled_snark_verifier_amd/csrc/gadgets/circuits.hpp"  // driver + gadgets (reference layers L2 / L3); includes nothing of the engine

using namespace gsv;

static void chk(int rc, const char* what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + gsv_last_error());
}
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr size_t FLUSH = 1 << 16;     // gates per gsv_*_push_gates call
constexpr size_t WIRE_BLOCK = 4096;   // wires per gsv_*_allocate_wires call

// Gates and wires of ONE recording (flat recorder or plan recorder), batched; the recorder is told about wires a block at a time and hands
// out consecutive ids, so the ids are the ones per-wire calls would have returned.
template <class Rec>
struct RecorderIo {
  Rec* r = nullptr;
  int (*alloc)(Rec*, size_t, uint64_t*) = nullptr;
  int (*push)(Rec*, const gsv_gate*, size_t) = nullptr;
  std::vector<gsv_gate> pending;
  uint64_t next = 0, end = 0;
  std::vector<uint8_t> written;  // by wire id: CircuitMode::lookup_wire of a wire nobody wrote is None (the driver unwraps it: panic)
  WireId allocate(Credits credits) {
    if (credits == 0) return UNREACHABLE;  // storage.rs:119-133
    if (next == end) {
      chk(alloc(r, WIRE_BLOCK, &next), "allocate_wires");
      end = next + WIRE_BLOCK;
      if (written.size() < end) written.resize(size_t(end), 0);
    }
    return next++;
  }
  void gate(const Gate& g) {
    pending.push_back(gsv_gate{g.a, g.b, g.c, uint8_t(g.t), {0, 0, 0, 0, 0, 0, 0}});
    if (g.c != UNREACHABLE && g.c < written.size()) written[size_t(g.c)] = 1;
    if (pending.size() >= FLUSH) flush();
  }
  void flush() {
    if (pending.empty()) return;
    chk(push(r, pending.data(), pending.size()), "push_gates");
    pending.clear();
  }
  bool has(WireId w) const { return w < 2 || (w < written.size() && written[size_t(w)]); }
  void mark(WireId w) { if (w < written.size()) written[size_t(w)] = 1; }
};

// `impl CircuitMode` over the flat recorder: the body of ONE unit component.
class AbiRecordMode final : public CircuitMode {
 public:
  AbiRecordMode() {
    chk(gsv_recorder_create(&io_.r), "gsv_recorder_create");
    io_.alloc = gsv_recorder_allocate_wires;
    io_.push = gsv_recorder_push_gates;
  }
  ~AbiRecordMode() override { gsv_recorder_destroy(io_.r); }
  WireId allocate_wire(Credits c) override { return io_.allocate(c); }
  void evaluate_gate(const Gate& g) override { io_.gate(g); }
  bool consume_wire(WireId w) override { return io_.has(w); }
  void add_credits(const WireId*, size_t, Credits) override {}  // the recorder keeps SSA wires: credits only decide dead gates, in allocate_wire
  void declare_input(WireId w) {
    if (w == UNREACHABLE) gsv_panic("input wire has zero fan-out and no root credit");
    chk(gsv_recorder_declare_input(io_.r, w), "gsv_recorder_declare_input");
    io_.mark(w);
  }
  gsv_program* compile(const Wires& outputs, const gsv_compile_opts& opts) {
    io_.flush();
    chk(gsv_recorder_declare_outputs(io_.r, outputs.data(), outputs.size()), "gsv_recorder_declare_outputs");
    gsv_program* p = nullptr;
    chk(gsv_program_compile_opts(io_.r, &opts, &p), "gsv_program_compile_opts");
    return p;
  }
  uint64_t n_gates() {
    io_.flush();
    uint64_t n = 0;
    chk(gsv_recorder_counts(io_.r, nullptr, nullptr, &n), "gsv_recorder_counts");
    return n;
  }

 private:
  RecorderIo<gsv_recorder> io_;
};

// One compiled (ComponentKey, output liveness) pair — the host's side of a unit.
struct Unit {
  gsv_program* program = nullptr;
  std::vector<int32_t> out_index;  // per component output: index into the program's outputs, -1 = dead, -2 = FALSE, -3 = TRUE, -(4+k) = input k passed through
  size_t n_program_outputs = 0;
};
struct UnitCache {  // shared by the driver's mode and the warm-up recorders
  std::mutex mu;
  std::condition_variable cv;
  std::unordered_map<std::string, int> index;  // -1: some thread is recording it
  std::vector<std::unique_ptr<Unit>> units;
  ~UnitCache() { for (auto& u : units) if (u->program) gsv_program_destroy(u->program); }
};

// `impl CircuitMode` + the with_named_child unit hook over the plan recorder.  `collect == false`: a warm-up recorder — it only fills the
// unit cache (its own gates and calls go nowhere), as the engine's built-in plan builder does with NamedCircuit::warmups.
class AbiPlanMode final : public CircuitMode, public UnitHook {
 public:
  AbiPlanMode(gsv_plan_recorder* pr, bool collect, std::vector<std::string> units, std::shared_ptr<UnitCache> cache)
      : pr_(pr), collect_(collect), unit_names_(std::move(units)), cache_(std::move(cache)) {
    io_.r = pr;
    io_.alloc = gsv_plan_recorder_allocate_wires;
    io_.push = gsv_plan_recorder_push_gates;
  }
  WireId allocate_wire(Credits c) override {
    if (collect_) return io_.allocate(c);
    return c == 0 ? UNREACHABLE : local_next_++;
  }
  void evaluate_gate(const Gate& g) override { if (collect_) io_.gate(g); }
  bool consume_wire(WireId w) override { return collect_ ? io_.has(w) : true; }
  void add_credits(const WireId*, size_t, Credits) override {}
  UnitHook* unit_hook() override { return this; }
  void declare_input(WireId w) {
    if (!collect_) return;
    if (w == UNREACHABLE) gsv_panic("input wire has zero fan-out and no root credit");
    chk(gsv_plan_recorder_declare_input(pr_, w), "gsv_plan_recorder_declare_input");
    io_.mark(w);
  }
  void flush() { if (collect_) io_.flush(); }

  bool call_unit(const ComponentKey& key, const Wires& inputs, const std::vector<Credits>& out_credits, const ComponentMetaTemplate&, const ChildFn& body, size_t arity,
                 Wires& out) override {
    if (!is_unit(key)) return false;
    std::vector<uint8_t> live(arity, 0);
    std::string ck = key;
    ck.push_back('!');
    for (size_t i = 0; i < arity; ++i) { live[i] = out_credits[i] != 0; ck.push_back(live[i] ? '1' : '0'); }
    UnitCache& uc = *cache_;
    int id = -1;
    {
      std::unique_lock<std::mutex> lk(uc.mu);
      for (;;) {
        auto it = uc.index.find(ck);
        if (it == uc.index.end()) { uc.index.emplace(ck, -1); break; }  // ours to record
        if (it->second >= 0) { id = it->second; break; }
        uc.cv.wait(lk);
      }
    }
    if (id < 0) try {
      // the component ON ITS OWN: the body as the root of a fresh two-pass run (run_streaming, circuit/mod.rs:253-301) whose output i has a
      // credit only if the parent reads it — the stand-alone recording makes the parent's dead-gate decisions
      AbiRecordMode rec;
      StreamingRunner run(rec, inputs.size(), body);
      run.set_output_liveness(live);
      const Wires& in = run.prepare();
      for (WireId w : in) rec.declare_input(w);
      const Wires& o = run.execute();
      if (o.size() != arity) gsv_panic("unit returned wrong arity");
      auto u = std::make_unique<Unit>();
      Wires produced;
      for (size_t i = 0; i < arity; ++i) {
        const WireId w = o[i];
        if (w == UNREACHABLE) { u->out_index.push_back(-1); continue; }
        if (w == FALSE_WIRE) { u->out_index.push_back(-2); continue; }
        if (w == TRUE_WIRE) { u->out_index.push_back(-3); continue; }
        bool passed = false;
        for (size_t k = 0; k < in.size(); ++k) if (in[k] == w) { u->out_index.push_back(-int32_t(4 + k)); passed = true; break; }
        if (passed) continue;
        u->out_index.push_back(int32_t(produced.size()));
        produced.push_back(w);
      }
      u->n_program_outputs = produced.size();
      gsv_compile_opts co{};
      co.struct_size = sizeof co;
      co.keep_trace = 0; co.background = 1; co.consume_recorder = 1;
      co.for_plan = pr_;  // window_div and the plan file are the recorder's
      u->program = rec.compile(produced, co);
      {
        std::lock_guard<std::mutex> lk(uc.mu);
        id = int(uc.units.size());
        uc.units.push_back(std::move(u));
        uc.index[ck] = id;
      }
      uc.cv.notify_all();
    } catch (...) {
      { std::lock_guard<std::mutex> lk(uc.mu); auto it = uc.index.find(ck); if (it != uc.index.end() && it->second < 0) uc.index.erase(it); }
      uc.cv.notify_all();
      throw;
    }
    const Unit* u;
    { std::lock_guard<std::mutex> lk(uc.mu); u = uc.units[size_t(id)].get(); }
    std::vector<uint64_t> produced(u->n_program_outputs);
    if (collect_) {
      io_.flush();  // the glue gates in front of the call
      chk(gsv_plan_recorder_call(pr_, u->program, inputs.data(), produced.data()), "gsv_plan_recorder_call");
      for (uint64_t w : produced) { if (io_.written.size() <= w) io_.written.resize(size_t(w) + 1, 0); io_.written[size_t(w)] = 1; }
      ++n_calls;
    } else {
      for (uint64_t& w : produced) w = local_next_++;
    }
    out.assign(arity, UNREACHABLE);
    for (size_t i = 0; i < arity; ++i) {
      const int32_t oi = u->out_index[i];
      if (oi == -1) continue;
      if (oi == -2) { out[i] = FALSE_WIRE; continue; }
      if (oi == -3) { out[i] = TRUE_WIRE; continue; }
      if (oi <= -4) { out[i] = inputs[size_t(-oi - 4)]; continue; }
      out[i] = produced[size_t(oi)];
    }
    return true;
  }
  uint64_t n_calls = 0;

 private:
  bool is_unit(const ComponentKey& key) const {  // a ComponentKey starts with the component's name (component_key.rs:16-39)
    for (const std::string& n : unit_names_)
      if (key.size() > n.size() && key.compare(0, n.size(), n) == 0 && (key[n.size()] == '#' || key[n.size()] == '|')) return true;
    return false;
  }
  gsv_plan_recorder* pr_;
  bool collect_;
  std::vector<std::string> unit_names_;
  std::shared_ptr<UnitCache> cache_;
  RecorderIo<gsv_plan_recorder> io_;
  WireId local_next_ = WIRE_MIN;
};

struct MacSink {  // CiphertextHandler::handle for ONE instance: AESAccumulatingHash (h <- AES_K(h ^ ct), ciphertext_hasher.rs:23-29)
  uint8_t state[16] = {0};
  uint64_t next_record = 0;
  bool in_order = true;
  // --destroy-in-sink: what a host's `Drop` (or a garbage collector) does on the handler's thread in the middle of a pass — a second
  // session and its plan are destroyed from INSIDE the callback (round 6: the engine defers the release to the end of the pass; before,
  // hipFree waited for the ring pass, which waited for this callback: a 60 s stall ended by the device's watchdog)
  gsv_session* victim_session = nullptr;
  gsv_plan* victim_plan = nullptr;
  uint64_t destroy_after = 0, deferred_before = 0, deferred_after = 0;
  double destroy_s = 0;
};
static int mac_sink(void* user, size_t instance, uint64_t first, const uint8_t* records, uint64_t n) {
  MacSink& m = *static_cast<MacSink*>(user);
  if (instance != 0 || first != m.next_record) m.in_order = false;
  m.next_record = first + n;
  if (m.victim_session && first + n >= m.destroy_after) {
    const double t0 = now_s();
    m.deferred_before = gsv_deferred_release_count();
    gsv_session_destroy(m.victim_session); m.victim_session = nullptr;
    gsv_plan_destroy(m.victim_plan); m.victim_plan = nullptr;
    m.deferred_after = gsv_deferred_release_count();
    m.destroy_s = now_s() - t0;
  }
  return gsv_cbcmac_update(m.state, records, n);
}
static std::string hex(const uint8_t* b, size_t n) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (size_t i = 0; i < n; ++i) { s.push_back(d[b[i] >> 4]); s.push_back(d[b[i] & 15]); }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: ext_host <circuit spec> <units csv> <plan file> [--window-div N] [--warmup-threads N] [--garble SEED] [--ring] [--destroy-in-sink]\n"); return 2; }
  const std::string spec = argv[1], units_csv = argv[2], path = argv[3];
  uint32_t window_div = 4;
  int warm_threads = -1;
  bool garble = false, ring = false, destroy_in_sink = false, engine_mac_too = true;
  uint64_t seed = 0;
  for (int i = 4; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--window-div") && i + 1 < argc) window_div = uint32_t(atoi(argv[++i]));
    else if (!std::strcmp(argv[i], "--warmup-threads") && i + 1 < argc) warm_threads = atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--garble") && i + 1 < argc) { garble = true; seed = std::strtoull(argv[++i], nullptr, 10); }
    else if (!std::strcmp(argv[i], "--no-engine-mac")) engine_mac_too = false;   // only the host's own CBC-MAC (one serial chain on the sink thread instead of two)
    else if (!std::strcmp(argv[i], "--ring")) ring = true;                        // the whole pass as one launch over a ciphertext ring (GSV_STREAM_RING)
    else if (!std::strcmp(argv[i], "--destroy-in-sink")) destroy_in_sink = true;  // destroy a second session + plan from the sink callback, mid-pass
    else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    std::vector<std::string> units;
    { std::string cur; for (char c : units_csv + ",") { if (c == ',') { if (!cur.empty()) units.push_back(cur); cur.clear(); } else cur.push_back(c); } }
    const double t0 = now_s();
    NamedCircuit nc = make_circuit(spec);
    gsv_plan_recorder_opts po{};
    po.struct_size = sizeof po; po.window_div = window_div; po.plan_file = path.c_str();
    gsv_plan_recorder* pr = nullptr;
    chk(gsv_plan_recorder_create_opts(&po, &pr), "gsv_plan_recorder_create_opts");
    auto cache = std::make_shared<UnitCache>();
    // warm-up recorders: the circuit's key-specific units (the verifier: 182 constant line functions of 18 M gates) recorded side by side
    // with the driver, each under a mode of its own that shares the unit cache — host-side parallelism, nothing the engine knows about
    const size_t n_warm = nc.warmups.size();
    size_t n_rec = warm_threads >= 0 ? size_t(warm_threads) : std::max<size_t>(1, std::min<size_t>(16, std::thread::hardware_concurrency()) / 4);
    n_rec = std::min(n_rec, n_warm);
    std::atomic<size_t> next{0};
    std::mutex err_mu;
    std::string err;
    std::vector<std::thread> crew;
    for (size_t t = 0; t < n_rec; ++t)
      crew.emplace_back([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n_warm) return;
          try {
            AbiPlanMode wm(pr, false, units, cache);
            StreamingRunner wrun(wm, nc.warmups[i].n_inputs, nc.warmups[i].fn);
            (void)wrun.prepare();
            (void)wrun.execute();
          } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (err.empty()) err = e.what();
            next.store(n_warm);
            return;
          }
        }
      });
    struct Joiner { std::vector<std::thread>& th; std::atomic<size_t>& next; size_t n; ~Joiner() { next.store(n); for (auto& t : th) if (t.joinable()) t.join(); } } joiner{crew, next, n_warm};
    AbiPlanMode mode(pr, true, units, cache);
    StreamingRunner run(mode, nc.n_inputs, nc.fn);
    for (WireId w : run.prepare()) mode.declare_input(w);
    const Wires outs = run.execute();
    mode.flush();
    for (auto& t : crew) t.join();
    if (!err.empty()) throw std::runtime_error("warm-up recorder: " + err);
    const double t_rec = now_s();
    gsv_plan* meta = nullptr;
    chk(gsv_plan_recorder_finish(pr, outs.data(), outs.size(), &meta), "gsv_plan_recorder_finish");
    const double t_built = now_s();
    uint64_t n_gates = 0, n_ct = 0, n_calls = 0, n_in = 0, n_out = 0;
    chk(gsv_plan_counts(meta, &n_gates, &n_ct, &n_calls), "gsv_plan_counts");
    chk(gsv_plan_io(meta, &n_in, &n_out), "gsv_plan_io");
    gsv_plan_destroy(meta);
    gsv_plan_recorder_destroy(pr);
    std::fprintf(stderr, "PLAN_FILE_READY %s\n", path.c_str());  // (tests: the file is complete — digest it while this process garbles)
    std::fflush(stderr);
    const size_t n_units = cache->units.size();
    cache.reset();  // the unit programs (their records are in the file)
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    const double build_rss_gb = double(ru.ru_maxrss) / 1e6;
    std::string extra;
    if (garble) {
      gsv_engine* e = nullptr;
      chk(gsv_engine_create(0, &e), "gsv_engine_create");
      gsv_plan* plan = nullptr;
      const double t1 = now_s();
      chk(gsv_plan_load(path.c_str(), e, &plan), "gsv_plan_load");
      const double t_load = now_s() - t1;
      gsv_plan_session_opts so{};
      so.retain_stream = ring ? GSV_STREAM_RING : 0;  // nothing retained: the stream leaves the device segment by segment of the running window
      if (ring) { so.max_concurrent_calls = 16; so.drain_segment_records = 300000; }
      gsv_session* s = nullptr;
      chk(gsv_session_create_plan_opts(e, plan, 1, &so, &s), "gsv_session_create_plan_opts");
      MacSink sink;
      if (destroy_in_sink) {
        chk(gsv_plan_load(path.c_str(), e, &sink.victim_plan), "gsv_plan_load (second plan)");
        chk(gsv_session_create_plan_opts(e, sink.victim_plan, 1, &so, &sink.victim_session), "gsv_session_create_plan_opts (second session)");
        sink.destroy_after = n_ct / 3;
      }
      std::vector<uint8_t> delta(16), consts(32), inputs(size_t(n_in) * 16);
      chk(gsv_labels_from_seed(seed, size_t(n_in), delta.data(), consts.data(), consts.data() + 16, inputs.data()), "gsv_labels_from_seed");
      chk(gsv_session_set_garble_inputs(s, delta.data(), consts.data(), inputs.data()), "gsv_session_set_garble_inputs");
      uint8_t engine_mac[16];
      const double t2 = now_s();
      std::memset(engine_mac, 0, sizeof engine_mac);
      chk(gsv_session_garble_streaming_sink(s, 0, 0, 0, mac_sink, &sink, 1, engine_mac_too ? engine_mac : nullptr), "gsv_session_garble_streaming_sink");
      if (!engine_mac_too) std::memcpy(engine_mac, sink.state, 16);
      const double t_garble = now_s() - t2;
      std::vector<uint8_t> out(size_t(n_out) * 16);
      chk(gsv_session_read_outputs(s, out.data(), nullptr), "gsv_session_read_outputs");
      gsv_session_destroy(s);
      gsv_plan_destroy(plan);
      gsv_engine_destroy(e);
      char buf[256];
      std::snprintf(buf, sizeof buf, ", \"seed\": %llu, \"load_s\": %.2f, \"garble_s\": %.2f, \"sink_records\": %llu, \"sink_in_order\": %s", (unsigned long long)seed, t_load, t_garble,
                    (unsigned long long)sink.next_record, sink.in_order ? "true" : "false");
      extra = buf;
      if (destroy_in_sink) {
        std::snprintf(buf, sizeof buf, ", \"destroyed_in_sink\": %s, \"deferred_releases\": %llu, \"destroy_call_s\": %.4f, \"deferred_total_after_pass\": %llu",
                      sink.victim_session ? "false" : "true", (unsigned long long)(sink.deferred_after - sink.deferred_before), sink.destroy_s, (unsigned long long)gsv_deferred_release_count());
        extra += buf;
      }
      extra += ", \"ct_hash\": \"" + hex(sink.state, 16) + "\", \"engine_ct_hash\": \"" + hex(engine_mac, 16) + "\", \"output_label0\": \"" + hex(out.data(), out.size()) + "\"";
    }
    getrusage(RUSAGE_SELF, &ru);
    std::printf("{\"plan_file\": \"%s\", \"window_div\": %u, \"n_gates\": %llu, \"n_ciphertexts\": %llu, \"n_calls\": %llu, \"n_inputs\": %llu, \"n_outputs\": %llu, \"unit_programs\": %zu, "
                "\"unit_calls\": %llu, \"warmup_recorders\": %zu, \"record_s\": %.2f, \"build_s\": %.2f, \"build_peak_rss_gb\": %.2f, \"peak_rss_gb\": %.2f%s}\n",
                path.c_str(), window_div, (unsigned long long)n_gates, (unsigned long long)n_ct, (unsigned long long)n_calls, (unsigned long long)n_in, (unsigned long long)n_out, n_units,
                (unsigned long long)mode.n_calls, n_rec, t_rec - t0, t_built - t0, build_rss_gb, double(ru.ru_maxrss) / 1e6, extra.c_str());
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ext_host: %s\n", e.what());
    return 1;
  }
}
