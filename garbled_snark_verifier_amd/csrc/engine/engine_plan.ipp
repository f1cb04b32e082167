// Part of engine.cpp: plans — the built-in builder (single and dual builds), plan files, background compilation, the plan recorder.
// ---------------------------------------------------------------- plans
int gsv_plan_create(gsv_plan** out) {
  if (!out) return fail(GSV_ERR_INVALID, "null out");
  *out = new gsv_plan();
  return GSV_OK;
}
void gsv_plan_destroy(gsv_plan* p) {
  if (!p) return;
  release_or_defer([p] {
    for (gsv_program* q : p->owned) program_destroy_now(q);
    delete p;
  });
}
int gsv_plan_add_call(gsv_plan* p, const gsv_program* prog, const uint32_t* in_globals, const uint32_t* out_globals) {
  if (!p || !prog || p->finished) return fail(GSV_ERR_INVALID, "bad argument / plan already finished");
  { int rc = program_ready(prog); if (rc) return rc; }
  const Program& g = prog->prog;
  if ((!in_globals && !g.input_slots.empty()) || (!out_globals && !g.output_slots.empty())) return fail(GSV_ERR_INVALID, "null wire list");
  if (!g.fb_src_slot.empty()) return fail(GSV_ERR_INVALID, "a program compiled with feedback cannot be a plan call");
  PlanCall c;
  c.prog = const_cast<gsv_program*>(prog);
  c.in_globals.assign(in_globals, in_globals + g.input_slots.size());
  c.out_globals.assign(out_globals, out_globals + g.output_slots.size());
  c.gid_off = p->n_gates; c.ct_off = p->n_ct;
  p->n_gates += g.n_gates; p->n_ct += g.n_ct;
  for (uint32_t w : c.in_globals) if (w < PLAN_WIRE_FALSE) p->n_globals = std::max(p->n_globals, w + 1);
  for (uint32_t w : c.out_globals) {
    if (w >= PLAN_WIRE_FALSE) return fail(GSV_ERR_INVALID, "a call cannot write a constant");
    p->n_globals = std::max(p->n_globals, w + 1);
  }
  p->calls.push_back(std::move(c));
  return GSV_OK;
}
int gsv_plan_finish(gsv_plan* p, uint32_t n_inputs, const uint32_t* output_globals, size_t n_outputs) {
  if (!p || p->finished || (!output_globals && n_outputs)) return fail(GSV_ERR_INVALID, "bad argument");
  // global wires 0..n_inputs-1 are the plan's inputs; every other global must be written by a call before it is read
  std::vector<uint8_t> defined(std::max<uint32_t>(p->n_globals, n_inputs), 0);
  for (uint32_t i = 0; i < n_inputs; ++i) defined[i] = 1;
  for (const PlanCall& c : p->calls) {
    for (uint32_t w : c.in_globals) if (w < PLAN_WIRE_FALSE && !defined[w]) return fail(GSV_ERR_CIRCUIT, "plan call reads global wire " + std::to_string(w) + " before any call wrote it");
    for (uint32_t w : c.out_globals) defined[w] = 1;
  }
  for (size_t i = 0; i < n_outputs; ++i)
    if (output_globals[i] < PLAN_WIRE_FALSE && (output_globals[i] >= defined.size() || !defined[output_globals[i]])) return fail(GSV_ERR_CIRCUIT, "plan output is never written");
  p->n_globals = uint32_t(defined.size());
  p->n_inputs = n_inputs;
  p->outputs.assign(output_globals, output_globals + n_outputs);
  p->finished = true;
  return GSV_OK;
}
// Record one of the built-in restated circuits under the two-pass driver with the named components (comma separated,
// e.g. "fq12::mul_montgomery,fq12::square_montgomery") turned into calls of separately compiled programs; everything
// between them is compiled as glue programs (plan_builder.hpp).
static int plan_window_div(uint32_t* window_div) {
  *window_div = 1;
  if (getenv("GSV_PLAN_HALF_WINDOW") && atoi(getenv("GSV_PLAN_HALF_WINDOW")) != 0) *window_div = 2;
  if (const char* e = getenv("GSV_PLAN_WINDOW_DIV")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) *window_div = uint32_t(v); else return fail(GSV_ERR_INVALID, "GSV_PLAN_WINDOW_DIV must be 1, 2 or 4"); }
  return GSV_OK;
}
// A dual build (gsv_plan_build_file_pair): the second image of every program — compiled from the same recording for 1 / window_div of the
// LDS window, handed to `sink` — and the second plan.
struct DualBuild {
  const char* units_csv = nullptr;  // the second plan's units; null or equal to the first plan's: one recorder serves both plans
  uint32_t window_div = 1;
  std::function<void(Program&)> sink;
  gsv_plan** out = nullptr;
};
// sink: see PlanUnitCache::sink (gsv_plan_build_file); empty = the programs stay in memory.  window_div_override: 0 = GSV_PLAN_WINDOW_DIV.
static int plan_from_circuit_impl(const char* spec, const char* units_csv, const std::function<void(Program&)>& sink, gsv_plan** out, uint32_t window_div_override = 0,
                                  const DualBuild* dual = nullptr) {
  if (!spec || !units_csv || !out) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  auto split_csv = [](const char* csv) {
    std::vector<std::string> v;
    std::string cur;
    for (const char* q = csv;; ++q) {
      if (*q == ',' || *q == 0) { if (!cur.empty()) v.push_back(cur); cur.clear(); if (!*q) break; }
      else cur.push_back(*q);
    }
    return v;
  };
  const std::vector<std::string> names = split_csv(units_csv);
  const bool two_recorders = dual && dual->units_csv && split_csv(dual->units_csv) != names;
  NamedCircuit nc = make_circuit(spec);
  PlanRecordMode mode(names);
  CompileOptions opt;
  if (const char* e = getenv("GSV_FUSE")) opt.fuse = atoi(e) != 0;
  // GSV_PLAN_WINDOW_DIV=2|4: compile every program once, for half / a quarter of the LDS window; the same image then serves every
  // layout of up to that many instances per workgroup and the recorded traces are not kept (less host memory and no second
  // compilation for plans with hundreds of programs, at a smaller window when sessions have few instances).
  // GSV_PLAN_HALF_WINDOW=1 is the older spelling of GSV_PLAN_WINDOW_DIV=2.
  uint32_t window_div = 1;
  if (window_div_override) window_div = window_div_override;
  else { int rc = plan_window_div(&window_div); if (rc) return rc; }
  if (dual) {
    if (!sink || !dual->sink || !dual->out) return fail(GSV_ERR_INVALID, "internal: a dual build writes both plans to files");
    CompileOptions ob = opt;
    ob.lds_slots = std::min<uint32_t>(ob.lds_slots, LDS_WINDOW_SLOTS / dual->window_div);
    mode.cache()->dual = true; mode.cache()->bg_opt_b = ob; mode.cache()->sink_b = dual->sink;
    if (two_recorders) { mode.cache()->names_a = names; mode.cache()->names_b = split_csv(dual->units_csv); }
  }
  // (a plan built straight into a file keeps ONE image per program and no trace: with GSV_PLAN_WINDOW_DIV=1 that image has the full LDS
  // window and serves one instance per workgroup only — the small-batch plan of bench.py: 3 % faster steps for 1 and 16 instances)
  const bool single_image = window_div > 1 || bool(sink);
  mode.cache()->sink = sink;
  if (single_image) opt.lds_slots = std::min<uint32_t>(opt.lds_slots, LDS_WINDOW_SLOTS / window_div);
  mode.compile_in_background(opt, single_image);  // units are compiled while the driver records the rest of the circuit
  std::vector<uint32_t> in_ssa, out_ssa;
  const bool dbg = getenv("GSV_PLAN_DEBUG") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  size_t n_recorders = 0;
  // Two plans with different units: the second plan's driver walks the circuit on a thread of its own, over the SAME unit cache — the
  // units the plans share (the verifier: its 182 constant line functions, 3.3 B of the 3.5 B gates a build records) are recorded once,
  // by whoever gets there first (the other waits for them), and compiled for both plans.
  std::unique_ptr<PlanRecordMode> mode_b;
  std::vector<uint32_t> in_ssa_b, out_ssa_b;
  std::thread walk_b;
  std::exception_ptr walk_b_err;
  if (two_recorders) {
    mode_b.reset(new PlanRecordMode(split_csv(dual->units_csv), mode.cache()));
    walk_b = std::thread([&] {
      try {
        size_t nr = 0;
        record_plan(*mode_b, nc.n_inputs, nc.fn, std::vector<NamedCircuit::Warmup>(), in_ssa_b, out_ssa_b, &nr);
      } catch (...) { walk_b_err = std::current_exception(); }
    });
  }
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } walk_b_joiner{walk_b};
  record_plan(mode, nc.n_inputs, nc.fn, nc.warmups, in_ssa, out_ssa, &n_recorders);
  if (walk_b.joinable()) walk_b.join();
  if (walk_b_err) std::rethrow_exception(walk_b_err);
  if (dbg) std::fprintf(stderr, "plan: recorded at %.1f s (%zu units, %zu glue classes, %zu warm-ups on %zu threads)\n", since(), mode.units.size(), mode.glue_classes.size(), nc.warmups.size(), n_recorders);
  mode.wait_for_compilations();
  if (dbg) std::fprintf(stderr, "plan: background compilations finished at %.1f s\n", since());
  BuiltPlan bp = finish_plan(mode, in_ssa, out_ssa, opt, dual && !two_recorders ? 2 : 0);
  BuiltPlan bp_second;
  if (two_recorders) bp_second = finish_plan(*mode_b, in_ssa_b, out_ssa_b, opt, 1);
  if (dbg) std::fprintf(stderr, "plan: all programs compiled at %.1f s\n", since());
  if (dbg) {  // per program: how often it is called, its size and shape (latency-bound programs carry four-wire records)
    std::vector<size_t> n_calls(bp.programs.size(), 0);
    for (const BuiltPlan::Call& c : bp.calls) if (c.program >= 0) n_calls[size_t(c.program)]++;
    for (size_t k = 0; k < bp.programs.size(); ++k) {
      const Program& g = bp.programs[k];
      std::fprintf(stderr, "plan: program %3zu: %5zu calls, %9llu gates, %8u steps (%.0f records per step), and_terms %u, lds slots %u of %u, label reads from hbm %.0f %%\n", k, n_calls[k],
                   (unsigned long long)g.n_gates, g.n_steps, g.n_steps ? double(g.n_ct + g.n_fused_free) / g.n_steps : 0.0, g.and_terms, g.n_lds_slots, g.lds_slots_limit,
                   100.0 * double(g.reads_hbm) / std::max<double>(1.0, double(g.reads_hbm + g.reads_lds)));
    }
  }
  // the plan object over one set of images (a dual build makes two: same calls, same globals)
  auto make_plan = [&](BuiltPlan& bp, const PlanRecordMode& mode, std::vector<Program>& programs, uint32_t wdiv, bool keep_traces, gsv_plan** dst) -> int {
    std::unique_ptr<gsv_plan> plan(new gsv_plan());
    for (size_t k = 0; k < programs.size(); ++k) {
      gsv_program* q = new gsv_program();
      plan->owned.push_back(q);
      q->prog = std::move(programs[k]);
      q->window_div = wdiv;
      if (keep_traces) q->src.reset(new ProgramSource{std::move(bp.traces[k]), bp.prog_inputs[k], bp.prog_outputs[k], {}, opt});
      for (size_t i = 0; i < q->prog.input_slots.size(); ++i)
        if (q->prog.input_slots[i] != SLOT_FIRST_INPUT + i) { gsv_plan_destroy(plan.release()); return fail(GSV_ERR_CIRCUIT, "internal: inputs are not slot-contiguous"); }
    }
    for (const BuiltPlan::Call& c : bp.calls) {
      int rc = gsv_plan_add_call(plan.get(), plan->owned[size_t(c.program)], c.in_globals.data(), c.out_globals.data());
      if (rc) { gsv_plan_destroy(plan.release()); return rc; }
    }
    int rc = gsv_plan_finish(plan.get(), bp.n_inputs, bp.outputs.data(), bp.outputs.size());
    if (rc) { gsv_plan_destroy(plan.release()); return rc; }
    if (plan->n_gates != mode.n_gates()) { gsv_plan_destroy(plan.release()); return fail(GSV_ERR_CIRCUIT, "internal: plan gate count differs from the recorded stream"); }
    *dst = plan.release();
    return GSV_OK;
  };
  gsv_plan* second = nullptr;
  if (dual) { int rc = two_recorders ? make_plan(bp_second, *mode_b, bp_second.programs, dual->window_div, false, &second) : make_plan(bp, mode, bp.programs_b, dual->window_div, false, &second); if (rc) return rc; }
  if (single_image) for (Trace& t : bp.traces) t = Trace();
  int rc = make_plan(bp, mode, bp.programs, window_div, !single_image, out);
  if (rc) { gsv_plan_destroy(second); return rc; }
  if (dual) *dual->out = second;
  return GSV_OK;
  GSV_CATCH
}
int gsv_plan_from_circuit(const char* spec, const char* units_csv, gsv_plan** out) { return plan_from_circuit_impl(spec, units_csv, nullptr, out); }
int gsv_plan_io(const gsv_plan* p, uint64_t* n_inputs, uint64_t* n_outputs) {
  if (!p || !p->finished) return fail(GSV_ERR_INVALID, "plan not finished");
  if (n_inputs) *n_inputs = p->n_inputs;
  if (n_outputs) *n_outputs = p->outputs.size();
  return GSV_OK;
}
int gsv_plan_counts(const gsv_plan* p, uint64_t* n_gates, uint64_t* n_ciphertexts, uint64_t* n_calls) {
  if (!p) return fail(GSV_ERR_INVALID, "null plan");
  if (n_gates) *n_gates = p->n_gates;
  if (n_ciphertexts) *n_ciphertexts = p->n_ct;
  if (n_calls) *n_calls = p->calls.size();
  return GSV_OK;
}
// ---- plan files ---------------------------------------------------------------------------------------------------------
// A built plan (compiled programs + calls) as one file, so that the ~100 s / ~50 GB build of the verifier plan is paid once per
// machine: rank 0 of a node builds and saves, every other rank (and every later process) loads.  gsv_plan_load with an engine
// streams each program's records from the (memory-mapped, page-cache shared) file straight into that GPU's memory; the host
// keeps only the metadata a session needs, so a loading rank's private memory stays small.  Layout (little endian, every array
// padded to 16 bytes):  PlanFileHeader | program blocks in any order, each: PlanFileProgram, steps, ands, xors, ct_pos, input_slots,
// output_slots | at calls_off, per call: {program, n_in, n_out}, in_globals, out_globals | outputs | at table_off: one uint64 file
// offset per program.  The table is what lets gsv_plan_build_file append a program the moment a worker has compiled it.
namespace gsv_plan_file {
constexpr char PLAN_MAGIC[8] = {'G', 'S', 'V', 'P', 'L', 'A', 'N', '4'};
struct PlanFileHeader {
  char magic[8];
  uint32_t n_programs, n_calls, n_globals, n_inputs, n_outputs, lds_window_slots;
  uint64_t n_gates, n_ct, rec_sizes;  // rec_sizes: sizeof(StepDesc) | sizeof(AndRec) << 16 | sizeof(XorRec) << 32 (format guard)
  uint64_t calls_off, table_off;
};
struct PlanFileProgram {
  uint64_t n_steps, n_ands, n_xors, n_ct_pos, n_inputs, n_outputs;
  uint64_t n_gates, n_ct, n_dead, n_fused_free, reads_lds, reads_hbm, writes_lds, writes_hbm;
  uint64_t gate_count[GATE_TYPE_COUNT];
  uint32_t n_slots, n_lds_slots, lds_slots_limit, fb_stage_base, and_depth, n_and_steps, max_step_width, peak_live, window_div, and_terms;
};
constexpr uint64_t plan_rec_sizes() { return uint64_t(sizeof(StepDesc)) | (uint64_t(sizeof(AndRec)) << 16) | (uint64_t(sizeof(XorRec)) << 32); }
inline size_t pad16(size_t n) { return (n + 15) & ~size_t(15); }
struct FileCloser { FILE* f; ~FileCloser() { if (f) std::fclose(f); } };
struct Mapping {
  const uint8_t* base = nullptr; size_t size = 0; int fd = -1;
  ~Mapping() { if (base) munmap(const_cast<uint8_t*>(base), size); if (fd >= 0) close(fd); }
};
// Writes a plan file: program blocks may be appended from several threads (each reserves its range, then pwrite()s it), the calls,
// the offset table and the header follow when the plan is complete; the file appears under its name only then (temp file + rename).
class PlanFileWriter {
 public:
  ~PlanFileWriter() { if (fd_ >= 0) { close(fd_); std::remove(tmp_.c_str()); } }
  int open_file(const std::string& path) {
    path_ = path;
    tmp_ = path + ".tmp." + std::to_string(long(getpid()));
    fd_ = ::open(tmp_.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd_ < 0) return fail(GSV_ERR_INVALID, "cannot create " + tmp_);
    next_.store(pad16(sizeof(PlanFileHeader)));
    return GSV_OK;
  }
  bool ok() const { return !bad_.load(); }
  // -> file offset of the block
  uint64_t append_program(const Program& g, uint32_t window_div) {
    PlanFileProgram m{};
    m.n_steps = g.steps.size(); m.n_ands = g.ands.size(); m.n_xors = g.xors.size(); m.n_ct_pos = g.ct_pos.size(); m.n_inputs = g.input_slots.size(); m.n_outputs = g.output_slots.size();
    m.n_gates = g.n_gates; m.n_ct = g.n_ct; m.n_dead = g.n_dead; m.n_fused_free = g.n_fused_free;
    m.reads_lds = g.reads_lds; m.reads_hbm = g.reads_hbm; m.writes_lds = g.writes_lds; m.writes_hbm = g.writes_hbm;
    for (int i = 0; i < GATE_TYPE_COUNT; ++i) m.gate_count[i] = g.gate_count[i];
    m.n_slots = g.n_slots; m.n_lds_slots = g.n_lds_slots; m.lds_slots_limit = g.lds_slots_limit; m.fb_stage_base = g.fb_stage_base; m.and_depth = g.and_depth;
    m.n_and_steps = g.n_and_steps; m.max_step_width = g.max_step_width; m.peak_live = g.peak_live; m.window_div = window_div; m.and_terms = g.and_terms;
    const void* parts[7] = {&m, g.steps.data(), g.ands.data(), g.xors.data(), g.ct_pos.data(), g.input_slots.data(), g.output_slots.data()};
    const size_t lens[7] = {sizeof m, g.steps.size() * sizeof(StepDesc), g.ands.size() * sizeof(AndRec), g.xors.size() * sizeof(XorRec), g.ct_pos.size() * 4, g.input_slots.size() * 4, g.output_slots.size() * 4};
    size_t total = 0;
    for (size_t l : lens) total += pad16(l);
    const uint64_t off = next_.fetch_add(total);
    uint64_t pos = off;
    for (int i = 0; i < 7; ++i) { put_at(pos, parts[i], lens[i]); pos += pad16(lens[i]); }
    return off;
  }
  // single-threaded tail: calls, outputs, table, header; then the rename
  int finish(const gsv_plan* p, const std::vector<uint64_t>& program_off, const std::map<const gsv_program*, uint32_t>& index) {
    PlanFileHeader h{};
    std::memcpy(h.magic, PLAN_MAGIC, 8);
    h.n_programs = uint32_t(program_off.size()); h.n_calls = uint32_t(p->calls.size()); h.n_globals = p->n_globals; h.n_inputs = p->n_inputs; h.n_outputs = uint32_t(p->outputs.size());
    h.lds_window_slots = LDS_WINDOW_SLOTS; h.n_gates = p->n_gates; h.n_ct = p->n_ct; h.rec_sizes = plan_rec_sizes();
    std::vector<uint8_t> tail;
    auto put = [&](const void* d, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(d); tail.insert(tail.end(), b, b + n); tail.resize(pad16(tail.size()), 0); };
    for (const PlanCall& c : p->calls) {
      const uint32_t hdr[4] = {index.at(c.prog), uint32_t(c.in_globals.size()), uint32_t(c.out_globals.size()), 0};
      put(hdr, sizeof hdr);
      put(c.in_globals.data(), c.in_globals.size() * 4);
      put(c.out_globals.data(), c.out_globals.size() * 4);
    }
    put(p->outputs.data(), p->outputs.size() * 4);
    h.calls_off = next_.load();
    h.table_off = h.calls_off + tail.size();
    put(program_off.data(), program_off.size() * 8);
    put_at(h.calls_off, tail.data(), tail.size());
    put_at(0, &h, sizeof h);
    const bool closed = close(fd_) == 0;
    fd_ = -1;
    if (!ok() || !closed || std::rename(tmp_.c_str(), path_.c_str()) != 0) { std::remove(tmp_.c_str()); return fail(GSV_ERR_INVALID, "cannot write " + path_); }
    return GSV_OK;
  }

 private:
  void put_at(uint64_t off, const void* d, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(d);
    while (n) {
      const ssize_t w = pwrite(fd_, b, n, off_t(off));
      if (w <= 0) { bad_.store(true); return; }
      b += w; off += uint64_t(w); n -= size_t(w);
    }
  }
  int fd_ = -1;
  std::string path_, tmp_;
  std::atomic<uint64_t> next_{0};
  std::atomic<bool> bad_{false};
};
}  // namespace gsv_plan_file
using namespace gsv_plan_file;

// ---- background compilation (gsv_program_compile_opts) -------------------------------------------------------------------------------
// One pool for the process, created on first use: GSV_COMPILE_THREADS workers (default: the hardware's, at most 16).  submit() blocks while
// as many jobs as workers are queued, which bounds the traces and compiler temporaries in flight.
static CompilePool& abi_compile_pool() {
  static CompilePool pool(plan_compile_threads());
  return pool;
}
// Waits for a program's background compilation (no-op otherwise) and returns its status.
static int program_ready(const gsv_program* cp) {
  gsv_program* p = const_cast<gsv_program*>(cp);
  std::unique_lock<std::mutex> lk(p->cmu);
  p->ccv.wait(lk, [p] { return !p->compiling; });
  if (p->compile_rc) return fail(p->compile_rc, p->compile_err);
  return GSV_OK;
}

// ---- plan recorder: the plan builder behind the C ABI, for a host that runs its own two-pass driver
struct gsv_plan_recorder {
  PlanRecordMode mode{std::vector<std::string>()};
  std::vector<uint32_t> inputs;
  std::vector<const gsv_program*> externals;
  std::map<const gsv_program*, int> unit_of;
  uint32_t window_div = 1;
  bool single_image = false;             // every program of the plan exists as ONE image (window_div > 1 or a plan file): no trace is kept
  std::unique_ptr<PlanFileWriter> file;  // set: programs are appended to the plan file as soon as they are compiled, their records dropped
  std::mutex mu;
  std::vector<gsv_program*> compiled_for;  // programs compiled with gsv_compile_opts.for_plan = this recorder: their jobs write to `file`
  bool finished = false;
  void spill(Program& g) const {
    g.file_off = file->append_program(g, window_div);
    g.spilled = true;
    std::vector<StepDesc>().swap(g.steps); std::vector<AndRec>().swap(g.ands); std::vector<XorRec>().swap(g.xors); std::vector<uint32_t>().swap(g.ct_pos);
  }
  void wait_for_compilations() {
    std::vector<gsv_program*> v;
    { std::lock_guard<std::mutex> lk(mu); v = compiled_for; }
    for (gsv_program* q : v) (void)program_ready(q);
  }
};
// program <-> plan recorder registration (gsv_compile_opts.for_plan): whichever side is destroyed first takes itself out of the other
static std::mutex g_recorder_link_mu;
static void unlink_from_recorder(gsv_program* p) {
  std::lock_guard<std::mutex> lk(g_recorder_link_mu);
  if (gsv_plan_recorder* r = p->for_recorder) {
    std::lock_guard<std::mutex> lk2(r->mu);
    r->compiled_for.erase(std::remove(r->compiled_for.begin(), r->compiled_for.end(), p), r->compiled_for.end());
    p->for_recorder = nullptr;
  }
}
int gsv_plan_recorder_create_opts(const gsv_plan_recorder_opts* o, gsv_plan_recorder** out) {
  if (!out) return fail(GSV_ERR_INVALID, "null out");
  if (o && o->struct_size != sizeof(gsv_plan_recorder_opts)) return fail(GSV_ERR_INVALID, "gsv_plan_recorder_opts.struct_size does not match this library");
  std::unique_ptr<gsv_plan_recorder> r(new gsv_plan_recorder());
  if (o) {
    if (o->window_div != 0 && o->window_div != 1 && o->window_div != 2 && o->window_div != 4) return fail(GSV_ERR_INVALID, "window_div must be 0, 1, 2 or 4");
    r->window_div = std::max<uint32_t>(1, o->window_div);
    if (o->plan_file) {
      r->file.reset(new PlanFileWriter());
      int rc = r->file->open_file(o->plan_file);
      if (rc) return rc;
      gsv_plan_recorder* rp = r.get();
      r->mode.cache()->sink = [rp](Program& g) { rp->spill(g); };  // the glue programs finish_plan compiles
    }
  }
  r->single_image = r->window_div > 1 || bool(r->file);
  *out = r.release();
  return GSV_OK;
}
int gsv_plan_recorder_create(gsv_plan_recorder** out) { return gsv_plan_recorder_create_opts(nullptr, out); }
void gsv_plan_recorder_destroy(gsv_plan_recorder* r) {
  if (!r) return;
  r->wait_for_compilations();  // their jobs hold a pointer to this recorder's plan file
  {
    std::lock_guard<std::mutex> lk(g_recorder_link_mu);
    std::lock_guard<std::mutex> lk2(r->mu);
    for (gsv_program* q : r->compiled_for) q->for_recorder = nullptr;
    r->compiled_for.clear();
  }
  delete r;
}
int gsv_plan_recorder_allocate_wire(gsv_plan_recorder* r, uint16_t credits, uint64_t* wire_out) {
  if (!r || !wire_out) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  *wire_out = r->mode.allocate_wire(credits);
  return GSV_OK;
  GSV_CATCH
}
int gsv_plan_recorder_allocate_wires(gsv_plan_recorder* r, size_t n, uint64_t* first_wire_out) {
  if (!r || !first_wire_out || n == 0) return fail(GSV_ERR_INVALID, "null argument / n == 0");
  GSV_TRY
  *first_wire_out = r->mode.allocate_wire(1);
  for (size_t i = 1; i < n; ++i) (void)r->mode.allocate_wire(1);
  return GSV_OK;
  GSV_CATCH
}
int gsv_plan_recorder_declare_input(gsv_plan_recorder* r, uint64_t wire) {
  if (!r) return fail(GSV_ERR_INVALID, "null recorder");
  GSV_TRY
  r->inputs.push_back(r->mode.define_input(wire));
  return GSV_OK;
  GSV_CATCH
}
int gsv_plan_recorder_push_gates(gsv_plan_recorder* r, const gsv_gate* gates, size_t n) {
  if (!r || (!gates && n)) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  for (size_t i = 0; i < n; ++i) {
    if (gates[i].gate_type >= GATE_TYPE_COUNT) return fail(GSV_ERR_INVALID, "unknown gate type");
    r->mode.evaluate_gate(Gate{gates[i].wire_a, gates[i].wire_b, gates[i].wire_c, static_cast<GateType>(gates[i].gate_type)});
  }
  return GSV_OK;
  GSV_CATCH
}
int gsv_plan_recorder_call(gsv_plan_recorder* r, const gsv_program* program, const uint64_t* in_wires, uint64_t* out_wires) {
  if (!r || !program) return fail(GSV_ERR_INVALID, "null argument");
  // arity and gate count come from the recording (a program whose compilation still runs in the background is accepted as it is)
  uint64_t n_in = program->decl_inputs, n_out = program->decl_outputs, n_gates = program->decl_gates;
  bool fb = program->has_feedback;
  if (!program->has_decl) {  // loaded / built elsewhere: read the image's own tables
    const Program& g = program->prog;
    n_in = g.input_slots.size(); n_out = g.output_slots.size(); n_gates = g.n_gates; fb = !g.fb_src_slot.empty();
  }
  if ((!in_wires && n_in) || (!out_wires && n_out)) return fail(GSV_ERR_INVALID, "null wire list");
  if (fb) return fail(GSV_ERR_INVALID, "a program compiled with feedback cannot be a plan call");
  GSV_TRY
  auto it = r->unit_of.find(program);
  if (it == r->unit_of.end()) {
    r->externals.push_back(program);
    it = r->unit_of.emplace(program, r->mode.add_external_unit(int(r->externals.size()) - 1, n_gates, size_t(n_out))).first;
  }
  Wires in(in_wires, in_wires + n_in), out;
  r->mode.call_external(it->second, in, out);
  for (size_t i = 0; i < out.size(); ++i) out_wires[i] = out[i];
  return GSV_OK;
  GSV_CATCH
}
int gsv_plan_recorder_finish(gsv_plan_recorder* r, const uint64_t* output_wires, size_t n_outputs, gsv_plan** out) {
  if (!r || !out || (!output_wires && n_outputs)) return fail(GSV_ERR_INVALID, "null argument");
  if (r->finished) return fail(GSV_ERR_INVALID, "plan recorder already finished");
  GSV_TRY
  std::vector<uint32_t> out_ssa;
  for (size_t i = 0; i < n_outputs; ++i) out_ssa.push_back(r->mode.current(output_wires[i]));
  CompileOptions opt;
  if (const char* e = getenv("GSV_FUSE")) opt.fuse = atoi(e) != 0;
  if (r->single_image) opt.lds_slots = std::min<uint32_t>(opt.lds_slots, LDS_WINDOW_SLOTS / r->window_div);
  r->wait_for_compilations();
  for (const gsv_program* q : r->externals) { int rc = program_ready(q); if (rc) return rc; }
  BuiltPlan bp = finish_plan(r->mode, r->inputs, out_ssa, opt);
  r->finished = true;
  std::unique_ptr<gsv_plan> plan(new gsv_plan());
  for (size_t k = 0; k < bp.programs.size(); ++k) {
    gsv_program* q = new gsv_program();
    plan->owned.push_back(q);
    q->prog = std::move(bp.programs[k]);
    q->window_div = r->window_div;
    if (!r->single_image) q->src.reset(new ProgramSource{std::move(bp.traces[k]), bp.prog_inputs[k], bp.prog_outputs[k], {}, opt});
  }
  for (const BuiltPlan::Call& c : bp.calls) {
    const gsv_program* q = c.program >= 0 ? plan->owned[size_t(c.program)] : r->externals[size_t(-1 - c.program)];
    int rc = gsv_plan_add_call(plan.get(), q, c.in_globals.data(), c.out_globals.data());
    if (rc) { gsv_plan_destroy(plan.release()); return rc; }
  }
  int rc = gsv_plan_finish(plan.get(), bp.n_inputs, bp.outputs.data(), bp.outputs.size());
  if (rc) { gsv_plan_destroy(plan.release()); return rc; }
  if (r->file) {
    // the offset table in the order of the programs' first calls — the order gsv_plan_build_file and gsv_plan_save use
    std::vector<uint64_t> off;
    std::map<const gsv_program*, uint32_t> index;
    for (const PlanCall& c : plan->calls)
      if (index.emplace(c.prog, uint32_t(off.size())).second) {
        if (!c.prog->prog.spilled) { gsv_plan_destroy(plan.release()); return fail(GSV_ERR_INVALID, "a unit program of this plan was not compiled for its recorder (gsv_compile_opts.for_plan): its records are not in the plan file"); }
        off.push_back(c.prog->prog.file_off);
      }
    rc = r->file->finish(plan.get(), off, index);
    r->file.reset();
    if (rc) { gsv_plan_destroy(plan.release()); return rc; }
  }
  *out = plan.release();
  return GSV_OK;
  GSV_CATCH
}

// ---- compile with options: one image for a share of the LDS window, background compilation, records straight into a plan file
static int compile_impl(gsv_recorder* r, const uint32_t* fb_out_idx, const uint32_t* fb_in_idx, size_t n_feedback, const gsv_compile_opts* o, gsv_program** out) {
  if (!r || !out) return fail(GSV_ERR_INVALID, "null argument");
  if (!r->outputs_declared) return fail(GSV_ERR_INVALID, "outputs not declared");
  if (o && o->struct_size != sizeof(gsv_compile_opts)) return fail(GSV_ERR_INVALID, "gsv_compile_opts.struct_size does not match this library");
  GSV_TRY
  gsv_plan_recorder* const pr = o ? o->for_plan : nullptr;
  uint32_t window_div = o ? o->window_div : 0;
  if (window_div != 0 && window_div != 1 && window_div != 2 && window_div != 4) return fail(GSV_ERR_INVALID, "window_div must be 0, 1, 2 or 4");
  if (pr) {
    if (window_div != 0 && std::max<uint32_t>(1, window_div) != pr->window_div) return fail(GSV_ERR_INVALID, "window_div differs from the plan recorder's");
    if (pr->finished) return fail(GSV_ERR_INVALID, "plan recorder already finished");
    if (n_feedback) return fail(GSV_ERR_INVALID, "a program compiled with feedback cannot be a plan call");
    window_div = pr->window_div;
  }
  window_div = std::max<uint32_t>(1, window_div);
  const bool spill = pr && pr->file;
  const bool keep = !spill && (!o || o->keep_trace) && !(pr && pr->single_image);
  auto fb = std::make_shared<std::vector<std::pair<uint32_t, uint32_t>>>();
  for (size_t i = 0; i < n_feedback; ++i) fb->push_back({fb_out_idx[i], fb_in_idx[i]});
  std::unique_ptr<gsv_program> p(new gsv_program());
  CompileOptions opt;
  if (const char* e = getenv("GSV_LDS_LIFETIME")) opt.lds_max_lifetime = uint32_t(atoi(e));  // tuning knobs (defaults are the measured best)
  if (const char* e = getenv("GSV_FUSE")) opt.fuse = atoi(e) != 0;
  if (const char* e = getenv("GSV_FUSE_DUP")) opt.fuse_dup_fanout = uint32_t(atoi(e));
  if (const char* e = getenv("GSV_ORDER_BY_READER")) opt.order_by_reader = atoi(e) != 0;
  if (const char* e = getenv("GSV_HBM_ARENA")) opt.hbm_arena_factor = uint32_t(atoi(e));
  if (const char* e = getenv("GSV_LDS_SLOTS")) opt.lds_slots = std::min<uint32_t>(uint32_t(atoi(e)), LDS_WINDOW_SLOTS);
  if (window_div > 1 || (pr && pr->single_image)) opt.lds_slots = std::min<uint32_t>(opt.lds_slots, LDS_WINDOW_SLOTS / window_div);
  p->window_div = window_div;
  p->has_decl = true;
  p->decl_inputs = r->inputs.size(); p->decl_outputs = r->outputs.size(); p->decl_gates = r->mode.trace().size();
  p->has_feedback = n_feedback != 0;
  // the trace: moved out of the recorder (consume_recorder) or copied
  auto trace = std::make_shared<Trace>();
  if (o && o->consume_recorder) { *trace = std::move(r->mode.trace()); r->mode.trace() = Trace(); }
  else *trace = r->mode.trace();
  auto inputs = std::make_shared<std::vector<uint32_t>>(r->inputs), outputs = std::make_shared<std::vector<uint32_t>>(r->outputs);
  gsv_program* const q = p.get();
  auto work = [q, trace, inputs, outputs, fb, opt, keep, spill, pr]() -> std::pair<int, std::string> {
    try {
      q->prog = compile_program(*trace, *inputs, *outputs, *fb, opt);
      for (size_t i = 0; i < q->prog.input_slots.size(); ++i)
        if (q->prog.input_slots[i] != SLOT_FIRST_INPUT + i) return {GSV_ERR_CIRCUIT, "internal: inputs are not slot-contiguous"};
      if (keep) q->src.reset(new ProgramSource{std::move(*trace), *inputs, *outputs, *fb, opt});
      else *trace = Trace();
      if (spill) pr->spill(q->prog);
      return {GSV_OK, std::string()};
    } catch (const std::exception& e) { return {GSV_ERR_CIRCUIT, e.what()};
    } catch (...) { return {GSV_ERR_CIRCUIT, "unknown exception"}; }
  };
  if (pr) { std::lock_guard<std::mutex> lk0(g_recorder_link_mu); std::lock_guard<std::mutex> lk(pr->mu); pr->compiled_for.push_back(q); q->for_recorder = pr; }
  if (o && o->background) {
    q->compiling = true;
    abi_compile_pool().submit([q, work] {
      auto res = work();
      { std::lock_guard<std::mutex> lk(q->cmu); q->compile_rc = res.first; q->compile_err = res.second; q->compiling = false; }
      q->ccv.notify_all();
    });
  } else {
    auto res = work();
    if (res.first) {
      unlink_from_recorder(q);
      return fail(res.first, res.second);
    }
  }
  *out = p.release();
  return GSV_OK;
  GSV_CATCH
}
int gsv_program_compile(gsv_recorder* r, const uint32_t* fb_out_idx, const uint32_t* fb_in_idx, size_t n_feedback, gsv_program** out) {
  return compile_impl(r, fb_out_idx, fb_in_idx, n_feedback, nullptr, out);
}
int gsv_program_compile_opts(gsv_recorder* r, const gsv_compile_opts* opts, gsv_program** out) { return compile_impl(r, nullptr, nullptr, 0, opts, out); }
int gsv_program_wait(gsv_program* p) {
  if (!p) return fail(GSV_ERR_INVALID, "null program");
  return program_ready(p);
}

int gsv_plan_wire_file(const gsv_plan* p, uint64_t* n_global_wires, uint64_t* max_program_slots) {
  if (!p || !p->finished) return fail(GSV_ERR_INVALID, "plan not finished");
  uint64_t mx = SLOT_FIRST_INPUT;
  for (const PlanCall& c : p->calls) mx = std::max<uint64_t>(mx, c.prog->prog.n_slots);
  if (n_global_wires) *n_global_wires = p->n_globals;
  if (max_program_slots) *max_program_slots = mx;
  return GSV_OK;
}
int gsv_plan_image_bytes(const gsv_plan* p, uint64_t* bytes, uint64_t* n_programs) {
  if (!p) return fail(GSV_ERR_INVALID, "null plan");
  std::set<const gsv_program*> seen;
  uint64_t b = 0;
  for (const PlanCall& c : p->calls) if (seen.insert(c.prog).second) b += c.prog->image_bytes();
  if (bytes) *bytes = b;
  if (n_programs) *n_programs = seen.size();
  return GSV_OK;
}

int gsv_plan_save(const gsv_plan* p, const char* path) {
  if (!p || !path || !p->finished) return fail(GSV_ERR_INVALID, "null argument / plan not finished");
  std::vector<const gsv_program*> progs;
  std::map<const gsv_program*, uint32_t> index;
  for (const PlanCall& c : p->calls)
    if (index.emplace(c.prog, uint32_t(progs.size())).second) {
      if (c.prog->device_only || c.prog->prog.spilled) return fail(GSV_ERR_INVALID, "this plan holds no program records on the host (loaded straight to a device / built straight to a file)");
      progs.push_back(c.prog);
    }
  PlanFileWriter w;
  { int rc = w.open_file(path); if (rc) return rc; }
  std::vector<uint64_t> off;
  for (const gsv_program* q : progs) off.push_back(w.append_program(q->prog, q->window_div));
  return w.finish(p, off, index);
}
// Build a plan and write it to `path` without ever holding it: every program is appended to the file by the worker that compiled
// it and its records are dropped (the verifier's plan is 41 GB of records; built in memory it peaks at ~54 GB of host RSS).
// Load the file with gsv_plan_load (with an engine: streamed to the device).  One image per program: GSV_PLAN_WINDOW_DIV=2|4.
static std::function<void(Program&)> spill_to(PlanFileWriter& w, uint32_t window_div) {
  return [&w, window_div](Program& g) {
    g.file_off = w.append_program(g, window_div);
    g.spilled = true;
    std::vector<StepDesc>().swap(g.steps); std::vector<AndRec>().swap(g.ands); std::vector<XorRec>().swap(g.xors); std::vector<uint32_t>().swap(g.ct_pos);
  };
}
static int finish_built_file(PlanFileWriter& w, const gsv_plan* plan) {
  std::vector<uint64_t> off;
  std::map<const gsv_program*, uint32_t> index;
  for (const gsv_program* q : plan->owned) {
    if (!q->prog.spilled) return fail(GSV_ERR_CIRCUIT, "internal: a program was not written to the plan file");
    index.emplace(q, uint32_t(off.size()));
    off.push_back(q->prog.file_off);
  }
  return w.finish(plan, off, index);
}
int gsv_plan_build_file(const char* spec, const char* units_csv, const char* path) {
  if (!spec || !units_csv || !path) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  uint32_t window_div = 1;
  { int rc = plan_window_div(&window_div); if (rc) return rc; }
  PlanFileWriter w;
  { int rc = w.open_file(path); if (rc) return rc; }
  gsv_plan* plan = nullptr;
  int rc = plan_from_circuit_impl(spec, units_csv, spill_to(w, window_div), &plan, window_div);
  if (rc) return rc;
  struct PlanOwner { gsv_plan* p; ~PlanOwner() { gsv_plan_destroy(p); } } po{plan};
  return finish_built_file(w, plan);
  GSV_CATCH
}
// TWO plan files from ONE recording of the circuit: every program is compiled twice — for 1 / window_div_a and for 1 / window_div_b of the
// LDS label window — by the worker that takes it off the recorder, and appended to both files.  A deployment that serves large batches
// (four instances per workgroup: window_div 4) AND small ones (full window: window_div 1) builds both plans for the price of one
// recording, which is the critical path of a build (bench.py: 94 s -> ~55 s to the first launch).  Each file is byte for byte what
// gsv_plan_build_file writes for its window_div (tools/plan_digest.py; tests/test_ext_host.py).
int gsv_plan_build_file_pair(const char* spec, const char* units_csv_a, const char* path_a, uint32_t window_div_a, const char* units_csv_b, const char* path_b, uint32_t window_div_b) {
  const char* units_csv = units_csv_a;
  if (!spec || !units_csv || !path_a || !path_b) return fail(GSV_ERR_INVALID, "null argument");
  for (uint32_t d : {window_div_a, window_div_b}) if (d != 1 && d != 2 && d != 4) return fail(GSV_ERR_INVALID, "window_div must be 1, 2 or 4");
  if (std::string(path_a) == path_b) return fail(GSV_ERR_INVALID, "the two plan files must differ");
  GSV_TRY
  PlanFileWriter wa, wb;
  { int rc = wa.open_file(path_a); if (rc) return rc; }
  { int rc = wb.open_file(path_b); if (rc) return rc; }
  gsv_plan *plan_a = nullptr, *plan_b = nullptr;
  DualBuild dual;
  dual.units_csv = units_csv_b; dual.window_div = window_div_b; dual.sink = spill_to(wb, window_div_b); dual.out = &plan_b;
  int rc = plan_from_circuit_impl(spec, units_csv, spill_to(wa, window_div_a), &plan_a, window_div_a, &dual);
  if (rc) return rc;
  struct PlanOwner { gsv_plan* p; ~PlanOwner() { gsv_plan_destroy(p); } } oa{plan_a}, ob{plan_b};
  rc = finish_built_file(wa, plan_a);
  if (rc) return rc;
  return finish_built_file(wb, plan_b);
  GSV_CATCH
}

int gsv_plan_load(const char* path, gsv_engine* e, gsv_plan** out) {
  if (!path || !out) return fail(GSV_ERR_INVALID, "null argument");
  Mapping mp;
  mp.fd = open(path, O_RDONLY);
  if (mp.fd < 0) return fail(GSV_ERR_INVALID, std::string("cannot open ") + path);
  struct stat st;
  if (fstat(mp.fd, &st) != 0 || size_t(st.st_size) < sizeof(PlanFileHeader)) return fail(GSV_ERR_INVALID, std::string(path) + ": not a plan file");
  mp.size = size_t(st.st_size);
  void* mm = mmap(nullptr, mp.size, PROT_READ, MAP_PRIVATE, mp.fd, 0);
  if (mm == MAP_FAILED) return fail(GSV_ERR_INVALID, std::string("cannot map ") + path);
  mp.base = static_cast<const uint8_t*>(mm);
  (void)madvise(mm, mp.size, MADV_SEQUENTIAL);
  size_t pos = 0;
  bool bad = false;
  auto take = [&](size_t n) -> const uint8_t* { const size_t m = pad16(n); if (m > mp.size - pos) { bad = true; return mp.base; } const uint8_t* q = mp.base + pos; pos += m; return q; };
  PlanFileHeader h;
  std::memcpy(&h, take(sizeof h), sizeof h);
  if (bad || std::memcmp(h.magic, PLAN_MAGIC, 8) != 0 || h.rec_sizes != plan_rec_sizes() || h.lds_window_slots != LDS_WINDOW_SLOTS)
    return fail(GSV_ERR_INVALID, std::string(path) + ": not a plan file of this engine build");
  if (h.table_off > mp.size || h.calls_off > h.table_off || (h.table_off & 15) || (h.calls_off & 15) || uint64_t(h.n_programs) > (mp.size - h.table_off) / 8)
    return fail(GSV_ERR_INVALID, std::string(path) + ": truncated or inconsistent plan file");
  const uint64_t* const table = reinterpret_cast<const uint64_t*>(mp.base + h.table_off);
  if (e) HIPCHK(hipSetDevice(e->device));
  const size_t bounce_bytes = 64u << 20;
  struct Bounce {
    void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
    ~Bounce() { for (void* q : buf) if (q) (void)hipHostFree(q); for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x); }
  } bounce_owner;
  void** bounce = bounce_owner.buf;
  hipEvent_t* bounce_ev = bounce_owner.ev;
  int bounce_next = 0;
  if (e)
    for (int b = 0; b < 2; ++b) {
      HIPCHK(hipHostMalloc(&bounce[b], bounce_bytes, hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&bounce_ev[b], hipEventDisableTiming));
      HIPCHK(hipEventRecord(bounce_ev[b], e->stream));
    }
  struct PlanOwner { gsv_plan* p; ~PlanOwner() { if (p) gsv_plan_destroy(p); } } po{new gsv_plan()};
  gsv_plan* plan = po.p;
  GSV_TRY
  for (uint32_t k = 0; k < h.n_programs; ++k) {
    PlanFileProgram m;
    if ((table[k] & 15) || table[k] < sizeof(PlanFileHeader) || table[k] > h.calls_off) { bad = true; break; }
    pos = size_t(table[k]);
    std::memcpy(&m, take(sizeof m), sizeof m);
    if (bad) break;
    gsv_program* q = new gsv_program();
    plan->owned.push_back(q);
    Program& g = q->prog;
    g.n_steps = uint32_t(m.n_steps); g.n_gates = m.n_gates; g.n_ct = m.n_ct; g.n_dead = m.n_dead; g.n_fused_free = m.n_fused_free;
    g.reads_lds = m.reads_lds; g.reads_hbm = m.reads_hbm; g.writes_lds = m.writes_lds; g.writes_hbm = m.writes_hbm;
    for (int i = 0; i < GATE_TYPE_COUNT; ++i) g.gate_count[i] = m.gate_count[i];
    g.n_slots = m.n_slots; g.n_lds_slots = m.n_lds_slots; g.lds_slots_limit = m.lds_slots_limit; g.fb_stage_base = m.fb_stage_base; g.and_depth = m.and_depth;
    g.n_and_steps = m.n_and_steps; g.max_step_width = m.max_step_width; g.peak_live = m.peak_live; g.and_terms = m.and_terms;
    if ((m.window_div != 1 && m.window_div != 2 && m.window_div != 4) || (m.and_terms != 2 && m.and_terms != 4)) { bad = true; break; }
    // The file is input: counts are checked against the file size BEFORE they are multiplied, slot counts against the record
    // format's 20-bit slot space, and every step's record ranges against the record arrays (the records themselves — 40 GB for the
    // verifier — are not re-validated: plan files live in a directory only their owner can write, bench.py / _plan_cache_path).
    const uint64_t lim = mp.size;
    if (m.n_steps > 0xFFFFFFFFull || m.n_steps > lim / sizeof(StepDesc) || m.n_ands > lim / sizeof(AndRec) || m.n_xors > lim / sizeof(XorRec) || m.n_ct_pos > lim / 4 || m.n_inputs > lim / 4 ||
        m.n_outputs > lim / 4 || m.n_ct_pos != m.n_ct || m.n_ands != m.n_ct || m.n_slots < SLOT_FIRST_INPUT + m.n_inputs || m.n_slots > SLOT_LDS_FLAG || m.lds_slots_limit > LDS_WINDOW_SLOTS ||
        m.n_lds_slots > m.lds_slots_limit) { bad = true; break; }
    q->window_div = m.window_div;
    const uint8_t* steps = take(m.n_steps * sizeof(StepDesc));
    const uint8_t* ands = take(m.n_ands * sizeof(AndRec));
    const uint8_t* xors = take(m.n_xors * sizeof(XorRec));
    const uint8_t* ctp = take(m.n_ct_pos * 4);
    const uint8_t* ins = take(m.n_inputs * 4);
    const uint8_t* outs = take(m.n_outputs * 4);
    if (bad) break;
    {
      const StepDesc* sd = reinterpret_cast<const StepDesc*>(steps);
      for (uint64_t i = 0; i < m.n_steps && !bad; ++i)
        bad = uint64_t(sd[i].and_off) + sd[i].and_cnt > m.n_ands || uint64_t(sd[i].xor_off) + sd[i].xor_cnt > m.n_xors;
      const uint32_t* cp = reinterpret_cast<const uint32_t*>(ctp);
      for (uint64_t i = 0; i < m.n_ct_pos && !bad; ++i) bad = cp[i] >= m.n_ct;
      const uint32_t *is = reinterpret_cast<const uint32_t*>(ins), *os = reinterpret_cast<const uint32_t*>(outs);
      for (uint64_t i = 0; i < m.n_inputs && !bad; ++i) bad = is[i] >= m.n_slots;
      for (uint64_t i = 0; i < m.n_outputs && !bad; ++i) bad = os[i] >= m.n_slots;  // (an LDS-window slot carries bit 20: rejected as well)
      if (bad) break;
    }
    g.input_slots.assign(reinterpret_cast<const uint32_t*>(ins), reinterpret_cast<const uint32_t*>(ins) + m.n_inputs);
    g.output_slots.assign(reinterpret_cast<const uint32_t*>(outs), reinterpret_cast<const uint32_t*>(outs) + m.n_outputs);
    if (!e) {  // host copy: a complete program (hostsim, saving again, uploading to any device later)
      g.steps.assign(reinterpret_cast<const StepDesc*>(steps), reinterpret_cast<const StepDesc*>(steps) + m.n_steps);
      g.ands.assign(reinterpret_cast<const AndRec*>(ands), reinterpret_cast<const AndRec*>(ands) + m.n_ands);
      g.xors.assign(reinterpret_cast<const XorRec*>(xors), reinterpret_cast<const XorRec*>(xors) + m.n_xors);
      g.ct_pos.assign(reinterpret_cast<const uint32_t*>(ctp), reinterpret_cast<const uint32_t*>(ctp) + m.n_ct_pos);
      continue;
    }
    q->device_only = true;
    DevProgram d;
    // Records go from the file to the device through two page-locked bounce buffers (pread + async copy): the process never
    // holds more than the buffers, whatever the size of the plan (the mapping above is only dereferenced for the metadata).
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {  // same padding rule as upload_program
      HIPCHK(hipMalloc(dst, bytes + 32));
      HIPCHK(hipMemsetAsync(*dst, 0, bytes + 32, e->stream));
      d.bytes += bytes;
      if (!bytes) return GSV_OK;
      if (!src) return fail(GSV_ERR_INVALID, "internal: missing source");
      size_t off = size_t(static_cast<const uint8_t*>(src) - mp.base);
      for (size_t done_b = 0; done_b < bytes;) {
        const size_t nb = std::min(bounce_bytes, bytes - done_b);
        const int b = bounce_next;
        bounce_next ^= 1;
        HIPCHK(hipEventSynchronize(bounce_ev[b]));  // the previous copy out of this buffer has finished
        size_t got = 0;
        while (got < nb) {
          const ssize_t r = pread(mp.fd, static_cast<uint8_t*>(bounce[b]) + got, nb - got, off_t(off + done_b + got));
          if (r <= 0) return fail(GSV_ERR_INVALID, std::string(path) + ": short read");
          got += size_t(r);
        }
        HIPCHK(hipMemcpyAsync(static_cast<uint8_t*>(*dst) + done_b, bounce[b], nb, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipEventRecord(bounce_ev[b], e->stream));
        done_b += nb;
      }
      return GSV_OK;
    };
    int rc = GSV_OK;
    void** dsts[7] = {&d.steps, &d.ands, &d.xors, &d.fb_src, &d.fb_dst, &d.out_slots, &d.ct_pos};
    const void* srcs[7] = {steps, ands, xors, nullptr, nullptr, outs, ctp};
    const size_t lens[7] = {size_t(m.n_steps) * sizeof(StepDesc), size_t(m.n_ands) * sizeof(AndRec), size_t(m.n_xors) * sizeof(XorRec), 0, 0, size_t(m.n_outputs) * 4, size_t(m.n_ct_pos) * 4};
    for (int i = 0; i < 7 && rc == GSV_OK; ++i) rc = up(dsts[i], srcs[i], lens[i]);
    // file it before checking rc: gsv_plan_destroy then releases whatever was allocated
    q->dev[{e->device, 1}] = d;  // image key 1 = `prog` itself (gsv_program::image_key)
    if (rc != GSV_OK) return rc;
    q->loaded_image_bytes = d.bytes;
  }
  if (e) { HIPCHK(hipStreamSynchronize(e->stream)); plan->device = e->device; }
  pos = size_t(h.calls_off);
  for (uint32_t k = 0; k < h.n_calls && !bad; ++k) {
    uint32_t hdr[4];
    std::memcpy(hdr, take(sizeof hdr), sizeof hdr);
    if (bad || hdr[0] >= plan->owned.size()) { bad = true; break; }
    const uint8_t* ig = take(size_t(hdr[1]) * 4);
    const uint8_t* og = take(size_t(hdr[2]) * 4);
    if (bad) break;
    const Program& g = plan->owned[hdr[0]]->prog;
    if (hdr[1] != g.input_slots.size() || hdr[2] != g.output_slots.size()) { bad = true; break; }
    int rc = gsv_plan_add_call(plan, plan->owned[hdr[0]], reinterpret_cast<const uint32_t*>(ig), reinterpret_cast<const uint32_t*>(og));
    if (rc) return rc;
  }
  if (!bad) {
    const uint8_t* og = take(size_t(h.n_outputs) * 4);
    if (!bad) {
      int rc = gsv_plan_finish(plan, h.n_inputs, reinterpret_cast<const uint32_t*>(og), h.n_outputs);
      if (rc) return rc;
    }
  }
  if (bad || plan->n_gates != h.n_gates || plan->n_ct != h.n_ct || plan->n_globals != h.n_globals) return fail(GSV_ERR_INVALID, std::string(path) + ": truncated or inconsistent plan file");
  po.p = nullptr;
  *out = plan;
  return GSV_OK;
  GSV_CATCH
}
