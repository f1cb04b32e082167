// Host runtime + C ABI (include/gsv_engine.h) of the MI355X garbling engine.
// Device memory, streams and events are plain HIP runtime calls; there is NO CPU execution path for
// garble/evaluate — without a HIP device gsv_engine_create fails with GSV_ERR_DEVICE.
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/gsv_engine.h"
#include "../gadgets/circuits.hpp"
#include "host_crypto.hpp"
#include "kernel_api.h"
#include "plan_builder.hpp"
#include "schedule.hpp"
#include "program.hpp"

using namespace gsv;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define GSV_TRY try {
#define GSV_CATCH                                                                 \
  }                                                                               \
  catch (const std::exception& e) { return fail(GSV_ERR_CIRCUIT, e.what()); }     \
  catch (...) { return fail(GSV_ERR_CIRCUIT, "unknown exception"); }

// A failed HIP call leaves its error behind for hipGetLastError(); the kernel launchers report hipGetLastError(), so the stale
// error of e.g. an out-of-memory hipMalloc would make every later launch of the process look failed: clear it here.
#define HIPCHK(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess) {                                                                                  \
      (void)hipGetLastError();                                                                               \
      return fail(GSV_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                        \
    }                                                                                                        \
  } while (0)

// Large device allocations report what was asked for and what the device had left.
static int dev_alloc(void** p, size_t bytes, const char* what) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e == hipSuccess) return GSV_OK;
  (void)hipGetLastError();
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  char msg[256];
  std::snprintf(msg, sizeof msg, "hipMalloc of %.2f GB for %s failed (%s): device has %.2f of %.2f GB free", double(bytes) / 1e9, what, hipGetErrorString(e), double(free_b) / 1e9,
                double(total_b) / 1e9);
  return fail(GSV_ERR_DEVICE, msg);
}
#define DEVALLOC(p, bytes, what) do { int _rc = dev_alloc(reinterpret_cast<void**>(p), (bytes), (what)); if (_rc) return _rc; } while (0)
// ---- deferred release ---------------------------------------------------------------------------------------------------------------
// hipFree / hipStreamDestroy synchronise the device.  While a streaming pass runs that is at best a stall of whoever destroys something
// and at worst a deadlock: a ring pass waits for the host's stream position, which waits for a sink / source callback — and a host that
// drops a session, plan, program or engine FROM that callback (a Rust `Drop` inside `CiphertextHandler::handle`, Python's collector on
// the callback thread: profiles/r05_debug/) would wait in hipFree for that very pass until the device's watchdog ends it.  The destroy
// entry points therefore never free while a streaming pass is in flight in this process: the request is queued and runs, in order, when
// the last pass in flight has synchronised (the handle is invalid for the host from the moment destroy returns, as always).  Outside a
// pass a destroy runs at once, under the gate's lock: a pass that starts meanwhile waits for it instead of being stalled by it.
namespace {
struct ReleaseGate {
  std::recursive_mutex mu;                      // recursive: a queued plan destroy runs its programs' destroys
  int active = 0;                               // streaming passes in flight (any session of this process)
  std::vector<std::function<void()>> pending;   // destroy requests that arrived meanwhile, in arrival order
  uint64_t n_deferred = 0;                      // statistics (gsv_deferred_release_count)
};
ReleaseGate& release_gate() { static ReleaseGate g; return g; }
// First local of every streaming entry point: declared before anything else so that it is destroyed LAST — a session destroyed from
// its own pass's callback is still alive while the entry point uses it.
struct PassGuard {
  PassGuard() { ReleaseGate& g = release_gate(); std::lock_guard<std::recursive_mutex> lk(g.mu); ++g.active; }
  ~PassGuard() {
    ReleaseGate& g = release_gate();
    std::lock_guard<std::recursive_mutex> lk(g.mu);
    if (--g.active != 0) return;
    std::vector<std::function<void()>> run;
    run.swap(g.pending);
    for (auto& f : run) f();
  }
  PassGuard(const PassGuard&) = delete;
  PassGuard& operator=(const PassGuard&) = delete;
};
void release_or_defer(std::function<void()> fn) {
  ReleaseGate& g = release_gate();
  std::lock_guard<std::recursive_mutex> lk(g.mu);
  if (g.active > 0) { g.pending.push_back(std::move(fn)); ++g.n_deferred; return; }
  fn();
}
}  // namespace

// A stream that must make progress WHILE a window runs on the engine's stream (the drain's gathers and copies, the other half of a
// garble -> evaluate pair).  The runtime multiplexes streams onto a few hardware queues per priority level, in order within a queue: a
// side stream that lands on the main stream's queue would sit behind the running window — which, with a ciphertext ring, itself waits
// for that side stream's work (observed: the ring stalls until the device watchdog fires, depending on how many streams the process
// had created before).  Streams of another priority level come from another pool of hardware queues, so these ask for the highest.
// (Not for the evaluator of a garble -> evaluate pair: two long launches on queues of DIFFERENT priority, whichever way round, took
// 46.5 s for the verifier instead of 42.9 s on equal terms — ensure_pair probes for a stream of the same priority that overlaps.)
static hipError_t create_side_stream(hipStream_t* st) {
  int least = 0, greatest = 0;
  const char* off = getenv("GSV_SIDE_STREAM_PRIORITY");
  if ((off && off[0] == '0') || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
}

struct gsv_recorder {
  RecordMode mode;
  std::vector<uint32_t> inputs, outputs;  // SSA ids
  bool outputs_declared = false;
};

struct DevProgram {
  void *steps = nullptr, *ands = nullptr, *xors = nullptr, *fb_src = nullptr, *fb_dst = nullptr, *out_slots = nullptr, *ct_pos = nullptr;
  size_t bytes = 0;
};

// What a program was compiled from: kept so that the half-window variant (two instances per workgroup) can be
// compiled the first time a session needs it.
struct ProgramSource {
  Trace trace;
  std::vector<uint32_t> inputs, outputs;
  std::vector<std::pair<uint32_t, uint32_t>> feedback;
  CompileOptions opt;
};

struct gsv_program {
  Program prog;                    // compiled for 1/window_div of the LDS label window: serves every layout of up to window_div instances per workgroup
  std::map<uint32_t, std::unique_ptr<Program>> variants;  // instances per workgroup (2, 4) -> the program compiled for that share of the window, on demand, from `src`
  uint32_t window_div = 1;         // 1: full window (the other layouts are compiled on demand); 2 / 4: `prog` itself was compiled for half / a quarter of the
                                   // window and is the only image (GSV_PLAN_WINDOW_DIV: no second variant, no trace kept)
  bool device_only = false;        // loaded by gsv_plan_load straight into device memory: the host keeps the metadata, not the records
  uint64_t loaded_image_bytes = 0; // size of the records of a device_only program
  uint32_t image_key(uint32_t ni) const { return ni <= window_div ? 1u : ni; }  // which compiled image a layout runs
  const Program& variant(uint32_t ni) const { return ni <= window_div ? prog : *variants.at(ni); }
  std::unique_ptr<ProgramSource> src;
  std::mutex mu;
  std::map<std::pair<int, int>, DevProgram> dev;  // per (device, instances per workgroup)
  // gsv_program_compile_opts(background = 1): the handle exists at once, `prog` is filled by a worker of the library's compile pool.  What a
  // plan recorder needs to take a call of the program (arity, gate count) is known from the recording and kept here; everything that
  // reads `prog` goes through program_ready() first.
  uint64_t decl_inputs = 0, decl_outputs = 0, decl_gates = 0;
  bool has_feedback = false, has_decl = false;
  struct gsv_plan_recorder* for_recorder = nullptr;  // compiled with gsv_compile_opts.for_plan: registered there until either side is destroyed (g_recorder_link_mu)
  std::mutex cmu;
  std::condition_variable ccv;
  bool compiling = false;
  int compile_rc = 0;
  std::string compile_err;
  size_t image_bytes() const {
    if (device_only) return size_t(loaded_image_bytes);
    return prog.steps.size() * sizeof(StepDesc) + prog.ands.size() * sizeof(AndRec) + prog.xors.size() * sizeof(XorRec) +
           (prog.fb_src_slot.size() * 2 + prog.output_slots.size() + prog.ct_pos.size()) * sizeof(uint32_t);
  }
};

// A plan = a sequence of calls to compiled programs over ONE wire file per instance (component-level programs: the
// reference instantiates the same component shapes thousands of times, streaming_mode.rs:150-247).  Wires that cross
// calls live in a "global" region behind the programs' own slots; a call copies its inputs in, runs, copies its outputs out.
struct PlanCall {
  gsv_program* prog;
  std::vector<uint32_t> in_globals, out_globals;
  uint64_t gid_off = 0, ct_off = 0;  // gate ids / ciphertext records consumed by the calls before this one
};
struct gsv_plan {
  std::vector<gsv_program*> owned;  // programs created by gsv_plan_from_circuit (destroyed with the plan)
  std::vector<PlanCall> calls;
  uint32_t n_globals = 0, n_inputs = 0;
  std::vector<uint32_t> outputs;
  uint64_t n_gates = 0, n_ct = 0;
  bool finished = false;
  int device = -1;  // >= 0: loaded by gsv_plan_load straight into that device's memory (device_only programs): serves that device only
};

struct gsv_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  void* te = nullptr;  // device T-tables
};

struct gsv_drain;
struct PairState;
extern "C" {
static void destroy_drain(gsv_drain* d);
static void destroy_pair(PairState* ps);
}
struct gsv_session {
  gsv_engine* e = nullptr;
  gsv_program* p = nullptr;
  DevProgram dp;
  size_t n_inst = 0;
  uint64_t replays = 1, ct_cap = 1;
  void *W = nullptr, *VB = nullptr, *CT = nullptr, *delta = nullptr, *out = nullptr, *out_bits = nullptr, *in_bits = nullptr, *step_clock = nullptr, *ct_stage = nullptr, *ct_gate = nullptr;
  size_t ct_gate_bytes = 0;  // capacity of ct_gate and of every buffer of ct_gate_more (ensure_ct_gate)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint32_t ni = 1;  // instances per workgroup of this session's launches
  // plan sessions: `facade` stands in for the program (slots = wire-file stride, inputs / outputs in the global region)
  const gsv_plan* plan = nullptr;
  Program facade;
  uint32_t global_base = 0;  // first slot of the plan's global region
  bool plan_retain = true;   // plan sessions: whole ciphertext stream kept on the device (else one call block: streaming only)
  uint64_t plan_max_block = 0;  // ciphertext records per instance of the device block: the largest WINDOW of the schedule
  uint64_t plan_max_segment = 0;  // ... of a gate-order buffer: the largest drain SEGMENT (schedule.hpp)
  bool ct_ring = false;              // the device block is a ring of plan_max_block records (schedule.hpp, SchedParams::ring_ct)
  std::string ring_diag;  // ring mode: the longest interval between two publications of the host's position in the last pass, and where it went
  unsigned long long* host_ct_pos = nullptr;  // ring mode: the host's stream-position counter (page-locked, mapped into the device)
  unsigned long long* dev_ct_pos = nullptr;   // ... its device address
  hipStream_t aux_stream = nullptr;  // gather kernels and flag polls of the drain, beside the running window
  uint32_t* host_done = nullptr;     // per call of the plan: workgroups that have finished it in the current pass (mapped host memory, written by the device)
  uint32_t* dev_done = nullptr;      // ... its device address
  struct CallDev { DevProgram dp; };
  std::vector<CallDev> call_dev;
  // Call-level schedule (schedule.hpp): windows of consecutive calls; the calls of a window run as a dataflow inside ONE launch
  // (grid.y = calls), each waiting for the completion flags of the calls it depends on.  Device tables in stream order: the call
  // descriptors, the concatenated wire hand-over lists (globals -> the call's scratch region -> globals), the dependency lists
  // (window-relative call indices) and the completion flags [instance group][call] (compared with the launch epoch: never reset).
  Schedule sched;
  void *d_calls = nullptr, *d_copy_src = nullptr, *d_copy_dst = nullptr, *d_deps = nullptr, *d_flags = nullptr, *d_error = nullptr;
  uint32_t flag_stride = 0, epoch = 0;
  // Safe-schedule fallback (round 6): the options the session was created with, the host's last inputs (re-staged when a pass is
  // repeated) and what the big allocations hold, so that a second schedule can be installed into the same session.
  gsv_plan_session_opts opts{};
  bool safe_mode = false;                    // the schedule is the safe one: ONE call per launch, no dependency waits on the device
  bool dep_fault = false;                    // the last pass ended with status 1 (a dependency wait gave up)
  uint64_t n_fallbacks = 0;
  size_t w_slots_cap = 0;                    // 16-byte slots per instance W / VB were allocated for
  uint64_t ct_records_cap = 0;               // ciphertext records per instance CT was allocated for
  std::vector<uint8_t> stash_delta, stash_consts, stash_inputs, stash_bits;
  int stash_kind = 0;                        // 0 nothing, 1 garble inputs, 2 evaluate inputs
  size_t drain_instances = 0;               // streaming calls: only the first this-many instances' streams leave the device (0 = all)
  uint64_t next_call = 0;                   // streaming slices: the call the next slice must start with
  bool unchecked_slices = false;            // benchmarks may garble slices out of order (results are then meaningless)
  void* plan_out_slots = nullptr;
  const Program& prog() const { return plan ? facade : p->variant(ni); }
  const Program& call_prog(size_t k) const { return plan->calls[k].prog->variant(ni); }
  uint32_t first_input_slot() const { return plan ? global_base : SLOT_FIRST_INPUT; }
  bool ran = false, last_eval = false, garbled = false;
  int hasher = 0;  // 0 AesNiHasher, 1 Blake3Hasher
  std::vector<uint64_t> ct_uploaded;  // per instance: records supplied by gsv_session_upload_ciphertexts
  struct gsv_drain* drain = nullptr;   // streaming drain: copy streams, pinned buffers, per-instance MAC states (created on first use)
  std::vector<void*> ct_gate_more;     // further gate-order buffers of the drain pipeline (ct_gate is the first)
  void* ct_alt = nullptr;              // garble -> evaluate on the device: the second program-order ciphertext block
  struct PairState* pair = nullptr;    // ... and its stream / events (created on first use)
  uint64_t ct_stride() const { return plan ? (plan_retain ? plan->n_ct : plan_max_block) : ct_cap * p->prog.n_ct; }  // n_ct does not depend on the variant
};

// Failure paths release whatever was allocated so far through the public destroy functions (a failed hipMalloc on a
// multi-GB session must not leave the GPU full).
struct SessionDeleter { void operator()(gsv_session* s) const { gsv_session_destroy(s); } };
struct EngineDeleter { void operator()(gsv_engine* e) const { gsv_engine_destroy(e); } };
typedef std::unique_ptr<gsv_session, SessionDeleter> SessionPtr;
typedef std::unique_ptr<gsv_engine, EngineDeleter> EnginePtr;

extern "C" {

const char* gsv_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------- recorder
int gsv_recorder_create(gsv_recorder** out) {
  if (!out) return fail(GSV_ERR_INVALID, "null out");
  *out = new gsv_recorder();
  return GSV_OK;
}
void gsv_recorder_destroy(gsv_recorder* r) { delete r; }

int gsv_recorder_allocate_wire(gsv_recorder* r, uint16_t credits, uint64_t* wire_out) {
  if (!r || !wire_out) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  *wire_out = r->mode.allocate_wire(credits);
  return GSV_OK;
  GSV_CATCH
}
int gsv_recorder_allocate_wires(gsv_recorder* r, size_t n, uint64_t* first_wire_out) {
  if (!r || !first_wire_out || n == 0) return fail(GSV_ERR_INVALID, "null argument / n == 0");
  GSV_TRY
  *first_wire_out = r->mode.allocate_wire(1);
  for (size_t i = 1; i < n; ++i) (void)r->mode.allocate_wire(1);
  return GSV_OK;
  GSV_CATCH
}
int gsv_recorder_declare_input(gsv_recorder* r, uint64_t wire) {
  if (!r) return fail(GSV_ERR_INVALID, "null recorder");
  GSV_TRY
  r->inputs.push_back(r->mode.define_input(wire));
  return GSV_OK;
  GSV_CATCH
}
int gsv_recorder_push_gates(gsv_recorder* r, const gsv_gate* gates, size_t n) {
  if (!r || (!gates && n)) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  for (size_t i = 0; i < n; ++i) {
    if (gates[i].gate_type > 10) return fail(GSV_ERR_INVALID, "gate_type out of range");
    r->mode.evaluate_gate(Gate{gates[i].wire_a, gates[i].wire_b, gates[i].wire_c, GateType(gates[i].gate_type)});
  }
  return GSV_OK;
  GSV_CATCH
}
int gsv_recorder_declare_outputs(gsv_recorder* r, const uint64_t* wires, size_t n) {
  if (!r || (!wires && n)) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  r->outputs.clear();
  for (size_t i = 0; i < n; ++i) r->outputs.push_back(r->mode.current(wires[i]));
  r->outputs_declared = true;
  return GSV_OK;
  GSV_CATCH
}
int gsv_recorder_record_circuit(gsv_recorder* r, const char* spec) {
  if (!r || !spec) return fail(GSV_ERR_INVALID, "null argument");
  GSV_TRY
  if (!r->inputs.empty() || r->mode.trace().size()) return fail(GSV_ERR_INVALID, "recorder already holds a circuit");
  NamedCircuit nc = make_circuit(spec);
  StreamingRunner run(r->mode, nc.n_inputs, nc.fn);  // two-pass credit driver, circuit/mod.rs:253-301
  const Wires& in = run.prepare();
  for (WireId w : in) r->inputs.push_back(r->mode.define_input(w));
  const Wires& out = run.execute();
  for (WireId w : out) r->outputs.push_back(r->mode.current(w));
  r->outputs_declared = true;
  return GSV_OK;
  GSV_CATCH
}

int gsv_recorder_counts(const gsv_recorder* r, uint64_t* n_inputs, uint64_t* n_outputs, uint64_t* n_gates) {
  if (!r) return fail(GSV_ERR_INVALID, "null recorder");
  if (n_inputs) *n_inputs = r->inputs.size();
  if (n_outputs) *n_outputs = r->outputs.size();
  if (n_gates) *n_gates = const_cast<gsv_recorder*>(r)->mode.trace().size();
  return GSV_OK;
}

// ---------------------------------------------------------------- program
static int program_ready(const gsv_program* cp);
static void unlink_from_recorder(gsv_program* p);
static void program_destroy_now(gsv_program* p) {
  (void)program_ready(p);  // a background compilation still writes into it
  unlink_from_recorder(p);  // its plan recorder must not wait on a destroyed program (gsv_plan_recorder_finish / _destroy)
  std::set<void*> freed;  // a half-window image loaded from a plan file is filed under both layouts
  for (auto& kv : p->dev) {
    (void)hipSetDevice(kv.first.first);
    for (void* q : {kv.second.steps, kv.second.ands, kv.second.xors, kv.second.fb_src, kv.second.fb_dst, kv.second.out_slots, kv.second.ct_pos})
      if (q && freed.insert(q).second) (void)hipFree(q);
  }
  delete p;
}
void gsv_program_destroy(gsv_program* p) {
  if (!p) return;
  release_or_defer([p] { program_destroy_now(p); });
}
int gsv_program_get_info(const gsv_program* p, gsv_program_info* info) {
  if (!p || !info) return fail(GSV_ERR_INVALID, "null argument");
  { int rc = program_ready(p); if (rc) return rc; }
  const Program& g = p->prog;
  std::memset(info, 0, sizeof *info);
  info->n_inputs = g.input_slots.size(); info->n_outputs = g.output_slots.size();
  info->n_gates = g.n_gates; info->n_ciphertexts = g.n_ct; info->n_dead = g.n_dead;
  for (int i = 0; i < 11; ++i) info->gate_count[i] = g.gate_count[i];
  info->n_steps = g.n_steps; info->and_depth = g.and_depth; info->n_and_steps = g.n_and_steps; info->max_step_width = g.max_step_width;
  info->n_slots = g.n_slots; info->peak_live = g.peak_live; info->device_bytes = p->image_bytes();
  info->n_lds_slots = g.n_lds_slots; info->reads_lds = g.reads_lds; info->reads_hbm = g.reads_hbm; info->writes_lds = g.writes_lds; info->writes_hbm = g.writes_hbm;
  info->n_fused_free = g.n_fused_free;
  info->and_terms = g.and_terms;
  return GSV_OK;
}

// ---------------------------------------------------------------- engine
int gsv_engine_create(int device, gsv_engine** out) {
  if (!out) return fail(GSV_ERR_INVALID, "null out");
  int n = 0;
  hipError_t er = hipGetDeviceCount(&n);
  if (er != hipSuccess || n <= 0) return fail(GSV_ERR_DEVICE, "no HIP device available: the garbling engine has no CPU fallback");
  if (device < 0 || device >= n) return fail(GSV_ERR_DEVICE, "device index out of range");
  HIPCHK(hipSetDevice(device));
  EnginePtr e(new gsv_engine());
  e->device = device;
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  const AesTables& t = AesTables::fixed_key();
  HIPCHK(hipMalloc(&e->te, sizeof t.te));
  HIPCHK(hipMemcpy(e->te, t.te, sizeof t.te, hipMemcpyHostToDevice));
  if (gsvk_upload_round_keys(t.rk) != 0) return fail(GSV_ERR_DEVICE, "round key upload failed");
  *out = e.release();
  return GSV_OK;
}
static void engine_destroy_now(gsv_engine* e) {
  (void)hipSetDevice(e->device);
  if (e->te) (void)hipFree(e->te);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}
void gsv_engine_destroy(gsv_engine* e) {
  if (!e) return;
  release_or_defer([e] { engine_destroy_now(e); });
}
uint64_t gsv_deferred_release_count(void) {
  ReleaseGate& g = release_gate();
  std::lock_guard<std::recursive_mutex> lk(g.mu);
  return g.n_deferred;
}

int gsv_labels_from_seed(uint64_t seed, size_t n_inputs, uint8_t delta[16], uint8_t false_label0[16], uint8_t true_label0[16], uint8_t* input_label0) {
  if (!delta || !false_label0 || !true_label0 || (!input_label0 && n_inputs)) return fail(GSV_ERR_INVALID, "null argument");
  ChaCha20Seed rng(seed);
  rng.next_label(delta);
  rng.next_label(false_label0);
  rng.next_label(true_label0);
  for (size_t i = 0; i < n_inputs; ++i) rng.next_label(input_label0 + 16 * i);
  return GSV_OK;
}

// ---------------------------------------------------------------- sessions
// The variant of a program for `ni` instances per workgroup (1/ni of the LDS window each), compiled on first use.  Throws on failure; p->mu held by the caller.
static void compile_window_variant(gsv_program* p, uint32_t ni) {
  if (ni <= p->window_div || p->variants.count(ni)) return;
  if (!p->src) gsv_panic("this program was compiled for 1/" + std::to_string(p->window_div) + " of the LDS window and its trace was not kept: it cannot serve " + std::to_string(ni) +
                         " instances per workgroup (build the plan with GSV_PLAN_WINDOW_DIV=" + std::to_string(ni) + ")");
  CompileOptions opt = p->src->opt;
  opt.lds_slots = std::min<uint32_t>(opt.lds_slots, LDS_WINDOW_SLOTS / ni);
  std::unique_ptr<Program> q(new Program(compile_program(p->src->trace, p->src->inputs, p->src->outputs, p->src->feedback, opt)));
  for (size_t i = 0; i < q->input_slots.size(); ++i)
    if (q->input_slots[i] != SLOT_FIRST_INPUT + i) gsv_panic("internal: inputs are not slot-contiguous");
  p->variants[ni] = std::move(q);
}
// Instances per workgroup of a session: as many (1, 2, 4) as keep every CU busy — the latency-bound narrow steps then cost their fixed
// time once for all of them (kernels.hip) — limited to what the programs can serve; GSV_INSTANCES_PER_WG=1|2|4 overrides.
static uint32_t choose_instances_per_wg(size_t n_instances, int n_cus, uint32_t max_servable) {
  uint32_t ni = n_instances > 2 * size_t(n_cus) ? 4u : n_instances > size_t(n_cus) ? 2u : 1u;
  if (const char* ev = getenv("GSV_INSTANCES_PER_WG")) { int v = atoi(ev); if (v == 1 || v == 2 || v == 4) ni = uint32_t(v); }
  while (ni > 1 && (ni > max_servable || ni > n_instances)) ni /= 2;
  return ni;
}
static int upload_program(gsv_engine* e, gsv_program* p, uint32_t ni, DevProgram* out) {
  std::lock_guard<std::mutex> lk(p->mu);
  // one image per compiled variant: a program compiled for a share of the window serves every layout up to it from ONE copy in HBM
  // (the verifier plan's images are 41 GB)
  const int key = int(p->image_key(ni));
  auto it = p->dev.find({e->device, key});
  if (it != p->dev.end()) { *out = it->second; return GSV_OK; }
  // a program loaded by gsv_plan_load(path, engine) has no host copy of its records: there is nothing to upload to another device
  if (p->prog.spilled) return fail(GSV_ERR_INVALID, "this program's records were written to a plan file and dropped (gsv_plan_build_file / a plan recorder with a plan file): load the file with gsv_plan_load");
  if (p->device_only) return fail(GSV_ERR_INVALID, "this program was loaded straight into another device's memory (gsv_plan_load with an engine): it has no image for device " + std::to_string(e->device));
  if (ni > p->window_div) {  // first session with this many instances per workgroup: compile for that share of the LDS window
    GSV_TRY
    compile_window_variant(p, ni);
    GSV_CATCH
  }
  DevProgram d;
  const Program& g = p->variant(ni);
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    // +32 bytes of zero padding: the kernel's record prefetch reads 24 bytes wherever a lane's record starts
    HIPCHK(hipMalloc(dst, bytes + 32));
    HIPCHK(hipMemset(*dst, 0, bytes + 32));
    if (bytes) HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    d.bytes += bytes;
    return GSV_OK;
  };
  int rc;
  if ((rc = up(&d.steps, g.steps.data(), g.steps.size() * sizeof(StepDesc)))) return rc;
  if ((rc = up(&d.ands, g.ands.data(), g.ands.size() * sizeof(AndRec)))) return rc;
  if ((rc = up(&d.xors, g.xors.data(), g.xors.size() * sizeof(XorRec)))) return rc;
  if ((rc = up(&d.fb_src, g.fb_src_slot.data(), g.fb_src_slot.size() * 4))) return rc;
  if ((rc = up(&d.fb_dst, g.fb_dst_slot.data(), g.fb_dst_slot.size() * 4))) return rc;
  if ((rc = up(&d.out_slots, g.output_slots.data(), g.output_slots.size() * 4))) return rc;
  if ((rc = up(&d.ct_pos, g.ct_pos.data(), g.ct_pos.size() * 4))) return rc;
  p->dev[{e->device, key}] = d;
  *out = d;
  return GSV_OK;
}

int gsv_session_create(gsv_engine* e, const gsv_program* cp, size_t n_instances, uint64_t replays, uint64_t ct_capacity_replays, gsv_session** out) {
  if (!e || !cp || !out || n_instances == 0 || replays == 0) return fail(GSV_ERR_INVALID, "bad argument");
  gsv_program* p = const_cast<gsv_program*>(cp);
  { int rc = program_ready(p); if (rc) return rc; }
  if (ct_capacity_replays == 0 || ct_capacity_replays > replays) ct_capacity_replays = replays;
  if (replays > 0xFFFFFFFFull) return fail(GSV_ERR_INVALID, "too many replays");
  HIPCHK(hipSetDevice(e->device));
  SessionPtr s(new gsv_session());
  s->e = e; s->p = p; s->n_inst = n_instances; s->replays = replays; s->ct_cap = ct_capacity_replays;
  s->ct_uploaded.assign(n_instances, 0);
  // Two instances per workgroup once there are more instances than CUs (each then works with half of the LDS label
  // window, see kernels.hip); GSV_INSTANCES_PER_WG=1|2 overrides.
  {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, e->device));
    s->ni = choose_instances_per_wg(n_instances, prop.multiProcessorCount, p->src ? 4u : p->window_div);
  }
  int rc = upload_program(e, p, s->ni, &s->dp);
  if (rc) return rc;
  const Program& g = s->prog();
  DEVALLOC(&s->W, n_instances * size_t(g.n_slots) * 16, "the wire files");
  HIPCHK(hipMalloc(&s->VB, n_instances * size_t(g.n_slots)));
  HIPCHK(hipMemset(s->VB, 0, n_instances * size_t(g.n_slots)));
  size_t ct_bytes = n_instances * size_t(s->ct_stride()) * 16;
  DEVALLOC(&s->CT, ct_bytes, "the ciphertext blocks");
  HIPCHK(hipMalloc(&s->delta, n_instances * 16));
  HIPCHK(hipMalloc(&s->out, n_instances * g.output_slots.size() * 16 + 16));
  HIPCHK(hipMalloc(&s->out_bits, n_instances * g.output_slots.size() + 16));
  HIPCHK(hipMalloc(&s->in_bits, n_instances * g.input_slots.size() + 16));
  HIPCHK(hipEventCreate(&s->ev0));
  HIPCHK(hipEventCreate(&s->ev1));
  *out = s.release();
  return GSV_OK;
}
static void session_destroy_now(gsv_session* s) {
  (void)hipSetDevice(s->e->device);
  (void)hipStreamSynchronize(s->e->stream);
  for (void* q : {s->W, s->VB, s->CT, s->delta, s->out, s->out_bits, s->in_bits, s->step_clock, s->ct_stage, s->ct_gate}) if (q) (void)hipFree(q);
  for (void* q : {s->d_calls, s->d_copy_src, s->d_copy_dst, s->d_deps, s->d_flags, s->d_error}) if (q) (void)hipFree(q);
  if (s->plan_out_slots) (void)hipFree(s->plan_out_slots);
  for (void* q : s->ct_gate_more) if (q) (void)hipFree(q);
  if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
  if (s->host_done) (void)hipHostFree(s->host_done);
  if (s->host_ct_pos) (void)hipHostFree(s->host_ct_pos);
  destroy_drain(s->drain);
  destroy_pair(s->pair);
  if (s->ct_alt) (void)hipFree(s->ct_alt);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  delete s;
}
void gsv_session_destroy(gsv_session* s) {
  if (!s) return;
  release_or_defer([s] { session_destroy_now(s); });
}

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
namespace {
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
}  // namespace

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

int gsv_session_create_plan(gsv_engine* e, const gsv_plan* plan, size_t n_instances, gsv_session** out) { return gsv_session_create_plan_ex(e, plan, n_instances, 1, out); }
int gsv_session_create_plan_ex(gsv_engine* e, const gsv_plan* plan, size_t n_instances, int retain_stream, gsv_session** out) {
  gsv_plan_session_opts o{};
  o.retain_stream = retain_stream;
  return gsv_session_create_plan_opts(e, plan, n_instances, &o, out);
}
// The call-level schedule of a plan session (schedule.hpp) for `n_wg` workgroups per call on a device with `n_cus` CUs.
static int install_schedule(gsv_session* s, const gsv_plan_session_opts& o, int n_cus, size_t free_b);
static Schedule make_schedule(const gsv_plan* plan, uint32_t ni, size_t n_instances, int n_cus, size_t free_bytes, const gsv_plan_session_opts& o, uint64_t* max_call_ct) {
  std::vector<SchedCall> calls(plan->calls.size());
  uint64_t max_block = 0;
  uint32_t max_slots = 0;
  for (size_t k = 0; k < plan->calls.size(); ++k) {
    const PlanCall& c = plan->calls[k];
    const Program& g = c.prog->variant(ni);
    calls[k].in = c.in_globals.data(); calls[k].n_in = c.in_globals.size();
    calls[k].out = c.out_globals.data(); calls[k].n_out = c.out_globals.size();
    calls[k].n_slots = g.n_slots; calls[k].n_ct = g.n_ct; calls[k].n_steps = g.n_steps;
    max_block = std::max<uint64_t>(max_block, g.n_ct);
    max_slots = std::max(max_slots, g.n_slots);
  }
  *max_call_ct = max_block;
  SchedParams sp;
  const size_t n_wg = (n_instances + ni - 1) / ni;
  // calls side by side: as many as it takes to give every CU a workgroup (GSV_PLAN_CONCURRENCY / opts override)
  uint32_t conc = o.max_concurrent_calls ? o.max_concurrent_calls : uint32_t(std::max<size_t>(1, size_t(n_cus) / std::max<size_t>(1, n_wg)));
  if (!o.max_concurrent_calls) if (const char* ev = getenv("GSV_PLAN_CONCURRENCY")) conc = uint32_t(std::max(1, atoi(ev)));
  // a session that drains its stream leaves a few CUs to the gather kernels that bring finished segments into gate order beside the
  // running window (a workgroup of the garbling kernel takes a whole CU, also while it waits for a dependency)
  if (!o.max_concurrent_calls && o.retain_stream != 1 && conc > 1 && n_wg * size_t(conc) + 16 > size_t(n_cus)) conc = uint32_t(std::max<size_t>(1, (size_t(n_cus) - std::min<size_t>(16, size_t(n_cus) / 2)) / n_wg));
  sp.max_calls_in_flight = std::min<uint32_t>(conc, 65535u);
  // the scratch ring: at most ~1/16 of the free device memory over all instances, and 2^30 slots (slot offsets are 32 bits)
  uint64_t slots = o.max_scratch_slots ? o.max_scratch_slots : uint64_t(free_bytes / 16 / 16 / std::max<size_t>(1, n_instances));
  sp.max_scratch_slots = std::min<uint64_t>(std::max<uint64_t>(slots, max_slots), 1ull << 30);
  if (conc == 1) sp.max_scratch_slots = max_slots;
  // ciphertext window: the whole stream when it is retained, else about a quarter of the free memory for the two window buffers
  if (o.retain_stream == 1) sp.max_window_ct = ~0ull;
  else {
    // Default for sessions that do not retain the stream: the device block (= one window, the scope inside which independent call chains
    // overlap: schedule.hpp) takes up to 40 % of the free memory, at most 48 GB over all instances (one instance of the verifier, 47.7 GB
    // of ciphertexts, is ONE window: 26.7 s instead of the 27.6 s of two — profiles/r04_e2e/verifier_mixed_units.log).  The stream leaves the
    // device in SEGMENTS of a window (below), so a large window costs the drain nothing.  Round 3's default cut one instance's pass into
    // 2 windows and drained whole windows (48.2 s with the commitment: half of the 27-s CBC-MAC chain uncovered); 46 windows of 1 GB hid
    // the chain but cost the garbling 4.7 s — the verifier's line-coefficient chain precedes the Miller loop in stream order and only
    // runs beside it inside one window (29.6 s with 2 windows, 33.4 s with 18, 34.3 s with 46: profiles/r04_e2e/one_instance_windows.log).
    // (... and at most 48 GB over all instances: device memory that has been freed is scrubbed before it is handed out again, ~25 GB/s,
    // so a session of 16 instances with a 96-GB block took 6 s to create; its garbling is 3 % faster with 6-GB windows than with 2-GB ones)
    const double block_bytes = std::min(double(free_bytes) * 0.4, 48e9);
    uint64_t w = o.window_ct_records ? o.window_ct_records : uint64_t(block_bytes / 16.0 / double(std::max<size_t>(1, n_instances)));
    if (!o.window_ct_records && conc == 1) w = 0;  // sequential sessions keep the one-call block of rounds 1-2 (smallest footprint)
    sp.max_window_ct = std::max<uint64_t>(w, max_block);
  }
  // Drain segments: at most 64 M records (1 GB) per instance — the serial CBC-MAC chain of a segment takes 0.6 s —, less when three
  // gate-order buffers of that size would take more than a tenth of the free memory; never smaller than the largest call.
  {
    uint64_t sg = o.drain_segment_records ? o.drain_segment_records : std::min<uint64_t>(uint64_t(double(free_bytes) * 0.1 / (3.0 * 16.0) / double(std::max<size_t>(1, n_instances))), 1ull << 26);
    if (const char* ev = getenv("GSV_DRAIN_SEGMENT_RECORDS")) if (!o.drain_segment_records) sg = uint64_t(std::max(1ll, atoll(ev)));
    sp.segment_ct = std::min<uint64_t>(std::max<uint64_t>(sg, max_block), sp.max_window_ct);
  }
  // Ciphertext ring (schedule.hpp), retain_stream = GSV_STREAM_RING or GSV_CT_RING=1 in the environment: a session that does not
  // retain the stream and runs calls side by side keeps THREE
  // segments' worth of ciphertexts on the device instead of a window's, and the window becomes the whole pass (one instance: 48 GB of
  // device block -> 3.2 GB, 2 windows -> 1; sixteen: 17 windows -> 1 over a 27-GB ring).  Opt-in: with large windows + segments the
  // pass is already bounded by the dependent depth and the host's MAC chain (tools/ring_ab.py: 30.7 s either way for one instance,
  // 32.9 s vs 32.5-33.4 s for sixteen), and a ring makes the running launch WAIT for the host — it must never share a hardware queue
  // with the side streams (create_side_stream).  Not with an explicit window_ct_records (the caller sizes the launches: garble ||
  // evaluate pairs, tests), not for sequential sessions, and not when the whole stream fits the ring anyway.
  const bool ring_wanted = o.retain_stream == GSV_STREAM_RING || (o.retain_stream == 0 && getenv("GSV_CT_RING") && atoi(getenv("GSV_CT_RING")) == 1);
  if (ring_wanted && !o.window_ct_records && conc > 1) {
    uint64_t ring = std::max<uint64_t>(3 * sp.segment_ct, 2 * sp.segment_ct + max_block);
    if (const char* ev = getenv("GSV_CT_RING_RECORDS")) ring = std::max<uint64_t>(uint64_t(std::max(1ll, atoll(ev))), 2 * sp.segment_ct + max_block);  // tests: small rings on small circuits
    if (ring < plan->n_ct && ring <= sp.max_window_ct) { sp.ring_ct = ring; sp.max_window_ct = ~0ull; }
  }
  sp.max_window_calls = std::min<uint32_t>(o.max_window_calls ? o.max_window_calls : 32768u, 65535u);
  Schedule sc = schedule_calls(calls, plan->n_globals, plan->outputs, sp);
  {  // always: the O(calls) ring checks (a violation would otherwise show up as a 60 s device stall and status 2)
    const std::string err = verify_ring_bounds(calls, sc);
    if (!err.empty()) gsv_panic("plan schedule: " + err);
  }
  if (getenv("GSV_PLAN_DEBUG") || getenv("GSV_VERIFY_SCHEDULE")) {
    const std::string err = verify_schedule(calls, plan->n_globals, plan->outputs, sc);
    if (!err.empty()) gsv_panic("internal: plan schedule violates a hazard: " + err);
    std::fprintf(stderr, "plan schedule: %zu calls, %zu windows, <= %u calls in flight (width %u), scratch ring %llu slots, depth %llu of %llu steps, %zu dependencies\n", calls.size(),
                 sc.windows.size(), sp.max_calls_in_flight, sc.max_width, (unsigned long long)sc.scratch_slots, (unsigned long long)sc.critical_steps, (unsigned long long)sc.total_steps, sc.deps.size());
  }
  return sc;
}
int gsv_session_create_plan_opts(gsv_engine* e, const gsv_plan* plan, size_t n_instances, const gsv_plan_session_opts* opts, gsv_session** out) {
  if (!e || !plan || !out || n_instances == 0 || !plan->finished || plan->calls.empty()) return fail(GSV_ERR_INVALID, "bad argument / plan not finished");
  gsv_plan_session_opts o{};
  o.retain_stream = 1;
  if (opts) o = *opts;
  HIPCHK(hipSetDevice(e->device));
  if (plan->device >= 0 && plan->device != e->device) return fail(GSV_ERR_INVALID, "this plan was loaded into device " + std::to_string(plan->device) + " (gsv_plan_load with an engine): it serves sessions on that device only");
  SessionPtr s(new gsv_session());
  s->e = e; s->p = plan->calls[0].prog; s->plan = plan; s->n_inst = n_instances; s->replays = 1; s->ct_cap = 1;
  s->ct_uploaded.assign(n_instances, 0);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, e->device));
  {
    uint32_t servable = 4;
    for (const auto& c : plan->calls) if (!c.prog->src) servable = std::min(servable, c.prog->window_div);
    s->ni = choose_instances_per_wg(n_instances, prop.multiProcessorCount, servable);
  }
  s->call_dev.resize(plan->calls.size());
  if (s->ni > 1) {  // the plan's programs are independent: compile their missing window variants in parallel
    GSV_TRY
    std::vector<gsv_program*> todo;
    for (const auto& c : plan->calls) if (std::find(todo.begin(), todo.end(), c.prog) == todo.end()) todo.push_back(c.prog);
    const uint32_t ni = s->ni;
    parallel_for_programs(todo.size(), [&](size_t i) { std::lock_guard<std::mutex> lk(todo[i]->mu); compile_window_variant(todo[i], ni); });
    GSV_CATCH
  }
  for (size_t k = 0; k < plan->calls.size(); ++k) {
    int rc = upload_program(e, plan->calls[k].prog, s->ni, &s->call_dev[k].dp);
    if (rc) return rc;
  }
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  s->opts = o;
  {
    int rc = install_schedule(s.get(), o, prop.multiProcessorCount, free_b);
    if (rc) return rc;
  }
  const Program& f = s->facade;
  s->w_slots_cap = f.n_slots;
  s->ct_records_cap = s->ct_stride();
  DEVALLOC(&s->W, n_instances * size_t(f.n_slots) * 16, "the wire files");
  HIPCHK(hipMalloc(&s->VB, n_instances * size_t(f.n_slots)));
  HIPCHK(hipMemset(s->VB, 0, n_instances * size_t(f.n_slots)));
  const size_t ct_bytes = n_instances * size_t(s->ct_stride()) * 16;
  DEVALLOC(&s->CT, ct_bytes, "the ciphertext blocks");
  HIPCHK(hipMalloc(&s->delta, n_instances * 16));
  HIPCHK(hipMalloc(&s->out, n_instances * f.output_slots.size() * 16 + 16));
  HIPCHK(hipMalloc(&s->out_bits, n_instances * f.output_slots.size() + 16));
  HIPCHK(hipMalloc(&s->in_bits, n_instances * f.input_slots.size() + 16));
  HIPCHK(hipEventCreate(&s->ev0));
  HIPCHK(hipEventCreate(&s->ev1));
  *out = s.release();
  return GSV_OK;
}
// Everything of a plan session that depends on its SCHEDULE: the schedule itself, the wire-file layout (scratch regions in front of the
// global wires), the ring's position counter, the completion counters, and the device tables of the window launches (call descriptors,
// hand-over lists, dependency lists, completion flags).  Called by gsv_session_create_plan_opts and again, with the safe options, by
// fall_back_to_safe_schedule (after drop_schedule).
static int install_schedule(gsv_session* s, const gsv_plan_session_opts& o, int n_cus, size_t free_b) {
  const gsv_plan* plan = s->plan;
  const size_t n_instances = s->n_inst;
  uint64_t max_call_ct = 0;
  GSV_TRY
  s->sched = make_schedule(plan, s->ni, n_instances, n_cus, free_b, o, &max_call_ct);
  GSV_CATCH
  const Schedule& sc = s->sched;
  const uint32_t scratch = uint32_t((std::max<uint64_t>(sc.scratch_slots, SLOT_FIRST_INPUT) + 7) / 8 * 8);
  s->global_base = scratch;
  s->plan_retain = o.retain_stream == 1;
  s->ct_ring = sc.ring_ct != 0;
  s->plan_max_block = s->ct_ring ? sc.ring_ct : sc.max_window_ct;
  s->plan_max_segment = sc.max_segment_ct;
  if (s->ct_ring) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->host_ct_pos), 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->dev_ct_pos), s->host_ct_pos, 0));
    *s->host_ct_pos = 0;
  }
  if (uint64_t(scratch) + plan->n_globals > 0xFFFFFFF0ull) return fail(GSV_ERR_CIRCUIT, "plan wire file too large");
  Program& f = s->facade;
  f = Program();
  f.n_slots = scratch + plan->n_globals;
  f.n_gates = plan->n_gates; f.n_ct = plan->n_ct;
  for (uint32_t i = 0; i < plan->n_inputs; ++i) f.input_slots.push_back(scratch + i);
  auto global_slot = [&](uint32_t w) -> uint32_t { return w == PLAN_WIRE_FALSE ? SLOT_FALSE : w == PLAN_WIRE_TRUE ? SLOT_TRUE : scratch + w; };
  for (uint32_t w : plan->outputs) f.output_slots.push_back(global_slot(w));
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    HIPCHK(hipMalloc(dst, bytes + 64));
    if (bytes) HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return GSV_OK;
  };
  // descriptors, wire hand-over lists and dependency lists, all in stream order
  {
    const size_t n = plan->calls.size();
    std::vector<dev::CallDesc> cds(n);
    std::vector<uint32_t> csrc, cdst, deps;
    {
      // the device-written completion counters (one per call of the PLAN: windows enqueued back to back never share a counter), in mapped host memory
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->host_done), n * 4 + 64, hipHostMallocMapped | hipHostMallocCoherent));
      std::memset(s->host_done, 0, n * 4 + 64);
      HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->dev_done), s->host_done, 0));
    }
    for (const Schedule::Window& w : sc.windows) {
      for (uint32_t k = w.call0; k < w.call1; ++k) {
        const PlanCall& c = plan->calls[k];
        const Program& g = s->call_prog(k);
        const uint32_t base = sc.scratch_base[k];
        dev::CallDesc& d = cds[k];
        std::memset(&d, 0, sizeof d);
        d.steps = s->call_dev[k].dp.steps; d.ands = s->call_dev[k].dp.ands; d.xors = s->call_dev[k].dp.xors;
        d.gid_off = c.gid_off; d.ct_off = s->plan_retain ? c.ct_off : s->ct_ring ? sc.ring_off[k] : c.ct_off - w.ct0;
        if (s->ct_ring) { d.ct_need = sc.ring_need[k]; d.ct_ready = sc.seg_end[k]; d.ct_pos = s->dev_ct_pos; }
        d.done_host = s->dev_done + k;
        d.w_base = base; d.n_steps = g.n_steps; d.and_terms = g.and_terms;
        d.pre_off = uint32_t(csrc.size());
        if (base != 0)  // the call's own copies of the constant labels (FALSE, TRUE, the all-zero label) in front of its scratch region
          for (uint32_t q = 0; q < SLOT_FIRST_INPUT; ++q) { csrc.push_back(q); cdst.push_back(base + q); }
        for (size_t i = 0; i < c.in_globals.size(); ++i) { csrc.push_back(global_slot(c.in_globals[i])); cdst.push_back(base + g.input_slots[i]); }
        d.n_pre = uint32_t(csrc.size()) - d.pre_off;
        d.post_off = uint32_t(csrc.size());
        for (size_t i = 0; i < c.out_globals.size(); ++i) {
          if (g.output_slots[i] & SLOT_LDS_FLAG) return fail(GSV_ERR_CIRCUIT, "internal: a program output lives in the LDS window");
          csrc.push_back(base + g.output_slots[i]); cdst.push_back(scratch + c.out_globals[i]);
        }
        d.n_post = uint32_t(csrc.size()) - d.post_off;
        d.dep_off = uint32_t(deps.size());
        for (uint32_t q = sc.dep_off[k]; q < sc.dep_off[k + 1]; ++q) deps.push_back(sc.deps[q] - w.call0);
        d.n_deps = uint32_t(deps.size()) - d.dep_off;
        if (csrc.size() > 0xFFFFFF00ull) return fail(GSV_ERR_CIRCUIT, "plan hand-over lists too large");
      }
      s->flag_stride = std::max<uint32_t>(s->flag_stride, w.call1 - w.call0 + 2);  // + a slot nobody writes (fault injection below) + the group's progress counter (kernels.hip, watchdog)
    }
    // GSV_FAULT_WITHHOLD_DEP=1 (tests): the first dependency of the first call that has one is pointed at the slot nobody writes — on
    // the device exactly what a violated dispatch-order assumption looks like (a dependency that never completes).  Never for the safe schedule.
    if (!s->safe_mode && getenv("GSV_FAULT_WITHHOLD_DEP") && atoi(getenv("GSV_FAULT_WITHHOLD_DEP")) == 1)
      for (size_t k = 0; k < n; ++k) if (cds[k].n_deps) { deps[cds[k].dep_off] = s->flag_stride - 2; break; }
    const size_t n_wg = (n_instances + s->ni - 1) / s->ni;
    int rc;
    if ((rc = up(&s->d_calls, cds.data(), cds.size() * sizeof(dev::CallDesc))) || (rc = up(&s->d_copy_src, csrc.data(), csrc.size() * 4)) || (rc = up(&s->d_copy_dst, cdst.data(), cdst.size() * 4)) ||
        (rc = up(&s->d_deps, deps.data(), deps.size() * 4)))
      return rc;
    HIPCHK(hipMalloc(&s->d_flags, n_wg * size_t(s->flag_stride) * 4 + 64));
    HIPCHK(hipMemset(s->d_flags, 0, n_wg * size_t(s->flag_stride) * 4 + 64));
    HIPCHK(hipMalloc(&s->d_error, 64));
    HIPCHK(hipMemset(s->d_error, 0, 64));
    if ((rc = up(&s->plan_out_slots, f.output_slots.data(), f.output_slots.size() * 4))) return rc;
  }
  return GSV_OK;
}
// the schedule-dependent state of a session, released (the caller has synchronised the device's streams)
static void drop_schedule(gsv_session* s) {
  for (void** q : {&s->d_calls, &s->d_copy_src, &s->d_copy_dst, &s->d_deps, &s->d_flags, &s->d_error, &s->plan_out_slots}) { if (*q) (void)hipFree(*q); *q = nullptr; }
  if (s->host_done) (void)hipHostFree(s->host_done);
  if (s->host_ct_pos) (void)hipHostFree(s->host_ct_pos);
  s->host_done = nullptr; s->dev_done = nullptr; s->host_ct_pos = nullptr; s->dev_ct_pos = nullptr;
  s->flag_stride = 0; s->next_call = 0;
}
int gsv_session_plan_schedule_info(const gsv_session* s, gsv_plan_schedule_info* info) {
  if (!s || !s->plan || !info) return fail(GSV_ERR_INVALID, "null argument / not a plan session");
  const Schedule& sc = s->sched;
  info->n_calls = s->plan->calls.size(); info->n_windows = sc.windows.size(); info->n_dependencies = sc.deps.size();
  info->max_width = sc.max_width;
  info->scratch_slots = s->global_base; info->wire_file_slots = s->facade.n_slots; info->window_ct_records = sc.max_window_ct;
  info->critical_steps = sc.critical_steps; info->total_steps = sc.total_steps;
  info->n_segments = sc.segments.size(); info->segment_ct_records = sc.max_segment_ct;
  info->ct_ring_records = sc.ring_ct;
  return GSV_OK;
}
int gsv_session_plan_window(const gsv_session* s, uint64_t window, uint64_t* first_call, uint64_t* n_calls, uint64_t* max_width) {
  if (!s || !s->plan || window >= s->sched.windows.size()) return fail(GSV_ERR_INVALID, "null argument / window index out of range");
  const Schedule::Window& w = s->sched.windows[size_t(window)];
  if (first_call) *first_call = w.call0;
  if (n_calls) *n_calls = w.call1 - w.call0;
  if (max_width) *max_width = w.max_width;
  return GSV_OK;
}
int gsv_session_set_drain_instances(gsv_session* s, size_t n) {
  if (!s || n > s->n_inst) return fail(GSV_ERR_INVALID, "null session / more instances than the session holds");
  s->drain_instances = n;  // (the gate-order buffers are re-allocated by the next streaming call if they were sized for fewer: ensure_ct_gate)
  return GSV_OK;
}
int gsv_session_set_unchecked_slices(gsv_session* s, int on) {
  if (!s) return fail(GSV_ERR_INVALID, "null session");
  s->unchecked_slices = on != 0;
  return GSV_OK;
}

static int stage_labels(gsv_session* s, const uint8_t* consts, const uint8_t* inputs) {
  // Per instance the wire file starts [FALSE, TRUE, ZERO, input0, input1, ...]: one strided copy.
  const Program& g = s->prog();
  const size_t n_in = g.input_slots.size();
  if (s->plan) {  // constants at slots 0..2, inputs at the head of the global region: two strided copies
    std::vector<uint8_t> host(s->n_inst * 48, 0);
    for (size_t i = 0; i < s->n_inst; ++i) std::memcpy(&host[i * 48], consts + 32 * i, 32);
    HIPCHK(hipMemcpy2D(s->W, size_t(g.n_slots) * 16, host.data(), 48, 48, s->n_inst, hipMemcpyHostToDevice));
    if (n_in) HIPCHK(hipMemcpy2D(static_cast<uint8_t*>(s->W) + size_t(s->global_base) * 16, size_t(g.n_slots) * 16, inputs, n_in * 16, n_in * 16, s->n_inst, hipMemcpyHostToDevice));
    return GSV_OK;
  }
  const size_t row = (SLOT_FIRST_INPUT + n_in) * 16;
  std::vector<uint8_t> host(s->n_inst * row, 0);
  for (size_t i = 0; i < s->n_inst; ++i) {
    std::memcpy(&host[i * row], consts + 32 * i, 32);
    if (n_in) std::memcpy(&host[i * row + SLOT_FIRST_INPUT * 16], inputs + i * n_in * 16, n_in * 16);
  }
  HIPCHK(hipMemcpy2D(s->W, size_t(g.n_slots) * 16, host.data(), row, row, s->n_inst, hipMemcpyHostToDevice));
  return GSV_OK;
}

// Plan sessions keep the host's last inputs: a pass that is repeated on the safe schedule (fall_back_to_safe_schedule) starts from them —
// the wire file's input region is recycled by the plan's later calls, and the safe schedule lays the wire file out differently.
static void stash_inputs(gsv_session* s, int kind, const uint8_t* delta, const uint8_t* consts, const uint8_t* inputs, const uint8_t* bits) {
  if (!s->plan) return;
  const size_t n_in = s->prog().input_slots.size();
  s->stash_kind = kind;
  if (delta) s->stash_delta.assign(delta, delta + s->n_inst * 16); else s->stash_delta.clear();
  s->stash_consts.assign(consts, consts + s->n_inst * 32);
  if (n_in) s->stash_inputs.assign(inputs, inputs + s->n_inst * n_in * 16); else s->stash_inputs.clear();
  if (bits && n_in) s->stash_bits.assign(bits, bits + s->n_inst * n_in); else s->stash_bits.clear();
}
static int set_garble_inputs_impl(gsv_session* s, const uint8_t* delta, const uint8_t* const_label0, const uint8_t* input_label0) {
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipMemcpy(s->delta, delta, s->n_inst * 16, hipMemcpyHostToDevice));
  return stage_labels(s, const_label0, input_label0);
}
int gsv_session_set_garble_inputs(gsv_session* s, const uint8_t* delta, const uint8_t* const_label0, const uint8_t* input_label0) {
  if (!s || !delta || !const_label0 || (!input_label0 && !s->prog().input_slots.empty())) return fail(GSV_ERR_INVALID, "null argument");
  stash_inputs(s, 1, delta, const_label0, input_label0, nullptr);
  return set_garble_inputs_impl(s, delta, const_label0, input_label0);
}
static int set_evaluate_inputs_impl(gsv_session* s, const uint8_t* const_active, const uint8_t* input_active, const uint8_t* input_bits);
int gsv_session_set_evaluate_inputs(gsv_session* s, const uint8_t* const_active, const uint8_t* input_active, const uint8_t* input_bits) {
  if (!s || !const_active || ((!input_active || !input_bits) && !s->prog().input_slots.empty())) return fail(GSV_ERR_INVALID, "null argument");
  stash_inputs(s, 2, nullptr, const_active, input_active, input_bits);
  return set_evaluate_inputs_impl(s, const_active, input_active, input_bits);
}
static int set_evaluate_inputs_impl(gsv_session* s, const uint8_t* const_active, const uint8_t* input_active, const uint8_t* input_bits) {
  HIPCHK(hipSetDevice(s->e->device));
  int rc = stage_labels(s, const_active, input_active);
  if (rc) return rc;
  const Program& g = s->prog();
  const size_t n_in = g.input_slots.size();
  // plaintext bits: constants FALSE=0 / TRUE=1 (evaluate_mode.rs:104-121), then the input bits
  HIPCHK(hipMemset(s->VB, 0, s->n_inst * size_t(g.n_slots)));
  std::vector<uint8_t> two(s->n_inst * 2);
  for (size_t i = 0; i < s->n_inst; ++i) { two[2 * i] = 0; two[2 * i + 1] = 1; }
  HIPCHK(hipMemcpy2D(s->VB, g.n_slots, two.data(), 2, 2, s->n_inst, hipMemcpyHostToDevice));
  if (n_in) {
    std::vector<uint8_t> nb(s->n_inst * n_in);
    for (size_t i = 0; i < nb.size(); ++i) nb[i] = input_bits[i] ? 1 : 0;
    HIPCHK(hipMemcpy(s->in_bits, nb.data(), nb.size(), hipMemcpyHostToDevice));
    if (gsvk_scatter_bits(s->VB, g.n_slots, s->first_input_slot(), s->in_bits, uint32_t(n_in), uint32_t(s->n_inst), nullptr) != 0) return fail(GSV_ERR_DEVICE, "scatter_bits launch failed");
    HIPCHK(hipDeviceSynchronize());
  }
  return GSV_OK;
}
// The device stream of an instance holds each replay's ciphertexts in PROGRAM order (coalesced stores, program.hpp);
// every host-facing call speaks GATE order (the reference's stream / gc_{i}.bin order) through a staging buffer
// and a gather / scatter kernel.
static const uint64_t CT_STAGE_RECORDS = 1ull << 20;  // 16 MiB
static int ensure_ct_stage(gsv_session* s) {
  if (!s->ct_stage) HIPCHK(hipMalloc(&s->ct_stage, CT_STAGE_RECORDS * 16));
  return GSV_OK;
}
// stage[0..n) <-> gate-order records [first, first+n) of one instance's stream.  Program sessions: one permutation per replay
// block; plan sessions: one per call block.
static int permute_range(gsv_session* s, size_t instance, uint64_t first, uint64_t n, int scatter) {
  uint8_t* stream = static_cast<uint8_t*>(s->CT) + instance * s->ct_stride() * 16;
  if (!s->plan) return gsvk_permute_ciphertexts(stream, s->dp.ct_pos, s->prog().n_ct, first, n, s->ct_stage, scatter, s->e->stream);
  for (size_t k = 0; k < s->plan->calls.size(); ++k) {
    const uint64_t b0 = s->plan->calls[k].ct_off, b1 = b0 + s->call_prog(k).n_ct;
    const uint64_t lo = std::max(first, b0), hi = std::min(first + n, b1);
    if (lo >= hi) continue;
    int rc = gsvk_permute_ciphertexts(stream + b0 * 16, s->call_dev[k].dp.ct_pos, b1 - b0, lo - b0, hi - lo, static_cast<uint8_t*>(s->ct_stage) + (lo - first) * 16, scatter, s->e->stream);
    if (rc) return rc;
  }
  return 0;
}
// copies stream records [first, first+n) of one instance, in gate order, to host memory
static int fetch_ciphertexts(gsv_session* s, size_t instance, uint64_t first, uint64_t n, uint8_t* out) {
  int rc = ensure_ct_stage(s);
  if (rc) return rc;
  if (permute_range(s, instance, first, n, 0) != 0) return fail(GSV_ERR_DEVICE, "ciphertext gather launch failed");
  HIPCHK(hipMemcpyAsync(out, s->ct_stage, n * 16, hipMemcpyDeviceToHost, s->e->stream));
  HIPCHK(hipStreamSynchronize(s->e->stream));
  return GSV_OK;
}

int gsv_session_upload_ciphertexts(gsv_session* s, size_t instance, const uint8_t* cts, uint64_t n_records) {
  if (!s || instance >= s->n_inst || (!cts && n_records)) return fail(GSV_ERR_INVALID, "bad argument");
  if (n_records > s->ct_stride()) return fail(GSV_ERR_INVALID, "more ciphertexts than the session's stream capacity");
  HIPCHK(hipSetDevice(s->e->device));
  // gate-order records from the host -> program-order positions of the device stream (staged in chunks)
  int rc = ensure_ct_stage(s);
  if (rc) return rc;
  for (uint64_t off = 0; off < n_records; off += CT_STAGE_RECORDS) {
    const uint64_t n = std::min<uint64_t>(CT_STAGE_RECORDS, n_records - off);
    HIPCHK(hipMemcpyAsync(s->ct_stage, cts + off * 16, n * 16, hipMemcpyHostToDevice, s->e->stream));
    if (permute_range(s, instance, off, n, 1) != 0) return fail(GSV_ERR_DEVICE, "ciphertext scatter launch failed");
    HIPCHK(hipStreamSynchronize(s->e->stream));
  }
  s->ct_uploaded[instance] = n_records;
  return GSV_OK;
}

static int launch_plan(gsv_session* s, uint64_t gate_id_base, bool eval);
static int launch(gsv_session* s, uint64_t gate_id_base, bool eval, uint64_t rep_base = 0, uint64_t n_replays = 0) {
  if (s->plan) return launch_plan(s, gate_id_base, eval);
  const Program& g = s->prog();
  HIPCHK(hipSetDevice(s->e->device));
  dev::KernelArgs ka{};
  ka.steps = s->dp.steps; ka.ands = s->dp.ands; ka.xors = s->dp.xors;
  ka.W = static_cast<uint4*>(s->W); ka.VB = static_cast<uint8_t*>(s->VB); ka.CT = static_cast<uint4*>(s->CT);
  ka.delta = static_cast<const uint4*>(s->delta); ka.te = static_cast<const uint32_t*>(s->e->te);
  ka.fb_src = static_cast<const uint32_t*>(s->dp.fb_src); ka.fb_dst = static_cast<const uint32_t*>(s->dp.fb_dst);
  ka.ct_stride = s->ct_stride(); ka.gid_base = gate_id_base; ka.n_gates = g.n_gates; ka.n_ct = g.n_ct;
  ka.n_steps = g.n_steps; ka.n_slots = g.n_slots; ka.replays = uint32_t(n_replays ? n_replays : s->replays); ka.rep_base = uint32_t(rep_base); ka.ct_cap_replays = uint32_t(s->ct_cap);
  ka.n_fb = uint32_t(g.fb_src_slot.size()); ka.fb_stage_base = g.fb_stage_base;
  ka.n_instances = uint32_t(s->n_inst);
  ka.hasher = uint32_t(s->hasher);
  ka.and_terms = g.and_terms; ka.any_four_wire = g.and_terms == 4;
  ka.step_clock = static_cast<unsigned long long*>(s->step_clock);
  ka.instances_per_wg = s->ni;
  if (const char* dg = getenv("GSV_DIAG")) ka.diag = uint32_t(atoi(dg));  // timing experiments (libgsv_engine_diag.so only): outputs are wrong when set
  HIPCHK(hipEventRecord(s->ev0, s->e->stream));
  if (ka.n_steps) {
    int lrc = gsvk_launch_program(&ka, uint32_t(s->n_inst), eval ? 1 : 0, s->e->stream);
    if (lrc != 0) return fail(GSV_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(hipError_t(lrc)));
  }
  HIPCHK(hipEventRecord(s->ev1, s->e->stream));
  if (!g.output_slots.empty()) {
    if (gsvk_gather_outputs(s->W, s->VB, g.n_slots, static_cast<const uint32_t*>(s->dp.out_slots), uint32_t(g.output_slots.size()),
                            uint32_t(s->n_inst), s->out, eval ? s->out_bits : nullptr, s->e->stream) != 0)
      return fail(GSV_ERR_DEVICE, "gather launch failed");
  }
  s->ran = true; s->last_eval = eval;
  return GSV_OK;
}
// One WINDOW of a plan session = one launch: grid = (instance groups, calls of the window); every workgroup waits for the
// completion flags of the calls it depends on, fetches its inputs from the global wires, runs its program in its own scratch region
// and publishes its outputs (kernels.hip).  A sequential schedule (one call in flight) is the same launch with each call depending on
// its predecessor: the instance groups still drift apart instead of meeting at a launch boundary after every call.
static int launch_plan_window(gsv_session* s, size_t w, uint64_t gate_id_base, bool eval, void* ct_block = nullptr, hipStream_t stream = nullptr) {
  const Program& f = s->facade;
  if (!ct_block) ct_block = s->CT;       // (garble -> evaluate: the garbler's current block, for both sessions)
  if (!stream) stream = s->e->stream;
  const Schedule::Window& win = s->sched.windows[w];
  dev::KernelArgs ka{};
  ka.calls = static_cast<const dev::CallDesc*>(s->d_calls) + win.call0;
  ka.copy_src = static_cast<const uint32_t*>(s->d_copy_src); ka.copy_dst = static_cast<const uint32_t*>(s->d_copy_dst);
  ka.deps = static_cast<const uint32_t*>(s->d_deps); ka.flags = static_cast<uint32_t*>(s->d_flags); ka.error = static_cast<uint32_t*>(s->d_error);
  if (s->host_done) {  // (the counters of THIS window's calls: its launch of the previous pass has long finished — every pass ends synchronised)
    std::memset(s->host_done + win.call0, 0, size_t(win.call1 - win.call0) * 4);
    __atomic_thread_fence(__ATOMIC_RELEASE);
  }
  ka.flag_stride = s->flag_stride; ka.epoch = ++s->epoch;
  {
    // dependency watchdog (kernels.hip): seconds without ANY completed call of the instance group before a wait gives up
    double secs = 60.0;
    if (const char* ev = getenv("GSV_DEP_WAIT_SECONDS")) { char* end = nullptr; const double v = std::strtod(ev, &end); if (end != ev && v > 0) secs = v; }
    ka.wait_ticks = (unsigned long long)(std::min(secs, 86400.0) * 1e8);
  }
  ka.W = static_cast<uint4*>(s->W); ka.VB = static_cast<uint8_t*>(s->VB); ka.CT = static_cast<uint4*>(ct_block);
  ka.delta = static_cast<const uint4*>(s->delta); ka.te = static_cast<const uint32_t*>(s->e->te);
  ka.ct_stride = s->ct_stride(); ka.gid_base = gate_id_base; ka.n_gates = 0; ka.n_ct = 0;
  ka.n_steps = 0; ka.n_slots = f.n_slots; ka.replays = 1; ka.rep_base = 0; ka.ct_cap_replays = 1;
  ka.n_instances = uint32_t(s->n_inst); ka.hasher = uint32_t(s->hasher); ka.instances_per_wg = s->ni;
  for (uint32_t k = win.call0; k < win.call1 && !ka.any_four_wire; ++k) ka.any_four_wire = s->call_prog(k).and_terms == 4;
  int lrc = gsvk_launch_batch(&ka, uint32_t(s->n_inst), win.call1 - win.call0, eval ? 1 : 0, stream);
  if (lrc != 0) return fail(GSV_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(hipError_t(lrc)));
  return GSV_OK;
}
// after a synchronisation: did a dependency wait give up?
static int check_plan_error(gsv_session* s) {
  uint32_t ew[16] = {0};
  HIPCHK(hipMemcpy(ew, s->d_error, 64, hipMemcpyDeviceToHost));
  const uint32_t err = ew[0];
  if (err == 2) {
    // which calls of the last window have not finished everywhere, and where the host's position stood (diagnostics)
    std::string open_calls;
    if (s->host_done && !s->sched.windows.empty()) {
      const uint32_t n_wg = uint32_t((s->n_inst + s->ni - 1) / s->ni);
      const Schedule::Window& win = s->sched.windows.back();
      int shown = 0;
      for (uint32_t k = win.call0; k < win.call1 && shown < 12; ++k)
        if (s->host_done[k] != n_wg) { open_calls += " " + std::to_string(k) + "(" + std::to_string(s->host_done[k]) + "/" + std::to_string(n_wg) + ", need " + std::to_string(s->sched.ring_need[k]) + ")"; ++shown; }
    }
    return fail(GSV_ERR_DEVICE, "a call waited for the host's stream position (ciphertext ring) and saw it stand still at " + std::to_string(s->host_ct_pos ? *s->host_ct_pos : 0) +
                                    "; unfinished calls:" + open_calls + "; the call that gave up: " + std::to_string(ew[8]) + " of the window (instance group " + std::to_string(ew[9]) + "), it wanted position " +
                                    std::to_string((uint64_t(ew[11]) << 32) | ew[10]) + ", saw " + std::to_string((uint64_t(ew[13]) << 32) | ew[12]) + " unchanged for " +
                                    std::to_string(double((uint64_t(ew[15]) << 32) | ew[14]) * 1e-8) + " s" + (s->ring_diag.empty() ? "" : "; host: " + s->ring_diag) + "; results are invalid");
  }
  s->dep_fault = err == 1;
  if (err) return fail(GSV_ERR_DEVICE, "a call of the plan waited for a dependency that never completed (dispatch-order assumption of schedule.hpp violated); results are invalid");
  return GSV_OK;
}
// Gate order <-> program order for calls [k0, k1) of window w (a drain segment, or the whole window): gate-order buffer, records
// relative to `gate_ct0` (the stream index of the buffer's first record) <-> the window's device block.
static int permute_plan_calls(gsv_session* s, size_t w, uint32_t k0, uint32_t k1, uint64_t gate_ct0, uint64_t gate_stride, int scatter, void* ct_block, void* gate_buf, hipStream_t stream) {
  const Schedule::Window& win = s->sched.windows[w];
  if (!ct_block) ct_block = s->CT;
  if (!gate_buf) gate_buf = s->ct_gate;
  if (!stream) stream = s->e->stream;
  for (uint32_t k = k0; k < k1; ++k) {
    const Program& cp = s->call_prog(k);
    if (!cp.n_ct) continue;
    const uint64_t rel = s->plan->calls[k].ct_off - win.ct0;
    uint8_t* block = static_cast<uint8_t*>(ct_block) + (s->plan_retain ? s->plan->calls[k].ct_off : s->ct_ring ? s->sched.ring_off[k] : rel) * 16;
    const size_t n_gather = (!scatter && s->drain_instances) ? std::min(s->drain_instances, s->n_inst) : s->n_inst;  // gsv_session_set_drain_instances
    if (gsvk_gather_segment(block, s->ct_stride(), s->call_dev[k].dp.ct_pos, cp.n_ct, 1, uint32_t(n_gather), static_cast<uint8_t*>(gate_buf) + (s->plan->calls[k].ct_off - gate_ct0) * 16, gate_stride, scatter, stream) != 0)
      return fail(GSV_ERR_DEVICE, scatter ? "ciphertext scatter launch failed" : "ciphertext gather launch failed");
  }
  return GSV_OK;
}
// windows [w0, w1) that cover exactly the calls [c0, c1), or an error: a slice of a plan starts and ends on window boundaries
static int window_range(const gsv_session* s, size_t c0, size_t c1, size_t* w0, size_t* w1) {
  const auto& ws = s->sched.windows;
  size_t a = 0;
  while (a < ws.size() && ws[a].call0 < c0) ++a;
  size_t b = a;
  while (b < ws.size() && ws[b].call1 <= c1) ++b;
  if (c0 == c1) { *w0 = *w1 = a; return GSV_OK; }
  if (a >= ws.size() || ws[a].call0 != c0 || b == a || ws[b - 1].call1 != c1)
    return fail(GSV_ERR_INVALID, "a slice of a plan session must start and end on window boundaries of its schedule (gsv_session_plan_window)");
  *w0 = a; *w1 = b;
  return GSV_OK;
}
static int gather_plan_outputs(gsv_session* s, bool eval) {
  const Program& f = s->facade;
  if (!f.output_slots.empty()) {
    if (gsvk_gather_outputs(s->W, s->VB, f.n_slots, static_cast<const uint32_t*>(s->plan_out_slots), uint32_t(f.output_slots.size()), uint32_t(s->n_inst), s->out,
                            eval ? s->out_bits : nullptr, s->e->stream) != 0) return fail(GSV_ERR_DEVICE, "gather launch failed");
  }
  s->ran = true; s->last_eval = eval;
  return GSV_OK;
}
static int launch_plan(gsv_session* s, uint64_t gate_id_base, bool eval) {
  if (!s->plan_retain) return fail(GSV_ERR_INVALID, "this plan session keeps one window of ciphertexts only: use gsv_session_garble_streaming");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipMemsetAsync(s->d_error, 0, 4, s->e->stream));  // every pass starts with a clean dependency-wait flag
  HIPCHK(hipEventRecord(s->ev0, s->e->stream));
  for (size_t w = 0; w < s->sched.windows.size(); ++w) {
    int rc = launch_plan_window(s, w, gate_id_base, eval);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(s->ev1, s->e->stream));
  return gather_plan_outputs(s, eval);
}

int gsv_session_garble(gsv_session* s, uint64_t gate_id_base) {
  if (!s) return fail(GSV_ERR_INVALID, "null session");
  int rc = launch(s, gate_id_base, false);
  if (rc == GSV_OK) s->garbled = true;
  return rc;
}
// Garble + drain.  The launch is cut into segments of one device ring (ct_cap replays).  After a segment the ring
// (program order) is gathered into a second device buffer in GATE order (a ~ms kernel between two garbling launches,
// which own every CU while they run); while the next segment is garbled, host threads copy that buffer out with plain
// sequential D2H copies (the copy engines work beside the kernel), fold each instance's bytes into its CBC-MAC (strictly
// serial per instance, hence the host: ciphertext_hasher.rs:23-29) and optionally append them to gc_<index>.bin
// (ciphertext_repository.rs:94-127).
//
// The drain machinery (copy streams, pinned chunk buffers, the per-instance MAC states) lives in the session, so that a plan can
// be garbled in SLICES of consecutive calls (gsv_session_garble_streaming_calls): the MACs chain from slice to slice and the
// page-locked buffers are set up once.
struct gsv_drain {
  // Many host threads are wanted for the MACs (one serial chain per instance) but only a few D2H copies should be in
  // flight at once: measured on the MI355X box, 128 concurrent copy streams move 9 GB/s where a handful move 22 GB/s
  // (and the number of STREAMS matters as much as the number of copies: the copies share a small pool of streams).
  struct CopyGate {
    std::mutex mu; std::condition_variable cv; std::vector<hipStream_t> idle;
    hipStream_t acquire() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !idle.empty(); }); hipStream_t st = idle.back(); idle.pop_back(); return st; }
    void release(hipStream_t st) { { std::lock_guard<std::mutex> lk(mu); idle.push_back(st); } cv.notify_one(); }
  } copy_gate;
  std::vector<hipStream_t> copy_streams;
  // Instances whose MAC chains one worker advances side by side: four (AES-NI, CbcMacHost::update_interleaved) or, on hosts with
  // VAES + AVX-512 and sessions with at least 128 instances (eight workers' worth), sixteen (update_interleaved16_vaes: one core
  // then MACs ~3 x as many blocks per second, so a node's GPUs need a third of the host cores for their commitments).
  static constexpr int GROUP_MAX = 16;
  // CPUs this process may actually use: the visible ones, capped by the container's CPU bandwidth quota (cgroup cpu.max)
  static size_t usable_cores() {
    size_t n = std::max<size_t>(1, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64] = {0}; double per = 0;
      if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) n = std::min<size_t>(n, std::max<size_t>(1, size_t(std::atof(q) / per)));
      std::fclose(f);
    }
    return n;
  }
  // One chain per worker while there is a core per instance: a chain alone advances at ~1.1e8 blocks/s, one of four interleaved at
  // ~0.75e8 — sixteen instances (BASELINE config 5 on one GPU) hashed four to a worker took 40 s for 34.8 s of garbling, their
  // sixteen chains one per core take 27 s.  More instances than cores: four (AES-NI) or sixteen (VAES, >= 128 instances) per worker.
  static int group_for(size_t n_inst) {
    if (const char* e = getenv("GSV_DRAIN_GROUP")) { const int v = atoi(e); if (v == 1 || v == 4 || v == 16) return v; }
    if (n_inst <= usable_cores()) return 1;
    return CbcMacHost::have_vaes() && n_inst >= 128 ? 16 : 4;
  }
  int group = 4;
  struct Worker {
    void* pinned[2][GROUP_MAX] = {};  // two sets of pinned chunk buffers: copy set j+1 while set j is hashed
    hipEvent_t done = nullptr;    // blocking-sync event: a worker waiting for its copies sleeps instead of spinning on a core
  };
  std::vector<Worker> workers;
  uint64_t chunk = 0;  // records per chunk buffer
  std::vector<CbcMacHost> macs;
  ~gsv_drain() {
    for (Worker& w : workers) {
      for (auto& set : w.pinned) for (void*& q : set) if (q) (void)hipHostFree(q);
      if (w.done) (void)hipEventDestroy(w.done);
    }
    for (hipStream_t st : copy_streams) (void)hipStreamDestroy(st);
  }
};
static void destroy_drain(gsv_drain* d) { delete d; }
static int ensure_drain(gsv_session* s, size_t T, uint64_t seg_records, int group) {
  // records per chunk: 16 MiB by default — measured on the MI355X box (tools/d2h_bw.py) a D2H copy stream moves 39-48 GB/s in 4 MiB
  // pieces and 54-57 GB/s from 16 MiB up; the buffers are page-locked once per session, not per call as in round 1
  const uint64_t chunk_mb = getenv("GSV_DRAIN_CHUNK_MB") ? std::max(1, atoi(getenv("GSV_DRAIN_CHUNK_MB"))) : 16;
  const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(seg_records, 1), (chunk_mb << 20) / 16);
  if (s->drain && s->drain->workers.size() >= T && s->drain->chunk == chunk && s->drain->group == group) return GSV_OK;
  std::vector<CbcMacHost> keep;
  if (s->drain) keep = s->drain->macs;
  destroy_drain(s->drain);
  s->drain = new gsv_drain();
  gsv_drain& d = *s->drain;
  d.macs = keep;
  d.chunk = chunk;
  d.group = group;
  // Copy sets in flight at once.  Round 3, whole Miller-loop pass at 64 instances (tools/e2e_plan_drain.py, profiles/r03_e2e/): 1 set
  // 48 GB/s, 2-4 sets 50 GB/s, 6 sets 40 GB/s, 12 sets 41 GB/s — the link is full with two or three 16 MiB copies queued.
  const int n_copy_streams = getenv("GSV_DRAIN_COPIES") ? std::max(1, atoi(getenv("GSV_DRAIN_COPIES"))) : 3;
  bool ok = true;
  for (int k = 0; k < n_copy_streams && ok; ++k) {
    hipStream_t st;
    ok = create_side_stream(&st) == hipSuccess;
    if (ok) { d.copy_streams.push_back(st); d.copy_gate.idle.push_back(st); }
  }
  d.workers.resize(T);
  for (gsv_drain::Worker& w : d.workers) {
    for (auto& set : w.pinned) for (int g = 0; g < group; ++g) ok = ok && hipHostMalloc(&set[g], chunk * 16, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&w.done, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) { destroy_drain(s->drain); s->drain = nullptr; return fail(GSV_ERR_DEVICE, "cannot allocate the drain buffers"); }
  return GSV_OK;
}

// Discarding form: calls [c0, c1) of a plan (or the whole program launch), ciphertexts stay in / are overwritten on the device.
static int garble_discard(gsv_session* s, uint64_t gate_id_base, size_t c0, size_t c1) {
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipEventRecord(s->ev0, s->e->stream));
  int rc = GSV_OK;
  if (s->plan) {
    size_t w0 = 0, w1 = 0;
    if (c0 == 0) HIPCHK(hipMemsetAsync(s->d_error, 0, 4, s->e->stream));
    if (s->ct_ring) __atomic_store_n(s->host_ct_pos, ~0ull, __ATOMIC_RELEASE);  // nothing reads the ciphertexts: every block of the ring is free at once
    rc = window_range(s, c0, c1, &w0, &w1);
    for (size_t w = w0; w < w1 && rc == GSV_OK; ++w) rc = launch_plan_window(s, w, gate_id_base, false);
    if (rc == GSV_OK) {
      HIPCHK(hipEventRecord(s->ev1, s->e->stream));
      if (c1 == s->plan->calls.size()) rc = gather_plan_outputs(s, false); else { s->ran = true; s->last_eval = false; }
    }
  } else {
    rc = launch(s, gate_id_base, false);
  }
  if (rc == GSV_OK) { HIPCHK(hipStreamSynchronize(s->e->stream)); s->garbled = !s->plan || s->plan_retain; }
  if (rc == GSV_OK && s->plan) rc = check_plan_error(s);
  return rc;
}

// Where a drained stream goes (any combination): the per-instance CBC-MAC (AESAccumulatingHash), gc_<index>.bin files, a host callback.
struct DrainSink {
  uint8_t* hashes = nullptr;           // n_inst x 16: the MAC states after this call
  const char* dir = nullptr;           // gc_<index>.bin, index = indexes ? indexes[i] : first_index + i
  uint64_t first_index = 0;
  gsv_ct_sink_fn fn = nullptr;         // CiphertextHandler::handle over a run of records of one instance
  void* user = nullptr;
  bool any() const { return hashes || dir || fn; }
};
// Garble -> evaluate on the device (gsv_session_garble_evaluate): the evaluator session consumes window k from the garbler's
// program-order block while the garbler writes window k+1 into the other one of two blocks.
struct PairState {
  hipStream_t stream = nullptr;                           // the evaluator's launches
  hipStream_t gstream = nullptr;                          // CU-masked pairs: the garbler's launches (else they go to the engine's stream)
  hipEvent_t ready = nullptr;                             // engine stream -> gstream hand-over at the start of a pass
  hipEvent_t garbled[2] = {nullptr, nullptr}, evaluated[2] = {nullptr, nullptr};
};
static int ensure_pair(gsv_session* s) {
  if (!s->ct_alt) DEVALLOC(&s->ct_alt, s->n_inst * size_t(s->ct_stride()) * 16, "the second ciphertext block (garble -> evaluate)");
  if (!s->pair) {
    std::unique_ptr<PairState> ps(new PairState());
    // The evaluator's stream: same priority as the engine's, on ANOTHER hardware queue.  Which queue a new stream lands on is the
    // runtime's business (round-robin over a few), so every candidate is probed — a one-thread kernel on the engine's stream waits up to
    // 5 ms for a one-thread kernel on the candidate — and the ones that queue up behind the engine's stream are kept alive until a
    // good one is found (the round-robin moves on), then destroyed.  No overlapping stream among eight: the last one serves (the pair
    // is still correct, window k is then evaluated after window k+1 has been garbled instead of beside it).
    // Round 5: the two long launches get DISJOINT sets of CUs through CU-masked streams (hipExtStreamCreateWithCUMask): a masked stream
    // owns a hardware queue of its own (the mask is a queue property), so the overlap no longer depends on which queue the runtime's
    // round-robin picks, and neither launch's waiting workgroups — a window holds more calls than run at once, the rest spin on their
    // dependency flags with a whole CU's LDS each — can sit on the CUs the other one needs (the 40 - 65 s run-to-run spread of round 4).
    // Three quarters of the CUs garble (two AES blocks per AND), a quarter evaluates (one).  The mask bits alternate in blocks of eight,
    // 3 : 1: whichever way the runtime maps bits to XCDs / shader engines, every XCD keeps CUs of both launches.  GSV_PAIR_CU_MASK=0, or a
    // runtime that refuses the masks, falls back to the probed unmasked stream below.
    if (!(getenv("GSV_PAIR_CU_MASK") && atoi(getenv("GSV_PAIR_CU_MASK")) == 0)) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, s->e->device) == hipSuccess && prop.multiProcessorCount >= 32) {
        const uint32_t n_cu = uint32_t(prop.multiProcessorCount), words = (n_cu + 31) / 32;
        std::vector<uint32_t> gm(words, 0), em(words, 0);
        for (uint32_t i = 0; i < n_cu; ++i) (((i / 8) % 4 == 3) ? em : gm)[i / 32] |= 1u << (i % 32);
        hipStream_t g = nullptr, e2 = nullptr;
        if (hipExtStreamCreateWithCUMask(&g, words, gm.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&e2, words, em.data()) == hipSuccess &&
            hipEventCreateWithFlags(&ps->ready, hipEventDisableTiming) == hipSuccess) {
          ps->gstream = g; ps->stream = e2;
          if (getenv("GSV_DRAIN_DEBUG")) std::fprintf(stderr, "garble -> evaluate: CU-masked streams, %u CUs garble, %u evaluate\n", n_cu - n_cu / 4, n_cu / 4);
        } else {
          (void)hipGetLastError();
          if (g) (void)hipStreamDestroy(g);
          if (e2) (void)hipStreamDestroy(e2);
        }
      }
    }
    if (!ps->stream) {
      std::vector<hipStream_t> rejected;
      uint32_t* const word = static_cast<uint32_t*>(s->d_error) + 4;
      for (int attempt = 0; attempt < 8 && !ps->stream; ++attempt) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        uint32_t result[2] = {0, 0};
        const bool probed = hipMemsetAsync(word, 0, 8, s->e->stream) == hipSuccess && hipStreamSynchronize(s->e->stream) == hipSuccess &&
                            gsvk_probe_overlap(word, 500000ull, s->e->stream, cand) == 0 && hipStreamSynchronize(cand) == hipSuccess &&
                            hipStreamSynchronize(s->e->stream) == hipSuccess && hipMemcpy(result, word, 8, hipMemcpyDeviceToHost) == hipSuccess;
        if (!probed || result[1] == 1u || attempt == 7) ps->stream = cand;
        else rejected.push_back(cand);
        if (getenv("GSV_DRAIN_DEBUG")) std::fprintf(stderr, "garble -> evaluate: candidate stream %d %s\n", attempt, !probed ? "could not be probed" : result[1] == 1u ? "overlaps the engine's stream" : "queues behind the engine's stream");
      }
      for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
      if (!ps->stream) return fail(GSV_ERR_DEVICE, "cannot create the evaluator's stream");
    }
    for (int b = 0; b < 2; ++b) {
      HIPCHK(hipEventCreateWithFlags(&ps->garbled[b], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ps->evaluated[b], hipEventDisableTiming));
    }
    s->pair = ps.release();
  }
  return GSV_OK;
}
static void destroy_pair(PairState* ps) {
  if (!ps) return;
  for (int b = 0; b < 2; ++b) { if (ps->garbled[b]) (void)hipEventDestroy(ps->garbled[b]); if (ps->evaluated[b]) (void)hipEventDestroy(ps->evaluated[b]); }
  if (ps->stream) (void)hipStreamDestroy(ps->stream);
  if (ps->gstream) (void)hipStreamDestroy(ps->gstream);
  if (ps->ready) (void)hipEventDestroy(ps->ready);
  delete ps;
}

// Follows the RUNNING window w through the completion counters its workgroups write into mapped host memory (kernels.hip, epilogue)
// until calls [k0, k1) of the plan have completed for every instance group, or the window's launch itself has finished (*window_done).
static int wait_calls_done(gsv_session* s, size_t w, uint32_t k0, uint32_t k1, bool* window_done, hipStream_t launch_stream = nullptr) {
  if (!launch_stream) launch_stream = s->e->stream;  // the stream the running window was launched on
  const Schedule::Window& win = s->sched.windows[w];
  const uint32_t n_wg = uint32_t((s->n_inst + s->ni - 1) / s->ni);
  const auto t0 = std::chrono::steady_clock::now();
  bool reported = false;
  // Host-side deadline, progress based like the device's watchdog (kernels.hip) and longer than it: the device gives up after
  // GSV_DEP_WAIT_SECONDS (default 60) without a completed call of an instance group and then ENDS its launch, which the stream query below
  // sees; this deadline covers the device that never comes back at all (no counter of the window has moved for twice that time + 30 s).
  double dev_secs = 60.0;
  if (const char* ev = getenv("GSV_DEP_WAIT_SECONDS")) { char* end = nullptr; const double v = std::strtod(ev, &end); if (end != ev && v > 0) dev_secs = std::min(v, 86400.0); }
  const double deadline = 2.0 * dev_secs + 30.0;
  uint64_t last_sum = ~0ull;
  auto last_move = t0;
  uint32_t polls = 0;
  while (!*window_done && k0 < k1) {
    bool all = true;
    for (uint32_t k = k0; k < k1 && all; ++k) all = __atomic_load_n(s->host_done + k, __ATOMIC_ACQUIRE) == n_wg;
    if (all) break;
    const hipError_t q = hipStreamQuery(launch_stream);
    if (q == hipSuccess) { *window_done = true; break; }
    if (q != hipErrorNotReady) {  // a failed launch / a lost device is neither "done" nor "running": the caller's error path must run
      (void)hipGetLastError();
      return fail(GSV_ERR_DEVICE, std::string("the window's launch failed while its stream was being drained: ") + hipGetErrorString(q));
    }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
    if ((++polls & 1023u) == 0) {  // every ~0.1 s: has any call of the window completed for another workgroup?
      uint64_t sum = 0;
      for (uint32_t k = win.call0; k < win.call1; ++k) sum += __atomic_load_n(s->host_done + k, __ATOMIC_RELAXED);
      const auto now = std::chrono::steady_clock::now();
      if (sum != last_sum) { last_sum = sum; last_move = now; }
      else if (std::chrono::duration<double>(now - last_move).count() > deadline)
        return fail(GSV_ERR_DEVICE, "no call of the running window has completed for " + std::to_string(int(deadline)) + " s and its launch has not ended: giving up on the device");
    }
    if (!reported && getenv("GSV_DRAIN_DEBUG") && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3.0) {
      reported = true;
      std::string msg;
      for (uint32_t k = win.call0; k < win.call1; ++k)
        if (s->host_done[k] != n_wg) msg += " " + std::to_string(k) + "(" + std::to_string(s->host_done[k]) + "/" + std::to_string(n_wg) + (s->ct_ring ? ",need " + std::to_string(s->sched.ring_need[k]) + ",ready " + std::to_string(s->sched.seg_end[k]) : "") + ")";
      std::fprintf(stderr, "drain debug: waiting > 3 s for calls [%u, %u) of window %zu; host position %llu; unfinished:%s\n", k0, k1, w, s->host_ct_pos ? (unsigned long long)*s->host_ct_pos : 0ull, msg.substr(0, 1500).c_str());
    }
  }
  return GSV_OK;
}
// the side stream and the device-written completion counters of a session whose stream leaves the device while a window runs
// The gate-order buffers (ct_gate + the drain pipeline's further ones) hold `bytes` each — or are released and ct_gate re-allocated.  A
// sample drain (gsv_session_set_drain_instances) sizes them for the sample; a later call over more instances (a full drain, or any
// evaluate_streaming: the evaluator uploads EVERY instance's stream) must not write past them.
static int ensure_ct_gate(gsv_session* s, size_t bytes) {
  if (s->ct_gate && s->ct_gate_bytes >= bytes) return GSV_OK;
  if (s->ct_gate || !s->ct_gate_more.empty()) {
    HIPCHK(hipStreamSynchronize(s->e->stream));
    if (s->aux_stream) HIPCHK(hipStreamSynchronize(s->aux_stream));
    for (void* q : s->ct_gate_more) if (q) (void)hipFree(q);
    s->ct_gate_more.clear();
    if (s->ct_gate) (void)hipFree(s->ct_gate);
    s->ct_gate = nullptr; s->ct_gate_bytes = 0;
  }
  DEVALLOC(&s->ct_gate, bytes, "the gate-order ciphertext buffer");
  s->ct_gate_bytes = bytes;
  return GSV_OK;
}
static int ensure_aux(gsv_session* s) {
  if (!s->aux_stream) HIPCHK(create_side_stream(&s->aux_stream));
  if (!s->host_done) return fail(GSV_ERR_INVALID, "internal: a plan session without completion counters");  // (allocated with its call descriptors)
  return GSV_OK;
}

// `ev`: an evaluator session over the same plan / schedule (plan sessions that do not retain the stream): every window is evaluated
// straight from the garbler's device block while the next window is garbled.
static int garble_streaming_pass(gsv_session* s, uint64_t gate_id_base, size_t c0, size_t c1, const DrainSink& sink, int n_threads, gsv_session* ev) {
  if (!sink.any() && !ev) return garble_discard(s, gate_id_base, c0, c1);  // garble only (output labels, device-rate measurements of long plans / chains)
  const Program& g = s->prog();
  // program sessions: segments of one ring (ct_cap replays of n_ct records); plan sessions: one WINDOW of the schedule per segment
  size_t pw0 = 0, pw1 = 0;
  if (s->plan) { int wrc = window_range(s, c0, c1, &pw0, &pw1); if (wrc) return wrc; }
  // (plan sessions: the stream leaves the device in SEGMENTS of a window, a gate-order buffer holds the largest segment)
  const uint64_t n_ct = s->plan ? s->plan_max_segment : g.n_ct, seg = s->plan ? 1 : s->ct_cap;
  const uint64_t first = s->plan ? pw0 : 0, total = s->plan ? pw1 : s->replays;
  const bool new_pass = s->plan ? c0 == 0 : true;
  // the instances whose streams leave the device: all of them, or the first drain_instances (every instance is garbled either way)
  const size_t n_inst = s->drain_instances ? std::min(s->drain_instances, s->n_inst) : s->n_inst;
  const bool want_drain = sink.any();
  const bool want_mac = sink.hashes != nullptr;
  const size_t GROUP = size_t(gsv_drain::group_for(n_inst));
  const size_t n_groups = (n_inst + GROUP - 1) / GROUP;
  // a worker MACs GROUP streams side by side at ~3e8 blocks/s (four chains, AES-NI) or ~1e9 (sixteen, VAES): a dozen / four of them keep
  // up with the PCIe link, 32 leave room for slow cores without page-locking more than 4 GB (16 GB) of chunk buffers
  size_t T = n_threads > 0 ? size_t(n_threads) : std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n_groups, GROUP == 16 ? 8 : 32), std::thread::hardware_concurrency()));
  T = std::min(T, n_groups);
  HIPCHK(hipSetDevice(s->e->device));
  const uint64_t seg_records = seg * n_ct;  // per instance
  if (want_drain) {
    if (seg_records) { int grc = ensure_ct_gate(s, n_inst * size_t(seg_records) * 16); if (grc) return grc; }
    int rc = ensure_drain(s, T, seg_records, int(GROUP));
    if (rc) return rc;
  }
  if (ev && (s->ct_ring || ev->ct_ring)) return fail(GSV_ERR_INVALID, "garble || evaluate pairs need sessions with explicit launch windows (window_ct_records, e.g. 1 << 28): the default is one whole-pass window over a ciphertext ring");
  if (ev) { int rc = ensure_pair(s); if (rc) return rc; }
  if (s->ct_ring) __atomic_store_n(s->host_ct_pos, (unsigned long long)(s->plan && pw0 < s->sched.windows.size() ? s->sched.windows[pw0].ct0 : 0), __ATOMIC_RELEASE);
  if (s->plan && want_drain) { int rc = ensure_aux(s); if (rc) return rc; }
  if (s->plan && new_pass) HIPCHK(hipMemsetAsync(s->d_error, 0, 4, s->e->stream));  // a new pass starts with a clean dependency-wait flag
  if (ev && new_pass) HIPCHK(hipMemsetAsync(ev->d_error, 0, 4, s->e->stream));
  std::vector<CbcMacHost> no_macs;
  if (want_drain && (new_pass || s->drain->macs.size() != n_inst)) s->drain->macs.assign(n_inst, CbcMacHost());  // a new pass starts from h = 0; later slices chain
  std::vector<CbcMacHost>& macs = want_drain ? s->drain->macs : no_macs;
  std::vector<FILE*> files(n_inst, nullptr);
  std::vector<std::string> paths(n_inst);
  auto close_files = [&]() { for (FILE*& f : files) if (f) { std::fclose(f); f = nullptr; } };
  // a failed pass must not leave a plausible-looking prefix of a ciphertext file behind
  auto remove_files = [&]() { if (sink.dir) for (const std::string& q : paths) if (!q.empty()) std::remove(q.c_str()); };
  if (sink.dir)
    for (size_t i = 0; i < n_inst; ++i) {
      paths[i] = std::string(sink.dir) + "/gc_" + std::to_string(sink.first_index + i) + ".bin";
      files[i] = std::fopen(paths[i].c_str(), new_pass ? "wb" : "ab");
      if (!files[i]) { close_files(); return fail(GSV_ERR_INVALID, "cannot create " + paths[i]); }
    }
  std::atomic<int> err{0};
  const uint64_t chunk = want_drain ? s->drain->chunk : 0;
  // The drain is a PIPELINE of segments: the device side (garble a window, bring it into gate order in one of `depth` gate-order
  // buffers) runs ahead of the host side (copy out, CBC-MAC, files, sink) by up to `depth` segments.  A window's ciphertext count is
  // fixed but its garbling time is not (the ladders and inversions produce a gigabyte of ciphertexts in seconds, the Miller loop in
  // half a second), while the serial CBC-MAC chain takes the same 0.58 s for every gigabyte: with ONE buffer a pass costs
  // sum(max(garble_w, mac_w)) — 37.0 s for one instance whose garbling takes 31 s and whose chain takes 27 s — with a few buffers
  // max(sum garble, sum mac).  Workers are persistent for the call and take the segments strictly in order (an instance's chain must
  // see its stream in order); a buffer is reused once every worker is done with the segment that held it.
  std::vector<void*> gate_bufs;
  if (want_drain) {
    gate_bufs.push_back(s->ct_gate);
    size_t want = 1;
    size_t n_units = size_t((total - first + seg - 1) / seg);  // drain units of this call: segments (plans) or rings
    if (s->plan) { n_units = 0; for (size_t w = pw0; w < pw1; ++w) n_units += s->sched.windows[w].seg1 - s->sched.windows[w].seg0; }
    if (seg_records && n_units > 1) {
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const size_t buf_bytes = n_inst * size_t(seg_records) * 16;
      // up to eight buffers, within half of the free memory and 32 GB (allocating device memory takes time too: ~25 GB/s)
      want = std::min<size_t>(std::min<size_t>(8, 1 + size_t(double(free_b) * 0.5 / double(buf_bytes))), std::max<size_t>(2, size_t(32e9 / double(buf_bytes))));
      if (const char* e = getenv("GSV_DRAIN_DEPTH")) want = size_t(std::max(1, atoi(e)));
    }
    while (1 + s->ct_gate_more.size() < want) {
      void* q = nullptr;
      if (hipMalloc(&q, s->ct_gate_bytes) != hipSuccess) { (void)hipGetLastError(); break; }  // (every buffer of the pipeline has ct_gate's capacity)
      s->ct_gate_more.push_back(q);
    }
    for (void* q : s->ct_gate_more) if (gate_bufs.size() < want) gate_bufs.push_back(q);
  }
  const size_t depth = std::max<size_t>(1, gate_bufs.size());
  struct Segment { uint64_t n, base; size_t buf; };
  std::mutex q_mu;
  std::condition_variable q_cv;
  std::vector<Segment> segments;          // pushed by the device side, in stream order
  std::vector<size_t> seg_done;           // per segment: workers that have finished it
  bool q_closed = false;
  auto worker_main = [&](size_t t) {
    if (hipSetDevice(s->e->device) != hipSuccess) { err = 1; }
    gsv_drain& dr = *s->drain;
    gsv_drain::Worker& w = dr.workers[t];
    for (size_t j = 0;; ++j) {
      Segment sg;
      {
        std::unique_lock<std::mutex> lk(q_mu);
        q_cv.wait(lk, [&] { return j < segments.size() || q_closed; });
        if (j >= segments.size()) return;
        sg = segments[j];
      }
      const uint64_t n = sg.n, base = sg.base;
      const uint8_t* const gate = static_cast<const uint8_t*>(gate_bufs[sg.buf]);
      for (size_t grp = t; grp < n_groups && !err && n; grp += T) {
        const size_t i0 = grp * GROUP, ng = std::min(GROUP, n_inst - i0);  // instances i0 .. i0+ng-1 advance together
        // the copies of one chunk set share a stream of the pool (a set holds a slot of the gate from issue to completion)
        auto copy = [&](uint64_t off, int b) {
          hipStream_t st = dr.copy_gate.acquire();
          bool ok = true;
          for (size_t g = 0; g < ng && ok; ++g)
            ok = hipMemcpyAsync(w.pinned[b][g], gate + ((i0 + g) * seg_records + off) * 16, std::min(chunk, n - off) * 16, hipMemcpyDeviceToHost, st) == hipSuccess;
          // many workers: sleep on the blocking-sync event (spinning workers eat the cores the MACs need); a handful of
          // workers (one instance: the whole-stream check) spin instead, a blocking wait's wake-up latency would be paid per chunk
          ok = ok && (T > 8 ? hipEventRecord(w.done, st) == hipSuccess && hipEventSynchronize(w.done) == hipSuccess : hipStreamSynchronize(st) == hipSuccess);
          dr.copy_gate.release(st);
          return ok;
        };
        int b = 0;
        if (!copy(0, 0)) { err = 1; break; }
        for (uint64_t off = 0; off < n; off += chunk, b ^= 1) {
          const uint64_t m = std::min(chunk, n - off);
          if (want_mac) {
            CbcMacHost* mp[gsv_drain::GROUP_MAX];
            const uint8_t* cp[gsv_drain::GROUP_MAX];
            for (size_t g = 0; g < ng; ++g) { mp[g] = &macs[i0 + g]; cp[g] = static_cast<const uint8_t*>(w.pinned[b][g]); }
            CbcMacHost::update_many(mp, cp, ng, m);  // sixteen / four chains per step, a ragged last group chain by chain
          }
          if (sink.dir)
            for (size_t g = 0; g < ng; ++g)
              if (std::fwrite(w.pinned[b][g], 16, m, files[i0 + g]) != m) { err = 2; break; }
          if (sink.fn && !err)
            for (size_t g = 0; g < ng; ++g)
              if (sink.fn(sink.user, i0 + g, base + off, static_cast<const uint8_t*>(w.pinned[b][g]), m) != 0) { err = 3; break; }
          if (err) break;
          if (off + chunk < n && !copy(off + chunk, b ^ 1)) { err = 1; break; }
        }
      }
      {
        std::lock_guard<std::mutex> lk(q_mu);
        seg_done[j]++;
      }
      q_cv.notify_all();
    }
  };
  std::vector<std::thread> workers;
  if (want_drain) for (size_t t = 0; t < T; ++t) workers.emplace_back(worker_main, t);
  // GSV_DRAIN_STATS=1: where the host thread of the pipeline waits (for a free gate-order buffer = the host side is the slower stage;
  // for the kernel + gather = the device is), printed once per call
  const bool stats = getenv("GSV_DRAIN_STATS") != nullptr;
  double t_wait_drain = 0, t_wait_device = 0, t_gather = 0;
  uint64_t drained_records = 0;
  const auto t_begin = std::chrono::steady_clock::now();
  auto secs = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  // blocks until the segment that last used buffer `b` has been consumed by every worker (segment index = its position in `segments`)
  auto wait_buffer = [&](size_t n_pushed) {
    if (!want_drain || n_pushed < depth) return;
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(q_mu);
    q_cv.wait(lk, [&] { return seg_done[n_pushed - depth] == T; });
    t_wait_drain += secs(t0);
  };
  auto push_segment = [&](uint64_t n, uint64_t base, size_t buf) {
    { std::lock_guard<std::mutex> lk(q_mu); segments.push_back(Segment{n, base, buf}); seg_done.push_back(0); }
    q_cv.notify_all();
  };
  auto finish_workers = [&]() {
    { std::lock_guard<std::mutex> lk(q_mu); q_closed = true; }
    q_cv.notify_all();
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& th : workers) th.join();
    workers.clear();
    t_wait_drain += secs(t0);
  };
  size_t n_pushed = 0;
  auto t_last_pub = std::chrono::steady_clock::now();
  double worst_gap = 0;
  s->ring_diag.clear();
  struct BlockingEvent { hipEvent_t ev = nullptr; ~BlockingEvent() { if (ev) (void)hipEventDestroy(ev); } } device_done_owner;
  if (want_drain && T + 1 > gsv_drain::usable_cores()) (void)hipEventCreateWithFlags(&device_done_owner.ev, hipEventBlockingSync | hipEventDisableTiming);
  const hipEvent_t device_done = device_done_owner.ev;
  int rc = GSV_OK;
  // The stream the garbler's windows are launched on: the engine's, or — a garble || evaluate pair with CU-masked streams (ensure_pair) —
  // the pair's masked garbler stream, which first waits for whatever the engine's stream still holds (input staging, the memsets above).
  hipStream_t gs = s->e->stream;
  if (ev && s->pair->gstream) {
    gs = s->pair->gstream;
    if (hipEventRecord(s->pair->ready, s->e->stream) != hipSuccess || hipStreamWaitEvent(gs, s->pair->ready, 0) != hipSuccess) { finish_workers(); close_files(); return fail(GSV_ERR_DEVICE, "stream hand-over failed"); }
  }
  if (hipEventRecord(s->ev0, gs) != hipSuccess) { finish_workers(); close_files(); return fail(GSV_ERR_DEVICE, "hipEventRecord failed"); }
  for (uint64_t r0 = first; r0 < total && rc == GSV_OK; r0 += seg) {
    const uint64_t r1 = std::min(total, r0 + seg);
    uint64_t n_records, base;  // per instance, in this segment; stream index of its first record
    if (s->plan) {
      const size_t w = size_t(r0);
      void* block = s->CT;
      if (ev) {
        // window w goes to block w & 1; the evaluator must be done with what that block held (window w - 2)
        const int b = int(w & 1);
        block = b ? s->ct_alt : s->CT;
        if (w >= pw0 + 2 && hipStreamWaitEvent(gs, s->pair->evaluated[b], 0) != hipSuccess) { rc = fail(GSV_ERR_DEVICE, "hipStreamWaitEvent failed"); break; }
      }
      rc = launch_plan_window(s, w, gate_id_base, false, block, gs);
      if (rc != GSV_OK) break;
      if (ev) {
        const int b = int(w & 1);
        if (hipEventRecord(s->pair->garbled[b], gs) != hipSuccess || hipStreamWaitEvent(s->pair->stream, s->pair->garbled[b], 0) != hipSuccess) { rc = fail(GSV_ERR_DEVICE, "event hand-over failed"); break; }
        rc = launch_plan_window(ev, w, gate_id_base, true, block, s->pair->stream);
        if (rc != GSV_OK) break;
        if (hipEventRecord(s->pair->evaluated[b], s->pair->stream) != hipSuccess) { rc = fail(GSV_ERR_DEVICE, "hipEventRecord failed"); break; }
      }
      // The window is running.  Its segments leave the device one after the other, in stream order, each as soon as every call of it
      // has completed for every instance group: the host follows the completion flags of the running launch (a page-locked copy,
      // refreshed through a side stream), brings the finished segment into gate order with a gather kernel on that side stream — the
      // session's schedule leaves it a few CUs — and hands it to the workers, while the window goes on garbling.
      const Schedule::Window& win = s->sched.windows[w];
      n_records = win.n_ct;
      base = win.ct0;
      if (want_drain) {
        bool window_done = false;
        for (uint32_t q = win.seg0; q < win.seg1 && rc == GSV_OK; ++q) {
          const Schedule::Segment& sg = s->sched.segments[q];
          const bool last = q + 1 == win.seg1;
          const auto t0 = std::chrono::steady_clock::now();
          const double drain_before = t_wait_drain;
          if (!last) rc = wait_calls_done(s, w, sg.call0, sg.call1, &window_done, gs);
          if (rc != GSV_OK) break;
          if (last && !window_done) {
            // the last segment ends with the window: sleep on the stream (on a blocking-sync event when the workers own the cores)
            const bool ok = device_done ? hipEventRecord(device_done, gs) == hipSuccess && hipEventSynchronize(device_done) == hipSuccess : hipStreamSynchronize(gs) == hipSuccess;
            if (!ok) { rc = fail(GSV_ERR_DEVICE, "kernel failed"); break; }
            window_done = true;
          }
          t_wait_device += secs(t0);
          wait_buffer(n_pushed);  // the gate-order buffer this segment goes to is free again
          const auto tg = std::chrono::steady_clock::now();
          rc = permute_plan_calls(s, w, sg.call0, sg.call1, sg.ct0, seg_records, 0, block, gate_bufs[n_pushed % depth], s->aux_stream);
          if (rc != GSV_OK) break;
          if (hipStreamSynchronize(s->aux_stream) != hipSuccess) { rc = fail(GSV_ERR_DEVICE, "ciphertext gather failed"); break; }
          t_gather += secs(tg);
          if (s->ct_ring) {
            __atomic_store_n(s->host_ct_pos, (unsigned long long)(sg.ct0 + sg.n_ct), __ATOMIC_RELEASE);  // the calls whose blocks overlap this segment's may write now
            const double gap = secs(t_last_pub);
            if (gap > worst_gap) {
              worst_gap = gap;
              char buf[256];
              std::snprintf(buf, sizeof buf, "longest interval between two positions %.2f s, before segment %u (calls [%u, %u)): %.2f s waiting for its calls, %.2f s for a free gate-order buffer, %.2f s gathering", gap, q,
                            sg.call0, sg.call1, std::chrono::duration<double>(tg - t0).count() - (t_wait_drain - drain_before), t_wait_drain - drain_before, secs(tg));
              s->ring_diag = buf;
            }
            t_last_pub = std::chrono::steady_clock::now();
          }
          drained_records += sg.n_ct;
          push_segment(sg.n_ct, sg.ct0, n_pushed % depth);
          ++n_pushed;
        }
        if (rc != GSV_OK) break;
      }
      if (hipStreamSynchronize(gs) != hipSuccess) { rc = fail(GSV_ERR_DEVICE, "kernel failed"); break; }
      continue;
    } else {
      // ring slots are (replay % ct_cap): a segment starts at a multiple of ct_cap, so its replays sit in slots 0..n_rep-1
      rc = launch(s, gate_id_base, false, r0, r1 - r0);
      if (rc != GSV_OK) break;
      wait_buffer(n_pushed);
      n_records = (r1 - r0) * n_ct;
      base = r0 * n_ct;
      if (gsvk_gather_segment(s->CT, s->ct_stride(), s->dp.ct_pos, n_ct, uint32_t(r1 - r0), uint32_t(n_inst), gate_bufs[n_pushed % depth], seg_records, 0, s->e->stream) != 0) {
        rc = fail(GSV_ERR_DEVICE, "ciphertext gather launch failed");
        break;
      }
    }
    {
      // the workers own the cores when there is one chain per core: this thread then sleeps on a blocking-sync event instead of
      // spinning in hipStreamSynchronize
      const auto t0 = std::chrono::steady_clock::now();
      const bool ok = device_done ? hipEventRecord(device_done, s->e->stream) == hipSuccess && hipEventSynchronize(device_done) == hipSuccess : hipStreamSynchronize(s->e->stream) == hipSuccess;
      if (!ok) { rc = fail(GSV_ERR_DEVICE, "kernel failed"); break; }
      t_wait_device += secs(t0);
    }
    drained_records += n_records;
    if (want_drain) { push_segment(n_records, base, n_pushed % depth); ++n_pushed; }
  }
  if (s->ct_ring && rc != GSV_OK) {
    // a failed pass: calls of the running window may still wait for room in the ring — let them run out (the results are discarded)
    __atomic_store_n(s->host_ct_pos, ~0ull, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(gs);
  }
  finish_workers();
  if (ev && hipStreamSynchronize(s->pair->stream) != hipSuccess && rc == GSV_OK) rc = fail(GSV_ERR_DEVICE, "evaluation kernel failed");
  if (stats) {
    const double tot = secs(t_begin);
    std::fprintf(stderr, "drain: %.2f s for %zu instances x %llu records (%.1f GB/s), %zu MAC workers x %zu chains, %zu gate-order buffers; host thread waited %.2f s for drains, %.2f s for the device and %.2f s for the gathers of running windows\n", tot, n_inst,
                 (unsigned long long)drained_records, double(drained_records) * double(n_inst) * 16e-9 / tot, T, GROUP, depth, t_wait_drain, t_wait_device, t_gather);
  }
  if (rc == GSV_OK && s->plan) {
    (void)hipEventRecord(s->ev1, gs);  // (every window on gs has been synchronised: the output gather on the engine's stream follows safely)
    if (c1 == s->plan->calls.size()) {
      rc = gather_plan_outputs(s, false);
      if (rc == GSV_OK && ev) rc = gather_plan_outputs(ev, true);
    } else { s->ran = true; s->last_eval = false; }
  }
  close_files();
  if (rc == GSV_OK && s->plan) rc = check_plan_error(s);
  if (rc == GSV_OK && ev) rc = check_plan_error(ev);
  if (rc == GSV_OK && err) rc = fail(err == 2 ? GSV_ERR_INVALID : err == 3 ? GSV_ERR_INVALID : GSV_ERR_DEVICE,
                                     err == 2 ? "short write to a gc file" : err == 3 ? "the ciphertext sink reported an error" : "device copy failed while draining ciphertexts");
  if (rc != GSV_OK) { remove_files(); return rc; }
  if (want_mac) for (size_t i = 0; i < n_inst; ++i) macs[i].digest(sink.hashes + 16 * i);
  s->garbled = true;
  return GSV_OK;
}
// A dependency wait gave up (status 1): the schedule's one assumption — a workgroup only waits for workgroups with a smaller linear index,
// which the hardware dispatches first (include/gsv_engine.h, max_concurrent_calls) — did not hold on this device / driver.  The session
// is switched, in place, to the SAFE schedule: one call per launch, in stream order, no dependency wait on the device at all (the stream
// orders the launches) and no ciphertext ring.  Same plan images, same wire-file and ciphertext allocations (the safe schedule needs
// less of both; re-allocated if not), the host's last inputs re-staged.  Slower (every call ends with a launch boundary), never wrong.
static int fall_back_to_safe_schedule(gsv_session* s) {
  if (!s->plan || s->safe_mode) return fail(GSV_ERR_DEVICE, "internal: no safe schedule to fall back to");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipDeviceSynchronize());
  drop_schedule(s);
  gsv_plan_session_opts o = s->opts;
  o.max_concurrent_calls = 1;
  o.max_window_calls = 1;
  if (o.retain_stream == GSV_STREAM_RING) o.retain_stream = 0;
  s->safe_mode = true;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, s->e->device));
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  int rc = install_schedule(s, o, prop.multiProcessorCount, free_b);
  if (rc) return rc;
  const Program& f = s->facade;
  if (f.n_slots > s->w_slots_cap) {
    (void)hipFree(s->W); (void)hipFree(s->VB); s->W = s->VB = nullptr;
    DEVALLOC(&s->W, s->n_inst * size_t(f.n_slots) * 16, "the wire files");
    HIPCHK(hipMalloc(&s->VB, s->n_inst * size_t(f.n_slots)));
    s->w_slots_cap = f.n_slots;
  }
  HIPCHK(hipMemset(s->VB, 0, s->n_inst * size_t(f.n_slots)));
  if (s->ct_stride() > s->ct_records_cap) {
    (void)hipFree(s->CT); s->CT = nullptr;
    if (s->ct_alt) { (void)hipFree(s->ct_alt); s->ct_alt = nullptr; }
    DEVALLOC(&s->CT, s->n_inst * size_t(s->ct_stride()) * 16, "the ciphertext blocks");
    s->ct_records_cap = s->ct_stride();
  }
  ++s->n_fallbacks;
  s->dep_fault = false;
  s->garbled = false;
  std::fill(s->ct_uploaded.begin(), s->ct_uploaded.end(), 0);
  if (s->stash_kind == 1) return set_garble_inputs_impl(s, s->stash_delta.data(), s->stash_consts.data(), s->stash_inputs.data());
  if (s->stash_kind == 2) return set_evaluate_inputs_impl(s, s->stash_consts.data(), s->stash_inputs.data(), s->stash_bits.data());
  return GSV_OK;
}
int gsv_session_fallback_count(const gsv_session* s, uint64_t* n) {
  if (!s || !n) return fail(GSV_ERR_INVALID, "null argument");
  *n = s->n_fallbacks;
  return GSV_OK;
}
// A whole pass whose results the engine alone has seen (discarded, MAC'ed, written to gc files) is repeated on the safe schedule by
// itself; a pass that fed a host callback or an evaluator session, or a slice of a pass, fails as before — the host has consumed a
// prefix of a stream that is invalid, and a slice's call range follows the old schedule's windows — but leaves the session on the
// safe schedule, so that the host's own repeat of the pass (from gsv_session_set_garble_inputs on) succeeds.
static int garble_streaming_range(gsv_session* s, uint64_t gate_id_base, size_t c0, size_t c1, const DrainSink& sink, int n_threads, gsv_session* ev = nullptr) {
  int rc = garble_streaming_pass(s, gate_id_base, c0, c1, sink, n_threads, ev);
  if (rc != GSV_ERR_DEVICE || !s->plan || !s->dep_fault || s->safe_mode) return rc;
  const std::string first_error = g_err;
  const bool whole = c0 == 0 && c1 == s->plan->calls.size();
  int frc = fall_back_to_safe_schedule(s);
  if (frc) return fail(GSV_ERR_DEVICE, first_error + "; the fall-back to the safe schedule failed too: " + g_err);
  if (ev) { frc = fall_back_to_safe_schedule(ev); if (frc) return fail(GSV_ERR_DEVICE, first_error + "; the evaluator's fall-back to the safe schedule failed: " + g_err); }
  if (!whole || sink.fn || ev)
    return fail(GSV_ERR_DEVICE, first_error + "; the session now runs the safe schedule (one call per launch): repeat the pass from gsv_session_set_garble_inputs");
  if (getenv("GSV_DRAIN_DEBUG") || getenv("GSV_PLAN_DEBUG")) std::fprintf(stderr, "plan session: %s -- repeating the pass on the safe schedule (one call per launch)\n", first_error.c_str());
  return garble_streaming_pass(s, gate_id_base, 0, s->plan->calls.size(), sink, n_threads, nullptr);
}
static DrainSink mac_file_sink(uint8_t* hashes, const char* dir, uint64_t first_index) { DrainSink k; k.hashes = hashes; k.dir = dir; k.first_index = first_index; return k; }
int gsv_session_garble_streaming(gsv_session* s, uint64_t gate_id_base, const char* dir, uint64_t first_index, int n_threads, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  if (!s) return fail(GSV_ERR_INVALID, "null argument");
  if (dir && !hashes) return fail(GSV_ERR_INVALID, "null hash buffer");
  return garble_streaming_range(s, gate_id_base, 0, s->plan ? s->plan->calls.size() : 1, mac_file_sink(hashes, dir, first_index), n_threads);
}
// a slice must start a new pass or continue where the previous one ended
static int check_slice(gsv_session* s, uint64_t first_call, uint64_t n_calls) {
  if (!s || !s->plan) return fail(GSV_ERR_INVALID, "null session / not a plan session");
  if (first_call > s->plan->calls.size() || n_calls > s->plan->calls.size() - first_call) return fail(GSV_ERR_INVALID, "call range outside the plan");
  // wires, gate ids and the MAC states continue from slice to slice: a slice either starts a new pass or continues the previous one
  if (first_call != 0 && first_call != s->next_call && !s->unchecked_slices)
    return fail(GSV_ERR_INVALID, "slice starts at call " + std::to_string(first_call) + " but the previous slice ended at call " + std::to_string(s->next_call) +
                                     " (gsv_session_set_unchecked_slices for timing runs)");
  return GSV_OK;
}
int gsv_session_garble_streaming_calls(gsv_session* s, uint64_t gate_id_base, uint64_t first_call, uint64_t n_calls, const char* dir, uint64_t first_index, int n_threads, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  int rc = check_slice(s, first_call, n_calls);
  if (rc) return rc;
  if (dir && !hashes) return fail(GSV_ERR_INVALID, "null hash buffer");
  rc = garble_streaming_range(s, gate_id_base, size_t(first_call), size_t(first_call + n_calls), mac_file_sink(hashes, dir, first_index), n_threads);
  if (rc == GSV_OK) { s->next_call = first_call + n_calls; s->garbled = s->plan_retain && s->next_call == s->plan->calls.size(); }
  return rc;
}
// The generic CiphertextHandler: every drained run of records is handed to `sink` (gate order, per instance in stream order).
int gsv_session_garble_streaming_sink(gsv_session* s, uint64_t gate_id_base, uint64_t first_call, uint64_t n_calls, gsv_ct_sink_fn sink, void* user, int n_threads, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  if (!s || !sink) return fail(GSV_ERR_INVALID, "null argument");
  DrainSink k;
  k.hashes = hashes; k.fn = sink; k.user = user;
  if (!s->plan) return garble_streaming_range(s, gate_id_base, 0, 1, k, n_threads);
  if (first_call == 0 && n_calls == 0) n_calls = s->plan->calls.size();
  int rc = check_slice(s, first_call, n_calls);
  if (rc) return rc;
  rc = garble_streaming_range(s, gate_id_base, size_t(first_call), size_t(first_call + n_calls), k, n_threads);
  if (rc == GSV_OK) { s->next_call = first_call + n_calls; s->garbled = s->plan_retain && s->next_call == s->plan->calls.size(); }
  return rc;
}
// Garble and evaluate side by side on the device (examples/groth16_garble.rs:171-230: the garbler thread feeds the evaluator thread
// through a channel; here window k of the garbler's device block is evaluated while window k+1 is garbled).
int gsv_session_garble_evaluate(gsv_session* gs, gsv_session* es, uint64_t gate_id_base, int n_threads, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  if (!gs || !es || gs == es) return fail(GSV_ERR_INVALID, "null / identical sessions");
  if (!gs->plan || gs->plan != es->plan || gs->e != es->e || gs->n_inst != es->n_inst || gs->ni != es->ni || gs->hasher != es->hasher)
    return fail(GSV_ERR_INVALID, "garbler and evaluator must be plan sessions of the same plan, engine, instance count and hasher");
  if (gs->plan_retain || es->plan_retain) return fail(GSV_ERR_INVALID, "gsv_session_garble_evaluate is for sessions that do not retain the stream (retain_stream = 0)");
  const auto &wa = gs->sched.windows, &wb = es->sched.windows;
  if (wa.size() != wb.size() || gs->plan_max_block != es->plan_max_block) return fail(GSV_ERR_INVALID, "garbler and evaluator sessions have different schedules (create both with the same options)");
  for (size_t i = 0; i < wa.size(); ++i) if (wa[i].call0 != wb[i].call0 || wa[i].call1 != wb[i].call1) return fail(GSV_ERR_INVALID, "garbler and evaluator sessions have different schedules (create both with the same options)");
  DrainSink k;
  k.hashes = hashes;
  int rc = garble_streaming_range(gs, gate_id_base, 0, gs->plan->calls.size(), k, n_threads, es);
  if (rc == GSV_OK) { gs->next_call = gs->plan->calls.size(); gs->garbled = false; }
  return rc;
}
int gsv_plan_call_info(const gsv_plan* p, uint64_t call, uint64_t* gate_offset, uint64_t* n_gates, uint64_t* ct_offset, uint64_t* n_ciphertexts, uint64_t* n_steps) {
  if (!p || call >= p->calls.size()) return fail(GSV_ERR_INVALID, "null plan / call index out of range");
  const PlanCall& c = p->calls[size_t(call)];
  if (gate_offset) *gate_offset = c.gid_off;
  if (n_gates) *n_gates = c.prog->prog.n_gates;
  if (ct_offset) *ct_offset = c.ct_off;
  if (n_ciphertexts) *n_ciphertexts = c.prog->prog.n_ct;
  if (n_steps) *n_steps = c.prog->prog.n_steps;
  return GSV_OK;
}
int gsv_plan_call_record_form(const gsv_plan* p, uint64_t call, uint32_t* and_terms) {
  if (!p || !and_terms || call >= p->calls.size()) return fail(GSV_ERR_INVALID, "null argument / call index out of range");
  { int rc = program_ready(p->calls[size_t(call)].prog); if (rc) return rc; }
  *and_terms = p->calls[size_t(call)].prog->prog.and_terms;
  return GSV_OK;
}
int gsv_session_evaluate(gsv_session* s, uint64_t gate_id_base) {
  if (!s) return fail(GSV_ERR_INVALID, "null session");
  // EvaluateMode panics with "Ciphertext source exhausted at gate .." when the source runs dry (evaluate_mode.rs:139-142).
  const uint64_t need = s->prog().n_ct * s->replays;
  if (s->ct_cap != s->replays || (s->plan && !s->plan_retain)) return fail(GSV_ERR_INVALID, "evaluate needs the whole ciphertext stream resident (ct_capacity_replays == replays)");
  if (!s->garbled)
    for (size_t i = 0; i < s->n_inst; ++i)
      if (s->ct_uploaded[i] < need)
        return fail(GSV_ERR_EXHAUSTED, "Ciphertext source exhausted: instance " + std::to_string(i) + " holds " + std::to_string(s->ct_uploaded[i]) + " of " + std::to_string(need) + " ciphertexts");
  return launch(s, gate_id_base, true);
}

// Evaluate with the ciphertexts coming from a CiphertextSource (ciphertext_source.rs:14-107), segment by segment: program sessions one
// ring at a time, plan sessions one window of the schedule at a time.  The records arrive in gate order in bounded chunks (a
// page-locked 16 MiB staging buffer: a window may be gigabytes), are folded into the per-instance CBC-MAC as FileSource does while
// reading (ciphertext_source.rs:36-107), uploaded, scattered to the program-order positions the kernel reads, and evaluated.
//   read(instance, first_record, dst, n) -> 0, or non-zero when the source runs dry ("Ciphertext source exhausted", evaluate_mode.rs:139-142)
static int evaluate_streaming_impl(gsv_session* s, uint64_t gate_id_base, const std::function<int(size_t, uint64_t, uint8_t*, uint64_t)>& read, uint8_t* hashes) {
  const Program& g = s->prog();
  // plan sessions: one window of the schedule per launch, its ciphertexts uploaded SEGMENT by segment (schedule.hpp: a gate-order buffer
  // holds the largest segment, the program-order device block the largest window); program sessions: one ring per launch
  const size_t n_inst = s->n_inst;
  HIPCHK(hipSetDevice(s->e->device));
  const uint64_t seg_records = s->plan ? s->plan_max_segment : s->ct_cap * g.n_ct;  // per instance: stride of the gate-order buffer
  if (seg_records) { int grc = ensure_ct_gate(s, n_inst * size_t(seg_records) * 16); if (grc) return grc; }  // (a sample drain before may have sized it for fewer instances)
  // The CBC-MAC of one instance is a serial chain (ciphertext_source.rs:36-107 folds it while reading), the chains of different instances
  // are independent: with hashes asked for, the chunks are read CHUNK-major (every instance's chunk at one offset, then the next offset)
  // and instance i's chunks are folded in order by worker i mod T, beside the uploads; a staging buffer is reused once its copy AND its MAC
  // are done.  (The reference's evaluator runs its finalized cases under into_par_iter: cut_and_choose/evaluator.rs:118-181.)
  const size_t T = hashes ? std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n_inst, 16), gsv_drain::usable_cores() > 1 ? gsv_drain::usable_cores() - 1 : 1)) : 0;
  const size_t NB = 2 + 2 * T;
  struct Pinned { std::vector<void*> p; std::vector<hipEvent_t> ev; ~Pinned() { for (void* q : p) if (q) (void)hipHostFree(q); for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); } } stage;
  const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(seg_records, 1), CT_STAGE_RECORDS);
  stage.p.assign(NB, nullptr); stage.ev.assign(NB, nullptr);
  for (size_t k = 0; k < NB; ++k) { HIPCHK(hipHostMalloc(&stage.p[k], size_t(chunk) * 16, hipHostMallocDefault)); HIPCHK(hipEventCreateWithFlags(&stage.ev[k], hipEventDisableTiming)); }
  std::vector<CbcMacHost> macs(n_inst);
  struct MacPool {  // declared after `stage` and `macs`: joined before either goes away
    struct Job { size_t inst; const uint8_t* p; uint64_t n; size_t buf; };
    std::vector<CbcMacHost>& macs;
    std::vector<std::deque<Job>> q;
    std::vector<char> busy;  // per staging buffer: a MAC job still reads it
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_free;
    bool closed = false;
    MacPool(std::vector<CbcMacHost>& m, size_t T, size_t NB) : macs(m), q(T), busy(NB, 0) {
      for (size_t t = 0; t < T; ++t) th.emplace_back([this, t] {
        for (;;) {
          Job j;
          { std::unique_lock<std::mutex> lk(mu); cv_job.wait(lk, [&] { return closed || !q[t].empty(); }); if (q[t].empty()) return; j = q[t].front(); q[t].pop_front(); }
          macs[j.inst].update(j.p, j.n);
          { std::lock_guard<std::mutex> lk(mu); busy[j.buf] = 0; }
          cv_free.notify_all();
        }
      });
    }
    void push(size_t inst, const uint8_t* p, uint64_t n, size_t buf) {
      { std::lock_guard<std::mutex> lk(mu); busy[buf] = 1; q[inst % q.size()].push_back(Job{inst, p, n, buf}); }
      cv_job.notify_all();
    }
    void wait_free(size_t buf) { std::unique_lock<std::mutex> lk(mu); cv_free.wait(lk, [&] { return !busy[buf]; }); }
    void finish() {  // every queued chunk folded, workers gone
      { std::lock_guard<std::mutex> lk(mu); closed = true; }
      cv_job.notify_all();
      for (std::thread& t : th) t.join();
      th.clear();
    }
    ~MacPool() { finish(); }
  } pool(macs, T, NB);
  if (s->plan) HIPCHK(hipMemsetAsync(s->d_error, 0, 4, s->e->stream));
  HIPCHK(hipEventRecord(s->ev0, s->e->stream));
  int rc = GSV_OK;
  size_t b = 0;
  // `n` records per instance starting at stream index `base` -> the gate-order buffer through `st` (bounded page-locked chunks, hashed as read)
  auto upload = [&](uint64_t base, uint64_t n, hipStream_t st) -> int {
    for (uint64_t off = 0; off < n; off += chunk)
      for (size_t i = 0; i < n_inst; ++i, b = (b + 1) % NB) {
        const uint64_t m = std::min(chunk, n - off);
        if (hipEventSynchronize(stage.ev[b]) != hipSuccess) return fail(GSV_ERR_DEVICE, "event wait failed");  // the copy that last used this staging buffer has finished
        if (T) pool.wait_free(b);  // ... and so has its MAC
        uint8_t* host = static_cast<uint8_t*>(stage.p[b]);
        if (read(i, base + off, host, m) != 0) return fail(GSV_ERR_EXHAUSTED, "Ciphertext source exhausted: instance " + std::to_string(i) + " ran dry at record " + std::to_string(base + off));
        if (T) pool.push(i, host, m, b);
        if (hipMemcpyAsync(static_cast<uint8_t*>(s->ct_gate) + (i * seg_records + off) * 16, host, m * 16, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(stage.ev[b], st) != hipSuccess)
          return fail(GSV_ERR_DEVICE, "ciphertext upload failed");
      }
    return GSV_OK;
  };
  if (s->plan && s->ct_ring) {
    // Ring mode: the window (the whole pass) is launched FIRST; its calls wait on the device until the host's position counter says their
    // segment has been uploaded.  Segment after segment: wait until the calls whose blocks this segment's blocks overwrite have
    // completed (their flags), upload and scatter on the side stream, publish the segment's end.
    rc = ensure_aux(s);
    for (size_t w = 0; w < s->sched.windows.size() && rc == GSV_OK; ++w) {
      const Schedule::Window& win = s->sched.windows[w];
      __atomic_store_n(s->host_ct_pos, (unsigned long long)win.ct0, __ATOMIC_RELEASE);
      rc = launch_plan_window(s, w, gate_id_base, true);
      bool window_done = false;
      for (uint32_t q = win.seg0; q < win.seg1 && rc == GSV_OK; ++q) {
        const Schedule::Segment& sg = s->sched.segments[q];
        uint32_t o0 = ~0u, o1 = 0;
        for (uint32_t k = sg.call0; k < sg.call1; ++k) if (s->sched.ovl1[k] > s->sched.ovl0[k]) { o0 = std::min(o0, s->sched.ovl0[k]); o1 = std::max(o1, s->sched.ovl1[k]); }
        // [ovl0, ovl1) is a RANGE around the overwritten calls: the calls it spans beside them are earlier calls too, but those of this
        // very segment cannot run before this upload — and are never among the overwritten ones (the ring holds two segments and a call)
        o1 = std::min(o1, sg.call0);
        if (o1 > o0) rc = wait_calls_done(s, w, o0, o1, &window_done);
        if (rc != GSV_OK) break;
        if (window_done) { rc = fail(GSV_ERR_DEVICE, "internal: the window finished before its ciphertexts were uploaded"); break; }
        // uploads and the scatter go through the side stream (the main stream holds the running window)
        rc = upload(sg.ct0, sg.n_ct, s->aux_stream);
        if (rc == GSV_OK) rc = permute_plan_calls(s, w, sg.call0, sg.call1, sg.ct0, seg_records, 1, nullptr, nullptr, s->aux_stream);
        if (rc == GSV_OK && hipStreamSynchronize(s->aux_stream) != hipSuccess) rc = fail(GSV_ERR_DEVICE, "ciphertext scatter failed");
        if (rc == GSV_OK) __atomic_store_n(s->host_ct_pos, (unsigned long long)(sg.ct0 + sg.n_ct), __ATOMIC_RELEASE);
      }
      if (rc != GSV_OK) {
        // let the calls that still wait for ciphertexts run out (their results are discarded with the error) instead of hanging the stream
        __atomic_store_n(s->host_ct_pos, ~0ull, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(s->e->stream);
        break;
      }
      if (hipStreamSynchronize(s->e->stream) != hipSuccess) rc = fail(GSV_ERR_DEVICE, "kernel failed");
    }
  } else if (s->plan) {
    for (size_t w = 0; w < s->sched.windows.size() && rc == GSV_OK; ++w) {
      const Schedule::Window& win = s->sched.windows[w];
      for (uint32_t q = win.seg0; q < win.seg1 && rc == GSV_OK; ++q) {
        const Schedule::Segment& sg = s->sched.segments[q];
        // (the stream orders this segment's uploads behind the scatter of the previous one, which read the same buffer)
        rc = upload(sg.ct0, sg.n_ct, s->e->stream);
        if (rc == GSV_OK) rc = permute_plan_calls(s, w, sg.call0, sg.call1, sg.ct0, seg_records, 1, nullptr, nullptr, nullptr);
      }
      if (rc == GSV_OK) rc = launch_plan_window(s, w, gate_id_base, true);
    }
  } else {
    const uint64_t n_ct = g.n_ct, total = s->replays, seg = s->ct_cap;
    for (uint64_t r0 = 0; r0 < total && rc == GSV_OK; r0 += seg) {
      const uint64_t r1 = std::min(total, r0 + seg);
      rc = upload(r0 * n_ct, (r1 - r0) * n_ct, s->e->stream);
      if (rc != GSV_OK) break;
      if (gsvk_gather_segment(s->CT, s->ct_stride(), s->dp.ct_pos, n_ct, uint32_t(r1 - r0), uint32_t(n_inst), s->ct_gate, seg_records, 1, s->e->stream) != 0) { rc = fail(GSV_ERR_DEVICE, "ciphertext scatter launch failed"); break; }
      rc = launch(s, gate_id_base, true, r0, r1 - r0);
    }
  }
  if (hipStreamSynchronize(s->e->stream) != hipSuccess && rc == GSV_OK) rc = fail(GSV_ERR_DEVICE, "kernel failed");
  if (rc != GSV_OK) return rc;
  if (s->plan) { HIPCHK(hipEventRecord(s->ev1, s->e->stream)); rc = gather_plan_outputs(s, true); if (rc) return rc; HIPCHK(hipStreamSynchronize(s->e->stream)); rc = check_plan_error(s); if (rc) return rc; }
  pool.finish();
  if (hashes) for (size_t i = 0; i < n_inst; ++i) macs[i].digest(hashes + 16 * i);
  return GSV_OK;
}
// FileSource: instance i reads <dir>/gc_<indexes[i]>.bin (indexes == NULL: first_index + i)
static int evaluate_from_files(gsv_session* s, uint64_t gate_id_base, const char* dir, const uint64_t* indexes, uint64_t first_index, uint8_t* hashes) {
  if (!s || !dir) return fail(GSV_ERR_INVALID, "null argument");
  std::vector<FILE*> files(s->n_inst, nullptr);
  struct Closer { std::vector<FILE*>& f; ~Closer() { for (FILE*& q : f) if (q) { std::fclose(q); q = nullptr; } } } closer{files};
  for (size_t i = 0; i < s->n_inst; ++i) {
    const std::string path = std::string(dir) + "/gc_" + std::to_string(indexes ? indexes[i] : first_index + i) + ".bin";
    files[i] = std::fopen(path.c_str(), "rb");
    if (!files[i]) return fail(GSV_ERR_INVALID, "cannot open " + path);
  }
  // the reads of one instance are sequential in the stream, but the instances alternate: seek when the position is not the expected one
  std::vector<uint64_t> pos(s->n_inst, 0);
  return evaluate_streaming_impl(s, gate_id_base, [&](size_t i, uint64_t first, uint8_t* dst, uint64_t n) -> int {
    if (pos[i] != first) { if (fseeko(files[i], off_t(first * 16), SEEK_SET) != 0) return 1; pos[i] = first; }
    if (n && std::fread(dst, 16, n, files[i]) != n) return 1;
    pos[i] += n;
    return 0;
  }, hashes);
}
int gsv_session_evaluate_streaming(gsv_session* s, uint64_t gate_id_base, const char* dir, uint64_t first_index, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  return evaluate_from_files(s, gate_id_base, dir, nullptr, first_index, hashes);
}
int gsv_session_evaluate_streaming_indexed(gsv_session* s, uint64_t gate_id_base, const char* dir, const uint64_t* indexes, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  if (!indexes) return fail(GSV_ERR_INVALID, "null index list");
  return evaluate_from_files(s, gate_id_base, dir, indexes, 0, hashes);
}
int gsv_session_evaluate_streaming_source(gsv_session* s, uint64_t gate_id_base, gsv_ct_source_fn source, void* user, uint8_t* hashes) {
  PassGuard pass_guard;  // destroys requested while this pass runs wait for its end (deferred release)
  if (!s || !source) return fail(GSV_ERR_INVALID, "null argument");
  return evaluate_streaming_impl(s, gate_id_base, [&](size_t i, uint64_t first, uint8_t* dst, uint64_t n) -> int { return source(user, i, first, dst, n); }, hashes);
}

int gsv_session_set_hasher(gsv_session* s, int kind) {
  if (!s || (kind != GSV_HASHER_AES && kind != GSV_HASHER_BLAKE3)) return fail(GSV_ERR_INVALID, "unknown hasher");
  s->hasher = kind;
  return GSV_OK;
}

int gsv_session_sync(gsv_session* s) {
  if (!s) return fail(GSV_ERR_INVALID, "null session");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipStreamSynchronize(s->e->stream));
  if (s->plan) return check_plan_error(s);
  return GSV_OK;
}
// Diagnostics: per-step wall-clock stamps (100 MHz) of instance 0's workgroup during the last replay of a launch.
int gsv_session_enable_step_clock(gsv_session* s) {
  if (!s) return fail(GSV_ERR_INVALID, "null session");
  HIPCHK(hipSetDevice(s->e->device));
  if (!s->step_clock) {
    const size_t bytes = (size_t(s->prog().n_steps) + 1) * sizeof(uint64_t);
    HIPCHK(hipMalloc(&s->step_clock, bytes));
    HIPCHK(hipMemset(s->step_clock, 0, bytes));
  }
  return GSV_OK;
}
int gsv_session_read_step_clock(gsv_session* s, uint64_t* out) {
  if (!s || !out || !s->step_clock || !s->ran) return fail(GSV_ERR_INVALID, "step clock not enabled / nothing ran");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipStreamSynchronize(s->e->stream));
  HIPCHK(hipMemcpy(out, s->step_clock, (size_t(s->prog().n_steps) + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return GSV_OK;
}
// Diagnostics: per step {and_cnt, xor_cnt, lds_reads, hbm_reads, lds_writes, hbm_writes} decoded from the compiled records.
int gsv_program_step_stats(const gsv_program* p, uint32_t* out6) {
  if (!p || !out6) return fail(GSV_ERR_INVALID, "null argument");
  { int rc = program_ready(p); if (rc) return rc; }
  const Program& g = p->prog;
  for (size_t s = 0; s < g.steps.size(); ++s) {
    const StepDesc& d = g.steps[s];
    uint32_t* o = out6 + 6 * s;
    o[0] = d.and_cnt; o[1] = d.xor_cnt; o[2] = o[3] = o[4] = o[5] = 0;
    auto rd = [&](uint32_t sl) { if (sl != SLOT_LDS_ZERO) o[(sl & SLOT_LDS_FLAG) ? 2 : 3]++; };
    auto wr = [&](uint32_t sl) { o[(sl & SLOT_LDS_FLAG) ? 4 : 5]++; };
    for (uint32_t k = 0; k < d.and_cnt; ++k) {
      const AndRec& r = g.ands[d.and_off + k];
      rd(uint32_t(r.w0) & SLOT_MASK); rd(uint32_t(r.w0 >> 21) & SLOT_MASK); rd(uint32_t(r.w0 >> 42) & SLOT_MASK);
      rd(uint32_t(r.w1) & SLOT_MASK); rd(uint32_t(r.w1 >> 21) & SLOT_MASK);
      if (g.and_terms == 4) { rd(uint32_t(r.w1 >> 42) & SLOT_MASK); rd(uint32_t(r.w2) & SLOT_MASK); rd(uint32_t(r.w2 >> 21) & SLOT_MASK); rd(uint32_t(r.w2 >> 42) & SLOT_MASK); wr(uint32_t(r.w3) & SLOT_MASK); }
      else wr(uint32_t(r.w1 >> 42) & SLOT_MASK);
    }
    for (uint32_t k = 0; k < d.xor_cnt; ++k) {
      const XorRec& r = g.xors[d.xor_off + k];
      rd(uint32_t(r.w0) & SLOT_MASK); rd(uint32_t(r.w0 >> 21) & SLOT_MASK); rd(uint32_t(r.w0 >> 42) & SLOT_MASK);
      rd(uint32_t(r.w1) & SLOT_MASK); wr(uint32_t(r.w1 >> 21) & SLOT_MASK);
    }
  }
  return GSV_OK;
}
int gsv_session_instances_per_workgroup(const gsv_session* s, int* n) {
  if (!s || !n) return fail(GSV_ERR_INVALID, "null argument");
  *n = int(s->ni);
  return GSV_OK;
}
int gsv_session_last_kernel_ms(gsv_session* s, double* ms) {
  if (!s || !ms || !s->ran) return fail(GSV_ERR_INVALID, "no launch recorded");
  HIPCHK(hipEventSynchronize(s->ev1));
  float f = 0;
  HIPCHK(hipEventElapsedTime(&f, s->ev0, s->ev1));
  *ms = f;
  return GSV_OK;
}
int gsv_session_read_outputs(gsv_session* s, uint8_t* labels, uint8_t* bits) {
  if (!s || !labels || !s->ran) return fail(GSV_ERR_INVALID, "bad argument / nothing ran");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipStreamSynchronize(s->e->stream));
  if (s->plan) { int rc = check_plan_error(s); if (rc) return rc; }
  const size_t n = s->n_inst * s->prog().output_slots.size();
  if (n) HIPCHK(hipMemcpy(labels, s->out, n * 16, hipMemcpyDeviceToHost));
  if (bits) {
    if (!s->last_eval) return fail(GSV_ERR_INVALID, "plaintext bits exist only after evaluate");
    if (n) HIPCHK(hipMemcpy(bits, s->out_bits, n, hipMemcpyDeviceToHost));
  }
  return GSV_OK;
}
int gsv_session_read_ciphertexts(gsv_session* s, size_t instance, uint64_t first, uint64_t n_records, uint8_t* out) {
  if (!s || instance >= s->n_inst || (!out && n_records)) return fail(GSV_ERR_INVALID, "bad argument");
  if (s->plan && !s->plan_retain) return fail(GSV_ERR_INVALID, "this plan session does not retain the ciphertext stream");
  if (first + n_records > s->ct_stride()) return fail(GSV_ERR_INVALID, "range exceeds the retained ciphertext stream");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipStreamSynchronize(s->e->stream));
  for (uint64_t off = 0; off < n_records; off += CT_STAGE_RECORDS) {
    const uint64_t n = std::min<uint64_t>(CT_STAGE_RECORDS, n_records - off);
    int rc = fetch_ciphertexts(s, instance, first + off, n, out + off * 16);
    if (rc) return rc;
  }
  return GSV_OK;
}
int gsv_session_ciphertext_hash(gsv_session* s, size_t instance, uint8_t hash[16]) {
  if (!s || instance >= s->n_inst || !hash) return fail(GSV_ERR_INVALID, "bad argument");
  if (s->ct_cap != s->replays || (s->plan && !s->plan_retain)) return fail(GSV_ERR_INVALID, "the session retains only part of the stream (ct_capacity_replays < replays)");
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipStreamSynchronize(s->e->stream));
  const uint64_t total = s->ct_stride();
  const uint64_t chunk = CT_STAGE_RECORDS;
  std::vector<uint8_t> buf(size_t(std::min<uint64_t>(chunk, total ? total : 1)) * 16);
  CbcMacHost mac;
  for (uint64_t off = 0; off < total; off += chunk) {
    uint64_t n = std::min(chunk, total - off);
    int rc = fetch_ciphertexts(s, instance, off, n, buf.data());
    if (rc) return rc;
    mac.update(buf.data(), n);
  }
  mac.digest(hash);
  return GSV_OK;
}
int gsv_cbcmac_update(uint8_t state[16], const uint8_t* cts, uint64_t n_records) {
  if (!state || (!cts && n_records)) return fail(GSV_ERR_INVALID, "null argument");
  // CbcMacHost starts from zero; chain by XOR-ing the state into the first block (h ^ ct).
  CbcMacHost mac;
  if (n_records == 0) return GSV_OK;
  uint8_t first[16];
  for (int i = 0; i < 16; ++i) first[i] = cts[i] ^ state[i];
  mac.update(first, 1);
  mac.update(cts + 16, n_records - 1);
  mac.digest(state);
  return GSV_OK;
}
int gsv_cbcmac_chains_per_step(void) { return CbcMacHost::have_vaes() ? 16 : GSV_HOST_AESNI ? 4 : 1; }
int gsv_cbcmac_update_many(uint8_t* states, const uint8_t* const* cts, size_t n_chains, uint64_t n_records) {
  if ((!states || !cts) && n_chains) return fail(GSV_ERR_INVALID, "null argument");
  for (size_t i = 0; i < n_chains; ++i) if (!cts[i] && n_records) return fail(GSV_ERR_INVALID, "null stream");
  // CbcMacHost starts from zero: chain by XOR-ing the state into a copy of the first block (h ^ ct), as gsv_cbcmac_update does
  if (n_records == 0) return GSV_OK;
  std::vector<CbcMacHost> macs(n_chains);
  for (size_t i = 0; i < n_chains; ++i) {
    uint8_t first[16];
    for (int k = 0; k < 16; ++k) first[k] = cts[i][k] ^ states[16 * i + k];
    macs[i].update(first, 1);
  }
  std::vector<CbcMacHost*> mp(n_chains);
  std::vector<const uint8_t*> cp(n_chains);
  for (size_t i = 0; i < n_chains; ++i) { mp[i] = &macs[i]; cp[i] = cts[i] + 16; }
  CbcMacHost::update_many(mp.data(), cp.data(), n_chains, n_records - 1);  // sixteen chains per step with VAES, else four
  for (size_t k = 0; k < n_chains; ++k) macs[k].digest(states + 16 * k);
  return GSV_OK;
}
int gsv_commit_labels(const uint8_t* labels, uint64_t n, uint8_t* out) {
  if ((!labels || !out) && n) return fail(GSV_ERR_INVALID, "null argument");
  for (uint64_t i = 0; i < n; ++i) {  // AES_K(label): one-block CBC-MAC from zero state
    CbcMacHost mac;
    mac.update(labels + 16 * i, 1);
    mac.digest(out + 16 * i);
  }
  return GSV_OK;
}

}  // extern "C"
