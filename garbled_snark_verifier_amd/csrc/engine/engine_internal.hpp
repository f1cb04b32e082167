// Types and helpers shared by the parts of the host runtime (engine.cpp is ONE translation unit; its parts are the engine_*.ipp files,
// included in order inside its extern "C" block): error reporting, device allocation, the deferred-release gate, and the objects behind the
// opaque handles of include/gsv_engine.h.
#pragma once
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
