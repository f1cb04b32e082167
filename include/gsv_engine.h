/* gsv_engine.h — C ABI of the MI355X (gfx950) garbling / evaluation engine.
 *
 * This is the drop-in boundary for the reference's hot path (BitVM/garbled-snark-verifier v0.4.0).
 * The reference has no FFI; its seam is the Rust trait trio
 *     CircuitMode        src/circuit/modes.rs:26-51
 *     CiphertextHandler  src/circuit/mod.rs:140-178
 *     CiphertextSource   src/circuit/ciphertext_source.rs:14-21
 * A `GpuGarbleMode` / `GpuEvaluateMode` (`impl CircuitMode`) placed next to garble_mode.rs forwards
 * to the functions below (binding shown in INTEGRATION.md).  Plain pointers and sizes only; every
 * function returns 0 on success and a non-zero gsv_status otherwise (the C side never unwinds; the
 * Rust shim turns non-zero into panic! to keep the reference's error behaviour).  All label buffers
 * are arrays of 16-byte records in `S::to_bytes()` order (big-endian u128, src/core/s.rs:25-31) —
 * the same bytes the reference writes to gc_{i}.bin (src/cut_and_choose/ciphertext_repository.rs:94-106).
 *
 * Object model
 *   gsv_recorder  one per circuit: records the gate stream that `CircuitMode::evaluate_gate`
 *                 receives (or a built-in restated gadget circuit) — replaces GarbleMode's per-gate
 *                 loop body (garble_mode.rs:160-222) with "enqueue".
 *   gsv_program   the recorded stream compiled into dependency-levelled device steps; shared by all
 *                 instances of a cut-and-choose run (cut_and_choose/garbler.rs:206-234 garbles the SAME
 *                 circuit once per seed).
 *   gsv_engine    one per GPU / per host thread; owns the HIP stream, device tables and buffers.
 *   gsv_session   one batch of instances garbled or evaluated on a program: wire files, ciphertext
 *                 streams, outputs.
 */
#ifndef GSV_ENGINE_H
#define GSV_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gsv_status {
  GSV_OK = 0,
  GSV_ERR_INVALID = 1,   /* bad argument / state */
  GSV_ERR_CIRCUIT = 2,   /* what the reference would panic! on (missing wire, arity mismatch, ...) */
  GSV_ERR_DEVICE = 3,    /* HIP failure or no gfx950 device: the engine has NO CPU fallback */
  GSV_ERR_EXHAUSTED = 4  /* ciphertext source exhausted (evaluate_mode.rs:141) */
} gsv_status;

/* Gate record = reference `Gate` (src/core/gate.rs:7-12) with u64 wire ids; `gate_type` is the
 * `GateType` repr(C) discriminant (src/core/gate_type.rs:3-15): And 0, Nand 1, Nimp 2, Imp 3,
 * Ncimp 4, Cimp 5, Nor 6, Or 7, Xor 8, Xnor 9, Not 10. */
typedef struct gsv_gate {
  uint64_t wire_a, wire_b, wire_c; /* wire_c == GSV_WIRE_UNREACHABLE: dead gate (gate_id still consumed) */
  uint8_t gate_type;
  uint8_t pad[7];
} gsv_gate;
#define GSV_WIRE_FALSE 0ull             /* circuit_context_trait.rs:2 */
#define GSV_WIRE_TRUE 1ull              /* circuit_context_trait.rs:3 */
#define GSV_WIRE_UNREACHABLE (~0ull)    /* src/core/wire.rs:8 */

typedef struct gsv_recorder gsv_recorder;
typedef struct gsv_program gsv_program;
typedef struct gsv_engine gsv_engine;
typedef struct gsv_session gsv_session;

const char* gsv_last_error(void); /* thread-local message for the last non-zero status */

/* ---- recording: the CircuitMode seam ------------------------------------------------------- */
int gsv_recorder_create(gsv_recorder** out);
void gsv_recorder_destroy(gsv_recorder* r);
/* CircuitMode::allocate_wire (modes.rs:38; storage.rs:119-133): credits == 0 -> GSV_WIRE_UNREACHABLE. */
int gsv_recorder_allocate_wire(gsv_recorder* r, uint16_t credits, uint64_t* wire_out);
/* The same for n wires with non-zero credits at once: they get consecutive ids from *first_wire_out (a host that asks per wire pays a
 * call per wire of the circuit; wires with zero credits are GSV_WIRE_UNREACHABLE and never reach the recorder). */
int gsv_recorder_allocate_wires(gsv_recorder* r, size_t n, uint64_t* first_wire_out);
/* CircuitMode::feed_wire for a root input (EncodeInput::encode): declares `wire` as the next circuit input. */
int gsv_recorder_declare_input(gsv_recorder* r, uint64_t wire);
/* CircuitMode::evaluate_gate (modes.rs:36), batched: n gates in stream order. */
int gsv_recorder_push_gates(gsv_recorder* r, const gsv_gate* gates, size_t n);
/* CircuitOutput::decode wire list (circuit/mod.rs:283): declares the circuit's output wires, in order. */
int gsv_recorder_declare_outputs(gsv_recorder* r, const uint64_t* wires, size_t n);
/* Record one of the built-in restated circuits (host mirror of src/gadgets, driven through the
 * two-pass credit driver of src/circuit/streaming_mode.rs) e.g. "u254_add", "fq_mul", "fq12_mul". */
int gsv_recorder_record_circuit(gsv_recorder* r, const char* circuit_spec);
/* What has been recorded so far: declared inputs / outputs and gates pushed (dead ones included). */
int gsv_recorder_counts(const gsv_recorder* r, uint64_t* n_inputs, uint64_t* n_outputs, uint64_t* n_gates);

/* ---- compile -------------------------------------------------------------------------------- */
/* feedback pairs (output index -> input index) are copied at the end of every replay, so that a
 * program garbled with `replays = K` equals K chained calls of the component (gate ids and the
 * ciphertext stream continue across replays). n_feedback may be 0. */
int gsv_program_compile(gsv_recorder* r, const uint32_t* fb_out_idx, const uint32_t* fb_in_idx, size_t n_feedback, gsv_program** out);
/* Compile with options (no feedback) — what a host that builds a PLAN of many unit programs needs (gsv_plan_recorder_* below; the
 * reference-side hook is StreamingMode::with_named_child, streaming_mode.rs:189-241):
 *   struct_size       sizeof(gsv_compile_opts) (a mismatch is GSV_ERR_INVALID)
 *   window_div        0 / 1: compiled for the full LDS label window (other layouts are compiled on first use from the kept trace);
 *                     2 / 4: ONE image for half / a quarter of the window that serves sessions with up to that many instances per workgroup
 *   keep_trace        0: the recorded trace is released once the image exists (no further variant can be compiled)
 *   background        1: the call returns at once and the compilation runs on the library's worker pool (GSV_COMPILE_THREADS, default: the
 *                     hardware's threads, at most 16; the call blocks while as many jobs as workers are queued).  The handle can be passed to
 *                     gsv_plan_recorder_call right away; everything that needs the image waits for it.  gsv_program_wait returns its status.
 *   consume_recorder  1: the trace is moved out of the recorder instead of copied (13 bytes per gate); the recorder is empty afterwards
 *   for_plan          the plan recorder the program is a unit of: window_div is the recorder's, and if the recorder writes a plan file the
 *                     records are appended to it the moment the image exists and dropped from memory.  The recorder must outlive the
 *                     compilation (gsv_plan_recorder_finish and _destroy wait for it). */
typedef struct gsv_plan_recorder gsv_plan_recorder;
typedef struct gsv_compile_opts {
  uint32_t struct_size;
  uint32_t window_div;
  uint32_t keep_trace;
  uint32_t background;
  uint32_t consume_recorder;
  uint32_t reserved;
  gsv_plan_recorder* for_plan;
} gsv_compile_opts;
int gsv_program_compile_opts(gsv_recorder* r, const gsv_compile_opts* opts, gsv_program** out);
int gsv_program_wait(gsv_program* p);
void gsv_program_destroy(gsv_program* p);
typedef struct gsv_program_info {
  uint64_t n_inputs, n_outputs;
  uint64_t n_gates;        /* per replay, dead gates included (= GateCount total, gate_type.rs:123-148) */
  uint64_t n_ciphertexts;  /* per replay */
  uint64_t n_dead;
  uint64_t gate_count[11]; /* by GateType discriminant */
  uint64_t n_steps, and_depth, n_and_steps, max_step_width; /* device steps; AND-depth of the DAG; steps holding AES work */
  uint64_t n_slots, peak_live;
  uint64_t device_bytes;   /* size of the program image in HBM */
  uint64_t n_lds_slots;    /* entries of the per-workgroup LDS label window in use */
  uint64_t reads_lds, reads_hbm, writes_lds, writes_hbm; /* label accesses per replay, by location */
  uint64_t n_fused_free;   /* free-gate records left after gate fusion (device records per replay = n_ciphertexts + this) */
  uint64_t and_terms;      /* wires per AND input in the device records: 2, or 4 for latency-bound programs (fewer device steps) */
} gsv_program_info;
int gsv_program_get_info(const gsv_program* p, gsv_program_info* info);

/* ---- plans: component-level programs ----------------------------------------------------------------
 * NOTE for a Rust host: the 11 B-gate verifier cannot go through the flat recorder above (13 bytes of trace per gate); it runs as a plan,
 * and a plan is NOT a pure "one more CircuitMode": it needs a hook inside the reference's StreamingMode::with_named_child
 * (streaming_mode.rs:189-241; bindings/rust/streaming_mode_unit_hook.patch) through which the mode receives whole component calls, and one
 * extra #[component] around the ladder chunks of exp_by_constant_montgomery (fp254impl.rs:713-722).  Both are stream-neutral.
 * The reference instantiates a few component shapes thousands of times (with_named_child, streaming_mode.rs:150-247); the
 * 11 B-gate verifier cannot be recorded as one flat program.  A plan is a sequence of CALLS to compiled programs over one
 * wire file per instance: wires that cross calls are "global" wires (dense ids chosen by the caller; ids 0..n_inputs-1 are
 * the plan's inputs); a call copies its inputs in, runs (one kernel launch), copies its outputs out.  Gate ids and the
 * ciphertext stream continue from call to call, so a plan produces exactly the stream of the flat circuit
 * call_1 ; call_2 ; ...  Programs used as calls are compiled without feedback. */
typedef struct gsv_plan gsv_plan;
int gsv_plan_create(gsv_plan** out);
void gsv_plan_destroy(gsv_plan* p);
int gsv_plan_add_call(gsv_plan* p, const gsv_program* prog, const uint32_t* in_globals /* n_inputs of prog */, const uint32_t* out_globals /* n_outputs of prog */);
int gsv_plan_finish(gsv_plan* p, uint32_t n_inputs, const uint32_t* output_globals, size_t n_outputs);
int gsv_plan_counts(const gsv_plan* p, uint64_t* n_gates, uint64_t* n_ciphertexts, uint64_t* n_calls);
int gsv_plan_io(const gsv_plan* p, uint64_t* n_inputs, uint64_t* n_outputs);
/* Plan recorder: the same builder for a host that runs its OWN two-pass driver (a `GpuPlanMode: CircuitMode` on the Rust side,
 * INTEGRATION.md §5).  Gates outside unit components are pushed as they come (they are cut into de-duplicated glue programs);
 * a unit component is one gsv_plan_recorder_call with a program the host recorded and compiled from the component's body
 * (gsv_recorder_* + gsv_program_compile, outputs = the wires the body PRODUCES): in_wires name the parent's wires, out_wires
 * receive one fresh parent wire per program output.  finish computes the global wires and returns the plan (which owns the
 * glue programs; the unit programs stay the caller's and must outlive the plan). */
int gsv_plan_recorder_create(gsv_plan_recorder** out);
/* With options: window_div as in struct gsv_compile_opts — for the glue programs the recorder compiles itself and for every unit compiled with
 * for_plan = this recorder; plan_file (may be NULL): every program of the plan is appended to this file when it has been compiled and its
 * records are dropped — the host never holds the plan's images (the verifier: 41 GB).  gsv_plan_recorder_finish then completes the file
 * (atomically: temp file + rename) and returns a plan that holds metadata only (counts, calls; gsv_plan_counts / _call_info / _io work,
 * sessions do not): load the file with gsv_plan_load(path, engine).  The file equals the one gsv_plan_build_file writes for the same
 * circuit, units and window_div. */
typedef struct gsv_plan_recorder_opts {
  uint32_t struct_size;   /* sizeof(gsv_plan_recorder_opts) */
  uint32_t window_div;    /* 0 / 1, 2, 4 */
  const char* plan_file;
} gsv_plan_recorder_opts;
int gsv_plan_recorder_create_opts(const gsv_plan_recorder_opts* opts, gsv_plan_recorder** out);
void gsv_plan_recorder_destroy(gsv_plan_recorder* r);
int gsv_plan_recorder_allocate_wire(gsv_plan_recorder* r, uint16_t credits, uint64_t* wire_out);
int gsv_plan_recorder_allocate_wires(gsv_plan_recorder* r, size_t n, uint64_t* first_wire_out);  /* as gsv_recorder_allocate_wires */
int gsv_plan_recorder_declare_input(gsv_plan_recorder* r, uint64_t wire);
int gsv_plan_recorder_push_gates(gsv_plan_recorder* r, const gsv_gate* gates, size_t n);
int gsv_plan_recorder_call(gsv_plan_recorder* r, const gsv_program* program, const uint64_t* in_wires, uint64_t* out_wires);
int gsv_plan_recorder_finish(gsv_plan_recorder* r, const uint64_t* output_wires, size_t n_outputs, gsv_plan** out);

/* Per-call facts of a finished plan: gate ids / ciphertext records consumed by the calls before `call`, and the call's own
 * gate, ciphertext and device-step counts.  Any pointer may be NULL. */
int gsv_plan_call_info(const gsv_plan* p, uint64_t call, uint64_t* gate_offset, uint64_t* n_gates, uint64_t* ct_offset, uint64_t* n_ciphertexts, uint64_t* n_steps);
/* Record form of the call's program: 2 = up to two wires per AND input (throughput-bound programs), 4 = up to four (latency-bound programs:
 * fewer dependent steps).  A window launch that holds a four-wire program runs the FW instantiation of the kernel. */
int gsv_plan_call_record_form(const gsv_plan* p, uint64_t call, uint32_t* and_terms);
/* Wire file of a plan session, in 16-byte slots per instance: the global region (wires that cross calls; ids are recycled once
 * their last reader has run) behind the largest program's own slots. */
int gsv_plan_wire_file(const gsv_plan* p, uint64_t* n_global_wires, uint64_t* max_program_slots);
/* Bytes of compiled program records (what a session uploads to HBM once per GPU) and the number of distinct programs. */
int gsv_plan_image_bytes(const gsv_plan* p, uint64_t* bytes, uint64_t* n_programs);
/* Plan files.  Building the verifier's plan takes minutes of host time and tens of GB of host memory (hundreds of
 * constant-specialised programs): gsv_plan_save writes a finished plan (its compiled programs and calls) to `path` (atomically:
 * temp file + rename), gsv_plan_load reads one back.  With an engine the program records are streamed from the memory-mapped
 * file straight into that GPU's memory and the host keeps only the metadata sessions need (the ranks of a node share one file
 * through the page cache: rank 0 builds and saves, the others load); such a plan serves sessions on that engine only and cannot
 * be saved again.  With e == NULL the plan is a complete host copy.  The file is specific to the engine build that wrote it. */
int gsv_plan_save(const gsv_plan* p, const char* path);
int gsv_plan_load(const char* path, gsv_engine* e, gsv_plan** out);
/* gsv_plan_from_circuit + gsv_plan_save without ever holding the plan: each program is appended to the file by the worker that
 * compiled it and its records are released at once, so the build's host memory is the programs still being compiled (the verifier:
 * ~25 GB instead of ~54 GB).  One image per program: GSV_PLAN_WINDOW_DIV=2|4 as below, or 1 = the full window, for sessions with one
 * instance per workgroup only (small batches: 3 % faster steps).  Then gsv_plan_load(path, engine). */
int gsv_plan_build_file(const char* circuit_spec, const char* units_csv, const char* path);
/* TWO plan files from ONE build (round 6): plan A = (units_csv_a, 1 / window_div_a of the LDS label window), plan B likewise (window_div 1,
 * 2 or 4; units_csv_b NULL = plan A's units).  The units the two plans share are recorded ONCE — the verifier's 182 constant line
 * functions are 3.3 B of the 3.5 B gates a build records, whichever granularity the Fq12 arithmetic around them is cut at — and compiled
 * for both plans by the worker that takes them off the recorder; with different unit lists the second plan's driver walks the circuit
 * on a thread of its own over the same unit cache.  Recording is the critical path of a build, so a deployment that serves full batches
 * (Fq12-level units, window_div 4) and small ones (Fq6-level units, window_div 1) gets both plans for about the time of one.  Each
 * file is byte for byte what gsv_plan_build_file writes for its units and window_div. */
int gsv_plan_build_file_pair(const char* circuit_spec, const char* units_csv_a, const char* path_a, uint32_t window_div_a, const char* units_csv_b, const char* path_b, uint32_t window_div_b);

/* Call operands naming the constant wires instead of a global wire. */
#define GSV_PLAN_WIRE_FALSE 0xFFFFFFFEu
#define GSV_PLAN_WIRE_TRUE 0xFFFFFFFFu
/* Stand-alone harness: record one of the built-in restated circuits under the two-pass driver with the named components
 * (comma separated, e.g. "fq12::mul_montgomery,fq12::square_montgomery") turned into calls; each distinct (component key,
 * output liveness) pair is recorded and compiled once, the gates between units become glue programs.  The plan owns its
 * programs.  (A Rust host reaches the same through a with_named_child hook; see INTEGRATION.md.)
 * Environment: GSV_PLAN_WINDOW_DIV=2|4 compiles every program ONCE, for half / a quarter of the LDS label window — the one image
 * then serves sessions with up to that many instances per workgroup and the recorded traces are freed (the verifier plan: 50 GB
 * of host memory instead of 92); unset, programs are compiled for the full window and the other layouts on first use. */
int gsv_plan_from_circuit(const char* spec, const char* units_csv, gsv_plan** out);

/* ---- engine --------------------------------------------------------------------------------- */
int gsv_engine_create(int device, gsv_engine** out); /* fails with GSV_ERR_DEVICE if no HIP device */
void gsv_engine_destroy(gsv_engine* e);
/* Deferred release (round 6).  gsv_session_destroy, gsv_plan_destroy, gsv_program_destroy and gsv_engine_destroy may be called from ANY
 * thread at ANY time — including from a gsv_ct_sink_fn / gsv_ct_source_fn callback in the middle of a streaming pass, which is where a
 * Rust host's `Drop` or a garbage collector runs them (CircuitMode values are dropped wherever the host drops them, modes.rs:26-51).
 * Freeing device memory synchronises the device, and a pass over a ciphertext ring waits for the host: so while any streaming pass
 * (gsv_session_garble_streaming*, _sink, _garble_evaluate, gsv_session_evaluate_streaming*) is in flight in this process the destroy
 * calls only queue the request; the queue runs, in order, when the last pass in flight has ended.  The handle is invalid for the host as
 * soon as destroy returns, as always.  Returns how many requests have been deferred so far (tests, diagnostics). */
uint64_t gsv_deferred_release_count(void);

/* Seed -> labels exactly as GarbleMode::new + issue_garbled_wire draw them (garble_mode.rs:80-97,
 * 116-118): delta, false.label0, true.label0, then n_inputs input label0s.  Host-only helper for the
 * stand-alone harness; a Rust host passes its own labels. */
int gsv_labels_from_seed(uint64_t seed, size_t n_inputs, uint8_t delta[16], uint8_t false_label0[16], uint8_t true_label0[16], uint8_t* input_label0);

/* ---- sessions: one batch of instances on one program ---------------------------------------- */
/* ct_capacity_replays: how many replays' worth of ciphertexts each instance's device stream holds
 * (a ring indexed by replay); pass `replays` to keep the whole stream. */
int gsv_session_create(gsv_engine* e, const gsv_program* p, size_t n_instances, uint64_t replays, uint64_t ct_capacity_replays, gsv_session** out);
void gsv_session_destroy(gsv_session* s);
/* A session over a plan: same calls as a program session (set_*_inputs, garble, evaluate, read_outputs, read / upload
 * ciphertexts, ciphertext_hash; one pass, whole stream retained).  The plan and its programs must outlive the session. */
int gsv_session_create_plan(gsv_engine* e, const gsv_plan* plan, size_t n_instances, gsv_session** out);
/* retain_stream = 0: the device keeps ONE WINDOW of ciphertexts (consecutive calls of the schedule, gsv_plan_session_opts below; plans
 * of any length); such a session is driven by the streaming calls only (gsv_session_garble_streaming*, _sink, _garble_evaluate,
 * gsv_session_evaluate_streaming*): each window's block is drained / evaluated while the next window runs.  Passing NULL options to
 * gsv_session_create_plan_opts means retain_stream = 1. */
int gsv_session_create_plan_ex(gsv_engine* e, const gsv_plan* plan, size_t n_instances, int retain_stream, gsv_session** out);
/* Call-level concurrency.  The reference gets its width from `total` instances garbled side by side, one per core
 * (cut_and_choose/garbler.rs:206-234); with 1-16 instances on a 256-CU GPU the width has to come from INSIDE an instance.  A plan
 * session therefore executes a SCHEDULE of the plan: the calls are taken in windows of consecutive stream order; a window is ONE
 * launch (grid = instance groups x calls) in which every call waits for the calls it depends on through the global wires (RAW / WAW
 * / WAR on the global ids) and otherwise runs side by side with the others, each in a scratch region of its own inside the
 * instance's wire file.  Gate ids and ciphertext positions are those of the stream order, so the result is bit-identical to the
 * sequential run.
 *   retain_stream         as gsv_session_create_plan_ex; GSV_STREAM_RING (2): like 0, and the whole pass is ONE launch over a ciphertext
 *                         ring (see window_ct_records).  A pass that only garbles (no sink: output labels, device rates) never waits in
 *                         the ring and gets the single launch's overlap scope for any number of instances (sixteen instances of the
 *                         verifier: 17 windows -> 1)
 *   max_concurrent_calls  calls of one instance in flight: 0 = as many as give every CU a workgroup (GSV_PLAN_CONCURRENCY overrides);
 *                         1 = sequential (the stream order).
 *                         ASSUMPTION for values > 1: a workgroup that waits for a dependency only waits for workgroups with a smaller
 *                         linear index, which the hardware dispatches first (in-order workgroup dispatch — true of every AMD GPU to
 *                         date, but not an architectural guarantee).  A wait that sees no call of its instance group complete for
 *                         GSV_DEP_WAIT_SECONDS (default 60) sets an error flag instead of hanging.  A STREAMING pass that ends that
 *                         way (gsv_session_garble_streaming*, _sink, _garble_evaluate) switches the session, in place, to the SAFE
 *                         schedule — one call per launch in stream order, no dependency wait on the device, no ciphertext ring; same
 *                         images and allocations, the host's last inputs re-staged — and, when the engine alone has seen the pass's
 *                         results (discarded / CBC-MAC / gc files, the whole plan in one call), repeats the pass by itself and
 *                         returns its result (gsv_session_fallback_count tells).  A pass that fed a host callback or an evaluator
 *                         session, and a slice of a pass, still fail with GSV_ERR_DEVICE (the host has consumed an invalid prefix;
 *                         gc files are removed) but the session is on the safe schedule afterwards: repeating the pass from
 *                         gsv_session_set_garble_inputs succeeds.  The streaming evaluator likewise: a pass over gc files is repeated
 *                         by the engine, a pass over a host source fails once and succeeds when repeated from
 *                         gsv_session_set_evaluate_inputs.  gsv_session_garble / _evaluate (asynchronous, whole stream
 *                         retained) report the flag at the next gsv_session_sync / read_outputs as before.
 *                         max_concurrent_calls = 1 keeps every call dependent on its predecessor only.
 *   window_ct_records     ciphertext records per instance of one window = ONE launch = the device block of a session that does not
 *                         retain the stream; independent call chains only overlap inside a window, so windows want to be large;
 *                         0 = 40 % of the free device memory, at most 48 GB over all instances (one instance of the verifier: one window).
 *                         With retain_stream = GSV_STREAM_RING (or GSV_CT_RING=1 in the environment when a retain_stream = 0 session
 *                         is created), 0 instead means the whole pass as ONE
 *                         window over a ciphertext RING of three drain segments (sessions with max_concurrent_calls != 1 whose
 *                         stream is longer than the ring; 3 GB instead of 48 for one instance): a garbling call waits until what
 *                         its block of the ring held on the previous lap has been taken off the device, an evaluating call until its
 *                         segment has been uploaded — the host publishes its stream position in mapped host memory.  Same pass time
 *                         as large windows, so it is opt-in.  gsv_session_garble_evaluate pairs need plain windows
 *                         (an explicit value)
 *   max_scratch_slots     16-byte slots per instance for the ring the calls' scratch regions are carved from; 0 = chosen from the
 *                         free device memory
 *   max_window_calls      0 = 32768 (a launch holds at most 65535 calls)
 *   drain_segment_records ciphertext records per instance of a drain SEGMENT: the streaming calls follow the completion flags of a running
 *                         window and take its stream off the device segment by segment (consecutive calls, gate order), so that the host
 *                         side — copies, the serial CBC-MAC chains, files, a sink — works beside the window that is still being garbled;
 *                         0 = 64 M records (1 GB) per instance or less (three gate-order buffers within a tenth of the free memory) */
#define GSV_STREAM_RING 2
typedef struct gsv_plan_session_opts {
  int retain_stream;                        /* 0 = windows, 1 = the whole stream stays on the device, GSV_STREAM_RING = one launch over a ring */
  uint32_t max_concurrent_calls;
  uint64_t window_ct_records;
  uint64_t max_scratch_slots;
  uint32_t max_window_calls;
  uint32_t drain_segment_records;
} gsv_plan_session_opts;
int gsv_session_create_plan_opts(gsv_engine* e, const gsv_plan* plan, size_t n_instances, const gsv_plan_session_opts* opts, gsv_session** out);
typedef struct gsv_plan_schedule_info {
  uint64_t n_calls, n_windows, n_dependencies, max_width;  /* max_width: most calls in flight at once by the schedule's step model */
  uint64_t scratch_slots, wire_file_slots;  /* per instance: scratch area / whole wire file (scratch + global wires) */
  uint64_t window_ct_records;               /* largest window, ciphertext records per instance */
  uint64_t critical_steps, total_steps;     /* device steps: sum over batches of the longest call / sum over all calls */
  uint64_t n_segments, segment_ct_records;  /* drain segments of the whole schedule; largest segment, ciphertext records per instance */
  uint64_t ct_ring_records;                 /* 0, or the size of the device's ciphertext ring (records per instance): see window_ct_records */
} gsv_plan_schedule_info;
int gsv_session_plan_schedule_info(const gsv_session* s, gsv_plan_schedule_info* info);
/* How often this session has fallen back to the safe schedule (0 or 1: see max_concurrent_calls above). */
int gsv_session_fallback_count(const gsv_session* s, uint64_t* n);
/* Window `window` of the session's schedule: calls [first_call, first_call + n_calls) of the plan.  Slices handed to
 * gsv_session_garble_streaming_calls start and end on window boundaries. */
int gsv_session_plan_window(const gsv_session* s, uint64_t window, uint64_t* first_call, uint64_t* n_calls, uint64_t* max_width);

/* Garble (GarbleMode): per instance i: delta[16i..], const_label0 = {false.label0, true.label0}
 * (32 B per instance), input_label0 (n_inputs*16 B per instance).  Asynchronous on the engine stream. */
int gsv_session_set_garble_inputs(gsv_session* s, const uint8_t* delta, const uint8_t* const_label0, const uint8_t* input_label0);
int gsv_session_garble(gsv_session* s, uint64_t gate_id_base);
/* Evaluate (EvaluateMode): const_active = {false active label, true active label} (evaluate_mode.rs:70),
 * input_active labels + plaintext bits (evaluate_mode.rs:15-18).  Ciphertexts come from the session's
 * device stream: either uploaded with gsv_session_upload_ciphertexts (CiphertextSource / gc_{i}.bin
 * bytes) or left there by a previous gsv_session_garble on the same session. */
int gsv_session_set_evaluate_inputs(gsv_session* s, const uint8_t* const_active, const uint8_t* input_active, const uint8_t* input_bits);
int gsv_session_upload_ciphertexts(gsv_session* s, size_t instance, const uint8_t* cts, uint64_t n_records);
int gsv_session_evaluate(gsv_session* s, uint64_t gate_id_base);

/* Garble every replay AND consume the ciphertext stream while the GPU keeps garbling (CiphertextHandler at full size:
 * circuit/mod.rs:140-178; AESAccumulatingHash, ciphertext_hasher.rs:23-29; the gc_{i}.bin writer,
 * cut_and_choose/ciphertext_repository.rs:94-127).  The launch is cut into segments of one device ring
 * (ct_capacity_replays replays); each finished segment is brought into gate order on the device (second buffer of the
 * ring's size, allocated on first use), copied out while the next segment is garbled, folded into every instance's CBC-MAC
 * by n_threads host threads (0 = up to 32; every thread advances the chains of four instances side by side, one chain alone
 * being bound by the latency of its dependent AES rounds) and, if dir is
 * not NULL, appended to <dir>/gc_<first_index + instance>.bin.  hashes receives n_instances x 16 bytes.  Output labels
 * are read with gsv_session_read_outputs as after gsv_session_garble.  With hashes == NULL and dir == NULL the stream is
 * discarded (garbling only: output labels, device-rate measurements). */
int gsv_session_garble_streaming(gsv_session* s, uint64_t gate_id_base, const char* dir, uint64_t first_index, int n_threads, uint8_t* hashes);

/* The same for calls [first_call, first_call + n_calls) of a plan session only, so that a long plan can be garbled in slices:
 * gate ids, wires and (with hashing) the per-instance CBC-MAC states continue from the previous slice; first_call == 0 starts a
 * new pass (MACs from zero, gc files truncated; later slices append).  `hashes` receives the MAC states after this slice — the
 * commitments once the last slice has run.  Output labels are gathered by the slice that ends with the plan's last call. */
int gsv_session_garble_streaming_calls(gsv_session* s, uint64_t gate_id_base, uint64_t first_call, uint64_t n_calls, const char* dir, uint64_t first_index, int n_threads, uint8_t* hashes);
/* A slice must start a new pass (first_call == 0) or continue where the previous slice ended, and start / end on window boundaries
 * of the session's schedule; anything else is GSV_ERR_INVALID.  Timing harnesses that garble slices out of order (stale wires,
 * meaningless MACs) say so explicitly: */
int gsv_session_set_unchecked_slices(gsv_session* s, int on);
/* Drain a SAMPLE of a large batch: the streaming calls (gsv_session_garble_streaming*, _sink) garble every instance but only the streams of
 * the first n instances leave the device (hashes then receives n x 16 bytes, gc files / the sink see instances 0 .. n-1); 0 = all (the
 * default).  For checking the ciphertexts of a full-GPU batch whose whole stream (1 024 x 47.7 GB) no PCIe link carries in reasonable
 * time: the reference holds no counterpart (its garbler hashes every instance, cut_and_choose/garbler.rs:219-222).  Set it before the
 * session's first streaming call (the gate-order buffers are sized by it). */
int gsv_session_set_drain_instances(gsv_session* s, size_t n);

/* The generic ciphertext sink: CiphertextHandler::handle (circuit/mod.rs:140-178) for ANY consumer — the reference has three impls, the
 * CBC-MAC accumulator, the hash + file writer and a channel Sender<S> (circuit/mod.rs:160-170) that feeds an evaluator thread; this is
 * the third.  The session is garbled window by window (program sessions: ring by ring) exactly as by gsv_session_garble_streaming, no
 * stream retained on the device, and every drained run of records is handed to `sink`:
 *     sink(user, instance, first_record, records, n_records)   records = n_records x 16 bytes, S::to_bytes() order, gate order;
 *                                                               first_record = index of records[0] in the instance's whole stream
 * The runs of ONE instance arrive in stream order, back to back, from one thread at a time; runs of different instances may arrive
 * concurrently from different host threads (n_threads = 1: everything from one thread).  `records` is only valid during the call.
 * A non-zero return aborts the pass (GSV_ERR_INVALID).  hashes (optional, n_instances x 16) additionally receives the CBC-MACs.
 * Plan sessions: calls [first_call, first_call + n_calls) as gsv_session_garble_streaming_calls (0, 0 = the whole plan). */
typedef int (*gsv_ct_sink_fn)(void* user, size_t instance, uint64_t first_record, const uint8_t* records, uint64_t n_records);
int gsv_session_garble_streaming_sink(gsv_session* s, uint64_t gate_id_base, uint64_t first_call, uint64_t n_calls, gsv_ct_sink_fn sink, void* user, int n_threads, uint8_t* hashes);

/* Garble and evaluate side by side on the device — the second phase of the reference's benchmark (examples/groth16_garble.rs:171-230,
 * tests/garbler_evaluator_connection.rs:64-172: a garbler thread feeds an evaluator thread through a channel of ciphertexts).  Both
 * sessions are plan sessions of the same plan on the same engine, created with the same options and retain_stream = 0; `evaluator` has
 * its inputs set (gsv_session_set_evaluate_inputs).  Window k of the garbler's device block is evaluated — on a second HIP stream,
 * straight from HBM, no PCIe, nothing retained — while window k+1 is garbled into the other one of two blocks.  hashes (optional)
 * receives the garbler's CBC-MACs (the stream is then drained to the host as well, as by gsv_session_garble_streaming).  Outputs:
 * gsv_session_read_outputs on either session. */
int gsv_session_garble_evaluate(gsv_session* garbler, gsv_session* evaluator, uint64_t gate_id_base, int n_threads, uint8_t* hashes);

/* Evaluate with the ciphertexts streamed from <dir>/gc_<first_index + instance>.bin (EvaluateMode over a FileSource:
 * evaluate_mode.rs:59-196, ciphertext_source.rs:36-107), one ring / one plan call at a time, for streams of any length.  Like
 * FileSource the call hashes what it reads: `hashes` (optional, n_instances x 16) receives each file's CBC-MAC so that the caller
 * can compare it with the garbler's commitment.  A file that is too short fails with GSV_ERR_EXHAUSTED. */
int gsv_session_evaluate_streaming(gsv_session* s, uint64_t gate_id_base, const char* dir, uint64_t first_index, uint8_t* hashes);
/* The same with an index per instance: instance i reads <dir>/gc_<indexes[i]>.bin.  This is how the finalized instances of a
 * cut-and-choose run — an arbitrary subset of 0..total-1 — are evaluated in ONE session / one launch per window
 * (Evaluator::evaluate_from is `into_par_iter` over the cases, cut_and_choose/evaluator.rs:354-475). */
int gsv_session_evaluate_streaming_indexed(gsv_session* s, uint64_t gate_id_base, const char* dir, const uint64_t* indexes, uint8_t* hashes);
/* The generic ciphertext source: CiphertextSource::recv (ciphertext_source.rs:14-34; the reference's impls are a channel Receiver<S>
 * and FileSource).  The engine pulls the stream in gate order, per instance, in bounded runs (<= 16 MiB):
 *     source(user, instance, first_record, records, n_records)   fill records[0 .. n_records x 16); return non-zero when the source
 *                                                                 has run dry -> GSV_ERR_EXHAUSTED (evaluate_mode.rs:139-142)
 * called from the calling thread.  hashes (optional) receives the CBC-MAC of what was read (FileSource hashes while reading). */
typedef int (*gsv_ct_source_fn)(void* user, size_t instance, uint64_t first_record, uint8_t* records, uint64_t n_records);
int gsv_session_evaluate_streaming_source(gsv_session* s, uint64_t gate_id_base, gsv_ct_source_fn source, void* user, uint8_t* hashes);

/* Gate PRF (`H: GateHasher`, src/hashers/mod.rs:15-20): GSV_HASHER_AES = AesNiHasher (default; the benchmarked
 * path, hashers/mod.rs:54-96), GSV_HASHER_BLAKE3 = Blake3Hasher (hashers/mod.rs:22-51, the crate's DefaultHasher). */
#define GSV_HASHER_AES 0
#define GSV_HASHER_BLAKE3 1
int gsv_session_set_hasher(gsv_session* s, int kind);

int gsv_session_sync(gsv_session* s);
/* 1, 2 or 4: how many instances share a workgroup (= a CU) in this session's launches.  Chosen at creation: 2 once the
 * session holds more instances than the device has CUs, 4 once it holds more than twice as many (each instance then works
 * with half / a quarter of the LDS label window; the program variant for that share is compiled on first use, or the plan was
 * built for it: GSV_PLAN_WINDOW_DIV), else 1; GSV_INSTANCES_PER_WG=1|2|4 overrides.  Results do not depend on it. */
int gsv_session_instances_per_workgroup(const gsv_session* s, int* n);
/* seconds of device time of the last garble/evaluate launch (HIP events on the engine stream) */
int gsv_session_last_kernel_ms(gsv_session* s, double* ms);
/* Diagnostics (tools/step_profile.py; no reference counterpart).  step clock: 100 MHz wall-clock stamps taken by instance 0's
 * workgroup at the start of every step of the LAST replay of a launch, n_steps+1 values.  step stats: per step
 * {and_cnt, xor_cnt, lds_reads, hbm_reads, lds_writes, hbm_writes}. */
int gsv_session_enable_step_clock(gsv_session* s);
int gsv_session_read_step_clock(gsv_session* s, uint64_t* out /* n_steps+1 */);
int gsv_program_step_stats(const gsv_program* p, uint32_t* out /* n_steps*6 */);

/* Outputs after sync.  Garble: label0 per output wire (label1 = label0 ^ delta).  Evaluate: active
 * label + plaintext bit per output wire. */
int gsv_session_read_outputs(gsv_session* s, uint8_t* labels /* n_inst*n_out*16 */, uint8_t* bits /* n_inst*n_out or NULL */);
/* CiphertextHandler side: copy `n_records` 16-byte ciphertexts of one instance, starting at stream
 * index `first`, to host memory (gate order, no framing). */
int gsv_session_read_ciphertexts(gsv_session* s, size_t instance, uint64_t first, uint64_t n_records, uint8_t* out);
/* AESAccumulatingHash over the instance's full retained stream (ciphertext_hasher.rs:23-29):
 * D2H in chunks + AES-NI CBC-MAC on the calling host thread. */
int gsv_session_ciphertext_hash(gsv_session* s, size_t instance, uint8_t hash[16]);
/* Stand-alone host CBC-MAC (AESAccumulatingHash) over a byte stream, chaining from `state`. */
int gsv_cbcmac_update(uint8_t state[16], const uint8_t* cts, uint64_t n_records);
/* The same for n_chains independent streams of n_records records each (states: n_chains x 16 bytes): chains are advanced four at
 * a time side by side — one chain is bound by the latency of its dependent AES rounds, four fill the AES unit — which is how the
 * engine's own drain hashes the instances' streams. */
int gsv_cbcmac_update_many(uint8_t* states, const uint8_t* const* cts, size_t n_chains, uint64_t n_records);
/* Chains gsv_cbcmac_update_many (and a session's drain) advances per step on this host: 16 with VAES + AVX-512, 4 with AES-NI, else 1. */
int gsv_cbcmac_chains_per_step(void);
/* AesLabelCommitHasher: AES_K(label) for n labels (cut_and_choose/mod.rs:41-48). */
int gsv_commit_labels(const uint8_t* labels, uint64_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* GSV_ENGINE_H */
