// Part of engine.cpp: plan sessions — schedule, wire-file layout and device tables (install_schedule), inputs, window launches, the non-streaming garble.
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
  if (const char* dg = getenv("GSV_DIAG")) ka.diag = uint32_t(atoi(dg));  // timing experiments (libgsv_engine_diag.so only): outputs are wrong when set
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
