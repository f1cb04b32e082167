// Part of engine.cpp: evaluation (resident stream and streaming sources) and the remaining small entry points.
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
static int evaluate_streaming_pass(gsv_session* s, uint64_t gate_id_base, const std::function<int(size_t, uint64_t, uint8_t*, uint64_t)>& read, uint8_t* hashes) {
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
// The evaluator's side of the safe-schedule fallback (engine_drain.ipp, fall_back_to_safe_schedule): a pass that ended with a dependency wait
// giving up switches the session to one call per launch; a source that can be read again from the start (gc files) is then evaluated
// again at once, any other source gets the error with the remedy (its records have been consumed) and the host's repeat succeeds.
static int evaluate_streaming_impl(gsv_session* s, uint64_t gate_id_base, const std::function<int(size_t, uint64_t, uint8_t*, uint64_t)>& read, uint8_t* hashes, bool rereadable = false) {
  int rc = evaluate_streaming_pass(s, gate_id_base, read, hashes);
  if (rc != GSV_ERR_DEVICE || !s->plan || !s->dep_fault || s->safe_mode) return rc;
  const std::string first_error = g_err;
  if (fall_back_to_safe_schedule(s)) return fail(GSV_ERR_DEVICE, first_error + "; the fall-back to the safe schedule failed too: " + g_err);
  if (!rereadable) return fail(GSV_ERR_DEVICE, first_error + "; the session now runs the safe schedule (one call per launch): repeat the pass from gsv_session_set_evaluate_inputs");
  if (getenv("GSV_DRAIN_DEBUG") || getenv("GSV_PLAN_DEBUG")) std::fprintf(stderr, "plan session: %s -- repeating the evaluation on the safe schedule (one call per launch)\n", first_error.c_str());
  return evaluate_streaming_pass(s, gate_id_base, read, hashes);
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
  }, hashes, /*rereadable=*/true);
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
