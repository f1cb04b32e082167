// Part of engine.cpp: the streaming garbler — drain pipeline (gather, D2H, CBC-MAC / files / sink), garble || evaluate pairs, the safe-schedule fallback.
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
  if (rc != GSV_ERR_DEVICE || !s->plan || !(s->dep_fault || (ev && ev->dep_fault)) || s->safe_mode) return rc;
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
