// Part of engine.cpp: program sessions — window variants, program upload, create / destroy.
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
