// Part of engine.cpp: recorder, program, engine and label entry points.

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
