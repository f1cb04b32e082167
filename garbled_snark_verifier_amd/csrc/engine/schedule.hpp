// Call-level scheduling of a plan: which calls may run side by side.
//
// A plan is a sequence of calls {program, input globals, output globals} in STREAM order (the order in which the reference's
// streaming driver meets the components, src/circuit/streaming_mode.rs:150-247); stream order fixes every call's gate ids and the
// position of its ciphertext block, execution order is free as long as the data flow through the global wires is kept.  The
// reference garbles one instance on one core and gets its width from running `total` instances side by side
// (cut_and_choose/garbler.rs:206-234); with 1-16 instances on a 256-CU GPU the width has to come from inside an instance:
// sibling components (an Fq12 multiplication is 15 independent Fq2 multiplications, fq12.rs:199-221 / fq6.rs:194-260; the three
// point decompressions of groth16.rs:250-268 are independent ladders; the 26 window multiplexers of g1.rs:309-368 are independent).
//
// Model.  Global wire ids are MEMORY LOCATIONS of the instance's wire file (ids are recycled, plan_builder.hpp): a call reads its
// input ids in a pre-copy, runs in a scratch region of its own, writes its output ids in a post-copy.  A BATCH is a set of calls
// executed as: all pre-copies (one launch) ; all kernels side by side (one launch, grid.y = calls) ; all post-copies (one launch).
// Hazards between call i and a later (stream order) call k:
//   RAW  k reads an id i writes            -> k in a later batch than i
//   WAW  k writes an id i writes           -> k in a later batch than i   (ids nobody ever reads — trash ids — are exempt)
//   WAR  k writes an id i reads            -> k in the same batch as i or a later one (every pre-copy of a batch precedes every
//                                             post-copy of it)
// Calls are taken in WINDOWS of consecutive stream order: a window's ciphertext blocks are contiguous in the stream, so the
// device needs one window's block per instance (twice with the drain's gate-order copy) and the host consumes the stream window
// by window, in order (CiphertextHandler semantics: ciphertexts leave in gate order, circuit/mod.rs:140-178).  Inside a window
// calls are levelled by the hazards above (calls of earlier windows have completed); a level is cut into batches by the caller's
// limits (calls per batch, scratch slots per batch).  With max_calls_per_batch = 1 the schedule is the stream order itself.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace gsv {

struct SchedCall {
  const uint32_t* in = nullptr; size_t n_in = 0;    // global ids read (ids >= id_limit are constants: ignored)
  const uint32_t* out = nullptr; size_t n_out = 0;  // global ids written
  uint32_t n_slots = 0;                              // scratch slots of the call's program
  uint64_t n_ct = 0;                                 // ciphertext records of the call
  uint32_t n_steps = 0;
};
struct SchedParams {
  uint32_t max_calls_per_batch = 1;
  uint64_t max_batch_slots = ~0ull;   // sum of the scratch regions of a batch (16-byte slots per instance)
  uint64_t max_window_ct = ~0ull;     // ciphertext records per instance and window (a single larger call still forms a window)
  uint32_t max_window_calls = 4096;
  uint32_t slot_align = 8;            // scratch regions start on 128-byte lines
};
struct Schedule {
  struct Batch { uint32_t first, count; uint32_t max_steps; };          // order[first .. first+count)
  struct Window { uint32_t call0, call1, batch0, batch1; uint64_t ct0, n_ct; };  // calls [call0, call1) of the stream
  std::vector<uint32_t> order;         // execution order: call indices, batch after batch
  std::vector<uint32_t> scratch_base;  // per call (stream index): first slot of its scratch region, relative to the scratch area
  std::vector<Batch> batches;
  std::vector<Window> windows;
  uint64_t scratch_slots = 0;          // size of the scratch area (max over batches)
  uint64_t max_window_ct = 0;
  uint64_t critical_steps = 0;         // sum over batches of the longest call (device steps): the schedule's depth
  uint64_t total_steps = 0;            // sum over calls
};

inline Schedule schedule_calls(const std::vector<SchedCall>& calls, uint32_t n_ids, const std::vector<uint32_t>& plan_outputs, const SchedParams& p) {
  Schedule s;
  const size_t n = calls.size();
  s.scratch_base.assign(n, 0);
  // ids that are read by some call or are outputs of the plan; the rest (trash ids) carry no WAW / WAR hazard
  std::vector<uint8_t> ever_read(n_ids, 0);
  for (const SchedCall& c : calls) for (size_t i = 0; i < c.n_in; ++i) if (c.in[i] < n_ids) ever_read[c.in[i]] = 1;
  for (uint32_t w : plan_outputs) if (w < n_ids) ever_read[w] = 1;
  std::vector<uint32_t> w_stamp(n_ids, 0), r_stamp(n_ids, 0), w_level(n_ids, 0), r_level(n_ids, 0);
  std::vector<uint32_t> level;
  uint32_t stamp = 0;
  size_t k0 = 0;
  uint64_t ct_off = 0;
  const uint32_t max_batch = std::max<uint32_t>(1, p.max_calls_per_batch);
  auto align_up = [&](uint64_t v) { const uint64_t a = std::max<uint32_t>(1, p.slot_align); return (v + a - 1) / a * a; };
  while (k0 < n) {
    // ---- the window: consecutive calls under the ciphertext budget
    size_t k1 = k0;
    uint64_t wct = 0;
    while (k1 < n && (k1 == k0 || (wct + calls[k1].n_ct <= p.max_window_ct && k1 - k0 < p.max_window_calls))) wct += calls[k1++].n_ct;
    ++stamp;
    level.assign(k1 - k0, 0);
    uint32_t n_levels = 0;
    for (size_t k = k0; k < k1; ++k) {
      const SchedCall& c = calls[k];
      uint32_t lv = 0;
      if (max_batch > 1) {
        for (size_t i = 0; i < c.n_in; ++i) { const uint32_t g = c.in[i]; if (g < n_ids && w_stamp[g] == stamp) lv = std::max(lv, w_level[g] + 1); }
        for (size_t i = 0; i < c.n_out; ++i) {
          const uint32_t g = c.out[i];
          if (g >= n_ids || !ever_read[g]) continue;
          if (w_stamp[g] == stamp) lv = std::max(lv, w_level[g] + 1);
          if (r_stamp[g] == stamp) lv = std::max(lv, r_level[g]);
        }
        for (size_t i = 0; i < c.n_in; ++i) {
          const uint32_t g = c.in[i];
          if (g >= n_ids) continue;
          if (r_stamp[g] != stamp) { r_stamp[g] = stamp; r_level[g] = lv; } else r_level[g] = std::max(r_level[g], lv);
        }
        for (size_t i = 0; i < c.n_out; ++i) { const uint32_t g = c.out[i]; if (g < n_ids) { w_stamp[g] = stamp; w_level[g] = lv; } }
      } else {
        lv = uint32_t(k - k0);  // sequential: one call per batch, stream order
      }
      level[k - k0] = lv;
      n_levels = std::max(n_levels, lv + 1);
    }
    // ---- levels -> batches (stream order inside a level: a WAR pair of one level keeps reader-batch <= writer-batch)
    std::vector<std::vector<uint32_t>> by_level(n_levels);
    for (size_t k = k0; k < k1; ++k) by_level[level[k - k0]].push_back(uint32_t(k));
    Schedule::Window w{uint32_t(k0), uint32_t(k1), uint32_t(s.batches.size()), 0, ct_off, wct};
    for (const auto& lv_calls : by_level) {
      size_t i = 0;
      while (i < lv_calls.size()) {
        Schedule::Batch b{uint32_t(s.order.size()), 0, 0};
        uint64_t slots = 0;
        while (i < lv_calls.size() && b.count < max_batch) {
          const uint32_t k = lv_calls[i];
          const uint64_t need = align_up(calls[k].n_slots);
          if (b.count && slots + need > p.max_batch_slots) break;
          s.scratch_base[k] = uint32_t(slots);
          slots += need;
          s.order.push_back(k);
          b.max_steps = std::max(b.max_steps, calls[k].n_steps);
          ++b.count; ++i;
        }
        s.scratch_slots = std::max(s.scratch_slots, slots);
        s.critical_steps += b.max_steps;
        s.batches.push_back(b);
      }
    }
    w.batch1 = uint32_t(s.batches.size());
    s.windows.push_back(w);
    s.max_window_ct = std::max(s.max_window_ct, wct);
    ct_off += wct;
    k0 = k1;
  }
  for (const SchedCall& c : calls) s.total_steps += c.n_steps;
  return s;
}

// Checks a schedule against the hazard rules by replaying it over id "versions" (test helper; also run by the engine in debug
// builds): returns an empty string when every call reads exactly the versions it reads in stream order.
inline std::string verify_schedule(const std::vector<SchedCall>& calls, uint32_t n_ids, const std::vector<uint32_t>& plan_outputs, const Schedule& s) {
  const size_t n = calls.size();
  if (s.order.size() != n) return "order does not cover every call";
  // stream-order semantics: version of an id = index of the call that wrote it last (+1), 0 = initial
  std::vector<uint32_t> ver(n_ids, 0);
  std::vector<std::vector<uint32_t>> want(n);
  for (size_t k = 0; k < n; ++k) {
    for (size_t i = 0; i < calls[k].n_in; ++i) { const uint32_t g = calls[k].in[i]; want[k].push_back(g < n_ids ? ver[g] : 0); }
    for (size_t i = 0; i < calls[k].n_out; ++i) { const uint32_t g = calls[k].out[i]; if (g < n_ids) ver[g] = uint32_t(k + 1); }
  }
  std::vector<uint32_t> final_want(plan_outputs.size());
  for (size_t i = 0; i < plan_outputs.size(); ++i) final_want[i] = plan_outputs[i] < n_ids ? ver[plan_outputs[i]] : 0;
  std::fill(ver.begin(), ver.end(), 0);
  std::vector<uint8_t> ever_read(n_ids, 0);
  for (const SchedCall& c : calls) for (size_t i = 0; i < c.n_in; ++i) if (c.in[i] < n_ids) ever_read[c.in[i]] = 1;
  for (uint32_t w : plan_outputs) if (w < n_ids) ever_read[w] = 1;
  std::vector<uint8_t> seen(n, 0);
  for (const Schedule::Batch& b : s.batches) {
    for (uint32_t j = 0; j < b.count; ++j) {  // all pre-copies
      const uint32_t k = s.order[b.first + j];
      if (k >= n || seen[k]) return "call scheduled twice / out of range";
      seen[k] = 1;
      for (size_t i = 0; i < calls[k].n_in; ++i) { const uint32_t g = calls[k].in[i]; if (g < n_ids && ver[g] != want[k][i]) return "call " + std::to_string(k) + " reads a stale or clobbered global"; }
    }
    std::vector<std::pair<uint32_t, uint32_t>> written;  // all post-copies: two calls of a batch must not write one live id
    for (uint32_t j = 0; j < b.count; ++j) {
      const uint32_t k = s.order[b.first + j];
      for (size_t i = 0; i < calls[k].n_out; ++i) { const uint32_t g = calls[k].out[i]; if (g < n_ids) written.push_back({g, k}); }
    }
    std::sort(written.begin(), written.end());
    for (size_t i = 0; i < written.size(); ++i) {
      // several writers of one id inside a batch (their order inside the post-copy launch is undefined): only for ids nobody reads
      if (i && written[i].first == written[i - 1].first && ever_read[written[i].first]) return "two calls of a batch write global " + std::to_string(written[i].first);
      ver[written[i].first] = written[i].second + 1;
    }
  }
  for (size_t i = 0; i < plan_outputs.size(); ++i) if (plan_outputs[i] < n_ids && ver[plan_outputs[i]] != final_want[i]) return "a plan output ends with the wrong version";
  // scratch regions of a batch must not overlap
  for (const Schedule::Batch& b : s.batches) {
    std::vector<std::pair<uint64_t, uint64_t>> r;
    for (uint32_t j = 0; j < b.count; ++j) { const uint32_t k = s.order[b.first + j]; r.push_back({s.scratch_base[k], uint64_t(s.scratch_base[k]) + calls[k].n_slots}); }
    std::sort(r.begin(), r.end());
    for (size_t i = 1; i < r.size(); ++i) if (r[i].first < r[i - 1].second) return "scratch regions of a batch overlap";
    for (auto& x : r) if (x.second > s.scratch_slots) return "scratch region outside the scratch area";
  }
  return std::string();
}

}  // namespace gsv
