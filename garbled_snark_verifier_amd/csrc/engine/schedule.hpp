// Call-level scheduling of a plan: which calls may run side by side, as a DATAFLOW over the calls of a window.
//
// A plan is a sequence of calls {program, input globals, output globals} in STREAM order (the order in which the reference's
// streaming driver meets the components, src/circuit/streaming_mode.rs:150-247); stream order fixes every call's gate ids and the
// position of its ciphertext block, execution order is free as long as the data flow through the global wires is kept.  The
// reference garbles one instance on one core and gets its width from running `total` instances side by side
// (cut_and_choose/garbler.rs:206-234); with 1-16 instances on a 256-CU GPU the width has to come from inside an instance:
// sibling components (an Fq12 multiplication is 15 independent Fq2 multiplications, fq12.rs:199-221 / fq6.rs:194-260; the three
// point decompressions of groth16.rs:250-268 are independent ladders; the 26 window multiplexers of g1.rs:309-368 are independent).
//
// Execution model (kernels.hip, run_program_kernel with KernelArgs::calls): ONE launch per window, grid = (instance groups, calls of
// the window in stream order).  A workgroup (x, c) = call c for instance group x: it waits until the calls it depends on have set
// their completion flags for group x, copies its inputs from the instance's global wires into a scratch region of its own, runs the
// program, copies its outputs back to the globals and sets its flag.  Workgroups are dispatched in linear order (x fastest), every
// dependency points to an EARLIER call, so a waiting workgroup only ever waits for workgroups that are running or finished (the
// forward-progress argument of decoupled look-back scans); a bounded wait turns a violated assumption into an error, not a hang.
//
// Dependencies.  Global wire ids are MEMORY LOCATIONS of the instance's wire file (ids are recycled, plan_builder.hpp).  For a call k
// and an earlier call i of the same window:
//   RAW  k reads an id i wrote last                 -> k waits for i
//   WAW  k writes an id i wrote last                -> k waits for i   (ids nobody ever reads — trash ids — are exempt)
//   WAR  k writes an id i read (the old value)      -> k waits for i
//   in-flight bound: k waits for call k - C         -> at most C calls of an instance in flight (C = 1: the stream order itself)
//   scratch ring: k's scratch region is carved from a ring of slots; k waits for the earlier calls whose regions it overlaps
// Calls of earlier windows have completed (previous launch).  A window is a run of consecutive calls: its ciphertext blocks are
// contiguous in the stream, so the device needs one window's block per instance (twice with the drain's gate-order copy) and the
// host consumes the stream window by window, in order (CiphertextHandler semantics: circuit/mod.rs:140-178).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace gsv {

struct SchedCall {
  const uint32_t* in = nullptr; size_t n_in = 0;    // global ids read (ids >= n_ids are the constant wires: ignored)
  const uint32_t* out = nullptr; size_t n_out = 0;  // global ids written
  uint32_t n_slots = 0;                              // scratch slots of the call's program
  uint64_t n_ct = 0;                                 // ciphertext records of the call
  uint32_t n_steps = 0;
};
struct SchedParams {
  uint32_t max_calls_in_flight = 1;   // C above
  uint64_t max_scratch_slots = ~0ull; // size limit of the scratch ring (16-byte slots per instance)
  uint64_t max_window_ct = ~0ull;     // ciphertext records per instance and window (a single larger call still forms a window)
  // Drain segments: a window is ONE launch and the scope inside which independent call chains overlap (a window boundary joins every
  // chain: the verifier's line-coefficient chain comes before the Miller loop in stream order and runs beside it only when both lie in
  // one window), so windows want to be as large as the device block allows; the host side of a drain (copies, the serial CBC-MAC
  // chains) wants SMALL units, because it can only start on a unit when the device has finished it and the last unit's share is the
  // pass's tail.  A window is therefore cut into segments of consecutive calls of at most segment_ct ciphertext records: the host
  // follows the completion flags of a running window and drains segment after segment while the window is still being garbled.
  uint64_t segment_ct = 0;            // 0 = one segment per window
  // Ciphertext RING (sessions that do not retain the stream): the device's program-order block holds ring_ct records per instance
  // instead of a whole window's, so a window — the scope in which independent call chains overlap — can span the whole pass whatever
  // the number of instances.  A call's block sits at ring_off inside the ring; before a garbling call writes it, everything the ring
  // held there on the previous lap must have left the device: the call waits (kernels.hip, prologue) until the host's "gathered up to"
  // position has reached ring_need (the END of the drain segment that holds the last overlapped call).  An evaluating call waits until
  // its own segment has been uploaded: seg_end.  ring_ct >= 2 x the largest segment + the largest call, so that the segment a call
  // waits for never contains the call itself (or a call behind it).  0 = no ring: the block is one window.
  uint64_t ring_ct = 0;
  uint32_t max_window_calls = 32768;  // (grid.y of a launch is at most 65535)
  uint32_t slot_align = 8;            // scratch regions start on 128-byte lines
};
struct Schedule {
  struct Window { uint32_t call0, call1; uint64_t ct0, n_ct; uint32_t max_width; uint32_t seg0, seg1; };  // calls [call0, call1) of the stream; segments [seg0, seg1)
  std::vector<Window> windows;
  struct Segment { uint32_t call0, call1; uint64_t ct0, n_ct; };  // consecutive calls of one window: the unit the stream leaves the device in
  std::vector<Segment> segments;
  uint64_t max_segment_ct = 0;
  // ring mode (SchedParams::ring_ct): per call its block's offset inside the ring, the stream position that must have been gathered off
  // the device before a garbling call may write there (0 = nothing), the end of the call's own segment (what an evaluating call waits
  // for), and — for the host side of an evaluation — the calls of the previous lap the call's block overlaps: [ovl0, ovl1)
  uint64_t ring_ct = 0;
  std::vector<uint64_t> ring_off, ring_need, seg_end;
  std::vector<uint32_t> ovl0, ovl1;
  std::vector<uint32_t> scratch_base;  // per call: first slot of its scratch region
  std::vector<uint32_t> dep_off;       // per call (+1): its dependencies are deps[dep_off[k] .. dep_off[k+1])
  std::vector<uint32_t> deps;          // call indices (stream order, same window, < k)
  uint64_t scratch_slots = 0;          // size of the scratch ring
  uint64_t max_window_ct = 0;
  uint64_t critical_steps = 0;         // sum over windows of the longest dependency path (device steps): the schedule's depth
  uint64_t total_steps = 0;            // sum over calls
  uint32_t max_width = 0;              // most calls that can be in flight at once (by start / finish times of the step model)
};

inline Schedule schedule_calls(const std::vector<SchedCall>& calls, uint32_t n_ids, const std::vector<uint32_t>& plan_outputs, const SchedParams& p) {
  Schedule s;
  const size_t n = calls.size();
  s.scratch_base.assign(n, 0);
  s.dep_off.assign(n + 1, 0);
  s.ring_ct = p.ring_ct;
  if (p.ring_ct) { s.ring_off.assign(n, 0); s.ring_need.assign(n, 0); s.seg_end.assign(n, 0); s.ovl0.assign(n, 0); s.ovl1.assign(n, 0); }
  const uint32_t C = std::max<uint32_t>(1, p.max_calls_in_flight);
  auto align_up = [&](uint64_t v) { const uint64_t a = std::max<uint32_t>(1, p.slot_align); return (v + a - 1) / a * a; };
  // ids that are read by some call or are outputs of the plan; the rest (trash ids) carry no WAW hazard
  std::vector<uint8_t> ever_read(n_ids, 0);
  for (const SchedCall& c : calls) for (size_t i = 0; i < c.n_in; ++i) if (c.in[i] < n_ids) ever_read[c.in[i]] = 1;
  for (uint32_t w : plan_outputs) if (w < n_ids) ever_read[w] = 1;
  // the scratch ring: large enough for C consecutive regions (then the in-flight bound is the only ring constraint), within the limit
  uint64_t need = 0, max_one = 0;
  {
    uint64_t run = 0;
    for (size_t k = 0; k < n; ++k) {
      run += align_up(calls[k].n_slots);
      if (k >= C) run -= align_up(calls[k - C].n_slots);
      need = std::max(need, run);
      max_one = std::max<uint64_t>(max_one, align_up(calls[k].n_slots));
    }
  }
  const uint64_t ring = std::max<uint64_t>(std::min<uint64_t>(need, std::min<uint64_t>(p.max_scratch_slots, 0xFFFFFF00ull)), max_one);
  s.scratch_slots = ring;
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  std::vector<uint32_t> w_call(n_ids, NONE);          // last writer of the id inside the current window
  std::vector<uint32_t> r_head(n_ids, NONE);          // readers of the id's current value: list through `rnode`
  struct RNode { uint32_t call, next; };
  std::vector<RNode> rnode;
  std::vector<uint32_t> touched;                       // ids with window state (reset at the window's end)
  struct Occ { uint32_t call; uint64_t lo, hi; };
  std::vector<Occ> occ;                                // live scratch regions: handed out and not yet overwritten by a later region
  uint64_t cur = 0;
  std::vector<uint32_t> d;
  std::vector<uint64_t> finish(n, 0);                  // step model: a call starts when its dependencies have finished
  size_t k0 = 0;
  uint64_t ct_off = 0;
  while (k0 < n) {
    size_t k1 = k0;
    uint64_t wct = 0;
    while (k1 < n && (k1 == k0 || (wct + calls[k1].n_ct <= p.max_window_ct && k1 - k0 < p.max_window_calls))) wct += calls[k1++].n_ct;
    rnode.clear(); touched.clear(); occ.clear(); cur = 0;
    uint64_t depth = 0;
    std::vector<std::pair<uint64_t, int>> ev;  // (time, +1 start / -1 finish) for the width estimate
    for (size_t k = k0; k < k1; ++k) {
      const SchedCall& c = calls[k];
      d.clear();
      for (size_t i = 0; i < c.n_in; ++i) { const uint32_t g = c.in[i]; if (g < n_ids && w_call[g] != NONE) d.push_back(w_call[g]); }            // RAW
      for (size_t i = 0; i < c.n_out; ++i) {
        const uint32_t g = c.out[i];
        if (g >= n_ids) continue;
        if (ever_read[g] && w_call[g] != NONE) d.push_back(w_call[g]);                                                                        // WAW
        for (uint32_t r = r_head[g]; r != NONE; r = rnode[r].next) if (rnode[r].call != uint32_t(k)) d.push_back(rnode[r].call);               // WAR
      }
      if (k - k0 >= C) d.push_back(uint32_t(k - C));                                                                                           // in-flight bound
      // scratch region from the ring; the earlier calls it overlaps must have finished
      const uint64_t need_k = align_up(c.n_slots);
      if (cur + need_k > ring) cur = 0;
      const uint64_t lo = cur, hi = cur + need_k;
      const size_t n_occ = occ.size();
      for (size_t i = 0; i < n_occ; ++i) {  // (the live regions are few: about as many as calls can be in flight)
        Occ o = occ[i];
        if (!(o.lo < hi && lo < o.hi)) continue;
        d.push_back(o.call);
        // what the new region leaves of the old one stays live (a later region may still run into it)
        occ[i].hi = occ[i].lo;  // emptied; compacted below
        if (o.lo < lo) occ.push_back(Occ{o.call, o.lo, lo});
        if (o.hi > hi) occ.push_back(Occ{o.call, hi, o.hi});
      }
      occ.erase(std::remove_if(occ.begin(), occ.end(), [](const Occ& o) { return o.lo >= o.hi; }), occ.end());
      s.scratch_base[k] = uint32_t(lo);
      cur = hi;
      occ.push_back(Occ{uint32_t(k), lo, hi});
      // bookkeeping of the ids
      for (size_t i = 0; i < c.n_in; ++i) {
        const uint32_t g = c.in[i];
        if (g >= n_ids) continue;
        if (w_call[g] == NONE && r_head[g] == NONE) touched.push_back(g);
        rnode.push_back(RNode{uint32_t(k), r_head[g]});
        r_head[g] = uint32_t(rnode.size() - 1);
      }
      for (size_t i = 0; i < c.n_out; ++i) {
        const uint32_t g = c.out[i];
        if (g >= n_ids) continue;
        if (w_call[g] == NONE && r_head[g] == NONE) touched.push_back(g);
        w_call[g] = uint32_t(k);
        r_head[g] = NONE;  // the new value has no readers yet
      }
      std::sort(d.begin(), d.end());
      d.erase(std::unique(d.begin(), d.end()), d.end());
      uint64_t start = 0;
      for (uint32_t j : d) { s.deps.push_back(j); start = std::max(start, finish[j]); }
      s.dep_off[k + 1] = uint32_t(s.deps.size());
      finish[k] = start + c.n_steps;
      depth = std::max(depth, finish[k]);
      ev.push_back({start, +1}); ev.push_back({finish[k], -1});
    }
    for (uint32_t g : touched) { w_call[g] = NONE; r_head[g] = NONE; }
    std::sort(ev.begin(), ev.end());
    int width = 0, max_width = 0;
    for (auto& e : ev) { width += e.second; max_width = std::max(max_width, width); }
    const uint32_t seg0 = uint32_t(s.segments.size());
    {
      size_t a = k0;
      uint64_t off = ct_off;
      while (a < k1) {
        size_t b = a;
        uint64_t sct = 0;
        while (b < k1 && (b == a || p.segment_ct == 0 || sct + calls[b].n_ct <= p.segment_ct)) sct += calls[b++].n_ct;
        s.segments.push_back(Schedule::Segment{uint32_t(a), uint32_t(b), off, sct});
        s.max_segment_ct = std::max(s.max_segment_ct, sct);
        off += sct;
        a = b;
      }
    }
    if (p.ring_ct) {
      // ring placement of the window's calls, in stream order; a block never wraps (a call that does not fit the rest of the ring starts
      // at 0 and leaves a gap).  `lap` holds the calls whose blocks are still in the ring, oldest first.
      struct Held { uint32_t call; uint64_t lo, hi; };
      std::vector<Held> held;  // blocks still in the ring (a few hundred at most: the ring holds a few segments)
      uint64_t pos = 0;
      std::vector<uint64_t> end_of_seg_of(k1 - k0, 0);
      for (uint32_t q = seg0; q < s.segments.size(); ++q)
        for (uint32_t k = s.segments[q].call0; k < s.segments[q].call1; ++k) end_of_seg_of[k - k0] = s.segments[q].ct0 + s.segments[q].n_ct;
      for (size_t k = k0; k < k1; ++k) {
        const uint64_t nct = calls[k].n_ct;
        if (nct > p.ring_ct) throw std::runtime_error("internal: ciphertext ring smaller than a call's block");
        if (pos + nct > p.ring_ct) pos = 0;
        const uint64_t lo = pos, hi = pos + nct;
        uint64_t need = 0;
        uint32_t o0 = uint32_t(k), o1 = uint32_t(k);
        const size_t n_held = held.size();
        size_t keep = 0;
        for (size_t i = 0; i < n_held; ++i) {
          const Held h = held[i];
          if (nct && h.lo < hi && lo < h.hi) {  // (partly) overwritten by this call: its segment must have left the device / its call must be done
            need = std::max(need, end_of_seg_of[h.call - k0]);
            o0 = std::min(o0, h.call);
            o1 = std::max(o1, h.call + 1);
            // what this call leaves of the block still holds that call's records: calls run in any order the dependencies allow, so a
            // later call that takes the rest has to wait for the same segment itself
            if (h.lo < lo) held.push_back(Held{h.call, h.lo, lo});
            if (h.hi > hi) held.push_back(Held{h.call, hi, h.hi});
          } else held[keep++] = h;
        }
        for (size_t i = n_held; i < held.size(); ++i) held[keep++] = held[i];
        held.resize(keep);
        s.ring_off[k] = lo; s.ring_need[k] = need; s.seg_end[k] = end_of_seg_of[k - k0]; s.ovl0[k] = o0; s.ovl1[k] = o1 > o0 ? o1 : o0;
        if (nct) held.push_back(Held{uint32_t(k), lo, hi});
        pos = hi;
      }
    }
    s.windows.push_back(Schedule::Window{uint32_t(k0), uint32_t(k1), ct_off, wct, uint32_t(max_width), seg0, uint32_t(s.segments.size())});
    s.max_width = std::max(s.max_width, uint32_t(max_width));
    s.max_window_ct = std::max(s.max_window_ct, wct);
    s.critical_steps += depth;
    ct_off += wct;
    k0 = k1;
  }
  for (const SchedCall& c : calls) s.total_steps += c.n_steps;
  return s;
}

// Test helper: checks a schedule by brute force — returns an empty string when (a) every hazard between two calls of a window is
// covered by a dependency PATH and (b) two calls whose scratch regions overlap are ordered by a path.  O(n^2 / 64) per window.
inline std::string verify_schedule(const std::vector<SchedCall>& calls, uint32_t n_ids, const std::vector<uint32_t>& plan_outputs, const Schedule& s) {
  const size_t n = calls.size();
  if (s.dep_off.size() != n + 1 || s.scratch_base.size() != n) return "schedule does not cover every call";
  std::vector<uint8_t> ever_read(n_ids, 0);
  for (const SchedCall& c : calls) for (size_t i = 0; i < c.n_in; ++i) if (c.in[i] < n_ids) ever_read[c.in[i]] = 1;
  for (uint32_t w : plan_outputs) if (w < n_ids) ever_read[w] = 1;
  size_t covered = 0;
  for (const Schedule::Window& w : s.windows) {
    if (w.call0 != covered || w.call1 <= w.call0 || w.call1 > n) return "windows are not a partition of the calls";
    covered = w.call1;
    const size_t m = w.call1 - w.call0, words = (m + 63) / 64;
    std::vector<uint64_t> reach(m * words, 0);  // reach[k] = set of calls k depends on, transitively
    auto row = [&](size_t k) { return &reach[k * words]; };
    for (size_t k = 0; k < m; ++k) {
      for (uint32_t q = s.dep_off[w.call0 + k]; q < s.dep_off[w.call0 + k + 1]; ++q) {
        const uint32_t j = s.deps[q];
        if (j < w.call0 || j >= w.call0 + k) return "a dependency points outside the window or forward";
        const size_t jj = j - w.call0;
        row(k)[jj / 64] |= 1ull << (jj % 64);
        for (size_t t = 0; t < words; ++t) row(k)[t] |= row(jj)[t];
      }
    }
    auto ordered = [&](size_t i, size_t k) { return (row(k)[i / 64] >> (i % 64)) & 1ull; };  // i before k
    // hazards
    std::vector<std::vector<uint32_t>> readers(n_ids), writers(n_ids);
    for (size_t k = 0; k < m; ++k) {
      const SchedCall& c = calls[w.call0 + k];
      for (size_t i = 0; i < c.n_in; ++i) {
        const uint32_t g = c.in[i];
        if (g >= n_ids) continue;
        if (!writers[g].empty() && !ordered(writers[g].back(), k)) return "RAW hazard not ordered: call " + std::to_string(w.call0 + k);
        readers[g].push_back(uint32_t(k));
      }
      for (size_t i = 0; i < c.n_out; ++i) {
        const uint32_t g = c.out[i];
        if (g >= n_ids) continue;
        if (ever_read[g] && !writers[g].empty() && writers[g].back() != k && !ordered(writers[g].back(), k)) return "WAW hazard not ordered: call " + std::to_string(w.call0 + k);
        for (uint32_t r : readers[g]) if (r != k && !ordered(r, k)) return "WAR hazard not ordered: call " + std::to_string(w.call0 + k);
        readers[g].clear();
        writers[g].push_back(uint32_t(k));
      }
    }
    // scratch regions
    for (size_t k = 0; k < m; ++k) {
      const uint64_t lo = s.scratch_base[w.call0 + k], hi = lo + calls[w.call0 + k].n_slots;
      if (hi > s.scratch_slots) return "scratch region outside the ring";
      for (size_t i = 0; i < k; ++i) {
        const uint64_t lo2 = s.scratch_base[w.call0 + i], hi2 = lo2 + calls[w.call0 + i].n_slots;
        if (lo < hi2 && lo2 < hi && !ordered(i, k)) return "overlapping scratch regions of unordered calls " + std::to_string(w.call0 + i) + " and " + std::to_string(w.call0 + k);
      }
    }
  }
  if (covered != n) return "windows do not cover every call";
  // segments partition every window's calls and ciphertexts in stream order
  for (const Schedule::Window& w : s.windows) {
    uint32_t c = w.call0;
    uint64_t ct = w.ct0;
    for (uint32_t q = w.seg0; q < w.seg1; ++q) {
      if (q >= s.segments.size() || s.segments[q].call0 != c || s.segments[q].call1 <= c || s.segments[q].ct0 != ct) return "segments do not partition a window";
      uint64_t sum = 0;
      for (uint32_t k = s.segments[q].call0; k < s.segments[q].call1; ++k) sum += calls[k].n_ct;
      if (sum != s.segments[q].n_ct) return "a segment's ciphertext count is wrong";
      c = s.segments[q].call1; ct += sum;
    }
    if (c != w.call1 || ct != w.ct0 + w.n_ct) return "segments do not cover a window";
  }
  // ring mode, by simulation (record by record: small rings only — the tests'): when call k takes its block, every record it overwrites
  // belongs to a call whose segment ends at or before ring_need[k], that segment does not contain k or a later call, the overwritten
  // calls lie in [ovl0, ovl1), and an evaluating call's own segment ends at seg_end[k]
  if (s.ring_ct && s.ring_ct <= (1u << 22)) {
    if (s.ring_off.size() != n || s.ring_need.size() != n || s.seg_end.size() != n || s.ovl0.size() != n || s.ovl1.size() != n) return "ring tables do not cover every call";
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    for (const Schedule::Window& w : s.windows) {
      std::vector<uint32_t> owner(size_t(s.ring_ct), NONE);
      std::vector<uint64_t> seg_end_of(w.call1 - w.call0, 0), seg_first_of(w.call1 - w.call0, 0);
      for (uint32_t q = w.seg0; q < w.seg1; ++q)
        for (uint32_t k = s.segments[q].call0; k < s.segments[q].call1; ++k) { seg_end_of[k - w.call0] = s.segments[q].ct0 + s.segments[q].n_ct; seg_first_of[k - w.call0] = s.segments[q].call0; }
      for (uint32_t k = w.call0; k < w.call1; ++k) {
        const uint64_t lo = s.ring_off[k], hi = lo + calls[k].n_ct;
        if (hi > s.ring_ct) return "a call's block leaves the ring";
        if (s.seg_end[k] != seg_end_of[k - w.call0]) return "seg_end of a call is not the end of its segment";
        for (uint64_t r = lo; r < hi; ++r) {
          const uint32_t i = owner[size_t(r)];
          if (i != NONE && i != k) {
            if (seg_end_of[i - w.call0] > s.ring_need[k]) return "ring: a call overwrites records of a segment it does not wait for";
            if (i < s.ovl0[k] || i >= s.ovl1[k]) return "ring: an overwritten call lies outside [ovl0, ovl1)";
          }
        }
        for (uint64_t r = lo; r < hi; ++r) owner[size_t(r)] = k;
        // the wait must be satisfiable without call k (or anything behind it) having completed: the awaited position ends in front of k's segment
        if (s.ring_need[k] > s.segments[0].ct0 && s.ring_need[k] != 0) {
          // the segment that ends at ring_need[k] must end at or before the first call of k's own segment
          uint64_t first_ct_of_own_seg = 0;
          for (uint32_t q = w.seg0; q < w.seg1; ++q) if (s.segments[q].call0 == seg_first_of[k - w.call0]) first_ct_of_own_seg = s.segments[q].ct0;
          if (s.ring_need[k] > first_ct_of_own_seg) return "ring: a call waits for its own segment (ring too small)";
        }
      }
    }
  }
  return std::string();
}

// The O(calls) part of the ring checks, run for EVERY session with a ciphertext ring (verify_schedule's record-by-record simulation only
// fits the tests' small rings): every call's block lies inside the ring, seg_end is the end of the call's own drain segment, and the
// position a garbling call waits for (ring_need) ends at or before its OWN segment's first record — a call that waited for its own
// segment would wait for itself: the 60 s device stall + status 2 that a ring too small for the plan used to end in.
inline std::string verify_ring_bounds(const std::vector<SchedCall>& calls, const Schedule& s) {
  if (!s.ring_ct) return std::string();
  const size_t n = calls.size();
  if (s.ring_off.size() != n || s.ring_need.size() != n || s.seg_end.size() != n) return "ring tables do not cover every call";
  for (const Schedule::Window& w : s.windows)
    for (uint32_t q = w.seg0; q < w.seg1; ++q) {
      if (q >= s.segments.size()) return "a window names a segment that does not exist";
      const Schedule::Segment& sg = s.segments[q];
      for (uint32_t k = sg.call0; k < sg.call1; ++k) {
        if (k >= n) return "a segment names a call that does not exist";
        if (s.ring_off[k] + calls[k].n_ct > s.ring_ct) return "ring: the block of call " + std::to_string(k) + " leaves the ring";
        if (s.seg_end[k] != sg.ct0 + sg.n_ct) return "ring: seg_end of call " + std::to_string(k) + " is not the end of its segment";
        if (s.ring_need[k] > sg.ct0) return "ring: call " + std::to_string(k) + " waits for its own drain segment (the ring is too small for this plan: raise window_ct_records / GSV_CT_RING_RECORDS)";
      }
    }
  return std::string();
}

}  // namespace gsv
