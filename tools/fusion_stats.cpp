// Analysis tool (not part of the product): how much of a circuit's free-gate work disappears when
// single-reader XOR/XNOR/NOT outputs are folded into their reader (operand lists of up to KX wires
// for a free gate, KA wires per AND input).
//   g++ -O2 -std=c++17 -I garbled_snark_verifier_amd/csrc tools/fusion_stats.cpp -o /tmp/fusion_stats
//   /tmp/fusion_stats fq12_mul 4 2
#include <cstdio>
#include <cstdlib>
#include <map>

#include "engine/program.hpp"
#include "gadgets/circuits.hpp"

using namespace gsv;

int main(int argc, char** argv) {
  std::string spec = argc > 1 ? argv[1] : "fq_mul";
  const unsigned KX = argc > 2 ? atoi(argv[2]) : 4, KA = argc > 3 ? atoi(argv[3]) : 2;
  RecordMode mode;
  NamedCircuit nc = make_circuit(spec);
  StreamingRunner run(mode, nc.n_inputs, nc.fn);
  std::vector<uint32_t> inputs, outputs;
  for (WireId w : run.prepare()) inputs.push_back(mode.define_input(w));
  for (WireId w : run.execute()) outputs.push_back(mode.current(w));
  const Trace& t = mode.trace();
  const size_t n = t.size();
  const uint32_t nw = t.n_wires;
  std::vector<uint32_t> fan(nw, 0);
  std::vector<uint8_t> pinned(nw, 0), is_free(nw, 0);
  for (uint32_t w : outputs) pinned[w] = 1;
  size_t live = 0, n_and = 0, n_free = 0;
  for (size_t i = 0; i < n; ++i) {
    if (t.c[i] == DEAD_WIRE) continue;
    ++live;
    fan[t.a[i]]++;
    if (t.type[i] != uint8_t(GateType::Not)) fan[t.b[i]]++;
    if (t.type[i] >= 8) { is_free[t.c[i]] = 1; ++n_free; } else ++n_and;
  }
  std::map<uint32_t, size_t> fh;
  size_t cand = 0;
  for (size_t i = 0; i < n; ++i) {
    if (t.c[i] == DEAD_WIRE || t.type[i] < 8) continue;
    uint32_t f = fan[t.c[i]];
    fh[std::min(f, 5u)]++;
    if (f == 1 && !pinned[t.c[i]]) ++cand;
  }
  printf("%s: %zu gates, %zu live, %zu and, %zu free; free-gate fan-out histogram:", spec.c_str(), n, live, n_and, n_free);
  for (auto& kv : fh) printf(" %u:%zu", kv.first, kv.second);
  printf("\nsingle-reader free outputs: %zu (%.1f%% of free gates)\n", cand, 100.0 * cand / n_free);

  // fold
  std::vector<std::vector<uint32_t>> expr(nw);  // operand list of a free gate's output (sorted, duplicates cancelled)
  std::vector<uint8_t> mat(nw, 1);              // materialised?
  auto ops_of = [&](uint32_t x, std::vector<uint32_t>& out) {
    if (is_free[x] && fan[x] == 1 && !pinned[x]) out = expr[x]; else out = {x};
  };
  auto symdiff = [](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    std::vector<uint32_t> r;
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
      if (j == b.size() || (i < a.size() && a[i] < b[j])) r.push_back(a[i++]);
      else if (i == a.size() || b[j] < a[i]) r.push_back(b[j++]);
      else { ++i; ++j; }
    }
    return r;
  };
  std::vector<uint32_t> lev0(nw, 0), lev1(nw, 0);
  size_t free_kept = 0, free_loads = 0, and_loads = 0;
  std::map<size_t, size_t> xh, ah;
  uint32_t steps0 = 0, steps1 = 0;
  std::vector<uint32_t> oa, ob;
  for (size_t i = 0; i < n; ++i) {
    uint32_t c = t.c[i];
    if (c == DEAD_WIRE) continue;
    uint32_t a = t.a[i], b = t.b[i];
    bool is_not = t.type[i] == uint8_t(GateType::Not);
    lev0[c] = std::max(lev0[a], is_not ? 0u : lev0[b]) + 1;
    steps0 = std::max(steps0, lev0[c]);
    ops_of(a, oa);
    if (is_not) ob.clear(); else ops_of(b, ob);
    if (t.type[i] >= 8) {
      std::vector<uint32_t> r = symdiff(oa, ob);
      if (r.size() > KX) {  // materialise the larger folded side, then the other
        if (oa.size() >= ob.size() && oa.size() > 1) { oa = {a}; } else if (ob.size() > 1) { ob = {b}; }
        r = symdiff(oa, ob);
        if (r.size() > KX) { oa = {a}; ob = {b}; r = symdiff(oa, ob); }
      }
      expr[c] = r;
    } else {
      if (oa.size() > KA) oa = {a};
      if (ob.size() > KA) ob = {b};
      and_loads += oa.size() + ob.size();
      ah[oa.size() + ob.size()]++;
      uint32_t l = 0;
      for (uint32_t w : oa) l = std::max(l, lev1[w]);
      for (uint32_t w : ob) l = std::max(l, lev1[w]);
      lev1[c] = l + 1;
      steps1 = std::max(steps1, lev1[c]);
    }
    for (int side = 0; side < 2; ++side) {
      uint32_t x = side ? b : a;
      if (side && is_not) continue;
      const std::vector<uint32_t>& o = side ? ob : oa;
      bool foldable = is_free[x] && fan[x] == 1 && !pinned[x];
      if (foldable && !(o.size() == 1 && o[0] == x)) mat[x] = 0;
    }
  }
  // materialised free gates: levels need a second pass in stream order (lev1 of a kept free gate = max over its operands + 1)
  std::fill(lev1.begin(), lev1.end(), 0);
  steps1 = 0;
  std::vector<uint32_t> width;
  for (size_t i = 0; i < n; ++i) {
    uint32_t c = t.c[i];
    if (c == DEAD_WIRE) continue;
    if (t.type[i] >= 8) {
      if (!mat[c]) continue;
      uint32_t l = 0;
      for (uint32_t w : expr[c]) l = std::max(l, lev1[w]);
      lev1[c] = l + 1;
      ++free_kept;
      free_loads += expr[c].size();
      xh[expr[c].size()]++;
    } else {
      uint32_t a = t.a[i], b = t.b[i];
      std::vector<uint32_t> o1, o2;
      if (is_free[a] && !mat[a]) o1 = expr[a]; else o1 = {a};
      if (is_free[b] && !mat[b]) o2 = expr[b]; else o2 = {b};
      uint32_t l = 0;
      for (uint32_t w : o1) l = std::max(l, lev1[w]);
      for (uint32_t w : o2) l = std::max(l, lev1[w]);
      lev1[c] = l + 1;
    }
    steps1 = std::max(steps1, lev1[c]);
  }
  printf("KX=%u KA=%u: free gates kept %zu of %zu (%.1f%%), steps %u -> %u\n", KX, KA, free_kept, n_free, 100.0 * free_kept / n_free, steps0, steps1);
  printf("label loads: before %zu, after %zu (free %zu + and %zu); label stores: before %zu, after %zu\n", 2 * live, free_loads + and_loads,
         free_loads, and_loads, live, free_kept + n_and);
  printf("free-gate operand histogram:");
  for (auto& kv : xh) printf(" %zu:%zu", kv.first, kv.second);
  printf("\nand-gate operand histogram:");
  for (auto& kv : ah) printf(" %zu:%zu", kv.first, kv.second);
  printf("\n");
  return 0;
}
