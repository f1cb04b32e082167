// Analysis tool (not part of the product): how deep is a circuit in device steps under different fusion limits?
// A device step is one dependency level of the fused op list (compile_program 1.).  The product folds free gates into their readers
// with up to KX = 4 wires per free op and KA = 2 wires per AND input, duplicating a folded expression of <= DT = 2 wires into up to
// DF = 2 readers.  This tool recomputes the level structure (levels only, no records) for other limits, to see how much of the
// depth is owed to free gates that survive the fusion.  AND depth is the floor.
//   g++ -O2 -std=c++17 -I garbled_snark_verifier_amd/csrc tools/depth_stats.cpp -o /tmp/depth_stats && /tmp/depth_stats fq_mul
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "engine/program.hpp"
#include "gadgets/circuits.hpp"

using namespace gsv;

struct Res { uint32_t steps; size_t ands, frees; double and_terms, free_terms; };

static Res analyse(const Trace& t, const std::vector<uint32_t>& inputs, const std::vector<uint32_t>& outputs, uint32_t KX, uint32_t KA, uint32_t DF, uint32_t DT) {
  constexpr uint32_t M = 16;
  struct Expr { uint32_t w[M]; uint8_t n; };
  const size_t n = t.size();
  const uint32_t nw = t.n_wires;
  constexpr uint8_t NOT = uint8_t(GateType::Not);
  std::vector<uint8_t> fan(nw, 0), pinned(nw, 0), isfree(nw, 0), need(nw, 0);
  for (size_t i = 0; i < n; ++i) {
    if (t.c[i] == DEAD_WIRE) continue;
    if (fan[t.a[i]] < 255) ++fan[t.a[i]];
    if (t.type[i] != NOT && fan[t.b[i]] < 255) ++fan[t.b[i]];
    if (t.type[i] >= 8) isfree[t.c[i]] = 1;
  }
  pinned[0] = pinned[1] = 1;
  for (uint32_t w : inputs) pinned[w] = 1;
  for (uint32_t w : outputs) pinned[w] = 1;
  std::vector<Expr> expr(nw);
  std::vector<uint32_t> lev(nw, 0);  // level of the wire when it is materialised
  auto single = [](uint32_t x) { Expr e; e.n = 1; e.w[0] = x; return e; };
  auto resolve = [&](uint32_t x, uint32_t cap) -> Expr {
    if (isfree[x] && !pinned[x]) {
      const Expr& ex = expr[x];
      if (ex.n <= cap && (fan[x] == 1 || (fan[x] <= DF && ex.n <= DT))) return ex;
    }
    need[x] = 1;
    return single(x);
  };
  auto symdiff = [](const Expr& a, const Expr& b, uint32_t* out) -> uint32_t {
    uint32_t i = 0, j = 0, k = 0;
    while (i < a.n || j < b.n) {
      if (j == b.n || (i < a.n && a.w[i] < b.w[j])) out[k++] = a.w[i++];
      else if (i == a.n || b.w[j] < a.w[i]) out[k++] = b.w[j++];
      else { ++i; ++j; }
    }
    return k;
  };
  Res r{0, 0, 0, 0, 0};
  std::vector<uint32_t> free_n(nw, 0);
  std::vector<uint32_t> cons(nw, DEAD_WIRE);
  for (size_t i = 0; i < n; ++i) {
    if (t.c[i] == DEAD_WIRE) continue;
    cons[t.a[i]] = uint32_t(i);
    if (t.type[i] != NOT) cons[t.b[i]] = uint32_t(i);
  }
  std::vector<uint8_t> absorbed(n, 0);
  constexpr uint8_t XOR = uint8_t(GateType::Xor), XNOR = uint8_t(GateType::Xnor);
  for (size_t i = 0; i < n; ++i) {
    const uint32_t c = t.c[i];
    if (c == DEAD_WIRE || absorbed[i]) continue;
    const uint8_t ty = t.type[i];
    if (ty >= 8) {
      Expr ea = resolve(t.a[i], KX), eb;
      if (ty == NOT) eb.n = 0; else eb = resolve(t.b[i], KX);
      uint32_t w[2 * M];
      uint32_t k = symdiff(ea, eb, w);
      if (k > KX) {
        if (ea.n >= eb.n && ea.n > 1) { ea = single(t.a[i]); need[t.a[i]] = 1; } else { eb = single(t.b[i]); need[t.b[i]] = 1; }
        k = symdiff(ea, eb, w);
        if (k > KX) { ea = single(t.a[i]); need[t.a[i]] = 1; eb = single(t.b[i]); need[t.b[i]] = 1; k = symdiff(ea, eb, w); }
      }
      Expr e; e.n = uint8_t(k);
      uint32_t l = 0;
      for (uint32_t q = 0; q < k; ++q) { e.w[q] = w[q]; l = std::max(l, lev[w[q]]); }
      expr[c] = e;
      lev[c] = l + 1;
      free_n[c] = k;
    } else {
      const Expr ea = resolve(t.a[i], KA), eb = resolve(t.b[i], KA);
      uint32_t l = 0;
      for (uint32_t q = 0; q < ea.n; ++q) l = std::max(l, lev[ea.w[q]]);
      for (uint32_t q = 0; q < eb.n; ++q) l = std::max(l, lev[eb.w[q]]);
      uint32_t out = c;
      // the product's output fold (fuse_trace): the single reader of this AND is a XOR/XNOR with an operand that already exists
      if (fan[c] == 1 && !pinned[c]) {
        const size_t j = cons[c];
        const uint8_t tj = t.type[j];
        if ((tj == XOR || tj == XNOR) && t.c[j] != DEAD_WIRE) {
          const uint32_t y = t.a[j] == c ? t.b[j] : t.a[j];
          if (y != c && y < c) {
            const Expr ey = resolve(y, 1);
            for (uint32_t q = 0; q < ey.n; ++q) l = std::max(l, lev[ey.w[q]]);
            out = t.c[j];
            absorbed[j] = 1;
            isfree[out] = 0;
            r.and_terms += ey.n;
          }
        }
      }
      lev[out] = l + 1;
      r.ands++; r.and_terms += ea.n + eb.n;
      r.steps = std::max(r.steps, lev[out]);
    }
  }
  for (uint32_t w = 2; w < nw; ++w)
    if (isfree[w] && (need[w] || pinned[w])) { r.frees++; r.free_terms += free_n[w]; r.steps = std::max(r.steps, lev[w]); }
  return r;
}

int main(int argc, char** argv) {
  const std::string spec = argc > 1 ? argv[1] : "fq_mul";
  RecordMode mode;
  NamedCircuit nc = make_circuit(spec);
  StreamingRunner run(mode, nc.n_inputs, nc.fn);
  std::vector<uint32_t> inputs, outputs;
  for (WireId w : run.prepare()) inputs.push_back(mode.define_input(w));
  for (WireId w : run.execute()) outputs.push_back(mode.current(w));
  const Trace& t = mode.trace();
  std::printf("%s: %zu gates\n", spec.c_str(), t.size());
  const uint32_t cfg[][4] = {{4, 2, 2, 2}, {4, 3, 2, 2}, {4, 3, 2, 3}, {4, 3, 4, 3}, {4, 3, 255, 3}, {4, 4, 2, 2}, {4, 4, 2, 4}, {4, 4, 255, 4}, {8, 8, 255, 8}, {16, 16, 255, 16}};
  for (const auto& c : cfg) {
    const Res r = analyse(t, inputs, outputs, c[0], c[1], c[2], c[3]);
    std::printf("  free op <= %2u wires, AND input <= %2u, duplicate <= %2u-wire expressions into <= %3u readers: %7u steps, %9zu AND ops (%.2f wires), %9zu free ops (%.2f wires)\n",
                c[0], c[1], c[3], c[2], r.steps, r.ands, r.and_terms / std::max<size_t>(1, r.ands), r.frees, r.free_terms / std::max<size_t>(1, r.frees));
  }
  return 0;
}
