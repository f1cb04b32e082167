// Per-component gate counts of a restated circuit, in seconds even for the 11 B-gate verifier.
//
//   g++ -O2 -std=c++17 -I garbled_snark_verifier_amd/csrc tools/gate_counts.cpp -o /tmp/gate_counts
//   /tmp/gate_counts <circuit spec> [depth] [--json] [--verified true|false]
//   /tmp/gate_counts --pairing-csv      the `test_name,total_gates` rows of the reference's examples/pairing_gate_counts.rs (same operations,
//                                       same deterministic inputs: g1 = 5 G, g2 = 7 G2, the Fq2 constant (123, 456), ...), so that first contact
//                                       with cargo is `cargo run --release --example pairing_gate_counts | sort | diff - <(sort this)`
//
// --json prints the schema of the reference's own counter (examples/groth16_gc_gate_count.rs:126-141: circuit_size, gate_count
// {nonfree, free, total, *_formatted, breakdown = the eleven per-GateType counts in discriminant order}, verification_result,
// compressed) so that `cargo run --example groth16_gc_gate_count -- --json [--compressed]` and this tool diff key by key, plus
// "components": per component NAME calls / distinct keys / own gates, and "tree": the inclusive call tree.  verification_result is
// what --verified says (the counting context does not execute the circuit; tests/test_gate_counts.py takes it from the oracle's
// Execute mode), circuit_size.k is null: the restated verifier's size depends on the number of public inputs only, not on k.
//
// The gadget headers are run under a COUNTING context: add_gate counts, and a component (with_named_child) is run once per
// distinct key (component key = name + off-circuit parameters + arity + input length, component_key.rs:16-39) and its counts
// are reused for every later call — gate counts are a function of the key alone (a component's body only depends on its
// off-circuit parameters; which of its gates are dead depends on the caller, but dead gates are counted too:
// streaming_mode.rs:140).  Prints, per component NAME: calls, distinct keys, gates inclusive / exclusive of nested components,
// non-free gates; then the call tree down to [depth].  Used for DESIGN.md's reconciliation table against the reference's
// published 11,174,708,821 (README.md:12).
#include <cstdio>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "gadgets/circuits.hpp"

using namespace gsv;

struct Counts {
  uint64_t total = 0, nonfree = 0;
  uint64_t by_type[11] = {0};
  void add(const Counts& o) { total += o.total; nonfree += o.nonfree; for (int i = 0; i < 11; ++i) by_type[i] += o.by_type[i]; }
};
struct Node { std::string name; uint64_t calls = 0; Counts inc; std::map<std::string, Node> kids; };

struct CountCtx final : CircuitContext {
  WireId next = WIRE_MIN;
  Counts cur;                                  // gates of the component being run (inclusive)
  struct Memo { Counts inc; std::map<std::string, std::pair<uint64_t, Counts>> by_name_self; Node tree; };
  std::map<ComponentKey, Memo> memo;
  std::map<std::string, std::pair<uint64_t, Counts>> self_by_name;  // name -> (calls, exclusive counts) of the current subtree
  std::map<std::string, uint64_t> keys_by_name;
  Node* tree = nullptr;
  Counts self;                                 // exclusive gates of the current component

  WireId issue_wire() override { return next++; }
  void add_gate(const Gate& g) override {
    cur.total++; self.total++;
    cur.by_type[int(g.t)]++; self.by_type[int(g.t)]++;
    if (!gate_is_free(g.t)) { cur.nonfree++; self.nonfree++; }
  }
  static std::string name_of(const ComponentKey& k) { size_t p = k.find_first_of("|#"); return k.substr(0, p); }
  static void merge(std::map<std::string, std::pair<uint64_t, Counts>>& dst, const std::map<std::string, std::pair<uint64_t, Counts>>& src) {
    for (auto& kv : src) { auto& d = dst[kv.first]; d.first += kv.second.first; d.second.add(kv.second.second); }
  }
  static void merge_tree(Node& dst, const Node& src) {
    dst.calls += src.calls; dst.inc.add(src.inc);
    for (auto& kv : src.kids) { Node& d = dst.kids[kv.first]; d.name = kv.first; merge_tree(d, kv.second); }
  }
  Wires with_named_child(const ComponentKey& key, const Wires& inputs, const ChildFn& f, size_t arity) override {
    const std::string name = name_of(key);
    auto it = memo.find(key);
    if (it == memo.end()) {
      // run the body once in a fresh accounting frame
      Counts save_cur = cur, save_self = self;
      auto save_names = std::move(self_by_name);
      Node* save_tree = tree;
      Memo m;
      m.tree.name = name;
      cur = Counts(); self = Counts(); self_by_name.clear(); tree = &m.tree;
      Wires out = f(*this, inputs);
      if (out.size() != arity) gsv_panic("component returned wrong arity: " + name);
      m.inc = cur;
      auto& me = self_by_name[name];
      me.first += 1; me.second.add(self);
      m.by_name_self = self_by_name;
      m.tree.calls = 1; m.tree.inc = cur;
      cur = save_cur; self = save_self; self_by_name = std::move(save_names); tree = save_tree;
      keys_by_name[name]++;
      it = memo.emplace(key, std::move(m)).first;
    }
    const Memo& m = it->second;
    cur.add(m.inc);
    merge(self_by_name, m.by_name_self);
    if (tree) { Node& d = tree->kids[name]; d.name = name; merge_tree(d, m.tree); }
    Wires out(arity);
    for (auto& w : out) w = next++;
    return out;
  }
};

static void print_tree(const Node& n, int depth, int max_depth, uint64_t total) {
  if (depth > max_depth) return;
  std::printf("%*s%-*s calls %8llu  gates %15llu  %6.2f%%  non-free %14llu\n", 2 * depth, "", 62 - 2 * depth, n.name.c_str(), (unsigned long long)n.calls,
              (unsigned long long)n.inc.total, 100.0 * double(n.inc.total) / double(total), (unsigned long long)n.inc.nonfree);
  std::vector<const Node*> kids;
  for (auto& kv : n.kids) kids.push_back(&kv.second);
  std::sort(kids.begin(), kids.end(), [](const Node* a, const Node* b) { return a->inc.total > b->inc.total; });
  for (const Node* k : kids) print_tree(*k, depth + 1, max_depth, total);
}

static std::string human(uint64_t n) {  // format_number of examples/groth16_gc_gate_count.rs:17-27
  char b[32];
  if (n >= 1000000000ull) std::snprintf(b, sizeof b, "%.1fB", double(n) / 1e9);
  else if (n >= 1000000ull) std::snprintf(b, sizeof b, "%.1fM", double(n) / 1e6);
  else if (n >= 1000ull) std::snprintf(b, sizeof b, "%.1fK", double(n) / 1e3);
  else std::snprintf(b, sizeof b, "%llu", (unsigned long long)n);
  return b;
}
static void json_tree(const Node& n, int depth, int max_depth) {
  std::printf("{\"name\": \"%s\", \"calls\": %llu, \"gates\": %llu, \"nonfree\": %llu, \"children\": [", n.name.c_str(), (unsigned long long)n.calls, (unsigned long long)n.inc.total,
              (unsigned long long)n.inc.nonfree);
  if (depth < max_depth) {
    std::vector<const Node*> kids;
    for (auto& kv : n.kids) kids.push_back(&kv.second);
    std::sort(kids.begin(), kids.end(), [](const Node* a, const Node* b) { return a->inc.total > b->inc.total; });
    for (size_t i = 0; i < kids.size(); ++i) { if (i) std::printf(", "); json_tree(*kids[i], depth + 1, max_depth); }
  }
  std::printf("]}");
}


// ---- --pairing-csv: examples/pairing_gate_counts.rs, row for row (the reference prints the rows as its threads finish: compare sorted)
namespace {
using namespace gsv::gadgets;
HFq hfq_inv(const HFq& a) {  // a^(p-2): the host only needs it for 7 G2 in affine form
  const BigU e = HFq::sub_raw(HFq::p(), HFq::from_u64(2)).to_bigu();
  HFq r = HFq::from_u64(1);
  for (size_t i = e.bits(); i-- > 0;) { r = HFq::mul(r, r); if (e.bit(i)) r = HFq::mul(r, a); }
  return r;
}
HFq2 hfq2_inv(const HFq2& a) {
  const HFq n = hfq_inv(HFq::add(HFq::mul(a.c0, a.c0), HFq::mul(a.c1, a.c1)));
  return {HFq::mul(a.c0, n), HFq::neg(HFq::mul(a.c1, n))};
}
uint64_t count_of(const std::function<void(CircuitContext&, const G1Wires&, const G2Wires&)>& body) {
  CountCtx ctx;
  Node root; root.name = "<root>";
  ctx.tree = &root;
  G1Wires g1{ctx.issue_wires(254), ctx.issue_wires(254), ctx.issue_wires(254)};  // Inputs::allocate: G1Projective::new, then G2Projective::new
  G2Wires g2 = pairing::g2_from_wires(ctx.issue_wires(1524));
  body(ctx, g1, g2);
  return ctx.cur.total;  // StreamingResult::gate_count.total_gate_count(): every gate of the execution pass, dead ones included
}
int pairing_csv() {
  // g2_aff = (7 * G2::generator()).into_affine(): 7 Q = 2 (2 Q + Q) + Q in the pairing's own homogeneous projective coordinates (x / z, y / z)
  const HFq2 gx = test_g2_generator_x(), gy = test_g2_generator_y();
  HG2 r{gx, gy, HFq2{HFq::from_u64(1), HFq()}};
  (void)h_double_in_place(r); (void)h_add_in_place(r, gx, gy); (void)h_double_in_place(r); (void)h_add_in_place(r, gx, gy);
  const HFq2 zi = hfq2_inv(r.z), q7x = HFq2::mul(r.x, zi), q7y = HFq2::mul(r.y, zi);
  // on the twist: y^2 = x^3 + b'
  if (HFq::cmp(HFq2::sub(HFq2::sq(q7y), HFq2::add(HFq2::mul(HFq2::sq(q7x), q7x), PairingConst::coeff_b())).c0, HFq()) != 0) { std::fprintf(stderr, "7 G2 is not on the curve\n"); return 1; }
  const std::vector<HEllCoeff> ell7 = h_ell_coeffs(q7x, q7y);
  auto mont = [](uint64_t v) { return fq_as_montgomery_const(BigU(v)); };
  const Fq12 f0 = fq12::one_constant();
  std::printf("test_name,total_gates\n");
  auto row = [](const char* name, uint64_t n) { std::printf("%s,%llu\n", name, (unsigned long long)n); };
  row("fq_mul_montgomery", count_of([&](CircuitContext& c, const G1Wires&, const G2Wires&) {
    (void)fq::mul_montgomery(c, constant_wires(mont(0xFFFFFFFFull), 254), constant_wires(mont(0xFFFFFFFFFFFFFFFFull), 254)); }));
  row("mul_constant_by_fq_montgomery", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires&) {
    (void)fq2::mul_constant_by_fq_montgomery(c, HFq2{HFq::from_u64(123), HFq::from_u64(456)}, g1.x); }));
  row("fq_mul_by_constant_montgomery", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires&) { (void)fq::mul_by_constant_montgomery(c, g1.x, mont(123)); }));
  row("test_double_in_place_montgomery", count_of([&](CircuitContext& c, const G1Wires&, const G2Wires& g2) { (void)pairing::double_in_place_circuit_montgomery(c, g2); }));
  row("test_add_in_place_montgomery", count_of([&](CircuitContext& c, const G1Wires&, const G2Wires& g2) { (void)pairing::add_in_place_montgomery(c, g2, g2); }));
  row("test_mul_by_char_montgomery", count_of([&](CircuitContext& c, const G1Wires&, const G2Wires& g2) { (void)pairing::mul_by_char_montgomery(c, g2); }));
  row("test_ell_montgomery", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires& g2) {
    const std::vector<Fq6> coeffs = pairing::ell_coeffs_montgomery(c, g2);
    (void)pairing::ell_montgomery(c, f0, coeffs[0], g1); }));
  row("test_ell_by_constant_montgomery", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires&) { (void)pairing::ell_eval_const(c, f0, ell7[0], g1); }));
  row("test_ell_coeffs_evaluate_montgomery_fast", count_of([&](CircuitContext& c, const G1Wires&, const G2Wires& g2) { (void)pairing::ell_coeffs_montgomery(c, g2); }));
  row("test_miller_loop_evaluate_montgomery_fast", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires& g2) { (void)pairing::miller_loop_montgomery_fast(c, g1, g2); }));
  row("test_miller_loop", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires&) { (void)pairing::miller_loop_const_q(c, g1, q7x, q7y); }));
  row("test_deserialized_compressed_g1", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires&) {
    const Wires& x = g1.x;
    Wires x2 = fq::square_montgomery(c, x);
    Wires x3 = fq::mul_montgomery(c, x2, x);
    Wires y2 = fq::add(c, x3, constant_wires(mont(3), 254));  // Fq::add with the constant as WIRES (g1::Config::COEFF_B = 3), not add_constant
    Wires y = fq::sqrt_montgomery(c, y2);
    Wires neg_y = fq::neg(c, y);
    (void)gadgets::select(c, y, neg_y, TRUE_WIRE); }));
  row("test_deserialized_compressed_g2", count_of([&](CircuitContext& c, const G1Wires&, const G2Wires& g2) {
    const Fq2& x = g2.x;
    Fq2 x2 = fq2::square_montgomery(c, x);
    Fq2 x3 = fq2::mul_montgomery(c, x2, x);
    Fq2 y2 = fq2::add_constant(c, x3, PairingConst::coeff_b().as_montgomery_const());
    Fq2 y = fq2::sqrt_general_montgomery(c, y2);
    Fq2 neg_y = fq2::neg(c, y);
    (void)gadgets::select(c, y.c[0], neg_y.c[0], TRUE_WIRE);
    (void)gadgets::select(c, y.c[1], neg_y.c[1], TRUE_WIRE); }));
  row("test_multi_miller_loop", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires&) { (void)pairing::multi_miller_loop_const_q(c, {g1}, {{q7x, q7y}}); }));
  row("test_multi_miller_loop_evaluate_montgomery_fast", count_of([&](CircuitContext& c, const G1Wires& g1, const G2Wires& g2) { (void)pairing::multi_miller_loop_montgomery_fast(c, {g1}, {g2}); }));
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: gate_counts <circuit spec> [tree depth] [--json] [--verified true|false]\n       gate_counts --pairing-csv\n"); return 2; }
  if (std::string(argv[1]) == "--pairing-csv") {
    try { return pairing_csv(); } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  }
  int depth = 3;
  bool json = false;
  const char* verified = "null";
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--json") json = true;
    else if (a == "--verified" && i + 1 < argc) verified = std::string(argv[++i]) == "true" ? "true" : "false";
    else depth = atoi(argv[i]);
  }
  try {
    NamedCircuit nc = make_circuit(argv[1]);
    CountCtx ctx;
    Node root; root.name = "<root>";
    ctx.tree = &root;
    Wires in = ctx.issue_wires(nc.n_inputs);
    nc.fn(ctx, in);
    root.calls = 1; root.inc = ctx.cur;
    if (json) {
      const std::string spec = argv[1];
      const bool compressed = spec.rfind("groth16_verify_compressed", 0) == 0;
      const uint64_t total = ctx.cur.total, nonfree = ctx.cur.nonfree, free_g = total - nonfree;
      std::printf("{\n  \"circuit_size\": {\"k\": null, \"constraints\": null},\n  \"gate_count\": {\n    \"nonfree\": %llu,\n    \"nonfree_formatted\": \"%s\",\n    \"free\": %llu,\n    \"free_formatted\": \"%s\",\n"
                  "    \"total\": %llu,\n    \"total_formatted\": \"%s\",\n    \"breakdown\": [", (unsigned long long)nonfree, human(nonfree).c_str(), (unsigned long long)free_g, human(free_g).c_str(),
                  (unsigned long long)total, human(total).c_str());
      for (int i = 0; i < 11; ++i) std::printf("%s%llu", i ? ", " : "", (unsigned long long)ctx.cur.by_type[i]);
      std::printf("]\n  },\n  \"verification_result\": %s,\n  \"compressed\": %s,\n  \"inputs\": %zu,\n  \"root_level_gates\": %llu,\n  \"components\": [\n", verified, compressed ? "true" : "false", nc.n_inputs,
                  (unsigned long long)ctx.self.total);
      std::vector<std::pair<std::string, std::pair<uint64_t, Counts>>> rows(ctx.self_by_name.begin(), ctx.self_by_name.end());
      std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.second.second.total > b.second.second.total; });
      for (size_t i = 0; i < rows.size(); ++i)
        std::printf("    {\"name\": \"%s\", \"calls\": %llu, \"keys\": %llu, \"gates_self\": %llu, \"nonfree_self\": %llu}%s\n", rows[i].first.c_str(), (unsigned long long)rows[i].second.first,
                    (unsigned long long)ctx.keys_by_name[rows[i].first], (unsigned long long)rows[i].second.second.total, (unsigned long long)rows[i].second.second.nonfree, i + 1 < rows.size() ? "," : "");
      std::printf("  ],\n  \"tree\": ");
      json_tree(root, 0, depth);
      std::printf("\n}\n");
      return 0;
    }
    std::printf("circuit: %.60s%s\ninputs %zu  total gates %llu  non-free %llu  (root-level gates outside components: %llu)\n\n", argv[1], std::string(argv[1]).size() > 60 ? "..." : "",
                nc.n_inputs, (unsigned long long)ctx.cur.total, (unsigned long long)ctx.cur.nonfree, (unsigned long long)ctx.self.total);
    std::printf("%-58s %10s %8s %16s %16s %15s\n", "component", "calls", "keys", "gates (self)", "share", "non-free (self)");
    std::vector<std::pair<std::string, std::pair<uint64_t, Counts>>> rows(ctx.self_by_name.begin(), ctx.self_by_name.end());
    std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.second.second.total > b.second.second.total; });
    for (auto& r : rows)
      std::printf("%-58s %10llu %8llu %16llu %15.3f%% %15llu\n", r.first.c_str(), (unsigned long long)r.second.first, (unsigned long long)ctx.keys_by_name[r.first],
                  (unsigned long long)r.second.second.total, 100.0 * double(r.second.second.total) / double(ctx.cur.total), (unsigned long long)r.second.second.nonfree);
    std::printf("\ncall tree (inclusive gates):\n");
    print_tree(root, 0, depth, ctx.cur.total);
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
