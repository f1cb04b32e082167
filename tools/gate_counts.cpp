// Per-component gate counts of a restated circuit, in seconds even for the 11 B-gate verifier.
//
//   g++ -O2 -std=c++17 -I garbled_snark_verifier_amd/csrc tools/gate_counts.cpp -o /tmp/gate_counts
//   /tmp/gate_counts <circuit spec> [depth] [--json] [--verified true|false]
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

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: gate_counts <circuit spec> [tree depth] [--json] [--verified true|false]\n"); return 2; }
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
