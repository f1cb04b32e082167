// Per-component gate counts of a restated circuit, in seconds even for the 11 B-gate verifier.
//
//   g++ -O2 -std=c++17 -I garbled_snark_verifier_amd/csrc tools/gate_counts.cpp -o /tmp/gate_counts
//   /tmp/gate_counts <circuit spec> [depth]
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

struct Counts { uint64_t total = 0, nonfree = 0; };
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
    if (!gate_is_free(g.t)) { cur.nonfree++; self.nonfree++; }
  }
  static std::string name_of(const ComponentKey& k) { size_t p = k.find_first_of("|#"); return k.substr(0, p); }
  static void merge(std::map<std::string, std::pair<uint64_t, Counts>>& dst, const std::map<std::string, std::pair<uint64_t, Counts>>& src) {
    for (auto& kv : src) { auto& d = dst[kv.first]; d.first += kv.second.first; d.second.total += kv.second.second.total; d.second.nonfree += kv.second.second.nonfree; }
  }
  static void merge_tree(Node& dst, const Node& src) {
    dst.calls += src.calls; dst.inc.total += src.inc.total; dst.inc.nonfree += src.inc.nonfree;
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
      me.first += 1; me.second.total += self.total; me.second.nonfree += self.nonfree;
      m.by_name_self = self_by_name;
      m.tree.calls = 1; m.tree.inc = cur;
      cur = save_cur; self = save_self; self_by_name = std::move(save_names); tree = save_tree;
      keys_by_name[name]++;
      it = memo.emplace(key, std::move(m)).first;
    }
    const Memo& m = it->second;
    cur.total += m.inc.total; cur.nonfree += m.inc.nonfree;
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

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: gate_counts <circuit spec> [tree depth]\n"); return 2; }
  const int depth = argc > 2 ? atoi(argv[2]) : 3;
  try {
    NamedCircuit nc = make_circuit(argv[1]);
    CountCtx ctx;
    Node root; root.name = "<root>";
    ctx.tree = &root;
    Wires in = ctx.issue_wires(nc.n_inputs);
    nc.fn(ctx, in);
    root.calls = 1; root.inc = ctx.cur;
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
