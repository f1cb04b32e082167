// The SEQUENCE of component calls of a restated circuit with its dataflow, as a chain of 64-bit hashes — the C++ half of
// tests/test_call_sequence.py, whose other half is the independent Python restatement (tests/ref_call_sequence.py over
// tests/ref_verifier_count.py / ref_gadgets.py, written from the Rust source and sharing no code with these gadget headers).
//
//   g++ -O2 -std=c++17 -I garbled_snark_verifier_amd/csrc tools/call_sequence.cpp -o /tmp/call_sequence
//   /tmp/call_sequence <circuit spec> <unit name> [<unit name> ...]      one line per event: "<unit name> <hash>"
//
// Every wire carries a provenance hash.  Constants (FALSE, TRUE) share one value — which constant a wire is depends on the instance,
// what is wired where does not; primary input i has hash(IN, i).  A gate outside the unit components hashes its operands' values, in
// order, with its type into its output (an in-place NOT included).  A call of a UNIT component (named on the command line; the first
// one met on the way down — their bodies are not run) is an event: hash(name, arity, the provenance of every input wire in the
// component's input order), and output j of the call carries hash(event, j).  Components that are not units are run like inline code.
// Two walks that print the same lines have issued the same unit calls in the same order with the same wiring between them and the same
// glue gates around them: groth16.rs:57-110,250-268, pairing.rs:945-1007, final_exponentiation.rs:99-135 as sequences, not as counts.
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "gadgets/circuits.hpp"

using namespace gsv;

static inline uint64_t mix(uint64_t x, uint64_t y) {
  uint64_t z = x * 0x9E3779B97F4A7C15ull + y;
  z ^= z >> 32; z *= 0xD6E8FEB86659FD93ull; z ^= z >> 32;
  return z;
}
static uint64_t name_hash(const std::string& s) {  // FNV-1a 64
  uint64_t h = 0xcbf29ce484222325ull;
  for (unsigned char ch : s) { h ^= ch; h *= 0x100000001b3ull; }
  return h;
}
constexpr uint64_t K_CONST = 0x1111111111111111ull, K_IN = 0x2222222222222222ull, K_GATE = 0x100, K_OUT = 0x3000;

struct SeqCtx final : CircuitContext {
  std::vector<uint64_t> prov{K_CONST, K_CONST};
  std::set<std::string> units;
  uint64_t glue_gates = 0, events = 0;
  WireId issue_wire() override { prov.push_back(0); return WireId(prov.size() - 1); }
  void add_gate(const Gate& g) override {
    ++glue_gates;
    if (g.c == UNREACHABLE) return;
    prov[g.c] = mix(mix(prov[g.a], prov[g.b]), K_GATE + uint64_t(g.t));
  }
  static std::string name_of(const ComponentKey& k) { size_t p = k.find_first_of("|#"); return k.substr(0, p); }
  Wires with_named_child(const ComponentKey& key, const Wires& inputs, const ChildFn& f, size_t arity) override {
    const std::string name = name_of(key);
    if (!units.count(name)) {
      Wires out = f(*this, inputs);
      if (out.size() != arity) gsv_panic("component returned wrong arity: " + name);
      return out;
    }
    uint64_t e = mix(name_hash(name), arity);
    for (WireId w : inputs) e = mix(e, prov[w]);
    std::printf("%s %016llx\n", name.c_str(), (unsigned long long)e);
    ++events;
    Wires out(arity);
    for (size_t j = 0; j < arity; ++j) { out[j] = issue_wire(); prov[out[j]] = mix(e, K_OUT + j); }
    return out;
  }
};

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: call_sequence <circuit spec> <unit name> ...\n"); return 2; }
  try {
    NamedCircuit nc = make_circuit(argv[1]);
    SeqCtx ctx;
    for (int i = 2; i < argc; ++i) ctx.units.insert(argv[i]);
    Wires in = ctx.issue_wires(nc.n_inputs);
    for (size_t i = 0; i < in.size(); ++i) ctx.prov[in[i]] = mix(K_IN, i);
    Wires out = nc.fn(ctx, in);
    uint64_t e = mix(name_hash("<outputs>"), out.size());
    for (WireId w : out) e = mix(e, ctx.prov[w]);
    std::printf("<outputs> %016llx\n", (unsigned long long)e);
    std::fprintf(stderr, "call_sequence: %llu events, %llu gates outside the units, %zu wires\n", (unsigned long long)ctx.events, (unsigned long long)ctx.glue_gates, ctx.prov.size());
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
