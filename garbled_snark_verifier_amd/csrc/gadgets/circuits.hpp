// Named root circuits: the BASELINE.json configs and the shapes of the reference's own tests, as
// (n_inputs, closure) pairs that any CircuitMode can be run under.
//   u254_add            BASELINE config 1   (src/gadgets/bigint/add.rs:9-26 on 254-bit operands)
//   fq_mul              BASELINE config 2   (src/gadgets/bn254/fp254impl.rs:219-230)
//   fq12_mul            BASELINE config 3   (tests/fq12_mul_e2e.rs:166-173)
//   fq12_mul_chain:K    Groth16-shaped SYNTHETIC for config 4: r <- Fq12::mul(r, b), K times
//   fq12_square, fq12_cyclotomic_square   Fq12::square_montgomery (fq12.rs:311-324), cyclotomic_square_montgomery (fq12.rs:326-392)
//   fq12_sqmul, fq12_sqmul_chain:K        square-and-multiply link r <- Fq12::mul(Fq12::square(r), b)
//   fq_inverse, fq2_inverse, fq12_inverse, fq12_frobenius:I, fq12_conjugate, final_exp   (bn254_ext.hpp)
//   g2_double, g2_add, g2_mul_by_char, ell_eval, ell_const:K, miller_loop               (bn254_pairing.hpp)
//   g1_add, g1_scalar_mul:W, g1_to_affine, fq_sqrt, fq2_sqrt, groth16_verify:<vk hex>, groth16_verify_compressed:<vk hex>   (bn254_groth16.hpp)
//   fq_complex          tests/streaming_evaluate.rs:401-407  ((a^2)*b + a)
//   gate:T              tests/streaming_evaluate.rs:136-213  (one gate of discriminant T at the root)
//   driver_mix          credits / dead-gate / pass-through / constant edge cases (circuit/mod.rs:419-836 shapes)
//   random_circuit:SEED pseudo-random DAG with nested components for differential tests
#pragma once
#include <algorithm>
#include <memory>
#include <string>

#include "bn254_groth16.hpp"

namespace gsv {

struct NamedCircuit {
  size_t n_inputs = 0;
  size_t n_outputs = 0;
  CircuitFn fn;
  // Mini-circuits that call, once each, components the circuit will instantiate with constants of its own (the 178 line functions
  // of a verifying key's gamma and delta).  A plan builder may run them on other threads to record those units side by side ahead of
  // the driver; they change nothing in the circuit's gate stream (a unit missing from the warm-up is simply recorded by the driver).
  struct Warmup { size_t n_inputs; CircuitFn fn; };
  std::vector<Warmup> warmups;
};

// G2 generator of BN254 (affine, standard form): the constant Q of the pairing test circuits
inline gadgets::HFq2 test_g2_generator_x() {
  return gadgets::hfq2_hex("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed", "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2");
}
inline gadgets::HFq2 test_g2_generator_y() {
  return gadgets::hfq2_hex("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa", "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b");
}

namespace detail {
inline Wires driver_mix(CircuitContext& c, const Wires& in) {
  // in: 6 wires.  Exercises: a dead gate (zero fan-out output), a constant-input AND (not folded),
  // a component whose outputs are {input pass-through, constant, internal}, same wire as a and b,
  // a child that ignores one of its inputs, nested children.
  using namespace gadgets;
  WireId dead = c.issue_wire();
  c.add_gate(Gate::and_(in[0], in[1], dead));  // never read: credits 0 -> UNREACHABLE, gate_id still consumed
  WireId k = c.issue_wire();
  c.add_gate(Gate::and_(in[0], FALSE_WIRE, k));  // AND with constant still costs a ciphertext
  WireId sq = c.issue_wire();
  c.add_gate(Gate::or_(in[2], in[2], sq));  // same wire twice: two credits
  Wires child_in = {in[3], in[4], in[5], k};
  Wires ch = component(c, KeyBuilder("test::mixed_outputs"), child_in, 4, [](CircuitContext& cc, const Wires& x) -> Wires {
    WireId t = cc.issue_wire();
    cc.add_gate(Gate::nimp(x[0], x[1], t));  // x[2] is never read by the child
    WireId u = cc.issue_wire();
    cc.add_gate(Gate::xnor(t, x[3], u));
    Wires inner = component(cc, KeyBuilder("test::inner"), Wires{t, u}, 2, [](CircuitContext& c3, const Wires& y) -> Wires {
      WireId p = c3.issue_wire(), q = c3.issue_wire();
      c3.add_gate(Gate::and_variant(y[0], y[1], p, true, true, false));
      c3.add_gate(Gate::xor_(y[0], TRUE_WIRE, q));  // q is dropped by the caller below -> dead inside the child
      return {p, q};
    });
    return {x[0], TRUE_WIRE, u, inner[0]};
  });
  Wires sum = add(c, Wires{ch[0], ch[2], sq}, Wires{ch[3], in[1], k});
  Wires nocarry = add_without_carry(c, Wires{sum[0], sum[1]}, Wires{sum[2], sum[3]});  // top carry dead
  WireId sel = selector(c, nocarry[0], nocarry[1], ch[1]);
  return {sel, ch[1], in[5], sum[3]};
}

// Deterministic pseudo-random circuit for differential tests: every gate type, constants as operands, the same
// wire on both inputs, wires that are never read (dead gates), nested components whose outputs mix inputs,
// constants and internal wires.  splitmix64 keeps the structure identical for every mode that runs it.
struct Mix64 {
  uint64_t s;
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  uint32_t below(uint32_t n) { return uint32_t(next() % n); }
};
inline Wires random_block(CircuitContext& c, const Wires& in, uint64_t seed, uint32_t n_gates, int depth) {
  Mix64 rng{seed};
  Wires pool = in;
  auto pick = [&]() -> WireId {
    uint32_t r = rng.below(20);
    if (r == 0) return FALSE_WIRE;
    if (r == 1) return TRUE_WIRE;
    // favour recent wires so that chains (depth) and early deaths both occur
    uint32_t span = std::min<uint32_t>(uint32_t(pool.size()), 1u + rng.below(24));
    return (rng.below(3) == 0) ? pool[rng.below(uint32_t(pool.size()))] : pool[pool.size() - 1 - rng.below(span)];
  };
  for (uint32_t g = 0; g < n_gates; ++g) {
    if (depth > 0 && rng.below(40) == 0) {
      Wires ci;
      uint32_t k = 2 + rng.below(5);
      for (uint32_t i = 0; i < k; ++i) ci.push_back(pick());
      uint64_t child_seed = rng.next();
      uint32_t child_gates = 5 + rng.below(30);
      uint32_t arity = 1 + rng.below(4);
      ComponentKey key = KeyBuilder("test::random_block").param_usize("seed", child_seed).param_usize("gates", child_gates).finish(arity, ci.size());
      ChildFn fn = [child_seed, child_gates, arity, depth](CircuitContext& cc, const Wires& x) -> Wires {
        Wires all = random_block(cc, x, child_seed, child_gates, depth - 1);
        Mix64 r2{child_seed ^ 0xABCDEFull};
        Wires out;
        for (uint32_t i = 0; i < arity; ++i) {
          uint32_t sel = r2.below(8);
          if (sel == 0) out.push_back(TRUE_WIRE);
          else if (sel == 1) out.push_back(x[r2.below(uint32_t(x.size()))]);  // input passed through as an output
          else out.push_back(all[all.size() - 1 - r2.below(std::min<uint32_t>(uint32_t(all.size()), 6u))]);
        }
        return out;
      };
      Wires o = c.with_named_child(key, ci, fn, arity);
      // (the metadata pass sees mock outputs where the execution pass may see constants or passed-through inputs:
      //  the generator must draw the same random numbers in both passes, so it never looks at the wire ids)
      for (WireId w : o) if (rng.below(4) != 0) pool.push_back(w);  // some child outputs are dropped
      continue;
    }
    GateType t = GateType(rng.below(10));  // And..Xnor (the in-place Not is covered by gate:10)
    WireId a = pick(), b = (rng.below(16) == 0) ? a : pick();
    WireId o = c.issue_wire();
    c.add_gate(Gate::make(t, a, b, o));
    if (rng.below(10) != 0) pool.push_back(o);  // ~10 % of the gates are never read: dead unless chosen as output
  }
  return pool;
}
}  // namespace detail

namespace detail {
// One warm-up per constant line function of the Miller loop's two fixed G2 points (pairing.rs:923-942 as called from 944-1007):
// inputs f (Fq12) and p (G1), every output live — the shape multi_miller_loop_groth16_evaluate_montgomery_fast calls them in.
inline void add_ell_warmups(NamedCircuit& nc, const gadgets::HFq2& q1x, const gadgets::HFq2& q1y, const gadgets::HFq2& q2x, const gadgets::HFq2& q2y) {
  using namespace gadgets;
  for (int which = 0; which < 2; ++which) {
    auto ell = std::make_shared<std::vector<HEllCoeff>>(which == 0 ? h_ell_coeffs(q1x, q1y) : h_ell_coeffs(q2x, q2y));
    for (size_t k = 0; k < ell->size(); ++k)
      nc.warmups.push_back({3048 + 762, [ell, k](CircuitContext& c, const Wires& in) {
        G1Wires p{slice(in, 3048, 3302), slice(in, 3302, 3556), slice(in, 3556, 3810)};
        return pairing::ell_by_constant_montgomery(c, Fq12::from_wires(slice(in, 0, 3048)), (*ell)[k], p).to_wires();
      }});
  }
}
}  // namespace detail

inline NamedCircuit make_circuit(const std::string& spec) {
  using namespace gadgets;
  std::string name = spec;
  uint64_t param = 0;
  bool has_param = false;
  size_t colon = spec.find(':');
  if (colon != std::string::npos) {
    name = spec.substr(0, colon);
    if (name != "groth16_verify" && name != "groth16_verify_compressed") {  // (its parameter is a hex blob, parsed below)
      param = std::stoull(spec.substr(colon + 1));
      has_param = true;
    }
  }
  NamedCircuit nc;
  auto two_fq = [&](std::function<Wires(CircuitContext&, const Wires&, const Wires&)> op) {
    nc.n_inputs = 508; nc.n_outputs = 254;
    nc.fn = [op](CircuitContext& c, const Wires& in) { return op(c, slice(in, 0, 254), slice(in, 254, 508)); };
  };
  auto one_fq = [&](std::function<Wires(CircuitContext&, const Wires&)> op) {
    nc.n_inputs = 254; nc.n_outputs = 254;
    nc.fn = [op](CircuitContext& c, const Wires& in) { return op(c, in); };
  };
  if (name == "u254_add" || name == "bigint_add") {
    size_t n = (name == "u254_add") ? 254 : size_t(param);
    nc.n_inputs = 2 * n; nc.n_outputs = n + 1;
    nc.fn = [n](CircuitContext& c, const Wires& in) { return add(c, slice(in, 0, n), slice(in, n, 2 * n)); };
  } else if (name == "bigint_sub") {
    size_t n = size_t(param);
    nc.n_inputs = 2 * n; nc.n_outputs = n + 1;
    nc.fn = [n](CircuitContext& c, const Wires& in) { return sub(c, slice(in, 0, n), slice(in, n, 2 * n)); };
  } else if (name == "bigint_mul") {
    size_t n = size_t(param);
    nc.n_inputs = 2 * n; nc.n_outputs = 2 * n;
    nc.fn = [n](CircuitContext& c, const Wires& in) { return mul(c, slice(in, 0, n), slice(in, n, 2 * n)); };
  } else if (name == "fq_add") { two_fq([](CircuitContext& c, const Wires& a, const Wires& b) { return fq::add(c, a, b); });
  } else if (name == "fq_sub") { two_fq([](CircuitContext& c, const Wires& a, const Wires& b) { return fq::sub(c, a, b); });
  } else if (name == "fq_mul") { two_fq([](CircuitContext& c, const Wires& a, const Wires& b) { return fq::mul_montgomery(c, a, b); });
  } else if (name == "fq_neg") { one_fq([](CircuitContext& c, const Wires& a) { return fq::neg(c, a); });
  } else if (name == "fq_double") { one_fq([](CircuitContext& c, const Wires& a) { return fq::double_(c, a); });
  } else if (name == "fq_half") { one_fq([](CircuitContext& c, const Wires& a) { return fq::half(c, a); });
  } else if (name == "fq_triple") { one_fq([](CircuitContext& c, const Wires& a) { return fq::triple(c, a); });
  } else if (name == "fq_div6") { one_fq([](CircuitContext& c, const Wires& a) { return fq::div6(c, a); });
  } else if (name == "fq_complex") {
    two_fq([](CircuitContext& c, const Wires& a, const Wires& b) {
      Wires a2 = fq::square_montgomery(c, a);
      Wires a2b = fq::mul_montgomery(c, a2, b);
      return fq::add(c, a2b, a);
    });
  } else if (name == "fq2_mul") {
    nc.n_inputs = 1016; nc.n_outputs = 508;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      return fq2::mul_montgomery(c, Fq2::from_wires(slice(in, 0, 508)), Fq2::from_wires(slice(in, 508, 1016))).to_wires();
    };
  } else if (name == "fq6_mul") {
    nc.n_inputs = 3048; nc.n_outputs = 1524;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      return fq6::mul_montgomery(c, Fq6::from_wires(slice(in, 0, 1524)), Fq6::from_wires(slice(in, 1524, 3048))).to_wires();
    };
  } else if (name == "fq12_mul" || name == "fq12_mul_chain") {
    size_t k = (name == "fq12_mul") ? 1 : size_t(param);
    if (k == 0) gsv_panic("fq12_mul_chain: K must be >= 1");
    nc.n_inputs = 6096; nc.n_outputs = 3048;
    nc.fn = [k](CircuitContext& c, const Wires& in) {
      Fq12 r = Fq12::from_wires(slice(in, 0, 3048));
      Fq12 b = Fq12::from_wires(slice(in, 3048, 6096));
      for (size_t i = 0; i < k; ++i) r = fq12::mul_montgomery(c, r, b);
      return r.to_wires();
    };
  } else if (name == "fq12_square" || name == "fq12_cyclotomic_square") {
    const bool cyc = name == "fq12_cyclotomic_square";
    nc.n_inputs = 3048; nc.n_outputs = 3048;
    nc.fn = [cyc](CircuitContext& c, const Wires& in) {
      Fq12 a = Fq12::from_wires(in);
      return (cyc ? fq12::cyclotomic_square_montgomery(c, a) : fq12::square_montgomery(c, a)).to_wires();
    };
  } else if (name == "fq12_sqmul" || name == "fq12_sqmul_chain") {
    // square-and-multiply link (the shape of the Miller loop / final exponentiation): r <- Fq12::mul(Fq12::square(r), b), K times
    size_t k = (name == "fq12_sqmul") ? 1 : size_t(param);
    if (k == 0) gsv_panic("fq12_sqmul_chain: K must be >= 1");
    nc.n_inputs = 6096; nc.n_outputs = 3048;
    nc.fn = [k](CircuitContext& c, const Wires& in) {
      Fq12 r = Fq12::from_wires(slice(in, 0, 3048));
      Fq12 b = Fq12::from_wires(slice(in, 3048, 6096));
      for (size_t i = 0; i < k; ++i) r = fq12::mul_montgomery(c, fq12::square_montgomery(c, r), b);
      return r.to_wires();
    };
  } else if (name == "fq_inverse") {  // Fq::inverse_montgomery (fp254impl.rs:333-678)
    nc.n_inputs = 254; nc.n_outputs = 254;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq::inverse_montgomery(c, in); };
  } else if (name == "fq2_inverse") {
    nc.n_inputs = 508; nc.n_outputs = 508;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq2::inverse_montgomery(c, Fq2::from_wires(in)).to_wires(); };
  } else if (name == "fq12_inverse") {
    nc.n_inputs = 3048; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq12::inverse_montgomery(c, Fq12::from_wires(in)).to_wires(); };
  } else if (name == "fq12_frobenius") {  // fq12_frobenius:I, I in 1..3
    if (!has_param || param < 1 || param > 3) gsv_panic("fq12_frobenius:I needs I in 1..3");
    const size_t i = size_t(param);
    nc.n_inputs = 3048; nc.n_outputs = 3048;
    nc.fn = [i](CircuitContext& c, const Wires& in) { return fq12::frobenius_montgomery(c, Fq12::from_wires(in), i).to_wires(); };
  } else if (name == "fq12_conjugate") {
    nc.n_inputs = 3048; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq12::conjugate(c, Fq12::from_wires(in)).to_wires(); };
  } else if (name == "final_exp") {  // final_exponentiation_montgomery (final_exponentiation.rs:99-135), ~2.9 B gates
    nc.n_inputs = 3048; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq12::final_exponentiation_montgomery(c, Fq12::from_wires(in)).to_wires(); };
  } else if (name == "g2_double") {  // pairing.rs:359-407: r -> (2r, line coefficients)
    nc.n_inputs = 1524; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      G2Step s = pairing::double_in_place_circuit_montgomery(c, pairing::g2_from_wires(in));
      return concat(pairing::g2_to_wires(s.r), s.coeffs.to_wires());
    };
  } else if (name == "g2_add") {  // pairing.rs:409-464: (r, q affine) -> (r + q, line coefficients)
    nc.n_inputs = 3048; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      G2Step s = pairing::add_in_place_montgomery(c, pairing::g2_from_wires(slice(in, 0, 1524)), pairing::g2_from_wires(slice(in, 1524, 3048)));
      return concat(pairing::g2_to_wires(s.r), s.coeffs.to_wires());
    };
  } else if (name == "g2_mul_by_char") {
    nc.n_inputs = 1524; nc.n_outputs = 1524;
    nc.fn = [](CircuitContext& c, const Wires& in) { return pairing::g2_to_wires(pairing::mul_by_char_montgomery(c, pairing::g2_from_wires(in))); };
  } else if (name == "ell_eval") {  // pairing.rs:160-171: f, coeffs (Fq6), p.x, p.y
    nc.n_inputs = 3048 + 1524 + 508; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      G1Wires p{slice(in, 4572, 4826), slice(in, 4826, 5080), Wires()};
      return pairing::ell_montgomery(c, Fq12::from_wires(slice(in, 0, 3048)), Fq6::from_wires(slice(in, 3048, 4572)), p).to_wires();
    };
  } else if (name == "ell_const") {  // pairing.rs:923-942 with the K-th line coefficient of the G2 generator: ell_const:K
    const size_t k = has_param ? size_t(param) : 0;
    nc.n_inputs = 3048 + 762; nc.n_outputs = 3048;
    nc.fn = [k](CircuitContext& c, const Wires& in) {
      static const std::vector<HEllCoeff> ell = h_ell_coeffs(test_g2_generator_x(), test_g2_generator_y());
      if (k >= ell.size()) gsv_panic("ell_const: coefficient index out of range");
      G1Wires p{slice(in, 3048, 3302), slice(in, 3302, 3556), slice(in, 3556, 3810)};
      return pairing::ell_by_constant_montgomery(c, Fq12::from_wires(slice(in, 0, 3048)), ell[k], p).to_wires();
    };
  } else if (name == "miller_loop") {  // pairing.rs:944-1007 with q1 = G2 generator, q2 = -generator; inputs p1, p2, p3 (x, y, z), q3 (x, y, z)
    nc.n_inputs = 3 * 762 + 1524; nc.n_outputs = 3048;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      auto g1 = [&](size_t o) { return G1Wires{slice(in, o, o + 254), slice(in, o + 254, o + 508), slice(in, o + 508, o + 762)}; };
      const HFq2 gx = test_g2_generator_x(), gy = test_g2_generator_y();
      return pairing::multi_miller_loop_groth16_evaluate_montgomery_fast(c, g1(0), g1(762), g1(1524), gx, gy, gx, HFq2::neg(gy), pairing::g2_from_wires(slice(in, 2286, 3810))).to_wires();
    };
    detail::add_ell_warmups(nc, test_g2_generator_x(), test_g2_generator_y(), test_g2_generator_x(), HFq2::neg(test_g2_generator_y()));
  } else if (name == "g1_add") {  // g1.rs:159-235
    nc.n_inputs = 1524; nc.n_outputs = 762;
    nc.fn = [](CircuitContext& c, const Wires& in) { return g1_to_wires(g1::add_montgomery(c, g1_from_wires(slice(in, 0, 762)), g1_from_wires(slice(in, 762, 1524)))); };
  } else if (name == "g1_scalar_mul") {  // g1.rs:309-368 with base = the G1 generator (1, 2); window W = param (the verifier uses 10)
    const size_t W = has_param ? size_t(param) : 10;
    if (W < 1 || W > 12) gsv_panic("g1_scalar_mul:W needs W in 1..12");
    nc.n_inputs = 254; nc.n_outputs = 762;
    nc.fn = [W](CircuitContext& c, const Wires& in) { return g1_to_wires(g1::scalar_mul_by_constant_base_montgomery(c, in, HG1::from_affine(HFq::from_u64(1), HFq::from_u64(2)), W)); };
  } else if (name == "g1_mux_add") {
    // plan test shape: constant tables behind multiplexers (unit calls whose inputs are mostly the constant wires), additions of
    // wire points and of a constant point.  in: 4 selector bits, a point p; out: ((T[s0 s1] + T'[s2 s3]) + p) + G
    nc.n_inputs = 4 + 762; nc.n_outputs = 762;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      const HG1 g = HG1::from_affine(HFq::from_u64(1), HFq::from_u64(2));
      std::vector<G1Wires> t1, t2;
      HG1 p = HG1::identity();
      for (int i = 0; i < 4; ++i) { t1.push_back(g1_new_constant_montgomery(p)); p = HG1::add(p, g); }
      for (int i = 0; i < 4; ++i) { t2.push_back(g1_new_constant_montgomery(p)); p = HG1::add(p, p); }
      G1Wires a = g1::multiplexer(c, t1, slice(in, 0, 2), 2), b = g1::multiplexer(c, t2, slice(in, 2, 4), 2);
      G1Wires r = g1::add_montgomery(c, g1::add_montgomery(c, a, b), g1_from_wires(slice(in, 4, 766)));
      return g1_to_wires(g1::add_montgomery(c, r, g1_new_constant_montgomery(g)));
    };
  } else if (name == "g1_to_affine") {  // groth16.rs:26-48
    nc.n_inputs = 762; nc.n_outputs = 762;
    nc.fn = [](CircuitContext& c, const Wires& in) { return g1_to_wires(groth16::projective_to_affine_montgomery(c, g1_from_wires(in))); };
  } else if (name == "groth16_verify") {  // groth16.rs:58-110; inputs: public scalars, A, B, C (CircuitInput order, groth16.rs:290-318)
    if (colon == std::string::npos) gsv_panic("groth16_verify needs a verifying key: groth16_verify:<hex>");
    size_t n_pub = 0;
    auto vk = std::make_shared<groth16::VerifyingKey>(groth16::vk_from_hex(spec.substr(colon + 1), &n_pub));
    nc.n_inputs = n_pub * 254 + 762 + 1524 + 762; nc.n_outputs = 1;
    nc.fn = [vk, n_pub](CircuitContext& c, const Wires& in) {
      std::vector<Wires> pub;
      for (size_t i = 0; i < n_pub; ++i) pub.push_back(slice(in, i * 254, i * 254 + 254));
      const size_t o = n_pub * 254;
      return Wires{groth16::verify(c, pub, g1_from_wires(slice(in, o, o + 762)), pairing::g2_from_wires(slice(in, o + 762, o + 2286)), g1_from_wires(slice(in, o + 2286, o + 3048)), *vk)};
    };
    detail::add_ell_warmups(nc, vk->gamma_x, HFq2::neg(vk->gamma_y), vk->delta_x, HFq2::neg(vk->delta_y));  // groth16.rs:58-110 hands -gamma, -delta to the Miller loop
  } else if (name == "groth16_verify_compressed") {  // groth16.rs:250-268; inputs: public scalars, (A.x, flag), (B.x, flag), (C.x, flag) (groth16.rs:410-421)
    if (colon == std::string::npos) gsv_panic("groth16_verify_compressed needs a verifying key: groth16_verify_compressed:<hex>");
    size_t n_pub = 0;
    auto vk = std::make_shared<groth16::VerifyingKey>(groth16::vk_from_hex(spec.substr(colon + 1), &n_pub));
    nc.n_inputs = n_pub * 254 + 255 + 509 + 255; nc.n_outputs = 1;
    nc.fn = [vk, n_pub](CircuitContext& c, const Wires& in) {
      std::vector<Wires> pub;
      for (size_t i = 0; i < n_pub; ++i) pub.push_back(slice(in, i * 254, i * 254 + 254));
      const size_t o = n_pub * 254;
      return Wires{groth16::verify_compressed(c, pub, slice(in, o, o + 254), in[o + 254], Fq2::from_wires(slice(in, o + 255, o + 763)), in[o + 763],
                                              slice(in, o + 764, o + 1018), in[o + 1018], *vk)};
    };
    detail::add_ell_warmups(nc, vk->gamma_x, HFq2::neg(vk->gamma_y), vk->delta_x, HFq2::neg(vk->delta_y));
  } else if (name == "fq_sqrt") {  // fq.rs:290-299
    one_fq([](CircuitContext& c, const Wires& a) { return fq::sqrt_montgomery(c, a); });
  } else if (name == "fq2_sqrt") {  // fq2.rs:425-446
    nc.n_inputs = 508; nc.n_outputs = 508;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq2::sqrt_general_montgomery(c, Fq2::from_wires(in)).to_wires(); };
  } else if (name == "fq_addmul") {  // (a + b) * b: glue followed by a component, for the C-ABI plan recorder test
    nc.n_inputs = 508; nc.n_outputs = 254;
    nc.fn = [](CircuitContext& c, const Wires& in) { return fq::mul_montgomery(c, fq::add(c, slice(in, 0, 254), slice(in, 254, 508)), slice(in, 254, 508)); };
  } else if (name == "fq12_mix") {
    // plan test shape: units (Fq12 square / mul) with glue between them, a unit output reused much later, and a final unit
    // whose upper half is never read (its gates are dead: a second liveness pattern of the same component)
    nc.n_inputs = 6096; nc.n_outputs = 1524;
    nc.fn = [](CircuitContext& c, const Wires& in) {
      Fq12 r = Fq12::from_wires(slice(in, 0, 3048));
      Fq12 b = Fq12::from_wires(slice(in, 3048, 6096));
      Fq12 t = fq12::square_montgomery(c, r);
      Fq12 u{{fq6::add(c, t.c[0], b.c[0]), fq6::sub(c, t.c[1], b.c[1])}};
      Fq12 v = fq12::mul_montgomery(c, u, b);
      Fq12 w = fq12::mul_montgomery(c, v, t);
      return w.c[0].to_wires();
    };
    // warm-ups for the plan builder's tests: the square and the fully live multiplication (the half-dead one is left to the driver)
    nc.warmups.push_back({3048, [](CircuitContext& c, const Wires& in) { return fq12::square_montgomery(c, Fq12::from_wires(in)).to_wires(); }});
    nc.warmups.push_back({6096, [](CircuitContext& c, const Wires& in) {
      return fq12::mul_montgomery(c, Fq12::from_wires(slice(in, 0, 3048)), Fq12::from_wires(slice(in, 3048, 6096))).to_wires();
    }});
  } else if (name == "gate") {
    if (!has_param || param > 10) gsv_panic("gate:T needs T in 0..10");
    GateType t = static_cast<GateType>(param);
    nc.n_inputs = 2; nc.n_outputs = 1;
    if (t == GateType::Not) {
      // Gate::not is in place (a == b == c), gate.rs:139-147; used only by the reference's tests.
      nc.fn = [](CircuitContext& c, const Wires& in) { c.add_gate(Gate::not_(in[0])); return Wires{in[0]}; };
    } else {
      nc.fn = [t](CircuitContext& c, const Wires& in) {
        WireId o = c.issue_wire();
        c.add_gate(Gate::make(t, in[0], in[1], o));
        return Wires{o};
      };
    }
  } else if (name == "random_circuit") {
    // random_circuit:SEED — 24 inputs, ~3000 gates, 16 outputs
    const uint64_t seed = param;
    nc.n_inputs = 24; nc.n_outputs = 16;
    nc.fn = [seed](CircuitContext& c, const Wires& in) {
      Wires pool = detail::random_block(c, in, seed * 7919 + 1, 3000, 2);
      detail::Mix64 r{seed ^ 0x5151ull};
      Wires out;
      for (int i = 0; i < 16; ++i) out.push_back(i == 0 ? in[3] : pool[pool.size() - 1 - r.below(200)]);
      return out;
    };
  } else if (name == "driver_mix") {
    nc.n_inputs = 6; nc.n_outputs = 4;
    nc.fn = detail::driver_mix;
  } else {
    gsv_panic("unknown circuit: " + spec);
  }
  return nc;
}

}  // namespace gsv
