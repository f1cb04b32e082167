// Gate-stream producers of the rest of groth16_verify: G1 addition, the constant-table multiplexers and
// the window-10 MSM with constant bases, projective -> affine, Fq12::equal_constant and the verifier's composition.
// Mirrors src/gadgets/basic.rs:73-105, src/gadgets/bigint/cmp.rs:171-194, src/gadgets/bn254/g1.rs:159-400,
// src/gadgets/bn254/{fq2,fq6,fq12}.rs equal_constant and src/gadgets/groth16.rs:26-110 call-for-call.
//
// What cannot be pinned without arkworks: the MSM's constant tables.  The reference fills them with ark_bn254::G1Projective
// values produced by ark-ec's own Jacobian formulas (`p += base`, `b + b`, g1.rs:319-352), so the (x, y, z) REPRESENTATIVE of
// every table entry is whatever that arithmetic leaves.  The host arithmetic below follows the formulas ark-ec 0.5 documents
// for short Weierstrass curves with a = 0 (add-2007-bl, dbl-2009-l, identity = (1, 1, 0)); the table entries are the same
// group elements and the circuit has the same gates in the same order, but which of the two constant wires an individual
// multiplexer gate reads is only as right as that restatement — flagged in DESIGN.md with the other unpinned constants.
#pragma once
#include "bn254_pairing.hpp"

namespace gsv {
namespace gadgets {
using fq::Fq;

// ------------------------------------------------------------------ basic.rs / bigint/cmp.rs multiplexers
inline WireId multiplexer_bit(CircuitContext& c, const Wires& a, const Wires& s, size_t w) {  // basic.rs:73-105 (#[component])
  if (a.size() != (size_t(1) << w) || s.size() != w) gsv_panic("multiplexer: wrong operand count");
  return component(c, KeyBuilder("basic::multiplexer").param_usize("w", w), concat(a, s), 1, [w](CircuitContext& cc, const Wires& in) -> Wires {
    const size_t n = size_t(1) << w;
    Wires cur(in.begin(), in.begin() + n);
    for (size_t k = 0; k < w; ++k) {  // pairs reduced under selector bits from LSB to MSB
      const WireId sel = in[n + k];
      size_t j = 0;
      for (size_t i = 0; i < cur.size(); i += 2) cur[j++] = selector(cc, cur[i + 1], cur[i], sel);
      cur.resize(cur.size() / 2);
    }
    return {cur[0]};
  })[0];
}
inline BigIntWires multiplexer(CircuitContext& c, const std::vector<BigIntWires>& a, const Wires& s, size_t w) {  // cmp.rs:171-194 (#[bn_component])
  const size_t n = size_t(1) << w, n_bits = a.at(0).size();
  if (a.size() != n || s.size() != w) gsv_panic("bigint::multiplexer: wrong operand count");
  Wires in;
  for (const auto& x : a) { if (x.size() != n_bits) gsv_panic("bigint::multiplexer: inconsistent widths"); in.insert(in.end(), x.begin(), x.end()); }
  in.insert(in.end(), s.begin(), s.end());
  return component(c, KeyBuilder("bigint::multiplexer").param_usize("w", w), in, n_bits, [n, n_bits, w](CircuitContext& cc, const Wires& x) {
    const Wires sel(x.begin() + n * n_bits, x.end());
    Wires out;
    for (size_t i = 0; i < n_bits; ++i) {
      Wires ith(n);
      for (size_t k = 0; k < n; ++k) ith[k] = x[k * n_bits + i];
      out.push_back(multiplexer_bit(cc, ith, sel, w));
    }
    return out;
  });
}

// ------------------------------------------------------------------ host-side G1 (Jacobian, standard form) for the constant tables
struct HG1 {
  HFq x, y, z;
  static HG1 identity() { return {HFq::from_u64(1), HFq::from_u64(1), HFq()}; }
  static HG1 from_affine(const HFq& ax, const HFq& ay) { return {ax, ay, HFq::from_u64(1)}; }
  bool is_zero() const { return z.is_zero(); }
  static HG1 dbl(const HG1& p) {  // dbl-2009-l
    if (p.is_zero()) return p;
    HFq a = HFq::mul(p.x, p.x), b = HFq::mul(p.y, p.y), c = HFq::mul(b, b);
    HFq xb = HFq::add(p.x, b);
    HFq d = HFq::sub(HFq::sub(HFq::mul(xb, xb), a), c); d = HFq::add(d, d);
    HFq e = HFq::add(HFq::add(a, a), a), f = HFq::mul(e, e);
    HFq z3 = HFq::mul(p.z, p.y); z3 = HFq::add(z3, z3);
    HFq x3 = HFq::sub(f, HFq::add(d, d));
    HFq c8 = HFq::add(c, c); c8 = HFq::add(c8, c8); c8 = HFq::add(c8, c8);
    HFq y3 = HFq::sub(HFq::mul(e, HFq::sub(d, x3)), c8);
    return {x3, y3, z3};
  }
  static HG1 add(const HG1& p, const HG1& q) {  // add-2007-bl
    if (p.is_zero()) return q;
    if (q.is_zero()) return p;
    HFq z1z1 = HFq::mul(p.z, p.z), z2z2 = HFq::mul(q.z, q.z);
    HFq u1 = HFq::mul(p.x, z2z2), u2 = HFq::mul(q.x, z1z1);
    HFq s1 = HFq::mul(HFq::mul(p.y, q.z), z2z2), s2 = HFq::mul(HFq::mul(q.y, p.z), z1z1);
    if (HFq::cmp(u1, u2) == 0 && HFq::cmp(s1, s2) == 0) return dbl(p);
    HFq h = HFq::sub(u2, u1);
    HFq i = HFq::add(h, h); i = HFq::mul(i, i);
    HFq j = HFq::mul(h, i);
    HFq r = HFq::sub(s2, s1); r = HFq::add(r, r);
    HFq v = HFq::mul(u1, i);
    HFq x3 = HFq::sub(HFq::sub(HFq::mul(r, r), j), HFq::add(v, v));
    HFq s1j = HFq::mul(s1, j);
    HFq y3 = HFq::sub(HFq::mul(r, HFq::sub(v, x3)), HFq::add(s1j, s1j));
    HFq zz = HFq::add(p.z, q.z);
    HFq z3 = HFq::mul(HFq::sub(HFq::sub(HFq::mul(zz, zz), z1z1), z2z2), h);
    return {x3, y3, z3};
  }
};

// ------------------------------------------------------------------ g1.rs
inline Wires g1_to_wires(const G1Wires& p) { return concat(concat(p.x, p.y), p.z); }
inline G1Wires g1_from_wires(const Wires& w) { return G1Wires{slice(w, 0, 254), slice(w, 254, 508), slice(w, 508, 762)}; }
inline Wires fq_constant_wires_montgomery(const HFq& v) {  // Fq::new_constant(&Fq::as_montgomery(v)): bits of v * 2^254 mod p
  static const HFq r254 = [] { HFq t = HFq::from_u64(1); for (int i = 0; i < 254; ++i) t = HFq::add(t, t); return t; }();
  const HFq m = HFq::mul(v, r254);
  Wires w(254);
  for (size_t i = 0; i < 254; ++i) w[i] = ((m.l[i / 64] >> (i % 64)) & 1) ? TRUE_WIRE : FALSE_WIRE;
  return w;
}
inline G1Wires g1_new_constant_montgomery(const HG1& p) {  // G1Projective::new_constant(&as_montgomery(p)), g1.rs:69-75,115-121
  return G1Wires{fq_constant_wires_montgomery(p.x), fq_constant_wires_montgomery(p.y), fq_constant_wires_montgomery(p.z)};
}

namespace g1 {
// g1.rs:159-235 (#[component]); no doubling case: equal operands give the all-zero point, as in the reference
inline G1Wires add_montgomery(CircuitContext& c, const G1Wires& p, const G1Wires& q) {
  fq::check_len(p.x); fq::check_len(p.y); fq::check_len(p.z); fq::check_len(q.x); fq::check_len(q.y); fq::check_len(q.z);
  Wires out = component(c, KeyBuilder("g1::add_montgomery"), concat(g1_to_wires(p), g1_to_wires(q)), 762, [](CircuitContext& cc, const Wires& in) {
    const Fq x1 = slice(in, 0, 254), y1 = slice(in, 254, 508), z1 = slice(in, 508, 762);
    const Fq x2 = slice(in, 762, 1016), y2 = slice(in, 1016, 1270), z2 = slice(in, 1270, 1524);
    Fq z1s = fq::square_montgomery(cc, z1);
    Fq z2s = fq::square_montgomery(cc, z2);
    Fq z1c = fq::mul_montgomery(cc, z1s, z1);
    Fq z2c = fq::mul_montgomery(cc, z2s, z2);
    Fq u1 = fq::mul_montgomery(cc, x1, z2s);
    Fq u2 = fq::mul_montgomery(cc, x2, z1s);
    Fq s1 = fq::mul_montgomery(cc, y1, z2c);
    Fq s2 = fq::mul_montgomery(cc, y2, z1c);
    Fq r = fq::sub(cc, s1, s2);
    Fq h = fq::sub(cc, u1, u2);
    Fq h2 = fq::square_montgomery(cc, h);
    Fq g = fq::mul_montgomery(cc, h, h2);
    Fq v = fq::mul_montgomery(cc, u1, h2);
    Fq r2 = fq::square_montgomery(cc, r);
    Fq r2g = fq::add(cc, r2, g);
    Fq vd = fq::double_(cc, v);
    Fq x3 = fq::sub(cc, r2g, vd);
    Fq vx3 = fq::sub(cc, v, x3);
    Fq w = fq::mul_montgomery(cc, r, vx3);
    Fq s1g = fq::mul_montgomery(cc, s1, g);
    Fq y3 = fq::sub(cc, w, s1g);
    Fq z1z2 = fq::mul_montgomery(cc, z1, z2);
    Fq z3 = fq::mul_montgomery(cc, z1z2, h);
    WireId z1_0 = fq::equal_constant(cc, z1, BigU());
    WireId z2_0 = fq::equal_constant(cc, z2, BigU());
    const Fq zero = constant_wires(BigU(), 254);
    const Wires s = {z1_0, z2_0};
    Fq x = multiplexer(cc, {x3, x2, x1, zero}, s, 2);
    Fq y = multiplexer(cc, {y3, y2, y1, zero}, s, 2);
    Fq z = multiplexer(cc, {z3, z2, z1, zero}, s, 2);
    return concat(concat(x, y), z);
  });
  return g1_from_wires(out);
}

// g1.rs:275-307 (#[component(offcircuit_args = "w")])
inline G1Wires multiplexer(CircuitContext& c, const std::vector<G1Wires>& a, const Wires& s, size_t w) {
  const size_t n = size_t(1) << w;
  if (a.size() != n || s.size() != w) gsv_panic("g1::multiplexer: wrong operand count");
  Wires in;
  in.reserve(n * 762 + w);
  for (const auto& p : a) { in.insert(in.end(), p.x.begin(), p.x.end()); in.insert(in.end(), p.y.begin(), p.y.end()); in.insert(in.end(), p.z.begin(), p.z.end()); }
  in.insert(in.end(), s.begin(), s.end());
  Wires out = component(c, KeyBuilder("g1::multiplexer").param_usize("w", w), in, 762, [n, w](CircuitContext& cc, const Wires& x) {
    const Wires sel(x.begin() + n * 762, x.end());
    Wires out;
    for (size_t coord = 0; coord < 3; ++coord) {
      std::vector<BigIntWires> col(n);
      for (size_t k = 0; k < n; ++k) col[k] = slice(x, k * 762 + coord * 254, k * 762 + coord * 254 + 254);
      Wires o = gadgets::multiplexer(cc, col, sel, w);
      out.insert(out.end(), o.begin(), o.end());
    }
    return out;
  });
  return g1_from_wires(out);
}

inline std::string g1_key_bytes(const HG1& p) { return p.x.to_bigu().key_bytes() + "," + p.y.to_bigu().key_bytes() + "," + p.z.to_bigu().key_bytes(); }

// g1.rs:309-368 (#[component(offcircuit_args = "base")]): windows of W scalar bits select from constant tables, then one chain of additions
inline G1Wires scalar_mul_by_constant_base_montgomery(CircuitContext& c, const Wires& s, const HG1& base, size_t W) {
  if (s.size() != 254) gsv_panic("Fr operand must have 254 wires");
  const std::string kb = g1_key_bytes(base);
  // the constant tables (host work, identical in the metadata and the execution pass of the component): built once
  const size_t n = size_t(1) << W;
  std::vector<std::vector<G1Wires>> tables;
  {
    std::vector<HG1> bases;
    HG1 p = HG1::identity();
    for (size_t i = 0; i < n; ++i) { bases.push_back(p); p = HG1::add(p, base); }
    for (size_t index = 0; index < 254; index += W) {
      const size_t w = std::min(W, 254 - index), m = size_t(1) << w;
      std::vector<G1Wires> table;
      for (size_t k = 0; k < m; ++k) table.push_back(g1_new_constant_montgomery(bases[k]));
      tables.push_back(std::move(table));
      for (auto& b : bases) for (size_t k = 0; k < w; ++k) b = HG1::add(b, b);
    }
  }
  Wires out = component(c, KeyBuilder("g1::scalar_mul_by_constant_base_montgomery").param_usize("W", W).param("base", kb.data(), kb.size()), s, 762,
                        [&tables, W](CircuitContext& cc, const Wires& sc) {
    std::vector<G1Wires> to_be_added;
    for (size_t index = 0, t = 0; index < 254; index += W, ++t) {
      const size_t w = std::min(W, 254 - index);
      to_be_added.push_back(multiplexer(cc, tables[t], slice(sc, index, index + w), w));
    }
    G1Wires acc = to_be_added[0];
    for (size_t i = 1; i < to_be_added.size(); ++i) acc = add_montgomery(cc, acc, to_be_added[i]);
    return g1_to_wires(acc);
  });
  return g1_from_wires(out);
}

// g1.rs:370-400 (#[component(offcircuit_args = "bases")])
inline G1Wires msm_with_constant_bases_montgomery(CircuitContext& c, const std::vector<Wires>& scalars, const std::vector<HG1>& bases, size_t W) {
  if (scalars.empty()) return g1_new_constant_montgomery(HG1::identity());
  if (scalars.size() != bases.size()) gsv_panic("msm: scalars and bases differ in length");
  Wires in;
  std::string kb;
  for (const auto& s : scalars) in.insert(in.end(), s.begin(), s.end());
  for (const auto& b : bases) { kb += g1_key_bytes(b); kb.push_back(';'); }
  const size_t n = scalars.size();
  Wires out = component(c, KeyBuilder("g1::msm_with_constant_bases_montgomery").param_usize("W", W).param("bases", kb.data(), kb.size()), in, 762,
                        [&bases, n, W](CircuitContext& cc, const Wires& x) {
    std::vector<G1Wires> to_be_added;
    for (size_t i = 0; i < n; ++i) to_be_added.push_back(scalar_mul_by_constant_base_montgomery(cc, slice(x, i * 254, i * 254 + 254), bases[i], W));
    G1Wires acc = to_be_added[0];
    for (size_t i = 1; i < n; ++i) acc = add_montgomery(cc, acc, to_be_added[i]);
    return g1_to_wires(acc);
  });
  return g1_from_wires(out);
}
}  // namespace g1

// ------------------------------------------------------------------ equal_constant over the tower (fq2.rs:148-158, fq6.rs:139-152, fq12.rs:158-168)
namespace fq2 {
inline WireId equal_constant(CircuitContext& c, const Fq2& a, const Fq2Const& b) {
  WireId u = fq::equal_constant(c, a.c[0], b.c0), v = fq::equal_constant(c, a.c[1], b.c1), w = c.issue_wire();
  c.add_gate(Gate::and_(u, v, w));
  return w;
}
}  // namespace fq2
struct Fq12Const { Fq2Const c[6]; };  // wire order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
namespace fq6 {
inline WireId equal_constant(CircuitContext& c, const Fq6& a, const Fq2Const* b) {
  WireId u = fq2::equal_constant(c, a.c[0], b[0]), v = fq2::equal_constant(c, a.c[1], b[1]), w = fq2::equal_constant(c, a.c[2], b[2]);
  WireId x = c.issue_wire(), y = c.issue_wire();
  c.add_gate(Gate::and_(u, v, x));
  c.add_gate(Gate::and_(x, w, y));
  return y;
}
}  // namespace fq6
namespace fq12 {
inline WireId equal_constant(CircuitContext& c, const Fq12& a, const Fq12Const& b) {
  WireId u = fq6::equal_constant(c, a.c[0], &b.c[0]), v = fq6::equal_constant(c, a.c[1], &b.c[3]), w = c.issue_wire();
  c.add_gate(Gate::and_(u, v, w));
  return w;
}
}  // namespace fq12

// ------------------------------------------------------------------ square roots (fp254impl.rs:691-725, fq.rs:175-192,290-299, fq2.rs:396-446)
namespace fq {
// fp254impl.rs:691-725 (#[bn_component(offcircuit_args = "exp")]): left-to-right square-and-multiply over the exponent's bits
inline size_t exp_chunk_bits() { static const size_t v = [] { const char* e = getenv("GSV_EXP_CHUNK"); const int n = e ? atoi(e) : 4; return size_t(n >= 1 && n <= 64 ? n : 4); }(); return v; }  // ladder steps per fp254::exp_chunk wrapper (experiments: GSV_EXP_CHUNK)
inline Fq exp_by_constant_montgomery(CircuitContext& c, const Fq& a, const BigU& exp) {
  check_len(a);
  const std::string kb = exp.key_bytes();
  return component(c, KeyBuilder("fp254::exp_by_constant_montgomery").param("exp", kb.data(), kb.size()), a, N, [&exp](CircuitContext& cc, const Wires& x) -> Wires {
    if (exp.is_zero()) return constant_wires(BigU(1), x.size());  // (the plain constant 1, as in the reference)
    if (exp == BigU(1)) return x;
    // The square-and-multiply ladder, exp_chunk_bits() (= 4) exponent bits at a time.  Each chunk is wrapped in a component of this
    // restatement's own ("fp254::exp_chunk", keyed by its bit pattern): a component boundary never changes the gate stream (a
    // wire is dead iff nothing reads it — the committed fixtures were produced before the wrapping existed), and it gives the
    // plan builder a unit of a few multiplications (<= 16 distinct programs per exponent) between the 380-multiplication ladder
    // and its single multiplications.
    Fq result = x;
    for (size_t hi = exp.bits() - 1; hi > 0;) {
      const size_t n = std::min(exp_chunk_bits(), hi);
      std::string pat;
      for (size_t k = 0; k < n; ++k) pat.push_back(exp.bit(hi - 1 - k) ? '1' : '0');
      result = component(cc, KeyBuilder("fp254::exp_chunk").param("bits", pat.data(), pat.size()), concat(x, result), N, [&pat](CircuitContext& c3, const Wires& in) -> Wires {
        const Fq base = slice(in, 0, N);
        Fq r = slice(in, N, 2 * N);
        for (char b : pat) {
          Fq sq = square_montgomery(c3, r);
          r = b == '1' ? mul_montgomery(c3, base, sq) : sq;
        }
        return r;
      });
      hi -= n;
    }
    return result;
  });
}
inline const BigU& modulus_add_1_div_4() { static BigU v = BigU::from_hex("0c19139cb84c680a6e14116da060561765e05aa45a1c72a34f082305b61f3f52"); return v; }  // fp254impl.rs:36-37
inline const BigU& modulus_minus_one_div_two() { static BigU v = BigU::from_hex("183227397098d014dc2822db40c0ac2ecbc0b548b438e5469e10460b6c3e7ea3"); return v; }
inline Fq sqrt_montgomery(CircuitContext& c, const Fq& a) { return exp_by_constant_montgomery(c, a, modulus_add_1_div_4()); }  // fq.rs:290-299
inline WireId is_qnr_montgomery(CircuitContext& c, const Fq& x) {  // fq.rs:177-192: x^((p-1)/2) == -1
  Fq y = exp_by_constant_montgomery(c, x, modulus_minus_one_div_two());
  const Fq neg_one = constant_wires(fq_as_montgomery_const(bigu_sub(FqConst::modulus(), BigU(1))), N);
  return gadgets::equal(c, y, neg_one);
}
}  // namespace fq
namespace fq2 {
// fq2.rs:425-446 (#[component]): the complex method (eprint 2012/685 algorithm 8), general case c1 != 0, the root is assumed to exist
inline Fq2 sqrt_general_montgomery(CircuitContext& c, const Fq2& a) {
  Wires out = component(c, KeyBuilder("fq2::sqrt_general_montgomery"), a.to_wires(), 508, [](CircuitContext& cc, const Wires& in) {
    const Fq a0 = slice(in, 0, 254), a1 = slice(in, 254, 508);
    Fq c0s = fq::square_montgomery(cc, a0);
    Fq c1s = fq::square_montgomery(cc, a1);
    Fq alpha = fq::add(cc, c0s, c1s);  // norm_montgomery, fq2.rs:397-402
    Fq alpha_sqrt = fq::sqrt_montgomery(cc, alpha);
    Fq delta_plus = fq::add(cc, alpha_sqrt, a0);
    Fq delta = fq::half(cc, delta_plus);
    WireId is_qnr = fq::is_qnr_montgomery(cc, delta);
    Fq delta_alt = fq::sub(cc, delta, alpha_sqrt);
    Fq delta_final = select(cc, delta_alt, delta, is_qnr);
    Fq c0_final = fq::sqrt_montgomery(cc, delta_final);
    Fq c0_inv = fq::inverse_montgomery(cc, c0_final);
    Fq c1_half = fq::half(cc, a1);
    Fq c1_final = fq::mul_montgomery(cc, c0_inv, c1_half);
    return concat(c0_final, c1_final);
  });
  return Fq2::from_wires(out);
}
}  // namespace fq2

// ------------------------------------------------------------------ groth16.rs
namespace groth16 {
// groth16.rs:26-48 (#[component])
inline G1Wires projective_to_affine_montgomery(CircuitContext& c, const G1Wires& p) {
  Wires out = component(c, KeyBuilder("groth16::projective_to_affine_montgomery"), g1_to_wires(p), 762, [](CircuitContext& cc, const Wires& in) {
    const Fq x = slice(in, 0, 254), y = slice(in, 254, 508), z = slice(in, 508, 762);
    Fq z_inverse = fq::inverse_montgomery(cc, z);
    Fq z_inverse_square = fq::square_montgomery(cc, z_inverse);
    Fq z_inverse_cube = fq::mul_montgomery(cc, z_inverse, z_inverse_square);
    Fq new_x = fq::mul_montgomery(cc, x, z_inverse_square);
    Fq new_y = fq::mul_montgomery(cc, y, z_inverse_cube);
    return concat(concat(new_x, new_y), constant_wires(fq_as_montgomery_const(BigU(1)), 254));
  });
  return g1_from_wires(out);
}

// The constant part of a Groth16 verifying key as the circuit needs it (standard form, affine points).  `alpha_beta` is the
// value the reference computes on the host with arkworks (groth16.rs:98-105): final_exponentiation(miller_loop(alpha_g1,
// -beta_g2)) inverted; the host side hands it over precomputed.
struct VerifyingKey {
  std::vector<std::pair<HFq, HFq>> gamma_abc_g1;  // [0] = constant term, then one base per public input
  HFq2 gamma_x, gamma_y, delta_x, delta_y;
  HFq alpha_beta[12];
};

// groth16.rs:58-110 (not a component): inputs are the public scalars (plain bits), A, B (G2), C in Montgomery form
inline WireId verify(CircuitContext& c, const std::vector<Wires>& pub, const G1Wires& a, const G2Wires& b, const G1Wires& cpt, const VerifyingKey& vk) {
  if (vk.gamma_abc_g1.size() < pub.size() + 1) gsv_panic("groth16: verifying key has too few gamma_abc_g1 entries");
  std::vector<HG1> bases;
  for (size_t i = 0; i < pub.size(); ++i) bases.push_back(HG1::from_affine(vk.gamma_abc_g1[i + 1].first, vk.gamma_abc_g1[i + 1].second));
  G1Wires msm_temp = g1::msm_with_constant_bases_montgomery(c, pub, bases, 10);
  G1Wires gamma0 = g1_new_constant_montgomery(HG1::from_affine(vk.gamma_abc_g1[0].first, vk.gamma_abc_g1[0].second));
  G1Wires msm = g1::add_montgomery(c, msm_temp, gamma0);
  G1Wires msm_affine = projective_to_affine_montgomery(c, msm);
  Fq12 f = pairing::multi_miller_loop_groth16_evaluate_montgomery_fast(c, msm_affine, cpt, a, vk.gamma_x, HFq2::neg(vk.gamma_y), vk.delta_x, HFq2::neg(vk.delta_y), b);
  f = fq12::final_exponentiation_montgomery(c, f);
  Fq12Const ab;
  for (int i = 0; i < 6; ++i) ab.c[i] = Fq2Const{fq_as_montgomery_const(vk.alpha_beta[2 * i].to_bigu()), fq_as_montgomery_const(vk.alpha_beta[2 * i + 1].to_bigu())};
  return fq12::equal_constant(c, f, ab);
}

// groth16.rs:116-143 (#[component]): y = +-sqrt(x^3 + b), the flag picks the circuit's own root or its negative; z = 1
inline G1Wires decompress_g1_from_compressed(CircuitContext& c, const Fq& x_m, WireId y_flag) {
  Wires in = x_m; in.push_back(y_flag);
  Wires out = component(c, KeyBuilder("groth16::decompress_g1_from_compressed"), in, 762, [](CircuitContext& cc, const Wires& w) {
    const Fq x = slice(w, 0, 254);
    const WireId flag = w[254];
    Fq x2 = fq::square_montgomery(cc, x);
    Fq x3 = fq::mul_montgomery(cc, x2, x);
    Fq rhs = fq::add_constant(cc, x3, fq_as_montgomery_const(BigU(3)));  // g1::Config::COEFF_B = 3
    Fq sy = fq::sqrt_montgomery(cc, rhs);
    Fq sy_neg = fq::neg(cc, sy);
    Fq y = select(cc, sy, sy_neg, flag);
    return concat(concat(x, y), constant_wires(fq_as_montgomery_const(BigU(1)), 254));
  });
  return g1_from_wires(out);
}
// groth16.rs:145-182 (#[component])
inline G2Wires decompress_g2_from_compressed(CircuitContext& c, const Fq2& x, WireId y_flag) {
  Wires in = x.to_wires(); in.push_back(y_flag);
  Wires out = component(c, KeyBuilder("groth16::decompress_g2_from_compressed"), in, 1524, [](CircuitContext& cc, const Wires& w) {
    const Fq2 x = Fq2::from_wires(slice(w, 0, 508));
    const WireId flag = w[508];
    Fq2 x2 = fq2::square_montgomery(cc, x);
    Fq2 x3 = fq2::mul_montgomery(cc, x2, x);
    Fq2 y2 = fq2::add_constant(cc, x3, PairingConst::coeff_b().as_montgomery_const());
    Fq2 y = fq2::sqrt_general_montgomery(cc, y2);
    Fq2 neg_y = fq2::neg(cc, y);
    Fq y0 = select(cc, y.c[0], neg_y.c[0], flag);
    Fq y1 = select(cc, y.c[1], neg_y.c[1], flag);
    Wires out = x.to_wires();
    out.insert(out.end(), y0.begin(), y0.end());
    out.insert(out.end(), y1.begin(), y1.end());
    Wires one = constant_wires(fq_as_montgomery_const(BigU(1)), 254), zero = constant_wires(BigU(), 254);
    out.insert(out.end(), one.begin(), one.end());
    out.insert(out.end(), zero.begin(), zero.end());
    return out;
  });
  return pairing::g2_from_wires(out);
}
// groth16.rs:250-268: decompress A, B, C, then the verifier
inline WireId verify_compressed(CircuitContext& c, const std::vector<Wires>& pub, const Fq& ax, WireId a_flag, const Fq2& bx, WireId b_flag, const Fq& cx, WireId c_flag,
                                const VerifyingKey& vk) {
  G1Wires a = decompress_g1_from_compressed(c, ax, a_flag);
  G2Wires b = decompress_g2_from_compressed(c, bx, b_flag);
  G1Wires cp = decompress_g1_from_compressed(c, cx, c_flag);
  return verify(c, pub, a, b, cp, vk);
}

// Binary form of a verifying key inside a circuit name ("groth16_verify:<hex>"): n_pub (1 byte), then 32-byte big-endian
// field elements: (n_pub + 1) x (x, y) of gamma_abc_g1; gamma_g2 (x.c0, x.c1, y.c0, y.c1); delta_g2 likewise; alpha_beta (12).
inline VerifyingKey vk_from_hex(const std::string& hex, size_t* n_pub_out) {
  if (hex.size() < 2 || hex.size() % 2) gsv_panic("groth16 vk: bad hex length");
  auto byte = [&](size_t i) { return uint8_t(std::stoul(hex.substr(2 * i, 2), nullptr, 16)); };
  const size_t n_pub = byte(0), n_fe = 2 * (n_pub + 1) + 8 + 12;
  if (hex.size() != 2 * (1 + 32 * n_fe)) gsv_panic("groth16 vk: wrong length for its public-input count");
  size_t pos = 1;
  auto fe = [&]() { HFq v = HFq::from_bigu(BigU::from_hex(hex.substr(2 * pos, 64))); pos += 32; if (HFq::cmp(v, HFq::p()) >= 0) gsv_panic("groth16 vk: field element not reduced"); return v; };
  VerifyingKey vk;
  for (size_t i = 0; i <= n_pub; ++i) { HFq x = fe(), y = fe(); vk.gamma_abc_g1.push_back({x, y}); }
  vk.gamma_x.c0 = fe(); vk.gamma_x.c1 = fe(); vk.gamma_y.c0 = fe(); vk.gamma_y.c1 = fe();
  vk.delta_x.c0 = fe(); vk.delta_x.c1 = fe(); vk.delta_y.c0 = fe(); vk.delta_y.c1 = fe();
  for (int i = 0; i < 12; ++i) vk.alpha_beta[i] = fe();
  *n_pub_out = n_pub;
  return vk;
}
}  // namespace groth16

}  // namespace gadgets
}  // namespace gsv
