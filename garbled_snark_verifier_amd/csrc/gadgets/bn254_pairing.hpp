// Gate-stream producers of the Groth16 Miller loop: sparse Fq12 multiplications, the G2 line
// functions with wire and with constant Q, and multi_miller_loop_groth16_evaluate_montgomery_fast.
// Mirrors src/gadgets/bn254/{fq2,fq6,fq12,pairing}.rs call-for-call.
//
// Off-circuit values the reference takes from ark_bn254 (the crate is an un-vendored Cargo dependency, ark-bn254 0.5):
//   g2::Config::COEFF_B = 3 / (9 + u), Config::TWIST_MUL_BY_Q_X = xi^((p-1)/3), TWIST_MUL_BY_Q_Y = xi^((p-1)/2),
//   Config::ATE_LOOP_COUNT = the signed digits of 6x + 2 below.  The first three are derived (tests check them against
//   Python arithmetic); the DIGIT SEQUENCE is restated from the published arkworks configuration and is only value-checked
//   (sum d_i 2^i == 6x + 2) — like every other parity statement of this repository it needs a first contact with cargo.
#pragma once
#include "bn254_ext.hpp"

namespace gsv {
namespace gadgets {

// ------------------------------------------------------------------ host-side Fq / Fq2 arithmetic for constants
struct HFq {
  uint64_t l[4] = {0, 0, 0, 0};
  static HFq from_bigu(const BigU& b) {
    HFq r;
    const auto& v = b.limbs();
    for (size_t i = 0; i < v.size() && i < 8; ++i) r.l[i / 2] |= uint64_t(v[i]) << (32 * (i % 2));
    return r;
  }
  BigU to_bigu() const {
    std::vector<uint32_t> v;
    for (int i = 0; i < 4; ++i) { v.push_back(uint32_t(l[i])); v.push_back(uint32_t(l[i] >> 32)); }
    return bigu_from_limbs(v);
  }
  static const HFq& p() { static HFq v = from_bigu(FqConst::modulus()); return v; }
  static int cmp(const HFq& a, const HFq& b) { for (int i = 3; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1; return 0; }
  bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
  static HFq add(const HFq& a, const HFq& b) {
    HFq r; unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = uint64_t(c); c >>= 64; }
    if (c || cmp(r, p()) >= 0) r = sub_raw(r, p());
    return r;
  }
  static HFq sub_raw(const HFq& a, const HFq& b) {
    HFq r; unsigned __int128 bw = 0;
    for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - bw; r.l[i] = uint64_t(d); bw = (d >> 64) & 1; }
    return r;
  }
  static HFq sub(const HFq& a, const HFq& b) { return cmp(a, b) >= 0 ? sub_raw(a, b) : sub_raw(add_raw(a, p()), b); }
  static HFq add_raw(const HFq& a, const HFq& b) {
    HFq r; unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = uint64_t(c); c >>= 64; }
    return r;  // callers only use it where the sum fits (a < p, b = p < 2^254)
  }
  static HFq neg(const HFq& a) { return a.is_zero() ? a : sub_raw(p(), a); }
  // a * b * 2^-256 mod p (word-serial Montgomery product; p < 2^254 leaves room for the carries)
  static HFq mont(const HFq& a, const HFq& b) {
    static const uint64_t n0 = [] { uint64_t p0 = p().l[0], x = 1; for (int i = 0; i < 6; ++i) x *= 2 - p0 * x; return ~x + 1; }();  // -p^-1 mod 2^64
    const HFq& m = p();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 acc = 0;
      for (int j = 0; j < 4; ++j) { acc += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = uint64_t(acc); acc >>= 64; }
      acc += t[4]; t[4] = uint64_t(acc); t[5] = uint64_t(acc >> 64);
      const uint64_t q = t[0] * n0;
      acc = (unsigned __int128)q * m.l[0] + t[0]; acc >>= 64;
      for (int j = 1; j < 4; ++j) { acc += (unsigned __int128)q * m.l[j] + t[j]; t[j - 1] = uint64_t(acc); acc >>= 64; }
      acc += t[4]; t[3] = uint64_t(acc); t[4] = t[5] + uint64_t(acc >> 64);
    }
    HFq r; for (int i = 0; i < 4; ++i) r.l[i] = t[i];
    if (t[4] || cmp(r, m) >= 0) r = sub_raw(r, m);
    return r;
  }
  static HFq mul(const HFq& a, const HFq& b) {  // standard-form product
    static const HFq r2 = [] { HFq v = from_u64(1); for (int i = 0; i < 512; ++i) v = add(v, v); return v; }();  // 2^512 mod p
    return mont(mont(a, b), r2);
  }
  static HFq from_u64(uint64_t v) { HFq r; r.l[0] = v; return r; }
};
struct HFq2 {
  HFq c0, c1;
  static HFq2 add(const HFq2& a, const HFq2& b) { return {HFq::add(a.c0, b.c0), HFq::add(a.c1, b.c1)}; }
  static HFq2 sub(const HFq2& a, const HFq2& b) { return {HFq::sub(a.c0, b.c0), HFq::sub(a.c1, b.c1)}; }
  static HFq2 neg(const HFq2& a) { return {HFq::neg(a.c0), HFq::neg(a.c1)}; }
  static HFq2 dbl(const HFq2& a) { return add(a, a); }
  static HFq2 mul(const HFq2& a, const HFq2& b) {
    HFq t0 = HFq::mul(a.c0, b.c0), t1 = HFq::mul(a.c1, b.c1);
    HFq s = HFq::mul(HFq::add(a.c0, a.c1), HFq::add(b.c0, b.c1));
    return {HFq::sub(t0, t1), HFq::sub(HFq::sub(s, t0), t1)};
  }
  static HFq2 sq(const HFq2& a) { return mul(a, a); }
  static HFq2 mul_fp(const HFq2& a, const HFq& k) { return {HFq::mul(a.c0, k), HFq::mul(a.c1, k)}; }
  static HFq2 conj(const HFq2& a) { return {a.c0, HFq::neg(a.c1)}; }  // frobenius_map(1)
  Fq2Const as_montgomery_const() const { return Fq2Const{fq_as_montgomery_const(c0.to_bigu()), fq_as_montgomery_const(c1.to_bigu())}; }
};
inline HFq2 hfq2_hex(const char* a, const char* b) { return {HFq::from_bigu(BigU::from_hex(a)), HFq::from_bigu(BigU::from_hex(b))}; }

struct PairingConst {
  // ark_bn254::g2::Config::COEFF_B = 3 / (9 + u)
  static const HFq2& coeff_b() {
    static HFq2 v = hfq2_hex("2b149d40ceb8aaae81be18991be06ac3b5b4c5e559dbefa33267e6dc24a138e5", "009713b03af0fed4cd2cafadeed8fdf4a74fa084e52d1852e4a2bd0685c315d2");
    return v;
  }
  // Config::TWIST_MUL_BY_Q_X = xi^((p-1)/3), TWIST_MUL_BY_Q_Y = xi^((p-1)/2)
  static const HFq2& twist_x() {
    static HFq2 v = hfq2_hex("2fb347984f7911f74c0bec3cf559b143b78cc310c2c3330c99e39557176f553d", "16c9e55061ebae204ba4cc8bd75a079432ae2a1d0b7c9dce1665d51c640fcba2");
    return v;
  }
  static const HFq2& twist_y() {
    static HFq2 v = hfq2_hex("063cf305489af5dcdc5ec698b6e2f9b9dbaae0eda9c95998dc54014671a0135a", "07c03cbcac41049a0704b5a7ec796f2b21807dc98fa25bd282d37f632623b0e3");
    return v;
  }
  static const HFq& half() { static HFq v = HFq::from_bigu(FqConst::half_modulus()); return v; }
};
// Config::ATE_LOOP_COUNT: signed digits of 6x + 2 = 29793968203157093288, least significant first (65 entries)
static const int8_t ATE_LOOP_COUNT[65] = {0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 1, 1, 1,
                                          0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, 1, 1};

// pairing.rs:30-133 on the host: line coefficients of a CONSTANT G2 point (affine, standard form)
struct HG2 { HFq2 x, y, z; };
struct HEllCoeff { HFq2 c0, c1, c2; };
inline HEllCoeff h_double_in_place(HG2& r) {
  HFq2 a = HFq2::mul_fp(HFq2::mul(r.x, r.y), PairingConst::half());
  HFq2 b = HFq2::sq(r.y), c = HFq2::sq(r.z);
  HFq2 e = HFq2::mul(PairingConst::coeff_b(), HFq2::add(HFq2::dbl(c), c));
  HFq2 f = HFq2::add(HFq2::dbl(e), e);
  HFq2 g = HFq2::mul_fp(HFq2::add(b, f), PairingConst::half());
  HFq2 h = HFq2::sub(HFq2::sq(HFq2::add(r.y, r.z)), HFq2::add(b, c));
  HFq2 i = HFq2::sub(e, b);
  HFq2 j = HFq2::sq(r.x);
  HFq2 e2 = HFq2::sq(e);
  HG2 n{HFq2::mul(a, HFq2::sub(b, f)), HFq2::sub(HFq2::sq(g), HFq2::add(HFq2::dbl(e2), e2)), HFq2::mul(b, h)};
  r = n;
  return {HFq2::neg(h), HFq2::add(HFq2::dbl(j), j), i};
}
inline HEllCoeff h_add_in_place(HG2& r, const HFq2& qx, const HFq2& qy) {
  HFq2 theta = HFq2::sub(r.y, HFq2::mul(qy, r.z));
  HFq2 lambda = HFq2::sub(r.x, HFq2::mul(qx, r.z));
  HFq2 c = HFq2::sq(theta), d = HFq2::sq(lambda);
  HFq2 e = HFq2::mul(lambda, d), f = HFq2::mul(r.z, c), g = HFq2::mul(r.x, d);
  HFq2 h = HFq2::sub(HFq2::add(e, f), HFq2::dbl(g));
  HFq2 j = HFq2::sub(HFq2::mul(theta, qx), HFq2::mul(lambda, qy));
  HG2 n{HFq2::mul(lambda, h), HFq2::sub(HFq2::mul(theta, HFq2::sub(g, h)), HFq2::mul(e, r.y)), HFq2::mul(r.z, e)};
  r = n;
  return {lambda, HFq2::neg(theta), j};
}
inline std::vector<HEllCoeff> h_ell_coeffs(const HFq2& qx, const HFq2& qy) {
  std::vector<HEllCoeff> out;
  HG2 r{qx, qy, HFq2{HFq::from_u64(1), HFq()}};
  const HFq2 nqy = HFq2::neg(qy);
  for (int k = 63; k >= 0; --k) {  // ATE_LOOP_COUNT.iter().rev().skip(1)
    out.push_back(h_double_in_place(r));
    if (ATE_LOOP_COUNT[k] == 1) out.push_back(h_add_in_place(r, qx, qy));
    else if (ATE_LOOP_COUNT[k] == -1) out.push_back(h_add_in_place(r, qx, nqy));
  }
  const HFq2 q1x = HFq2::mul(HFq2::conj(qx), PairingConst::twist_x()), q1y = HFq2::mul(HFq2::conj(qy), PairingConst::twist_y());
  const HFq2 q2x = HFq2::mul(HFq2::conj(q1x), PairingConst::twist_x()), q2y = HFq2::neg(HFq2::mul(HFq2::conj(q1y), PairingConst::twist_y()));
  out.push_back(h_add_in_place(r, q1x, q1y));
  out.push_back(h_add_in_place(r, q2x, q2y));
  return out;
}

// ------------------------------------------------------------------ fq2.rs / fq6.rs / fq12.rs additions
namespace fq2 {
inline Fq2 add_constant(CircuitContext& c, const Fq2& a, const Fq2Const& b) { return Fq2{{fq::add_constant(c, a.c[0], b.c0), fq::add_constant(c, a.c[1], b.c1)}}; }  // fq2.rs:170-177
inline Fq2 mul_by_fq_montgomery(CircuitContext& c, const Fq2& a, const Wires& b) {  // fq2.rs:282-291
  return Fq2{{fq::mul_montgomery(c, a.c[0], b), fq::mul_montgomery(c, a.c[1], b)}};
}
// fq2.rs:307-322 (#[component(offcircuit_args = "a")]): a is a STANDARD-form constant, b a wire
inline Fq2 mul_constant_by_fq_montgomery(CircuitContext& c, const HFq2& a, const Wires& b) {
  const BigU a0m = fq_as_montgomery_const(a.c0.to_bigu()), a1m = fq_as_montgomery_const(a.c1.to_bigu());
  std::string k0 = a0m.key_bytes(), k1 = a1m.key_bytes();
  Wires out = component(c, KeyBuilder("fq2::mul_constant_by_fq_montgomery").param("a0", k0.data(), k0.size()).param("a1", k1.data(), k1.size()), b, 508,
                        [&a0m, &a1m](CircuitContext& cc, const Wires& in) {
    Wires c0 = fq::mul_by_constant_montgomery(cc, in, a0m);
    Wires c1 = fq::mul_by_constant_montgomery(cc, in, a1m);
    return concat(c0, c1);
  });
  return Fq2::from_wires(out);
}
}  // namespace fq2

namespace fq6 {
// The three sparse Fq6 multiplications below are wrapped in components of their own like fq6::mul_montgomery (bn254.hpp): stream-
// neutral, and the units at which a line evaluation (mul_by_034: two mul_by_01 + one mul_by_fq2, mutually independent up to a few
// additions) has its width for a plan session.
inline Fq6 mul_by_fq2_montgomery(CircuitContext& c0_, const Fq6& a_, const Fq2& b_) {  // fq6.rs:326-332
  Wires out = component(c0_, KeyBuilder("fq6::mul_by_fq2_montgomery"), concat(a_.to_wires(), b_.to_wires()), 1524, [](CircuitContext& c, const Wires& in) {
    const Fq6 a = Fq6::from_wires(slice(in, 0, 1524));
    const Fq2 b = Fq2::from_wires(slice(in, 1524, 2032));
    return Fq6{{fq2::mul_montgomery(c, a.c[0], b), fq2::mul_montgomery(c, a.c[1], b), fq2::mul_montgomery(c, a.c[2], b)}}.to_wires();
  });
  return Fq6::from_wires(out);
}
inline Fq6 mul_by_01_montgomery(CircuitContext& c0_, const Fq6& a_, const Fq2& c0_in, const Fq2& c1_in) {  // fq6.rs:351-379
  Wires out = component(c0_, KeyBuilder("fq6::mul_by_01_montgomery"), concat(concat(a_.to_wires(), c0_in.to_wires()), c1_in.to_wires()), 1524, [](CircuitContext& c, const Wires& in) {
    const Fq6 a = Fq6::from_wires(slice(in, 0, 1524));
    const Fq2 c0 = Fq2::from_wires(slice(in, 1524, 2032)), c1 = Fq2::from_wires(slice(in, 2032, 2540));
    const Fq2 &a0 = a.c[0], &a1 = a.c[1], &a2 = a.c[2];
    Fq2 w1 = fq2::mul_montgomery(c, a0, c0);
    Fq2 w2 = fq2::mul_montgomery(c, a1, c1);
    Fq2 w3 = fq2::add(c, a1, a2);
    Fq2 w4 = fq2::mul_montgomery(c, w3, c1);
    Fq2 w5 = fq2::sub(c, w4, w2);
    Fq2 w6 = fq2::mul_by_nonresidue(c, w5);
    Fq2 w7 = fq2::add(c, w6, w1);
    Fq2 w8 = fq2::add(c, a0, a1);
    Fq2 w9 = fq2::add(c, c0, c1);
    Fq2 w10 = fq2::mul_montgomery(c, w8, w9);
    Fq2 w11 = fq2::sub(c, w10, w1);
    Fq2 w12 = fq2::sub(c, w11, w2);
    Fq2 w13 = fq2::add(c, a0, a2);
    Fq2 w14 = fq2::mul_montgomery(c, w13, c0);
    Fq2 w15 = fq2::sub(c, w14, w1);
    Fq2 w16 = fq2::add(c, w15, w2);
    return Fq6{{w7, w12, w16}}.to_wires();
  });
  return Fq6::from_wires(out);
}
// fq6.rs:381-410: c1 is a constant handed over in Montgomery form
inline Fq6 mul_by_01_constant1_montgomery(CircuitContext& c0_, const Fq6& a_, const Fq2& c0_in, const Fq2Const& c1) {
  const std::string k0 = c1.c0.key_bytes(), k1 = c1.c1.key_bytes();
  Wires out = component(c0_, KeyBuilder("fq6::mul_by_01_constant1_montgomery").param("c1_0", k0.data(), k0.size()).param("c1_1", k1.data(), k1.size()),
                        concat(a_.to_wires(), c0_in.to_wires()), 1524, [&c1](CircuitContext& c, const Wires& in) {
    const Fq6 a = Fq6::from_wires(slice(in, 0, 1524));
    const Fq2 c0 = Fq2::from_wires(slice(in, 1524, 2032));
    const Fq2 &a0 = a.c[0], &a1 = a.c[1], &a2 = a.c[2];
    Fq2 w1 = fq2::mul_montgomery(c, a0, c0);
    Fq2 w2 = fq2::mul_by_constant_montgomery(c, a1, c1);
    Fq2 w3 = fq2::add(c, a1, a2);
    Fq2 w4 = fq2::mul_by_constant_montgomery(c, w3, c1);
    Fq2 w5 = fq2::sub(c, w4, w2);
    Fq2 w6 = fq2::mul_by_nonresidue(c, w5);
    Fq2 w7 = fq2::add(c, w6, w1);
    Fq2 w8 = fq2::add(c, a0, a1);
    Fq2 w9 = fq2::add_constant(c, c0, c1);
    Fq2 w10 = fq2::mul_montgomery(c, w8, w9);
    Fq2 w11 = fq2::sub(c, w10, w1);
    Fq2 w12 = fq2::sub(c, w11, w2);
    Fq2 w13 = fq2::add(c, a0, a2);
    Fq2 w14 = fq2::mul_montgomery(c, w13, c0);
    Fq2 w15 = fq2::sub(c, w14, w1);
    Fq2 w16 = fq2::add(c, w15, w2);
    return Fq6{{w7, w12, w16}}.to_wires();
  });
  return Fq6::from_wires(out);
}
}  // namespace fq6

namespace fq12 {
inline Fq12 mul_by_034_montgomery(CircuitContext& c, const Fq12& a, const Fq2& c0, const Fq2& c3, const Fq2& c4) {  // fq12.rs:266-285 (#[component])
  Wires in = concat(concat(concat(a.to_wires(), c0.to_wires()), c3.to_wires()), c4.to_wires());
  Wires out = component(c, KeyBuilder("fq12::mul_by_034_montgomery"), in, N, [](CircuitContext& cc, const Wires& x) {
    Fq12 a = Fq12::from_wires(slice(x, 0, N));
    Fq2 c0 = Fq2::from_wires(slice(x, N, N + 508)), c3 = Fq2::from_wires(slice(x, N + 508, N + 1016)), c4 = Fq2::from_wires(slice(x, N + 1016, N + 1524));
    Fq6 w1 = fq6::mul_by_01_montgomery(cc, a.c[1], c3, c4);
    Fq6 w2 = fq6::mul_by_nonresidue(cc, w1);
    Fq6 w3 = fq6::mul_by_fq2_montgomery(cc, a.c[0], c0);
    Fq6 new_c0 = fq6::add(cc, w2, w3);
    Fq6 w4 = fq6::add(cc, a.c[0], a.c[1]);
    Fq2 w5 = fq2::add(cc, c3, c0);
    Fq6 w6 = fq6::mul_by_01_montgomery(cc, w4, w5, c4);
    Fq6 w7 = fq6::add(cc, w1, w3);
    Fq6 new_c1 = fq6::sub(cc, w6, w7);
    return Fq12{{new_c0, new_c1}}.to_wires();
  });
  return Fq12::from_wires(out);
}
// fq12.rs:287-310 (#[component(offcircuit_args = "c4")]): c4 is a constant in Montgomery form
inline Fq12 mul_by_034_constant4_montgomery(CircuitContext& c, const Fq12& a, const Fq2& c0, const Fq2& c3, const Fq2Const& c4) {
  Wires in = concat(concat(a.to_wires(), c0.to_wires()), c3.to_wires());
  std::string k0 = c4.c0.key_bytes(), k1 = c4.c1.key_bytes();
  Wires out = component(c, KeyBuilder("fq12::mul_by_034_constant4_montgomery").param("c4_0", k0.data(), k0.size()).param("c4_1", k1.data(), k1.size()), in, N,
                        [&c4](CircuitContext& cc, const Wires& x) {
    Fq12 a = Fq12::from_wires(slice(x, 0, N));
    Fq2 c0 = Fq2::from_wires(slice(x, N, N + 508)), c3 = Fq2::from_wires(slice(x, N + 508, N + 1016));
    Fq6 w1 = fq6::mul_by_01_constant1_montgomery(cc, a.c[1], c3, c4);
    Fq6 w2 = fq6::mul_by_nonresidue(cc, w1);
    Fq6 w3 = fq6::mul_by_fq2_montgomery(cc, a.c[0], c0);
    Fq6 new_c0 = fq6::add(cc, w2, w3);
    Fq6 w4 = fq6::add(cc, a.c[0], a.c[1]);
    Fq2 w5 = fq2::add(cc, c3, c0);
    Fq6 w6 = fq6::mul_by_01_constant1_montgomery(cc, w4, w5, c4);
    Fq6 w7 = fq6::add(cc, w1, w3);
    Fq6 new_c1 = fq6::sub(cc, w6, w7);
    return Fq12{{new_c0, new_c1}}.to_wires();
  });
  return Fq12::from_wires(out);
}
}  // namespace fq12

// ------------------------------------------------------------------ pairing.rs
struct G1Wires { Wires x, y, z; };                 // g1.rs:13-17 (affine where the Miller loop uses it: z = Montgomery ONE)
struct G2Wires { Fq2 x, y, z; };                   // g2.rs:16-20
struct G2Step { G2Wires r; Fq6 coeffs; };

namespace pairing {
inline G2Wires g2_affine_neg_evaluate(CircuitContext& c, const G2Wires& q) { return G2Wires{q.x, fq2::neg(c, q.y), q.z}; }  // pairing.rs:466-473

inline Wires g2_to_wires(const G2Wires& g) { return concat(concat(g.x.to_wires(), g.y.to_wires()), g.z.to_wires()); }
inline G2Wires g2_from_wires(const Wires& w) { return G2Wires{Fq2::from_wires(slice(w, 0, 508)), Fq2::from_wires(slice(w, 508, 1016)), Fq2::from_wires(slice(w, 1016, 1524))}; }

inline G2Step double_in_place_circuit_montgomery(CircuitContext& c, const G2Wires& r) {  // pairing.rs:359-407 (#[component])
  Wires out = component(c, KeyBuilder("pairing::double_in_place_circuit_montgomery"), g2_to_wires(r), 1524 + 1524, [](CircuitContext& cc, const Wires& in) {
    G2Wires r = g2_from_wires(in);
    const Fq2 &rx = r.x, &ry = r.y, &rz = r.z;
    Fq2 a = fq2::mul_montgomery(cc, rx, ry);
    a = fq2::half(cc, a);
    Fq2 b = fq2::square_montgomery(cc, ry);
    Fq2 cq = fq2::square_montgomery(cc, rz);
    Fq2 c_triple = fq2::triple(cc, cq);
    Fq2 e = fq2::mul_by_constant_montgomery(cc, c_triple, PairingConst::coeff_b().as_montgomery_const());
    Fq2 f = fq2::triple(cc, e);
    Fq2 g = fq2::add(cc, b, f);
    g = fq2::half(cc, g);
    Fq2 ryrz = fq2::add(cc, ry, rz);
    Fq2 ryrzs = fq2::square_montgomery(cc, ryrz);
    Fq2 bc = fq2::add(cc, b, cq);
    Fq2 h = fq2::sub(cc, ryrzs, bc);
    Fq2 i = fq2::sub(cc, e, b);
    Fq2 j = fq2::square_montgomery(cc, rx);
    Fq2 es = fq2::square_montgomery(cc, e);
    Fq2 j_triple = fq2::triple(cc, j);
    Fq2 bf = fq2::sub(cc, b, f);
    Fq2 new_x = fq2::mul_montgomery(cc, a, bf);
    Fq2 es_triple = fq2::triple(cc, es);
    Fq2 gs = fq2::square_montgomery(cc, g);
    Fq2 new_y = fq2::sub(cc, gs, es_triple);
    Fq2 new_z = fq2::mul_montgomery(cc, b, h);
    Fq2 hn = fq2::neg(cc, h);
    return concat(g2_to_wires(G2Wires{new_x, new_y, new_z}), Fq6{{hn, j_triple, i}}.to_wires());
  });
  return G2Step{g2_from_wires(slice(out, 0, 1524)), Fq6::from_wires(slice(out, 1524, 3048))};
}
inline G2Step add_in_place_montgomery(CircuitContext& c, const G2Wires& r, const G2Wires& q) {  // pairing.rs:409-464 (#[component])
  Wires out = component(c, KeyBuilder("pairing::add_in_place_montgomery"), concat(g2_to_wires(r), g2_to_wires(q)), 1524 + 1524, [](CircuitContext& cc, const Wires& in) {
    G2Wires r = g2_from_wires(slice(in, 0, 1524)), q = g2_from_wires(slice(in, 1524, 3048));
    const Fq2 &rx = r.x, &ry = r.y, &rz = r.z, &qx = q.x, &qy = q.y;
    Fq2 wires_1 = fq2::mul_montgomery(cc, qy, rz);
    Fq2 theta = fq2::sub(cc, ry, wires_1);
    Fq2 wires_2 = fq2::mul_montgomery(cc, qx, rz);
    Fq2 lambda = fq2::sub(cc, rx, wires_2);
    Fq2 cq = fq2::square_montgomery(cc, theta);
    Fq2 d = fq2::square_montgomery(cc, lambda);
    Fq2 e = fq2::mul_montgomery(cc, lambda, d);
    Fq2 f = fq2::mul_montgomery(cc, rz, cq);
    Fq2 g = fq2::mul_montgomery(cc, rx, d);
    Fq2 wires_3 = fq2::add(cc, e, f);
    Fq2 wires_4 = fq2::double_(cc, g);
    Fq2 h = fq2::sub(cc, wires_3, wires_4);
    Fq2 neg_theta = fq2::neg(cc, theta);
    Fq2 wires_5 = fq2::mul_montgomery(cc, theta, qx);
    Fq2 wires_6 = fq2::mul_montgomery(cc, lambda, qy);
    Fq2 j = fq2::sub(cc, wires_5, wires_6);
    Fq2 new_r_x = fq2::mul_montgomery(cc, lambda, h);
    Fq2 wires_7 = fq2::sub(cc, g, h);
    Fq2 wires_8 = fq2::mul_montgomery(cc, theta, wires_7);
    Fq2 wires_9 = fq2::mul_montgomery(cc, e, ry);
    Fq2 new_r_y = fq2::sub(cc, wires_8, wires_9);
    Fq2 new_r_z = fq2::mul_montgomery(cc, rz, e);
    return concat(g2_to_wires(G2Wires{new_r_x, new_r_y, new_r_z}), Fq6{{lambda, neg_theta, j}}.to_wires());
  });
  return G2Step{g2_from_wires(slice(out, 0, 1524)), Fq6::from_wires(slice(out, 1524, 3048))};
}
inline G2Wires mul_by_char_montgomery(CircuitContext& c, const G2Wires& r) {  // pairing.rs:475-501 (#[component])
  Wires out = component(c, KeyBuilder("pairing::mul_by_char_montgomery"), g2_to_wires(r), 1524, [](CircuitContext& cc, const Wires& in) {
    G2Wires r = g2_from_wires(in);
    Fq2 s_x = fq2::frobenius_montgomery(cc, r.x, 1);
    s_x = fq2::mul_by_constant_montgomery(cc, s_x, PairingConst::twist_x().as_montgomery_const());
    Fq2 s_y = fq2::frobenius_montgomery(cc, r.y, 1);
    s_y = fq2::mul_by_constant_montgomery(cc, s_y, PairingConst::twist_y().as_montgomery_const());
    return g2_to_wires(G2Wires{s_x, s_y, r.z});
  });
  return g2_from_wires(out);
}
inline std::vector<Fq6> ell_coeffs_montgomery(CircuitContext& c, const G2Wires& q) {  // pairing.rs:507-547
  G2Wires neg_q = g2_affine_neg_evaluate(c, q);
  std::vector<Fq6> ellc;
  G2Wires r = q;
  for (int k = 63; k >= 0; --k) {
    G2Step s = double_in_place_circuit_montgomery(c, r);
    ellc.push_back(s.coeffs);
    r = s.r;
    if (ATE_LOOP_COUNT[k] == 1) { G2Step t = add_in_place_montgomery(c, r, q); ellc.push_back(t.coeffs); r = t.r; }
    else if (ATE_LOOP_COUNT[k] == -1) { G2Step t = add_in_place_montgomery(c, r, neg_q); ellc.push_back(t.coeffs); r = t.r; }
  }
  G2Wires q1 = mul_by_char_montgomery(c, q);
  G2Wires q2 = mul_by_char_montgomery(c, q1);
  q2 = g2_affine_neg_evaluate(c, q2);
  G2Step t = add_in_place_montgomery(c, r, q1);
  ellc.push_back(t.coeffs);
  r = t.r;
  G2Step t2 = add_in_place_montgomery(c, r, q2);
  ellc.push_back(t2.coeffs);
  return ellc;
}
inline Fq12 ell_montgomery(CircuitContext& c, const Fq12& f, const Fq6& coeffs, const G1Wires& p) {  // pairing.rs:160-171
  Fq2 c0_fq2 = fq2::mul_by_fq_montgomery(c, coeffs.c[0], p.y);
  Fq2 c3_fq2 = fq2::mul_by_fq_montgomery(c, coeffs.c[1], p.x);
  return fq12::mul_by_034_montgomery(c, f, c0_fq2, c3_fq2, coeffs.c[2]);
}
inline Fq12 ell_by_constant_montgomery(CircuitContext& c, const Fq12& f, const HEllCoeff& coeffs, const G1Wires& p) {  // pairing.rs:923-942 (#[component(offcircuit_args)])
  Wires in = concat(concat(f.to_wires(), p.x), concat(p.y, p.z));
  const Fq2Const k0 = coeffs.c0.as_montgomery_const(), k1 = coeffs.c1.as_montgomery_const(), k2 = coeffs.c2.as_montgomery_const();
  std::string kb = k0.c0.key_bytes() + k0.c1.key_bytes() + k1.c0.key_bytes() + k1.c1.key_bytes() + k2.c0.key_bytes() + k2.c1.key_bytes();
  Wires out = component(c, KeyBuilder("pairing::ell_by_constant_montgomery").param("coeffs", kb.data(), kb.size()), in, fq12::N, [&coeffs, &k2](CircuitContext& cc, const Wires& x) {
    Fq12 f = Fq12::from_wires(slice(x, 0, fq12::N));
    Wires px = slice(x, fq12::N, fq12::N + 254), py = slice(x, fq12::N + 254, fq12::N + 508);
    Fq2 new_c0 = fq2::mul_constant_by_fq_montgomery(cc, coeffs.c0, py);
    Fq2 new_c1 = fq2::mul_constant_by_fq_montgomery(cc, coeffs.c1, px);
    return fq12::mul_by_034_constant4_montgomery(cc, f, new_c0, new_c1, k2).to_wires();
  });
  return Fq12::from_wires(out);
}
// pairing.rs:944-1007 (#[component(offcircuit_args = "q1,q2")]): q1, q2 constant affine G2 points (standard form), q3 wires
inline Fq12 multi_miller_loop_groth16_evaluate_montgomery_fast(CircuitContext& c, const G1Wires& p1, const G1Wires& p2, const G1Wires& p3, const HFq2& q1x, const HFq2& q1y,
                                                               const HFq2& q2x, const HFq2& q2y, const G2Wires& q3) {
  auto g1w = [](const G1Wires& p) { return concat(concat(p.x, p.y), p.z); };
  Wires in = concat(concat(concat(g1w(p1), g1w(p2)), g1w(p3)), g2_to_wires(q3));
  const std::vector<HEllCoeff> q1ell = h_ell_coeffs(q1x, q1y), q2ell = h_ell_coeffs(q2x, q2y);
  Wires out = component(c, KeyBuilder("pairing::multi_miller_loop_groth16_evaluate_montgomery_fast"), in, fq12::N, [&q1ell, &q2ell](CircuitContext& cc, const Wires& x) {
    auto g1 = [&](size_t o) { return G1Wires{slice(x, o, o + 254), slice(x, o + 254, o + 508), slice(x, o + 508, o + 762)}; };
    G1Wires p1 = g1(0), p2 = g1(762), p3 = g1(1524);
    G2Wires q3 = g2_from_wires(slice(x, 2286, 2286 + 1524));
    std::vector<Fq6> q3ell = ell_coeffs_montgomery(cc, q3);
    size_t i1 = 0, i2 = 0, i3 = 0;
    Fq12 f = fq12::one_constant();
    auto step = [&]() {
      f = ell_by_constant_montgomery(cc, f, q1ell[i1++], p1);
      f = ell_by_constant_montgomery(cc, f, q2ell[i2++], p2);
      f = ell_montgomery(cc, f, q3ell[i3++], p3);
    };
    for (int i = 64; i >= 1; --i) {  // (1..ATE_LOOP_COUNT.len()).rev()
      if (i != 64) f = fq12::square_montgomery(cc, f);
      step();
      const int8_t bit = ATE_LOOP_COUNT[i - 1];
      if (bit == 1 || bit == -1) step();
    }
    step();
    step();
    if (i1 != q1ell.size() || i3 != q3ell.size()) gsv_panic("internal: line coefficient count mismatch");
    return f.to_wires();
  });
  return Fq12::from_wires(out);
}

// ---- pairing.rs functions that are NOT on the verifier's path, restated for `tools/gate_counts --pairing-csv` (the rows of the reference's
// examples/pairing_gate_counts.rs).  Plain functions here: the reference wraps the two const-Q loops in #[component(offcircuit_args)]
// (pairing.rs:737, 775), which changes neither the gate stream nor the count.
inline G1Wires g1_normalize_to_affine(CircuitContext& c, const G1Wires& p) {  // pairing.rs:173-186
  Wires inv_z = fq::inverse_montgomery(c, p.z);
  Wires inv_z2 = fq::square_montgomery(c, inv_z);
  Wires inv_z3 = fq::mul_montgomery(c, inv_z2, inv_z);
  Wires x = fq::mul_montgomery(c, p.x, inv_z2);
  Wires y = fq::mul_montgomery(c, p.y, inv_z3);
  return G1Wires{x, y, constant_wires(fq_as_montgomery_const(BigU(1)), 254)};
}
inline Fq12 ell_eval_const(CircuitContext& c, const Fq12& f, const HEllCoeff& coeffs, const G1Wires& p) {  // pairing.rs:134-151
  Fq2 c0_fq2 = fq2::mul_constant_by_fq_montgomery(c, coeffs.c0, p.y);
  Fq2 c3_fq2 = fq2::mul_constant_by_fq_montgomery(c, coeffs.c1, p.x);
  return fq12::mul_by_034_constant4_montgomery(c, f, c0_fq2, c3_fq2, coeffs.c2.as_montgomery_const());
}
// pairing.rs:776-843 (one P per constant Q; every P normalised to affine first); with one pair it is miller_loop_const_q's stream (:738-774)
inline Fq12 multi_miller_loop_const_q(CircuitContext& c, const std::vector<G1Wires>& ps, const std::vector<std::pair<HFq2, HFq2>>& qs) {
  if (ps.size() != qs.size()) gsv_panic("multi_miller_loop_const_q: |ps| != |qs|");
  if (ps.empty()) return fq12::one_constant();
  std::vector<std::vector<HEllCoeff>> qells;
  for (const auto& q : qs) qells.push_back(h_ell_coeffs(q.first, q.second));
  std::vector<G1Wires> ps_aff;
  for (const G1Wires& p : ps) ps_aff.push_back(g1_normalize_to_affine(c, p));
  Fq12 f = fq12::one_constant();
  size_t step = 0;
  auto eval_step = [&]() { for (size_t k = 0; k < ps_aff.size(); ++k) f = ell_eval_const(c, f, qells[k][step], ps_aff[k]); ++step; };
  for (int i = 64; i >= 1; --i) {
    if (i != 64) f = fq12::square_montgomery(c, f);
    eval_step();
    const int8_t bit = ATE_LOOP_COUNT[i - 1];
    if (bit == 1 || bit == -1) eval_step();
  }
  eval_step();
  eval_step();
  if (step != qells[0].size()) gsv_panic("internal: line coefficient count mismatch");
  return f;
}
inline Fq12 miller_loop_const_q(CircuitContext& c, const G1Wires& p, const HFq2& qx, const HFq2& qy) {  // pairing.rs:738-774
  return multi_miller_loop_const_q(c, {p}, {{qx, qy}});
}
// pairing.rs:640-698 (variable Qs: their line coefficients come from ell_coeffs_montgomery, all of them BEFORE the loop; inputs assumed
// affine, no normalisation); with one pair it is miller_loop_montgomery_fast's stream (:845-878)
inline Fq12 multi_miller_loop_montgomery_fast(CircuitContext& c, const std::vector<G1Wires>& ps, const std::vector<G2Wires>& qs) {
  std::vector<std::vector<Fq6>> qells;
  for (const G2Wires& q : qs) qells.push_back(ell_coeffs_montgomery(c, q));
  Fq12 f = fq12::one_constant();
  size_t step = 0;
  auto eval_step = [&]() { for (size_t k = 0; k < ps.size(); ++k) f = ell_montgomery(c, f, qells[k][step], ps[k]); ++step; };
  for (int i = 64; i >= 1; --i) {
    if (i != 64) f = fq12::square_montgomery(c, f);
    eval_step();
    const int8_t bit = ATE_LOOP_COUNT[i - 1];
    if (bit == 1 || bit == -1) eval_step();
  }
  eval_step();
  eval_step();
  return f;
}
inline Fq12 miller_loop_montgomery_fast(CircuitContext& c, const G1Wires& p, const G2Wires& q) { return multi_miller_loop_montgomery_fast(c, {p}, {q}); }  // pairing.rs:845-878
}  // namespace pairing

}  // namespace gadgets
}  // namespace gsv
