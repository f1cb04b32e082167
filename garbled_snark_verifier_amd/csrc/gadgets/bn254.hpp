// Gate-stream producers for the BN254 tower on 254-wire Montgomery-form operands:
// Fq (Fp254Impl), Fq2, Fq6, Fq12 — the parts needed by BASELINE configs 2–4.
// Mirrors src/gadgets/bn254/{fp254impl,fq,fq2,fq6,fq12}.rs call-for-call (see bigint.hpp header
// for why order, operands and component boundaries matter).
#pragma once
#include <array>

#include "bigint.hpp"

namespace gsv {
namespace gadgets {

// ----- off-circuit constants (fq.rs:56-76, fp254impl.rs:21-66).  Derivations are checked in
// tests/test_host_constants.py against Python integer arithmetic.
struct FqConst {
  static constexpr size_t N_BITS = 254;
  static const BigU& modulus() { static BigU v = BigU::from_hex("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47"); return v; }
  // MODULUS^-1 mod 2^254 (fq.rs:59-60)
  static const BigU& m_inverse() { static BigU v = BigU::from_hex("0a85dd486e7773942750342fe7cc257f6121829ae1359536782df87d1b799c77"); return v; }
  // 2^254 - MODULUS (fp254impl.rs:58-62)
  static const BigU& not_modulus() { static BigU v = BigU::from_hex("0f9bb18d1ece5fd647afba497e7ea7a2687e956e978e3572c3df73e9278302b9"); return v; }
  // 1/2, 1/3, 2/3 mod p (fq.rs:64-75)
  static const BigU& half_modulus() { static BigU v = BigU::from_hex("183227397098d014dc2822db40c0ac2ecbc0b548b438e5469e10460b6c3e7ea4"); return v; }
  static const BigU& one_third_modulus() { static BigU v = BigU::from_hex("2042def740cbc01bd03583cf0100e593ba56470b9af68708d2c05d6490535385"); return v; }
  static const BigU& two_third_modulus() { static BigU v = BigU::from_hex("10216f7ba065e00de81ac1e7808072c9dd2b2385cd7b438469602eb24829a9c3"); return v; }
  // Fq(1) - Fq(2^254 - p) = (1 - not_modulus) mod p  (fp254impl.rs:164)
  static const BigU& neg_addend() { static BigU v = BigU::from_hex("20c89ce5c263405370a08b6d0302b0bb2f02d522d0e3951a7841182db0f9fa8f"); return v; }
};

namespace fq {
constexpr size_t N = FqConst::N_BITS;
using Fq = BigIntWires;  // fq.rs:22-23

inline void check_len(const Wires& a) { if (a.size() != N) gsv_panic("Fq operand must have 254 wires"); }

// tail shared by add / add_constant / double (fp254impl.rs:101-114)
inline Fq reduce_once_select(CircuitContext& cc, const Wires& wires1, WireId u) {
  Wires wires2 = gadgets::add_constant(cc, wires1, FqConst::not_modulus());
  wires2.pop_back();
  WireId v = gadgets::less_than_constant(cc, wires1, FqConst::modulus());
  WireId s = cc.issue_wire();
  cc.add_gate(Gate::and_variant(u, v, s, true, false, false));
  return gadgets::select(cc, wires1, wires2, s);
}

inline Fq add(CircuitContext& c, const Fq& a, const Fq& b) {  // fp254impl.rs:96-115
  check_len(a); check_len(b);
  return component(c, KeyBuilder("fp254::add"), concat(a, b), N, [](CircuitContext& cc, const Wires& in) {
    Wires wires1 = gadgets::add(cc, slice(in, 0, N), slice(in, N, 2 * N));
    WireId u = wires1.back(); wires1.pop_back();
    return reduce_once_select(cc, wires1, u);
  });
}

inline Fq add_constant(CircuitContext& c, const Fq& a, const BigU& b) {  // fp254impl.rs:117-141
  check_len(a);
  std::string kb = b.key_bytes();
  return component(c, KeyBuilder("fp254::add_constant").param("b", kb.data(), kb.size()), a, N,
                   [&b](CircuitContext& cc, const Wires& in) -> Wires {
    if (b.is_zero()) return in;
    Wires wires1 = gadgets::add_constant(cc, in, b);
    WireId u = wires1.back(); wires1.pop_back();
    return reduce_once_select(cc, wires1, u);
  });
}

inline Fq neg(CircuitContext& c, const Fq& a) {  // fp254impl.rs:153-168
  check_len(a);
  return component(c, KeyBuilder("fp254::neg"), a, N, [](CircuitContext& cc, const Wires& in) {
    Wires not_a = cc.issue_wires(N);
    for (size_t i = 0; i < N; ++i) cc.add_gate(Gate::xor_(in[i], TRUE_WIRE, not_a[i]));
    return fq::add_constant(cc, not_a, FqConst::neg_addend());
  });
}

inline Fq sub(CircuitContext& c, const Fq& a, const Fq& b) {  // fp254impl.rs:143-151
  check_len(a); check_len(b);
  return component(c, KeyBuilder("fp254::sub"), concat(a, b), N, [](CircuitContext& cc, const Wires& in) {
    Fq neg_b = fq::neg(cc, slice(in, N, 2 * N));
    return fq::add(cc, slice(in, 0, N), neg_b);
  });
}

inline Fq double_(CircuitContext& c, const Fq& a) {  // fp254impl.rs:170-191
  check_len(a);
  return component(c, KeyBuilder("fp254::double"), a, N, [](CircuitContext& cc, const Wires& in) {
    Wires shifted = in;
    WireId u = shifted.back(); shifted.pop_back();
    shifted.insert(shifted.begin(), FALSE_WIRE);
    return reduce_once_select(cc, shifted, u);
  });
}

inline Fq half(CircuitContext& c, const Fq& a) {  // fp254impl.rs:193-203
  check_len(a);
  return component(c, KeyBuilder("fp254::half"), a, N, [](CircuitContext& cc, const Wires& in) {
    WireId sel = in[0];
    Wires wires1 = gadgets::half(in);
    Wires wires2 = gadgets::add_constant_without_carry(cc, wires1, FqConst::half_modulus());
    return gadgets::select(cc, wires2, wires1, sel);
  });
}

inline Fq triple(CircuitContext& c, const Fq& a) {  // fp254impl.rs:727-732
  check_len(a);
  return component(c, KeyBuilder("fp254::triple"), a, N, [](CircuitContext& cc, const Wires& in) {
    Fq a2 = fq::double_(cc, in);
    return fq::add(cc, a2, in);
  });
}

inline Fq div6(CircuitContext& c, const Fq& a) {  // fp254impl.rs:734-792
  check_len(a);
  return component(c, KeyBuilder("fp254::div6"), a, N, [](CircuitContext& cc, const Wires& in) {
    Fq h = fq::half(cc, in);
    Wires result = cc.issue_wires(N);  // :739 pre-issued, every entry is overwritten below (zero fan-out wires)
    WireId r1 = FALSE_WIRE, r2 = FALSE_WIRE;
    for (size_t i = 0; i < N; ++i) {
      size_t j = N - 1 - i;
      WireId r2_and_hj = cc.issue_wire();
      cc.add_gate(Gate::and_(r2, h[j], r2_and_hj));
      WireId result_wire = cc.issue_wire();
      cc.add_gate(Gate::or_(r1, r2_and_hj, result_wire));
      result[j] = result_wire;
      WireId new_r1 = cc.issue_wire();
      cc.add_gate(Gate::xor_(r2, result_wire, new_r1));
      r1 = new_r1;
      WireId new_r2 = cc.issue_wire();
      cc.add_gate(Gate::xor_(h[j], result_wire, new_r2));
      r2 = new_r2;
      WireId edge_case = cc.issue_wire();
      cc.add_gate(Gate::nimp(result_wire, h[j], edge_case));
      WireId new_r1b = cc.issue_wire();
      cc.add_gate(Gate::xor_(r1, edge_case, new_r1b));
      r1 = new_r1b;
    }
    Wires plus_third = gadgets::add_constant_without_carry(cc, result, FqConst::one_third_modulus());
    result = gadgets::select(cc, plus_third, result, r2);
    Wires plus_two_third = gadgets::add_constant_without_carry(cc, result, FqConst::two_third_modulus());
    return gadgets::select(cc, plus_two_third, result, r1);
  });
}

inline Fq montgomery_reduce(CircuitContext& c, const Wires& x) {  // fp254impl.rs:303-331
  if (x.size() != 2 * N) gsv_panic("montgomery_reduce: need 508 wires");
  return component(c, KeyBuilder("fp254::montgomery_reduce"), x, N, [](CircuitContext& cc, const Wires& in) {
    Wires x_low = slice(in, 0, 254), x_high = slice(in, 254, in.size());
    Wires q = mul_by_constant_modulo_power_two(cc, x_low, FqConst::m_inverse(), 254);
    Wires prod = mul_by_constant(cc, q, FqConst::modulus());
    Wires sub_ = slice(prod, 254, 508);  // .split_at(254).1.truncate(254)
    WireId bound_check = greater_than(cc, sub_, x_high);
    Wires modulus_wires;  // BigIntWires::new_constant(x_high.len(), modulus)
    {
      std::vector<bool> mb = FqConst::modulus().bits_with_len(x_high.size());
      for (bool b : mb) modulus_wires.push_back(b ? TRUE_WIRE : FALSE_WIRE);
    }
    Wires subtract_if_too_much = self_or_zero(cc, modulus_wires, bound_check);
    Wires new_sub = sub_without_borrow(cc, sub_, subtract_if_too_much);
    return sub_without_borrow(cc, x_high, new_sub);
  });
}

inline Fq mul_montgomery(CircuitContext& c, const Fq& a, const Fq& b) {  // fp254impl.rs:219-230 (not a component)
  check_len(a); check_len(b);
  Wires m = gadgets::mul(c, a, b);
  return montgomery_reduce(c, m);
}
inline Fq square_montgomery(CircuitContext& c, const Fq& a) { return mul_montgomery(c, a, a); }  // fp254impl.rs:283-285
}  // namespace fq

// ------------------------------------------------------------------ fq2.rs (none of these are components)
struct Fq2 {
  std::array<Wires, 2> c;
  Wires to_wires() const { return concat(c[0], c[1]); }  // fq2.rs:33-41
  static Fq2 from_wires(const Wires& w) {                 // fq2.rs:49-60
    if (w.size() != 508) gsv_panic("Fq2::from_wires: need 508 wires");
    return Fq2{{slice(w, 0, 254), slice(w, 254, 508)}};
  }
};
namespace fq2 {
inline Fq2 add(CircuitContext& c, const Fq2& a, const Fq2& b) { return {{fq::add(c, a.c[0], b.c[0]), fq::add(c, a.c[1], b.c[1])}}; }   // fq2.rs:160-168
inline Fq2 sub(CircuitContext& c, const Fq2& a, const Fq2& b) { return {{fq::sub(c, a.c[0], b.c[0]), fq::sub(c, a.c[1], b.c[1])}}; }   // fq2.rs:188-199
inline Fq2 neg(CircuitContext& c, const Fq2& a) { return {{fq::neg(c, a.c[0]), fq::neg(c, a.c[1])}}; }                                   // fq2.rs:179-186
inline Fq2 double_(CircuitContext& c, const Fq2& a) { return {{fq::double_(c, a.c[0]), fq::double_(c, a.c[1])}}; }                       // fq2.rs:201-209
inline Fq2 half(CircuitContext& c, const Fq2& a) { return {{fq::half(c, a.c[0]), fq::half(c, a.c[1])}}; }                                 // fq2.rs:211-219
inline Fq2 triple(CircuitContext& c, const Fq2& a) { Fq2 a2 = double_(c, a); return add(c, a, a2); }                                      // fq2.rs:221-228
inline Fq2 div6(CircuitContext& c, const Fq2& a) { return {{fq::div6(c, a.c[0]), fq::div6(c, a.c[1])}}; }                                 // fq2.rs:386-394
// fq2.rs:230-255.  The reference does not make the Fq2 multiplication / squaring components; here they are wrapped in one (as the
// cyclotomic squaring below): a component boundary changes nothing in the gate stream — a wire is dead iff nothing reads it, wherever
// the boundary is; the oracle's fixtures, generated before the wrappers existed, did not move — and it gives a plan a unit of the
// granularity at which an instance has width: an Fq12 multiplication is 15 independent Fq2 multiplications (fq12.rs:199-221,
// fq6.rs:194-260), which a session runs side by side on 15 CUs (schedule.hpp).
inline Fq2 mul_montgomery(CircuitContext& c0, const Fq2& a_, const Fq2& b_) {
  Wires out = component(c0, KeyBuilder("fq2::mul_montgomery"), concat(a_.to_wires(), b_.to_wires()), 508, [](CircuitContext& c, const Wires& in) {
    const Fq2 a = Fq2::from_wires(slice(in, 0, 508)), b = Fq2::from_wires(slice(in, 508, 1016));
    Wires a_sum = fq::add(c, a.c[0], a.c[1]);
    Wires b_sum = fq::add(c, b.c[0], b.c[1]);
    Wires a0_b0 = fq::mul_montgomery(c, a.c[0], b.c[0]);
    Wires a1_b1 = fq::mul_montgomery(c, a.c[1], b.c[1]);
    Wires sum_prod = fq::mul_montgomery(c, a_sum, b_sum);
    Wires c0 = fq::sub(c, a0_b0, a1_b1);
    Wires sum_a0b0_a1b1 = fq::add(c, a0_b0, a1_b1);
    Wires c1 = fq::sub(c, sum_prod, sum_a0b0_a1b1);
    return Fq2{{c0, c1}}.to_wires();
  });
  return Fq2::from_wires(out);
}
inline Fq2 mul_by_nonresidue(CircuitContext& c, const Fq2& a) {  // fq2.rs:324-339
  Wires a0_3 = fq::triple(c, a.c[0]);
  Wires a0_9 = fq::triple(c, a0_3);
  Wires a1_3 = fq::triple(c, a.c[1]);
  Wires a1_9 = fq::triple(c, a1_3);
  Wires c0 = fq::sub(c, a0_9, a.c[1]);
  Wires c1 = fq::add(c, a1_9, a.c[0]);
  return {{c0, c1}};
}
inline Fq2 square_montgomery(CircuitContext& c0, const Fq2& a_) {  // fq2.rs:341-354 (wrapped like mul_montgomery above)
  Wires out = component(c0, KeyBuilder("fq2::square_montgomery"), a_.to_wires(), 508, [](CircuitContext& c, const Wires& in) {
    const Fq2 a = Fq2::from_wires(in);
    Wires a0_plus_a1 = fq::add(c, a.c[0], a.c[1]);
    Wires a0_minus_a1 = fq::sub(c, a.c[0], a.c[1]);
    Wires a0_a1 = fq::mul_montgomery(c, a.c[0], a.c[1]);
    Wires c0 = fq::mul_montgomery(c, a0_plus_a1, a0_minus_a1);
    Wires c1 = fq::double_(c, a0_a1);
    return Fq2{{c0, c1}}.to_wires();
  });
  return Fq2::from_wires(out);
}
}  // namespace fq2

// ------------------------------------------------------------------ fq6.rs
struct Fq6 {
  std::array<Fq2, 3> c;
  Wires to_wires() const { return concat(concat(c[0].to_wires(), c[1].to_wires()), c[2].to_wires()); }  // fq6.rs:17-25
  static Fq6 from_wires(const Wires& w) {  // fq6.rs:37-47
    if (w.size() != 1524) gsv_panic("Fq6::from_wires: need 1524 wires");
    return Fq6{{Fq2::from_wires(slice(w, 0, 508)), Fq2::from_wires(slice(w, 508, 1016)), Fq2::from_wires(slice(w, 1016, 1524))}};
  }
};
namespace fq6 {
inline Fq6 add(CircuitContext& c, const Fq6& a, const Fq6& b) { return {{fq2::add(c, a.c[0], b.c[0]), fq2::add(c, a.c[1], b.c[1]), fq2::add(c, a.c[2], b.c[2])}}; }  // fq6.rs:154-160
inline Fq6 sub(CircuitContext& c, const Fq6& a, const Fq6& b) { return {{fq2::sub(c, a.c[0], b.c[0]), fq2::sub(c, a.c[1], b.c[1]), fq2::sub(c, a.c[2], b.c[2])}}; }  // fq6.rs:170-176
inline Fq6 double_(CircuitContext& c, const Fq6& a) { return {{fq2::double_(c, a.c[0]), fq2::double_(c, a.c[1]), fq2::double_(c, a.c[2])}}; }                       // fq6.rs:178-184
inline Fq6 div6(CircuitContext& c, const Fq6& a) { return {{fq2::div6(c, a.c[0]), fq2::div6(c, a.c[1]), fq2::div6(c, a.c[2])}}; }                                   // fq6.rs:186-192
inline Fq6 mul_by_nonresidue(CircuitContext& c, const Fq6& a) {  // fq6.rs:346-349
  Fq2 u = fq2::mul_by_nonresidue(c, a.c[2]);
  return {{u, a.c[0], a.c[1]}};
}
// fq6.rs:194-260, wrapped in a component like the Fq2 multiplication above (stream-neutral): the unit at which an Fq12 multiplication /
// squaring has its width — three / two Fq6 multiplications that a plan session runs side by side, each with its own pre- and
// post-additions inside (cutting finer, at the Fq2 multiplications, doubles the dependency depth: schedule.hpp).
inline Fq6 mul_montgomery(CircuitContext& c0_, const Fq6& a_, const Fq6& b_) {
  Wires out = component(c0_, KeyBuilder("fq6::mul_montgomery"), concat(a_.to_wires(), b_.to_wires()), 1524, [](CircuitContext& c, const Wires& in) {
  const Fq6 a = Fq6::from_wires(slice(in, 0, 1524)), b = Fq6::from_wires(slice(in, 1524, 3048));
  const Fq2 &a_c0 = a.c[0], &a_c1 = a.c[1], &a_c2 = a.c[2];
  const Fq2 &b_c0 = b.c[0], &b_c1 = b.c[1], &b_c2 = b.c[2];
  Fq2 v0 = fq2::mul_montgomery(c, a_c0, b_c0);

  Fq2 wires_2 = fq2::add(c, a_c0, a_c2);
  Fq2 wires_3 = fq2::add(c, wires_2, a_c1);
  Fq2 wires_4 = fq2::sub(c, wires_2, a_c1);
  Fq2 wires_5 = fq2::double_(c, a_c1);
  Fq2 wires_6 = fq2::double_(c, a_c2);
  Fq2 wires_7 = fq2::double_(c, wires_6);
  Fq2 wires_8 = fq2::add(c, a_c0, wires_5);
  Fq2 wires_9 = fq2::add(c, wires_8, wires_7);

  Fq2 wires_10 = fq2::add(c, b_c0, b_c2);
  Fq2 wires_11 = fq2::add(c, wires_10, b_c1);
  Fq2 wires_12 = fq2::sub(c, wires_10, b_c1);
  Fq2 wires_13 = fq2::double_(c, b_c1);
  Fq2 wires_14 = fq2::double_(c, b_c2);
  Fq2 wires_15 = fq2::double_(c, wires_14);
  Fq2 wires_16 = fq2::add(c, b_c0, wires_13);
  Fq2 wires_17 = fq2::add(c, wires_16, wires_15);

  Fq2 v1 = fq2::mul_montgomery(c, wires_3, wires_11);
  Fq2 v2 = fq2::mul_montgomery(c, wires_4, wires_12);
  Fq2 v3 = fq2::mul_montgomery(c, wires_9, wires_17);
  Fq2 v4 = fq2::mul_montgomery(c, a_c2, b_c2);

  Fq2 v2_2 = fq2::double_(c, v2);

  Fq2 v0_3 = fq2::triple(c, v0);
  Fq2 v1_3 = fq2::triple(c, v1);
  Fq2 v2_3 = fq2::triple(c, v2);
  Fq2 v4_3 = fq2::triple(c, v4);

  Fq2 v0_6 = fq2::double_(c, v0_3);
  Fq2 v1_6 = fq2::double_(c, v1_3);
  Fq2 v4_6 = fq2::double_(c, v4_3);

  Fq2 v4_12 = fq2::double_(c, v4_6);

  Fq2 wires_18 = fq2::sub(c, v0_3, v1_3);
  Fq2 wires_19 = fq2::sub(c, wires_18, v2);
  Fq2 wires_20 = fq2::add(c, wires_19, v3);
  Fq2 wires_21 = fq2::sub(c, wires_20, v4_12);
  Fq2 wires_22 = fq2::mul_by_nonresidue(c, wires_21);
  Fq2 c0 = fq2::add(c, wires_22, v0_6);

  Fq2 wires_23 = fq2::sub(c, v1_6, v0_3);
  Fq2 wires_24 = fq2::sub(c, wires_23, v2_2);
  Fq2 wires_25 = fq2::sub(c, wires_24, v3);
  Fq2 wires_26 = fq2::add(c, wires_25, v4_12);
  Fq2 wires_27 = fq2::mul_by_nonresidue(c, v4_6);
  Fq2 c1 = fq2::add(c, wires_26, wires_27);

  Fq2 wires_28 = fq2::sub(c, v1_3, v0_6);
  Fq2 wires_29 = fq2::add(c, wires_28, v2_3);
  Fq2 c2 = fq2::sub(c, wires_29, v4_6);

  Fq6 result{{c0, c1, c2}};
  return div6(c, result).to_wires();
  });
  return Fq6::from_wires(out);
}
}  // namespace fq6

// ------------------------------------------------------------------ fq12.rs
struct Fq12 {
  std::array<Fq6, 2> c;
  Wires to_wires() const { return concat(c[0].to_wires(), c[1].to_wires()); }  // fq12.rs:17-24
  static Fq12 from_wires(const Wires& w) {  // fq12.rs:37-49
    if (w.size() != 3048) gsv_panic("Fq12::from_wires: need 3048 wires");
    return Fq12{{Fq6::from_wires(slice(w, 0, 1524)), Fq6::from_wires(slice(w, 1524, 3048))}};
  }
};
namespace fq12 {
constexpr size_t N = 3048;
inline Fq12 mul_montgomery(CircuitContext& c, const Fq12& a, const Fq12& b) {  // fq12.rs:198-221  (#[component])
  Wires out = component(c, KeyBuilder("fq12::mul_montgomery"), concat(a.to_wires(), b.to_wires()), N,
                        [](CircuitContext& cc, const Wires& in) {
    Fq12 a = Fq12::from_wires(slice(in, 0, N)), b = Fq12::from_wires(slice(in, N, 2 * N));
    Fq6 a_sum = fq6::add(cc, a.c[0], a.c[1]);
    Fq6 b_sum = fq6::add(cc, b.c[0], b.c[1]);
    Fq6 a0_b0 = fq6::mul_montgomery(cc, a.c[0], b.c[0]);
    Fq6 a1_b1 = fq6::mul_montgomery(cc, a.c[1], b.c[1]);
    Fq6 sum_a0b0_a1b1 = fq6::add(cc, a0_b0, a1_b1);
    Fq6 sum_prod = fq6::mul_montgomery(cc, a_sum, b_sum);
    Fq6 a1_b1_nonres = fq6::mul_by_nonresidue(cc, a1_b1);
    Fq6 c0 = fq6::add(cc, a0_b0, a1_b1_nonres);
    Fq6 c1 = fq6::sub(cc, sum_prod, sum_a0b0_a1b1);
    return Fq12{{c0, c1}}.to_wires();
  });
  return Fq12::from_wires(out);
}
inline Fq12 square_montgomery(CircuitContext& c, const Fq12& a) {  // fq12.rs:311-324  (#[component])
  Wires out = component(c, KeyBuilder("fq12::square_montgomery"), a.to_wires(), N, [](CircuitContext& cc, const Wires& in) {
    Fq12 a = Fq12::from_wires(in);
    Fq6 w1 = fq6::add(cc, a.c[0], a.c[1]);
    Fq6 w2 = fq6::mul_by_nonresidue(cc, a.c[1]);
    Fq6 w3 = fq6::add(cc, a.c[0], w2);
    Fq6 w4 = fq6::mul_montgomery(cc, a.c[0], a.c[1]);
    Fq6 w5 = fq6::mul_montgomery(cc, w1, w3);
    Fq6 w6 = fq6::mul_by_nonresidue(cc, w4);
    Fq6 w7 = fq6::add(cc, w4, w6);
    Fq6 c0 = fq6::sub(cc, w5, w7);
    Fq6 c1 = fq6::double_(cc, w4);
    return Fq12{{c0, c1}}.to_wires();
  });
  return Fq12::from_wires(out);
}
// Granger-Scott squaring in the cyclotomic subgroup (fq12.rs:326-392).  The reference does not make this a component; a
// component boundary changes nothing in the gate stream (a wire is dead iff nothing reads it, wherever the boundary is),
// and having one lets a plan record the 186 cyclotomic squarings of the final exponentiation once instead of 186 times.
inline Fq12 cyclotomic_square_montgomery(CircuitContext& c0_, const Fq12& a_) {
  Wires out = component(c0_, KeyBuilder("fq12::cyclotomic_square_montgomery"), a_.to_wires(), N, [](CircuitContext& c, const Wires& in) {
  const Fq12 a = Fq12::from_wires(in);
  const Fq2 &c0 = a.c[0].c[0], &c1 = a.c[0].c[1], &c2 = a.c[0].c[2], &c3 = a.c[1].c[0], &c4 = a.c[1].c[1], &c5 = a.c[1].c[2];
  // one "Fq4 squaring" of the pair (x, y); which operand is multiplied by the non-residue follows the reference line by line
  auto fp4 = [&](const Fq2& x, const Fq2& y, const Fq2& beta_of, const Fq2& added_to, Fq2& t_even, Fq2& t_odd) {
    Fq2 xy = fq2::mul_montgomery(c, x, y);
    Fq2 x_plus_y = fq2::add(c, x, y);
    Fq2 y_beta = fq2::mul_by_nonresidue(c, beta_of);
    Fq2 x_plus_y_beta = fq2::add(c, added_to, y_beta);
    Fq2 xy_beta = fq2::mul_by_nonresidue(c, xy);
    Fq2 w1 = fq2::mul_montgomery(c, x_plus_y, x_plus_y_beta);
    Fq2 w2 = fq2::add(c, xy, xy_beta);
    t_even = fq2::sub(c, w1, w2);
    t_odd = fq2::double_(c, xy);
  };
  Fq2 t0, t1, t2, t3, t4, t5;
  fp4(c0, c4, c4, c0, t0, t1);  // fq12.rs:337-345
  fp4(c2, c3, c2, c3, t2, t3);  // fq12.rs:347-355: y_beta = nonresidue * c2, x_plus_y_beta = c3 + y_beta
  fp4(c1, c5, c5, c1, t4, t5);  // fq12.rs:357-365
  auto three_minus = [&](const Fq2& t, const Fq2& cc) { Fq2 w1 = fq2::sub(c, t, cc); Fq2 w2 = fq2::double_(c, w1); return fq2::add(c, w2, t); };
  auto three_plus = [&](const Fq2& t, const Fq2& cc) { Fq2 w1 = fq2::add(c, t, cc); Fq2 w2 = fq2::double_(c, w1); return fq2::add(c, w2, t); };
  Fq2 z0 = three_minus(t0, c0);  // fq12.rs:367-369
  Fq2 z4 = three_minus(t2, c1);  // :371-373
  Fq2 z3 = three_minus(t4, c2);  // :375-377
  Fq2 t5_beta = fq2::mul_by_nonresidue(c, t5);
  Fq2 z2 = three_plus(t5_beta, c3);  // :379-382
  Fq2 z1 = three_plus(t1, c4);       // :384-386
  Fq2 z5 = three_plus(t3, c5);       // :388-390
  return Fq12{{Fq6{{z0, z4, z3}}, Fq6{{z2, z1, z5}}}}.to_wires();
  });
  return Fq12::from_wires(out);
}
}  // namespace fq12

}  // namespace gadgets
}  // namespace gsv
