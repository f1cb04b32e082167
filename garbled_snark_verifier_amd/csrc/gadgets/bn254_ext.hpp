// Gate-stream producers towards the verifier's final exponentiation: modular inverse, constant
// multiplications, Frobenius maps, conjugation, Fq2/Fq6/Fq12 inverses, exp_by_neg_x and
// final_exponentiation_montgomery.  Mirrors src/gadgets/bigint/{add,cmp}.rs, src/gadgets/bn254/{fp254impl,fq2,fq6,fq12,
// final_exponentiation}.rs call-for-call: gate order fixes gate ids, operand order fixes the half-gate ciphertexts.
// (Component boundaries do not change results — a wire is dead iff nothing reads it — but they are kept where the
// reference has them because they are what a plan can turn into calls.)
#pragma once
#include "bn254.hpp"

namespace gsv {
namespace gadgets {

// ------------------------------------------------------------------ off-circuit constants
struct ExtConst {
  // R = 2^254 mod p: Fq::as_montgomery(ONE) (fp254impl.rs:236-238)
  static const BigU& r_mod_p() { static BigU v = BigU::from_hex("0f9bb18d1ece5fd647afba497e7ea7a2687e956e978e3572c3df73e9278302b9"); return v; }
  // R^3 mod p: the constant of inverse_montgomery (fp254impl.rs:668-678)
  static const BigU& r3_mod_p() { static BigU v = BigU::from_hex("11e801eb89a33a2411fad539c6b5e3ed8ffe4a8fdc2585c19c62e80ef5353fc9"); return v; }
  static const BigU& p_minus_1() { static BigU v = BigU::from_hex("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd46"); return v; }
  static const BigU& one() { static BigU v(1); return v; }
  static const BigU& two() { static BigU v(2); return v; }
};

// (a + b) mod m / (a - b) mod m / a * 2^254 mod p on host constants (all < m)
inline BigU bigu_from_limbs(std::vector<uint32_t> l) {
  static const char* hx = "0123456789abcdef";
  std::string s;
  for (size_t i = l.size(); i-- > 0;)
    for (int n = 7; n >= 0; --n) s.push_back(hx[(l[i] >> (4 * n)) & 15u]);
  if (s.empty()) s = "0";
  return BigU::from_hex(s);
}
inline int bigu_cmp(const BigU& a, const BigU& b) {
  const auto &x = a.limbs(), &y = b.limbs();
  if (x.size() != y.size()) return x.size() < y.size() ? -1 : 1;
  for (size_t i = x.size(); i-- > 0;) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}
inline BigU bigu_add(const BigU& a, const BigU& b) {
  std::vector<uint32_t> r;
  uint64_t carry = 0;
  for (size_t i = 0; i < std::max(a.limbs().size(), b.limbs().size()) || carry; ++i) {
    uint64_t s = carry + (i < a.limbs().size() ? a.limbs()[i] : 0) + (i < b.limbs().size() ? b.limbs()[i] : 0);
    r.push_back(uint32_t(s));
    carry = s >> 32;
  }
  return bigu_from_limbs(r);
}
inline BigU bigu_sub(const BigU& a, const BigU& b) {  // a >= b
  std::vector<uint32_t> r;
  int64_t borrow = 0;
  for (size_t i = 0; i < a.limbs().size(); ++i) {
    int64_t d = int64_t(a.limbs()[i]) - (i < b.limbs().size() ? int64_t(b.limbs()[i]) : 0) - borrow;
    borrow = d < 0;
    if (d < 0) d += (int64_t(1) << 32);
    r.push_back(uint32_t(d));
  }
  return bigu_from_limbs(r);
}
inline BigU fq_add_const(const BigU& a, const BigU& b) {
  BigU s = bigu_add(a, b);
  return bigu_cmp(s, FqConst::modulus()) >= 0 ? bigu_sub(s, FqConst::modulus()) : s;
}
inline BigU fq_sub_const(const BigU& a, const BigU& b) { return bigu_cmp(a, b) >= 0 ? bigu_sub(a, b) : bigu_sub(bigu_add(a, FqConst::modulus()), b); }
inline BigU fq_double_const(const BigU& a) { return fq_add_const(a, a); }
struct Fq2Const { BigU c0, c1; };

inline Wires constant_wires(const BigU& v, size_t n) {  // BigIntWires::new_constant (bigint/mod.rs:76-88)
  Wires w;
  for (bool b : v.bits_with_len(n)) w.push_back(b ? TRUE_WIRE : FALSE_WIRE);
  return w;
}

// ------------------------------------------------------------------ bigint/add.rs, cmp.rs
inline BigIntWires double_without_overflow(CircuitContext& c, const BigIntWires& a) {  // add.rs:134-141 (a component without gates)
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::double_without_overflow"), a, n, [n](CircuitContext&, const Wires& in) {
    Wires r;
    r.push_back(FALSE_WIRE);
    for (size_t i = 0; i + 1 < n; ++i) r.push_back(in[i]);
    return r;
  });
}
inline BigIntWires self_or_zero_inv(CircuitContext& c, const BigIntWires& a, WireId s) {  // cmp.rs:25-41: a AND NOT s
  const size_t n = a.size();
  Wires in = a; in.push_back(s);
  return component(c, KeyBuilder("bigint::self_or_zero_inv"), in, n, [n](CircuitContext& cc, const Wires& in) {
    Wires bits; bits.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      WireId w = cc.issue_wire();
      cc.add_gate(Gate::and_variant(in[i], in[n], w, false, true, false));
      bits.push_back(w);
    }
    return bits;
  });
}
// add.rs:155-190: {odd part of a, 2^(trailing zeros) as a one-hot-ish "even part"}
inline std::array<BigIntWires, 2> odd_part(CircuitContext& c, const BigIntWires& a) {
  const size_t n = a.size();
  Wires select_bn = c.issue_wires(n - 1);
  select_bn.insert(select_bn.begin(), a[0]);
  for (size_t i = 1; i < n; ++i) c.add_gate(Gate::or_(select_bn[i - 1], a[i], select_bn[i]));
  Wires k = c.issue_wires(n - 1);
  k.insert(k.begin(), a[0]);
  for (size_t i = 1; i < n; ++i) c.add_gate(Gate::and_variant(select_bn[i - 1], a[i], k[i], true, false, false));
  Wires odd_acc = a;
  for (size_t i = 0; i < n; ++i) {
    Wires half_res = half(odd_acc);
    odd_acc = select(c, odd_acc, half_res, select_bn[i]);
  }
  return {odd_acc, k};
}

namespace fq {
constexpr size_t INV_PER_CHUNK = 4;  // fp254impl.rs:397
constexpr size_t INV_GROUP_CHUNKS = 64;  // engine extension: `inverse_iteration` components per `inverse::iteration_group` wrapper (see fq::inverse)
inline WireId equal_constant(CircuitContext& c, const Fq& a, const BigU& b) { return gadgets::equal_constant(c, a, b); }  // fp254impl.rs:87-93

// fp254impl.rs:254-275 (b is what the reference passes as `&ark_bn254::Fq`: its integer value)
inline Fq mul_by_constant_montgomery(CircuitContext& c, const Fq& a, const BigU& b) {
  check_len(a);
  std::string kb = b.key_bytes();
  return component(c, KeyBuilder("fp254::mul_by_constant_montgomery").param("b", kb.data(), kb.size()), a, N, [&b](CircuitContext& cc, const Wires& in) -> Wires {
    if (b.is_zero()) return constant_wires(BigU(), in.size());
    if (b == ExtConst::r_mod_p()) return in;
    Wires m = gadgets::mul_by_constant(cc, in, b);
    return montgomery_reduce(cc, m);
  });
}

// fp254impl.rs:333-663: binary extended Euclid ("almost inverse" with the power of two divided out afterwards)
inline Fq inverse(CircuitContext& c, const Fq& a) {
  check_len(a);
  return component(c, KeyBuilder("fp254::inverse"), a, N, [](CircuitContext& cc, const Wires& a) -> Wires {
    std::array<BigIntWires, 2> oe = gadgets::odd_part(cc, a);
    const BigIntWires &odd = oe[0], &even_part = oe[1];
    Fq neg_odd = fq::neg(cc, odd);
    Wires u = gadgets::half(neg_odd), v = odd;
    Wires k = constant_wires(ExtConst::one(), N), r = constant_wires(ExtConst::one(), N), s = constant_wires(ExtConst::two(), N);
    // Engine extension (stream-neutral, like fp254::exp_chunk): the 127 `inverse_iteration` components (fp254impl.rs:397-640) are wrapped in
    // groups of INV_GROUP_CHUNKS consecutive ones — a component boundary the reference does not have.  A boundary never changes the gate
    // stream (a wire is dead iff nothing reads it, wherever the boundary is: the oracle's hashes of fq_inverse / fq12_inverse did not move);
    // as a plan UNIT a group is one call whose iterations' chained adders overlap, where 64 separate calls of four iterations each cost
    // 37 % more device steps.  Its ciphertext block (4.4 M records) stays below an Fq12 multiplication's.
    for (size_t g0 = 0; g0 < 2 * N; g0 += INV_PER_CHUNK * INV_GROUP_CHUNKS) {
      const size_t g1 = std::min(2 * N, g0 + INV_PER_CHUNK * INV_GROUP_CHUNKS);
      uint64_t n_group_it = g1 - g0;
      ComponentKey gkey = KeyBuilder("inverse::iteration_group").param("iterations", &n_group_it, sizeof n_group_it).finish(5 * N, 5 * N);
      Wires gout = cc.with_named_child(gkey, concat(concat(concat(concat(u, v), r), s), k), [g0, g1](CircuitContext& cg, const Wires& gx) -> Wires {
        Wires u = slice(gx, 0, N), v = slice(gx, N, 2 * N), r = slice(gx, 2 * N, 3 * N), s = slice(gx, 3 * N, 4 * N), k = slice(gx, 4 * N, 5 * N);
          for (size_t it0 = g0; it0 < g1; it0 += INV_PER_CHUNK) {
            const size_t n_it = std::min(INV_PER_CHUNK, 2 * N - it0);
            Wires in = concat(concat(concat(concat(u, v), r), s), k);
            ComponentKey key = KeyBuilder("inverse_iteration").finish(5 * N, 5 * N);
            Wires out = cg.with_named_child(key, in, [n_it](CircuitContext& c3, const Wires& x) -> Wires {
              Wires u = slice(x, 0, N), v = slice(x, N, 2 * N), r = slice(x, 2 * N, 3 * N), s = slice(x, 3 * N, 4 * N), k = slice(x, 4 * N, 5 * N);
              for (size_t it = 0; it < n_it; ++it) {
                const WireId not_x1 = u[0], not_x2 = v[0];
                const WireId x3 = gadgets::greater_than(c3, u, v);
                const WireId p2 = c3.issue_wire();
                c3.add_gate(Gate::and_variant(not_x1, not_x2, p2, false, true, false));
                const WireId p3 = c3.issue_wire();
                const WireId wires_2 = c3.issue_wire();
                c3.add_gate(Gate::and_(not_x1, not_x2, wires_2));
                c3.add_gate(Gate::and_(wires_2, x3, p3));
                const WireId p4 = c3.issue_wire();
                c3.add_gate(Gate::nimp(wires_2, x3, p4));
                // part 1
                Wires u1 = gadgets::half(u), v1 = v, r1 = r;
                Wires s1 = double_without_overflow(c3, s);
                Wires k1 = gadgets::add_constant_without_carry(c3, k, ExtConst::one());
                // part 2
                Wires u2 = u, v2 = gadgets::half(v);
                Wires r2 = double_without_overflow(c3, r);
                Wires s2 = s;
                Wires k2 = gadgets::add_constant_without_carry(c3, k, ExtConst::one());
                // part 3
                Wires u3 = gadgets::sub_without_borrow(c3, u1, v2);
                Wires v3 = v;
                Wires r3 = gadgets::add_without_carry(c3, r, s);
                Wires s3 = double_without_overflow(c3, s);
                Wires k3 = gadgets::add_constant_without_carry(c3, k, ExtConst::one());
                // part 4
                Wires u4 = u;
                Wires v4 = gadgets::sub_without_borrow(c3, v2, u1);
                Wires r4 = double_without_overflow(c3, r);
                Wires s4 = gadgets::add_without_carry(c3, r, s);
                Wires k4 = gadgets::add_constant_without_carry(c3, k, ExtConst::one());
                auto mix = [&](const Wires& w1, const Wires& w2, const Wires& w3, const Wires& w4) {
                  Wires t1 = self_or_zero_inv(c3, w1, not_x1);
                  Wires t2 = gadgets::self_or_zero(c3, w2, p2);
                  Wires t3 = gadgets::self_or_zero(c3, w3, p3);
                  Wires t4 = gadgets::self_or_zero(c3, w4, p4);
                  Wires a1 = gadgets::add_without_carry(c3, t1, t2);
                  Wires a2 = gadgets::add_without_carry(c3, a1, t3);
                  return gadgets::add_without_carry(c3, a2, t4);
                };
                Wires new_u = mix(u1, u2, u3, u4);
                Wires new_v = mix(v1, v2, v3, v4);
                Wires new_r = mix(r1, r2, r3, r4);
                Wires new_s = mix(s1, s2, s3, s4);
                Wires new_k = mix(k1, k2, k3, k4);
                const WireId v_equals_one = gadgets::equal_constant(c3, v, ExtConst::one());
                u = gadgets::select(c3, u, new_u, v_equals_one);
                v = gadgets::select(c3, v, new_v, v_equals_one);
                r = gadgets::select(c3, r, new_r, v_equals_one);
                s = gadgets::select(c3, s, new_s, v_equals_one);
                k = gadgets::select(c3, k, new_k, v_equals_one);
              }
              return concat(concat(concat(concat(u, v), r), s), k);
            }, 5 * N);
            u = slice(out, 0, N); v = slice(out, N, 2 * N); r = slice(out, 2 * N, 3 * N); s = slice(out, 3 * N, 4 * N); k = slice(out, 4 * N, 5 * N);
          }
        return concat(concat(concat(concat(u, v), r), s), k);
      }, 5 * N);
      u = slice(gout, 0, N); v = slice(gout, N, 2 * N); r = slice(gout, 2 * N, 3 * N); s = slice(gout, 3 * N, 4 * N); k = slice(gout, 4 * N, 5 * N);
    }
    // Engine extension (stream-neutral): the two division chains — sibling components of the reference, consecutive in its stream — sit in ONE
    // wrapper component.  As one plan unit the 2^k chain's counter updates (508 full-width comparisons: 131 k dependent device steps) run
    // beside the even-part chain (64 k steps) instead of behind it, as they do in the flat stream.
    ComponentKey dkey = KeyBuilder("inverse::divide_chains").finish(N, 3 * N);
    return cc.with_named_child(dkey, concat(concat(s, even_part), k), [](CircuitContext& cd, const Wires& dx) -> Wires {
      Wires s = slice(dx, 0, N), even_part = slice(dx, N, 2 * N), k = slice(dx, 2 * N, 3 * N);
      // divide the result by the even part of the input
      {
        ComponentKey key = KeyBuilder("inverse::divide_result_by_even_part").finish(2 * N, 2 * N);
        Wires even = even_part;
        Wires out = cd.with_named_child(key, concat(s, even), [](CircuitContext& c3, const Wires& x) -> Wires {
          Wires s = slice(x, 0, N), even_part = slice(x, N, 2 * N);
          size_t chunk_idx = 0;
          for (size_t it0 = 0; it0 < N; it0 += INV_PER_CHUNK, ++chunk_idx) {
            const size_t n_it = std::min(INV_PER_CHUNK, N - it0);
            uint64_t ci = chunk_idx;
            ComponentKey ck = KeyBuilder("inverse::divide_result_by_even_part::chunk").param("chunk_idx", &ci, sizeof ci).finish(2 * N, 2 * N);
            Wires o2 = c3.with_named_child(ck, concat(s, even_part), [n_it](CircuitContext& c4, const Wires& y) -> Wires {
              Wires s = slice(y, 0, N), even_part = slice(y, N, 2 * N);
              for (size_t it = 0; it < n_it; ++it) {
                Wires updated_s = fq::half(c4, s);
                Wires updated_even = fq::half(c4, even_part);
                const WireId sel = gadgets::equal_constant(c4, even_part, ExtConst::one());
                s = gadgets::select(c4, s, updated_s, sel);
                even_part = gadgets::select(c4, even_part, updated_even, sel);
              }
              return concat(s, even_part);
            }, 2 * N);
            s = slice(o2, 0, N); even_part = slice(o2, N, 2 * N);
          }
          return s;
        }, N);
        s = out;
      }
      // divide the result by 2^k
      ComponentKey key = KeyBuilder("inverse::divide_result_by_2^k").finish(2 * N, 2 * N);
      return cd.with_named_child(key, concat(s, k), [](CircuitContext& c3, const Wires& x) -> Wires {
        Wires s = slice(x, 0, N), k = slice(x, N, 2 * N);
        for (size_t it0 = 0; it0 < 2 * N; it0 += INV_PER_CHUNK) {
          const size_t n_it = std::min(INV_PER_CHUNK, 2 * N - it0);
          ComponentKey ck = KeyBuilder("inverse::divide_result_by_2^k::chunk").finish(2 * N, 2 * N);
          Wires o2 = c3.with_named_child(ck, concat(s, k), [n_it](CircuitContext& c4, const Wires& y) -> Wires {
            Wires s = slice(y, 0, N), k = slice(y, N, 2 * N);
            for (size_t it = 0; it < n_it; ++it) {
              Wires updated_s = fq::half(c4, s);
              Wires updated_k = fq::add_constant(c4, k, ExtConst::p_minus_1());
              const WireId sel = fq::equal_constant(c4, k, BigU());
              s = gadgets::select(c4, s, updated_s, sel);
              k = gadgets::select(c4, k, updated_k, sel);
            }
            return concat(s, k);
          }, 2 * N);
          s = slice(o2, 0, N); k = slice(o2, N, 2 * N);
        }
        return s;
      }, N);
    }, N);
  });
}
inline Fq inverse_montgomery(CircuitContext& c, const Fq& a) {  // fp254impl.rs:665-678
  Fq b = inverse(c, a);
  return mul_by_constant_montgomery(c, b, ExtConst::r3_mod_p());
}
}  // namespace fq

namespace fq2 {
// fq2.rs:257-280.  `b` is the Fq2 constant exactly as the reference passes it.
inline Fq2 mul_by_constant_montgomery(CircuitContext& c, const Fq2& a, const Fq2Const& b) {
  if (b.c0 == ExtConst::one() && b.c1.is_zero()) return a;
  Wires a_sum = fq::add(c, a.c[0], a.c[1]);
  Wires a0_b0 = fq::mul_by_constant_montgomery(c, a.c[0], b.c0);
  Wires a1_b1 = fq::mul_by_constant_montgomery(c, a.c[1], b.c1);
  BigU bsum = fq_add_const(b.c0, b.c1);
  Wires sum_mul_sum = fq::mul_by_constant_montgomery(c, a_sum, bsum);
  Wires c0 = fq::sub(c, a0_b0, a1_b1);
  Wires a0b0_plus_a1b1 = fq::add(c, a0_b0, a1_b1);
  Wires c1 = fq::sub(c, sum_mul_sum, a0b0_plus_a1b1);
  return Fq2{{c0, c1}};
}
inline Fq2 inverse_montgomery(CircuitContext& c, const Fq2& a) {  // fq2.rs:356-372 (#[component])
  Wires out = component(c, KeyBuilder("fq2::inverse_montgomery"), a.to_wires(), 508, [](CircuitContext& cc, const Wires& in) {
    Fq2 a = Fq2::from_wires(in);
    Wires a0_square = fq::square_montgomery(cc, a.c[0]);
    Wires a1_square = fq::square_montgomery(cc, a.c[1]);
    Wires norm = fq::add(cc, a0_square, a1_square);
    Wires inverse_norm = fq::inverse_montgomery(cc, norm);
    Wires c0 = fq::mul_montgomery(cc, a.c[0], inverse_norm);
    Wires neg_a1 = fq::neg(cc, a.c[1]);
    Wires c1 = fq::mul_montgomery(cc, neg_a1, inverse_norm);
    return Fq2{{c0, c1}}.to_wires();
  });
  return Fq2::from_wires(out);
}
// fq2.rs:374-384: FROBENIUS_COEFF_FP2_C1 = [1, -1]; the constant is handed over in Montgomery form
inline Fq2 frobenius_montgomery(CircuitContext& c, const Fq2& a, size_t i) {
  const BigU coef_mont = (i % 2 == 0) ? ExtConst::r_mod_p() : fq_sub_const(BigU(), ExtConst::r_mod_p());
  Wires c1 = fq::mul_by_constant_montgomery(c, a.c[1], coef_mont);
  return Fq2{{a.c[0], c1}};
}
}  // namespace fq2

// Frobenius coefficients of ark_bn254 in MONTGOMERY form (value * 2^254 mod p), indices 1..3:
//   FP6_C1[i] = xi^((p^i-1)/3), FP6_C2[i] = xi^((2p^i-2)/3), FP12_C1[i] = xi^((p^i-1)/6), xi = 9 + u
// (standard-form values below; checked by tests/test_gadgets_execute.py: frobenius == x^(p^i))
inline BigU fq_as_montgomery_const(const BigU& v) {  // v * 2^254 mod p by 254 modular doublings
  BigU r = v;
  for (int i = 0; i < 254; ++i) r = fq_double_const(r);
  return r;
}
inline Fq2Const frob_const(int which, size_t i) {
  static const char* T[3][4][2] = {
      {{"1", "0"},
       {"2fb347984f7911f74c0bec3cf559b143b78cc310c2c3330c99e39557176f553d", "16c9e55061ebae204ba4cc8bd75a079432ae2a1d0b7c9dce1665d51c640fcba2"},
       {"30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd48", "0"},
       {"0856e078b755ef0abaff1c77959f25ac805ffd3d5d6942d37b746ee87bdcfb6d", "04f1de41b3d1766fa9f30e6dec26094f0fdf31bf98ff2631380cab2baaa586de"}},
      {{"1", "0"},
       {"05b54f5e64eea80180f3c0b75a181e84d33365f7be94ec72848a1f55921ea762", "2c145edbe7fd8aee9f3a80b03b0b1c923685d2ea1bdec763c13b4711cd2b8126"},
       {"59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe", "0"},
       {"0bc58c6611c08dab19bee0f7b5b2444ee633094575b06bcb0e1a92bc3ccbf066", "23d5e999e1910a12feb0f6ef0cd21d04a44a9e08737f96e55fe3ed9d730c239f"}},
      {{"1", "0"},
       {"1284b71c2865a7dfe8b99fdd76e68b605c521e08292f2176d60b35dadcc9e470", "246996f3b4fae7e6a6327cfe12150b8e747992778eeec7e5ca5cf05f80f362ac"},
       {"30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd49", "0"},
       {"19dc81cfcc82e4bbefe9608cd0acaa90894cb38dbe55d24ae86f7d391ed4a67f", "00abf8b60be77d7306cbeee33576139d7f03a5e397d439ec7694aa2bf4c0c101"}}};
  if (i > 3) gsv_panic("Frobenius power above 3 is not tabulated");
  return Fq2Const{fq_as_montgomery_const(BigU::from_hex(T[which][i][0])), fq_as_montgomery_const(BigU::from_hex(T[which][i][1]))};
}

namespace fq6 {
inline Fq6 neg(CircuitContext& c, const Fq6& a) { return {{fq2::neg(c, a.c[0]), fq2::neg(c, a.c[1]), fq2::neg(c, a.c[2])}}; }  // fq6.rs:162-168
inline Fq6 mul_by_constant_fq2_montgomery(CircuitContext& c, const Fq6& a, const Fq2Const& b) {  // fq6.rs:334-344
  return {{fq2::mul_by_constant_montgomery(c, a.c[0], b), fq2::mul_by_constant_montgomery(c, a.c[1], b), fq2::mul_by_constant_montgomery(c, a.c[2], b)}};
}
inline Fq6 square_montgomery(CircuitContext& c, const Fq6& a) {  // fq6.rs:421-448 (eprint 2006/471)
  const Fq2 &a_c0 = a.c[0], &a_c1 = a.c[1], &a_c2 = a.c[2];
  Fq2 s_0 = fq2::square_montgomery(c, a_c0);
  Fq2 wires_1 = fq2::add(c, a_c0, a_c2);
  Fq2 wires_2 = fq2::add(c, wires_1, a_c1);
  Fq2 wires_3 = fq2::sub(c, wires_1, a_c1);
  Fq2 s_1 = fq2::square_montgomery(c, wires_2);
  Fq2 s_2 = fq2::square_montgomery(c, wires_3);
  Fq2 wires_4 = fq2::mul_montgomery(c, a_c1, a_c2);
  Fq2 s_3 = fq2::double_(c, wires_4);
  Fq2 s_4 = fq2::square_montgomery(c, a_c2);
  Fq2 wires_5 = fq2::add(c, s_1, s_2);
  Fq2 t_1 = fq2::half(c, wires_5);
  Fq2 wires_6 = fq2::mul_by_nonresidue(c, s_3);
  Fq2 res_c0 = fq2::add(c, s_0, wires_6);
  Fq2 wires_7 = fq2::mul_by_nonresidue(c, s_4);
  Fq2 wires_8 = fq2::sub(c, s_1, s_3);
  Fq2 wires_9 = fq2::sub(c, wires_8, t_1);
  Fq2 res_c1 = fq2::add(c, wires_9, wires_7);
  Fq2 wires_10 = fq2::sub(c, t_1, s_0);
  Fq2 res_c2 = fq2::sub(c, wires_10, s_4);
  return {{res_c0, res_c1, res_c2}};
}
inline Fq6 inverse_montgomery(CircuitContext& cc, const Fq6& r) {  // fq6.rs:450-487
  const Fq2 &a = r.c[0], &b = r.c[1], &c = r.c[2];
  Fq2 a_square = fq2::square_montgomery(cc, a);
  Fq2 b_square = fq2::square_montgomery(cc, b);
  Fq2 c_square = fq2::square_montgomery(cc, c);
  Fq2 ab = fq2::mul_montgomery(cc, a, b);
  Fq2 ac = fq2::mul_montgomery(cc, a, c);
  Fq2 bc = fq2::mul_montgomery(cc, b, c);
  Fq2 bc_beta = fq2::mul_by_nonresidue(cc, bc);
  Fq2 a_square_minus_bc_beta = fq2::sub(cc, a_square, bc_beta);
  Fq2 c_square_beta = fq2::mul_by_nonresidue(cc, c_square);
  Fq2 c_square_beta_minus_ab = fq2::sub(cc, c_square_beta, ab);
  Fq2 b_square_minus_ac = fq2::sub(cc, b_square, ac);
  Fq2 wires_1 = fq2::mul_montgomery(cc, c_square_beta_minus_ab, c);
  Fq2 wires_2 = fq2::mul_montgomery(cc, b_square_minus_ac, b);
  Fq2 wires_1_plus_wires_2 = fq2::add(cc, wires_1, wires_2);
  Fq2 wires_3 = fq2::mul_by_nonresidue(cc, wires_1_plus_wires_2);
  Fq2 wires_4 = fq2::mul_montgomery(cc, a, a_square_minus_bc_beta);
  Fq2 norm = fq2::add(cc, wires_4, wires_3);
  Fq2 inverse_norm = fq2::inverse_montgomery(cc, norm);
  Fq2 res_c0 = fq2::mul_montgomery(cc, a_square_minus_bc_beta, inverse_norm);
  Fq2 res_c1 = fq2::mul_montgomery(cc, c_square_beta_minus_ab, inverse_norm);
  Fq2 res_c2 = fq2::mul_montgomery(cc, b_square_minus_ac, inverse_norm);
  return {{res_c0, res_c1, res_c2}};
}
inline Fq6 frobenius_montgomery(CircuitContext& c, const Fq6& a, size_t i) {  // fq6.rs:489-515
  Fq2 f0 = fq2::frobenius_montgomery(c, a.c[0], i);
  Fq2 f1 = fq2::frobenius_montgomery(c, a.c[1], i);
  Fq2 f2 = fq2::frobenius_montgomery(c, a.c[2], i);
  Fq2 f1u = fq2::mul_by_constant_montgomery(c, f1, frob_const(0, i % 6));
  Fq2 f2u = fq2::mul_by_constant_montgomery(c, f2, frob_const(1, i % 6));
  return {{f0, f1u, f2u}};
}
}  // namespace fq6

namespace fq12 {
inline Fq12 conjugate(CircuitContext& c, const Fq12& a) { return Fq12{{a.c[0], fq6::neg(c, a.c[1])}}; }  // fq12.rs:444-447
inline Fq12 inverse_montgomery(CircuitContext& c, const Fq12& a) {  // fq12.rs:413-428 (#[component])
  Wires out = component(c, KeyBuilder("fq12::inverse_montgomery"), a.to_wires(), N, [](CircuitContext& cc, const Wires& in) {
    Fq12 a = Fq12::from_wires(in);
    Fq6 a_c0_square = fq6::square_montgomery(cc, a.c[0]);
    Fq6 a_c1_square = fq6::square_montgomery(cc, a.c[1]);
    Fq6 a_c1_square_beta = fq6::mul_by_nonresidue(cc, a_c1_square);
    Fq6 norm = fq6::sub(cc, a_c0_square, a_c1_square_beta);
    Fq6 inverse_norm = fq6::inverse_montgomery(cc, norm);
    Fq6 res_c0 = fq6::mul_montgomery(cc, a.c[0], inverse_norm);
    Fq6 neg_a_c1 = fq6::neg(cc, a.c[1]);
    Fq6 res_c1 = fq6::mul_montgomery(cc, inverse_norm, neg_a_c1);
    return Fq12{{res_c0, res_c1}}.to_wires();
  });
  return Fq12::from_wires(out);
}
inline Fq12 frobenius_montgomery(CircuitContext& c, const Fq12& a, size_t i) {  // fq12.rs:430-442
  Fq6 f0 = fq6::frobenius_montgomery(c, a.c[0], i);
  Fq6 f1 = fq6::frobenius_montgomery(c, a.c[1], i);
  Fq6 x = fq6::mul_by_constant_fq2_montgomery(c, f1, frob_const(2, i % 12));
  return Fq12{{f0, x}};
}
inline Fq12 one_constant() {  // Fq12::new_constant(ONE): Montgomery form of 1 in c0.c0.c0, zero elsewhere (fq12.rs:74-112)
  Wires w = constant_wires(ExtConst::r_mod_p(), 254);
  Wires z = constant_wires(BigU(), 254);
  for (int k = 0; k < 11; ++k) w = concat(w, z);
  return Fq12::from_wires(w);
}

// final_exponentiation.rs:65-99: NAF of the BN parameter x = 4965661367192848881, most significant digit first
inline Fq12 cyclotomic_exp_fast_inverse_montgomery_fast(CircuitContext& c, const Fq12& f) {
  static const int8_t NAF[63] = {1, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 1, 0, -1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0,
                                 0, 0, 1, 0, -1, 0, -1, 0, -1, 0, 1, 0, 1, 0, 0, -1, 0, 1, 0, 1, 0, -1, 0, 0, 1, 0, 1, 0, 0, 0, 1};
  Fq12 res = one_constant();
  Fq12 f_inverse = inverse_montgomery(c, f);
  bool found_nonzero = false;
  for (int idx = 62; idx >= 0; --idx) {
    const int8_t value = NAF[idx];
    if (found_nonzero) res = cyclotomic_square_montgomery(c, res);
    if (value != 0) {
      found_nonzero = true;
      res = value > 0 ? mul_montgomery(c, res, f) : mul_montgomery(c, res, f_inverse);
    }
  }
  return res;
}
inline Fq12 exp_by_neg_x_montgomery(CircuitContext& c, const Fq12& f) {  // final_exponentiation.rs:94-97
  Fq12 f2 = cyclotomic_exp_fast_inverse_montgomery_fast(c, f);
  return conjugate(c, f2);
}
inline Fq12 final_exponentiation_montgomery(CircuitContext& c, const Fq12& f) {  // final_exponentiation.rs:99-135 (#[component])
  Wires out = component(c, KeyBuilder("final_exponentiation_montgomery"), f.to_wires(), N, [](CircuitContext& cc, const Wires& in) {
    Fq12 f = Fq12::from_wires(in);
    Fq12 f_inv = inverse_montgomery(cc, f);
    Fq12 f_conjugate = conjugate(cc, f);
    Fq12 u = mul_montgomery(cc, f_inv, f_conjugate);
    Fq12 u_frobenius = frobenius_montgomery(cc, u, 2);
    Fq12 r = mul_montgomery(cc, u_frobenius, u);
    Fq12 y0 = exp_by_neg_x_montgomery(cc, r);
    Fq12 y1 = square_montgomery(cc, y0);
    Fq12 y2 = square_montgomery(cc, y1);
    Fq12 y3 = mul_montgomery(cc, y1, y2);
    Fq12 y4 = exp_by_neg_x_montgomery(cc, y3);
    Fq12 y5 = square_montgomery(cc, y4);
    Fq12 y6 = exp_by_neg_x_montgomery(cc, y5);
    Fq12 y7 = conjugate(cc, y3);
    Fq12 y8 = conjugate(cc, y6);
    Fq12 y9 = mul_montgomery(cc, y8, y4);
    Fq12 y10 = mul_montgomery(cc, y9, y7);
    Fq12 y11 = mul_montgomery(cc, y10, y1);
    Fq12 y12 = mul_montgomery(cc, y10, y4);
    Fq12 y13 = mul_montgomery(cc, y12, r);
    Fq12 y14 = frobenius_montgomery(cc, y11, 1);
    Fq12 y15 = mul_montgomery(cc, y14, y13);
    Fq12 y16 = frobenius_montgomery(cc, y10, 2);
    Fq12 y17 = mul_montgomery(cc, y16, y15);
    Fq12 r2 = conjugate(cc, r);
    Fq12 y18 = mul_montgomery(cc, r2, y11);
    Fq12 y19 = frobenius_montgomery(cc, y18, 3);
    return mul_montgomery(cc, y19, y17).to_wires();
  });
  return Fq12::from_wires(out);
}
}  // namespace fq12

}  // namespace gadgets
}  // namespace gsv
