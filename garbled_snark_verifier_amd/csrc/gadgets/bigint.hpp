// Gate-stream producers: basic + bigint gadgets, generic over CircuitContext.
// Host-side mirror of the reference's gadget layer for the hot path's callers
// (ref: src/gadgets/basic.rs:7-72, src/gadgets/bigint/{add,mul,cmp}.rs).  The order of
// add_gate / issue_wire calls, the operand order of every gate and the component boundaries
// (which functions are `#[component]`s and what their input lists are) follow the reference
// exactly: gate order fixes gate_id, operand order fixes the half-gate ciphertext, component
// boundaries fix which wires have zero fan-out (dead gates).
#pragma once
#include <utility>

#include "../circuit/bigu.hpp"
#include "../circuit/circuit.hpp"

namespace gsv {
namespace gadgets {

using BigIntWires = Wires;  // bigint/mod.rs:50-53: LSB-first wire vector

inline Wires concat(const Wires& a, const Wires& b) {
  Wires r; r.reserve(a.size() + b.size());
  r.insert(r.end(), a.begin(), a.end()); r.insert(r.end(), b.begin(), b.end());
  return r;
}
inline Wires slice(const Wires& a, size_t lo, size_t hi) { return Wires(a.begin() + lo, a.begin() + hi); }

// The `#[component]` / `#[bn_component]` wrapper (circuit_component_macro/src/gen_wrapper.rs:270-286):
// key = name + off-circuit params + arity + input length; body runs on the wires it is handed.
template <class Body>
inline Wires component(CircuitContext& c, KeyBuilder kb, const Wires& inputs, size_t arity, Body&& body) {
  ChildFn fn = [&body](CircuitContext& cc, const Wires& in) -> Wires { return body(cc, in); };
  return c.with_named_child(kb.finish(arity, inputs.size()), inputs, fn, arity);
}

// ------------------------------------------------------------------ basic.rs
struct SumCarry { WireId sum, carry; };

inline SumCarry half_adder(CircuitContext& c, WireId a, WireId b) {  // basic.rs:7-15
  WireId result = c.issue_wire(), carry = c.issue_wire();
  c.add_gate(Gate::xor_(a, b, result));
  c.add_gate(Gate::and_(a, b, carry));
  return {result, carry};
}
inline SumCarry full_adder(CircuitContext& c, WireId a, WireId b, WireId cin) {  // basic.rs:17-32
  WireId axc = c.issue_wire(), bxc = c.issue_wire(), result = c.issue_wire(), t = c.issue_wire(), carry = c.issue_wire();
  c.add_gate(Gate::xor_(a, cin, axc));
  c.add_gate(Gate::xor_(b, cin, bxc));
  c.add_gate(Gate::xor_(a, bxc, result));
  c.add_gate(Gate::and_(axc, bxc, t));
  c.add_gate(Gate::xor_(cin, t, carry));
  return {result, carry};
}
inline SumCarry half_subtracter(CircuitContext& c, WireId a, WireId b) {  // basic.rs:34-46
  WireId result = c.issue_wire(), borrow = c.issue_wire();
  c.add_gate(Gate::xor_(a, b, result));
  c.add_gate(Gate::and_variant(a, b, borrow, true, false, false));
  return {result, borrow};
}
inline SumCarry full_subtracter(CircuitContext& c, WireId a, WireId b, WireId cin) {  // basic.rs:48-63
  WireId bxa = c.issue_wire(), bxc = c.issue_wire(), result = c.issue_wire(), t = c.issue_wire(), carry = c.issue_wire();
  c.add_gate(Gate::xor_(a, b, bxa));
  c.add_gate(Gate::xor_(b, cin, bxc));
  c.add_gate(Gate::xor_(bxa, cin, result));
  c.add_gate(Gate::and_(bxa, bxc, t));
  c.add_gate(Gate::xor_(cin, t, carry));
  return {result, carry};
}
inline WireId selector(CircuitContext& c, WireId a, WireId b, WireId s) {  // basic.rs:65-72: s ? a : b
  WireId d = c.issue_wire(), f = c.issue_wire(), g = c.issue_wire();
  c.add_gate(Gate::nand(a, s, d));
  c.add_gate(Gate::and_variant(s, b, f, true, false, true));
  c.add_gate(Gate::nand(d, f, g));
  return g;
}

// ------------------------------------------------------------------ bigint/add.rs
inline BigIntWires add(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // add.rs:8-26
  if (a.size() != b.size()) gsv_panic("bigint::add: length mismatch");
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::add"), concat(a, b), n + 1, [n](CircuitContext& cc, const Wires& in) {
    Wires bits; bits.reserve(n + 1);
    SumCarry r = half_adder(cc, in[0], in[n]);
    bits.push_back(r.sum);
    WireId carry = r.carry;
    for (size_t i = 1; i < n; ++i) {
      SumCarry f = full_adder(cc, in[i], in[n + i], carry);
      bits.push_back(f.sum);
      carry = f.carry;
    }
    bits.push_back(carry);
    return bits;
  });
}
inline BigIntWires add_without_carry(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // add.rs:28-36
  BigIntWires r = add(c, a, b);
  r.pop_back();
  return r;
}
inline BigIntWires add_constant(CircuitContext& c, const BigIntWires& a, const BigU& b) {  // add.rs:38-84
  const size_t n = a.size();
  std::string kb = b.key_bytes();
  return component(c, KeyBuilder("bigint::add_constant").param("b", kb.data(), kb.size()), a, n + 1,
                   [n, &b](CircuitContext& cc, const Wires& in) {
    if (b.is_zero()) gsv_panic("add_constant: b must be non-zero");
    std::vector<bool> bb = b.bits_with_len(n);
    size_t first_one = 0;
    while (!bb[first_one]) ++first_one;
    Wires bits; bits.reserve(n + 1);
    WireId carry = UNREACHABLE;
    for (size_t i = 0; i < n; ++i) {
      WireId a_i = in[i];
      if (i < first_one) {
        bits.push_back(a_i);
      } else if (i == first_one) {
        WireId w = cc.issue_wire();
        cc.add_gate(Gate::not_with_xor(a_i, w));
        bits.push_back(w);
        carry = a_i;
      } else if (bb[i]) {
        WireId w1 = cc.issue_wire(), w2 = cc.issue_wire();
        cc.add_gate(Gate::xnor(a_i, carry, w1));
        cc.add_gate(Gate::or_(a_i, carry, w2));
        bits.push_back(w1);
        carry = w2;
      } else {
        WireId w1 = cc.issue_wire(), w2 = cc.issue_wire();
        cc.add_gate(Gate::xor_(a_i, carry, w1));
        cc.add_gate(Gate::and_(a_i, carry, w2));
        bits.push_back(w1);
        carry = w2;
      }
    }
    bits.push_back(carry);
    return bits;
  });
}
inline BigIntWires add_constant_without_carry(CircuitContext& c, const BigIntWires& a, const BigU& b) {  // add.rs:86-94
  BigIntWires r = add_constant(c, a, b);
  r.pop_back();
  return r;
}
inline BigIntWires sub(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // add.rs:96-115
  if (a.size() != b.size()) gsv_panic("bigint::sub: length mismatch");
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::sub"), concat(a, b), n + 1, [n](CircuitContext& cc, const Wires& in) {
    Wires bits; bits.reserve(n + 1);
    SumCarry r = half_subtracter(cc, in[0], in[n]);
    bits.push_back(r.sum);
    WireId borrow = r.carry;
    for (size_t i = 1; i < n; ++i) {
      SumCarry f = full_subtracter(cc, in[i], in[n + i], borrow);
      borrow = f.carry;
      bits.push_back(f.sum);
    }
    bits.push_back(borrow);
    return bits;
  });
}
inline BigIntWires sub_without_borrow(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // add.rs:117-125
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::sub_without_borrow"), concat(a, b), n, [n](CircuitContext& cc, const Wires& in) {
    Wires bits = sub(cc, slice(in, 0, n), slice(in, n, 2 * n));
    bits.pop_back();
    return bits;
  });
}
inline BigIntWires half(const BigIntWires& a) {  // add.rs:146-156 (no gates)
  BigIntWires r(a.begin() + 1, a.end());
  r.push_back(FALSE_WIRE);
  return r;
}

// ------------------------------------------------------------------ bigint/cmp.rs
inline BigIntWires self_or_zero(CircuitContext& c, const BigIntWires& a, WireId s) {  // cmp.rs:10-22
  const size_t n = a.size();
  Wires in = a; in.push_back(s);
  return component(c, KeyBuilder("bigint::self_or_zero"), in, n, [n](CircuitContext& cc, const Wires& in) {
    Wires bits; bits.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      WireId w = cc.issue_wire();
      cc.add_gate(Gate::and_(in[i], in[n], w));
      bits.push_back(w);
    }
    return bits;
  });
}
inline WireId equal_zero(CircuitContext& c, const BigIntWires& a) {  // cmp.rs:88-108
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::equal_zero"), a, 1, [n](CircuitContext& cc, const Wires& in) -> Wires {
    if (n == 1) {
      WireId z = cc.issue_wire();
      cc.add_gate(Gate::not_with_xor(in[0], z));
      return {z};
    }
    WireId res = cc.issue_wire();
    cc.add_gate(Gate::xnor(in[0], in[1], res));
    for (size_t i = 1; i < n; ++i) {
      WireId next = cc.issue_wire();
      cc.add_gate(Gate::and_variant(in[i], res, next, true, false, false));
      res = next;
    }
    return {res};
  })[0];
}
inline WireId equal_constant(CircuitContext& c, const BigIntWires& a, const BigU& b) {  // cmp.rs:61-86
  const size_t n = a.size();
  std::string kb = b.key_bytes();
  return component(c, KeyBuilder("bigint::equal_constant").param("b", kb.data(), kb.size()), a, 1,
                   [n, &b](CircuitContext& cc, const Wires& in) -> Wires {
    if (b.is_zero()) return {equal_zero(cc, in)};
    std::vector<bool> bb = b.bits_with_len(n);
    size_t one_ind = 0;
    while (!bb[one_ind]) ++one_ind;
    WireId res = in[one_ind];
    for (size_t i = 0; i < n; ++i) {
      if (i == one_ind) continue;
      WireId nr = cc.issue_wire();
      cc.add_gate(Gate::and_variant(in[i], res, nr, !bb[i], false, false));
      res = nr;
    }
    return {res};
  })[0];
}
inline WireId equal(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // cmp.rs:43-59
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::equal"), concat(a, b), 1, [n](CircuitContext& cc, const Wires& in) -> Wires {
    Wires x; x.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      WireId w = cc.issue_wire();
      cc.add_gate(Gate::xor_(in[i], in[n + i], w));
      x.push_back(w);
    }
    return {equal_constant(cc, x, BigU())};
  })[0];
}
inline WireId greater_than(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // cmp.rs:110-131
  const size_t n = a.size();
  return component(c, KeyBuilder("bigint::greater_than"), concat(a, b), 1, [n](CircuitContext& cc, const Wires& in) -> Wires {
    Wires not_b; not_b.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      WireId w = cc.issue_wire();
      cc.add_gate(Gate::not_with_xor(in[n + i], w));
      not_b.push_back(w);
    }
    BigIntWires sum = add(cc, slice(in, 0, n), not_b);
    return {sum.back()};
  })[0];
}
inline WireId less_than_constant(CircuitContext& c, const BigIntWires& a, const BigU& b) {  // cmp.rs:133-154
  const size_t n = a.size();
  std::string kb = b.key_bytes();
  return component(c, KeyBuilder("bigint::less_than_constant").param("b", kb.data(), kb.size()), a, 1,
                   [n, &b](CircuitContext& cc, const Wires& in) -> Wires {
    Wires not_a; not_a.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      WireId w = cc.issue_wire();
      cc.add_gate(Gate::not_with_xor(in[i], w));
      not_a.push_back(w);
    }
    BigIntWires sum = add_constant(cc, not_a, b);
    return {sum.back()};
  })[0];
}
inline BigIntWires select(CircuitContext& c, const BigIntWires& a, const BigIntWires& b, WireId s) {  // cmp.rs:156-173
  if (a.size() != b.size()) gsv_panic("bigint::select: length mismatch");
  const size_t n = a.size();
  Wires in = concat(a, b); in.push_back(s);
  return component(c, KeyBuilder("bigint::select"), in, n, [n](CircuitContext& cc, const Wires& in) {
    Wires bits; bits.reserve(n);
    for (size_t i = 0; i < n; ++i) bits.push_back(selector(cc, in[i], in[n + i], in[2 * n]));
    return bits;
  });
}

// ------------------------------------------------------------------ bigint/mul.rs
constexpr bool is_use_karatsuba(size_t len) { return len == 21 ? false : len > 19; }  // mul.rs:8-13

inline BigIntWires mul_naive(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // mul.rs:19-56
  if (a.size() != b.size()) gsv_panic("mul_naive: length mismatch");
  const size_t len = a.size();
  return component(c, KeyBuilder("bigint::mul_naive"), concat(a, b), len * 2, [len](CircuitContext& cc, const Wires& in) {
    Wires result(len * 2, FALSE_WIRE);
    for (size_t i = 0; i < len; ++i) {
      WireId current_bit = in[len + i];
      Wires add0 = slice(result, i, i + len);
      Wires add1; add1.reserve(len);
      for (size_t j = 0; j < len; ++j) {
        WireId w = cc.issue_wire();
        cc.add_gate(Gate::and_(in[j], current_bit, w));
        add1.push_back(w);
      }
      BigIntWires r = add(cc, add0, add1);
      for (size_t k = 0; k < len + 1; ++k) result[i + k] = r[k];
    }
    return result;
  });
}

inline BigIntWires mul_karatsuba(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // mul.rs:58-183
  if (a.size() != b.size()) gsv_panic("mul_karatsuba: length mismatch");
  const size_t len = a.size();
  return component(c, KeyBuilder("bigint::mul_karatsuba"), concat(a, b), len * 2, [len](CircuitContext& cc, const Wires& in) {
    Wires a = slice(in, 0, len), b = slice(in, len, 2 * len);
    if (len < 5) return mul_naive(cc, a, b);
    Wires result(len * 2, FALSE_WIRE);
    const size_t len0 = len / 2, len1 = (len + 1) / 2;
    Wires a0 = slice(a, 0, len0), a1 = slice(a, len0, len);
    Wires b0 = slice(b, 0, len0), b1 = slice(b, len0, len);
    Wires sq0 = is_use_karatsuba(len0) ? mul_karatsuba(cc, a0, b0) : mul_naive(cc, a0, b0);
    Wires sq1 = is_use_karatsuba(len1) ? mul_karatsuba(cc, a1, b1) : mul_naive(cc, a1, b1);
    Wires ea0 = a0, eb0 = b0, esq0 = sq0;
    if (len0 < len1) {
      ea0.push_back(FALSE_WIRE);
      eb0.push_back(FALSE_WIRE);
      esq0.push_back(FALSE_WIRE);
      esq0.push_back(FALSE_WIRE);
    }
    Wires sum_a = add(cc, ea0, a1);
    Wires sum_b = add(cc, eb0, b1);
    Wires sq_sum = add(cc, esq0, sq1);
    sq_sum.push_back(FALSE_WIRE);
    Wires sum_mul = is_use_karatsuba(sum_a.size()) ? mul_karatsuba(cc, sum_a, sum_b) : mul_naive(cc, sum_a, sum_b);
    Wires cross_full = sub_without_borrow(cc, sum_mul, sq_sum);
    Wires cross = slice(cross_full, 0, len + 1);
    for (size_t k = 0; k < len0 * 2; ++k) result[k] = sq0[k];
    Wires segment = slice(result, len0, len0 + len + 1);
    Wires new_segment = add(cc, segment, cross);
    for (size_t k = 0; k < len + 2; ++k) result[len0 + k] = new_segment[k];
    Wires segment2 = slice(result, 2 * len0, result.size());
    Wires new_segment2 = add(cc, segment2, sq1);
    for (size_t k = 0; k < 2 * len1; ++k) result[2 * len0 + k] = new_segment2[k];
    return result;
  });
}

inline BigIntWires mul(CircuitContext& c, const BigIntWires& a, const BigIntWires& b) {  // mul.rs:185-207
  const size_t len = a.size();
  if (len < 5) return mul_naive(c, a, b);
  if (len > 4000) gsv_panic("Bit length exceeds maximum supported 4000");
  return is_use_karatsuba(len) ? mul_karatsuba(c, a, b) : mul_naive(c, a, b);
}

inline BigIntWires mul_by_constant(CircuitContext& c, const BigIntWires& a, const BigU& k) {  // mul.rs:209-241
  const size_t len = a.size();
  std::string kb = k.key_bytes();
  return component(c, KeyBuilder("bigint::mul_by_constant").param("c", kb.data(), kb.size()), a, len * 2,
                   [len, &k](CircuitContext& cc, const Wires& a) {
    Wires acc(len * 2, FALSE_WIRE);
    std::vector<bool> kbits = k.bits_with_len(len);
    for (size_t i = 0; i < len; ++i) {
      if (!kbits[i]) continue;
      Wires addw = slice(acc, i, i + len);
      BigIntWires nb = add(cc, a, addw);
      for (size_t t = 0; t < len + 1; ++t) acc[i + t] = nb[t];
    }
    return acc;
  });
}

inline BigIntWires mul_by_constant_modulo_power_two(CircuitContext& c, const BigIntWires& a, const BigU& k, size_t power) {
  // mul.rs:243-329
  const size_t len = a.size();
  std::string kb = k.key_bytes();
  return component(c, KeyBuilder("bigint::mul_by_constant_modulo_power_two").param("c", kb.data(), kb.size()).param_usize("power", power),
                   a, power, [len, power, &k](CircuitContext& cc, const Wires& a) {
    constexpr size_t PER_CHUNK = 8;
    if (!(power < 2 * len)) gsv_panic("power must be < 2*len");
    std::vector<bool> kbits = k.bits_with_len(len);
    std::vector<size_t> ones;
    for (size_t i = 0; i < len; ++i) if (i < power && kbits[i]) ones.push_back(i);
    Wires result(power, FALSE_WIRE);
    if (ones.empty()) return result;
    for (size_t chunk_idx = 0; chunk_idx * PER_CHUNK < ones.size(); ++chunk_idx) {
      std::vector<size_t> chunk(ones.begin() + chunk_idx * PER_CHUNK,
                                ones.begin() + std::min(ones.size(), (chunk_idx + 1) * PER_CHUNK));
      Wires prev = result;
      Wires inputs = concat(a, prev);
      ComponentKey key = KeyBuilder("mul_by_const_mod_2p").param_usize("a_len", len).param_usize("power", power)
                             .param_usize("chunk_idx", chunk_idx).finish(power, inputs.size());
      ChildFn fn = [len, power, &chunk](CircuitContext& ctx, const Wires& in) -> Wires {
        Wires a(in.begin(), in.begin() + len);
        Wires res(in.begin() + len, in.end());
        for (size_t i : chunk) {
          size_t nb = std::min(power - i, len);
          if (nb == 0) continue;
          Wires a_slice = slice(a, 0, nb);
          Wires addw = slice(res, i, i + nb);
          BigIntWires nbts = add(ctx, a_slice, addw);
          if (i + nb < power) for (size_t t = 0; t < nb + 1; ++t) res[i + t] = nbts[t];
          else for (size_t t = 0; t < nb; ++t) res[i + t] = nbts[t];
        }
        return res;
      };
      result = cc.with_named_child(key, inputs, fn, power);
    }
    return result;
  });
}

}  // namespace gadgets
}  // namespace gsv
