// Per-gate label arithmetic of the MI355X engine: fixed-key AES-128 (T-table form for LDS),
// gate-id tweak, privacy-free half-gate garble / degarble.
//
// Written for the gfx950 kernels in garble_kernels.hip; the functions are also compilable as plain
// host C++ (GSV_HD expands to nothing) so that tests/hostsim can unit-test exactly this code on a
// machine without a GPU.  The product's C-ABI never takes the host route.
//
// Data convention: a label is the 16 bytes of the reference's `S::to_bytes()` (big-endian u128,
// src/core/s.rs:25-31) = the AES block, held as four little-endian 32-bit words w[0..3] of those
// bytes, i.e. w[i] is AES state column i with row 0 in the low byte.  XOR is layout-agnostic.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GSV_HD __host__ __device__ __forceinline__
#else
#define GSV_HD inline
#endif

namespace gsv {
namespace dev {

struct Label {
  uint32_t w[4];
};
GSV_HD Label lxor(const Label& a, const Label& b) { return Label{{a.w[0] ^ b.w[0], a.w[1] ^ b.w[1], a.w[2] ^ b.w[2], a.w[3] ^ b.w[3]}}; }
// XOR with delta iff `on` (branch-free: mask is 0 or ~0).
GSV_HD Label lxor_if(const Label& a, const Label& d, uint32_t on) {
  uint32_t m = 0u - (on & 1u);
  return Label{{a.w[0] ^ (d.w[0] & m), a.w[1] ^ (d.w[1] & m), a.w[2] ^ (d.w[2] & m), a.w[3] ^ (d.w[3] & m)}};
}

// Gate discriminants = reference `GateType` repr(C) values (src/core/gate_type.rs:3-15).
enum : uint32_t { GT_AND = 0, GT_XOR = 8, GT_XNOR = 9, GT_NOT = 10 };
// alphas_const (gate_type.rs:20-37): for discriminant t<8, (alpha_a, alpha_b, alpha_c) = bits (4,2,1) of t.
GSV_HD uint32_t alpha_a(uint32_t t) { return (t >> 2) & 1u; }
GSV_HD uint32_t alpha_b(uint32_t t) { return (t >> 1) & 1u; }
GSV_HD uint32_t alpha_c(uint32_t t) { return t & 1u; }
// Plain truth function ((a^aa)&(b^ab))^ac for t<8; Xor/Xnor/Not otherwise (gate_type.rs:39-61).
GSV_HD uint32_t gate_eval_bit(uint32_t t, uint32_t a, uint32_t b) {
  if (t < 8) return (((a ^ alpha_a(t)) & (b ^ alpha_b(t))) ^ alpha_c(t)) & 1u;
  if (t == GT_XOR) return (a ^ b) & 1u;
  if (t == GT_XNOR) return (a ^ b ^ 1u) & 1u;
  return (a ^ 1u) & 1u;
}

// Tweak (src/hashers/mod.rs:57-64,90-96): bytes[0..8) = LE64(g ^ C0), bytes[8..16) = LE64(g * C1).
GSV_HD Label tweak_of(uint64_t gate_id) {
  uint64_t t0 = gate_id ^ 0x123456789ABCDEF0ull;
  uint64_t t1 = gate_id * 0xDEADBEEFCAFEBABEull;
  return Label{{uint32_t(t0), uint32_t(t0 >> 32), uint32_t(t1), uint32_t(t1 >> 32)}};
}

// AES tables as the kernels see them.  A table accessor `Tab` provides
//     template <int K, int BYTE> uint32_t lk(uint32_t s) const   = Te_K[ byte BYTE of s ]
//     uint32_t rk(int i) const                                   = round-key word i (0..43) of the fixed key
//                                                                  0x42*16 (src/hashers/aes_ni.rs:165)
//   Te0[x] = (2s, s, s, 3s)   Te1[x] = (3s, 2s, s, s)   Te2[x] = (s, 3s, 2s, s)   Te3[x] = (s, s, 3s, 2s)
// as little-endian words (byte k = state row k), s = SBOX[x].
// Two accessors exist: PlainTables (host / tests: four 256-entry arrays) and, in kernels.hip,
// LdsBankedTables (each entry replicated once per LDS bank so that a wave's 64 random lookups
// never conflict).
// three-input XOR: one V_BITOP3_B32 (truth table 0x96) on gfx950
GSV_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

struct PlainTables {
  static constexpr bool kPairedRotation = false;  // four tables, plain round keys
  const uint32_t* te[4];
  const uint32_t* rkp;  // 44 round-key words
  template <int K, int BYTE>
  GSV_HD uint32_t lk(uint32_t s) const { return te[K][(s >> (8 * BYTE)) & 0xff]; }
  GSV_HD uint32_t rk(int i) const { return rkp[i]; }
};

// One middle round for one column: T0[b0 of x0] ^ T1[b1 of x1] ^ T2[b2 of x2] ^ T3[b3 of x3] ^ k
// A table set that keeps only Te0 and Te2 (kPairedRotation) pays ONE rotation per column instead of one per rotated term:
//   Te1[b] ^ Te3[d] ^ k = rotl8(Te0[b] ^ Te2[d] ^ rotr8(k)),
// and its rk(4..39) — the middle rounds' keys — are stored as rotr8(k), so the key rides in the same three-input XOR.
template <class Tab>
GSV_HD uint32_t aes_col(const Tab& T, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t k) {
  if constexpr (Tab::kPairedRotation) {
    const uint32_t odd = xor3(T.template lk<0, 1>(x1), T.template lk<2, 3>(x3), k);
    return xor3(T.template lk<0, 0>(x0), T.template lk<2, 2>(x2), (odd << 8) | (odd >> 24));
  } else {
    return xor3(xor3(T.template lk<0, 0>(x0), T.template lk<1, 1>(x1), T.template lk<2, 2>(x2)), T.template lk<3, 3>(x3), k);
  }
}
// Final round column: SubBytes + ShiftRows + AddRoundKey; the plain S-box byte sits in Te2 byte0 and byte3
// and in Te0 byte1 and byte2 (only the two tables the device keeps un-rotated are used).
template <class Tab>
GSV_HD uint32_t aes_last_col(const Tab& T, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t k) {
  const uint32_t m0 = T.template lk<2, 0>(x0), m1 = T.template lk<0, 1>(x1), m2 = T.template lk<0, 2>(x2), m3 = T.template lk<2, 3>(x3);
#if defined(__HIP_DEVICE_COMPILE__)
  // byte assembly with two v_perm_b32 (bytes 0..3 of the second operand are selector values 0..3, of the first 4..7, 0x0c = zero)
  // instead of four v_and: {m0.b0, m1.b1, 0, 0} ^ {0, 0, m2.b2, m3.b3} ^ k
  return xor3(__builtin_amdgcn_perm(m1, m0, 0x0c0c0500u), __builtin_amdgcn_perm(m3, m2, 0x07020c0cu), k);
#else
  return xor3(xor3(m0 & 0x000000ffu, m1 & 0x0000ff00u, m2 & 0x00ff0000u), m3 & 0xff000000u, k);
#endif
}

// One full AES-128 encryption of `in` (FIPS-197; equals _mm_aesenc x9 + _mm_aesenclast, aes_ni.rs:39-54).
template <class Tab>
GSV_HD Label aes128_encrypt(const Tab& T, const Label& in) {
  uint32_t s0 = in.w[0] ^ T.rk(0), s1 = in.w[1] ^ T.rk(1), s2 = in.w[2] ^ T.rk(2), s3 = in.w[3] ^ T.rk(3);
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    uint32_t t0 = aes_col(T, s0, s1, s2, s3, T.rk(4 * r + 0));
    uint32_t t1 = aes_col(T, s1, s2, s3, s0, T.rk(4 * r + 1));
    uint32_t t2 = aes_col(T, s2, s3, s0, s1, T.rk(4 * r + 2));
    uint32_t t3 = aes_col(T, s3, s0, s1, s2, T.rk(4 * r + 3));
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  Label o;
  o.w[0] = aes_last_col(T, s0, s1, s2, s3, T.rk(40));
  o.w[1] = aes_last_col(T, s1, s2, s3, s0, T.rk(41));
  o.w[2] = aes_last_col(T, s2, s3, s0, s1, T.rk(42));
  o.w[3] = aes_last_col(T, s3, s0, s1, s2, T.rk(43));
  return o;
}

// Two independent blocks interleaved (the reference's encrypt2_blocks, aes_ni.rs:68-94): two
// dependency chains to overlap LDS latency with.
template <class Tab>
GSV_HD void aes128_encrypt2(const Tab& T, const Label& in0, const Label& in1, Label& out0, Label& out1) {
  uint32_t a0 = in0.w[0] ^ T.rk(0), a1 = in0.w[1] ^ T.rk(1), a2 = in0.w[2] ^ T.rk(2), a3 = in0.w[3] ^ T.rk(3);
  uint32_t b0 = in1.w[0] ^ T.rk(0), b1 = in1.w[1] ^ T.rk(1), b2 = in1.w[2] ^ T.rk(2), b3 = in1.w[3] ^ T.rk(3);
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    const uint32_t k0 = T.rk(4 * r), k1 = T.rk(4 * r + 1), k2 = T.rk(4 * r + 2), k3 = T.rk(4 * r + 3);
    uint32_t t0 = aes_col(T, a0, a1, a2, a3, k0), u0 = aes_col(T, b0, b1, b2, b3, k0);
    uint32_t t1 = aes_col(T, a1, a2, a3, a0, k1), u1 = aes_col(T, b1, b2, b3, b0, k1);
    uint32_t t2 = aes_col(T, a2, a3, a0, a1, k2), u2 = aes_col(T, b2, b3, b0, b1, k2);
    uint32_t t3 = aes_col(T, a3, a0, a1, a2, k3), u3 = aes_col(T, b3, b0, b1, b2, k3);
    a0 = t0; a1 = t1; a2 = t2; a3 = t3;
    b0 = u0; b1 = u1; b2 = u2; b3 = u3;
  }
  out0.w[0] = aes_last_col(T, a0, a1, a2, a3, T.rk(40)); out1.w[0] = aes_last_col(T, b0, b1, b2, b3, T.rk(40));
  out0.w[1] = aes_last_col(T, a1, a2, a3, a0, T.rk(41)); out1.w[1] = aes_last_col(T, b1, b2, b3, b0, T.rk(41));
  out0.w[2] = aes_last_col(T, a2, a3, a0, a1, T.rk(42)); out1.w[2] = aes_last_col(T, b2, b3, b0, b1, T.rk(42));
  out0.w[3] = aes_last_col(T, a3, a0, a1, a2, T.rk(43)); out1.w[3] = aes_last_col(T, b3, b0, b1, b2, T.rk(43));
}

// H(x, g) = AES_K(x ^ tweak(g)), no feed-forward (src/hashers/mod.rs:66-86).
template <class Tab>
GSV_HD Label hash_with_gate(const Tab& T, const Label& x, uint64_t gate_id) { return aes128_encrypt(T, lxor(x, tweak_of(gate_id))); }

// ---- Blake3Hasher (src/hashers/mod.rs:22-51): first 16 bytes of BLAKE3(label_bytes || LE64(gate_id)).
// The 24-byte message is one block of one chunk: compress(IV, m, counter 0, block_len 24,
// CHUNK_START | CHUNK_END | ROOT).  Message words 0..3 are the label's little-endian words, 4..5 the gate id.
GSV_HD uint32_t b3_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define GSV_B3_G(a, b, c, d, mx, my)                                             \
  a = a + b + (mx); d = b3_rotr(d ^ a, 16); c = c + d; b = b3_rotr(b ^ c, 12);   \
  a = a + b + (my); d = b3_rotr(d ^ a, 8);  c = c + d; b = b3_rotr(b ^ c, 7);
GSV_HD Label blake3_hash_with_gate(const Label& x, uint64_t gate_id) {
  uint32_t m[16] = {x.w[0], x.w[1], x.w[2], x.w[3], uint32_t(gate_id), uint32_t(gate_id >> 32), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t s0 = 0x6A09E667u, s1 = 0xBB67AE85u, s2 = 0x3C6EF372u, s3 = 0xA54FF53Au, s4 = 0x510E527Fu, s5 = 0x9B05688Cu, s6 = 0x1F83D9ABu,
           s7 = 0x5BE0CD19u, s8 = 0x6A09E667u, s9 = 0xBB67AE85u, s10 = 0x3C6EF372u, s11 = 0xA54FF53Au, s12 = 0u, s13 = 0u, s14 = 24u,
           s15 = 1u | 2u | 8u;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    GSV_B3_G(s0, s4, s8, s12, m[0], m[1]) GSV_B3_G(s1, s5, s9, s13, m[2], m[3])
    GSV_B3_G(s2, s6, s10, s14, m[4], m[5]) GSV_B3_G(s3, s7, s11, s15, m[6], m[7])
    GSV_B3_G(s0, s5, s10, s15, m[8], m[9]) GSV_B3_G(s1, s6, s11, s12, m[10], m[11])
    GSV_B3_G(s2, s7, s8, s13, m[12], m[13]) GSV_B3_G(s3, s4, s9, s14, m[14], m[15])
    if (r < 6) {  // MSG_PERMUTATION = {2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8}
      uint32_t p[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
#pragma unroll
      for (int i = 0; i < 16; ++i) m[i] = p[i];
    }
  }
  return Label{{s0 ^ s8, s1 ^ s9, s2 ^ s10, s3 ^ s11}};
}
// Same half-gate algebra as below with the Blake3 PRF (no tweak: the gate id is part of the message).
GSV_HD void garble_and_blake3(uint32_t t, const Label& a0, const Label& b0, const Label& delta, uint64_t gate_id, Label& c0, Label& ct) {
  Label sel = lxor_if(a0, delta, alpha_a(t));
  Label oth = lxor(sel, delta);
  Label hs = blake3_hash_with_gate(sel, gate_id), ho = blake3_hash_with_gate(oth, gate_id);
  Label bsel = lxor_if(b0, delta, alpha_b(t));
  ct = lxor(lxor(hs, ho), bsel);
  c0 = lxor_if(hs, delta, alpha_c(t));
}
GSV_HD Label degarble_and_blake3(uint32_t t, const Label& ct, const Label& a, uint32_t a_value, const Label& b, uint64_t gate_id) {
  Label h = blake3_hash_with_gate(a, gate_id);
  uint32_t use_ct = (a_value ^ alpha_a(t)) & 1u;
  Label z{{0, 0, 0, 0}};
  return lxor(h, lxor_if(z, lxor(ct, b), use_ct));
}

// garble_gate, AND-family arm (halfgates_garbling.rs:17-35).  t < 8.
template <class Tab>
GSV_HD void garble_and(const Tab& T, uint32_t t, const Label& a0, const Label& b0, const Label& delta, uint64_t gate_id,
                       Label& c0, Label& ct) {
  Label tw = tweak_of(gate_id);
  Label sel = lxor_if(a0, delta, alpha_a(t));  // selected_a
  Label oth = lxor(sel, delta);                // other_a
  Label hs, ho;
  aes128_encrypt2(T, lxor(sel, tw), lxor(oth, tw), hs, ho);
  Label bsel = lxor_if(b0, delta, alpha_b(t));
  ct = lxor(lxor(hs, ho), bsel);
  c0 = lxor_if(hs, delta, alpha_c(t));
}
// garble_gate, free arm (halfgates_garbling.rs:14-16).  t in {Xor, Xnor, Not}.
GSV_HD Label garble_free(uint32_t t, const Label& a0, const Label& b0, const Label& delta) {
  Label r = (t == GT_NOT) ? a0 : lxor(a0, b0);
  return lxor_if(r, delta, (t != GT_XOR) ? 1u : 0u);
}
// degarble_gate, AND-family arm (halfgates_garbling.rs:57-67).
template <class Tab>
GSV_HD Label degarble_and(const Tab& T, uint32_t t, const Label& ct, const Label& a, uint32_t a_value, const Label& b, uint64_t gate_id) {
  Label h = hash_with_gate(T, a, gate_id);
  uint32_t use_ct = (a_value ^ alpha_a(t)) & 1u;
  Label z{{0, 0, 0, 0}};
  Label m = lxor(ct, b);
  return lxor(h, lxor_if(z, m, use_ct));
}
// degarble_gate, free arm (halfgates_garbling.rs:49-55).
GSV_HD Label degarble_free(uint32_t t, const Label& a, const Label& b) { return (t == GT_NOT) ? a : lxor(a, b); }

}  // namespace dev
}  // namespace gsv
