// Minimal host-side unsigned big integer for OFF-CIRCUIT constants only (moduli, Montgomery
// constants, shift-add schedules).  Stands in for num_bigint::BigUint as the gadgets use it
// (ref: src/gadgets/bigint/mod.rs:24-48 bits_from_biguint / bits_from_biguint_with_len).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "circuit.hpp"

namespace gsv {

class BigU {
 public:
  BigU() = default;
  explicit BigU(uint64_t v) { if (v) { limbs_.push_back(uint32_t(v)); if (v >> 32) limbs_.push_back(uint32_t(v >> 32)); } }
  static BigU from_hex(const std::string& hex) {
    BigU r;
    size_t start = (hex.size() > 1 && hex[0] == '0' && (hex[1] == 'x' || hex[1] == 'X')) ? 2 : 0;
    size_t nd = hex.size() - start;
    r.limbs_.assign((nd + 7) / 8, 0);
    for (size_t i = 0; i < nd; ++i) {
      char ch = hex[hex.size() - 1 - i];
      uint32_t d = (ch >= '0' && ch <= '9') ? uint32_t(ch - '0')
                 : (ch >= 'a' && ch <= 'f') ? uint32_t(ch - 'a' + 10)
                 : (ch >= 'A' && ch <= 'F') ? uint32_t(ch - 'A' + 10) : 0xFFu;
      if (d > 15) gsv_panic("BigU::from_hex: bad digit");
      r.limbs_[i / 8] |= d << (4 * (i % 8));
    }
    r.trim();
    return r;
  }
  bool is_zero() const { return limbs_.empty(); }
  bool bit(size_t i) const { return (i / 32 < limbs_.size()) && ((limbs_[i / 32] >> (i % 32)) & 1u); }
  size_t bits() const {  // BigUint::bits()
    if (limbs_.empty()) return 0;
    uint32_t top = limbs_.back();
    size_t n = 0;
    while (top) { ++n; top >>= 1; }
    return (limbs_.size() - 1) * 32 + n;
  }
  // bigint/mod.rs:33-48: LSB-first, exactly `len` bits; error if the value needs more.
  std::vector<bool> bits_with_len(size_t len) const {
    if (bits() > len) gsv_panic("BigUint overflow: value requires more bits than limit");
    std::vector<bool> b(len);
    for (size_t i = 0; i < len; ++i) b[i] = bit(i);
    return b;
  }
  // BigUint::to_bytes_le(): minimal little-endian bytes, [0] for zero.  Only feeds component keys.
  std::string key_bytes() const {
    std::string s;
    size_t nb = (bits() + 7) / 8;
    if (nb == 0) nb = 1;
    for (size_t i = 0; i < nb; ++i) {
      uint32_t limb = (i / 4 < limbs_.size()) ? limbs_[i / 4] : 0;
      s.push_back(char((limb >> (8 * (i % 4))) & 0xFF));
    }
    return s;
  }
  bool operator==(const BigU& o) const { return limbs_ == o.limbs_; }
  bool operator!=(const BigU& o) const { return !(*this == o); }
  const std::vector<uint32_t>& limbs() const { return limbs_; }

 private:
  void trim() { while (!limbs_.empty() && limbs_.back() == 0) limbs_.pop_back(); }
  std::vector<uint32_t> limbs_;
};

}  // namespace gsv
