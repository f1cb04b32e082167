// gfx950 (MI355X / CDNA4) kernels of the garbling engine.
//
// Execution model: one 1024-thread workgroup = one or two garbling instances on one CU.
// The circuit is deep and narrow (Fq12 mul after gate fusion: 9.1 M device records over 8 197 dependent
// steps, 6 665 of them forced by the AND depth), so a step is latency-bound as often as throughput-bound:
// the workgroup walks the program's steps with `s_barrier` between them — no grid-wide synchronisation, no
// inter-CU traffic — and independent cut-and-choose instances fill the other CUs.
//   * two-level wire file: short-lived labels in a 90 KiB LDS window chosen by the compiler's
//     fan-out/lifetime pass, long-lived ones in W[instance][slot] in HBM (128-bit accesses; the compiler
//     orders every step's records so that a wave's label accesses fall into few 128-byte lines)
//   * AES tables Te0/Te2: 64 KiB in LDS, bank-replicated so lookups never conflict, 256-byte entry stride so
//     a lookup address is one v_perm_b32; round keys in LDS too
//   * fused gates: out = AND_t(a1^a2, b1^b2) ^ p  and  out = x1^x2^x3^x4 (program.hpp); within a step the
//     AND-family records and the free records are two contiguous runs, so only one wave per step diverges
//     on the gate kind; records of step s+2 are prefetched into registers before the barrier of step s
//   * ciphertexts are written at the AND record's own index (program order: coalesced), gate order is
//     restored for the host by permute_ciphertexts_kernel / gather_segment_kernel
// No MFMA: the work is byte-table lookups and 128-bit XORs.
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>

#include "gate_math.hpp"
#include "kernel_api.h"
#include "limits.h"

// Kernel variants that were measured and rejected (branch-free operand loads, free gates first on odd waves, stores one pass late, free
// gates from the top lane down, two other issue-priority schemes, the four-lane quad form for every program) live as a patch against this
// file in profiles/r05_kernel/rejected_variants.patch, with their A/B logs beside it; this file holds the adopted code only.

namespace gsv {
namespace dev {

__constant__ uint32_t c_rk[44];

// Native 128-bit / 64-bit vectors (copyable across address spaces, one dwordx4 / dwordx2 access each).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// LDS map of a workgroup (limits.h; 163,632 of the CU's 163,840 bytes):
//   [0, 64 KiB)   AES tables.  Entry x occupies the 256-byte stride [x*256, x*256+256): dwords 0..31 hold Te0[x]
//                 replicated once per LDS bank, dwords 32..63 hold Te2[x] = rotl16(Te0[x]) likewise.
//                 * bank = (address/4) mod 32 = lane & 31 for both halves, and a wave's ds_read_b32 is served in
//                   two 32-lane groups (MI355X_MICROARCH.md §LDS): 64 data-dependent lookups never conflict;
//                 * the 256-byte stride puts the index byte on a byte boundary of the address, so the whole lookup
//                   address {byte1 = state byte, byte0 = lane*4} is ONE v_perm_b32 (an address built as
//                   bfe + lshl_or cost two VALU ops per lookup = 45 % of the AES instruction stream);
//                 * Te1 = rotl8(Te0), Te3 = rotl8(Te2): one v_alignbit for half of the lookups, none in the last round.
//   [64 KiB, +90 KiB)      label window: GSV_LDS_SLOTS x 16 B, the short-lived wires chosen by the compiler
//   [.., +5.6 KiB)         plaintext bits of window wires (evaluate mode)
//   [.., +16 B)            arrival counters of the per-group step barrier, one per instance group (FW instantiations)
//   [.., +176 B)           the 44 round-key words (read with a wave-uniform address = LDS broadcast; keeping them
//                          in SGPRs instead spilled ~60 SGPRs and put v_readlane/v_writelane into every step)
#define GSV_SLOT_LDS_FLAG (1u << 20)
#define GSV_SLOT_INDEX_MASK (GSV_SLOT_LDS_FLAG - 1u)
#define GSV_SLOT_MASK ((1u << 21) - 1u)

// Address-space-qualified pointers keep every access a plain ds_* or global_* instruction (no
// generic/flat pointers, which would also need a null-checked LDS->flat cast).
#define GSV_LDS __attribute__((address_space(3)))
#define GSV_GLB __attribute__((address_space(1)))
typedef uint32_t GSV_LDS lds_u32;
typedef u32x4 GSV_LDS lds_u128;
typedef uint8_t GSV_LDS lds_u8;
typedef u32x4 GSV_GLB glb_u128;
typedef u32x2 GSV_GLB glb_u64;
typedef uint8_t GSV_GLB glb_u8;
#define GSV_CST __attribute__((address_space(4)))
typedef const u32x4 GSV_CST cst_u128;   // read-only data behind a wave-uniform address: s_load (scalar cache), not a vector load
typedef const uint32_t GSV_CST cst_u32;

struct LdsBankedTable {
  static constexpr bool kPairedRotation = true;  // Te0 / Te2 only; c_rk[4..39] hold rotr8 of the middle rounds' keys (gate_math.hpp, aes_col)
  uint32_t lane4;  // (lane & 31) * 4
  template <int K, int BYTE>
  __device__ __forceinline__ uint32_t lk(uint32_t s) const {
    // LDS address = {0, 0, byte BYTE of s, lane4}: v_perm_b32 over {s (bytes 4..7), lane4 (bytes 0..3)}; 0x0c = zero byte
    const uint32_t addr = __builtin_amdgcn_perm(s, lane4, 0x0c0c0000u | (uint32_t(4 + BYTE) << 8));
    const uint32_t v = *reinterpret_cast<const lds_u32*>(uintptr_t(addr + ((K & 2) ? 128u : 0u)));  // Te0 or Te2 half
    return (K & 1) ? __builtin_amdgcn_alignbit(v, v, 24) : v;                                       // Te1 / Te3 = rotl8
  }
  // Round keys of the one-gate-per-lane AES come through the scalar cache (s_load from constant memory): the LDS pipe is
  // the co-limiter of that loop (364 ds_read per block pair, 44 of them round keys), and although the 44 SGPRs do not all
  // stay resident (the compiler parks ~90 scalars in VGPR lanes), trading 44 LDS reads for v_readlane is +2.4 %.
  // The quad form keeps its 11 per-lane round-key words in VGPRs, loaded once from the LDS copy.
  cst_u32* rkp;  // = c_rk; laundered (asm volatile) at the top of every one-gate-per-lane AES pass so that the 44 s_loads are
                 // issued THERE and their SGPRs are live only across the pass, not across the whole step loop
  __device__ __forceinline__ uint32_t rk(int i) const { return rkp[i]; }
};

typedef uint32_t GSV_GLB glb_u32;

struct WireFile {
  glb_u128* hbm;      // this instance's wire file
  glb_u8* hbm_bits;   // evaluate: plaintext bits
  uint32_t win_base;  // LDS byte address of this instance's label window
  uint32_t bit_base;  // LDS byte address of this instance's window plaintext bits
  // `slot` still carries GSV_SLOT_LDS_FLAG: (slot << 4) + (win_base - FLAG * 16) is ONE v_lshl_add_u32 (mod 2^32), where masking
  // the flag off first costs an AND and a separate shift
  __device__ __forceinline__ lds_u128* win(uint32_t slot) const { return reinterpret_cast<lds_u128*>(uintptr_t((slot << 4) + (win_base - (GSV_SLOT_LDS_FLAG << 4)))); }
  __device__ __forceinline__ lds_u32* win_word(uint32_t slot, uint32_t c) const { return reinterpret_cast<lds_u32*>(uintptr_t((slot << 4) + (win_base - (GSV_SLOT_LDS_FLAG << 4)) + c * 4u)); }
  __device__ __forceinline__ lds_u8* win_bit(uint32_t idx) const { return reinterpret_cast<lds_u8*>(uintptr_t(bit_base + idx)); }
  // A label lives EITHER in the LDS window or in the HBM wire file, lane by lane.  Written as `if (lds) v = *win(slot); else v = hbm[slot];`
  // both loads target the same registers, and since LDS and vector-memory results return out of order with respect to each other the
  // compiler puts `s_waitcnt lgkmcnt(0)` between them: a wave that holds both kinds of operands waits for each operand's LDS round trip
  // before it issues that operand's global load.  Round 3 measured the alternative (two register sets and a select, all ten loads of a
  // gate in flight at once): 24 fewer waits per step body, but +13 VGPRs, 60 more spilled scalars (the lane masks of the selects) and
  // 20 v_cndmask per gate — 5 % SLOWER on wide programs (9.77 against 10.27 x 10^10 gates/s at 1 024 instances), 6 % on the ladders, 4 %
  // on the inversions, +-0 for a single instance (profiles/r03_kernel): the other waves of the CU already hide those waits.
  __device__ __forceinline__ Label ld(uint32_t slot) const {
    u32x4 v;
    if (slot & GSV_SLOT_LDS_FLAG) v = *win(slot); else v = hbm[slot];
    return Label{{v.x, v.y, v.z, v.w}};
  }
  __device__ __forceinline__ void st(uint32_t slot, const Label& l) const {
    const u32x4 v = {l.w[0], l.w[1], l.w[2], l.w[3]};
    if (slot & GSV_SLOT_LDS_FLAG) *win(slot) = v; else hbm[slot] = v;
  }
  // one 32-bit column of a label (narrow-step mode: a label is spread over the 4 lanes of a quad)
  __device__ __forceinline__ uint32_t ld_word(uint32_t slot, uint32_t c) const {
    uint32_t v;
    if (slot & GSV_SLOT_LDS_FLAG) v = *win_word(slot, c); else v = ((const glb_u32*)hbm)[slot * 4u + c];
    return v;
  }
  __device__ __forceinline__ void st_word(uint32_t slot, uint32_t c, uint32_t v) const {
    if (slot & GSV_SLOT_LDS_FLAG) *win_word(slot, c) = v; else ((glb_u32*)hbm)[slot * 4u + c] = v;
  }
  __device__ __forceinline__ uint32_t ld_bit(uint32_t slot) const {
    uint32_t b;
    if (slot & GSV_SLOT_LDS_FLAG) b = *win_bit(slot & GSV_SLOT_INDEX_MASK); else b = hbm_bits[slot];
    return b;
  }
  __device__ __forceinline__ void st_bit(uint32_t slot, uint32_t b) const {
    if (slot & GSV_SLOT_LDS_FLAG) *win_bit(slot & GSV_SLOT_INDEX_MASK) = uint8_t(b); else hbm_bits[slot] = uint8_t(b);
  }
};

// ---- AES-128 with ONE BLOCK SPREAD OVER THE 4 LANES OF A QUAD (lane c holds state column c). -------------
// A lone wave needs ~5 us for the per-lane two-block AES (~1400 dependent-ish VALU + 320 LDS ops), and most
// steps of the circuit hold fewer AND gates than one wave has lanes.  Spreading a block over a quad cuts the
// per-lane instruction stream ~6x: every lane looks up the four T-table terms of ITS OWN column and the
// ShiftRows routing becomes three quad_perm DPP moves:
//   new s_c = Te0[b0(s_c)] ^ Te1[b1(s_c+1)] ^ Te2[b2(s_c+2)] ^ Te3[b3(s_c+3)] ^ rk[4r+c]
template <int CTRL>
__device__ __forceinline__ uint32_t quad_from(uint32_t v) {  // lane 4k+i reads lane 4k+perm[i]
  return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xF, 0xF, false));
}
#define GSV_QP_NEXT1 0x39  // quad_perm [1,2,3,0]
#define GSV_QP_NEXT2 0x4E  // quad_perm [2,3,0,1]
#define GSV_QP_NEXT3 0x93  // quad_perm [3,0,1,2]
__device__ __forceinline__ uint32_t aes128_quad(const LdsBankedTable& T, const uint32_t (&rkc)[11], uint32_t s) {
  s ^= rkc[0];
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    // rotated terms paired under one rotation (aes_col): rkc[1..9] are rotr8 of the round keys
    const uint32_t u0 = T.lk<0, 0>(s), u1 = T.lk<0, 1>(s), u2 = T.lk<2, 2>(s), u3 = T.lk<2, 3>(s);
    const uint32_t odd = rkc[r] ^ quad_from<GSV_QP_NEXT1>(u1) ^ quad_from<GSV_QP_NEXT3>(u3);
    s = __builtin_amdgcn_alignbit(odd, odd, 24) ^ (u0 ^ quad_from<GSV_QP_NEXT2>(u2));
  }
  const uint32_t m0 = T.lk<2, 0>(s) & 0x000000ffu, m1 = T.lk<0, 1>(s) & 0x0000ff00u;  // S-box byte from the un-rotated tables
  const uint32_t m2 = T.lk<0, 2>(s) & 0x00ff0000u, m3 = T.lk<2, 3>(s) & 0xff000000u;
  return m0 ^ quad_from<GSV_QP_NEXT1>(m1) ^ quad_from<GSV_QP_NEXT2>(m2) ^ quad_from<GSV_QP_NEXT3>(m3) ^ rkc[10];
}
// TWO blocks in the same quad, interleaved (garbling hashes x and x ^ delta under one tweak): the lookups per block are the same, the two
// dependency chains run in each other's latency shadow, and an AND gate takes FOUR lanes instead of eight.
__device__ __forceinline__ void aes128_quad_x2(const LdsBankedTable& T, const uint32_t (&rkc)[11], uint32_t& s, uint32_t& z) {
  s ^= rkc[0]; z ^= rkc[0];
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    const uint32_t u0 = T.lk<0, 0>(s), u1 = T.lk<0, 1>(s), u2 = T.lk<2, 2>(s), u3 = T.lk<2, 3>(s);
    const uint32_t v0 = T.lk<0, 0>(z), v1 = T.lk<0, 1>(z), v2 = T.lk<2, 2>(z), v3 = T.lk<2, 3>(z);
    const uint32_t odd = rkc[r] ^ quad_from<GSV_QP_NEXT1>(u1) ^ quad_from<GSV_QP_NEXT3>(u3);
    const uint32_t odz = rkc[r] ^ quad_from<GSV_QP_NEXT1>(v1) ^ quad_from<GSV_QP_NEXT3>(v3);
    s = __builtin_amdgcn_alignbit(odd, odd, 24) ^ (u0 ^ quad_from<GSV_QP_NEXT2>(u2));
    z = __builtin_amdgcn_alignbit(odz, odz, 24) ^ (v0 ^ quad_from<GSV_QP_NEXT2>(v2));
  }
  const uint32_t m0 = T.lk<2, 0>(s) & 0x000000ffu, m1 = T.lk<0, 1>(s) & 0x0000ff00u, m2 = T.lk<0, 2>(s) & 0x00ff0000u, m3 = T.lk<2, 3>(s) & 0xff000000u;
  const uint32_t n0 = T.lk<2, 0>(z) & 0x000000ffu, n1 = T.lk<0, 1>(z) & 0x0000ff00u, n2 = T.lk<0, 2>(z) & 0x00ff0000u, n3 = T.lk<2, 3>(z) & 0xff000000u;
  s = m0 ^ quad_from<GSV_QP_NEXT1>(m1) ^ quad_from<GSV_QP_NEXT2>(m2) ^ quad_from<GSV_QP_NEXT3>(m3) ^ rkc[10];
  z = n0 ^ quad_from<GSV_QP_NEXT1>(n1) ^ quad_from<GSV_QP_NEXT2>(n2) ^ quad_from<GSV_QP_NEXT3>(n3) ^ rkc[10];
}
__device__ __forceinline__ uint32_t tweak_word(uint64_t gate_id, uint32_t c) {  // column c of tweak_of(gate_id)
  const uint64_t t0 = gate_id ^ 0x123456789ABCDEF0ull, t1 = gate_id * 0xDEADBEEFCAFEBABEull;
  const uint64_t t = (c & 2u) ? t1 : t0;
  return (c & 1u) ? uint32_t(t >> 32) : uint32_t(t);
}

// NI = instances per workgroup (1, 2 or 4).  NI > 1 splits the 1024 threads into NI groups that garble NI instances of the
// same program in lockstep (they share the step barrier and the AES table; each has 1/NI of the label window): the many
// steps that are narrower than a group then cost their fixed latency once for NI instances.  The host picks the largest NI
// that still gives every CU a workgroup (2 above 256 instances: +16 %; 4 above 512: at 1024 instances the same rate on wide
// programs as NI = 2, +21 % on the decompression ladders and +29 % on the inversions, profiles/r02_final/ni4_vs_ni2.txt).
// Measured and rejected: per-group software barriers (LDS arrival counters) that let the groups drift into different
// phases, with and without a start skew (no gain over lockstep: the waves of the groups already overlap each other's
// latencies).
// HASH = 0: AesNiHasher (fixed-key AES, the hot path); HASH = 1: Blake3Hasher (src/hashers/mod.rs:22-51; the PRF most of
// the reference's own tests use) — pure 32-bit add/xor/rotate, one gate per lane in every step (no multi-lane form).
// FW: the launch contains a program in the four-wire record form (program.hpp pack_and4) — the decode and the four extra operand
// loads exist only in these instantiations, so the throughput-bound launches (all programs two-wire) run the code they always ran.
template <bool EVAL, int NI, int HASH, bool FW>
__global__ __launch_bounds__(GSV_BLOCK_THREADS) void run_program_kernel(KernelArgs ka) {
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  (void)s_mem;  // the dynamic LDS block starts at LDS address 0 (no static __shared__ in this kernel)
  // dword i of the table region: entry x = i / 64; first 32 dwords Te0[x], next 32 dwords Te2[x] (ka.te = Te0..Te3, 256 words each)
  for (uint32_t i = threadIdx.x; i < GSV_LDS_TABLE_BYTES / 4; i += GSV_BLOCK_THREADS) *reinterpret_cast<lds_u32*>(uintptr_t(i * 4u)) = ka.te[((i & 32u) ? 512u : 0u) + (i >> 6)];
  if (threadIdx.x < 44) *reinterpret_cast<lds_u32*>(uintptr_t(GSV_LDS_RK_BASE + 4u * threadIdx.x)) = c_rk[threadIdx.x];
  // window entry 0 of every instance: the all-zero label (and plaintext bit 0) that absent operands of fused gates name
  if (threadIdx.x < 4u * NI) *reinterpret_cast<lds_u32*>(uintptr_t(GSV_LDS_TABLE_BYTES + (threadIdx.x / 4u) * (GSV_LDS_SLOTS / NI) * 16u + (threadIdx.x % 4u) * 4u)) = 0u;
  if (threadIdx.x < uint32_t(NI)) *reinterpret_cast<lds_u8*>(uintptr_t(GSV_LDS_TABLE_BYTES + GSV_LDS_SLOTS * 16u + threadIdx.x * (GSV_LDS_SLOTS / NI))) = uint8_t(0);
  __syncthreads();
  // Window launch (plans, schedule.hpp): blockIdx.y selects a call of the window; everything that differs between calls comes from
  // a 96-byte descriptor behind a wave-uniform address (scalar loads).
  typedef const CallDesc GSV_CST cst_call;
  cst_call* const cd = ka.calls ? (cst_call*)ka.calls + blockIdx.y : nullptr;
  uint32_t w_base = 0;
  if (cd) {
    ka.steps = cd->steps; ka.ands = cd->ands; ka.xors = cd->xors;
    ka.gid_base += cd->gid_off; ka.ct_offset = cd->ct_off; ka.n_steps = cd->n_steps;
    ka.and_terms = cd->and_terms;
    w_base = cd->w_base;
  }
  // record form of this program (program.hpp): up to two or up to four wires per AND input — wave-uniform, the same for every gate of the launch's call
  const bool four_wire = FW && __builtin_amdgcn_readfirstlane(int(ka.and_terms)) == 4;
  constexpr uint32_t BT = GSV_BLOCK_THREADS / NI;  // threads per instance
  // which instance of this workgroup: wave-uniform (BT is a multiple of 64), so say so — every per-instance base
  // address below then lives in SGPRs instead of costing a VGPR each
  const uint32_t sub = NI == 1 ? 0u : uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x / BT)));
  // Lane index inside the instance's thread group; for the latency-bound (four-wire) programs ROTATED by one wave per group.  A
  // workgroup's waves go to the four SIMDs cyclically, so with four groups of four waves, wave j of every group sits on the same SIMD: the
  // partly filled pass of a narrow step (lanes 0 .. r of every group: a ladder step holds 137 AND gates per instance) loads the same two or
  // three SIMDs in all four groups and leaves the rest idle.  With the rotation group g's lane 0 is its wave g, and the groups' waves land
  // on different SIMDs: ladders +11 %, inversions +8 % at 1 024 instances (profiles/r05_kernel/kernel_ab_rot.log).  A pure relabelling of
  // lanes in wave units: quad roles (tid & 7), the LDS bank (tid & 31) and wave uniformity are unchanged.  NOT for the wide (two-wire)
  // programs: there the imbalance lets the less loaded SIMD's waves start the step's free gates under the others' last AES pass, and
  // balancing the SIMDs costs 3 %.
  const uint32_t tid = (NI > 1 && four_wire) ? (((threadIdx.x - sub * BT) + sub * 64u) & (BT - 1u)) : threadIdx.x - sub * BT;
  const LdsBankedTable aes{(tid & 31u) * 4u, (cst_u32*)c_rk};
  // narrow-step mode: LPG lanes per AND gate (garble: two blocks x 4 columns; evaluate: one block x 4 columns)
  constexpr uint32_t LPG = EVAL ? 4u : 8u;
  // Second multi-lane form (garbling four-wire programs: the latency-bound ladders and inversions): both blocks of a gate interleaved in ONE
  // quad (aes128_quad_x2), four lanes per gate.  At four instances per workgroup an instance has 256 lanes = 32 gates per eight-lane pass,
  // and half of an inversion's steps hold 33..48 AND gates, a fifth of a ladder's 65..96: they took two passes, or one partly filled
  // one-gate-per-lane pass at its ~5 us, and take one / two four-lane passes now.
  // (compiled into every garbling instantiation it costs the wide programs 2.3 % and the Miller loop 2.6 % at 1 024 instances,
  // profiles/r05_kernel/kernel_ab_dualall.log: it exists in the FW instantiations only)
  constexpr bool DUAL = FW && !EVAL && HASH == 0;
  const bool dual_prog = four_wire;
  constexpr uint32_t LPG2 = DUAL ? 4u : LPG;
  const uint32_t col = tid & 3u, blk = (tid >> 2) & 1u;
  uint32_t rkc[11];
#pragma unroll
  for (int r = 0; r < 11; ++r) rkc[r] = *reinterpret_cast<const lds_u32*>(uintptr_t(GSV_LDS_RK_BASE + 4u * (4u * uint32_t(r) + col)));

  uint32_t inst = blockIdx.x * NI + sub;
  const bool inst_active = inst < ka.n_instances;  // an odd batch leaves the last half idle (it still joins every barrier)
  if (!inst_active) inst = ka.n_instances - 1;     // harmless addresses; every gate loop below is masked off
  WireFile wf;
  wf.win_base = GSV_LDS_TABLE_BYTES + sub * (GSV_LDS_SLOTS / NI) * 16u;
  wf.bit_base = GSV_LDS_TABLE_BYTES + GSV_LDS_SLOTS * 16u + sub * (GSV_LDS_SLOTS / NI);
  wf.hbm = (glb_u128*)(ka.W + size_t(inst) * ka.n_slots + w_base);      // C-style cast: generic -> global address space
  wf.hbm_bits = (glb_u8*)(ka.VB + size_t(inst) * ka.n_slots + w_base);
  glb_u128* __restrict__ CT = (glb_u128*)(ka.CT + size_t(inst) * ka.ct_stride + ka.ct_offset);
  glb_u32* __restrict__ CTw = (glb_u32*)CT;
  Label delta{{0, 0, 0, 0}};
  if (!EVAL) { const u32x4 d = ((const glb_u128*)ka.delta)[inst]; delta = Label{{d.x, d.y, d.z, d.w}}; }
  // this lane's column of delta (selects, not a runtime-indexed array: that would be promoted to static LDS)
  const uint32_t dq = col == 0 ? delta.w[0] : col == 1 ? delta.w[1] : col == 2 ? delta.w[2] : delta.w[3];
  if (cd) {
    // ---- dataflow prologue: wait for the calls this one depends on (same instance group), then fetch the inputs.
    // One lane polls; the flags are written with agent-scope release by the producer's epilogue (possibly on another XCD: the
    // acquire fence below invalidates this CU's L1 and the non-local L2 lines before any wave reads the global wires).
    // The watchdog is PROGRESS based: the last word of the group's flag row counts the calls the group has completed (any window,
    // never reset); the wait gives up only when that counter has not moved for ka.wait_ticks (100 MHz; 60 s by default) — a long
    // window, a second session on the device or a slow hasher stretch the wait but keep the counter moving.
    if (threadIdx.x == 0 && cd->n_deps) {
      const uint32_t* const fl = ka.flags + size_t(blockIdx.x) * ka.flag_stride;
      const uint32_t* const progress = fl + (ka.flag_stride - 1u);
      unsigned long long t_start = wall_clock64();
      uint32_t seen = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool gave_up = false;
      for (uint32_t i = 0; i < cd->n_deps && !gave_up; ++i) {
        const uint32_t* const f = fl + ka.deps[cd->dep_off + i];
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ka.epoch) {
          __builtin_amdgcn_s_sleep(16);
          if (wall_clock64() - t_start > ka.wait_ticks) {
            const uint32_t now = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (now != seen) { seen = now; t_start = wall_clock64(); continue; }
            // no call of this instance group has completed for the whole interval: a dependency that never completes (violated
            // dispatch-order assumption).  The host refuses the results (engine.cpp, check_plan_error).
            __hip_atomic_store(ka.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
            break;
          }
        }
      }
    }
    // ---- ciphertext ring: the call's block inside the ring is free (garbling: what it held on the previous lap has been gathered off
    // the device) / filled (evaluating: the call's segment has been uploaded) once the host's position counter says so.  The counter
    // lives in host memory mapped into the device (fine-grained, system scope); the same progress rule as above bounds the wait.
    const unsigned long long* const ct_pos = threadIdx.x == 0 ? cd->ct_pos : nullptr;
    if (ct_pos) {
      const unsigned long long want = EVAL ? cd->ct_ready : cd->ct_need;
      unsigned long long t_start = wall_clock64(), seen = __hip_atomic_load(ct_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      while (seen < want) {
        __builtin_amdgcn_s_sleep(64);
        const unsigned long long now = __hip_atomic_load(ct_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (now != seen) { seen = now; t_start = wall_clock64(); continue; }
        if (wall_clock64() - t_start > ka.wait_ticks) {  // (the host publishes a position per drain segment: fractions of a second apart)
          __hip_atomic_store(ka.error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // who gave up, on what (diagnostics for engine.cpp, check_plan_error; whichever call writes last is reported)
          const unsigned long long waited = wall_clock64() - t_start;
          ka.error[8] = blockIdx.y; ka.error[9] = blockIdx.x;
          ka.error[10] = uint32_t(want); ka.error[11] = uint32_t(want >> 32);
          ka.error[12] = uint32_t(seen); ka.error[13] = uint32_t(seen >> 32);
          ka.error[14] = uint32_t(waited); ka.error[15] = uint32_t(waited >> 32);
          break;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t n_pre = inst_active ? cd->n_pre : 0u;
    const glb_u128* const gw = (const glb_u128*)(ka.W + size_t(inst) * ka.n_slots);
    glb_u128* const gwd = (glb_u128*)(ka.W + size_t(inst) * ka.n_slots);
    for (uint32_t i = tid; i < n_pre; i += BT) {
      const uint32_t sidx = ka.copy_src[cd->pre_off + i], didx = ka.copy_dst[cd->pre_off + i];
      gwd[didx] = gw[sidx];
      if (EVAL) { glb_u8* const vb = (glb_u8*)(ka.VB + size_t(inst) * ka.n_slots); vb[didx] = vb[sidx]; }
    }
    __syncthreads();
  }
  // Per-GROUP step barrier (round 6) for the latency-bound (four-wire) programs at several instances per workgroup: the instance groups of a
  // workgroup garble independent instances and need not pass their steps together — an arrival counter per group in LDS, polled by the
  // group's own waves, replaces the workgroup-wide s_barrier, so a group whose waves are done with a narrow step starts the next one without
  // waiting for the slowest wave of the other three groups.  Ladders +7.1 %, inversions +2.0 % at 1 024 instances; the wide (two-wire)
  // programs lose 0.5 % to the polling (profiles/r06_kernel/kernel_ab_gbar.log): the launches without a four-wire program (two thirds of a
  // pass) keep s_barrier, the FW instantiations use the group barrier for every call of their windows (one barrier form per instantiation:
  // the compiler's output — and build.py's ISA check of it — stays simple).  Round 2 had measured +-0 for the same idea at two groups per
  // workgroup on the programs of that time; a start skew between the groups adds nothing (gbar2 in the same log).
  constexpr bool group_barrier = FW && NI > 1;
  uint32_t bar_target = 0;  // arrivals of this group's waves so far
  uint32_t bar_addr = GSV_LDS_GROUP_BAR_BASE + 4u * sub, bar_one = 1u, bar_val = 0u;  // the barrier's own VGPRs (see the step barrier below)
  if (group_barrier) {
    asm volatile("" : "+v"(bar_addr), "+v"(bar_one), "+v"(bar_val));
    if (threadIdx.x < uint32_t(NI)) *reinterpret_cast<lds_u32*>(uintptr_t(GSV_LDS_GROUP_BAR_BASE + 4u * threadIdx.x)) = 0u;
    __syncthreads();
  }
  cst_u128* const step_q = (cst_u128*)ka.steps;
  // Timing ablations (GSV_DIAG, kernel_api.h) exist only in a library built with -DGSV_DIAG_BUILD (build.py --diag): as run-time
  // flags they cost the production loop a dozen register initialisations and several branches per gate.
#ifdef GSV_DIAG_BUILD
  const bool no_store = (ka.diag & 8u) != 0, no_load = (ka.diag & 4u) != 0, no_aes = (ka.diag & 1u) != 0, no_narrow = (ka.diag & 16u) != 0;
  const bool no_barrier = (ka.diag & 32u) != 0, no_refill = (ka.diag & 64u) != 0, no_hi = (ka.diag & 128u) != 0;  // step skeleton: what is the floor made of?
  // bit 256: phase clock of the narrow steps.  Wave 0 of workgroup (0, 0) stamps the 100 MHz wall clock (s_memrealtime, consumed only
  // behind the step barrier: no extra waits) at seven points of every narrow step in which it garbles an AND gate and accumulates the six
  // intervals + the time between two steps; the sums land in ka.step_clock[0..8] at the end of the kernel (tools/narrow_phase_clock.py).
  const bool phase_clock = (ka.diag & 256u) != 0 && ka.step_clock && blockIdx.x == 0 && blockIdx.y == 0;
  unsigned long long pc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc_t[7] = {0, 0, 0, 0, 0, 0, 0}, pc_prev_end = 0;
  // (each stamp waits for its own result — and with it for the wave's outstanding LDS operations: at the stamp points none are in flight
  // except the label store of point 4, whose completion the step barrier would wait for anyway)
#define GSV_PC_STAMP(i, dep) do { if (phase_clock) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pc_t[i]) : "v"(dep) : "memory"); } while (0)
#else
#define GSV_PC_STAMP(i, dep) do { } while (0)
  constexpr bool no_store = false, no_load = false, no_aes = false, no_narrow = false, no_barrier = false, no_refill = false, no_hi = false;
#endif

  for (uint32_t rep = 0; rep < ka.replays; ++rep) {
    const uint64_t gid_base = ka.gid_base + uint64_t(ka.rep_base + rep) * ka.n_gates;
    const uint64_t ct_base = uint64_t((ka.rep_base + rep) % ka.ct_cap_replays) * ka.n_ct;

    // A step descriptor is {and_off, and_cnt, xor_off, xor_cnt}.  Lane -> gate mapping of a step:
    //   wide   (default)                : lane i takes AND gate i for i < and_cnt, else free gate i - and_cnt;
    //                                     more than 1024 gates => several passes.
    //   narrow (and_cnt*LPG + xor_cnt <= 1024): LPG consecutive lanes share AND gate i/LPG, free gates follow.
    // Either way AND work and XOR work are contiguous lane ranges: only a boundary wave diverges on the gate kind.
    //
    // Records (program.hpp): AND-family 32 bytes {a1,a2,b1 | b2,p,out | gate id, type}, free 16 bytes {x1,x2,x3 | x4,out}:
    //   out = AND_t(a1 ^ a2, b1 ^ b2) ^ p      /      out = x1 ^ x2 ^ x3 ^ x4 (^ delta)
    // Absent operands name window entry 0 (an all-zero label), so every lane runs the same instruction stream.
    // The ciphertext of AND record k goes to position k of the replay's block of the stream.
    //
    // Software pipeline, two steps deep: while step s computes, the descriptor (scalar load: the whole
    // step bookkeeping stays on the scalar unit, which matters because all 16 waves run it even when only one
    // has gates) + this lane's first record of step s+2 are in flight (program records stream from HBM).
    //
    // The step barrier is `s_waitcnt lgkmcnt(0); s_barrier`: it waits for the wave's LDS stores (window labels) and scalar
    // loads, NOT for its vector-memory operations.  The waves of a workgroup share their CU's vector L1 (this code object is
    // not built for threadgroup-split mode), which serves their global accesses in issue order: a label stored to the HBM
    // wire file before the barrier is seen by any wave of the workgroup that loads it after the barrier, whether or not the
    // store has been acknowledged by L2.  That is the AMDGPU memory model's workgroup-scope release/acquire on gfx90a+,
    // and exactly what hipcc emits for `global store; __syncthreads(); global load` on gfx950 (no vmcnt wait in front of
    // s_barrier, no cache invalidate behind it; tests/test_engine_host.py pins that on the compiler's output).  So a step
    // never waits for store acknowledgements (~1 us each), and the record prefetch and the ciphertext stores stay in
    // flight across the barrier by themselves.
    const uint32_t last_step = ka.n_steps - 1;
    // narrow step: its AND gates in the multi-lane form (LPG lanes each) and — from the next wave boundary — its free gates fit ONE pass
    // (the value: lanes per gate of the pass — LPG, or LPG2 when the gates only fit at four lanes each; 0: not a narrow step)
    auto narrow_lpg = [&](const u32x4& d) -> uint32_t {
      if (HASH != 0 || no_narrow || d.y == 0) return 0u;
      if (((d.y * LPG + 63u) & ~63u) + d.w <= BT) return LPG;
      if (DUAL && dual_prog && ((d.y * LPG2 + 63u) & ~63u) + d.w <= BT) return LPG2;
      return 0u;
    };
    auto is_narrow = [&](const u32x4& d) -> bool { return narrow_lpg(d) != 0u; };
    // wide steps: is the AND remainder small enough for a multi-lane form?  (lanes per gate of the remainder's passes; 0: one partly filled
    // one-gate-per-lane pass)
    auto rem_lpg = [&](uint32_t n_and) -> uint32_t {
      if (HASH != 0) return 0u;
      const uint32_t rem = n_and % BT;
      // (up to two four-lane passes; three: -3 %, profiles/r05_kernel/kernel_ab_dualrem.log)
      if (DUAL && dual_prog) return rem <= BT / LPG ? LPG : rem <= 2u * (BT / LPG2) ? LPG2 : rem <= 2u * (BT / LPG) ? LPG : 0u;
      return rem <= 2u * (BT / LPG) ? LPG : 0u;
    };
    auto small_rem = [&](uint32_t n_and) -> bool { return rem_lpg(n_and) != 0u; };
    auto load_desc = [&](uint32_t s) -> u32x4 {
      u32x4 d = step_q[s < last_step ? s : last_step];  // wave-uniform address: a scalar (SMEM) load, two steps ahead of its use
      if (!inst_active) { d.y = 0; d.w = 0; }
      return d;
    };
    // First 16 bytes (the operand / output slots) of this lane's first (or only) record of step d.  Lanes without a gate
    // read the step table's first entry (always present): one unconditional load, no divergence.  The second half of an AND record
    // (gate id, third type bit) is only needed when the AES starts: it is loaded together with the operands.
    const glb_u8* const and_bytes = (const glb_u8*)ka.ands;
    const glb_u8* const xor_bytes = (const glb_u8*)ka.xors;
    typedef u32x4 Rec;
    // Narrow steps: AND lanes first (LPG per gate), free-gate lanes behind them from the next WAVE boundary, so that no wave runs the AES
    // path and the free-gate path one after the other (round 4: the inversions' one-instance rate +15 %, the ladders' +5 %,
    // profiles/r04_kernel/kernel_ab_np1.log).
    auto xor_lane0 = [&](uint32_t na) -> uint32_t { return (na + 63u) & ~63u; };
    auto rec_ptr = [&](const u32x4& d) -> const glb_u8* {
      const glb_u8* p = (const glb_u8*)ka.steps;
      const uint32_t nl = narrow_lpg(d);
      if (nl) {
        const uint32_t na = d.y * nl, x0 = xor_lane0(na);
        if (tid < na) p = and_bytes + size_t(d.x + tid / nl) * 32u;
        else if (tid >= x0 && tid < x0 + d.w) p = xor_bytes + size_t(d.z + (tid - x0)) * 16u;
      } else if (d.y >= BT || !small_rem(d.y)) { if (tid < d.y) p = and_bytes + size_t(d.x + tid) * 32u; }  // wide: first one-per-lane pass
      else { const uint32_t rl = rem_lpg(d.y); if (tid / rl < d.y) p = and_bytes + size_t(d.x + tid / rl) * 32u; }  // wide with only a small remainder: first multi-lane pass
      return p;
    };
    auto load_rec = [&](const u32x4& d) -> Rec { return *(const glb_u128*)rec_ptr(d); };
    auto load_and_rec = [&](uint32_t k) -> Rec { return *(const glb_u128*)(and_bytes + size_t(k) * 32u); };      // record k of the AND array, first half
    auto load_and_hi = [&](uint32_t k) -> u32x4 {  // second half (two-wire form: its first 8 bytes)
      if (no_hi) return u32x4{k, 0u, 0u, 0u};
      if (FW) return *(const glb_u128*)(and_bytes + size_t(k) * 32u + 16u);
      const u32x2 h = *(const glb_u64*)(and_bytes + size_t(k) * 32u + 16u);
      return u32x4{h.x, h.y, 0u, 0u};
    };
    // Two record registers in ping-pong: step s consumes one (loaded at the end of step s-2) and, once it is done with it,
    // refills the SAME registers with the record of step s+2.  No in-flight load is ever copied to another register: a copy
    // would make the compiler wait for the load it has just issued at the top of every step (which is what a rotating
    // r0 <- n0 <- n2r form did, exposing a full L2 round trip per step).
    u32x4 sdA = load_desc(0), sdB = load_desc(1);
    Rec recA = load_rec(sdA), recB = load_rec(sdB);
    // decoded AND record
    // two-wire form: a1 a2 b1 b2 p c | gid, type.  four-wire form: a1 a2 a3 a4 b1 b2 | b3 b4 p, c, gid (a3 a4 b3 b4 are only read when four_wire)
    struct AndOp { uint32_t a1, a2, a3, a4, b1, b2, b3, b4, p, c, t; uint64_t gid; };
    auto decode_and = [&](const Rec& q, const u32x4& hi) -> AndOp {
      AndOp o;
      const uint32_t s0 = q.x & GSV_SLOT_MASK, s1 = ((q.x >> 21) | (q.y << 11)) & GSV_SLOT_MASK, s2 = (q.y >> 10) & GSV_SLOT_MASK;
      const uint32_t s3 = q.z & GSV_SLOT_MASK, s4 = ((q.z >> 21) | (q.w << 11)) & GSV_SLOT_MASK, s5 = (q.w >> 10) & GSV_SLOT_MASK;
      o.a1 = s0; o.a2 = s1;
      if (four_wire) {
        o.a3 = s2; o.a4 = s3; o.b1 = s4; o.b2 = s5;
        o.b3 = hi.x & GSV_SLOT_MASK;
        o.b4 = ((hi.x >> 21) | (hi.y << 11)) & GSV_SLOT_MASK;
        o.p = (hi.y >> 10) & GSV_SLOT_MASK;
        o.c = hi.z & GSV_SLOT_MASK;
        o.t = (q.y >> 31) | ((q.w >> 31) << 1) | ((hi.y >> 31) << 2);
        o.gid = gid_base + uint64_t((hi.z >> 21) | (hi.w << 11));
      } else {
        o.a3 = o.a4 = o.b3 = o.b4 = 0u;
        o.b1 = s2; o.b2 = s3; o.p = s4; o.c = s5;
        o.t = (q.y >> 31) | ((q.w >> 31) << 1) | (((hi.y >> 8) & 1u) << 2);
        o.gid = gid_base + (uint64_t(hi.x) | (uint64_t(hi.y & 0xFFu) << 32));
      }
      return o;
    };
    // One AND-family gate spread over LPG lanes (garble: two AES blocks x 4 columns, evaluate: one block x 4 columns).
    // Called with LPG-aligned groups of active lanes; `q` is the gate's record (same in all lanes of the group), `cti`
    // its index in the program's AND array.
    auto and_multilane = [&](const Rec& q, const u32x4& q_hi, uint32_t cti) {
      const AndOp o = decode_and(q, q_hi);
      const uint32_t t = o.t;
      uint32_t a_c = dq, b_c = dq, p_c = 0;
      GSV_PC_STAMP(1, o.c ^ o.t);  // the record's second half has arrived and is decoded
      if (!no_load) {
        a_c = wf.ld_word(o.a1, col) ^ wf.ld_word(o.a2, col);
        b_c = wf.ld_word(o.b1, col) ^ wf.ld_word(o.b2, col);
        p_c = wf.ld_word(o.p, col);
        if (four_wire) {
          a_c ^= wf.ld_word(o.a3, col) ^ wf.ld_word(o.a4, col);
          b_c ^= wf.ld_word(o.b3, col) ^ wf.ld_word(o.b4, col);
        }
      }
      const uint32_t twc = tweak_word(o.gid, col);
      if (!EVAL) {
        uint32_t x = a_c ^ (alpha_a(t) ? dq : 0u);  // selected_a ; lanes of the second quad take other_a
        if (blk) x ^= dq;
        GSV_PC_STAMP(2, x ^ b_c ^ p_c ^ twc);  // the operands have arrived
        const uint32_t h = no_aes ? (x ^ twc) : aes128_quad(aes, rkc, x ^ twc);
        GSV_PC_STAMP(3, h);                    // AES done
        const uint32_t o2 = uint32_t(__shfl_xor(int(h), 4));  // the other block's column c
        const uint32_t c0_c = h ^ (alpha_c(t) ? dq : 0u) ^ p_c;
        const uint32_t ct_c = h ^ o2 ^ b_c ^ (alpha_b(t) ? dq : 0u);
        if (!blk && !no_store) wf.st_word(o.c, col, c0_c);
        asm volatile("" ::: "memory");
        if (!blk && !no_store) __builtin_nontemporal_store(ct_c, &CTw[(ct_base + cti) * 4u + col]);
        GSV_PC_STAMP(4, ct_c);                 // stores issued
      } else {
        uint32_t va = wf.ld_bit(o.a1) ^ wf.ld_bit(o.a2), vb = wf.ld_bit(o.b1) ^ wf.ld_bit(o.b2);
        if (four_wire) { va ^= wf.ld_bit(o.a3) ^ wf.ld_bit(o.a4); vb ^= wf.ld_bit(o.b3) ^ wf.ld_bit(o.b4); }
        va &= 1u; vb &= 1u;
        const uint32_t vp = wf.ld_bit(o.p) & 1u;
        const uint32_t ct_c = CTw[(ct_base + cti) * 4u + col];
        const uint32_t h = no_aes ? (a_c ^ twc) : aes128_quad(aes, rkc, a_c ^ twc);
        const uint32_t use_ct = (va ^ alpha_a(t)) & 1u;
        wf.st_word(o.c, col, h ^ (use_ct ? (ct_c ^ b_c) : 0u) ^ p_c);
        if (col == 0) wf.st_bit(o.c, (gate_eval_bit(t, va, vb) ^ vp) & 1u);
      }
    };
    // The same with both blocks in one quad (garbling, four lanes per gate: aes128_quad_x2).
    auto and_multilane_x2 = [&](const Rec& q, const u32x4& q_hi, uint32_t cti) {
      const AndOp o = decode_and(q, q_hi);
      const uint32_t t = o.t;
      uint32_t a_c = dq, b_c = dq, p_c = 0;
      if (!no_load) {
        a_c = wf.ld_word(o.a1, col) ^ wf.ld_word(o.a2, col);
        b_c = wf.ld_word(o.b1, col) ^ wf.ld_word(o.b2, col);
        p_c = wf.ld_word(o.p, col);
        if (four_wire) {
          a_c ^= wf.ld_word(o.a3, col) ^ wf.ld_word(o.a4, col);
          b_c ^= wf.ld_word(o.b3, col) ^ wf.ld_word(o.b4, col);
        }
      }
      const uint32_t twc = tweak_word(o.gid, col);
      uint32_t h = a_c ^ (alpha_a(t) ? dq : 0u) ^ twc;  // selected_a
      uint32_t o2 = h ^ dq;                              // other_a
      if (!no_aes) aes128_quad_x2(aes, rkc, h, o2);
      const uint32_t c0_c = h ^ (alpha_c(t) ? dq : 0u) ^ p_c;
      const uint32_t ct_c = h ^ o2 ^ b_c ^ (alpha_b(t) ? dq : 0u);
      if (!no_store) wf.st_word(o.c, col, c0_c);
      asm volatile("" ::: "memory");
      if (!no_store) __builtin_nontemporal_store(ct_c, &CTw[(ct_base + cti) * 4u + col]);
    };
    // decoded free-gate record (u32x4 = w0 | w1)
    struct XorOp { uint32_t x1, x2, x3, x4, c, par; };
    auto decode_xor = [&](const u32x4& r) -> XorOp {
      XorOp o;
      o.x1 = r.x & GSV_SLOT_MASK;
      o.x2 = ((r.x >> 21) | (r.y << 11)) & GSV_SLOT_MASK;
      o.x3 = (r.y >> 10) & GSV_SLOT_MASK;
      o.par = r.y >> 31;
      o.x4 = r.z & GSV_SLOT_MASK;
      o.c = ((r.z >> 21) | (r.w << 11)) & GSV_SLOT_MASK;
      return o;
    };
    // One step.  sd: its descriptor; r0: this lane's first record of it — consumed here and refilled, as the wave's youngest
    // vector-memory operation, with the record of step s+2 (descriptor n2sd, a scalar load issued at the top of this step).
    auto run_step = [&](const uint32_t s, const u32x4& sd, Rec& r0, const u32x4& n2sd) __attribute__((always_inline)) {
      const uint32_t and_off = sd.x, and_cnt = sd.y, xor_off = sd.z, total = sd.y + sd.w;
      (void)total;
#ifdef GSV_DIAG_BUILD
      if (!phase_clock)
#endif
      if (ka.step_clock && blockIdx.x == 0 && threadIdx.x == 0 && rep + 1 == ka.replays) ka.step_clock[s] = wall_clock64();  // older than this step's stores
      if (is_narrow(sd)) {
        // ------------------------------------------------------------------ narrow step: one pass
        const uint32_t nl = narrow_lpg(sd);
        const uint32_t na = and_cnt * nl, x0 = xor_lane0(na);
        if (tid < na) {
          if (!DUAL || nl == LPG) and_multilane(r0, load_and_hi(and_off + tid / LPG), and_off + tid / LPG);
          else and_multilane_x2(r0, load_and_hi(and_off + tid / LPG2), and_off + tid / LPG2);
        } else if (tid >= x0 && tid < x0 + sd.w) {
          const XorOp o = decode_xor(r0);
          Label c0 = delta;
          if (!no_load) c0 = lxor(lxor(wf.ld(o.x1), wf.ld(o.x2)), lxor(wf.ld(o.x3), wf.ld(o.x4)));
          if (!EVAL) c0 = lxor_if(c0, delta, o.par);
          if (!no_store || c0.w[0] == 0x12345678u) {
            wf.st(o.c, c0);
            if (EVAL) wf.st_bit(o.c, (wf.ld_bit(o.x1) ^ wf.ld_bit(o.x2) ^ wf.ld_bit(o.x3) ^ wf.ld_bit(o.x4) ^ o.par) & 1u);
          }
        }
      } else {
        // ------------------------------------------------------------------ wide step
        // AND-family gates: passes of 1024 gates, one per lane (two interleaved AES blocks each).
        const uint32_t xor_cnt = sd.w;
        const glb_u128* const xq = (const glb_u128*)ka.xors + xor_off;
        // The free-gate phase is a software pipeline over batches of XB*BT gates: while batch k is XORed and stored,
        // the operand loads of batch k+1 and the records of batch k+2 are in flight, so a batch exposes ONE memory
        // latency (not record -> operands -> store back to back).  The records of batch 0 are issued BEFORE the AES
        // passes and land behind them.
        constexpr int XB = 2;
        const uint32_t xl = tid;
        u32x4 xr[XB], xrn[XB];
        Label xa[XB];
        uint32_t xv[XB];  // evaluate: XOR of the operands' plaintext bits
        auto load_xor_recs = [&](uint32_t base, u32x4 (&r)[XB]) {
#pragma unroll
          for (int j = 0; j < XB; ++j) {
            const uint32_t g = base + uint32_t(j) * BT + xl;
            r[j] = u32x4{0, 0, 0, 0};
            if (g < xor_cnt) r[j] = xq[g];
          }
        };
        auto issue_xor_operands = [&](uint32_t base) {
#pragma unroll
          for (int j = 0; j < XB; ++j) {
            const XorOp o = decode_xor(xr[j]);
            xa[j] = delta; xv[j] = 0;
            if (!no_load && base + uint32_t(j) * BT + xl < xor_cnt) {
              xa[j] = lxor(lxor(wf.ld(o.x1), wf.ld(o.x2)), lxor(wf.ld(o.x3), wf.ld(o.x4)));
              if (EVAL) xv[j] = wf.ld_bit(o.x1) ^ wf.ld_bit(o.x2) ^ wf.ld_bit(o.x3) ^ wf.ld_bit(o.x4);
            }
          }
        };
        auto finish_xor_batch = [&](uint32_t base) {
#pragma unroll
          for (int j = 0; j < XB; ++j) {
            if (base + uint32_t(j) * BT + xl < xor_cnt) {
              const XorOp o = decode_xor(xr[j]);
              Label c0 = xa[j];
              if (!EVAL) c0 = lxor_if(c0, delta, o.par);
              if (!no_store || c0.w[0] == 0x12345678u) {
                wf.st(o.c, c0);
                if (EVAL) wf.st_bit(o.c, (xv[j] ^ o.par) & 1u);
              }
            }
          }
        };
        load_xor_recs(0, xr);
        // ---- AND-family gates: whole passes of BT gates in the one-gate-per-lane form (two interleaved AES blocks
        // per lane), then the remainder in the LPG-lanes-per-gate form: a partly filled one-gate-per-lane pass would
        // cost the full ~5 us AES latency for a handful of waves, the multi-lane form ~1 us per BT/LPG gates.
        // (remainders larger than two multi-lane passes are cheaper as one partly filled one-per-lane pass)
        const uint32_t and_rem = and_cnt % BT;
        const uint32_t and_full = small_rem(and_cnt) ? and_cnt - and_rem : and_cnt;
        Rec qnext = r0;
        uint32_t pass_idx = 0;
        for (uint32_t i = tid; i < and_full; i += BT) {
          // Issue priority falls with the pass index: a wave that is AHEAD (the instruction arbiter prefers the oldest wave of a SIMD, i.e.
          // the first instance group's) yields to the waves that are still in an earlier pass, so that the groups move through the step's
          // passes together and no group is left to run its last pass alone, latency-bound, while the others wait at the barrier.
          if (pass_idx == 0) __builtin_amdgcn_s_setprio(3); else if (pass_idx == 1) __builtin_amdgcn_s_setprio(2); else if (pass_idx == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
          ++pass_idx;
          LdsBankedTable aes_pass = aes;
          asm volatile("" : "+s"(aes_pass.rkp));
          const Rec q = qnext;
          if (i + BT < and_full) qnext = load_and_rec(and_off + i + BT);
          const uint32_t cti = and_off + i;
          const AndOp o = decode_and(q, load_and_hi(cti));
          const uint32_t t = o.t;
          Label a = delta, b = delta, pl{{0, 0, 0, 0}};
          if (!no_load) {
            a = lxor(wf.ld(o.a1), wf.ld(o.a2));
            b = lxor(wf.ld(o.b1), wf.ld(o.b2));
            pl = wf.ld(o.p);
            if (four_wire) {
              a = lxor(a, lxor(wf.ld(o.a3), wf.ld(o.a4)));
              b = lxor(b, lxor(wf.ld(o.b3), wf.ld(o.b4)));
            }
          }
          Label c0, ct{{0, 0, 0, 0}};
          uint32_t vc = 0;
          if (!EVAL) {
            if (no_aes) { c0 = lxor(a, b); ct = lxor(a, tweak_of(o.gid)); }
            else if (HASH == 1) garble_and_blake3(t, a, b, delta, o.gid, c0, ct);
            else garble_and(aes_pass, t, a, b, delta, o.gid, c0, ct);
          } else {
            uint32_t va = wf.ld_bit(o.a1) ^ wf.ld_bit(o.a2), vb = wf.ld_bit(o.b1) ^ wf.ld_bit(o.b2);
            if (four_wire) { va ^= wf.ld_bit(o.a3) ^ wf.ld_bit(o.a4); vb ^= wf.ld_bit(o.b3) ^ wf.ld_bit(o.b4); }
            va &= 1u; vb &= 1u;
            const uint32_t vp = wf.ld_bit(o.p) & 1u;
            const u32x4 cv = CT[ct_base + cti];
            if (HASH == 1) c0 = degarble_and_blake3(t, Label{{cv.x, cv.y, cv.z, cv.w}}, a, va, b, o.gid);
            else c0 = degarble_and(aes_pass, t, Label{{cv.x, cv.y, cv.z, cv.w}}, a, va, b, o.gid);
            vc = (gate_eval_bit(t, va, vb) ^ vp) & 1u;
          }
          c0 = lxor(c0, pl);
          if (!no_store || c0.w[0] == 0x12345678u) {
            wf.st(o.c, c0);
            if (EVAL) wf.st_bit(o.c, vc);
          }
          if (!EVAL && !no_store) __builtin_nontemporal_store(u32x4{ct.w[0], ct.w[1], ct.w[2], ct.w[3]}, &CT[ct_base + cti]);
        }
        __builtin_amdgcn_s_setprio(0);  // remainder, free gates, barrier: behind every wave that is still in a whole pass
        if (!DUAL || rem_lpg(and_cnt) != LPG2) {
          for (uint32_t g = and_full + tid / LPG; g < and_cnt; g += BT / LPG) {
            // the first remainder record was prefetched two steps ago when the step has no whole pass
            const Rec q = (and_full == 0 && g == tid / LPG) ? r0 : load_and_rec(and_off + g);
            and_multilane(q, load_and_hi(and_off + g), and_off + g);
          }
        } else {
          for (uint32_t g = and_full + tid / LPG2; g < and_cnt; g += BT / LPG2) {
            const Rec q = (and_full == 0 && g == tid / LPG2) ? r0 : load_and_rec(and_off + g);
            and_multilane_x2(q, load_and_hi(and_off + g), and_off + g);
          }
        }
        // ---- free-gate batches (their label stores are the wave's youngest stores: no young ciphertext store)
        if (xor_cnt) { issue_xor_operands(0); load_xor_recs(uint32_t(XB) * BT, xrn); }
        for (uint32_t base = 0; base < xor_cnt; base += XB * BT) {
          finish_xor_batch(base);
          const uint32_t nb = base + XB * BT;
#pragma unroll
          for (int j = 0; j < XB; ++j) xr[j] = xrn[j];
          if (nb < xor_cnt) {
            issue_xor_operands(nb);
            load_xor_recs(nb + XB * BT, xrn);
          }
        }
      }
      // keep r0's registers reserved through the step: were they handed to a store's data in between, the refill below would
      // have to wait for that store (vmcnt(0) in front of the prefetch) before it could overwrite them
      asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w)::"memory");
      if (!no_refill) r0 = load_rec(n2sd);  // stays in flight across the barrier and the whole next step
      GSV_PC_STAMP(5, and_cnt);  // the refill has been issued (ordered by the asm's memory clobber; no dependency on its data)
      if (no_barrier) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else if (group_barrier) {
        // arrive (lane 0 adds one to the group's counter) once this wave's LDS stores are done — its wire-file stores are issued, and the
        // group's waves share the CU's L1 as above — then poll until all of the group's waves have arrived.  Inline assembly over three
        // VGPRs that nothing else touches: written in C++ the counter's address and value land in registers that the step's stores have just
        // used as sources, and the compiler waits for those stores' acknowledgements (s_waitcnt vmcnt) in front of every barrier.
        bar_target += BT / 64u;
        uint64_t sv;
        uint32_t tmp;
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "ds_add_u32 %[addr], %[one]\n\t"
                     "s_mov_b64 exec, %[sv]\n"
                     ".Lgsv_gbar_poll%=:\n\t"
                     "ds_read_b32 %[val], %[addr]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %[tmp], %[val]\n\t"
                     "s_sub_i32 %[tmp], %[tmp], %[target]\n\t"
                     "s_cmp_lt_i32 %[tmp], 0\n\t"
                     "s_cbranch_scc0 .Lgsv_gbar_done%=\n\t"
                     "s_sleep 2\n\t"
                     "s_branch .Lgsv_gbar_poll%=\n"
                     ".Lgsv_gbar_done%=:"
                     : [val] "+v"(bar_val), [sv] "=&s"(sv), [tmp] "=&s"(tmp)
                     : [addr] "v"(bar_addr), [one] "v"(bar_one), [target] "s"(bar_target)
                     : "memory", "scc");
      }
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef GSV_DIAG_BUILD
      if (phase_clock) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pc_t[6]) : : "memory");
        if (is_narrow(sd) && and_cnt != 0 && pc_prev_end) {  // wave 0 garbled gate 0 in this step and in the one before: every stamp was taken
          pc_t[0] = pc_prev_end;  // interval 0 starts behind the previous step's barrier: loop top, descriptor, branch, second-half load, decode
          for (int i = 0; i < 6; ++i) pc_acc[i] += pc_t[i + 1] - pc_t[i];
          pc_acc[7] += 1;
        }
        pc_prev_end = is_narrow(sd) && and_cnt != 0 ? pc_t[6] : 0;
      }
#endif
    };
    for (uint32_t s = 0; s < ka.n_steps; s += 2) {
      const u32x4 sdA2 = load_desc(s + 2);  // lands during the step; the record load that needs it is issued at the step's end
      run_step(s, sdA, recA, sdA2);
      sdA = sdA2;
      if (s + 1 >= ka.n_steps) break;        // wave-uniform: every wave passes the same number of barriers
      const u32x4 sdB2 = load_desc(s + 3);
      run_step(s + 1, sdB, recB, sdB2);
      sdB = sdB2;
    }
    __syncthreads();
    if (ka.step_clock && blockIdx.x == 0 && threadIdx.x == 0 && rep + 1 == ka.replays) ka.step_clock[ka.n_steps] = wall_clock64();
    // replay epilogue: feedback copies through staging slots (sources may alias destinations; all in HBM)
    if (ka.n_fb) {
      const uint32_t nfb = inst_active ? ka.n_fb : 0u;
      for (uint32_t i = tid; i < nfb; i += BT) {
        const u32x4 v = wf.hbm[ka.fb_src[i]];
        wf.hbm[ka.fb_stage_base + i] = v;
        if (EVAL) { const uint8_t bv = wf.hbm_bits[ka.fb_src[i]]; wf.hbm_bits[ka.fb_stage_base + i] = bv; }
      }
      __syncthreads();
      for (uint32_t i = tid; i < nfb; i += BT) {
        const u32x4 v = wf.hbm[ka.fb_stage_base + i];
        wf.hbm[ka.fb_dst[i]] = v;
        if (EVAL) { const uint8_t bv = wf.hbm_bits[ka.fb_stage_base + i]; wf.hbm_bits[ka.fb_dst[i]] = bv; }
      }
      __syncthreads();
    }
  }
#ifdef GSV_DIAG_BUILD
  if (phase_clock && threadIdx.x == 0) for (int i = 0; i < 8; ++i) ka.step_clock[i] = pc_acc[i];
#endif
  if (cd) {
    // ---- dataflow epilogue: outputs -> global wires, then publish completion for this instance group (agent-scope release: the
    // consumer may run on another XCD)
    const uint32_t n_post = inst_active ? cd->n_post : 0u;
    glb_u128* const gw = (glb_u128*)(ka.W + size_t(inst) * ka.n_slots);
    for (uint32_t i = tid; i < n_post; i += BT) {
      const uint32_t sidx = ka.copy_src[cd->post_off + i], didx = ka.copy_dst[cd->post_off + i];
      gw[didx] = gw[sidx];
      if (EVAL) { glb_u8* const vb = (glb_u8*)(ka.VB + size_t(inst) * ka.n_slots); vb[didx] = vb[sidx]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t* const fl = ka.flags + size_t(blockIdx.x) * ka.flag_stride;
      __hip_atomic_store(fl + blockIdx.y, ka.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(fl + (ka.flag_stride - 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the group's progress counter (watchdog)
      // the host's view of the running window (drain segments, ring positions): one more finished workgroup of this call
      if (cd->done_host) __hip_atomic_fetch_add(cd->done_host, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// outputs[inst][i] = W[inst][slots[i]]  (and the plaintext bit in evaluate mode)
__global__ void gather_outputs_kernel(const uint4* W, const uint8_t* VB, uint32_t n_slots, const uint32_t* slots, uint32_t n_out,
                                      uint4* out, uint8_t* out_bits) {
  const uint32_t inst = blockIdx.y;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const size_t src = size_t(inst) * n_slots + slots[i];
  out[size_t(inst) * n_out + i] = W[src];
  if (out_bits) out_bits[size_t(inst) * n_out + i] = VB[src];
}

// evaluate-mode input staging: VB[inst][first_slot + i] = bits[inst][i]
__global__ void scatter_bits_kernel(uint8_t* VB, uint32_t n_slots, uint32_t first_slot, const uint8_t* bits, uint32_t n) {
  const uint32_t inst = blockIdx.y;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  VB[size_t(inst) * n_slots + first_slot + i] = bits[size_t(inst) * n + i];
}

__global__ void copy_slots_kernel(uint4* W, uint8_t* VB, uint32_t n_slots, const uint32_t* src, const uint32_t* dst, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t base = size_t(blockIdx.y) * n_slots;
  W[base + dst[i]] = W[base + src[i]];
  if (VB) VB[base + dst[i]] = VB[base + src[i]];
}

// Gate order <-> program order of the ciphertext stream (see program.hpp, ct_pos): one record per thread.
__global__ void permute_ciphertexts_kernel(uint4* stream, const uint32_t* ct_pos, uint64_t n_ct, uint64_t first, uint64_t n, uint4* stage, int scatter) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t idx = first + i, rep = idx / n_ct, g = idx - rep * n_ct;
  const uint64_t pos = rep * n_ct + ct_pos[g];
  if (scatter) stream[pos] = stage[i]; else stage[i] = stream[pos];
}

// Whole-segment form for the streaming drain: for every instance, `n_rep` replays starting at ring slot 0 of the
// program-order ring go to a gate-order buffer: out[inst][r * n_ct + g] = ring[inst][r * n_ct + ct_pos[g]].
// scatter != 0 is the evaluator's direction: gate-order records (read from gc_<i>.bin) go to their program-order positions.
__global__ void gather_segment_kernel(uint4* ring, uint64_t ring_stride, const uint32_t* ct_pos, uint64_t n_ct, uint32_t n_rep, uint4* out, uint64_t out_stride, int scatter) {
  // The source may have been written by workgroups of a garbling launch that is STILL RUNNING, on other XCDs, into addresses this XCD
  // has read before (a ciphertext ring reuses its blocks lap after lap).  Their epilogue released the records at agent scope before the
  // completion counter the host saw, and this launch's own dispatch acquires at agent scope: no fence here (one per wave cost 5 s of
  // L2 invalidations over a 16-instance pass and slowed the garbling beside it).
  const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_ct) return;
  const uint32_t pos = ct_pos[g];
  uint4* src = ring + uint64_t(blockIdx.y) * ring_stride;
  uint4* dst = out + uint64_t(blockIdx.y) * out_stride;
  for (uint32_t r = 0; r < n_rep; ++r) {
    if (scatter) src[uint64_t(r) * n_ct + pos] = dst[uint64_t(r) * n_ct + g];
    else dst[uint64_t(r) * n_ct + g] = src[uint64_t(r) * n_ct + pos];
  }
}

// Do two streams run side by side?  (engine.cpp, streams_overlap: the runtime multiplexes streams onto a few hardware queues, in order
// within a queue.)  One thread waits — bounded — for a word the other kernel, launched on the other stream AFTER it, sets:
// result 1 = saw it (the streams overlap), 2 = gave up (the second launch sat behind this one).
__global__ void probe_wait_kernel(uint32_t* word, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  uint32_t seen = 0;
  while (!(seen = __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  word[1] = seen ? 1u : 2u;
}
__global__ void probe_set_kernel(uint32_t* word) { __hip_atomic_store(word, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace dev
}  // namespace gsv

extern "C" {

int gsvk_upload_round_keys(const uint32_t rk[44]) {
  uint32_t dev_rk[44];  // LdsBankedTable::kPairedRotation: the nine middle rounds' keys rotated right by one byte
  for (int i = 0; i < 44; ++i) dev_rk[i] = (i >= 4 && i < 40) ? ((rk[i] >> 8) | (rk[i] << 24)) : rk[i];
  return int(hipMemcpyToSymbol(HIP_SYMBOL(gsv::dev::c_rk), dev_rk, 44 * sizeof(uint32_t)));
}
int gsvk_launch_program(const gsv::dev::KernelArgs* ka, uint32_t n_instances, int evaluate, hipStream_t stream) { return gsvk_launch_batch(ka, n_instances, 1, evaluate, stream); }
int gsvk_launch_batch(const gsv::dev::KernelArgs* ka, uint32_t n_instances, uint32_t n_calls, int evaluate, hipStream_t stream) {
  if (n_calls == 0 || n_calls > 65535u || (n_calls > 1 && !ka->calls) || (ka->calls && (!ka->flags || !ka->error))) return int(hipErrorInvalidValue);
  const size_t lds = GSV_LDS_BYTES;
  // The opt-in to 160 KiB of dynamic LDS is a per-device function attribute: done once per device (an engine per GPU may live
  // in one process, and sessions may be driven from several host threads).
  {
    static std::mutex attr_mu;
    static std::set<int> attr_done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return int(hipErrorInvalidDevice);
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_done.count(dev)) {
      // the kernels address LDS from byte 0: there must be no static LDS in front of the dynamic block
      hipFuncAttributes fa;
#define GSV_K(E, N, H, F) reinterpret_cast<const void*>(gsv::dev::run_program_kernel<E, N, H, F>)
      const void* kernels[16] = {GSV_K(false, 1, 0, false), GSV_K(true, 1, 0, false), GSV_K(false, 2, 0, false), GSV_K(true, 2, 0, false), GSV_K(false, 4, 0, false), GSV_K(true, 4, 0, false),
                                 GSV_K(false, 1, 1, false), GSV_K(true, 1, 1, false), GSV_K(false, 1, 0, true),  GSV_K(true, 1, 0, true),  GSV_K(false, 2, 0, true),  GSV_K(true, 2, 0, true),
                                 GSV_K(false, 4, 0, true),  GSV_K(true, 4, 0, true),  GSV_K(false, 1, 1, true),  GSV_K(true, 1, 1, true)};
#undef GSV_K
      for (const void* k : kernels) {
        if (hipFuncGetAttributes(&fa, k) != hipSuccess || fa.sharedSizeBytes != 0) return int(hipErrorInvalidValue);
        hipError_t e0 = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e0 != hipSuccess) return int(e0);
      }
      attr_done.insert(dev);
    }
  }
  const bool blake3 = ka->hasher == 1;
  const uint32_t ni = blake3 ? 1u : (ka->instances_per_wg == 4 ? 4u : ka->instances_per_wg == 2 ? 2u : 1u);
  const dim3 grid((n_instances + ni - 1) / ni, n_calls);
  const bool fw = ka->any_four_wire != 0;
#define GSV_LAUNCH(E, N, H, F) hipLaunchKernelGGL((gsv::dev::run_program_kernel<E, N, H, F>), grid, dim3(GSV_BLOCK_THREADS), lds, stream, *ka)
#define GSV_LAUNCH_EF(N, H) do { if (evaluate) { if (fw) GSV_LAUNCH(true, N, H, true); else GSV_LAUNCH(true, N, H, false); } \
                                 else { if (fw) GSV_LAUNCH(false, N, H, true); else GSV_LAUNCH(false, N, H, false); } } while (0)
  if (blake3) GSV_LAUNCH_EF(1, 1);
  else if (ni == 4) GSV_LAUNCH_EF(4, 0);
  else if (ni == 2) GSV_LAUNCH_EF(2, 0);
  else GSV_LAUNCH_EF(1, 0);
#undef GSV_LAUNCH_EF
#undef GSV_LAUNCH
  return int(hipGetLastError());
}
int gsvk_gather_outputs(const void* W, const void* VB, uint32_t n_slots, const uint32_t* slots, uint32_t n_out, uint32_t n_instances,
                        void* out, void* out_bits, hipStream_t stream) {
  dim3 grid((n_out + 255) / 256, n_instances);
  hipLaunchKernelGGL(gsv::dev::gather_outputs_kernel, grid, dim3(256), 0, stream, static_cast<const uint4*>(W), static_cast<const uint8_t*>(VB),
                     n_slots, slots, n_out, static_cast<uint4*>(out), static_cast<uint8_t*>(out_bits));
  return int(hipGetLastError());
}
int gsvk_permute_ciphertexts(void* stream, const void* ct_pos, uint64_t n_ct, uint64_t first, uint64_t n, void* stage, int scatter, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gsv::dev::permute_ciphertexts_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, s, static_cast<uint4*>(stream),
                     static_cast<const uint32_t*>(ct_pos), n_ct, first, n, static_cast<uint4*>(stage), scatter);
  return int(hipGetLastError());
}
int gsvk_gather_segment(void* ring, uint64_t ring_stride, const void* ct_pos, uint64_t n_ct, uint32_t n_rep, uint32_t n_instances, void* out,
                        uint64_t out_stride, int scatter, hipStream_t s) {
  if (n_ct == 0 || n_rep == 0) return 0;
  hipLaunchKernelGGL(gsv::dev::gather_segment_kernel, dim3(uint32_t((n_ct + 255) / 256), n_instances), dim3(256), 0, s, static_cast<uint4*>(ring), ring_stride,
                     static_cast<const uint32_t*>(ct_pos), n_ct, n_rep, static_cast<uint4*>(out), out_stride, scatter);
  return int(hipGetLastError());
}
int gsvk_copy_slots(void* W, void* VB, uint32_t n_slots, const uint32_t* src, const uint32_t* dst, uint32_t n, uint32_t n_instances, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gsv::dev::copy_slots_kernel, dim3((n + 255) / 256, n_instances), dim3(256), 0, s, static_cast<uint4*>(W), static_cast<uint8_t*>(VB), n_slots, src, dst, n);
  return int(hipGetLastError());
}
int gsvk_scatter_bits(void* VB, uint32_t n_slots, uint32_t first_slot, const void* bits, uint32_t n, uint32_t n_instances, hipStream_t stream) {
  dim3 grid((n + 255) / 256, n_instances);
  hipLaunchKernelGGL(gsv::dev::scatter_bits_kernel, grid, dim3(256), 0, stream, static_cast<uint8_t*>(VB), n_slots, first_slot,
                     static_cast<const uint8_t*>(bits), n);
  return int(hipGetLastError());
}

int gsvk_probe_overlap(void* word, unsigned long long ticks, hipStream_t first, hipStream_t second) {
  hipLaunchKernelGGL(gsv::dev::probe_wait_kernel, dim3(1), dim3(1), 0, first, static_cast<uint32_t*>(word), ticks);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(gsv::dev::probe_set_kernel, dim3(1), dim3(1), 0, second, static_cast<uint32_t*>(word));
  return int(hipGetLastError());
}

}  // extern "C"
