// gfx950 (MI355X / CDNA4) kernels of the garbling engine.
//
// Execution model: one 1024-thread workgroup = one garbling instance on one CU.
// The circuit is deep and narrow (Fq12 mul: ~20.3 M gates over ~10^4 dependent AND levels, mean
// width a few hundred gates), so a step is latency-bound, not throughput-bound: the workgroup walks
// the program's steps with `s_barrier` between them — no grid-wide synchronisation, no inter-CU
// traffic — and independent cut-and-choose instances fill the other CUs.
//   * two-level wire file: short-lived labels (most of them) in a 120 KiB LDS window chosen by the
//     compiler's fan-out/lifetime pass, long-lived ones in W[instance][slot] in HBM (128-bit accesses,
//     next-fit slot order so that a step's stores coalesce)
//   * AES table Te0: 32 KiB in LDS, bank-replicated so lookups never conflict; round keys scalar
//   * within a step the AND-family records and the free-gate records are two contiguous runs, so
//     only one wave per step diverges on the gate kind (no per-lane ballot/compaction needed);
//     records of the NEXT step are prefetched into registers before the barrier
//   * ciphertexts are written straight to the instance's stream at their gate-order index
// No MFMA: the work is byte-table lookups and 128-bit XORs.
#include <hip/hip_runtime.h>

#include "gate_math.hpp"
#include "kernel_api.h"

namespace gsv {
namespace dev {

__constant__ uint32_t c_rk[44];

// Native 128-bit / 64-bit vectors (copyable across address spaces, one dwordx4 / dwordx2 access each).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// LDS map of a workgroup (163,328 of the CU's 163,840 bytes):
//   [0, 32 KiB)            AES table Te0, every entry replicated once per LDS bank: dword x*32 + (lane & 31).
//                          A wave's ds_read_b32 is served in two 32-lane groups (MI355X_MICROARCH.md §LDS);
//                          inside a group lane j only touches bank j, so 64 data-dependent lookups cost the
//                          conflict-free 2 cycles (SQ_LDS_BANK_CONFLICT = 0 in profiles/).  Te1..Te3 are byte
//                          rotations of Te0 (one v_alignbit each).
//   [32 KiB, +120 KiB)     label window: GSV_LDS_SLOTS x 16 B, the short-lived wires chosen by the compiler
//   [.., +7.5 KiB)         plaintext bits of window wires (evaluate mode)
#define GSV_LDS_SLOTS 7680u
#define GSV_LDS_TABLE_BYTES 32768u
#define GSV_LDS_BYTES (GSV_LDS_TABLE_BYTES + GSV_LDS_SLOTS * 16u + GSV_LDS_SLOTS)
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

struct LdsBankedTable {
  uint32_t lane4;  // (lane & 31) * 4; the table starts at LDS byte 0
  template <int K, int BYTE>
  __device__ __forceinline__ uint32_t lk(uint32_t s) const {
    // byte BYTE of s, times 128 (32 banks x 4 B), as bits 7..14
    const uint32_t x = BYTE == 0 ? (s << 7) : BYTE == 1 ? (s >> 1) : BYTE == 2 ? (s >> 9) : (s >> 17);
    const uint32_t v = *reinterpret_cast<const lds_u32*>(uintptr_t((x & 0x7f80u) | lane4));
    return K == 0 ? v : __builtin_amdgcn_alignbit(v, v, 32 - 8 * K);  // rotl(v, 8K): Te_K from Te0
  }
};

struct WireFile {
  glb_u128* hbm;      // this instance's wire file
  glb_u8* hbm_bits;   // evaluate: plaintext bits
  __device__ __forceinline__ static lds_u128* win(uint32_t idx) { return reinterpret_cast<lds_u128*>(uintptr_t(GSV_LDS_TABLE_BYTES + idx * 16u)); }
  __device__ __forceinline__ static lds_u8* win_bit(uint32_t idx) { return reinterpret_cast<lds_u8*>(uintptr_t(GSV_LDS_TABLE_BYTES + GSV_LDS_SLOTS * 16u + idx)); }
  __device__ __forceinline__ Label ld(uint32_t slot) const {
    u32x4 v;
    if (slot & GSV_SLOT_LDS_FLAG) v = *win(slot & GSV_SLOT_INDEX_MASK); else v = hbm[slot];
    return Label{{v.x, v.y, v.z, v.w}};
  }
  __device__ __forceinline__ void st(uint32_t slot, const Label& l) const {
    const u32x4 v = {l.w[0], l.w[1], l.w[2], l.w[3]};
    if (slot & GSV_SLOT_LDS_FLAG) *win(slot & GSV_SLOT_INDEX_MASK) = v; else hbm[slot] = v;
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

template <bool EVAL>
__global__ __launch_bounds__(GSV_BLOCK_THREADS) void run_program_kernel(KernelArgs ka) {
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  const uint32_t tid = threadIdx.x;
  (void)s_mem;  // the dynamic LDS block starts at LDS address 0 (no static __shared__ in this kernel)
  for (uint32_t i = tid; i < GSV_LDS_TABLE_BYTES / 4; i += GSV_BLOCK_THREADS) *reinterpret_cast<lds_u32*>(uintptr_t(i * 4u)) = ka.te[i >> 5];
  __syncthreads();
  const LdsBankedTable aes{(tid & 31u) * 4u};
  const uint32_t* rk = c_rk;

  const uint32_t inst = blockIdx.x;
  WireFile wf;
  wf.hbm = (glb_u128*)(ka.W + size_t(inst) * ka.n_slots);      // C-style cast: generic -> global address space
  wf.hbm_bits = (glb_u8*)(ka.VB + size_t(inst) * ka.n_slots);
  glb_u128* __restrict__ CT = (glb_u128*)(ka.CT + size_t(inst) * ka.ct_stride);
  Label delta{{0, 0, 0, 0}};
  if (!EVAL) { const u32x4 d = ((const glb_u128*)ka.delta)[inst]; delta = Label{{d.x, d.y, d.z, d.w}}; }
  const glb_u128* __restrict__ and_q = (const glb_u128*)ka.ands;  // 16 B records
  const glb_u64* __restrict__ xor_q = (const glb_u64*)ka.xors;    // 8 B records
  const glb_u128* __restrict__ step_q = (const glb_u128*)ka.steps;

  for (uint32_t rep = 0; rep < ka.replays; ++rep) {
    const uint64_t gid_base = ka.gid_base + uint64_t(rep) * ka.n_gates;
    const uint64_t ct_base = uint64_t(rep % ka.ct_cap_replays) * ka.n_ct;

    // step descriptor {and_off, and_cnt, xor_off, xor_cnt}: lanes [0, and_cnt) take AND-family gates,
    // lanes [and_cnt, and_cnt + xor_cnt) take free gates, so only the boundary wave diverges.
    u32x4 sd = step_q[0];
    // prefetched record for this thread's first gate of the step
    u32x4 r0 = {0, 0, 0, 0};
    if (tid < sd.y) r0 = and_q[size_t(sd.x + tid)];
    else if (tid < sd.y + sd.w) { const u32x2 x = xor_q[size_t(sd.z + (tid - sd.y))]; r0.x = x.x; r0.y = x.y; }
    for (uint32_t s = 0; s < ka.n_steps; ++s) {
      const uint32_t and_off = sd.x, and_cnt = sd.y, xor_off = sd.z, total = sd.y + sd.w;
      // issue next step's descriptor + record loads early; they complete while this step computes
      u32x4 nsd = sd, n0 = r0;
      if (s + 1 < ka.n_steps) {
        nsd = step_q[s + 1];
        if (tid < nsd.y) n0 = and_q[size_t(nsd.x + tid)];
        else if (tid < nsd.y + nsd.w) { const u32x2 x = xor_q[size_t(nsd.z + (tid - nsd.y))]; n0.x = x.x; n0.y = x.y; }
      }
      for (uint32_t i = tid; i < total; i += GSV_BLOCK_THREADS) {
        u32x4 q = r0;
        if (i < and_cnt) {
          if (i != tid) q = and_q[size_t(and_off + i)];
        } else if (i != tid) { const u32x2 x = xor_q[size_t(xor_off + (i - and_cnt))]; q.x = x.x; q.y = x.y; }
        // common slot fields: a = bits 0..20, b = 21..41, c = 42..62 of the low 64 bits
        const uint32_t sa = q.x & GSV_SLOT_MASK;
        const uint32_t sb = ((q.x >> 21) | (q.y << 11)) & GSV_SLOT_MASK;
        const uint32_t sc = (q.y >> 10) & GSV_SLOT_MASK;
        Label a = delta, b = delta;
        if (!(ka.diag & 4u)) { a = wf.ld(sa); b = wf.ld(sb); }
        if (i < and_cnt) {
          const uint32_t t = (q.y >> 31) | ((q.z & 3u) << 1);
          const uint64_t hi = (uint64_t(q.w) << 32) | q.z;
          const uint64_t gid = gid_base + ((hi >> 2) & 0x7FFFFFFFull);
          const uint32_t cti = uint32_t(hi >> 33);
          if (!EVAL) {
            Label c0, ct;
            if (ka.diag & 1u) { c0 = lxor(a, b); ct = lxor(a, tweak_of(gid)); }
            else garble_and(aes, rk, t, a, b, delta, gid, c0, ct);
            if (!(ka.diag & 8u)) {
              wf.st(sc, c0);
              CT[ct_base + cti] = u32x4{ct.w[0], ct.w[1], ct.w[2], ct.w[3]};
            } else if (c0.w[0] == 0x12345678u && ct.w[1] == 0x9abcdef0u) wf.st(sc, c0);
          } else {
            const uint32_t va = wf.ld_bit(sa), vb = wf.ld_bit(sb);
            const u32x4 cv = CT[ct_base + cti];
            const Label ct{{cv.x, cv.y, cv.z, cv.w}};
            wf.st(sc, degarble_and(aes, rk, t, ct, a, va, b, gid));
            wf.st_bit(sc, gate_eval_bit(t, va, vb));
          }
        } else {
          const uint32_t xnor = q.y >> 31;
          const Label x = lxor(a, b);
          if (!EVAL) {
            const Label c0 = lxor_if(x, delta, xnor);
            if (!(ka.diag & 8u) || c0.w[0] == 0x12345678u) wf.st(sc, c0);
          } else {
            wf.st(sc, x);
            wf.st_bit(sc, (wf.ld_bit(sa) ^ wf.ld_bit(sb) ^ xnor) & 1u);
          }
        }
      }
      __syncthreads();  // workgroup-scope release/acquire: this step's LDS + HBM stores are visible to every wave
      sd = nsd; r0 = n0;
    }
    // replay epilogue: feedback copies through staging slots (sources may alias destinations; all in HBM)
    if (ka.n_fb) {
      for (uint32_t i = tid; i < ka.n_fb; i += GSV_BLOCK_THREADS) {
        const u32x4 v = wf.hbm[ka.fb_src[i]];
        wf.hbm[ka.fb_stage_base + i] = v;
        if (EVAL) { const uint8_t bv = wf.hbm_bits[ka.fb_src[i]]; wf.hbm_bits[ka.fb_stage_base + i] = bv; }
      }
      __syncthreads();
      for (uint32_t i = tid; i < ka.n_fb; i += GSV_BLOCK_THREADS) {
        const u32x4 v = wf.hbm[ka.fb_stage_base + i];
        wf.hbm[ka.fb_dst[i]] = v;
        if (EVAL) { const uint8_t bv = wf.hbm_bits[ka.fb_stage_base + i]; wf.hbm_bits[ka.fb_dst[i]] = bv; }
      }
      __syncthreads();
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

}  // namespace dev
}  // namespace gsv

extern "C" {

int gsvk_upload_round_keys(const uint32_t rk[44]) {
  return int(hipMemcpyToSymbol(HIP_SYMBOL(gsv::dev::c_rk), rk, 44 * sizeof(uint32_t)));
}
int gsvk_launch_program(const gsv::dev::KernelArgs* ka, uint32_t n_instances, int evaluate, hipStream_t stream) {
  const size_t lds = GSV_LDS_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gsv::dev::run_program_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(gsv::dev::run_program_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e1 != hipSuccess || e2 != hipSuccess) return int(e1 != hipSuccess ? e1 : e2);
    attr_done = true;
  }
  if (evaluate) hipLaunchKernelGGL(gsv::dev::run_program_kernel<true>, dim3(n_instances), dim3(GSV_BLOCK_THREADS), lds, stream, *ka);
  else hipLaunchKernelGGL(gsv::dev::run_program_kernel<false>, dim3(n_instances), dim3(GSV_BLOCK_THREADS), lds, stream, *ka);
  return int(hipGetLastError());
}
int gsvk_gather_outputs(const void* W, const void* VB, uint32_t n_slots, const uint32_t* slots, uint32_t n_out, uint32_t n_instances,
                        void* out, void* out_bits, hipStream_t stream) {
  dim3 grid((n_out + 255) / 256, n_instances);
  hipLaunchKernelGGL(gsv::dev::gather_outputs_kernel, grid, dim3(256), 0, stream, static_cast<const uint4*>(W), static_cast<const uint8_t*>(VB),
                     n_slots, slots, n_out, static_cast<uint4*>(out), static_cast<uint8_t*>(out_bits));
  return int(hipGetLastError());
}
int gsvk_scatter_bits(void* VB, uint32_t n_slots, uint32_t first_slot, const void* bits, uint32_t n, uint32_t n_instances, hipStream_t stream) {
  dim3 grid((n + 255) / 256, n_instances);
  hipLaunchKernelGGL(gsv::dev::scatter_bits_kernel, grid, dim3(256), 0, stream, static_cast<uint8_t*>(VB), n_slots, first_slot,
                     static_cast<const uint8_t*>(bits), n);
  return int(hipGetLastError());
}

}  // extern "C"
