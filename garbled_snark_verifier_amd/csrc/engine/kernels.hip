// gfx950 (MI355X / CDNA4) kernels of the garbling engine.
//
// Execution model: one 1024-thread workgroup = one garbling instance on one CU.
// The circuit is deep and narrow (Fq12 mul: ~20.3 M gates over ~10^4 dependent AND levels, mean
// width a few hundred gates), so a step is latency-bound, not throughput-bound: the workgroup walks
// the program's steps with `s_barrier` between them — no grid-wide synchronisation, no inter-CU
// traffic — and independent cut-and-choose instances fill the other CUs.
//   * wire file W[instance][slot] : 16-byte labels in HBM, L2-resident working set, 128-bit
//     coalescable loads/stores (global_load_dwordx4)
//   * AES T-tables: 128 KiB in LDS per workgroup, bank-replicated so lookups never conflict;
//     round keys scalar (constant address space)
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

__device__ __forceinline__ Label ld_label(const uint4* p) {
  uint4 v = *p;
  return Label{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st_label(uint4* p, const Label& l) { *p = make_uint4(l.w[0], l.w[1], l.w[2], l.w[3]); }

// T-tables in LDS, every entry replicated once per LDS bank: dword index = K*8192 + x*32 + (lane & 31).
// A wave's ds_read_b32 is served in two 32-lane groups (MI355X_MICROARCH.md §LDS); inside a group lane j
// only ever touches bank j, so 64 data-dependent lookups cost the conflict-free 2 cycles instead of the
// ~3.5x of a shared 1 KiB table.  4 tables x 256 entries x 32 banks x 4 B = 128 KiB of the CU's 160 KiB.
struct LdsBankedTables {
  const char* base;   // LDS
  uint32_t off[4];    // (lane & 31) * 4 + K * 32768, bytes
  template <int K, int BYTE>
  __device__ __forceinline__ uint32_t lk(uint32_t s) const {
    // byte BYTE of s, times 128 (32 banks x 4 B), as bits 7..14
    uint32_t x = BYTE == 0 ? (s << 7) : BYTE == 1 ? (s >> 1) : BYTE == 2 ? (s >> 9) : (s >> 17);
    return *reinterpret_cast<const uint32_t*>(base + ((x & 0x7f80u) | off[K]));
  }
};
#define GSV_TE_LDS_WORDS (4 * 256 * 32)

template <bool EVAL>
__global__ __launch_bounds__(GSV_BLOCK_THREADS) void run_program_kernel(KernelArgs ka) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_te[];  // GSV_TE_LDS_WORDS
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < GSV_TE_LDS_WORDS; i += GSV_BLOCK_THREADS) s_te[i] = ka.te[((i >> 13) << 8) | ((i >> 5) & 255u)];
  __syncthreads();
  const uint32_t lane4 = (tid & 31u) * 4u;
  const LdsBankedTables aes{reinterpret_cast<const char*>(s_te), {lane4, lane4 + 32768u, lane4 + 65536u, lane4 + 98304u}};
  const uint32_t* rk = c_rk;

  const uint32_t inst = blockIdx.x;
  uint4* __restrict__ W = ka.W + size_t(inst) * ka.n_slots;
  uint8_t* __restrict__ VB = EVAL ? ka.VB + size_t(inst) * ka.n_slots : nullptr;
  uint4* __restrict__ CT = ka.CT + size_t(inst) * ka.ct_stride;
  const Label delta = EVAL ? Label{{0, 0, 0, 0}} : ld_label(ka.delta + inst);
  const uint4* __restrict__ and_q = reinterpret_cast<const uint4*>(ka.ands);
  const uint4* __restrict__ xor_q = reinterpret_cast<const uint4*>(ka.xors);

  for (uint32_t rep = 0; rep < ka.replays; ++rep) {
    const uint64_t gid_base = ka.gid_base + uint64_t(rep) * ka.n_gates;
    const uint64_t ct_base = uint64_t(rep % ka.ct_cap_replays) * ka.n_ct;

    // step descriptor {and_off, and_cnt, xor_off, xor_cnt}: lanes [0, and_cnt) take AND-family gates,
    // lanes [and_cnt, and_cnt + xor_cnt) take free gates, so only the boundary wave diverges.
    uint4 sd = reinterpret_cast<const uint4*>(ka.steps)[0];
    // prefetched records for this thread's first gate of the step
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
    if (tid < sd.y) { r0 = and_q[2 * size_t(sd.x + tid)]; r1 = and_q[2 * size_t(sd.x + tid) + 1]; }
    else if (tid < sd.y + sd.w) r0 = xor_q[size_t(sd.z + (tid - sd.y))];
    for (uint32_t s = 0; s < ka.n_steps; ++s) {
      const uint32_t and_off = sd.x, and_cnt = sd.y, xor_off = sd.z, total = sd.y + sd.w;
      // issue next step's descriptor + record loads early; they complete while this step computes
      uint4 nsd = sd, n0 = r0, n1 = r1;
      if (s + 1 < ka.n_steps && !(ka.diag & 2u)) {
        nsd = reinterpret_cast<const uint4*>(ka.steps)[s + 1];
        if (tid < nsd.y) { n0 = and_q[2 * size_t(nsd.x + tid)]; n1 = and_q[2 * size_t(nsd.x + tid) + 1]; }
        else if (tid < nsd.y + nsd.w) n0 = xor_q[size_t(nsd.z + (tid - nsd.y))];
      }
      for (uint32_t i = tid; i < total; i += GSV_BLOCK_THREADS) {
        if (i < and_cnt) {
          uint4 q0 = r0, q1 = r1;
          if (i != tid) { q0 = and_q[2 * size_t(and_off + i)]; q1 = and_q[2 * size_t(and_off + i) + 1]; }
          const uint32_t t = q0.w;
          const uint64_t gid = gid_base + q1.x;
          Label a = delta, b = delta;
          if (!(ka.diag & 4u)) { a = ld_label(W + q0.x); b = ld_label(W + q0.y); }
          if (!EVAL) {
            Label c0, ct;
            if (ka.diag & 1u) { c0 = lxor(a, b); ct = lxor(a, tweak_of(gid)); }
            else garble_and(aes, rk, t, a, b, delta, gid, c0, ct);
            if (!(ka.diag & 8u)) {
              st_label(W + q0.z, c0);
              st_label(CT + ct_base + q1.y, ct);
            } else if (c0.w[0] == 0x12345678u && ct.w[1] == 0x9abcdef0u) st_label(W + q0.z, c0);
          } else {
            const uint32_t va = VB[q0.x], vb = VB[q0.y];
            const Label ct = ld_label(CT + ct_base + q1.y);
            st_label(W + q0.z, degarble_and(aes, rk, t, ct, a, va, b, gid));
            VB[q0.z] = uint8_t(gate_eval_bit(t, va, vb));
          }
        } else {
          uint4 q0 = r0;
          if (i != tid) q0 = xor_q[size_t(xor_off + (i - and_cnt))];
          const uint32_t t = q0.w;
          Label a = delta, b = delta;
          if (!(ka.diag & 4u)) { a = ld_label(W + q0.x); b = ld_label(W + q0.y); }
          if (!EVAL) {
            const Label c0 = garble_free(t, a, b, delta);
            if (!(ka.diag & 8u) || c0.w[0] == 0x12345678u) st_label(W + q0.z, c0);
          } else {
            st_label(W + q0.z, degarble_free(t, a, b));
            VB[q0.z] = uint8_t(gate_eval_bit(t, VB[q0.x], VB[q0.y]));
          }
        }
      }
      __syncthreads();  // workgroup-scope release/acquire: this step's W stores are visible to every wave
      if (ka.diag & 2u) {
        if (s + 1 < ka.n_steps) {
          nsd = reinterpret_cast<const uint4*>(ka.steps)[s + 1];
          if (tid < nsd.y) { n0 = and_q[2 * size_t(nsd.x + tid)]; n1 = and_q[2 * size_t(nsd.x + tid) + 1]; }
          else if (tid < nsd.y + nsd.w) n0 = xor_q[size_t(nsd.z + (tid - nsd.y))];
        }
      }
      sd = nsd; r0 = n0; r1 = n1;
    }
    // replay epilogue: feedback copies through staging slots (sources may alias destinations)
    if (ka.n_fb) {
      for (uint32_t i = tid; i < ka.n_fb; i += GSV_BLOCK_THREADS) {
        W[ka.fb_stage_base + i] = W[ka.fb_src[i]];
        if (EVAL) VB[ka.fb_stage_base + i] = VB[ka.fb_src[i]];
      }
      __syncthreads();
      for (uint32_t i = tid; i < ka.n_fb; i += GSV_BLOCK_THREADS) {
        W[ka.fb_dst[i]] = W[ka.fb_stage_base + i];
        if (EVAL) VB[ka.fb_dst[i]] = VB[ka.fb_stage_base + i];
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
  const size_t lds = GSV_TE_LDS_WORDS * sizeof(uint32_t);
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
