// Round 6 experiment: does the ORDER in which one lane issues a round's 32 table lookups move the T-table AES ceiling?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/aes_sched.hip -o tools/ubench/aes_sched
// Variants of the production form (tools/ubench/aes_forms.hip, ttable_kernel = kernels.hip's one-gate-per-lane AES), all validated
// against the host's AES and timed with every CU full (16 waves per CU, 64 chained encryptions of two blocks per lane):
//   0  compiler's schedule (the production code: lookups consumed in groups of four, `s_waitcnt lgkmcnt(4/2/0)` every few instructions)
//   1  a round's 32 addresses, then its 32 lookups, then the column sums (sched_group_barrier pins the three groups)
//   2  half rounds: 16 addresses + 16 lookups of block A, the same for block B, then the sums of A under B's lookups
//   3  as 0 with four blocks per lane (two gates' worth: more independent chains per wave, 2x the state registers)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../garbled_snark_verifier_amd/csrc/engine/host_crypto.hpp"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__constant__ uint32_t c_rk[44];
#define LDS_U32 __attribute__((address_space(3))) uint32_t
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int BYTE> __device__ __forceinline__ uint32_t addr_of(uint32_t s, uint32_t lane4) { return __builtin_amdgcn_perm(s, lane4, 0x0c0c0000u | (uint32_t(4 + BYTE) << 8)); }
__device__ __forceinline__ uint32_t ld(uint32_t addr, bool te2) { return *reinterpret_cast<const LDS_U32*>(uintptr_t(addr + (te2 ? 128u : 0u))); }
template <int K, int BYTE> __device__ __forceinline__ uint32_t lk(uint32_t s, uint32_t lane4) { return ld(addr_of<BYTE>(s, lane4), (K & 2) != 0); }
__device__ __forceinline__ uint32_t col(uint32_t l4, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t k) {
  const uint32_t odd = xor3(lk<0, 1>(x1, l4), lk<2, 3>(x3, l4), k);
  return xor3(lk<0, 0>(x0, l4), lk<2, 2>(x2, l4), __builtin_amdgcn_alignbit(odd, odd, 24));
}
__device__ __forceinline__ uint32_t last(uint32_t l4, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t k) {
  const uint32_t m0 = lk<2, 0>(x0, l4), m1 = lk<0, 1>(x1, l4), m2 = lk<0, 2>(x2, l4), m3 = lk<2, 3>(x3, l4);
  return xor3(__builtin_amdgcn_perm(m1, m0, 0x0c0c0500u), __builtin_amdgcn_perm(m3, m2, 0x07020c0cu), k);
}
// one middle round of ONE block with the three phases written apart: a[16] addresses, v[16] lookups, then the sums
struct Blk { uint32_t s[4]; };
__device__ __forceinline__ void round_addr(const Blk& b, uint32_t l4, uint32_t (&a)[16]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    a[4 * c + 0] = addr_of<0>(b.s[c], l4); a[4 * c + 1] = addr_of<1>(b.s[(c + 1) & 3], l4);
    a[4 * c + 2] = addr_of<2>(b.s[(c + 2) & 3], l4); a[4 * c + 3] = addr_of<3>(b.s[(c + 3) & 3], l4);
  }
}
__device__ __forceinline__ void round_load(const uint32_t (&a)[16], uint32_t (&v)[16]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) { v[4 * c + 0] = ld(a[4 * c + 0], false); v[4 * c + 1] = ld(a[4 * c + 1], false); v[4 * c + 2] = ld(a[4 * c + 2], true); v[4 * c + 3] = ld(a[4 * c + 3], true); }
}
__device__ __forceinline__ void round_sum(Blk& b, const uint32_t (&v)[16], int r) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t odd = xor3(v[4 * c + 1], v[4 * c + 3], c_rk[4 * r + c]);
    b.s[c] = xor3(v[4 * c + 0], v[4 * c + 2], __builtin_amdgcn_alignbit(odd, odd, 24));
  }
}
template <int V>
__global__ __launch_bounds__(1024) void ttable_kernel(const uint32_t* te, uint4* io, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)smem;
  for (uint32_t i = threadIdx.x; i < 65536 / 4; i += 1024) *reinterpret_cast<LDS_U32*>(uintptr_t(i * 4u)) = te[((i & 32u) ? 512u : 0u) + (i >> 6)];
  __syncthreads();
  const uint32_t l4 = (threadIdx.x & 31u) * 4u;
  constexpr int NB = V == 3 ? 4 : 2;
  const size_t g = size_t(blockIdx.x) * 1024 + threadIdx.x;
  Blk b[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) { const uint4 x = io[NB * g + j]; b[j].s[0] = x.x; b[j].s[1] = x.y; b[j].s[2] = x.z; b[j].s[3] = x.w; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) b[j].s[c] ^= c_rk[c];
#pragma unroll
    for (int r = 1; r < 10; ++r) {
      if (V == 0 || V == 3) {
        uint32_t t[NB][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < NB; ++j) t[j][c] = col(l4, b[j].s[c], b[j].s[(c + 1) & 3], b[j].s[(c + 2) & 3], b[j].s[(c + 3) & 3], c_rk[4 * r + c]);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) b[j].s[c] = t[j][c];
      } else if (V == 1) {
        uint32_t a0[16], a1[16], v0[16], v1[16];
        round_addr(b[0], l4, a0); round_addr(b[1], l4, a1);
        round_load(a0, v0); round_load(a1, v1);
        round_sum(b[0], v0, r); round_sum(b[1], v1, r);
        __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);  // 32 VALU: the addresses
        __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);  // 32 DS reads
        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);  // the sums
      } else {
        uint32_t a0[16], a1[16], v0[16], v1[16];
        round_addr(b[0], l4, a0); round_load(a0, v0);
        round_addr(b[1], l4, a1); round_load(a1, v1);
        round_sum(b[0], v0, r); round_sum(b[1], v1, r);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
      }
    }
    uint32_t t[NB][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < NB; ++j) t[j][c] = last(l4, b[j].s[c], b[j].s[(c + 1) & 3], b[j].s[(c + 2) & 3], b[j].s[(c + 3) & 3], c_rk[40 + c]);
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) b[j].s[c] = t[j][c];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) io[NB * g + j] = uint4{b[j].s[0], b[j].s[1], b[j].s[2], b[j].s[3]};
}

template <int V>
static int run(const gsv::AesTables& t, int cus, const char* what) {
  constexpr int NB = V == 3 ? 4 : 2;
  const int iters = 64, wgs = cus * 8;
  void* te; CHK(hipMalloc(&te, sizeof t.te)); CHK(hipMemcpy(te, t.te, sizeof t.te, hipMemcpyHostToDevice));
  const size_t n_blocks = size_t(wgs) * 1024 * NB;
  std::vector<uint8_t> h(n_blocks * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = uint8_t(i * 131 + (i >> 9));
  uint4* d; CHK(hipMalloc(&d, h.size())); CHK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(ttable_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(ttable_kernel<V>, dim3(wgs), dim3(1024), 65536, 0, static_cast<const uint32_t*>(te), d, iters);
  std::vector<uint8_t> out(h.size());
  CHK(hipMemcpy(out.data(), d, h.size(), hipMemcpyDeviceToHost));
  bool ok = true;
  for (size_t bi : {size_t(0), size_t(1), size_t(12345), n_blocks - 1}) {
    uint8_t ref[16]; std::memcpy(ref, &h[bi * 16], 16);
    for (int i = 0; i < iters; ++i) { uint8_t o[16]; gsv::CbcMacHost::encrypt_portable(t, ref, o); std::memcpy(ref, o, 16); }
    ok = ok && std::memcmp(ref, &out[bi * 16], 16) == 0;
  }
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipEventRecord(e0));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(ttable_kernel<V>, dim3(wgs), dim3(1024), 65536, 0, static_cast<const uint32_t*>(te), d, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double rate = 5.0 * double(n_blocks) * iters / (ms * 1e-3);
    if (rate > best) best = rate;
  }
  hipFuncAttributes fa; CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(ttable_kernel<V>)));
  std::printf("variant %d (%s, %d VGPRs): %.3e blocks/s on %d CUs = %.3e per CU  [%s]\n", V, what, fa.numRegs, best, cus, best / cus, ok ? "matches host AES" : "MISMATCH");
  CHK(hipFree(d)); CHK(hipFree(te));
  return 0;
}
int main() {
  const gsv::AesTables& t = gsv::AesTables::fixed_key();
  uint32_t dev_rk[44];
  for (int i = 0; i < 44; ++i) dev_rk[i] = (i >= 4 && i < 40) ? ((t.rk[i] >> 8) | (t.rk[i] << 24)) : t.rk[i];
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_rk), dev_rk, sizeof dev_rk));
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  if (run<0>(t, cus, "compiler's schedule, 2 blocks per lane")) return 1;
  if (run<1>(t, cus, "32 addresses | 32 lookups | sums")) return 1;
  if (run<2>(t, cus, "16+16 | 16+16 | sums")) return 1;
  if (run<3>(t, cus, "compiler's schedule, 4 blocks per lane")) return 1;
  return 0;
}
