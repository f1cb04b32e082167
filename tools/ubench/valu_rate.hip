// Micro-benchmark: integer VALU issue rate per CU for the instruction mix of the AES rounds
// (v_xor / v_lshl_or / v_bfe / v_alignbit), W waves per CU.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* out, int iters) {
  unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = a ^ 9, g = a + 11, h = a * 13;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a = (a << 7 | b) ^ c; b = __builtin_amdgcn_alignbit(b, b, 8) ^ d; c = ((c >> 9) & 0xff) + e; d = (d << 3 | f) ^ g;
      e = (e << 7 | f) ^ g; f = __builtin_amdgcn_alignbit(f, f, 16) ^ h; g = ((g >> 9) & 0xff) + a; h = (h << 3 | b) ^ c;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main(int argc, char** argv) {
  int iters = 20000;
  unsigned* d; hipMalloc(&d, 256 * 1024 * 4 * 4);
  for (int threads : {64, 256, 512, 1024}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration: 16 * 8 statements, each ~2 VALU ops => count precisely from ISA if needed; report statements/s
    double stmts = double(iters) * 16 * 8 * (threads / 64);  // wave-statements per CU
    printf("threads/CU %4d: %.3f ms, %.2f wave-statements per us per CU (each statement = 2 VALU ops)\n", threads, ms, stmts / (ms * 1e3));
  }
  return 0;
}
