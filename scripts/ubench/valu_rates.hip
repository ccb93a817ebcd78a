// micro-benchmark: issue cost of v_bcnt_u32_b32 / v_and / v_add_f64 / v_mul_f64 / v_dot8_u32_u4 on gfx950 (one wave per SIMD and 4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__global__ void k(uint32_t *out, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = 11 * a0, a5 = 13 * a0, a6 = 17 * a0, a7 = 19 * a0;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const uint32_t m = out[0];
  const double dm = (double)m + 1.000001;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) { a0 = __popc(a0 ^ m) + a1; a1 = __popc(a1 ^ m) + a2; a2 = __popc(a2 ^ m) + a3; a3 = __popc(a3 ^ m) + a4; a4 = __popc(a4 ^ m) + a5; a5 = __popc(a5 ^ m) + a6; a6 = __popc(a6 ^ m) + a7; a7 = __popc(a7 ^ m) + a0; }
      if (OP == 1) { a0 = (a0 ^ m) + a1; a1 = (a1 ^ m) + a2; a2 = (a2 ^ m) + a3; a3 = (a3 ^ m) + a4; a4 = (a4 ^ m) + a5; a5 = (a5 ^ m) + a6; a6 = (a6 ^ m) + a7; a7 = (a7 ^ m) + a0; }
      if (OP == 2) { d0 = d0 * dm; d1 = d1 * dm; d2 = d2 * dm; d3 = d3 * dm; d4 = d4 * dm; d5 = d5 * dm; d6 = d6 * dm; d7 = d7 * dm; }
      if (OP == 3) { d0 = d0 + dm; d1 = d1 + dm; d2 = d2 + dm; d3 = d3 + dm; d4 = d4 + dm; d5 = d5 + dm; d6 = d6 + dm; d7 = d7 + dm; }
      if (OP == 4) { a0 = __builtin_amdgcn_udot8(a0, m, a1, false); a1 = __builtin_amdgcn_udot8(a1, m, a2, false); a2 = __builtin_amdgcn_udot8(a2, m, a3, false); a3 = __builtin_amdgcn_udot8(a3, m, a4, false);
                     a4 = __builtin_amdgcn_udot8(a4, m, a5, false); a5 = __builtin_amdgcn_udot8(a5, m, a6, false); a6 = __builtin_amdgcn_udot8(a6, m, a7, false); a7 = __builtin_amdgcn_udot8(a7, m, a0, false); }
      if (OP == 5) { a0 = __builtin_amdgcn_udot4(a0, m, a1, false); a1 = __builtin_amdgcn_udot4(a1, m, a2, false); a2 = __builtin_amdgcn_udot4(a2, m, a3, false); a3 = __builtin_amdgcn_udot4(a3, m, a4, false);
                     a4 = __builtin_amdgcn_udot4(a4, m, a5, false); a5 = __builtin_amdgcn_udot4(a5, m, a6, false); a6 = __builtin_amdgcn_udot4(a6, m, a7, false); a7 = __builtin_amdgcn_udot4(a7, m, a0, false); }
    }
  }
  long long t1 = clock64();
  uint32_t r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[1 + blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[100000] = t1 - t0;
}
template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd) {
  const int iters = 4000;
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpy(&cyc, ((long long *)d) + 100000, 8, hipMemcpyDeviceToHost);
  const double ops = (double)iters * 64;  // per wave
  printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction-slot (clock64 %lld cyc / %0.f ops x waves/SIMD), kernel %.3f ms\n", name, waves_per_simd,
         (double)cyc / ops / 1.0, cyc, ops, ms);
}
int main() {
  uint32_t *d; hipMalloc(&d, 8 << 20);
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("xor+bcnt(acc)", d, w); run<1>("xor+add", d, w); run<2>("v_mul_f64", d, w); run<3>("v_add_f64", d, w); run<4>("xor? udot8", d, w); run<5>("udot4", d, w);
  }
  return 0;
}
