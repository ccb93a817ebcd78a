// Do vector and matrix instructions of DIFFERENT waves on one SIMD overlap?  Every wave repeats {NV v_fma_f32, NM v_mfma (FP6 x FP4,
// 32x32x64)} - blocks of one kind at a time, like the phases of the shared-sweep kernel - at 1, 2 or 4 waves per SIMD.
// If the time of {NV, NM} is max(time of NV alone, time of NM alone) they overlap; if it is the sum they do not.
// Identical waves that start together stay in phase (all in their vector block, then all in their matrix block): the last line staggers
// every second wave of a SIMD by one block - if THAT gives the maximum, overlap is possible and only the phase lock prevents it.
// build: hipcc --offload-arch=gfx950 -O3 valu_mfma_overlap.hip -o valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int NV, int NM, int PIN>
__global__ __launch_bounds__(256) void k(float *sink, int iters, int stagger) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = 0x11111111 * (i & 1); }
  v16f c0 = {0}, c1 = {0};
  float f0 = threadIdx.x, f1 = 1.0f, f2 = 2.0f, f3 = 3.0f;
  if (stagger && ((blockIdx.x >> 8) & 1)) {  // the second, fourth, ... wave of every SIMD starts one matrix block ahead
#pragma unroll
    for (int i = 0; i < NM / 2; ++i) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f3), "v"(f0));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f0), "v"(f1));
    }
#pragma unroll
    for (int i = 0; i < NM / 2; ++i) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (PIN == 1) asm volatile("" : "+v"(c0), "+v"(c1));
      if (PIN == 2) {  // five vector instructions in each MFMA's shadow, like the expansion of the sweep kernel
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f3), "v"(f0));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f0), "v"(f1));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f3), "v"(f0));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f0), "v"(f1));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
      }
    }
  }
  if (c0[0] + c1[0] + f0 + f1 + f2 + f3 == 12345.f) sink[0] = 1.f;
}
template <int NV, int NM, int PIN = 0>
static float run(int waves_per_simd, int iters, float *sink, int stagger = 0) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM, PIN>), dim3(blocks), dim3(256), 0, 0, sink, iters, stagger);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float *sink;
  (void)hipMalloc(&sink, 4);
  const int iters = 400;
  for (int w : {1, 2, 4}) {
    const float tv = run<320, 0>(w, iters, sink), tm = run<0, 24>(w, iters, sink), tb = run<320, 24>(w, iters, sink);
    const float tv2 = run<160, 0>(w, iters, sink), tb2 = run<160, 24>(w, iters, sink);
    // per wave and iteration, in ns: time / iters / waves per SIMD
    const float s = 1e6f / iters / w;
    const float tmp = run<0, 24, 1>(w, iters, sink), tbp = run<320, 24, 1>(w, iters, sink), tsh = run<0, 24, 2>(w, iters, sink), tshv = run<200, 24, 2>(w, iters, sink);
    printf("%d waves/SIMD: accumulators named between MFMA pairs: 24 MFMA %6.0f ns, with 320 VALU %6.0f | 24 MFMA with 120 VALU in their shadows %6.0f ns, + a block of 200 VALU %6.0f\n", w,
           tmp * s, tbp * s, tsh * s, tshv * s);
    printf("%d waves/SIMD: per wave-iteration  320 VALU %6.0f ns   24 MFMA %6.0f ns   both %6.0f ns  (sum %6.0f, max %6.0f) | 160 VALU %6.0f, with 24 MFMA %6.0f\n", w,
           tv * s, tm * s, tb * s, (tv + tm) * s, (tv > tm ? tv : tm) * s, tv2 * s, tb2 * s);
    if (w > 1) {
      const float ts = run<320, 24>(w, iters, sink, 1), ts2 = run<160, 24>(w, iters, sink, 1), ts3 = run<80, 24>(w, iters, sink, 1), t3 = run<80, 24>(w, iters, sink, 0);
      printf("%d waves/SIMD, every second wave one block ahead: 320 VALU + 24 MFMA %6.0f ns   160 VALU + 24 MFMA %6.0f ns   80 + 24: %6.0f (in phase %6.0f)\n", w, ts * s, ts2 * s, ts3 * s, t3 * s);
    }
  }
  return 0;
}
