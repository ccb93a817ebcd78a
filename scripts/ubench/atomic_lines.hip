// How fast do returning global atomics retire when many workgroups add to a FEW counters?  (the append counters of a sweep: one per query)
//   mode 0: every atomic on ONE word           mode 1: on 32 adjacent words (one 128-byte line)
//   mode 2: on 32 words, one per 128-byte line mode 3: as 2, without using the returned value (fire and forget)
// build: hipcc --offload-arch=gfx950 -O3 atomic_lines.hip -o atomic_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(uint32_t *ctr, uint32_t *sink, int mode, int per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  for (int i = wave; i < per_wg; i += blockDim.x / 64) {
    const int q = (i + blockIdx.x) & 31;
    uint32_t *p = mode == 0 ? ctr : mode == 1 ? ctr + q : ctr + q * 32;
    if (lane == 0) {
      if (mode == 3) atomicAdd(p, 1u);
      else acc += atomicAdd(p, 1u);
    }
  }
  if (lane == 0 && acc == 0xffffffffu) sink[0] = acc;
}
int main() {
  uint32_t *ctr, *sink;
  hipMalloc(&ctr, 32 * 128 * 4);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {448, 3584}) for (int per_wg : {8, 32}) for (int mode = 0; mode < 4; ++mode) {
    hipMemset(ctr, 0, 32 * 128 * 4);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, ctr, sink, mode, per_wg);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, ctr, sink, mode, per_wg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)wgs * per_wg;
    printf("wgs %5d  atomics/wg %3d  mode %d: %8.1f us per launch, %6.2f ns per atomic\n", wgs, per_wg, mode, ms * 100, ms * 1e5 / n);
  }
  return 0;
}
