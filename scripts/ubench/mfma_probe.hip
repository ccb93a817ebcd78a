// probe: fragment layout of v_mfma_i32_32x32x32_i8 on gfx950 (A row = lane%32? B col = lane%32? C register -> row mapping?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__host__ __device__ inline int8_t av(int m, int k) { return (int8_t)(((m * 7 + k * 3) % 5) - 2); }
__host__ __device__ inline int8_t bv(int k, int n) { return (int8_t)(((k * 5 + n * 11) % 7) - 3); }
__global__ void probe(int *out) {
  const int l = threadIdx.x, m = l % 32, h = l / 32;
  int8_t ab[16], bb[16];
  for (int i = 0; i < 16; ++i) { const int k = 16 * h + i; ab[i] = av(m, k); bb[i] = bv(k, m); }
  i32x4 a, b;
  memcpy(&a, ab, 16); memcpy(&b, bb, 16);
  i32x16 c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
  int *d; hipMalloc(&d, 64 * 16 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h_[64 * 16]; hipMemcpy(h_, d, sizeof h_, hipMemcpyDeviceToHost);
  int ref[32][32];
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int k = 0; k < 32; ++k) s += av(m, k) * bv(k, n); ref[m][n] = s; }
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l / 32), col = l % 32;
    if (h_[l * 16 + r] != ref[row][col]) ++bad;
  }
  printf("hypothesis A[m=lane%%32][k=16*(lane/32)+i], B[k][n=lane%%32], C[r] -> row (r&3)+8*(r>>2)+4*(lane/32), col lane%%32: %d mismatches of 1024\n", bad);
  if (bad) { for (int r = 0; r < 16; ++r) printf("lane0 c[%d]=%d  ", r, h_[r]); printf("\nref row0: "); for (int n = 0; n < 4; ++n) printf("%d ", ref[0][n]); printf("\nref col0: "); for (int m = 0; m < 16; ++m) printf("%d ", ref[m][0]); printf("\n"); }
  return 0;
}
