// bbq_rerank_kernels.hip - exact rerank of oversampled candidates (gfx950).
//
// computeSimilarity (reference src/vectorSimilarity.ts:14-126) of a raw fp32 query against chosen rows of the original
// fp32 matrix, which stays resident in HBM.  The reference accumulates in f64 in index order, so every candidate is
// one sequential f64 chain: lane = candidate, 64 candidates per wave.  Rows are gathered 32 columns at a time through
// LDS (each wave load covers two 128-byte row pieces, row stride padded to 33 words so the per-lane reads are
// conflict-free) and the next stage's loads are issued before the current stage's chain, which hides the gather.
// HBM traffic is the candidates' rows once (dim*4 B per candidate); the chain is v_*_f64 bound (~8 ops per element).
#include <hip/hip_runtime.h>
#include "bbq_device.h"
#include "bbq_launch.h"

namespace bbq {

namespace {

constexpr int kRrCols = 32;                   // columns per stage
constexpr int kRrLoads = 64 * kRrCols / 64;   // loads per lane per stage (= 32)

__global__ __launch_bounds__(64) void bbq_rerank_kernel(RerankArgs a) {
  const int q = blockIdx.y;
  const int64_t beg = a.offsets[q], end = a.offsets[q + 1];
  const int64_t c0 = beg + (int64_t)blockIdx.x * 64;
  if (c0 >= end) return;  // uniform per workgroup
  const int lane = threadIdx.x;
  __shared__ float s_rows[64][kRrCols + 1];
  __shared__ float s_q[kRrCols];
  __shared__ int64_t s_base[64];
  const bool valid = c0 + lane < end;
  s_base[lane] = (int64_t)a.rows[valid ? c0 + lane : c0] * a.dim;  // idle lanes shadow the first candidate
  __syncthreads();
  const float *__restrict__ qv = a.queries + (int64_t)q * a.dim;
  // lane -> (row it*2 + lane/32, column lane%32) of the stage
  const int lc = lane & 31, lr = lane >> 5;
  float nxt[kRrLoads];
  float nq = 0.f;
  auto fetch = [&](int col0) {
    const bool in = col0 + lc < a.dim;
#pragma unroll
    for (int it = 0; it < kRrLoads; ++it) nxt[it] = in ? __builtin_nontemporal_load(a.vecs + s_base[it * 2 + lr] + col0 + lc) : 0.f;
    nq = (lane < kRrCols && col0 + lane < a.dim) ? qv[col0 + lane] : 0.f;
  };
  fetch(0);
  double dp = 0, na = 0, nb = 0;
  for (int col0 = 0; col0 < a.dim; col0 += kRrCols) {
#pragma unroll
    for (int it = 0; it < kRrLoads; ++it) s_rows[it * 2 + lr][lc] = nxt[it];
    if (lane < kRrCols) s_q[lane] = nq;
    __syncthreads();
    if (col0 + kRrCols < a.dim) fetch(col0 + kRrCols);
    const int nc = min(kRrCols, a.dim - col0);
    if (a.sim == 1) {  // COSINE :73-101
      for (int j = 0; j < nc; ++j) {
        const double av = (double)s_q[j], bv = (double)s_rows[lane][j];
        dp += av * bv;
        na += av * av;
        nb += bv * bv;
      }
    } else if (a.sim == 0) {  // EUCLIDEAN :38-70
      for (int j = 0; j < nc; ++j) {
        const double diff = (double)s_q[j] - (double)s_rows[lane][j];
        dp += diff * diff;
      }
    } else {  // MAXIMUM_INNER_PRODUCT :108-118
      for (int j = 0; j < nc; ++j) dp += (double)s_q[j] * (double)s_rows[lane][j];
    }
    __syncthreads();
  }
  double r;
  if (a.sim == 1) r = (na == 0 || nb == 0) ? 0.0 : dp / (sqrt(na) * sqrt(nb));
  else if (a.sim == 0) r = 1.0 / (1.0 + sqrt(dp));
  else r = dp;
  if (valid) a.out[c0 + lane] = r;
}

}  // namespace

hipError_t launch_rerank(const RerankArgs &a, int n_queries, int64_t max_count, hipStream_t s) {
  if (n_queries <= 0 || max_count <= 0) return hipSuccess;
  dim3 grid((unsigned)((max_count + 63) / 64), (unsigned)n_queries);
  hipLaunchKernelGGL(bbq_rerank_kernel, grid, dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace bbq
