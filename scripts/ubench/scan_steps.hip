// scan_steps.hip - which part of the scan kernel costs bandwidth?  The scan kernel's access shape (one 6400-B tile per wave, nt loads,
// query planes staged through LDS) with its work added step by step:
//   0 popcounts only   1 + f64 score bound per row (35 f64 operations)   2 + threshold test, ballot, LDS append of the survivors
//   3 + workgroup barrier at the end and one count word written per workgroup   4 + per-query parameters loaded from global memory
//   5 + the tile's side value (plain load)   6 + 0.3 survivors per chunk that gather 32 B of exact corrections and are written out
//   hipcc --offload-arch=gfx950 -O3 -o scan_steps scan_steps.hip && ./scan_steps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t popc4(u32x4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

struct QP { double ay, ly, y1, qadd, cdp, dimd; int sim, one_bit, mip, pad; };

template <int STEP>
__global__ __launch_bounds__(512) void k(const u32x4 *__restrict__ p, unsigned n_chunks, const u32x4 *__restrict__ planes, const QP *__restrict__ qps,
                                         const uint32_t *__restrict__ thetas, uint32_t *__restrict__ counts, uint64_t *__restrict__ entries, uint32_t *out,
                                         const float *__restrict__ add_range, const double *__restrict__ exact) {
  __shared__ u32x4 s_planes[24];
  __shared__ uint64_t s_ent[512];
  __shared__ uint32_t s_cnt;
  const unsigned c = blockIdx.x;
  const int q = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const u32x4 *tp = p + ((size_t)c * 8 + wave) * (6400 / 16) + lane;
  const u32x4 *gp = planes + (size_t)q * 24;
  if (tid < 24) s_planes[tid] = gp[tid];
  if (tid == 0) s_cnt = 0;
  QP qp;
  uint32_t theta = 0x7fffffffu;
  if (STEP >= 4) { qp = qps[q]; theta = thetas[q]; }
  else { qp.ay = -0.15; qp.ly = 0.02; qp.y1 = 5760.0; qp.qadd = -0.001; qp.cdp = 0.0009; qp.dimd = 768.0; qp.sim = 1; }
  __syncthreads();
  u32x4 v[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) v[j] = __builtin_nontemporal_load(tp + j * 64);
  const uint32_t cc = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(tp - lane + 6 * 64) + lane);
  float addr_ = 0.f;
  if (STEP >= 5) addr_ = add_range[((size_t)c * 8 + wave) * 2 + 1];   // per-tile side value, plain load, one address per wave
  uint32_t acc[4] = {0, 0, 0, 0}, ones = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] += popc4(v[j] & s_planes[j * 4 + b]);
    ones += popc4(v[j]);
  }
  const uint32_t qc = acc[0] + (acc[1] << 1) + (acc[2] << 2) + (acc[3] << 3);
  uint32_t r = qc + ones + cc;
  bool pass = false;
  float ub32 = 0.f;
  if (STEP >= 1) {
    const double al = (double)__uint_as_float(cc << 16), au = (double)__uint_as_float(cc & 0xffff0000u), x1 = (double)ones, qcd = (double)qc;
    const double lx = au - al;
    const double t1 = (al * qp.ay) * qp.dimd, t2 = (qp.ay * lx) * x1, t3 = (al * qp.ly) * qp.y1, t4 = (lx * qp.ly) * qcd;
    const double s = ((t1 + t2) + t3) + t4;
    const double A = qp.ay * (qp.dimd - x1) + qp.ly * (qp.y1 - qcd), B = qp.ay * x1 + qp.ly * qcd;
    const double ea = fabs(al) * 0.0078125 + 1e-37, eu = fabs(au) * 0.0078125 + 1e-37;
    const double err = fabs(A) * ea + fabs(B) * eu;
    const double t = (s + err) + (qp.qadd - qp.cdp) + 1e-4;
    const double sc = (1.0 + t) / 2.0;
    ub32 = (float)(sc * (1.0 + 1e-9));
    r ^= __float_as_uint(ub32);
    pass = __float_as_uint(ub32) > theta;   // practically never
    if (STEP >= 5) r ^= __float_as_uint(addr_);
    // the library's survivors: about 0.3 rows per chunk pass the bound, gather their exact corrections (32 B, plain) and are emitted
    if (STEP >= 6) pass = ((c * 2654435761u + (unsigned)q * 40503u + tid * 2246822519u) >> 7) % (STEP >= 7 ? 155u : 1700u) == 0u;   // step 7, 8: 3.3 per chunk (the 1 M-row index)
    if (STEP >= 6 && pass) {
      const double *ex = exact + ((size_t)c * 512 + tid) * 4;
      double e0, e1, e2;
      if (STEP >= 8) { e0 = __builtin_nontemporal_load(ex); e1 = __builtin_nontemporal_load(ex + 1); e2 = __builtin_nontemporal_load(ex + 2); }
      else { e0 = ex[0]; e1 = ex[1]; e2 = ex[2]; }
      ub32 = (float)(e0 + e1 + e2);
    }
  }
  if (STEP >= 2) {
    const uint64_t m = __ballot(pass);
    if (m) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
      base = __shfl(base, 0);
      if (pass) s_ent[(base + __popcll(m & ((1ull << lane) - 1))) & 511] = ((uint64_t)__float_as_uint(ub32) << 32) | (c * 512u + tid);
    }
  }
  if (STEP >= 3) {
    __syncthreads();
    const uint32_t n = s_cnt;
    if (STEP >= 8) {
      for (uint32_t i = tid; i < n; i += 512) __builtin_nontemporal_store(s_ent[i], &entries[((size_t)q * n_chunks + c) * 16 + (i & 15)]);
      if (tid == 0) __builtin_nontemporal_store(n, &counts[(size_t)q * n_chunks + c]);
    } else {
      for (uint32_t i = tid; i < n; i += 512) entries[((size_t)q * n_chunks + c) * 16 + (i & 15)] = s_ent[i];
      if (tid == 0) counts[(size_t)q * n_chunks + c] = n;
    }
  }
  if (r == 0x12345678u) atomicAdd(out, 1u);
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Bufs { u32x4 *d, *planes; QP *qps; uint32_t *thetas, *counts, *out; uint64_t *entries; float *add_range; double *exact; };

template <int STEP>
static int run(const Bufs &b, size_t bytes, int reps, unsigned drop, const char *name) {
  const unsigned n_chunks = (unsigned)(bytes / 51200) - drop;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0;
  for (int it = 0; it < 7; ++it) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<STEP>), dim3(n_chunks, reps), dim3(512), 0, 0, b.d, n_chunks, b.planes, b.qps, b.thetas, b.counts, b.entries, b.out, b.add_range, b.exact);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  const double by = (double)n_chunks * 51200 * reps;
  printf("%4zu MB x %3d sweeps  chunks %% 8 = %u  step %d %-46s: avg %7.1f GB/s  best %7.1f GB/s\n", bytes >> 20, reps, n_chunks % 8, STEP, name,
         by / (sum / 5 * 1e-3) / 1e9, by / (best * 1e-3) / 1e9);
  return 0;
}

int main() {
  const size_t cap = (size_t)1100 << 20;
  Bufs b;
  CHK(hipMalloc((void **)&b.d, cap));
  CHK(hipMemset(b.d, 0x5a, cap));
  CHK(hipMalloc((void **)&b.planes, 128 * 24 * 16)); CHK(hipMemset(b.planes, 0x33, 128 * 24 * 16));
  QP h[128];
  for (int i = 0; i < 128; ++i) { h[i].ay = -0.15; h[i].ly = 0.02; h[i].y1 = 5760.0; h[i].qadd = -0.001; h[i].cdp = 0.0009; h[i].dimd = 768.0; h[i].sim = 1; h[i].one_bit = 0; h[i].mip = 0; h[i].pad = 0; }
  CHK(hipMalloc((void **)&b.qps, sizeof h)); CHK(hipMemcpy(b.qps, h, sizeof h, hipMemcpyHostToDevice));
  uint32_t th[128]; for (int i = 0; i < 128; ++i) th[i] = 0x7fffffffu;
  CHK(hipMalloc((void **)&b.thetas, sizeof th)); CHK(hipMemcpy(b.thetas, th, sizeof th, hipMemcpyHostToDevice));
  CHK(hipMalloc((void **)&b.counts, (size_t)128 * 22000 * 4));
  CHK(hipMalloc((void **)&b.entries, (size_t)128 * 22000 * 16 * 8));
  CHK(hipMalloc((void **)&b.out, 4)); CHK(hipMemset(b.out, 0, 4));
  CHK(hipMalloc((void **)&b.add_range, (size_t)22000 * 8 * 2 * 4)); CHK(hipMemset(b.add_range, 0, (size_t)22000 * 8 * 2 * 4));
  CHK(hipMalloc((void **)&b.exact, (size_t)22000 * 512 * 32)); CHK(hipMemset(b.exact, 0, (size_t)22000 * 512 * 32));
  {  // the library's large launch on the 1 M-row index: 1378 chunks starting at chunk 576, 128 queries; x extent as is / padded to 1384
    Bufs o = b;
    o.d = b.d + (size_t)576 * 51200 / 16;
    if (run<6>(o, (size_t)1378 * 51200, 128, 0, "library geometry, 1378 chunks from chunk 576")) return 1;
    if (run<6>(o, (size_t)1384 * 51200, 128, 0, "library geometry, padded to 1384 chunks")) return 1;
    if (run<4>(o, (size_t)1384 * 51200, 128, 0, "library geometry, padded to 1384 chunks")) return 1;
    if (run<7>(o, (size_t)1384 * 51200, 128, 0, "1384 chunks, 3.3 survivors per chunk, plain")) return 1;
    if (run<8>(o, (size_t)1384 * 51200, 128, 0, "1384 chunks, 3.3 survivors per chunk, nt")) return 1;
    if (run<7>(o, (size_t)1378 * 51200, 128, 0, "1378 chunks, 3.3 survivors per chunk, plain")) return 1;
    if (run<8>(o, (size_t)1378 * 51200, 128, 0, "1378 chunks, 3.3 survivors per chunk, nt")) return 1;
  }
  for (size_t mb : {1000, 100}) {
    const size_t bytes = mb << 20;
    const int reps = mb == 1000 ? 8 : 40;
    for (unsigned drop : {3u, 0u}) {
      if (run<0>(b, bytes, reps, drop, "popcounts")) return 1;
      if (run<1>(b, bytes, reps, drop, "+ f64 bound")) return 1;
      if (run<2>(b, bytes, reps, drop, "+ threshold, ballot, LDS append")) return 1;
      if (run<3>(b, bytes, reps, drop, "+ end barrier, count word per workgroup")) return 1;
      if (run<4>(b, bytes, reps, drop, "+ query parameters from global memory")) return 1;
      if (run<5>(b, bytes, reps, drop, "+ per-tile side value (plain load)")) return 1;
      if (run<6>(b, bytes, reps, drop, "+ 0.3 survivors per chunk: gather 32 B, emit")) return 1;
    }
  }
  return 0;
}
