// bbq_mfma_kernels.hip - shared sweep on the matrix cores (API extension; NOT the one-sweep-per-query path).
//
// With 32 queries sharing a sweep the integer dot products ARE a dense contraction,
//     qcDist[row][query] = sum_d bit_d(row) * q[query][d],
// and the VALU popcount formulation is bound by v_bcnt (bbq_kernels.hip: shared kernel, ~1.4x).  Here every wave expands
// the 1-bit codes of its 64-row tile to int8 {0,1} fragments on the fly and multiplies them with the 32 queries' int8
// values on v_mfma_i32_32x32x32_i8 (fragment layout verified on gfx950 by scripts/ubench/mfma_probe.hip:
//   A[m = lane%32][k = 16*(lane/32)+i],  B[k][n = lane%32],  C[r] -> row (r&3) + 8*(r>>2) + 4*(lane/32), col lane%32).
// The k -> dimension assignment is free as long as A and B agree, so the host lays the query bytes out in the order the
// code bits fall out of the packed words (fill_query_mfma in bbq_core.cpp).
//
// Operand roles (round 3): the QUERIES are the A operand (M = 32 queries) and the index ROWS the B operand (N = 32 rows), so a
// lane's 16 accumulators are 16 queries against ONE row (column lane%32) - its own row, or its half-wave partner's.  The row's
// pre-filter constants therefore sit in the lane's registers, and only the per-query constants (one float4, the same for the whole
// half-wave) come from LDS: 48 + 32 = 80 ds_read_b128 per tile and wave where the rows-as-A layout needed 48 + 128.
// What binds the kernel (profiles/r03_mfma_pmc.json, DESIGN.md "Shared sweeps"): VALU issue - 1 349 vector instructions per tile
// and wave next to 48 MFMAs, SQ_ACTIVE_INST_VALU = 88 % of the launch's SIMD time, the matrix cores 24 % busy.  The pre-filter
// (~16 instructions per pair) and the bit -> int8 expansion (9 per MFMA) are that load.
//
// Per (row, query) pair a cheap, provably conservative f32 pre-filter in "z-space" (the monotone argument of the
// similarity transform) rejects almost everything; the rare survivors go through the f64 bound and the exact f64 score of
// the one-sweep kernel, so the emitted candidates - and therefore the results - are identical.
#include <hip/hip_runtime.h>
#include <float.h>
#include "bbq_device.h"
#include "bbq_launch.h"

#pragma clang fp contract(off)

namespace bbq {

typedef uint32_t u32x4m __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2m __attribute__((ext_vector_type(2)));
typedef int i32x4m __attribute__((ext_vector_type(4)));
typedef int i32x16m __attribute__((ext_vector_type(16)));
typedef double f64x2m __attribute__((ext_vector_type(2)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

constexpr int kMfmaQueries = 32;

// ---- exact pieces shared with bbq_kernels.hip (kept textually identical: same operation order) ----------------------
__device__ __forceinline__ double m_js_max0(double x) { return (x != x) ? x : (x > 0.0 ? x : 0.0); }

__device__ __forceinline__ double m_score_f64(double qc, double ax, double ux, double xadd, double x1, const QueryParams &p) {
  const double lx = ux - ax;
  const double t1 = (ax * p.ay) * p.dimd;
  const double t2 = (p.ay * lx) * x1;
  const double t3 = (ax * p.ly) * p.y1;
  const double t4 = (lx * p.ly) * qc;
  const double s = ((t1 + t2) + t3) + t4;
  if (p.sim == 0) {
    const double e = (p.qadd + xadd) - (2.0 * s);
    return m_js_max0(1.0 / (1.0 + e));
  }
  const double t = p.one_bit ? (s + ((p.qadd + xadd) - p.cdp)) : (((s + p.qadd) + xadd) - p.cdp);
  if (p.sim == 1) return m_js_max0((1.0 + t) / 2.0);
  if (p.one_bit) return t < 0.0 ? 1.0 / (1.0 - t) : t + 1.0;
  const double FBS = 1.0 / 15.0;
  return t < 0.0 ? 1.0 / (1.0 - t / FBS) : t / FBS + 1.0;
}

// conservative lower edge, in z-space, of "score > theta" for one query (see the pre-filter below).
//   COSINE / MIP: z = s + xadd,  score = f(z + qadd - cdp) with f increasing
//   EUCLIDEAN   : z = 2s - xadd, score = 1/(1 + qadd - z)  increasing in z while the denominator is positive
// Returns zmin with:  exact f32 score > theta_score  =>  z > zmin.   -inf accepts everything.
__device__ __forceinline__ double z_threshold(uint32_t theta_key, const QueryParams &p) {
  if (theta_key == 0u) return -DBL_MAX;
  const uint32_t bits = (theta_key & 0x80000000u) ? (theta_key & 0x7fffffffu) : ~theta_key;
  const double th = (double)__uint_as_float(bits);  // the threshold score (a float the reference produced)
  if (!(th == th)) return -DBL_MAX;
  double z;
  if (p.sim == 1) {                 // max((1+t)/2, 0) > th  =>  t > 2 th - 1        (th >= 0 always for scores)
    if (th < 0.0) return -DBL_MAX;
    z = (2.0 * th - 1.0) - (p.qadd - p.cdp);
  } else if (p.sim == 2) {
    double t;
    if (p.one_bit) t = th >= 1.0 ? th - 1.0 : (th > 0.0 ? 1.0 - 1.0 / th : -DBL_MAX);
    else {
      const double FBS = 1.0 / 15.0;
      t = th >= 1.0 ? (th - 1.0) * FBS : (th > 0.0 ? (1.0 - 1.0 / th) * FBS : -DBL_MAX);
    }
    if (t == -DBL_MAX) return -DBL_MAX;
    z = t - (p.qadd - p.cdp);
  } else {                          // 1/(1+e) > th, e = qadd + xadd - 2s = qadd - z   =>  z > qadd + 1 - 1/th
    if (!(th > 0.0)) return -DBL_MAX;
    z = p.qadd + 1.0 - 1.0 / th;
  }
  if (!(fabs(z) <= DBL_MAX)) return -DBL_MAX;
  return z - 1e-9 * (fabs(z) + fabs(p.qadd) + fabs(p.cdp) + 1.0);  // rounding allowance of this inversion
}

struct MfmaArgs {
  ScanArgs s;
  const uint8_t *qbytes;   // [groups][W*4 words][2 halves][32 queries][16 B]  int8 query values in fragment order
  const float *qmax;       // [groups][4]: max |ay|, max |ly|, max y1, max |qadd - cdp| over the group's queries
  int32_t nq_total;
};

// Pairs that pass the pre-filter are pushed to a per-wave LDS queue (packed qc | row-in-tile << 20 | query << 26) and
// scored exactly afterwards by ONE copy of the exact code, 64 pairs at a time.
constexpr int kMfmaQueueCap = 512;
constexpr int kMfmaChunksPerBlock = 8;
template <int W, bool COMPACT>
__global__ __launch_bounds__(kChunkRows, 4) void bbq_scan_mfma_kernel(const MfmaArgs a) {  // 4 waves per SIMD = 2 workgroups per CU: caps the allocation at 128 VGPRs
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = kChunkRows;
  constexpr int NW = kChunkRows / 64;
  constexpr int WORDS = W * 4;
  u32x4m *s_B = reinterpret_cast<u32x4m *>(smem);                                    // [WORDS*2][32]
  f32x4m *s_qk = reinterpret_cast<f32x4m *>(smem + (size_t)WORDS * 2 * 32 * 16);     // [32] per-query pre-filter constants {cs*ay, cs*ly, y1, zth - margin}
  float *s_x1 = reinterpret_cast<float *>(s_qk + kMfmaQueries);                      // [NW][64] popcount of every tile row (exact in f32), for the survivors' exact scores
  uint32_t *s_queue = reinterpret_cast<uint32_t *>(s_x1 + NW * 64);                  // [NW][kMfmaQueueCap]
  uint32_t *s_qcount = s_queue + NW * kMfmaQueueCap;                                 // [NW] (+ padding to 16 B)
  QueryParams *s_qp = reinterpret_cast<QueryParams *>(s_qcount + 8);                 // [32]
  double *s_zth = reinterpret_cast<double *>(s_qp + kMfmaQueries);                   // [32]
  uint32_t *s_theta = reinterpret_cast<uint32_t *>(s_zth + kMfmaQueries);            // [32]
  uint32_t *s_cnt = s_theta + kMfmaQueries;                                          // [32]
  uint64_t *s_ent = reinterpret_cast<uint64_t *>(s_cnt + kMfmaQueries);              // [32][cap]

  const int group = blockIdx.y;
  const int q0 = group * kMfmaQueries;
  const int nb = min(kMfmaQueries, a.nq_total - q0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  {
    const u32x4m *__restrict__ gb = reinterpret_cast<const u32x4m *>(a.qbytes) + (size_t)group * WORDS * 2 * 32;
    for (int i = tid; i < WORDS * 2 * 32; i += NT) s_B[i] = gb[i];
    if (tid < kMfmaQueries) {
      QueryParams p{};
      uint32_t th = 0xFFFFFFFFu;  // lanes without a query accept nothing
      if (tid < nb) { p = a.s.qparams[q0 + tid]; th = a.s.theta[q0 + tid]; }
      s_qp[tid] = p;
      s_theta[tid] = th;
      const double zt = tid < nb ? z_threshold(th, p) : DBL_MAX;
      s_zth[tid] = zt;
      s_cnt[tid] = 0;
      // what the pre-filter compares with, per query: cs = 2 for EUCLIDEAN (z = 2s - xadd), 1 otherwise.  Lanes without a query get a
      // threshold nothing passes.  (float)zt rounds; the compare carries its own margin
      const float csf = p.sim == 0 ? 2.0f : 1.0f;
      const float zth = (float)zt;
      f32x4m qk;
      qk.x = csf * (float)p.ay; qk.y = csf * (float)p.ly; qk.z = (float)p.y1; qk.w = zth - 1e-6f * (fabsf(zth) + 1.0f);
      s_qk[tid] = qk;
    }
    if (tid < NW) s_qcount[tid] = 0;
  }
  __syncthreads();

  const int64_t n_tiles = (a.s.idx.n_rows + kTileRows - 1) / kTileRows;
  // A workgroup is persistent over kMfmaChunksPerBlock consecutive chunks of the same 32 queries: the query fragments are
  // staged once, and the codes of the next chunk's tile are prefetched into registers while the current one is computed.
  const int lc0 = blockIdx.x * kMfmaChunksPerBlock;
  u32x4m cnext[W];
  u32x2m ccnext = {0u, 0u};  // .x: this lane's bf16 pair, .y: the tile's additive-correction bound (min for EUCLIDEAN, max otherwise)
  {
    const int64_t t0 = (a.s.chunk_begin + lc0) * kTilesPerChunk + wave;
    if (lc0 < a.s.n_chunks && t0 < n_tiles) {
      const uint8_t *__restrict__ tp0 = a.s.idx.tiles + t0 * (int64_t)a.s.idx.tile_stride;
#pragma unroll
      for (int j = 0; j < W; ++j) cnext[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4m *>(tp0) + lane + j * kTileRows);
      if constexpr (COMPACT) {
        const uint8_t *cr0 = tp0 + (size_t)W * (kTileRows * 16);
        ccnext.x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(cr0) + lane);
        ccnext.y = __float_as_uint(a.s.idx.add_range[t0 * 2 + (s_qp[0].sim == 0 ? 0 : 1)]);
      }
    }
  }
#pragma unroll 1
  for (int ci = 0; ci < kMfmaChunksPerBlock; ++ci) {
  // opaque zero, redefined every iteration: keeps the compiler from hoisting the 24 query-fragment LDS reads out of
  // the chunk loop (96 VGPRs held across the loop -> 252 VGPRs, occupancy 2 and a slower kernel)
  int lds_off;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lds_off));
  const int lc = lc0 + ci;           // chunk index inside this launch
  if (lc >= a.s.n_chunks) break;     // block-uniform
  const int64_t chunk = a.s.chunk_begin + lc;
  const int64_t tile = chunk * kTilesPerChunk + wave;

  if (tile < n_tiles) {  // wave-uniform
    const int sim = s_qp[0].sim;            // uniform over the call (lanes without a query hold zeros)
    const float *__restrict__ gm = a.qmax + (size_t)group * 4;
    const float AYmax = gm[0], LYmax = gm[1], Y1max = gm[2];

    const uint8_t *__restrict__ tp = a.s.idx.tiles + tile * (int64_t)a.s.idx.tile_stride;
    const uint8_t *__restrict__ cr = tp + (size_t)W * (kTileRows * 16);
    const int64_t row_l = tile * kTileRows + lane;  // the row whose codes this lane loads
    const u32x4m *__restrict__ cp = reinterpret_cast<const u32x4m *>(tp) + lane;
    u32x4m c[W];
#pragma unroll
    for (int j = 0; j < W; ++j) c[j] = cnext[j];
    const u32x2m cc_cur = ccnext;
    {  // prefetch the next chunk's tile (same wave slot) while this one is computed
      const int64_t tn = tile + kTilesPerChunk;
      if (ci + 1 < kMfmaChunksPerBlock && lc + 1 < a.s.n_chunks && tn < n_tiles) {
        const uint8_t *__restrict__ tpn = a.s.idx.tiles + tn * (int64_t)a.s.idx.tile_stride;
#pragma unroll
        for (int j = 0; j < W; ++j) cnext[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4m *>(tpn) + lane + j * kTileRows);
        if constexpr (COMPACT) {
          const uint8_t *crn = tpn + (size_t)W * (kTileRows * 16);
          ccnext.x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(crn) + lane);
          ccnext.y = __float_as_uint(a.s.idx.add_range[tn * 2 + (s_qp[0].sim == 0 ? 0 : 1)]);
        }
      }
    }
    (void)cp;
    double al, au, aadd, ea, eu, eadd;
    if constexpr (COMPACT) {
      const u32x2m cc = cc_cur;
      al = (double)__uint_as_float(cc.x << 16);
      au = (double)__uint_as_float(cc.x & 0xffff0000u);
      aadd = (double)__uint_as_float(cc.y);
      const double rel = 0.0078125 * (1.0 + 1.0 / 65536.0);
      ea = fabs(al) * rel + 1e-37;
      eu = fabs(au) * rel + 1e-37;
      eadd = fabs(aadd) * 1.1920928955078125e-07 + 1e-37;
    } else {
      const f64x2m lu = __builtin_nontemporal_load(reinterpret_cast<const f64x2m *>(cr) + lane);
      al = lu.x; au = lu.y;
      aadd = __builtin_nontemporal_load(reinterpret_cast<const double *>(cr + 1024) + lane);
      // the f32 copies used by the pre-filter are rounded: 2^-24 relative
      ea = fabs(al) * 6e-8 + 1e-37; eu = fabs(au) * 6e-8 + 1e-37; eadd = fabs(aadd) * 6e-8 + 1e-37;
    }
    uint32_t ones = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) ones += __popc(c[j].x) + __popc(c[j].y) + __popc(c[j].z) + __popc(c[j].w);
    double x1row = (double)ones;  // quantizedComponentSum of a 1-bit row is its popcount ...
    if (a.s.idx.has_x1) x1row = reinterpret_cast<const double *>(cr + 1536)[lane];  // ... unless the index says otherwise
    // row constants of the pre-filter for MY row (tile row `lane`), in registers:
    //   k0 = {R1, D - x1, x1, al}   k1 = {lx, ca*add + slack, ea, eu}
    f32x4m k0, k1;
    {
      const double D = s_qp[0].dimd;   // same for every query of the batch
      const double x1 = x1row, lx = au - al;
      const double R1 = al * D + lx * x1;
      const double cs_d = sim == 0 ? 2.0 : 1.0, ca_d = sim == 0 ? -1.0 : 1.0;
      // f32 evaluation slack: 8 roundings of terms bounded with the group's largest query constants, doubled
      const double F = (double)AYmax * fabs(R1) + (double)LYmax * (double)Y1max * (fabs(al) + fabs(lx)) + fabs(aadd) + 1.0;
      const double slack = cs_d * (2e-6 * F + 1e-3 * (ea + eu) * ((double)AYmax * D + 2.0 * (double)LYmax * (double)Y1max)) + eadd * 1.001;
      k0.x = (float)R1; k0.y = (float)(D - x1); k0.z = (float)x1; k0.w = (float)al;
      // non-finite or huge rows: force a pass (an infinite slack makes every compare below fail to reject)
      const bool weird = !(fabs(R1) + fabs(al) + fabs(lx) + fabs(aadd) < 1e30);
      const float slack32 = weird ? __uint_as_float(0x7f800000u) : (float)slack * 1.001f + 1e-30f;
      k1.x = (float)lx; k1.y = (float)(ca_d * aadd) + slack32; k1.z = (float)(ea * 1.001); k1.w = (float)(eu * 1.001);   // unscaled: the pair loop carries cs in A and B
      // (float)(ca*aadd) + slack32 rounds once more: one extra ulp of |ca*aadd| is inside the 1.001 factors of slack (eadd part)
      s_x1[wave * 64 + lane] = k0.z;   // for the survivors' exact scores (any lane may score any row of the tile)
    }
    // ---- the contraction, one row group at a time: C[m = query][n = row of the group] over WORDS k-steps of 32 dims, then the
    //      pre-filter of its 32 x 32 pairs (a lane: 16 queries x its column's row); one accumulator tile live keeps the kernel under 128 VGPRs
    uint32_t *__restrict__ queue = s_queue + (size_t)wave * kMfmaQueueCap;
    const int rows_here = (int)min((int64_t)kTileRows, a.s.idx.n_rows - tile * kTileRows);
#pragma unroll 1
    for (int rg = 0; rg < 2; ++rg) {
      i32x16m acc = {0};
#pragma unroll
      for (int g = 0; g < WORDS; ++g) {
        const uint32_t w = (g & 3) == 0 ? c[g >> 2].x : (g & 3) == 1 ? c[g >> 2].y : (g & 3) == 2 ? c[g >> 2].z : c[g >> 2].w;
        // v_permlane32_swap(w, w): [0] = {own | partner (lane-32)}, [1] = {partner (lane+32) | own}: exactly the word of
        // tile row 32*rg + lane%32 (lanes of the other half borrow their partner's row)
        const auto sw2 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
        const uint32_t aw = (rg == 0 ? sw2[0] : sw2[1]) >> (4 * h);
        // this half supplies 16 of the word's 32 dims: bits 4h+c+8i -> byte i of dword c (one shift + one AND per dword;
        // the host lays the query bytes out in the same order, fill_query_mfma)
        i32x4m R;
        R.x = (int)(aw & 0x01010101u); R.y = (int)((aw >> 1) & 0x01010101u);
        R.z = (int)((aw >> 2) & 0x01010101u); R.w = (int)((aw >> 3) & 0x01010101u);
        const u32x4m bq = s_B[(g * 2 + h) * 32 + n + lds_off];
        i32x4m Q;
        Q.x = (int)bq.x; Q.y = (int)bq.y; Q.z = (int)bq.z; Q.w = (int)bq.w;
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Q, R, acc, 0, 0, 0);   // A = queries (m), B = rows (n)
      }
      // per (query, row) pre-filter; this lane owns tile row rit and 16 queries of the group
      const int rit = 32 * rg + n;
      const bool row_ok = rit < rows_here;
      // the constants of tile row rit: my own row when rg == h, otherwise my half-wave partner's (fetched here, per row group, so that
      // only one set is live).  swap(v, v): [0] = {own | partner(lane - 32)}, [1] = {partner(lane + 32) | own}
      f32x4m r0 = k0, r1 = k1;
      {  // every lane takes part in the swaps (a swap under a half-wave branch would read inactive lanes); the select follows
        const bool other = rg != h;
        const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(k0.x), __float_as_uint(k0.x), false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(k0.y), __float_as_uint(k0.y), false, false);
        const auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(k0.z), __float_as_uint(k0.z), false, false);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(k0.w), __float_as_uint(k0.w), false, false);
        const auto tx = __builtin_amdgcn_permlane32_swap(__float_as_uint(k1.x), __float_as_uint(k1.x), false, false);
        const auto ty = __builtin_amdgcn_permlane32_swap(__float_as_uint(k1.y), __float_as_uint(k1.y), false, false);
        const auto tz = __builtin_amdgcn_permlane32_swap(__float_as_uint(k1.z), __float_as_uint(k1.z), false, false);
        const auto tw = __builtin_amdgcn_permlane32_swap(__float_as_uint(k1.w), __float_as_uint(k1.w), false, false);
        if (other) {
          r0.x = __uint_as_float(h ? sx[0] : sx[1]); r0.y = __uint_as_float(h ? sy[0] : sy[1]);
          r0.z = __uint_as_float(h ? sz[0] : sz[1]); r0.w = __uint_as_float(h ? sw[0] : sw[1]);
          r1.x = __uint_as_float(h ? tx[0] : tx[1]); r1.y = __uint_as_float(h ? ty[0] : ty[1]);
          r1.z = __uint_as_float(h ? tz[0] : tz[1]); r1.w = __uint_as_float(h ? tw[0] : tw[1]);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;  // query of this accumulator
        const int qc = acc[r];
        const f32x4m qk = s_qk[m];                    // {cs*ay, cs*ly, y1, zth - margin}: the same address for the whole half-wave
        const float ayz = qk.x, lyz = qk.y, y1q = qk.z;
        const float qcf = (float)qc;
        const float u = fmaf(r1.x, qcf, r0.w * y1q);                 // al*y1 + lx*qc
        const float z = fmaf(lyz, u, fmaf(ayz, r0.x, r1.y));         // cs*(ay*R1 + ly*u) + ca*add + slack
        // the error terms in the same scaling: cs*A, cs*B against ea, eu (cs is 1 or 2: scaling by it commutes with every rounding,
        // so this is bit for bit |A|*(cs*ea) + |B|*(cs*eu) with A, B from the unscaled ay, ly - two multiplications per pair less)
        const float Ae = fmaf(lyz, y1q - qcf, ayz * r0.y);           // cs*(ay*(D-x1) + ly*(y1-qc))
        const float Be = fmaf(lyz, qcf, ayz * r0.z);                 // cs*(ay*x1 + ly*qc)
        const float zu = fmaf(fabsf(Ae), r1.z, fmaf(fabsf(Be), r1.w, z));
        // NaN anywhere => the compare fails => pass; an overflowed (infinite) zu proves nothing either: pass
        const bool pass = m < nb && row_ok && (!(zu <= qk.w) || !(fabsf(zu) <= 3.0e38f));
        if (pass) {
          const uint32_t slot = atomicAdd(&s_qcount[wave], 1u);
          if (slot < (uint32_t)kMfmaQueueCap) queue[slot] = (uint32_t)qc | ((uint32_t)rit << 20) | ((uint32_t)m << 26);
          else atomicOr(a.s.flags + q0 + m, kFlagOverflow);  // more survivors than the queue holds: this query goes dense
        }
      }
    }
    // ---- exact scores of the survivors (same-wave LDS traffic: program order is enough)
    const uint32_t n_pass = min(s_qcount[wave], (uint32_t)kMfmaQueueCap);
    for (uint32_t i = lane; i < n_pass; i += 64) {
      const uint32_t e = queue[i];
      const int qc = (int)(e & 0xFFFFFu), rit = (int)((e >> 20) & 63u), qn = (int)(e >> 26);
      const QueryParams pq = s_qp[qn];
      const int64_t row = tile * kTileRows + rit;
      double lo, up, ad;
      if constexpr (COMPACT) {
        const double *__restrict__ ex = a.s.idx.exact + row * 4;
        lo = ex[0]; up = ex[1]; ad = ex[2];
      } else {
        lo = reinterpret_cast<const double *>(cr)[2 * rit];
        up = reinterpret_cast<const double *>(cr)[2 * rit + 1];
        ad = reinterpret_cast<const double *>(cr + 1024)[rit];
      }
      double x1d = (double)s_x1[wave * 64 + rit];  // popcount of the row: exact in f32 (<= 2^24)
      if (a.s.idx.has_x1) x1d = reinterpret_cast<const double *>(cr + 1536)[rit];  // explicit sums may not be f32-exact
      const double s64 = m_score_f64((double)qc, lo, up, ad, x1d, pq);
      const float s32 = (float)s64;
      const uint32_t bits = __float_as_uint(s32);
      if (s32 != s32) atomicOr(a.s.flags + q0 + qn, kFlagNaN);
      else if (key_of_bits(bits) > s_theta[qn]) {
        const uint64_t ent = ((uint64_t)(uint32_t)(a.s.row_id_base + row) << 32) | bits;
        if (a.s.append_lists) {
          // append mode: straight into the query's list (unordered inside the segment; the finalize launch takes its keys from there
          // and the rare host replay sorts).  No per-chunk staging, so the waves of a workgroup never wait for each other (the slot
          // mode has three barriers per tile): 49.6 -> 53.6 K q/s at 10 M x 768
          const uint32_t slot = atomicAdd(a.s.append_counts + q0 + qn, 1u);
          const int64_t at = (int64_t)a.s.append_base[2 * (q0 + qn)] + slot;
          if (at < a.s.append_cap) a.s.append_lists[(size_t)(q0 + qn) * a.s.append_cap + at] = ent;
          else atomicOr(a.s.flags + q0 + qn, kFlagOverflow);
        } else {
          const uint32_t slot = atomicAdd(&s_cnt[qn], 1u);
          if (slot < (uint32_t)a.s.cap) s_ent[(size_t)qn * a.s.cap + slot] = ent;
        }
      }
    }
    (void)row_l;
    if (a.s.append_lists) s_qcount[wave] = 0;  // this wave's queue is its own: ready for its next tile
  }
  if (a.s.append_lists) continue;  // workgroup-uniform: nothing to flush, nobody to wait for
  __syncthreads();
  for (int b = wave; b < nb; b += NW) {  // each wave writes the lists of its share of the queries
    uint32_t cnt = s_cnt[b];
    if (cnt > (uint32_t)a.s.cap) {
      if (lane == 0) atomicOr(a.s.flags + q0 + b, kFlagOverflow);
      cnt = (uint32_t)a.s.cap;
    }
    const uint64_t *__restrict__ src = s_ent + (size_t)b * a.s.cap;
    uint64_t *__restrict__ out = a.s.entries + ((size_t)(q0 + b) * a.s.n_chunks + lc) * (size_t)a.s.cap;
    for (uint32_t i = lane; i < cnt; i += 64) {
      const uint64_t e = src[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < cnt; ++j) rank += (src[j] < e) ? 1u : 0u;
      out[rank] = e;
    }
    if (lane == 0) a.s.counts[(size_t)(q0 + b) * a.s.n_chunks + lc] = cnt;
  }
  __syncthreads();                       // everybody has read the counters of this chunk ...
  if (tid < kMfmaQueries) s_cnt[tid] = 0;
  if (tid < NW) s_qcount[tid] = 0;
  __syncthreads();                       // ... and sees them cleared before the next chunk's survivors arrive
  }  // chunks of this workgroup
}

// ---------------------------------------------------------------------------------------------------------------------

template <int W, bool COMPACT>
static hipError_t launch_mfma_t(const MfmaArgs &a, int nq, int nc, hipStream_t s) {
  constexpr int NW = kChunkRows / 64;
  const size_t smem = (size_t)W * 4 * 2 * 32 * 16 + (size_t)kMfmaQueries * 16 + (size_t)NW * 64 * 4 + (size_t)NW * kMfmaQueueCap * 4 + 32 +
                      kMfmaQueries * (sizeof(QueryParams) + 8 + 4 + 4) + (size_t)kMfmaQueries * a.s.cap * 8 + 64;
  dim3 grid((unsigned)((nc + kMfmaChunksPerBlock - 1) / kMfmaChunksPerBlock), (unsigned)((nq + kMfmaQueries - 1) / kMfmaQueries), 1),
      block(kChunkRows, 1, 1);
  auto kern = bbq_scan_mfma_kernel<W, COMPACT>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, block, smem, s, a);
  return hipGetLastError();
}

bool mfma_sweep_supported(const ScanArgs &a) {
  const int w = a.idx.w16;
  if (a.idx.store_bits != 1 || !(w == 1 || w == 6 || w == 8 || w == 12)) return false;
  const size_t smem = (size_t)w * 4 * 2 * 32 * 16 + (size_t)(kChunkRows / 64) * (64 * 2 * 16 + kMfmaQueueCap * 4) + 4096 + (size_t)kMfmaQueries * a.cap * 8;
  return smem <= 150 * 1024;
}

hipError_t launch_scan_mfma(const ScanArgs &sa, const uint8_t *qbytes, const float *qmax, int n_queries, int n_chunks, hipStream_t s) {
  if (n_chunks <= 0 || n_queries <= 0) return hipSuccess;
  MfmaArgs a{sa, qbytes, qmax, n_queries};
  const bool compact = sa.idx.layout == kLayoutCompact;
  switch (sa.idx.w16) {
    case 1: return compact ? launch_mfma_t<1, true>(a, n_queries, n_chunks, s) : launch_mfma_t<1, false>(a, n_queries, n_chunks, s);
    case 6: return compact ? launch_mfma_t<6, true>(a, n_queries, n_chunks, s) : launch_mfma_t<6, false>(a, n_queries, n_chunks, s);
    case 8: return compact ? launch_mfma_t<8, true>(a, n_queries, n_chunks, s) : launch_mfma_t<8, false>(a, n_queries, n_chunks, s);
    case 12: return compact ? launch_mfma_t<12, true>(a, n_queries, n_chunks, s) : launch_mfma_t<12, false>(a, n_queries, n_chunks, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bbq
