// bbq_latency_kernels.hip - the sweeps of the single-query call (the reference's own call shape: one synchronous
// searchNearestNeighbors per query, src/binaryQuantizationFormat.ts:308-412).
//
//   bbq_lat_scan_kernel     the sweep of one segment: exactly the rows / bound test / exact scores / candidates of bbq_scan_kernel in
//                           append mode, but the query arrives in the KERNEL ARGUMENTS of every launch: no host-to-device copy stands
//                           in front of the chain (bbq_core.cpp search_latency_chain; the answer leaves through mapped host memory
//                           the same way, written by the last finalize launch).  The first launch covers the dense prefix: its
//                           threshold is zero, so every row is scored exactly and listed (the reference heap is still filling there).
//
// Tried and dropped (round 3, measured on MI355X): the workgroup that finishes LAST doing the finalize launch's work in place
// (thresholds / final selection inside the sweep, agent-scope exchange of the candidates).  Exact - it passed the parity tests - but
// slower: the election needs the returned value of a device-scope atomic (2.5 us during which a finished workgroup keeps its wave
// slots: the 10 M-row sweep went from 157 to 178 us), and the selection itself is latency-bound per wave, so 512 threads take twice
// as long over it as the finalize kernel's 1024 (11 vs 6 us).  0.265 ms per call against 0.24.
#include <hip/hip_runtime.h>
#include "bbq_device.h"
#include "bbq_kernel_common.h"
#include "bbq_launch.h"

#pragma clang fp contract(off)

namespace bbq {

constexpr int kLatThreads = kChunkRows;          // scan workgroup

// one segment of the single-query sweep: the sparse path of bbq_scan_kernel in append mode (same rows, same bound test, same exact
// scores, same candidates) with the query in the kernel arguments
template <int QB, int W, bool COMPACT>
__global__ __launch_bounds__(kLatThreads) void bbq_lat_scan_kernel(const LatScanArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4 s_planes[W * QB];
  __shared__ uint64_t s_ent[kChunkRows];
  __shared__ uint32_t s_misc[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    for (int i = tid; i < W * QB; i += kLatThreads) {  // the query: kernel arguments -> LDS
      const uint4 v = a.planes[i];
      const u32x4 u = {v.x, v.y, v.z, v.w};
      s_planes[i] = u;
    }
    if (tid == 0) s_misc[14] = 0;  // candidates of this workgroup
  }
  const QueryParams p = a.p;
  // the dense prefix runs with threshold 0: below the key of every number, so each of its rows is scored exactly and listed
  const uint32_t theta = a.first ? 0u : *a.theta;
  __syncthreads();

  const int64_t chunk = a.chunk_begin + blockIdx.x;
  const int64_t n_tiles = (a.idx.n_rows + kTileRows - 1) / kTileRows;
  const int64_t tile = chunk * kTilesPerChunk + wave;
  bool nan_seen = false;
  if (tile < n_tiles) {  // wave-uniform
    const uint8_t *__restrict__ tp = a.idx.tiles + tile * (int64_t)a.idx.tile_stride;
    const int64_t row = tile * kTileRows + lane;
    const bool valid = row < a.idx.n_rows;
    f64x2 lu = {0.0, 0.0};
    double xadd = 0.0, x1 = 0.0;
    uint32_t cpk0 = 0, cpk1 = 0;
    u32x4 c[W];
    load_tile<W, COMPACT ? 1 : 2>(tp, lane, a.idx.has_x1 != 0, chunk_is_resident(chunk, a.idx), a.idx.nt_delta, c, cpk0, lu, xadd, x1);
    if constexpr (COMPACT) cpk1 = __float_as_uint(a.idx.add_range[tile * 2 + (p.sim == 0 ? 0 : 1)]);
    uint32_t acc[QB], ones, qc = 0;
    tile_popcounts<QB, W>(c, s_planes, acc, ones);
#pragma unroll
    for (int pl = 0; pl < QB; ++pl) qc += acc[pl] << pl;
    if (!a.idx.has_x1) x1 = (double)ones;
    bool need_exact = true;
    if constexpr (COMPACT) {
      const double al = (double)__uint_as_float(cpk0 << 16);
      const double au = (double)__uint_as_float(cpk0 & 0xffff0000u);
      const double aadd = (double)__uint_as_float(cpk1);
      const double ub = score_upper_bound((double)qc, al, au, aadd, x1, p);
      const float ub32 = (float)ub;
      need_exact = valid && ((ub32 != ub32) || key_of_bits(__float_as_uint(ub32)) > theta);
      if (need_exact) {
        const f64x2 *__restrict__ ex = reinterpret_cast<const f64x2 *>(a.idx.exact + row * 4);
        lu = ex[0];
        xadd = reinterpret_cast<const double *>(ex + 1)[0];
      }
    }
    if (need_exact) {
      const double s64 = score_f64((double)qc, lu.x, lu.y, xadd, x1, p);
      const float s32 = (float)s64;
      const uint32_t bits = __float_as_uint(s32);
      if (valid && (s32 != s32)) nan_seen = true;
      if (valid && (s32 == s32) && key_of_bits(bits) > theta) {
        const uint32_t slot = atomicAdd(&s_misc[14], 1u);
        s_ent[slot] = ((uint64_t)(uint32_t)(a.row_id_base + row) << 32) | bits;  // at most one per thread: slot < kChunkRows
      }
    }
  }
  if (__any(nan_seen) && lane == 0) atomicOr(a.flags, kFlagNaN);
  __syncthreads();
  const uint32_t cnt = s_misc[14];
  if (cnt) {  // workgroup-uniform
    if (tid == 0) s_misc[15] = atomicAdd(a.append_count, cnt);
    __syncthreads();
    const int64_t at = (a.first ? 0 : (int64_t)a.list_counts[0]) + s_misc[15];
    if (at + cnt > a.list_cap) {
      if (tid == 0) atomicOr(a.flags, kFlagOverflow);
    } else {
      for (uint32_t i = tid; i < cnt; i += kLatThreads) a.list[at + i] = s_ent[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pre-sampled threshold, step 1: exact scores of the first rows (the dense paths of bbq_scan_kernel), the per_wave largest keys of
// every wave's 64 rows
template <int QB, int W, bool COMPACT>
__global__ __launch_bounds__(kLatThreads) void bbq_lat_pre_kernel(const LatPreArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4 s_planes[W * QB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < W * QB; i += kLatThreads) {
    const uint4 v = a.planes[i];
    const u32x4 u = {v.x, v.y, v.z, v.w};
    s_planes[i] = u;
  }
  const QueryParams p = a.p;
  __syncthreads();
  const int64_t tile = (int64_t)blockIdx.x * kTilesPerChunk + wave;
  const int64_t row = tile * kTileRows + lane;
  const bool valid = row < (int64_t)a.rows && row < a.idx.n_rows;
  const uint8_t *__restrict__ tp = a.idx.tiles + tile * (int64_t)a.idx.tile_stride;
  uint32_t key = 0;
  if (tile * kTileRows < (int64_t)a.rows) {  // wave-uniform
    f64x2 lu = {0.0, 0.0};
    double xadd = 0.0, x1 = 0.0;
    uint32_t unused = 0;
    u32x4 c[W];
    // the prefix the threshold is sampled from is also the part of the index that stays cache-resident
    load_tile<W, COMPACT ? 0 : 2>(tp, lane, a.idx.has_x1 != 0, chunk_is_resident((int64_t)blockIdx.x, a.idx), a.idx.nt_delta, c, unused, lu, xadd, x1);
    if constexpr (COMPACT) {
      const f64x2 *__restrict__ ex = reinterpret_cast<const f64x2 *>(a.idx.exact + (valid ? row : 0) * 4);
      lu = BBQ_STREAM_LOAD(ex);
      xadd = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(ex + 1));
    }
    uint32_t acc[QB], ones, qc = 0;
    tile_popcounts<QB, W>(c, s_planes, acc, ones);
#pragma unroll
    for (int pl = 0; pl < QB; ++pl) qc += acc[pl] << pl;
    if (!a.idx.has_x1) x1 = (double)ones;
    const float s32 = (float)score_f64((double)qc, lu.x, lu.y, xadd, x1, p);
    const bool nan = valid && (s32 != s32);
    if (__any(nan) && lane == 0) atomicOr(a.flags, kFlagNaN);
    if (valid && !nan) key = key_of_bits(__float_as_uint(s32));
  }
  // the per_wave largest keys of the wave, one per round: wave maximum, kept by lane `t`, one holder of it retires
  uint32_t out = 0;
  for (int t = 0; t < a.per_wave; ++t) {
    uint32_t m = key;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == t) out = m;
    const unsigned long long holders = __ballot(key == m);
    if (lane == __ffsll((long long)holders) - 1) key = 0;
  }
  if (lane < a.per_wave) a.pre_keys[(tile * a.per_wave) + lane] = out;
}

// step 2: theta := the rank-th largest of the sampled keys (0: fewer than rank keys, everything stays a candidate)
__global__ __launch_bounds__(kFinalizeThreads) void bbq_lat_select_kernel(const uint32_t *__restrict__ pre_keys, int n_keys, int rank, uint32_t *theta) {
  __shared__ uint32_t s_keys[kLatPreKeys];
  __shared__ uint32_t s_hist[512];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_misc[16];
  const int tid = threadIdx.x;
#pragma unroll 4
  for (int i = tid; i < n_keys; i += kFinalizeThreads) s_keys[i] = pre_keys[i];
  __syncthreads();
  uint32_t th = 0;
  if (n_keys >= rank) th = block_select_kth_largest(s_keys, (uint32_t)n_keys, (uint32_t)rank, s_hist, s_wave, s_misc);  // uniform
  if (tid == 0) *theta = th;
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int QB, bool COMPACT>
static hipError_t lat_pre_w(const LatPreArgs &a, hipStream_t s) {
  const dim3 grid((unsigned)(a.rows / kChunkRows)), block(kLatThreads);
  switch (a.idx.w16) {
    case 6: hipLaunchKernelGGL((bbq_lat_pre_kernel<QB, 6, COMPACT>), grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((bbq_lat_pre_kernel<QB, 8, COMPACT>), grid, block, 0, s, a); break;
    case 12: hipLaunchKernelGGL((bbq_lat_pre_kernel<QB, 12, COMPACT>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_lat_pre(const LatPreArgs &a, int planes, hipStream_t s) {
  if (a.rows <= 0 || a.rows % kChunkRows != 0 || a.per_wave < 1 || a.per_wave > 4) return hipErrorInvalidValue;
  const bool c = a.idx.layout == kLayoutCompact;
  switch (planes) {
    case 1: return c ? lat_pre_w<1, true>(a, s) : lat_pre_w<1, false>(a, s);
    case 2: return c ? lat_pre_w<2, true>(a, s) : lat_pre_w<2, false>(a, s);
    case 4: return c ? lat_pre_w<4, true>(a, s) : lat_pre_w<4, false>(a, s);
    case 8: return c ? lat_pre_w<8, true>(a, s) : lat_pre_w<8, false>(a, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_lat_select(const uint32_t *pre_keys, int n_keys, int rank, uint32_t *theta, hipStream_t s) {
  if (n_keys < 0 || n_keys > kLatPreKeys || rank < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bbq_lat_select_kernel, dim3(1), dim3(kFinalizeThreads), 0, s, pre_keys, n_keys, rank, theta);
  return hipGetLastError();
}

template <int QB, bool COMPACT>
static hipError_t lat_scan_w(const LatScanArgs &a, hipStream_t s) {
  const dim3 grid((unsigned)a.n_chunks), block(kLatThreads);
  switch (a.idx.w16) {
    case 6: hipLaunchKernelGGL((bbq_lat_scan_kernel<QB, 6, COMPACT>), grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((bbq_lat_scan_kernel<QB, 8, COMPACT>), grid, block, 0, s, a); break;
    case 12: hipLaunchKernelGGL((bbq_lat_scan_kernel<QB, 12, COMPACT>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

bool latency_path_supported(const IndexView &v, int planes) {
  return v.store_bits == 1 && (v.w16 == 6 || v.w16 == 8 || v.w16 == 12) && (planes == 1 || planes == 2 || planes == 4 || planes == 8) &&
         v.w16 * planes <= kLatPlaneMax;
}

hipError_t launch_lat_scan(const LatScanArgs &a, int planes, hipStream_t s) {
  if (a.n_chunks <= 0) return hipErrorInvalidValue;
  const bool c = a.idx.layout == kLayoutCompact;
  switch (planes) {
    case 1: return c ? lat_scan_w<1, true>(a, s) : lat_scan_w<1, false>(a, s);
    case 2: return c ? lat_scan_w<2, true>(a, s) : lat_scan_w<2, false>(a, s);
    case 4: return c ? lat_scan_w<4, true>(a, s) : lat_scan_w<4, false>(a, s);
    case 8: return c ? lat_scan_w<8, true>(a, s) : lat_scan_w<8, false>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bbq
