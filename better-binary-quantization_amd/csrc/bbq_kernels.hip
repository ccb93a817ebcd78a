// bbq_kernels.hip - hand-written gfx950 (CDNA4) kernels of the binary-quantized scan + top-k path.
//
//   bbq_scan_kernel      the hot kernel: streams tile records from HBM with coalesced 16-byte loads
//                        (lane r owns row r of a 64-row tile: no cross-lane reduction at all), ANDs them
//                        with the query bit-planes staged once per workgroup in LDS, accumulates popcounts
//                        per plane, evaluates the reference's float64 score formula in the reference's
//                        operation order, rounds to f32 and either stores every score (DENSE) or appends
//                        the rows above the per-query threshold to the chunk's candidate slots.
//                        Replaces createDirectPackedBuffer + computeBatchFourBitDotProductDirectPacked /
//                        computeBatchDotProductDirectPacked + computeBatch{FourBit,OneBit}SimilarityScores
//                        (reference src/batchDotProduct.ts:22-49,420-436,478-617,
//                        src/utils/computeBatchFourBitDotProductDirectPacked.ts:10-53) and the f32 store of
//                        src/binaryQuantizationFormat.ts:353,378.
//   bbq_finalize_kernel  per query: compacts the candidate slots of one launch into the row-ordered
//                        candidate list, and radix-selects the k-th largest key seen so far = the threshold
//                        of the next segment (a lower bound of the reference heap's minimum at that point,
//                        src/binaryQuantizationFormat.ts:387-400).
//   bbq_retile_kernel    index build: row-major packed rows + corrections -> tile records.
//
// HBM-bound integer/byte work: no MFMA.  Compiled with -ffp-contract=off; the pragma below repeats it.
#include <hip/hip_runtime.h>
#include "bbq_device.h"
#include "bbq_kernel_common.h"
#include "bbq_launch.h"

#pragma clang fp contract(off)

namespace bbq {

// Multi-bit index rows (indexBits > 1): qcDist = sum_d q[d] * x[d] (computeQuantizedDotProduct, src/bitwiseDotProduct.ts:14-30)
// over SB-bit fields with the packed-nibble / packed-byte dot instructions - 8 (v_dot8_u32_u4) or 4 (v_dot4_u32_u8) exact
// integer products per lane and instruction.  2-bit fields are unfolded in registers into two dwords of nibbles (even / odd
// fields: one AND, one shift + AND); query values above 15 (QB == 8) are split into low and high nibbles,
// q = lo + 16 hi, so the dot is dot(x, lo) + 16 dot(x, hi).  s_q holds, per row dword, the matching query dwords
// (query_units_per_chunk; written by the host in exactly this order).  `sum` = the row's component sum (= quantizedComponentSum
// of a freshly quantized row, src/optimizedScalarQuantizer.ts:204-209).
template <int QB, int SB>
__device__ __forceinline__ void dot_chunk_multibit(const u32x4 c, const uint32_t *__restrict__ sq, uint32_t &lo, uint32_t &hi, uint32_t &sum) {
  const uint32_t x[4] = {c.x, c.y, c.z, c.w};
  constexpr int QN = query_units_per_chunk(QB, SB);  // query dwords per row dword
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t *__restrict__ qw = sq + t * QN;
    if constexpr (SB == 2) {
      const uint32_t e = x[t] & 0x33333333u, o = (x[t] >> 2) & 0x33333333u;
      lo = __builtin_amdgcn_udot8(e, qw[0], lo, false);
      lo = __builtin_amdgcn_udot8(o, qw[1], lo, false);
      if constexpr (QB > 4) {
        hi = __builtin_amdgcn_udot8(e, qw[2], hi, false);
        hi = __builtin_amdgcn_udot8(o, qw[3], hi, false);
      }
      sum = __builtin_amdgcn_udot8(e + o, 0x11111111u, sum, false);  // nibbles of e + o are at most 6
    } else if constexpr (SB == 4) {
      lo = __builtin_amdgcn_udot8(x[t], qw[0], lo, false);
      if constexpr (QB > 4) hi = __builtin_amdgcn_udot8(x[t], qw[1], hi, false);
      sum = __builtin_amdgcn_udot8(x[t], 0x11111111u, sum, false);
    } else {
      lo = __builtin_amdgcn_udot4(x[t], qw[0], lo, false);
      sum = __builtin_amdgcn_udot4(x[t], 0x01010101u, sum, false);
    }
  }
}

template <int QB, int W, int SB>
__device__ __forceinline__ void tile_dot_multibit(const u32x4 (&c)[W], const u32x4 *__restrict__ s_planes, uint32_t &qc, uint32_t &sum) {
  const uint32_t *__restrict__ sq = reinterpret_cast<const uint32_t *>(s_planes);
  constexpr int QN = query_units_per_chunk(QB, SB);
  uint32_t lo = 0, hi = 0;
  sum = 0;
#pragma unroll
  for (int j = 0; j < W; ++j) dot_chunk_multibit<QB, SB>(c[j], sq + j * 4 * QN, lo, hi, sum);
  qc = lo + (hi << 4);
}
template <int QB, int SB>
__device__ __forceinline__ void tile_dot_multibit_any(const uint8_t *__restrict__ tp, int lane, int w16, const u32x4 *__restrict__ s_planes,
                                                      uint32_t &qc, uint32_t &sum) {
  const u32x4 *__restrict__ cp = reinterpret_cast<const u32x4 *>(tp) + lane;
  const uint32_t *__restrict__ sq = reinterpret_cast<const uint32_t *>(s_planes);
  constexpr int QN = query_units_per_chunk(QB, SB);
  uint32_t lo = 0, hi = 0;
  sum = 0;
  for (int j = 0; j < w16; ++j) {
    const u32x4 c = BBQ_STREAM_LOAD(cp + j * kTileRows);
    dot_chunk_multibit<QB, SB>(c, sq + j * 4 * QN, lo, hi, sum);
  }
  qc = lo + (hi << 4);
}

// MODE: 0 sparse / inline corrections, 1 dense / inline, 2 sparse / compact corrections + exact gather, 3 dense / compact
// grid = (chunks of kChunkRows rows, queries); block = kChunkRows/64 waves: wave w handles tile w of its chunk (one row per lane)
// SB = bits per stored field: 1 = packed 1-bit rows (QB bit-planes of the query), 2 / 4 / 8 = multi-bit rows (QB = 4: query values
// <= 15, QB = 8: any)
template <int QB, int W, int MODE, int SB = 1>
__global__ __launch_bounds__(kChunkRows) void bbq_scan_kernel(const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = kChunkRows;
  constexpr bool DENSE = (MODE & 1) != 0;
  constexpr bool COMPACT = (MODE & 2) != 0;
  constexpr int QU = query_units_per_chunk(QB, SB);
  const int w16 = W > 0 ? W : a.idx.w16;
  u32x4 *s_planes = reinterpret_cast<u32x4 *>(smem);
  uint64_t *s_ent = reinterpret_cast<uint64_t *>(smem + (size_t)w16 * QU * 16);
  // with a flood tier every passing row of the chunk is staged (it may have to move to the overflow area as a whole)
  const uint32_t stage_cap = DENSE ? 0u : ((a.ovf || a.append_lists) ? (uint32_t)kChunkRows : (uint32_t)a.cap);
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_ent + stage_cap);

  const int q = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  {  // stage the query bit-planes once per workgroup
    const u32x4 *__restrict__ gp = reinterpret_cast<const u32x4 *>(a.qplanes) + (size_t)q * w16 * QU;
    for (int i = tid; i < w16 * QU; i += NT) s_planes[i] = gp[i];
    if (!DENSE && tid == 0) *s_cnt = 0;
  }
  const QueryParams p = a.qparams[q];
  const uint32_t theta = DENSE ? 0u : a.theta[q];
  __syncthreads();

  const int64_t chunk = a.chunk_begin + blockIdx.x;
  const int64_t n_tiles = (a.idx.n_rows + kTileRows - 1) / kTileRows;
  const int64_t tile = chunk * kTilesPerChunk + wave;
  bool nan_seen = false;

  if (tile < n_tiles) {  // wave-uniform
    const uint8_t *__restrict__ tp = a.idx.tiles + tile * (int64_t)a.idx.tile_stride;
    const uint8_t *__restrict__ cr = tp + (size_t)w16 * (kTileRows * 16);
    const int64_t row = tile * kTileRows + lane;
    const bool valid = row < a.idx.n_rows;
    const bool resident = chunk_is_resident(chunk, a.idx);

    // every load of the tile is issued up front: the row's code chunks and its corrections
    f64x2 lu = {0.0, 0.0};
    double xadd = 0.0, x1 = 0.0;
    uint32_t cpk0 = 0, cpk1 = 0;
    uint32_t qc = 0, ones;
    constexpr int CORR = !COMPACT ? 2 : (DENSE ? 0 : 1);
    if constexpr (W > 0) {
      u32x4 c[W];
      load_tile<W, CORR>(tp, lane, a.idx.has_x1 != 0, resident, a.idx.nt_delta, c, cpk0, lu, xadd, x1);
      if constexpr (COMPACT && DENSE) {
        const f64x2 *__restrict__ ex = reinterpret_cast<const f64x2 *>(a.idx.exact + row * 4);
        lu = BBQ_STREAM_LOAD(ex);
        xadd = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(ex + 1));
      }
      // the tile's additive-correction range: EUCLIDEAN scores fall with it (take the minimum), the others rise (maximum)
      if constexpr (COMPACT && !DENSE) cpk1 = __float_as_uint(a.idx.add_range[tile * 2 + (p.sim == 0 ? 0 : 1)]);
      if constexpr (SB == 1) {
        uint32_t acc[QB];
        tile_popcounts<QB, W>(c, s_planes, acc, ones);
#pragma unroll
        for (int pl = 0; pl < QB; ++pl) qc += acc[pl] << pl;
      } else {
        tile_dot_multibit<QB, W, SB>(c, s_planes, qc, ones);
      }
    } else {  // a row width without a compiled kernel: streamed chunk by chunk
      if constexpr (!COMPACT) {
        lu = BBQ_STREAM_LOAD(reinterpret_cast<const f64x2 *>(cr) + lane);
        xadd = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(cr + 1024) + lane);
        if (a.idx.has_x1) x1 = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(cr + 1536) + lane);
      } else if constexpr (DENSE) {
        const f64x2 *__restrict__ ex = reinterpret_cast<const f64x2 *>(a.idx.exact + row * 4);
        lu = BBQ_STREAM_LOAD(ex);
        xadd = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(ex + 1));
      } else {
        cpk0 = BBQ_STREAM_LOAD(reinterpret_cast<const uint32_t *>(cr) + lane);
        cpk1 = __float_as_uint(a.idx.add_range[tile * 2 + (p.sim == 0 ? 0 : 1)]);
      }
      if constexpr (SB == 1) {
        uint32_t acc[QB];
        tile_popcounts_any<QB>(tp, lane, w16, s_planes, acc, ones);
#pragma unroll
        for (int pl = 0; pl < QB; ++pl) qc += acc[pl] << pl;
      } else {
        tile_dot_multibit_any<QB, SB>(tp, lane, w16, s_planes, qc, ones);
      }
    }
    if (!a.idx.has_x1) x1 = (double)ones;  // quantizedComponentSum of a freshly quantized row is its popcount / component sum

    bool need_exact = true;
    if constexpr (COMPACT && !DENSE) {
      const double al = (double)__uint_as_float(cpk0 << 16);
      const double au = (double)__uint_as_float(cpk0 & 0xffff0000u);
      const double aadd = (double)__uint_as_float(cpk1);
      // NaN (no bound) passes; otherwise the row can only matter if even its upper bound beats the threshold
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
      const float s32 = (float)s64;  // Float32Array store, src/binaryQuantizationFormat.ts:353,378
      const uint32_t bits = __float_as_uint(s32);
      if (valid && (s32 != s32)) nan_seen = true;
      if constexpr (DENSE) {
        if (valid) {
          const int64_t o = (int64_t)q * a.dense_stride + (row - a.chunk_begin * kChunkRows);
          if (a.dense_score32) a.dense_score32[o] = s32;
          if (a.dense_qcdist) a.dense_qcdist[o] = (int32_t)qc;
          if (a.dense_score64) a.dense_score64[o] = s64;
        }
      } else {
        if (valid && (s32 == s32) && key_of_bits(bits) > theta) {
          const uint32_t slot = atomicAdd(s_cnt, 1u);
          if (slot < stage_cap) s_ent[slot] = ((uint64_t)(uint32_t)(a.row_id_base + row) << 32) | bits;
        }
      }
    }
  }
  if (__any(nan_seen) && lane == 0) atomicOr(a.flags + q, kFlagNaN);

  if constexpr (!DENSE) {
    __syncthreads();
    if (a.append_lists) {  // workgroup-uniform
      const uint32_t cnt = *s_cnt;
      if (cnt == 0) return;
      __syncthreads();  // everyone has read *s_cnt
      if (tid == 0) *s_cnt = atomicAdd(a.append_counts + (size_t)q * kAppendStride, cnt);
      __syncthreads();
      const int64_t at = (int64_t)a.append_base[2 * q] + *s_cnt;
      if (at + cnt > a.append_cap) {
        if (tid == 0) atomicOr(a.flags + q, kFlagOverflow);
        return;
      }
      uint64_t *__restrict__ dst = a.append_lists + (size_t)q * a.append_cap + at;
      for (uint32_t i = tid; i < cnt; i += NT) dst[i] = s_ent[i];
      return;
    }
    uint32_t cnt = *s_cnt;
    uint64_t *__restrict__ slot0 = a.entries + ((size_t)q * a.n_chunks + blockIdx.x) * (size_t)a.cap;
    uint64_t *__restrict__ out = slot0;
    uint32_t count_word = cnt;
    if (cnt > (uint32_t)a.cap) {  // workgroup-uniform
      bool parked = false;
      if (a.ovf) {
        // flood (e.g. rows stored cluster by cluster and this is the query's cluster): one block of the query's overflow
        // area takes the whole chunk, so the list stays row-ordered and the query stays on the sparse path
        __syncthreads();  // everyone has read *s_cnt
        if (tid == 0) *s_cnt = atomicAdd(a.ovf_counts + q, cnt);
        __syncthreads();
        const uint32_t off = *s_cnt;
        if ((uint64_t)off + cnt <= (uint64_t)a.ovf_cap) {
          out = a.ovf + (size_t)q * a.ovf_cap + off;
          count_word = kCountRedirect | cnt;
          if (tid == 0) slot0[0] = off;
          parked = true;
        }
      }
      if (!parked) {
        if (tid == 0) atomicOr(a.flags + q, kFlagOverflow);
        cnt = min(cnt, (uint32_t)a.cap);
        count_word = cnt;
      }
    }
    for (uint32_t i = tid; i < cnt; i += NT) {  // rows are distinct: rank by counting puts them in row order
      const uint64_t e = s_ent[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < cnt; ++j) rank += (s_ent[j] < e) ? 1u : 0u;
      out[rank] = e;
    }
    if (tid == 0) a.counts[(size_t)q * a.n_chunks + blockIdx.x] = count_word;
  }
}

// ---------------------------------------------------------------------------------------------------
// Shared sweep (API extension; reported separately from the one-sweep-per-query metric): one workgroup
// scores its 1024 rows against NB queries.  The codes (and corrections) of a row are loaded ONCE into registers and
// reused for every query, so HBM traffic per query drops NB-fold and the kernel becomes VALU-bound (popcounts + f64
// bound/score per query).  Results are identical to NB separate sweeps: same thresholds, same slots, same lists.
template <int QB, int W, bool COMPACT, int NB>
__global__ __launch_bounds__(kChunkRows) void bbq_scan_shared_kernel(const ScanArgs a, const int nq_total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = kChunkRows;
  u32x4 *s_planes = reinterpret_cast<u32x4 *>(smem);                               // [NB][W*QB]
  QueryParams *s_qp = reinterpret_cast<QueryParams *>(smem + (size_t)NB * W * QB * 16);  // [NB]
  uint64_t *s_ent = reinterpret_cast<uint64_t *>(s_qp + NB);                        // [NB][cap]
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_ent + (size_t)NB * a.cap);       // [NB]
  uint32_t *s_theta = s_cnt + NB;                                                   // [NB]

  const int q0 = blockIdx.y * NB;
  const int nb = min(NB, nq_total - q0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  {
    const u32x4 *__restrict__ gp = reinterpret_cast<const u32x4 *>(a.qplanes) + (size_t)q0 * W * QB;
    for (int i = tid; i < nb * W * QB; i += NT) s_planes[i] = gp[i];
    if (tid < nb) {
      s_qp[tid] = a.qparams[q0 + tid];
      s_cnt[tid] = 0;
      s_theta[tid] = a.theta[q0 + tid];
    }
  }
  __syncthreads();

  const int64_t chunk = a.chunk_begin + blockIdx.x;
  const int64_t n_tiles = (a.idx.n_rows + kTileRows - 1) / kTileRows;
  const int64_t tile = chunk * kTilesPerChunk + wave;
  uint32_t nan_mask = 0;

  if (tile < n_tiles) {  // wave-uniform
    const uint8_t *__restrict__ tp = a.idx.tiles + tile * (int64_t)a.idx.tile_stride;
    const int64_t row = tile * kTileRows + lane;
    const bool valid = row < a.idx.n_rows;
    u32x4 c[W];
    f64x2 lu = {0.0, 0.0};
    double xadd = 0.0, x1 = 0.0, al = 0.0, au = 0.0, aadd = 0.0;
    uint32_t cw = 0;
    load_tile<W, COMPACT ? 1 : 2>(tp, lane, a.idx.has_x1 != 0, chunk_is_resident(chunk, a.idx), a.idx.nt_delta, c, cw, lu, xadd, x1);
    bool have_exact = !COMPACT;
    if constexpr (COMPACT) {
      al = (double)__uint_as_float(cw << 16);
      au = (double)__uint_as_float(cw & 0xffff0000u);
      // tile range of the additive correction; the queries of one call share the similarity function
      aadd = (double)a.idx.add_range[tile * 2 + (s_qp[0].sim == 0 ? 0 : 1)];
    }
    uint32_t ones = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) ones += popc4(c[j]);
    if (!a.idx.has_x1) x1 = (double)ones;

#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
      const u32x4 *__restrict__ pl = s_planes + (size_t)b * W * QB;
      uint32_t acc[QB];
#pragma unroll
      for (int pq = 0; pq < QB; ++pq) acc[pq] = 0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
#pragma unroll
        for (int pq = 0; pq < QB; ++pq) acc[pq] += popc4(c[j] & pl[j * QB + pq]);
      }
      uint32_t qc = 0;
#pragma unroll
      for (int pq = 0; pq < QB; ++pq) qc += acc[pq] << pq;
      const QueryParams p = s_qp[b];
      const uint32_t theta = s_theta[b];
      bool need_exact = valid;
      if constexpr (COMPACT) {
        const double ub = score_upper_bound((double)qc, al, au, aadd, x1, p);
        const float ub32 = (float)ub;
        need_exact = valid && ((ub32 != ub32) || key_of_bits(__float_as_uint(ub32)) > theta);
        if (need_exact && !have_exact) {
          const f64x2 *__restrict__ ex = reinterpret_cast<const f64x2 *>(a.idx.exact + row * 4);
          lu = ex[0];
          xadd = reinterpret_cast<const double *>(ex + 1)[0];
          have_exact = true;
        }
      }
      if (need_exact) {
        const double s64 = score_f64((double)qc, lu.x, lu.y, xadd, x1, p);
        const float s32 = (float)s64;
        const uint32_t bits = __float_as_uint(s32);
        if (s32 != s32) nan_mask |= 1u << b;
        if ((s32 == s32) && key_of_bits(bits) > theta) {
          const uint32_t slot = atomicAdd(&s_cnt[b], 1u);
          if (slot < (uint32_t)a.cap) s_ent[(size_t)b * a.cap + slot] = ((uint64_t)(uint32_t)(a.row_id_base + row) << 32) | bits;
        }
      }
    }
  }
  for (int b = 0; b < nb; ++b)
    if (__any((nan_mask >> b) & 1u) && lane == 0) atomicOr(a.flags + q0 + b, kFlagNaN);

  __syncthreads();
  for (int b = 0; b < nb; ++b) {
    uint32_t cnt = s_cnt[b];
    if (cnt > (uint32_t)a.cap) {
      if (tid == 0) atomicOr(a.flags + q0 + b, kFlagOverflow);
      cnt = (uint32_t)a.cap;
    }
    const uint64_t *__restrict__ src = s_ent + (size_t)b * a.cap;
    uint64_t *__restrict__ out = a.entries + ((size_t)(q0 + b) * a.n_chunks + blockIdx.x) * (size_t)a.cap;
    for (uint32_t i = tid; i < cnt; i += NT) {
      const uint64_t e = src[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < cnt; ++j) rank += (src[j] < e) ? 1u : 0u;
      out[rank] = e;
    }
    if (tid == 0) a.counts[(size_t)(q0 + b) * a.n_chunks + blockIdx.x] = cnt;
  }
}

// ---------------------------------------------------------------------------------------------------
// finalize: one workgroup (1024 threads) per query

__global__ __launch_bounds__(kFinalizeThreads) void bbq_finalize_kernel(const FinalizeArgs a) {
  // dynamic LDS: keys | histogram | scratch | copy jobs | chunk counters.  A launch that compacts chunk slots gets all of it
  // (kFinalizeLdsBytes, 146 KB); a launch behind an appending or a dense sweep uses neither the counters nor, beyond the 8 KB the final
  // selection sorts in, the job table, and is launched with kFinalizeLdsBytesSmall (58 KB): two of them fit a CU, and one fits beside a
  // workgroup of another slot's matrix-core sweep (measured: the dispatcher still serves the running sweep's workgroups first, so
  // the finalize launches of the other slots keep waiting for it - no gain there; launch_finalize)
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
  uint32_t *s_keys = reinterpret_cast<uint32_t *>(fin_smem);                      // [kFinalizeKeyCap]
  uint32_t *s_hist = s_keys + kFinalizeKeyCap;                                      // [2][256]
  uint32_t *s_wave = s_hist + 512;                                                  // [16]
  uint32_t *s_misc = s_wave + 16;                                                   // [16]: 0..7 this kernel's, 8..15 the key selection's
  uint32_t *s_jobs = s_misc + 16;                                                   // [3 * kFinalizeJobs] {chunk | redirect, list offset, source offset}
  uint16_t *s_counts = reinterpret_cast<uint16_t *>(s_jobs + 3 * kFinalizeJobs);    // [kFinalizeCountCap] chunk counters of the launch, 16 bits each
  const int q = blockIdx.x;
  const int tid = threadIdx.x;
  uint32_t flags = 0;
  // the query's control words, all fetched at once (each was a memory round trip of its own on the critical path of a launch that
  // lasts ten microseconds): the scan launch that wrote them is over, nothing but thread 0 below changes them
  const int64_t base = a.emit ? a.list_counts[2 * q] : 0;
  const uint32_t appended = a.append_counts ? a.append_counts[(size_t)q * kAppendStride] : 0u;
  const uint32_t tcount0 = (uint32_t)a.topk_counts[q];
  const uint32_t flags0 = a.flags[q];
  __syncthreads();  // every thread has read them before thread 0 rewrites them below
  uint64_t *__restrict__ list = a.lists + (size_t)q * a.list_cap;
  uint32_t m_new = 0;

  if (a.dense_rows > 0) {
    m_new = (uint32_t)a.dense_rows;
    const float *__restrict__ d = a.dense_score32 + (size_t)q * a.dense_stride;
    for (uint32_t i = tid; i < m_new; i += kFinalizeThreads) {
      const uint32_t bits = __float_as_uint(d[i]);
      if (i < (uint32_t)kFinalizeKeyCap) s_keys[i] = key_of_bits(bits);
      if (a.emit && base + i < a.list_cap) list[base + i] = ((uint64_t)(uint32_t)(a.dense_row_id_base + i) << 32) | bits;
    }
  } else if (a.append_counts) {
    // the scan launch appended its candidates to the list itself: only their keys are needed here
    m_new = appended;
    if (tid == 0) a.append_counts[(size_t)q * kAppendStride] = 0u;
    if (base + m_new > a.list_cap) m_new = 0;  // the workgroups that did not fit have flagged the query: it takes the dense path
#pragma unroll 4
    for (uint32_t j = tid; j < m_new && j < (uint32_t)kFinalizeKeyCap; j += kFinalizeThreads) s_keys[j] = key_of_bits((uint32_t)list[base + j]);
    __syncthreads();
  } else {
    // Compaction of the chunk slots into the row-ordered list.  Thread t owns cpt consecutive chunks; two block scans give it
    // its first list offset and its first JOB number (a job = one non-empty chunk), the jobs land in LDS in chunk order, and then
    // the ENTRIES are dealt to the threads one by one (binary search of the entry's job): every global load of the copy is
    // independent of every other, where a thread walking its chunks one after the other paid one memory latency per chunk
    // (31 us for the 19 K chunks of a 10 M-row segment, 5 us this way).
    const int cpt = (a.n_chunks + kFinalizeThreads - 1) / kFinalizeThreads;
    const int c0 = min(tid * cpt, a.n_chunks), c1 = min(c0 + cpt, a.n_chunks);
    const uint32_t *__restrict__ gcnts = a.counts + (size_t)q * a.n_chunks;
    // the chunk counters are read twice in chunk order by their owner; staged through LDS they are read from memory ONCE, coalesced
    // (a thread reading its 19 consecutive words from global memory, twice, was most of what was left of the compaction)
    const bool staged = a.n_chunks <= kFinalizeCountCap;
    if (staged) {  // 16 bits per counter: at most 512 candidates per chunk, bit 15 = the redirect flag
      for (int c = tid; c < a.n_chunks; c += kFinalizeThreads) {
        const uint32_t cw = gcnts[c];
        s_counts[c] = (uint16_t)((cw & 0x7fffu) | ((cw & kCountRedirect) ? 0x8000u : 0u));
      }
      __syncthreads();
    }
    auto count_word = [&](int c) -> uint32_t {
      if (!staged) return gcnts[c];
      const uint32_t h = s_counts[c];
      return (h & 0x7fffu) | ((h & 0x8000u) ? kCountRedirect : 0u);
    };
    uint32_t sum = 0, nz = 0;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
      const uint32_t n = count_word(c) & ~kCountRedirect;
      sum += n;
      nz += n ? 1u : 0u;
    }
    uint32_t total, total_jobs;
    uint32_t off = block_exclusive_scan_1024(sum, s_wave, total);
    uint32_t job = block_exclusive_scan_1024(nz, s_wave, total_jobs);
    m_new = total;
    if (total_jobs <= (uint32_t)kFinalizeJobs) {
      for (int c = c0; c < c1; ++c) {
        const uint32_t cw = count_word(c), n = cw & ~kCountRedirect;
        if (!n) continue;
        // a redirected chunk (flood tier) keeps its entries in a block of the query's overflow area; slot 0 says where
        const uint32_t src = (cw & kCountRedirect) ? (uint32_t)a.entries[((size_t)q * a.n_chunks + c) * (size_t)a.cap] : 0u;
        s_jobs[3 * job] = (uint32_t)c | (cw & kCountRedirect);
        s_jobs[3 * job + 1] = off;
        s_jobs[3 * job + 2] = src;
        off += n;
        ++job;
      }
      __syncthreads();
      for (uint32_t j = tid; j < total; j += kFinalizeThreads) {
        uint32_t lo = 0, hi = total_jobs;  // the last job whose list offset is <= j
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_jobs[3 * mid + 1] <= j) lo = mid; else hi = mid;
        }
        const uint32_t cw = s_jobs[3 * lo], i = j - s_jobs[3 * lo + 1];
        const uint64_t *__restrict__ e = (cw & kCountRedirect)
                                             ? a.ovf + (size_t)q * a.ovf_cap + s_jobs[3 * lo + 2]
                                             : a.entries + ((size_t)q * a.n_chunks + (cw & ~kCountRedirect)) * (size_t)a.cap;
        const uint64_t ent = e[i];
        if (a.emit && base + j < a.list_cap) list[base + j] = ent;
        if (j < (uint32_t)kFinalizeKeyCap) s_keys[j] = key_of_bits((uint32_t)ent);
      }
    } else {
      // more non-empty chunks than the job table holds (very long segments, floods): every thread copies its own chunks
      for (int c = c0; c < c1; ++c) {
        const uint32_t cw = count_word(c), n = cw & ~kCountRedirect;
        const uint64_t *__restrict__ e = a.entries + ((size_t)q * a.n_chunks + c) * (size_t)a.cap;
        if (cw & kCountRedirect) e = a.ovf + (size_t)q * a.ovf_cap + (uint32_t)e[0];
        for (uint32_t i = 0; i < n; ++i, ++off) {
          const uint64_t ent = e[i];
          if (a.emit && base + off < a.list_cap) list[base + off] = ent;
          if (off < (uint32_t)kFinalizeKeyCap) s_keys[off] = key_of_bits((uint32_t)ent);
        }
      }
    }
    __syncthreads();
  }
  if (a.emit) {
    if (base + m_new > a.list_cap) flags |= kFlagOverflow;
    if (tid == 0) a.list_counts[2 * q] = (int32_t)min((int64_t)(base + m_new), a.list_cap);
  }

  if (a.need_theta) {
    const uint32_t tcount = tcount0;
    uint32_t *__restrict__ tk = a.topk_keys + (size_t)q * a.k;
    {
      // more new keys than the LDS buffer holds (a flood): select among the ones that fit.  The k-th largest of any
      // SUBSET of the rows seen so far is a valid (weaker) lower bound of the reference heap's minimum, so the
      // query stays exact and on the sparse path
      const uint32_t m_use = min(m_new, (uint32_t)kFinalizeKeyCap - tcount);
      __syncthreads();  // a flood can have written new keys at indices >= m_use: those writes end before the running top-k lands there
      for (uint32_t i = tid; i < tcount; i += kFinalizeThreads) s_keys[m_use + i] = tk[i];
      const uint32_t M = m_use + tcount;
      __syncthreads();
      if (M < (uint32_t)a.k) {
        // fewer than k rows seen so far: the reference heap is still filling, everything stays a candidate
        for (uint32_t i = tid; i < M; i += kFinalizeThreads) tk[i] = s_keys[i];
        if (tid == 0) { a.topk_counts[q] = (int32_t)M; a.theta[q] = 0u; }
      } else {
        const uint32_t th = block_select_kth_largest(s_keys, M, (uint32_t)a.k, s_hist, s_wave, s_misc + 8);  // exactly the k-th largest key among the M keys
        if (tid == 0) s_misc[2] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < M; i += kFinalizeThreads) {
          const uint32_t key = s_keys[i];
          if (key > th) tk[atomicAdd(&s_misc[2], 1u)] = key;
        }
        __syncthreads();
        const uint32_t gt = s_misc[2];  // < k by construction
        for (uint32_t i = gt + tid; i < (uint32_t)a.k; i += kFinalizeThreads) tk[i] = th;
        if (tid == 0) { a.topk_counts[q] = a.k; a.theta[q] = th; }
      }
    }
  }
  const uint32_t f_all = flags0 | flags;  // every thread: uniform
  if (tid == 0) {
    a.flags[q] = f_all;
    a.list_counts[2 * q + 1] = (int32_t)f_all;
  }

  if (a.final_out) {  // last segment: select and sort the answer on the device when that is provably what the heap returns
    const int k2 = a.final_k;
    const bool shard = a.final_shard != 0;
    const int hdr_slots = shard ? 3 : 2;
    const int64_t total = base + m_new;  // entries of the complete list
    const uint32_t tcount = a.need_theta ? (uint32_t)a.topk_counts[q] : tcount0;  // need_theta (never set on a last segment) would have rewritten it
    bool ok = f_all == 0 && a.emit && total <= a.list_cap && k2 >= 1 && k2 <= kFinalSelectMax && k2 + hdr_slots <= a.final_stride &&
              m_new <= (uint32_t)kFinalizeKeyCap - tcount && a.k == k2 + 1;
    uint64_t *__restrict__ s_sel = reinterpret_cast<uint64_t *>(s_jobs);  // the copy jobs are done with
    auto bits_of = [](uint32_t key) -> uint32_t { return (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key; };
    uint32_t n_sel = 0, th1 = 0;  // rows with key > th1 are the answer
    // single index: a list of at most k2 rows holds every row of the index.  Shard: fewer than k2 + 1 rows SEEN (pilot replica included)
    // means no cut exists yet and every listed row stays in the running
    const bool take_all = shard ? (m_new + tcount < (uint32_t)(k2 + 1)) : (total <= (int64_t)k2);
    if (ok && !take_all) {
      // the (k2 + 1)-th largest key of everything seen = the (k2 + 1)-th largest of {running top keys (the k2 + 1 largest of the
      // earlier segments, a.k == k2 + 1) U this segment's keys}; s_keys[0, m_new) still holds the latter
      const uint32_t *__restrict__ tk = a.topk_keys + (size_t)q * a.k;
      for (uint32_t i = tid; i < tcount; i += kFinalizeThreads) s_keys[m_new + i] = tk[i];
      const uint32_t M = m_new + tcount;
      const uint32_t M4 = (M + 3u) & ~3u;
      if (tid < (int)(M4 - M)) s_keys[M + tid] = 0u;  // key 0 is below every key of a finite score
      __syncthreads();
      if (M < (uint32_t)(k2 + 1)) {
        ok = false;  // uniform
      } else if (M <= 2048u) {
        // few keys: the wanted key X is the one with  #(keys > X) <= k2 < #(keys >= X)  - counted per key, two barriers in all
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        const u32x4 *__restrict__ k4 = reinterpret_cast<const u32x4 *>(s_keys);
        for (uint32_t i = tid; i < M; i += kFinalizeThreads) {
          const uint32_t key = s_keys[i];
          uint32_t gt = 0, ge = 0;
          for (uint32_t j = 0; j < M4 / 4; ++j) {
            const u32x4 v = k4[j];
            gt += (v.x > key) + (v.y > key) + (v.z > key) + (v.w > key);
            ge += (v.x >= key) + (v.y >= key) + (v.z >= key) + (v.w >= key);
          }
          if (gt <= (uint32_t)k2 && (uint32_t)k2 < ge) s_misc[0] = key;  // every writer writes the same value
        }
        __syncthreads();
        th1 = s_misc[0];
      } else {
        th1 = block_select_kth_largest(s_keys, M, (uint32_t)(k2 + 1), s_hist, s_wave, s_misc + 8);
      }
    }
    if (ok) {
      if (tid == 0) { s_misc[2] = 0; s_misc[3] = 0; }
      __syncthreads();
      for (int64_t i0 = 0; i0 < total; i0 += 4 * kFinalizeThreads) {  // four independent loads in flight per thread
        uint64_t ent[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t i = i0 + (int64_t)u * kFinalizeThreads + tid;
          ent[u] = i < total ? list[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t i = i0 + (int64_t)u * kFinalizeThreads + tid;
          const uint32_t key = key_of_bits((uint32_t)ent[u]);
          if (i < total && (take_all || key > th1)) {
            const uint32_t slot = atomicAdd(&s_misc[2], 1u);
            if (slot < (uint32_t)kFinalSelectMax) s_sel[slot] = ((uint64_t)key << 32) | (ent[u] >> 32);
          }
        }
      }
      __syncthreads();
      n_sel = s_misc[2];
      // single index: exactly k2 rows above the boundary (fewer: the boundary value repeats; a list shorter than k2 + 1 rows holds them
      // all).  Shard: at most k2 rows lie above an order statistic of rank k2 + 1, however many of them are this shard's own
      ok = shard ? (n_sel <= (uint32_t)k2) : take_all ? (n_sel == (uint32_t)total) : (n_sel == (uint32_t)k2);
    }
    if (ok) {
      // sort by counting (no barriers): the place of a row = the number of rows above it.  Equal scores (as floats: +0 == -0)
      // anywhere in the answer or at its boundary mean that the heap's history decides: flagged, the host replays (a shard leaves
      // that check to the merge, which sees the global answer)
      uint64_t *__restrict__ fo = a.final_out + (size_t)q * a.final_stride + hdr_slots;
      const float fth = __uint_as_float(bits_of(th1));
      for (uint32_t i = tid; i < n_sel; i += kFinalizeThreads) {
        const uint64_t x = s_sel[i];
        const float fx = __uint_as_float(bits_of((uint32_t)(x >> 32)));
        uint32_t rank = 0, same = 0;
        for (uint32_t j = 0; j < n_sel; ++j) {
          const uint64_t y = s_sel[j];
          rank += y > x ? 1u : 0u;
          same += __uint_as_float(bits_of((uint32_t)(y >> 32))) == fx ? 1u : 0u;
        }
        if (!shard && (same > 1u || (!take_all && fx == fth))) s_misc[3] = 1;
        fo[rank] = ((uint64_t)(uint32_t)x << 32) | bits_of((uint32_t)(x >> 32));
      }
      __syncthreads();
      ok = s_misc[3] == 0;
    }
    if (a.done_flag) {  // latency path: the entries above went to mapped host memory - they must be there before the header and the sequence word
      __threadfence_system();
      __syncthreads();
    }
    if (tid == 0) {
      uint64_t *__restrict__ hdr = a.final_out + (size_t)q * a.final_stride;
      const uint32_t listed = a.emit ? (uint32_t)min((int64_t)(base + m_new), a.list_cap) : 0u;
      hdr[0] = (uint64_t)listed | ((uint64_t)f_all << 32);
      hdr[1] = (uint64_t)(ok ? n_sel : 0u) | ((uint64_t)(ok ? 0u : 1u) << 32);
      if (shard) hdr[2] = (ok && !take_all) ? (uint64_t)th1 : 0ull;
      if (a.done_flag) {
        // the next call's chain starts without a copy that would reset the control words: leave them clean
        a.flags[q] = 0u;
        a.theta[q] = 0u;
        a.topk_counts[q] = 0;
        a.list_counts[2 * q] = 0;
        a.list_counts[2 * q + 1] = 0;
        __threadfence_system();
        __hip_atomic_store(a.done_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// index build: row-major (codes [n][pb], corr [n][4]) -> tile records.  One thread per (row, chunk).

__device__ __forceinline__ uint32_t bf16_trunc_bits(double v) { return __float_as_uint((float)v) >> 16; }

__global__ __launch_bounds__(256) void bbq_retile_kernel(const uint8_t *__restrict__ codes, const double *__restrict__ corr,
                                                        int64_t n_rows, int32_t pb, uint8_t *__restrict__ tiles, int32_t w16,
                                                        int32_t tile_stride, int32_t has_x1, int64_t n_rows_padded, int32_t layout,
                                                        double *__restrict__ exact) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / (w16 + 1);
  const int j = (int)(gid % (w16 + 1));
  if (row >= n_rows_padded) return;
  const int64_t tile = row / kTileRows;
  const int r = (int)(row % kTileRows);
  uint8_t *tp = tiles + tile * (int64_t)tile_stride;
  if (j < w16) {
    u32x4 v = {0, 0, 0, 0};
    if (row < n_rows) {
      uint32_t w[4] = {0, 0, 0, 0};
      const uint8_t *src = codes + row * (int64_t)pb;
      for (int b = 0; b < 16; ++b) {
        const int byte = j * 16 + b;
        if (byte < pb) w[b >> 2] |= (uint32_t)src[byte] << (8 * (b & 3));
      }
      v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    }
    reinterpret_cast<u32x4 *>(tp)[j * kTileRows + r] = v;
  } else {
    uint8_t *cr = tp + (size_t)w16 * (kTileRows * 16);
    f64x2 lu = {0.0, 0.0};
    double add = 0.0, x1 = 0.0;
    if (row < n_rows) {
      lu.x = corr[row * 4 + 0]; lu.y = corr[row * 4 + 1]; add = corr[row * 4 + 2]; x1 = corr[row * 4 + 3];
    }
    if (layout == kLayoutCompact) {
      reinterpret_cast<uint32_t *>(cr)[r] = bf16_trunc_bits(lu.x) | (bf16_trunc_bits(lu.y) << 16);  // the additive term: bbq_tile_add_range_kernel
      double *e = exact + row * 4;
      e[0] = lu.x; e[1] = lu.y; e[2] = add; e[3] = 0.0;
    } else {
      reinterpret_cast<f64x2 *>(cr)[r] = lu;
      reinterpret_cast<double *>(cr + 1024)[r] = add;
      if (has_x1) reinterpret_cast<double *>(cr + 1536)[r] = x1;
    }
  }
}

// multi-bit index: unpacked rows (one byte per dimension, [n][dim]) -> store_bits-wide fields in tile records; the corrections
// block is written exactly as above.  A code that is not below 2^index_bits raises *bad (the index is refused).
__global__ __launch_bounds__(256) void bbq_retile_multibit_kernel(const uint8_t *__restrict__ codes, const double *__restrict__ corr,
                                                                 int64_t n_rows, int32_t dim, int32_t store_bits, int32_t index_bits, uint8_t *__restrict__ tiles,
                                                                 int32_t w16, int32_t tile_stride, int32_t has_x1, int64_t n_rows_padded,
                                                                 int32_t layout, double *__restrict__ exact, uint32_t *__restrict__ bad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / (w16 + 1);
  const int j = (int)(gid % (w16 + 1));
  if (row >= n_rows_padded) return;
  const int64_t tile = row / kTileRows;
  const int r = (int)(row % kTileRows);
  uint8_t *tp = tiles + tile * (int64_t)tile_stride;
  if (j < w16) {
    uint32_t w[4] = {0, 0, 0, 0};
    if (row < n_rows) {
      const int per_dword = 32 / store_bits;
      const uint8_t *src = codes + row * (int64_t)dim;
      const uint32_t limit = 1u << index_bits, field = (1u << store_bits) - 1u;  // values of an indexBits-bit quantizer are < 2^indexBits (include/bbq.h)
      for (int t = 0; t < 4; ++t)
        for (int f = 0; f < per_dword; ++f) {
          const int d = (j * 4 + t) * per_dword + f;
          if (d < dim) {
            const uint32_t v = src[d];
            if (v >= limit) atomicOr(bad, 1u);
            w[t] |= (v & field) << (f * store_bits);
          }
        }
    }
    u32x4 v = {w[0], w[1], w[2], w[3]};
    reinterpret_cast<u32x4 *>(tp)[j * kTileRows + r] = v;
  } else {
    uint8_t *cr = tp + (size_t)w16 * (kTileRows * 16);
    f64x2 lu = {0.0, 0.0};
    double add = 0.0, x1 = 0.0;
    if (row < n_rows) {
      lu.x = corr[row * 4 + 0]; lu.y = corr[row * 4 + 1]; add = corr[row * 4 + 2]; x1 = corr[row * 4 + 3];
    }
    if (layout == kLayoutCompact) {
      reinterpret_cast<uint32_t *>(cr)[r] = bf16_trunc_bits(lu.x) | (bf16_trunc_bits(lu.y) << 16);
      double *e = exact + row * 4;
      e[0] = lu.x; e[1] = lu.y; e[2] = add; e[3] = 0.0;
    } else {
      reinterpret_cast<f64x2 *>(cr)[r] = lu;
      reinterpret_cast<double *>(cr + 1024)[r] = add;
      if (has_x1) reinterpret_cast<double *>(cr + 1536)[r] = x1;
    }
  }
}

// does quantizedComponentSum equal the sum of the row's codes everywhere? (multi-bit rows, one byte per dimension)
__global__ __launch_bounds__(256) void bbq_check_x1_multibit_kernel(const uint8_t *__restrict__ codes, const double *__restrict__ corr,
                                                                   int64_t n_rows, int32_t dim, uint32_t *__restrict__ mismatch) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const uint8_t *src = codes + row * (int64_t)dim;
  uint32_t sum = 0;
  for (int d = 0; d < dim; ++d) sum += src[d];
  if (!(corr[row * 4 + 3] == (double)sum)) atomicOr(mismatch, 1u);
}

// compact layout: {min, max} of additionalCorrection over the valid rows of each tile, as f32 (one wave per tile; the f32
// rounding is inside the bound's allowance for the additive term).  A NaN anywhere makes both ends NaN: no bound, exact path.
__global__ __launch_bounds__(256) void bbq_tile_add_range_kernel(const double *__restrict__ exact, int64_t n_rows, float *__restrict__ add_range) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  if (tile >= n_tiles) return;
  const int64_t row = tile * kTileRows + lane;
  const bool valid = row < n_rows;
  const double v = valid ? exact[row * 4 + 2] : 0.0;
  bool nan = valid && (v != v);
  double lo = valid ? v : __longlong_as_double(0x7ff0000000000000ll), hi = valid ? v : __longlong_as_double(0xfff0000000000000ll);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, d, 64));
    hi = fmax(hi, __shfl_xor(hi, d, 64));
  }
  nan = __any(nan);
  if (lane == 0) {
    add_range[tile * 2] = nan ? __uint_as_float(0x7fc00000u) : (float)lo;
    add_range[tile * 2 + 1] = nan ? __uint_as_float(0x7fc00000u) : (float)hi;
  }
}

// does quantizedComponentSum equal the row's popcount everywhere? (then it need not be stored)
__global__ __launch_bounds__(256) void bbq_check_x1_kernel(const uint8_t *__restrict__ codes, const double *__restrict__ corr,
                                                          int64_t n_rows, int32_t pb, uint32_t *__restrict__ mismatch) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const uint8_t *src = codes + row * (int64_t)pb;
  uint32_t ones = 0;
  for (int b = 0; b < pb; ++b) ones += __popc((uint32_t)src[b]);
  if (!(corr[row * 4 + 3] == (double)ones)) atomicOr(mismatch, 1u);
}

// ---------------------------------------------------------------------------------------------------
// shard transport: per-query lists [nq][list_cap] -> one packed buffer + offsets (what goes over RCCL)

__global__ __launch_bounds__(1024) void bbq_pack_offsets_kernel(const int32_t *__restrict__ counts /*[nq][2]*/, int32_t nq,
                                                               int64_t *__restrict__ offsets /*[nq+1]*/, int32_t *__restrict__ flags_out,
                                                               int64_t *__restrict__ total_out, int64_t advertised_cap, int64_t packed_cap) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint64_t s_base;
  // lists may hold floods beyond the advertised per-query capacity.  Normally the other queries leave plenty of room in
  // the packed buffer; only if the sum does not fit, the flooded queries are dropped (flagged: they take the dense path),
  // after which everything fits into any buffer of at least nq * advertised_cap entries.
  for (int round = 0; round < 2; ++round) {
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const bool squeeze = round == 1;
    for (int32_t q0 = 0; q0 < nq; q0 += 1024) {
      const int32_t q = q0 + (int32_t)threadIdx.x;
      uint32_t c = 0;
      if (q < nq) {
        int32_t f = counts[2 * q + 1];
        c = (uint32_t)counts[2 * q];
        if (!f && squeeze && (int64_t)c > advertised_cap) f = (int32_t)kFlagOverflow;
        flags_out[q] = f;
        if (f) c = 0u;  // a flagged query carries no list: it takes the dense path
      }
      uint32_t total;
      const uint32_t ex = block_exclusive_scan_1024(c, s_wave, total);
      const uint64_t base = s_base;
      if (q < nq) offsets[q] = (int64_t)(base + ex);
      __syncthreads();
      if (threadIdx.x == 0) s_base = base + total;
      __syncthreads();
    }
    if ((int64_t)s_base <= packed_cap) break;  // uniform: s_base is shared
    __syncthreads();
  }
  if (threadIdx.x == 0) { offsets[nq] = (int64_t)s_base; *total_out = (int64_t)s_base; }
}

__global__ __launch_bounds__(256) void bbq_pack_copy_kernel(const uint64_t *__restrict__ lists, int64_t list_cap,
                                                           const int64_t *__restrict__ offsets, uint64_t *__restrict__ packed,
                                                           int64_t packed_cap) {
  const int q = blockIdx.x;
  const int64_t b = offsets[q], e = offsets[q + 1];
  if (e > packed_cap) return;  // the host sees total > packed_cap and reports it
  const uint64_t *__restrict__ src = lists + (size_t)q * list_cap;
  for (int64_t i = threadIdx.x; i < e - b; i += 256) packed[b + i] = src[i];
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers (declared in bbq_launch.h)

template <int QB, int W, int MODE, int SB = 1>
static hipError_t launch_scan_t(const ScanArgs &a, int n_queries, int n_chunks, hipStream_t s) {
  const int w16 = W > 0 ? W : a.idx.w16;
  const size_t smem = (size_t)w16 * query_units_per_chunk(QB, SB) * 16 + ((MODE & 1) ? 0 : (size_t)((a.ovf || a.append_lists) ? kChunkRows : a.cap) * 8) + 16;
  dim3 grid((unsigned)n_chunks, (unsigned)n_queries, 1), block(kChunkRows, 1, 1);
  hipLaunchKernelGGL((bbq_scan_kernel<QB, W, MODE, SB>), grid, block, smem, s, a);
  return hipGetLastError();
}

// multi-bit rows: compile-time widths for 768-d / 1024-d at 2 bits (12 / 16 chunks; 16 is also 512-d at 4 bits), runtime loop otherwise
template <int QB, int MODE, int SB>
static hipError_t launch_scan_mb_w(const ScanArgs &a, int nq, int nc, hipStream_t s) {
  switch (a.idx.w16) {
    case 12: return launch_scan_t<QB, 12, MODE, SB>(a, nq, nc, s);
    case 16: return launch_scan_t<QB, 16, MODE, SB>(a, nq, nc, s);
    default: return launch_scan_t<QB, 0, MODE, SB>(a, nq, nc, s);
  }
}
template <int MODE>
static hipError_t launch_scan_mb(const ScanArgs &a, int planes, int nq, int nc, hipStream_t s) {
  switch (a.idx.store_bits) {
    case 2: return planes > 4 ? launch_scan_mb_w<8, MODE, 2>(a, nq, nc, s) : launch_scan_mb_w<4, MODE, 2>(a, nq, nc, s);
    case 4: return planes > 4 ? launch_scan_mb_w<8, MODE, 4>(a, nq, nc, s) : launch_scan_mb_w<4, MODE, 4>(a, nq, nc, s);
    case 8: return launch_scan_t<8, 0, MODE, 8>(a, nq, nc, s);
    default: return hipErrorInvalidValue;
  }
}

template <int QB, int MODE>
static hipError_t launch_scan_w(const ScanArgs &a, int nq, int nc, hipStream_t s) {
  switch (a.idx.w16) {
    case 1: return launch_scan_t<QB, 1, MODE>(a, nq, nc, s);    // dim <= 128
    case 6: return launch_scan_t<QB, 6, MODE>(a, nq, nc, s);    // dim 768
    case 8: return launch_scan_t<QB, 8, MODE>(a, nq, nc, s);    // dim 1024
    case 12: return launch_scan_t<QB, 12, MODE>(a, nq, nc, s);  // dim 1536
    default: return launch_scan_t<QB, 0, MODE>(a, nq, nc, s);
  }
}

template <int MODE>
static hipError_t launch_scan_q(const ScanArgs &a, int planes, int nq, int nc, hipStream_t s) {
  switch (planes) {
    case 1: return launch_scan_w<1, MODE>(a, nq, nc, s);
    case 2: return launch_scan_w<2, MODE>(a, nq, nc, s);
    case 4: return launch_scan_w<4, MODE>(a, nq, nc, s);
    default: return launch_scan_w<8, MODE>(a, nq, nc, s);
  }
}

template <int QB, int W, bool COMPACT, int NB>
static hipError_t launch_shared_t(const ScanArgs &a, int nq, int nc, hipStream_t s) {
  const size_t smem = (size_t)NB * W * QB * 16 + (size_t)NB * sizeof(QueryParams) + (size_t)NB * a.cap * 8 + (size_t)NB * 8 + 16;
  dim3 grid((unsigned)nc, (unsigned)((nq + NB - 1) / NB), 1), block(kChunkRows, 1, 1);
  hipLaunchKernelGGL((bbq_scan_shared_kernel<QB, W, COMPACT, NB>), grid, block, smem, s, a, nq);
  return hipGetLastError();
}
template <int QB, bool COMPACT, int NB>
static hipError_t launch_shared_w(const ScanArgs &a, int nq, int nc, hipStream_t s) {
  switch (a.idx.w16) {
    case 1: return launch_shared_t<QB, 1, COMPACT, NB>(a, nq, nc, s);
    case 6: return launch_shared_t<QB, 6, COMPACT, NB>(a, nq, nc, s);
    case 8: return launch_shared_t<QB, 8, COMPACT, NB>(a, nq, nc, s);
    case 12: return launch_shared_t<QB, 12, COMPACT, NB>(a, nq, nc, s);
    default: return hipErrorInvalidValue;
  }
}
template <bool COMPACT, int NB>
static hipError_t launch_shared_q(const ScanArgs &a, int planes, int nq, int nc, hipStream_t s) {
  switch (planes) {
    case 1: return launch_shared_w<1, COMPACT, NB>(a, nq, nc, s);
    case 2: return launch_shared_w<2, COMPACT, NB>(a, nq, nc, s);
    case 4: return launch_shared_w<4, COMPACT, NB>(a, nq, nc, s);
    default: return launch_shared_w<8, COMPACT, NB>(a, nq, nc, s);
  }
}

bool shared_sweep_supported(const ScanArgs &a, int share) {
  const int w = a.idx.w16;
  return a.idx.store_bits == 1 && (share == 4 || share == 8) && (w == 1 || w == 6 || w == 8 || w == 12) && (size_t)share * a.cap * 8 < 48 * 1024;
}

// sparse segments only; `share` queries per workgroup read each row once
hipError_t launch_scan_shared(const ScanArgs &a, int planes, int share, int n_queries, int n_chunks, hipStream_t s) {
  if (n_chunks <= 0 || n_queries <= 0) return hipSuccess;
  const bool compact = a.idx.layout == kLayoutCompact;
  if (share == 8) return compact ? launch_shared_q<true, 8>(a, planes, n_queries, n_chunks, s) : launch_shared_q<false, 8>(a, planes, n_queries, n_chunks, s);
  return compact ? launch_shared_q<true, 4>(a, planes, n_queries, n_chunks, s) : launch_shared_q<false, 4>(a, planes, n_queries, n_chunks, s);
}

hipError_t launch_scan(const ScanArgs &a, int planes, bool dense, int n_queries, int n_chunks, hipStream_t s) {
  if (n_chunks <= 0 || n_queries <= 0) return hipSuccess;
  const int mode = (dense ? 1 : 0) | (a.idx.layout == kLayoutCompact ? 2 : 0);
  if (a.idx.store_bits > 1) {
    switch (mode) {
      case 0: return launch_scan_mb<0>(a, planes, n_queries, n_chunks, s);
      case 1: return launch_scan_mb<1>(a, planes, n_queries, n_chunks, s);
      case 2: return launch_scan_mb<2>(a, planes, n_queries, n_chunks, s);
      default: return launch_scan_mb<3>(a, planes, n_queries, n_chunks, s);
    }
  }
  switch (mode) {
    case 0: return launch_scan_q<0>(a, planes, n_queries, n_chunks, s);
    case 1: return launch_scan_q<1>(a, planes, n_queries, n_chunks, s);
    case 2: return launch_scan_q<2>(a, planes, n_queries, n_chunks, s);
    default: return launch_scan_q<3>(a, planes, n_queries, n_chunks, s);
  }
}

hipError_t launch_finalize(const FinalizeArgs &a, int n_queries, hipStream_t s) {
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(bbq_finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               kFinalizeLdsBytes);  // once per process: more than the 64 KB a launch gets by default
  if (attr != hipSuccess) return attr;
  const bool compacts = !(a.dense_rows > 0) && a.append_counts == nullptr;
  hipLaunchKernelGGL(bbq_finalize_kernel, dim3((unsigned)n_queries), dim3(kFinalizeThreads), compacts ? kFinalizeLdsBytes : kFinalizeLdsBytesSmall, s, a);
  return hipGetLastError();
}

hipError_t launch_pack(const int32_t *counts, const uint64_t *lists, int64_t list_stride, int64_t advertised_cap, int32_t nq, int64_t *offsets,
                       int32_t *flags_out, int64_t *total_out, uint64_t *packed, int64_t packed_cap, hipStream_t s) {
  if (nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(bbq_pack_offsets_kernel, dim3(1), dim3(1024), 0, s, counts, nq, offsets, flags_out, total_out, advertised_cap, packed_cap);
  hipLaunchKernelGGL(bbq_pack_copy_kernel, dim3((unsigned)nq), dim3(256), 0, s, lists, list_stride, (const int64_t *)offsets, packed, packed_cap);
  return hipGetLastError();
}

hipError_t launch_retile(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t pb, uint8_t *tiles, int32_t w16,
                         int32_t tile_stride, int32_t has_x1, int32_t layout, double *exact, hipStream_t s) {
  const int64_t n_pad = (n_rows + kTileRows - 1) / kTileRows * kTileRows;
  const int64_t threads = n_pad * (w16 + 1);
  if (threads == 0) return hipSuccess;
  const int64_t blocks = (threads + 255) / 256;
  hipLaunchKernelGGL(bbq_retile_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, corr, n_rows, pb, tiles, w16, tile_stride,
                     has_x1, n_pad, layout, exact);
  return hipGetLastError();
}

hipError_t launch_retile_multibit(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t store_bits, int32_t index_bits, uint8_t *tiles,
                                  int32_t w16, int32_t tile_stride, int32_t has_x1, int32_t layout, double *exact, uint32_t *bad, hipStream_t s) {
  const int64_t n_pad = (n_rows + kTileRows - 1) / kTileRows * kTileRows;
  const int64_t threads = n_pad * (w16 + 1);
  if (threads == 0) return hipSuccess;
  const int64_t blocks = (threads + 255) / 256;
  hipLaunchKernelGGL(bbq_retile_multibit_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, corr, n_rows, dim, store_bits, index_bits, tiles, w16,
                     tile_stride, has_x1, n_pad, layout, exact, bad);
  return hipGetLastError();
}

hipError_t launch_check_x1_multibit(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, uint32_t *mismatch, hipStream_t s) {
  if (n_rows == 0) return hipSuccess;
  const int64_t blocks = (n_rows + 255) / 256;
  hipLaunchKernelGGL(bbq_check_x1_multibit_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, corr, n_rows, dim, mismatch);
  return hipGetLastError();
}

hipError_t launch_tile_add_range(const double *exact, int64_t n_rows, float *add_range, hipStream_t s) {
  const int64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(bbq_tile_add_range_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, exact, n_rows, add_range);
  return hipGetLastError();
}

hipError_t launch_check_x1(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t pb, uint32_t *mismatch, hipStream_t s) {
  if (n_rows == 0) return hipSuccess;
  const int64_t blocks = (n_rows + 255) / 256;
  hipLaunchKernelGGL(bbq_check_x1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, corr, n_rows, pb, mismatch);
  return hipGetLastError();
}

}  // namespace bbq
