// bbq_kernel_common.h - device functions shared by the kernels of bbq_kernels.hip and bbq_latency_kernels.hip (gfx950 only): the
// reference's float64 score formulas, the per-tile popcount loop, the score bound of the compact layout, block-wide scan and
// order-statistic selection.  Everything here is __forceinline__: each kernel file gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include "bbq_device.h"

#pragma clang fp contract(off)

namespace bbq {

// streamed, read-once data: non-temporal loads (A/B on MI355X: see DESIGN.md); -DBBQ_PLAIN_LOADS for the experiment
#ifdef BBQ_PLAIN_LOADS
#define BBQ_STREAM_LOAD(p) (*(p))
#else
#define BBQ_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// a tile's bytes: default cache policy for the resident part of the index (IndexView::resident_tiles), streamed otherwise.
// `resident` must be wave-uniform (chunk_is_resident: decided per workgroup from blockIdx): a scalar branch, never both loads
template <class T> __device__ __forceinline__ const T *stream_ptr(const T *p, int64_t nt_delta) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(p) + nt_delta);
}
// the compiler merges the loads of two branches that differ only in the cache policy into ONE plain load (also through a phi of
// the two addresses): the streamed branch is fenced with compiler barriers, which its loads cannot be hoisted or sunk across
#define BBQ_BRANCH_FENCE() asm volatile("" ::: "memory")
__device__ __forceinline__ bool chunk_is_resident(int64_t chunk, const IndexView &v) {  // chunk comes from blockIdx: scalar
  return v.resident_share >= 0 ? (chunk & 63) < v.resident_share : chunk * kTilesPerChunk < v.resident_tiles;
}
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ALL loads of a wave's tile in ONE two-way branch (W > 0): the W code chunks of the lane's row and its corrections - CORR 0: none,
// 1: the compact word, 2: the inline f64 corrections (lower, upper | additional | component sum if stored).  Loads that are
// already in flight when the branch is reached make the compiler wait for them inside it (it assumes either arm may follow them).
template <int W, int CORR>
__device__ __forceinline__ void load_tile(const uint8_t *__restrict__ tp, int lane, bool has_x1, bool resident, int64_t nt_delta,
                                          u32x4 (&c)[W], uint32_t &cw, f64x2 &lu, double &xadd, double &x1) {
  const u32x4 *__restrict__ cp = reinterpret_cast<const u32x4 *>(tp) + lane;
  const uint8_t *__restrict__ cr = tp + (size_t)W * (kTileRows * 16);
  if (resident) {  // scalar branch
#pragma unroll
    for (int j = 0; j < W; ++j) c[j] = cp[j * kTileRows];
    if constexpr (CORR == 1) cw = *(reinterpret_cast<const uint32_t *>(cr) + lane);
    if constexpr (CORR == 2) {
      lu = *(reinterpret_cast<const f64x2 *>(cr) + lane);
      xadd = *(reinterpret_cast<const double *>(cr + 1024) + lane);
      if (has_x1) x1 = *(reinterpret_cast<const double *>(cr + 1536) + lane);
    }
  } else {
    BBQ_BRANCH_FENCE();
    const u32x4 *__restrict__ cs = stream_ptr(cp, nt_delta);
    const uint8_t *__restrict__ crs = stream_ptr(cr, nt_delta);
#pragma unroll
    for (int j = 0; j < W; ++j) c[j] = BBQ_STREAM_LOAD(cs + j * kTileRows);
    if constexpr (CORR == 1) cw = BBQ_STREAM_LOAD(reinterpret_cast<const uint32_t *>(crs) + lane);
    if constexpr (CORR == 2) {
      lu = BBQ_STREAM_LOAD(reinterpret_cast<const f64x2 *>(crs) + lane);
      xadd = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(crs + 1024) + lane);
      if (has_x1) x1 = BBQ_STREAM_LOAD(reinterpret_cast<const double *>(crs + 1536) + lane);
    }
    BBQ_BRANCH_FENCE();
  }
}

__device__ __forceinline__ uint32_t popc4(u32x4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
// acc + popcount(x) in ONE instruction (v_bcnt_u32_b32 has the addend built in).  Written as `acc += popc4(...)` the compiler sums the
// four counts of a 16-byte chunk in a tree of v_add3_u32 first: 56 extra vector instructions per 768-d row, 14 % of the sweep's
// vector work - and the sweep is bound by vector issue whenever its bytes come out of a cache (scan kernel).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ uint32_t popc4_acc(u32x4 v, uint32_t acc) { return bcnt_acc(v.w, bcnt_acc(v.z, bcnt_acc(v.y, bcnt_acc(v.x, acc)))); }

// Math.max(x, 0) of the reference: NaN propagates, -0 -> +0
__device__ __forceinline__ double js_max0(double x) { return (x != x) ? x : (x > 0.0 ? x : 0.0); }

// src/batchDotProduct.ts:478-541 (one_bit) / :554-617 (every other queryBits); SURVEY App. A.4.
// Parenthesised exactly as JavaScript evaluates the reference's expressions; no FMA contraction.
__device__ __forceinline__ double score_f64(double qc, double ax, double ux, double xadd, double x1, const QueryParams &p) {
  const double lx = ux - ax;
  const double t1 = (ax * p.ay) * p.dimd;
  const double t2 = (p.ay * lx) * x1;
  const double t3 = (ax * p.ly) * p.y1;
  const double t4 = (lx * p.ly) * qc;
  const double s = ((t1 + t2) + t3) + t4;
  if (p.sim == 0) {  // EUCLIDEAN
    const double e = (p.qadd + xadd) - (2.0 * s);
    return js_max0(1.0 / (1.0 + e));
  }
  const double t = p.one_bit ? (s + ((p.qadd + xadd) - p.cdp)) : (((s + p.qadd) + xadd) - p.cdp);
  if (p.sim == 1) return js_max0((1.0 + t) / 2.0);  // COSINE
  // scaleMaxInnerProductScore (src/utils.ts:171-176): the 1-bit batch form (:527-533) and the per-row scorer's form for
  // every query width (src/binaryQuantizedScorer.ts:148-153, :207-209), which is what answers for multi-bit indexes
  if (p.one_bit || p.mip_plain) return t < 0.0 ? 1.0 / (1.0 - t) : t + 1.0;
  const double FBS = 1.0 / 15.0;  // FOUR_BIT_SCALE, src/constants.ts:20 - a true division by it, not *15
  return t < 0.0 ? 1.0 / (1.0 - t / FBS) : t / FBS + 1.0;
}

// One tile = 64 rows, one row per lane.  W = compile-time number of 16-byte chunks per row: the chunks come in registers (load_tile)
template <int QB, int W>
__device__ __forceinline__ void tile_popcounts(const u32x4 (&c)[W], const u32x4 *__restrict__ s_planes, uint32_t (&acc)[QB], uint32_t &ones) {
#pragma unroll
  for (int p = 0; p < QB; ++p) acc[p] = 0;
  ones = 0;
#pragma unroll
  for (int j = 0; j < W; ++j) {
#pragma unroll
    for (int p = 0; p < QB; ++p) acc[p] = popc4_acc(c[j] & s_planes[j * QB + p], acc[p]);
    ones = popc4_acc(c[j], ones);
  }
}
// any width (w16 chunks, known at run time): streamed chunk by chunk
template <int QB>
__device__ __forceinline__ void tile_popcounts_any(const uint8_t *__restrict__ tp, int lane, int w16, const u32x4 *__restrict__ s_planes,
                                                   uint32_t (&acc)[QB], uint32_t &ones) {
  const u32x4 *__restrict__ cp = reinterpret_cast<const u32x4 *>(tp) + lane;
#pragma unroll
  for (int p = 0; p < QB; ++p) acc[p] = 0;
  ones = 0;
  for (int j = 0; j < w16; ++j) {
    const u32x4 c = BBQ_STREAM_LOAD(cp + j * kTileRows);
#pragma unroll
    for (int p = 0; p < QB; ++p) acc[p] += popc4(c & s_planes[j * QB + p]);
    ones += popc4(c);
  }
}

// Upper bound of the score when only the COMPACT corrections are known (kLayoutCompact).
// The raw score s is linear in (lower, upper): with x1 and qcDist fixed,
//     s(lower, upper) = lower * A + upper * B,   A = ay*(dim - x1) + ly*(y1 - qc),   B = ay*x1 + ly*qc,
// so replacing (lower, upper, add) by their compact values (al, au, aadd) changes s by exactly
// (lower-al)*A + (upper-au)*B and the additive term by (add-aadd).  |lower-al| <= |al|*kBf16Rel + kAbsSlack
// (f32 rounding + truncation to the upper 16 bits), |add-aadd| <= |aadd|*2^-23 + kAbsSlack.  All three similarity
// transforms are monotone in s (resp. in t), and a generous rounding allowance (kRoundRel, ~7 orders of magnitude
// above the real f64 round-off of these ~20 operations) covers the difference between exact-arithmetic reasoning and
// IEEE evaluation.  Returns a value U with  exact f64 score <= U  (NaN or +inf when no finite bound can be given:
// the caller then takes the exact path).  tests/test_bound_math_cpu.py restates this in numpy and checks dominance.
constexpr double kBf16Rel = 0.0078125 * (1.0 + 1.0 / 65536.0);  // 2^-7 (1 + 2^-16)
constexpr double kF32Rel = 1.1920928955078125e-07;             // 2^-23
constexpr double kAbsSlack = 1e-37;
constexpr double kRoundRel = 1e-9;

__device__ __forceinline__ double score_upper_bound(double qc, double al, double au, double aadd, double x1, const QueryParams &p) {
  const double lx = au - al;
  const double t1 = (al * p.ay) * p.dimd;
  const double t2 = (p.ay * lx) * x1;
  const double t3 = (al * p.ly) * p.y1;
  const double t4 = (lx * p.ly) * qc;
  const double s = ((t1 + t2) + t3) + t4;
  const double A = p.ay * (p.dimd - x1) + p.ly * (p.y1 - qc);
  const double B = p.ay * x1 + p.ly * qc;
  const double mag = fabs(t1) + fabs(t2) + fabs(t3) + fabs(t4) + fabs(p.qadd) + fabs(aadd) + fabs(p.cdp) + 1.0;
  if (!(mag < 1e290)) return __longlong_as_double(0x7ff8000000000000ll);  // non-finite / huge: no bound
  const double es = fabs(A) * (fabs(al) * kBf16Rel + kAbsSlack) + fabs(B) * (fabs(au) * kBf16Rel + kAbsSlack);
  const double eadd = fabs(aadd) * kF32Rel + kAbsSlack;
  const double slop = kRoundRel * (mag + fabs(A) + fabs(B));
  if (p.sim == 0) {  // EUCLIDEAN: score = max(1/(1+e), 0), decreasing in e while 1+e > 0
    const double e_low = ((p.qadd + aadd) - (2.0 * s)) - (2.0 * es + eadd + slop);
    const double den = 1.0 + e_low;
    if (!(den > 0.0)) return __longlong_as_double(0x7ff0000000000000ll);  // +inf: cannot exclude a tiny positive denominator
    const double u = 1.0 / den;
    return u + kRoundRel * (u + 1.0);
  }
  const double t_up = (((s + p.qadd) + aadd) - p.cdp) + (es + eadd + slop);
  double u;
  if (p.sim == 1) u = js_max0((1.0 + t_up) / 2.0);
  else if (p.one_bit || p.mip_plain) u = t_up < 0.0 ? 1.0 / (1.0 - t_up) : t_up + 1.0;
  else {
    const double FBS = 1.0 / 15.0;
    u = t_up < 0.0 ? 1.0 / (1.0 - t_up / FBS) : t_up / FBS + 1.0;
  }
  return u + kRoundRel * (fabs(u) + 1.0);
}

__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *s_wave, uint32_t &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t n = __shfl_up(incl, d, 64);
    if (lane >= d) incl += n;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t x = s_wave[w];
    if (w < wave) wave_off += x;
    tot += x;
  }
  total = tot;
  __syncthreads();
  return wave_off + incl - v;
}

// k-th largest of the M keys in LDS (M >= k >= 1): radix select over the key bytes that actually vary, one 1024-thread workgroup;
// every thread returns it.  s_hist: 2 x 256 words, s_wave: 16 words, s_scr: 8 words of scratch owned by this function.
// Barriers are what this costs (16 waves: ~0.4 us each), so there are two per pass: the histogram of a pass is built in one of two
// buffers while the other is being cleared, and every pass leaves its result in words of its own.
template <int NT>
__device__ __forceinline__ uint32_t block_select_kth_largest_t(const uint32_t *s_keys, uint32_t M, uint32_t k, uint32_t *s_hist, uint32_t *s_wave,
                                                               uint32_t *s_scr) {
  const int tid = threadIdx.x;
  // Scores of one query live in a narrow range: the upper bytes of their keys are the same for (nearly) all of them, and a
  // histogram pass over such a byte is thousands of atomic adds on ONE LDS word (measured: 10 us per pass at 6 K keys).
  // Bytes that are constant over all keys are therefore skipped: they belong to the answer as they are.
  uint32_t vary = 0;
  {
    const uint32_t key0 = s_keys[0];
#pragma unroll 4
    for (uint32_t i = tid; i < M; i += NT) vary |= s_keys[i] ^ key0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vary |= __shfl_xor(vary, d, 64);
    if ((tid & 63) == 0) s_wave[tid >> 6] = vary;
    for (int i = tid; i < 512; i += NT) s_hist[i] = 0;
    __syncthreads();
    vary = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) vary |= s_wave[w];
  }
  uint32_t prefix = 0, mask = 0, kk = k;
  int buf = 0;
  for (int pass = 3; pass >= 0; --pass) {
    const int sh = pass * 8;
    if (((vary >> sh) & 255u) == 0u) {  // uniform: every key has the same byte here
      prefix |= s_keys[0] & (255u << sh);
      mask |= 255u << sh;
      continue;
    }
    uint32_t *__restrict__ hist = s_hist + 256 * buf;
    // The pass is latency-bound per wave (a wave walks M / NT keys one after the other), so nothing in a step may wait: the keys of
    // four steps are loaded before any of them is used, the first active lane's bin is read with v_readlane (an LDS-routed shuffle
    // was a round trip per key) and the histogram adds return nothing.
    for (uint32_t i0 = 0; i0 < M; i0 += 4 * NT) {  // whole waves iterate together (ballots below)
      uint32_t kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * NT + tid;
        kv[u] = s_keys[i < M ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * NT + tid;
        const bool in = i < M && (kv[u] & mask) == prefix;
        const uint32_t bin = in ? (kv[u] >> sh) & 255u : 256u;
        // a byte with few distinct values would still pile the adds on a few words: the lanes that share the first active lane's
        // bin add once for all of them
        const unsigned long long act = __ballot(in);
        if (act) {
          const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, __ffsll((long long)act) - 1);
          const unsigned long long same = __ballot(in && bin == b0);
          if (in && bin == b0) {
            if ((tid & 63) == __ffsll((long long)same) - 1) atomicAdd(&hist[b0], (uint32_t)__popcll(same));
          } else if (in) {
            atomicAdd(&hist[bin], 1u);
          }
        }
      }
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0 finds the bin: lane l owns bins 4l..4l+3; suffix sums over lanes by shuffles (no LDS round trips)
      const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      uint32_t suf = h0 + h1 + h2 + h3;  // becomes the sum over bins >= 4*tid
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_down(suf, d, 64);
        if (tid + d < 64) suf += n;
      }
      const uint32_t above = suf - (h0 + h1 + h2 + h3);  // bins > 4*tid+3
      if (suf >= kk && above < kk) {                     // exactly one lane: the k-th largest lies in its 4 bins
        uint32_t cum = above;
        int b = 4 * tid + 3;
        if (cum + h3 < kk) { cum += h3; b = 4 * tid + 2;
          if (cum + h2 < kk) { cum += h2; b = 4 * tid + 1;
            if (cum + h1 < kk) { cum += h1; b = 4 * tid; } } }
        s_scr[2 * pass] = (uint32_t)b;
        s_scr[2 * pass + 1] = kk - cum;
      }
    } else if (tid >= NT - 256) {
      s_hist[256 * (buf ^ 1) + (tid - (NT - 256))] = 0;  // the other buffer, for the next pass
    }
    __syncthreads();
    prefix |= s_scr[2 * pass] << sh;
    mask |= 255u << sh;
    kk = s_scr[2 * pass + 1];
    buf ^= 1;
  }
  __syncthreads();  // the caller may reuse the scratch words and the keys
  return prefix;
}

__device__ __forceinline__ uint32_t block_select_kth_largest(const uint32_t *s_keys, uint32_t M, uint32_t k, uint32_t *s_hist, uint32_t *s_wave,
                                                             uint32_t *s_scr) {
  return block_select_kth_largest_t<kFinalizeThreads>(s_keys, M, k, s_hist, s_wave, s_scr);
}

}  // namespace bbq
