// bbq_mfma_kernels.hip - shared sweep on the matrix cores (API extension; NOT the one-sweep-per-query path).
//
// With 32 queries sharing a sweep the integer dot products ARE a dense contraction,
//     qcDist[row][query] = sum_d bit_d(row) * q[query][d],
// and the VALU popcount formulation is bound by v_bcnt (bbq_kernels.hip: shared kernel, ~1.4x).  Here every wave expands
// the 1-bit codes of its 64-row tile to int8 fragments on the fly and multiplies them with the 32 queries' int8
// values on v_mfma_i32_32x32x32_i8 (fragment layout verified on gfx950 by scripts/ubench/mfma_probe.hip:
//   A[m = lane%32][k = 16*(lane/32)+i],  B[k][n = lane%32],  C[r] -> row (r&3) + 8*(r>>2) + 4*(lane/32), col lane%32).
// The k -> dimension assignment is free as long as A and B agree, so the host lays the query bytes out in the order the
// code bits fall out of the packed words (fill_query_mfma in bbq_core.cpp).  The QUERIES are the A operand (M = 32 queries) and
// the index ROWS the B operand (N = 32 rows), so a lane's 16 accumulators of a row group are 16 queries against ONE row.
//
// Round 4: the kernel was bound by vector issue (profiles/r03_mfma_pmc.json: 1 349 vector instructions per tile and wave next to
// 48 MFMAs, the matrix cores 24 % busy) - 16 instructions per (row, query) pair of pre-filter and 9 per MFMA of bit -> int8
// expansion.  Both are gone (profiles/r04_mfma_pmc.json: 269 vector instructions and 30 MFMAs per tile and group of 32 queries):
//  * THE PRE-FILTER LIVES IN THE ACCUMULATOR.  "score > theta" is, for a row with upper > lower and a query with upper > lower,
//    an inequality on the integer itself:  qcDist > T(query, row), where T is a sum of four (per-query constant) x (per-row
//    constant) products (derivation at row_constants()).  The MFMA's C operand is initialised with the float
//        bias + slack(row) - S * T(query, row)
//    in a binade where one ulp is the grain of the threshold; the matrix cores add S * qcDist EXACTLY, and a pair can only be a
//    candidate if its accumulator ends above the bias: one v_max3_i32 per two pairs finds out whether a tile has any.  No
//    conversion, no compare per pair.  Tiles with a survivor (one in twelve in the largest segment) derive the start values of the
//    row group again and take qcDist from the difference.
//    The start values are themselves a contraction - C[query][row] = qk[query] . rk[row] + K[row] - and run on
//    v_mfma_f32_32x32x2_f32 (start_values(): three per row group), not on the vector ALUs.
//    The constants are f32 images of the EXACT f64 corrections (both layouts: the compact layout reads its side array exact[],
//    32 B/row more per 32 queries - this kernel is nowhere near the HBM bound), so no per-pair error terms exist: every rounding
//    is covered by a per-row slack of three grains, a few hundredths of a standard deviation of qcDist.
//  * QUERY VALUES <= 15 (queryBits <= 4) RUN ON v_mfma_f32_32x32x64_f8f6f4, FP6 x FP4 (template FP).  A 1-bit code IS an FP4
//    number where it stands: the nibble patterns 0001, 0010, 0100 are 0.5, 1.0 and 2.0 (e2m1), so  w & 0x11111111,  w & 0x22222222,
//    w & 0x44444444  and  (w >> 3) & 0x11111111  turn one 32-dimension word into the 32 FP4 operands of a lane: 5 vector
//    instructions per MFMA of 64 dimensions, no byte expansion.  The host stores the query values as FP6 (e2m3: four significant
//    bits, exactly q / 2, q / 4, q / 8 for q <= 15) scaled against the bit's weight, so every product is q / 4 and the f32 accumulator -
//    started in the binade [2^19, 2^20), on a grid of 1/16 - holds bias - T/4 + qcDist/4 exactly.  The instruction takes 64 dimensions
//    in the time v_mfma_i32_32x32x32_i8 takes for 32 (scripts/ubench/mfma_fp4_probe.hip: layout, exactness and rate measured on
//    gfx950): half the matrix-core time and half the expansion of the int8 form.
//    Query values up to 127 stay on v_mfma_i32_32x32x32_i8 with the {0, 1} byte expansion: there the start value's BITS (a float in
//    [2^23, 2^24): 0x4B000000 + its integer distance from 2^23) are the i32 accumulator.
//    One v_permlane32_swap hands BOTH row groups their operand words (own or half-wave partner's row).
//  * Both row groups' accumulators are live at once, so a query fragment is read from LDS once for two MFMAs, and the next tile's
//    loads are issued as soon as the contraction has consumed the codes - into the same registers.
//  * A workgroup serves TWO groups of 32 queries per tile load (template G): see the kernel.
// What binds it now (scripts/ubench/valu_mfma_overlap.hip, four waves per SIMD): a vector instruction issued in the shadow of the
// SAME wave's MFMA is nearly free (five per MFMA: + 13-19 %), but a wave in a vector-only phase and a wave in an MFMA phase barely
// overlap on one SIMD (a block of 320 v_fma_f32 and a block of 24 MFMAs: 660 ns together, 422 and 366 ns alone - also when every
// second wave runs one block ahead).  The contraction's shadows are full (the operand expansion), so the ~150 vector instructions
// outside it - the row's popcount and constants, the addresses of the next tile, the survivor test - add their time to the matrix
// cores' instead of hiding behind it (the counters: vector ALUs active 58 %, matrix cores busy 56 % of the SIMD time at 1.8 GHz).
// Packed f32 instructions (v_pk_fma_f32) are NOT used: on gfx950 they issue at half rate (MI355X_MICROARCH.md, cycle constants).
//
// The rare survivors go through the exact f64 score of the one-sweep kernel, so the emitted candidates - and therefore the
// results - are identical.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
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
typedef float f32x16m __attribute__((ext_vector_type(16)));
typedef int i32x8m __attribute__((ext_vector_type(8)));

constexpr int kMfmaQueries = 32;
// The accumulator's start value lies in a binade whose ulp is the grain of the threshold:
//   int8 form (S = 1):          [2^23, 2^24), ulp 1; the i32 accumulator holds the float's bits, 0x4B000000 + its integer distance from 2^23
//   FP form  (S = 1/4, 1/2, 1): [2^19, 2^20), ulp 1/16; the f32 accumulator holds the value.  The products are multiples of S, the start
//                               value a multiple of 1/16: every partial sum is a multiple of 1/16 below 2^20, i.e. exact in f32
// "bias" is the middle of the binade (1.5 x its lower end); the sum of the terms must stay below a quarter of the binade's lower end
// so that every partial sum stays inside it.
template <bool FP> struct MfmaNum;
template <> struct MfmaNum<false> {
  static constexpr float S = 1.0f, ulp = 1.0f, bias = 12582912.0f, mag_limit = 4000000.0f, pass_all = 12582912.0f + 2097152.0f;
};
template <> struct MfmaNum<true> {   // S is chosen per call: 1/4 for query values up to 15, 1/2 up to 7, 1 up to 3 (MfmaArgs::fp_scale)
  static constexpr float S = 0.25f, ulp = 0.0625f, bias = 786432.0f, mag_limit = 110000.0f, pass_all = 786432.0f + 131072.0f;
};

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

// conservative lower edge, in z-space, of "score > theta" for one query.
//   COSINE / MIP: z = s + xadd,  score = f(z + qadd - cdp) with f increasing
//   EUCLIDEAN   : z = 2s - xadd, score = 1/(1 + qadd - z)  increasing in z while the denominator is positive
// Returns zmin with:  exact f32 score > theta_score  =>  z > zmin.   -DBL_MAX accepts everything.
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

// ---- the threshold on the integer ------------------------------------------------------------------------------------------------
// With ax = lower, lx = upper - lower of the row, x1 its component sum, D the dimension, and ay, ly, y1 of the query
// (src/batchDotProduct.ts:478-617):
//     s = ay * (ax * D + lx * x1) + ly * (ax * y1 + lx * qc),       z = cs * s + ca * add      (cs, ca) = (2, -1) EUCLIDEAN, (1, 1) otherwise
// so for lx > 0 and ly > 0 (beta = cs * ly):
//     z > zth   <=>   qc > T = (zth / beta) * (1 / lx)  -  (ay / ly) * (rho * D + x1)  -  y1 * rho  -  (1 / beta) * (ca * add / lx),    rho = ax / lx.
// Per query (prologue, f64 -> f32):  qk = -S * {zth / beta, ay / ly, y1, 1 / beta}.   Per row (f32):  rk = {1 / lx, -(rho * D + x1), -rho, -ca * add / lx}.
// The accumulator starts at  K + qk . rk  with K = bias + slack: K first, then the four products one by one, the one with the row's
// popcount in it last (start_values(): the MFMAs that do not need the popcount are issued in front of it).
//
// Slack.  Let mag = S * sum_j max_q |q_j| * |r_j| (the row's magnitude budget, with |rho| * D + (a bound of |x1|) standing for |r_1|, plus
// S * the largest possible qcDist so that every term of s is covered).
//  * the sum: four fused multiply-adds onto K, each result inside the binade (guaranteed by mag < mag_limit, else the row is "weird"):
//    4 * 0.5 ulp (tests/test_prefilter_math_cpu.py checks this order, the products-first order and a plain FMA chain);
//  * the f32 images of the constants: q_j within 2^-24 (one rounding of an f64), 1 / lx within 2^-22 (v_rcp_f32 is good to 1 ulp, lx
//    itself is rounded once), rho and add / lx within 2^-21.4, r_1 within 2^-21.3 of |rho| * D + |x1|: every product within 2^-21 of its
//    magnitude, together below 2^-21 * mag;  the f64 evaluation of the reference score itself (~2^-50 of the same magnitudes) and
//    the inversion of the similarity transform (z_threshold's own allowance) are far inside.
// slack = (2.25 ulp + 2^-20 * mag) rounded up to whole ulps: above 2 ulp + twice the bound of the images.  A weird row (lx <= 0, non-finite, mag too large)
// gets zero constants and K = pass_all, a value inside the binade that no threshold reaches: every one of its pairs passes and is
// scored exactly, and the difference to the start value is still qcDist.  A query that accepts everything (no threshold yet) is not
// swept here at all: the prologue flags it, as its candidates would overflow every list of this kernel anyway.
struct RowK {
  float r0, r1, r2, r3, K;
};
// `x1` is what r1 is made of; `x1_bound` >= |x1| is what the magnitude budget uses - the dimension for a row whose component sum is
// its popcount: every constant but r1 is then known before the popcount, and the start-value MFMAs that do not need r1 run in front
// of it (start_values: K first)
template <bool FP>
__device__ __forceinline__ RowK row_constants_early(double al, double au, double add, float x1_bound, float D, int sim, const float *__restrict__ gmax, float &rho_out, bool &ok_out) {
  using N = MfmaNum<FP>;
  const float lxf = (float)(au - al), alf = (float)al, addf = (float)add;
  const float r0 = __builtin_amdgcn_rcpf(lxf);
  const float rho = alf * r0;
  RowK k;
  k.r0 = r0;
  k.r1 = 0.0f;
  k.r2 = -rho;
  k.r3 = (sim == 0 ? addf : -addf) * r0;
  // gmax: S * max over the workgroup's queries of |zth / beta|, |ay / ly|, |y1|, 1 / beta, and S * max sum of query values
  const float mag = fmaf(gmax[0], fabsf(r0), fmaf(gmax[1], fmaf(fabsf(rho), D, x1_bound), fmaf(gmax[2], fabsf(rho), fmaf(gmax[3], fabsf(k.r3), gmax[4]))));
  const bool ok = lxf > 0.0f && mag < N::mag_limit;  // NaN anywhere: not ok
  if (ok) {
    k.K = N::bias + N::ulp * ceilf(fmaf(mag, 9.5367431640625e-07f / N::ulp, 2.25f));  // whole ulps above 2.25 ulp + 2^-20 * mag: 3 for an ordinary row
  } else {
    k.r0 = k.r2 = k.r3 = 0.0f;
    k.K = N::pass_all;
  }
  rho_out = rho;
  ok_out = ok;
  return k;
}
struct MfmaArgs {
  ScanArgs s;
  const uint8_t *qbytes;   // int8 form: [groups][W*4 words][2 halves][32 queries][16 B] query values in fragment order; FP form: per group
                           // [W*2 steps][2 halves][32 queries][16 B] then [W*2][2][32][8 B]: the 32 FP6 values (24 B) of a lane and step
  const float *qmax;       // [groups][4]: S-free maxima over the group's queries: |ay / ly|, |y1|, 1 / (cs * ly), sum of the query's values
  int32_t nq_total;
  int32_t chunks_per_block;  // consecutive chunks one workgroup walks with the same queries (launch_mfma_t)
  float fp_scale;            // FP form: every product is fp_scale * q (1/4, 1/2 or 1: the largest the query values leave room for in e2m3)
  int32_t groups_per_block;  // 1 or 2 groups of 32 queries per workgroup: a tile is loaded (and its row constants derived) once for all of them
  int32_t stage_cap;         // candidates per query a workgroup stages in LDS (slot mode: the chunk slots' capacity)
  int32_t queue_cap;         // survivors per wave, tile and group that wait for their exact scores
};

// Pairs that pass the pre-filter are pushed to a per-wave LDS queue (packed qc | row-in-tile << 20 | query << 26) and
// scored exactly afterwards by ONE copy of the exact code, 64 pairs at a time.
constexpr int kMfmaQueueCap = 512;        // one group per workgroup
constexpr int kMfmaQueueCapTwo = 128;     // two groups per workgroup (append mode only): LDS for two workgroups per CU
constexpr int kMfmaStageCapTwo = 32;
constexpr int kMfmaMaxChunksPerBlock = 16;

template <int W, bool COMPACT>
struct TileRegs {
  u32x4m c[W];     // this lane's row: its code words
  f64x2m lu;       // {lower, upper}
  double add;      // additionalCorrection
  double x1;       // explicit quantizedComponentSum (has_x1 only)
};

template <int W, bool COMPACT>
__device__ __forceinline__ void load_tile_regs(TileRegs<W, COMPACT> &t, const IndexView &idx, int64_t tile, int lane) {
  const uint8_t *__restrict__ tp = idx.tiles + tile * (int64_t)idx.tile_stride;
#pragma unroll
  for (int j = 0; j < W; ++j) t.c[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4m *>(tp) + lane + j * kTileRows);
  const uint8_t *__restrict__ cr = tp + (size_t)W * (kTileRows * 16);
  if constexpr (COMPACT) {
    const double *__restrict__ ex = idx.exact + (tile * kTileRows + lane) * 4;
    t.lu = __builtin_nontemporal_load(reinterpret_cast<const f64x2m *>(ex));
    t.add = __builtin_nontemporal_load(ex + 2);
    t.x1 = 0.0;
  } else {
    t.lu = __builtin_nontemporal_load(reinterpret_cast<const f64x2m *>(cr) + lane);
    t.add = __builtin_nontemporal_load(reinterpret_cast<const double *>(cr + 1024) + lane);
    t.x1 = idx.has_x1 ? __builtin_nontemporal_load(reinterpret_cast<const double *>(cr + 1536) + lane) : 0.0;
  }
}

__device__ __forceinline__ int abits(float v) { return (int)__float_as_uint(v); }
__device__ __forceinline__ int abits(int v) { return v; }

// the 16 start values of a lane (16 queries x its row-group row): K + qk . rk as three chained v_mfma_f32_32x32x2_f32 on C = 0, K first
// (every addition rounds at the binade's ulp: four roundings, which the slack covers).  The same call on the same operands gives the
// same bits: the survivors' path calls it again.
__device__ __forceinline__ f32x16m start_values_early(float aq0, float aq1, float b0, float b1) {   // the two steps without the popcount
  f32x16m c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(aq0, b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(aq1, b1, c, 0, 0, 0);
  return c;
}
template <bool FP>
__device__ __forceinline__ typename std::conditional<FP, f32x16m, i32x16m>::type start_values_finish(f32x16m c, float aq2, float b2) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(aq2, b2, c, 0, 0, 0);
  if constexpr (FP) return c;
  else return __builtin_bit_cast(i32x16m, c);   // the int8 form accumulates onto the float's bits
}
template <bool FP>
__device__ __forceinline__ typename std::conditional<FP, f32x16m, i32x16m>::type start_values(float aq0, float aq1, float aq2, float b0, float b1, float b2) {
  return start_values_finish<FP>(start_values_early(aq0, aq1, b0, b1), aq2, b2);
}

// ---- one k-step of the FP form, as a template over the step index: the fragment reads are inline asm (ds_read with an immediate
// offset, waited for by hand) because they must stay where they are written - one step ahead of their use.  Left to the compiler
// every step's fragment is read ahead of the loop (12 x 6 registers at 768-d: spills); volatile reads leave the LDS address space.
template <int G, int STEPS, int W>
__device__ __forceinline__ void fp_step(f32x16m &acc0, f32x16m &acc1, const u32x4m (&c)[W], u32x4m &bq, u32x2m &bq2, uint32_t addr16, uint32_t addr8) {
  // a lane supplies 32 of a step's 64 dimensions: ONE code word - word 2G of its row group's row n in the lower half-wave, word
  // 2G + 1 in the upper.  The tile's words were swapped in place for that (swap_code_words): [0] = {row n: 2G | row n: 2G + 1}, [1] =
  // the same of row 32 + n
  const uint32_t sw[2] = {(G & 1) == 0 ? c[G >> 1].x : c[G >> 1].z, (G & 1) == 0 ? c[G >> 1].y : c[G >> 1].w};
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq), "+v"(bq2));   // this step's fragment has arrived
  i32x8m Q, R0, R1;   // 8 dwords wide for the builtin; FP6 uses 6 of them and FP4 4: the others stay undefined
  Q[0] = (int)bq.x; Q[1] = (int)bq.y; Q[2] = (int)bq.z; Q[3] = (int)bq.w; Q[4] = (int)bq2.x; Q[5] = (int)bq2.y;
  if constexpr (G + 1 < STEPS) {  // the next step's fragment, one step ahead: the read is issued BEFORE this step's MFMAs (their operand
                                  // passes through an asm behind it, or the compiler issues them first to reuse the registers)
    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%5"
                 : "=v"(bq), "=v"(bq2) : "v"(addr16), "v"(addr8), "n"((G + 1) * 1024), "n"((G + 1) * 512));
    asm volatile("" : "+v"(Q));
  }
  // bits 0, 1, 2 of every nibble where they stand are the FP4 numbers 0.5, 1.0, 2.0; bit 3 would be the sign: it moves to bit 0
  R0[0] = (int)(sw[0] & 0x11111111u); R0[1] = (int)(sw[0] & 0x22222222u); R0[2] = (int)(sw[0] & 0x44444444u); R0[3] = (int)((sw[0] >> 3) & 0x11111111u);
  R1[0] = (int)(sw[1] & 0x11111111u); R1[1] = (int)(sw[1] & 0x22222222u); R1[2] = (int)(sw[1] & 0x44444444u); R1[3] = (int)((sw[1] >> 3) & 0x11111111u);
  // A = queries (FP6 e2m3, cbsz 2), B = rows (FP4 e2m1, blgp 4), block scales 2^0
  acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Q, R0, acc0, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Q, R1, acc1, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  asm volatile("" : "+v"(acc0), "+v"(acc1));  // ordered with the reads: this step's MFMAs are issued before the next step begins
  if constexpr (G + 1 < STEPS) fp_step<G + 1, STEPS, W>(acc0, acc1, c, bq, bq2, addr16, addr8);
}

// FP form: swap(word 2s, word 2s + 1) of every lane's own row, in place, ONCE per tile: afterwards the first register of each pair
// holds {rows 0..31: word 2s | rows 0..31: word 2s + 1}, the second the same of rows 32..63 - the B operands of step s for the two row
// groups, for every group of queries the workgroup serves (per group and step it was a swap and, for all but the last group, two
// copies: the swap overwrites both of its operands)
template <int W>
__device__ __forceinline__ void swap_code_words(u32x4m (&c)[W]) {
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(c[j].x, c[j].y, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(c[j].z, c[j].w, false, false);
    c[j].x = s0[0]; c[j].y = s0[1]; c[j].z = s1[0]; c[j].w = s1[1];
  }
}

// bytes of one group's staged query operands
__host__ __device__ constexpr int mfma_query_bytes(int w16, bool fp) { return fp ? w16 * 2 * 2 * 32 * 24 : w16 * 4 * 2 * 32 * 16; }

// 4 waves per SIMD = 2 workgroups per CU (128 VGPRs) up to 1024-d; 1536-d rows hold 48 code registers: one workgroup per CU.
// A workgroup serves groups_per_block groups of 32 queries (grid y = pairs of groups): the tile's codes stay in registers and its row
// constants are derived once, then each group runs its own contraction and test.  With ONE group per tile load the 10 M-row sweep reads
// 120 B per row and 32 queries - 1.2 GB in ~175 us, within a few per cent of what the memory side delivers - so both bounds are met at
// once; two groups halve the bytes, the popcounts, the conversions and the loads per query.
template <int W, bool COMPACT, bool FP, int G>
__global__ __launch_bounds__(kChunkRows, W <= 8 ? 4 : 2) void bbq_scan_mfma_kernel(const MfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using N = MfmaNum<FP>;
  using Acc = typename std::conditional<FP, f32x16m, i32x16m>::type;
  constexpr int NT = kChunkRows;
  constexpr int NW = kChunkRows / 64;
  constexpr int WORDS = W * 4;
  constexpr int STEPS = FP ? WORDS / 2 : WORDS;                                      // k-steps of 64 (FP) or 32 dimensions
  constexpr int QBYTES = mfma_query_bytes(W, FP);
  const float S = FP ? a.fp_scale : 1.0f;                                            // accumulator units per qcDist unit
  constexpr int gpb = G, nqb = gpb * kMfmaQueries;                                   // query slots of this workgroup (G = a.groups_per_block)
  const int scap = a.stage_cap, qcap = a.queue_cap;
  // per group: u32x4 [STEPS*2][32] 16 B per lane and step; FP: then u32x2 [STEPS*2][32], the rest of the lane's 32 FP6 values
  f32x4m *s_qk = reinterpret_cast<f32x4m *>(smem + (size_t)gpb * QBYTES);            // [nqb] per-query constants -S * {zth/beta, ay/ly, y1, 1/beta}
  f64x2m *s_lu = reinterpret_cast<f64x2m *>(s_qk + nqb);                             // [NW][64] {lower, upper} of every tile row, for the survivors' exact scores ...
  f64x2m *s_ax = s_lu + NW * 64;                                                     // [NW][64] ... and {additionalCorrection, component sum}: no trip to memory there
  uint32_t *s_queue = reinterpret_cast<uint32_t *>(s_ax + NW * 64);                  // [NW][qcap]
  uint32_t *s_qcount = s_queue + NW * qcap;                                          // [NW] (+ padding to 16 B)
  QueryParams *s_qp = reinterpret_cast<QueryParams *>(s_qcount + 8);                 // [nqb]
  float *s_gmax = reinterpret_cast<float *>(s_qp + nqb);                             // [8]: the workgroup's magnitude maxima (row_constants)
  uint32_t *s_theta = reinterpret_cast<uint32_t *>(s_gmax + 8);                      // [nqb]
  uint32_t *s_cnt = s_theta + nqb;                                                   // [nqb]
  uint64_t *s_ent = reinterpret_cast<uint64_t *>(s_cnt + nqb);                       // [nqb][scap]

  const int groups_total = (a.nq_total + kMfmaQueries - 1) / kMfmaQueries;
  const int group0 = blockIdx.y * gpb;
  const int ng = min(gpb, groups_total - group0);                                    // groups present here
  const int q0 = group0 * kMfmaQueries;
  const int nb = min(nqb, a.nq_total - q0);                                          // queries present here
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.s.idx.n_rows + kTileRows - 1) / kTileRows;
  // A workgroup is persistent over chunks_per_block consecutive chunks of the same queries: the query fragments and constants are
  // staged once, and a wave's next tile is loaded while the current one is tested.  The first tile's loads are in flight during the
  // prologue.
  const int cpb = a.chunks_per_block;
  const int lc0 = blockIdx.x * cpb;
  TileRegs<W, COMPACT> t;
  {
    const int64_t t0 = (a.s.chunk_begin + lc0) * kTilesPerChunk + wave;
    if (lc0 < a.s.n_chunks && t0 < n_tiles) load_tile_regs<W, COMPACT>(t, a.s.idx, t0, lane);
  }
  // prologue: wave 0 derives the queries' constants (a few f64 divisions per query) while the other waves stage the query fragments
  if (wave == 0) {
    if (lane < 8) s_gmax[lane] = 0.0f;   // same wave as the atomicMax below: LDS operations of a wave execute in order
    if (lane < NW) s_qcount[lane] = 0;
    if (lane < nqb) {
      QueryParams p{};
      uint32_t th = 0xFFFFFFFFu;
      f32x4m qk;
      // lanes without a query: the largest finite threshold - nothing of an ordinary row passes (a weird row's pairs are dropped later)
      qk.x = -3.0e38f; qk.y = 0.0f; qk.z = 0.0f; qk.w = 0.0f;
      if (lane < nb) {
        p = a.s.qparams[q0 + lane];
        th = a.s.theta[q0 + lane];
        const double zt = z_threshold(th, p);
        const double beta = (p.sim == 0 ? 2.0 : 1.0) * p.ly;   // > 0 and finite: the host sends no other query here (mfma_query_ok)
        const double A = -(double)S * (zt / beta);
        if (A < 1.0e30) {  // (also false for NaN)
          qk.x = A > -3.0e38 ? (float)A : -3.0e38f;            // a threshold beyond every score: nothing passes
          qk.y = (float)(-(double)S * (p.ay / p.ly));
          qk.z = (float)(-(double)S * p.y1);
          qk.w = (float)(-(double)S / beta);
          if (fabsf(qk.x) < 1.0e30f) atomicMax(reinterpret_cast<uint32_t *>(s_gmax), __float_as_uint(fabsf(qk.x) * 1.0000002f));  // non-negative floats order like their bits
        } else {
          // no threshold yet (or one below every score): every pair of this query would be a candidate, far more than the lists of
          // this kernel hold - the query is flagged here and gets a sweep of its own (finish_replay), its lane sweeps nothing
          atomicOr(a.s.flags + q0 + lane, kFlagOverflow);
        }
      }
      s_qp[lane] = p;
      s_theta[lane] = th;
      s_cnt[lane] = 0;
      s_qk[lane] = qk;
      if (lane == 0) {  // the host's maxima are S-free and rounded up; the workgroup's groups share one budget
        float g1 = 0.0f, g2 = 0.0f, g3 = 0.0f, g4 = 0.0f;
        for (int g = 0; g < ng; ++g) {
          const float *__restrict__ gm = a.qmax + (size_t)(group0 + g) * 4;
          g1 = fmaxf(g1, gm[0]); g2 = fmaxf(g2, gm[1]); g3 = fmaxf(g3, gm[2]); g4 = fmaxf(g4, gm[3]);
        }
        s_gmax[1] = S * g1;
        s_gmax[2] = S * g2;
        s_gmax[3] = S * g3;
        s_gmax[4] = S * g4;
      }
    }
  } else {
    const u32x4m *__restrict__ gb = reinterpret_cast<const u32x4m *>(a.qbytes + (size_t)group0 * QBYTES);
    u32x4m *__restrict__ sb = reinterpret_cast<u32x4m *>(smem);
    for (int i = tid - 64; i < ng * (QBYTES / 16); i += NT - 64) sb[i] = gb[i];
  }
  __syncthreads();

  const int sim = s_qp[0].sim;                  // uniform over the call
  const float Df = (float)a.s.idx.dim;
#pragma unroll 1
  for (int ci = 0; ci < cpb; ++ci) {
  const int lc = lc0 + ci;           // chunk index inside this launch
  if (lc >= a.s.n_chunks) break;     // block-uniform
  const int64_t chunk = a.s.chunk_begin + lc;
  // (named scalar: left alone the tile number lives in a pair of vector registers and everything derived from it - the rows of the last
  // tile, the next tile's address - is 64-bit vector arithmetic)
  const int64_t tile_v = chunk * kTilesPerChunk + wave;
  const int64_t tile = ((int64_t)__builtin_amdgcn_readfirstlane((int)(tile_v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_v);

  if (tile < n_tiles) {  // wave-uniform
    // ---- my row's constants (tile row `lane`), then both row groups' through one swap each: [0] = row n, [1] = row 32 + n
    // (every constant but r1 first: the start-value MFMAs that do not need the popcount are issued in front of it)
    float rho;
    bool row_ok;
    RowK mine = row_constants_early<FP>(t.lu.x, t.lu.y, t.add, a.s.idx.has_x1 ? fabsf((float)t.x1) * 1.0000002f : Df, Df, sim, s_gmax, rho, row_ok);
    // (K first; the last step is the only one that needs the row's popcount)
    const auto b0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine.K), __float_as_uint(mine.r0), false, false);
    const auto b1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine.r2), __float_as_uint(mine.r3), false, false);
    // the first group's first two steps, both row groups: four MFMAs of 64 cycles in whose shadow the popcount runs.  The empty asm ties
    // the code words to their results: the popcount cannot be scheduled in front of them (left alone the compiler puts it there)
    int lds_off0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lds_off0));
    float aqf[3];
    {
      const f32x4m qkm = s_qk[n + lds_off0];
      aqf[0] = h ? qkm.x : 1.0f;    // step 0: K x 1 + qk[0] r0
      aqf[1] = h ? qkm.w : qkm.z;   // step 1: qk[2] r2 + qk[3] r3
      aqf[2] = h ? 0.0f : qkm.y;    // step 2: qk[1] r1 (the row's popcount is in r1)
    }
    f32x16m e0 = start_values_early(aqf[0], aqf[1], __uint_as_float(b0[0]), __uint_as_float(b1[0]));
    f32x16m e1 = start_values_early(aqf[0], aqf[1], __uint_as_float(b0[1]), __uint_as_float(b1[1]));
#pragma unroll
    for (int j = 0; j < W; ++j) asm volatile("" : "+v"(e0), "+v"(e1), "+v"(t.c[j]));
    uint32_t ones = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) ones += __popc(t.c[j].x) + __popc(t.c[j].y) + __popc(t.c[j].z) + __popc(t.c[j].w);
    double x1row = (double)ones;                 // quantizedComponentSum of a 1-bit row is its popcount ...
    if (a.s.idx.has_x1) x1row = t.x1;            // ... unless the index says otherwise
    mine.r1 = row_ok ? -fmaf(rho, Df, (float)x1row) : 0.0f;
    s_lu[wave * 64 + lane] = t.lu;               // for the survivors' exact scores (any lane may score any row of the tile)
    {
      f64x2m ax;
      ax.x = t.add; ax.y = x1row;
      s_ax[wave * 64 + lane] = ax;
    }
    // ---- the accumulators start at the (negated, biased) thresholds of their pairs:  C[query][row] = qk[query] . rk[row] + K[row],
    // itself a contraction (six terms: four products, K x 1, 0 x 0) - three v_mfma_f32_32x32x2_f32 per row group, which evaluate it
    // as a chain of f32 FMAs (as 128 v_fma_f32 per tile and wave it was a third of the kernel's vector instructions).  The B operand
    // of a k-step is {one constant in the lower half-wave, another in the upper} of the row group's row n: one swap of my own row's two
    // constants gives both row groups' operands ([0]: rows 0..31, [1]: rows 32..63).
    const auto b2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine.r1), 0u, false, false);
    float rb[2][3];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) { rb[rg][0] = __uint_as_float(b0[rg]); rb[rg][1] = __uint_as_float(b1[rg]); rb[rg][2] = __uint_as_float(b2[rg]); }
    const int rows_here = (int)min((int64_t)kTileRows, a.s.idx.n_rows - tile * kTileRows);
    uint32_t *__restrict__ queue = s_queue + (size_t)wave * qcap;
    if constexpr (FP) swap_code_words<W>(t.c);   // (the popcount above wanted the words as they were loaded)
    // (straight-line code over the groups: as a loop the tile's registers were carried around it and spilled.  A group that does not
    // exist - the last workgroup row of an odd number of groups - is computed all the same: its queries' start values are -inf)
#pragma unroll
    for (int g = 0; g < G; ++g) {
    // opaque zero, redefined for every group of every tile: keeps the compiler from hoisting the query-fragment LDS reads out of the loops
    int lds_off;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lds_off));
    const int gq = g * kMfmaQueries;             // first query slot of this group
    const int nbg = min(kMfmaQueries, nb - gq);  // its queries (<= 0: none)
    const u32x4m *__restrict__ s_B = reinterpret_cast<const u32x4m *>(smem + (size_t)g * QBYTES);
    const u32x2m *__restrict__ s_B2 = reinterpret_cast<const u32x2m *>(smem + (size_t)g * QBYTES + (size_t)STEPS * 2 * 32 * 16);
    // A operands of the start-value contraction (start_values): query n's constants, k = 0 in the lower half-wave, k = 1 in the upper
    float aq0, aq1, aq2;
    Acc acc0, acc1;
    if (g == 0) {   // (compile time: the groups are unrolled) begun in front of the popcount
      aq0 = aqf[0]; aq1 = aqf[1]; aq2 = aqf[2];
      acc0 = start_values_finish<FP>(e0, aq2, rb[0][2]);
      acc1 = start_values_finish<FP>(e1, aq2, rb[1][2]);
    } else {
      const f32x4m qkm = s_qk[gq + n + lds_off];
      aq0 = h ? qkm.x : 1.0f;
      aq1 = h ? qkm.w : qkm.z;
      aq2 = h ? 0.0f : qkm.y;
      acc0 = start_values<FP>(aq0, aq1, aq2, rb[0][0], rb[0][1], rb[0][2]);
      acc1 = start_values<FP>(aq0, aq1, aq2, rb[1][0], rb[1][1], rb[1][2]);
    }
    // ---- the contraction: C[m = query][n = row of the group] += sum over the k-steps
    if constexpr (FP) {
      // LDS byte addresses of this lane's fragment of step 0 (the pointers are LDS pointers: their low 32 bits are the offset)
      const uint32_t addr16 = (uint32_t)(uintptr_t)(s_B + h * 32 + n + lds_off), addr8 = (uint32_t)(uintptr_t)(s_B2 + h * 32 + n + lds_off);
      u32x4m bq;
      u32x2m bq2;
      // (the swapped words are the same for every group: named here, or their 96 masked forms are computed once and kept - spilled)
#pragma unroll
      for (int j = 0; j < W; ++j) asm volatile("" : "+v"(t.c[j]));
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3" : "=v"(bq), "=v"(bq2) : "v"(addr16), "v"(addr8));
      fp_step<0, STEPS, W>(acc0, acc1, t.c, bq, bq2, addr16, addr8);
    } else {
#pragma unroll
      for (int w8 = 0; w8 < WORDS; ++w8) {
        const uint32_t w = (w8 & 3) == 0 ? t.c[w8 >> 2].x : (w8 & 3) == 1 ? t.c[w8 >> 2].y : (w8 & 3) == 2 ? t.c[w8 >> 2].z : t.c[w8 >> 2].w;
        // swap(w, w >> 4): [0] = {row n's word | row n's word >> 4}, [1] = {row 32+n's word | row 32+n's word >> 4}: the lower half-wave
        // supplies the low nibble of every byte, the upper one the high nibble, of the row group's row n
        const auto sw = __builtin_amdgcn_permlane32_swap(w, w >> 4, false, false);
        const u32x4m bq = s_B[(w8 * 2 + h) * 32 + n + lds_off];
        i32x4m Q;
        Q.x = (int)bq.x; Q.y = (int)bq.y; Q.z = (int)bq.z; Q.w = (int)bq.w;
        i32x4m R0, R1;
        R0.x = (int)(sw[0] & 0x01010101u); R0.y = (int)((sw[0] >> 1) & 0x01010101u); R0.z = (int)((sw[0] >> 2) & 0x01010101u); R0.w = (int)((sw[0] >> 3) & 0x01010101u);
        R1.x = (int)(sw[1] & 0x01010101u); R1.y = (int)((sw[1] >> 1) & 0x01010101u); R1.z = (int)((sw[1] >> 2) & 0x01010101u); R1.w = (int)((sw[1] >> 3) & 0x01010101u);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Q, R0, acc0, 0, 0, 0);   // A = queries (m), B = rows (n)
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Q, R1, acc1, 0, 0, 0);
      }
    }
    // ---- the codes are consumed by the last group: the next tile of this wave slot goes into the same registers while this one is tested
    if (g + 1 == G) {
      const int64_t tn = tile + kTilesPerChunk;
      // (the lane passes through an empty asm: its share of the addresses is derived here, a few instructions per tile, instead of
      // being carried around the loop in four registers that the two-group kernel does not have)
      int lane_here = lane;
      asm volatile("" : "+v"(lane_here));
      if (ci + 1 < cpb && lc + 1 < a.s.n_chunks && tn < n_tiles) load_tile_regs<W, COMPACT>(t, a.s.idx, tn, lane_here);
    }
    // ---- any accumulator above the bias?
    // (compared as integers in both forms: positive floats order like their bits, a negative start value - a lane without a query - is
    // a negative integer, NaN cannot arise; v_max3_i32 takes two accumulators per instruction, the f32 maximum with its NaN rules one)
    // (written as a chain max(max(m, a), b): one v_max3_i32 per two accumulators - as max(m, max(a, b)) it compiled to three
    // instructions per four)
    int mx = max(max(abits(acc0[0]), abits(acc0[1])), abits(acc0[2]));
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max(max(mx, abits(acc0[r])), abits(acc0[r + 1]));
    mx = max(max(mx, abits(acc0[15])), abits(acc1[0]));
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max(max(mx, abits(acc1[r])), abits(acc1[r + 1]));
    mx = max(mx, abits(acc1[15]));
    const bool any_lane = mx > (int)__float_as_uint(N::bias);
    if (__any(any_lane)) {
      // survivors: take qcDist from the difference to the accumulator's start value, which is derived again (the same instructions on
      // the same operands: the same bits).  The operands pass through an empty asm so that the compiler cannot keep all 32 start values
      // alive across the contraction instead.
      float rc[2][3];
#pragma unroll
      for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          rc[rg][j] = rb[rg][j];
          asm volatile("" : "+v"(rc[rg][j]));
        }
      // (the lane's coordinates as well: everything derived from them here - 32 queue words, 16 query-exists masks - would otherwise be
      // hoisted out of the chunk loop and held in registers for the whole kernel)
      int h2 = h, n2 = n;
      asm volatile("" : "+v"(h2), "+v"(n2));
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        const int rit = 32 * rg + n2;
        const Acc &accv = rg == 0 ? acc0 : acc1;
        bool any_rg = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) any_rg = any_rg || abits(accv[r]) > (int)__float_as_uint(N::bias);
        if (!__any(any_rg)) continue;  // wave-uniform
        const Acc init = start_values<FP>(aq0, aq1, aq2, rc[rg][0], rc[rg][1], rc[rg][2]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const auto av = accv[r];
          const bool above = abits(av) > (int)__float_as_uint(N::bias);
          if (__any(above)) {  // wave-uniform: most accumulators have no survivor in any lane
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h2;
            if (above && m < nbg && rit < rows_here) {
              uint32_t qc;
              if constexpr (FP) qc = (uint32_t)((av - init[r]) * (1.0f / S));   // both on the 1/16 grid inside one binade: exact (S is a power of two)
              else qc = (uint32_t)(av - init[r]);
              const uint32_t slot = atomicAdd(&s_qcount[wave], 1u);
              if (slot < (uint32_t)qcap) queue[slot] = qc | ((uint32_t)rit << 20) | ((uint32_t)m << 26);
              else atomicOr(a.s.flags + q0 + gq + m, kFlagOverflow);  // more survivors than the queue holds: this query goes dense
            }
          }
        }
      }
      // ---- exact scores of the survivors (same-wave LDS traffic: program order is enough)
      const uint32_t n_pass = min(s_qcount[wave], (uint32_t)qcap);
      for (uint32_t i = lane; i < n_pass; i += 64) {
        const uint32_t e = queue[i];
        const int qc = (int)(e & 0xFFFFFu), rit = (int)((e >> 20) & 63u), qn = gq + (int)(e >> 26);
        const QueryParams pq = s_qp[qn];
        const int64_t row = tile * kTileRows + rit;
        const f64x2m lu = s_lu[wave * 64 + rit], ax = s_ax[wave * 64 + rit];
        const double lo = lu.x, up = lu.y, ad = ax.x, x1d = ax.y;
        const double s64 = m_score_f64((double)qc, lo, up, ad, x1d, pq);
        const float s32 = (float)s64;
        const uint32_t bits = __float_as_uint(s32);
        if (s32 != s32) atomicOr(a.s.flags + q0 + qn, kFlagNaN);
        else if (key_of_bits(bits) > s_theta[qn]) {
          const uint64_t ent = ((uint64_t)(uint32_t)(a.s.row_id_base + row) << 32) | bits;
          // staged in the workgroup's LDS lists.  Slot mode: written row-ordered after every chunk.  Append mode: flushed once, when the
          // workgroup is done - one atomic per query and workgroup reserves the room (an atomic that returns a value is a trip to the
          // memory side: per survivor it cost the early segments, where a few per cent of the pairs survive, more than the sweep itself)
          const uint32_t slot = atomicAdd(&s_cnt[qn], 1u);
          if (slot < (uint32_t)scap) s_ent[(size_t)qn * scap + slot] = ent;
          else if (a.s.append_lists) {  // the staging list is full: straight into the query's list
            const uint32_t gs = atomicAdd(a.s.append_counts + (size_t)(q0 + qn) * kAppendStride, 1u);
            const int64_t at = (int64_t)a.s.append_base[2 * (q0 + qn)] + gs;
            if (at < a.s.append_cap) a.s.append_lists[(size_t)(q0 + qn) * a.s.append_cap + at] = ent;
            else atomicOr(a.s.flags + q0 + qn, kFlagOverflow);
          }
        }
      }
      s_qcount[wave] = 0;  // this wave's queue is its own: ready for its next group or tile
    }
    }  // groups of this workgroup
  }
  if (a.s.append_lists) continue;  // workgroup-uniform: nothing to flush per chunk, nobody to wait for
  __syncthreads();
  for (int b = wave; b < nb; b += NW) {  // each wave writes the lists of its share of the queries
    uint32_t cnt = s_cnt[b];
    if (cnt > (uint32_t)scap) {
      if (lane == 0) atomicOr(a.s.flags + q0 + b, kFlagOverflow);
      cnt = (uint32_t)scap;
    }
    const uint64_t *__restrict__ src = s_ent + (size_t)b * scap;
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
  if (tid < nqb) s_cnt[tid] = 0;
  __syncthreads();                       // ... and sees them cleared before the next chunk's survivors arrive
  }  // chunks of this workgroup
  if (a.s.append_lists) {  // append mode: the workgroup's staged candidates go to the queries' lists (unordered inside the segment; the
                           // finalize launch takes its keys from there and the rare host replay sorts)
    __syncthreads();
    for (int b = wave; b < nb; b += NW) {
      const uint32_t cnt = min(s_cnt[b], (uint32_t)scap);
      if (cnt == 0) continue;  // wave-uniform
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(a.s.append_counts + (size_t)(q0 + b) * kAppendStride, cnt);
      base = __builtin_amdgcn_readfirstlane(base);
      const int64_t at0 = (int64_t)a.s.append_base[2 * (q0 + b)] + base;
      const uint64_t *__restrict__ src = s_ent + (size_t)b * scap;
      uint64_t *__restrict__ out = a.s.append_lists + (size_t)(q0 + b) * a.s.append_cap;
      for (uint32_t i = lane; i < cnt; i += 64) {
        if (at0 + i < a.s.append_cap) out[at0 + i] = src[i];
        else atomicOr(a.s.flags + q0 + b, kFlagOverflow);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------

static size_t mfma_smem_bytes(int w16, int stage_cap, bool fp, int gpb, int queue_cap) {
  constexpr int NW = kChunkRows / 64;
  const size_t nqb = (size_t)gpb * kMfmaQueries;
  return (size_t)gpb * mfma_query_bytes(w16, fp) + nqb * 16 + (size_t)NW * 64 * 32 + (size_t)NW * queue_cap * 4 + 32 +
         nqb * (sizeof(QueryParams) + 4 + 4) + 32 + nqb * (size_t)stage_cap * 8 + 64;
}

// Two groups per workgroup where the staging fits the CU as before (two workgroups per CU up to 1024-d, one beyond) - in append mode
// only, where the staged lists may be shorter than the chunk slots (a full list overflows into the query's list in memory)
// (the int8 form up to 1024-d stays at one group: with two its fragment reads are scheduled across the groups and spill)
constexpr bool mfma_two_groups_built(int w16, bool fp) { return fp || w16 > 8; }
static int mfma_groups_per_block(const ScanArgs &a, int groups, bool fp) {
  if (groups < 2 || !a.append_lists || !mfma_two_groups_built(a.idx.w16, fp)) return 1;
  // the chunk slots' capacity says how many candidates the plan expects here (cap_for: lam + 8 sqrt(lam) + 16 for lam candidates per
  // chunk and query): up to 64 slots a tile and group has 4 lam <= 62 survivors on average, eight standard deviations below the queue
  if (a.cap > 64) return 1;
  const size_t two = mfma_smem_bytes(a.idx.w16, std::min<int>(a.cap, kMfmaStageCapTwo), fp, 2, kMfmaQueueCapTwo);
  const size_t lds_alloc = (two + 1279) / 1280 * 1280;  // LDS is handed out in 1280-byte blocks
  return (a.idx.w16 <= 8 ? 2 * lds_alloc <= 160 * 1024 : two <= 150 * 1024) ? 2 : 1;
}

template <int W, bool COMPACT, bool FP>
static hipError_t launch_mfma_t(MfmaArgs a, int nq, int nc, hipStream_t s) {
  const int groups = (nq + kMfmaQueries - 1) / kMfmaQueries;
  a.groups_per_block = mfma_groups_per_block(a.s, groups, FP);
  a.stage_cap = a.groups_per_block == 2 ? std::min<int>(a.s.cap, kMfmaStageCapTwo) : a.s.cap;
  a.queue_cap = a.groups_per_block == 2 ? kMfmaQueueCapTwo : kMfmaQueueCap;
  const size_t smem = mfma_smem_bytes(W, a.stage_cap, FP, a.groups_per_block, a.queue_cap);
  // The chip holds 512 of these workgroups at a time (2 per CU) and a workgroup's prologue (staging the queries: 18 KB per group at
  // 768-d, the constants) costs about one chunk's worth of time.  Up to 12 chunks per workgroup: ONE round of workgroups, everything
  // resident at once.  Longer sweeps: about sqrt(2 x prologue x chunks / slots) chunks each, where the prologues and the idle tail of
  // the last workgroups cost the same.
  const int grid_y = (groups + a.groups_per_block - 1) / a.groups_per_block;
  const int64_t units = (int64_t)nc * grid_y;
  int64_t cpb = (units + 511) / 512;
  if (cpb > 12) cpb = std::min<int64_t>(kMfmaMaxChunksPerBlock, std::max<int64_t>(4, (int64_t)llround(sqrt((double)units / 220.0))));
  a.chunks_per_block = (int)cpb;
  dim3 grid((unsigned)((nc + a.chunks_per_block - 1) / a.chunks_per_block), (unsigned)grid_y, 1), block(kChunkRows, 1, 1);
  auto kern = bbq_scan_mfma_kernel<W, COMPACT, FP, 1>;
  if constexpr (mfma_two_groups_built(W, FP)) {
    if (a.groups_per_block == 2) kern = bbq_scan_mfma_kernel<W, COMPACT, FP, 2>;
  }
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, block, smem, s, a);
  return hipGetLastError();
}

template <int W>
static hipError_t launch_mfma_w(const MfmaArgs &a, bool compact, bool fp, int nq, int nc, hipStream_t s) {
  if (compact) return fp ? launch_mfma_t<W, true, true>(a, nq, nc, s) : launch_mfma_t<W, true, false>(a, nq, nc, s);
  return fp ? launch_mfma_t<W, false, true>(a, nq, nc, s) : launch_mfma_t<W, false, false>(a, nq, nc, s);
}

bool mfma_sweep_supported(const ScanArgs &a) {
  const int w = a.idx.w16;
  if (a.idx.store_bits != 1 || !(w == 1 || w == 6 || w == 8 || w == 12)) return false;
  return mfma_smem_bytes(w, a.cap, false, 1, kMfmaQueueCap) <= 150 * 1024;
}

int64_t mfma_query_bytes_per_group(int w16, bool fp) { return mfma_query_bytes(w16, fp); }

int mfma_queries_per_tile_load(const ScanArgs &a, int n_queries, bool fp) {
  return kMfmaQueries * mfma_groups_per_block(a, (n_queries + kMfmaQueries - 1) / kMfmaQueries, fp);
}

hipError_t launch_scan_mfma(const ScanArgs &sa, const uint8_t *qbytes, const float *qmax, float fp_scale, int n_queries, int n_chunks, hipStream_t s) {
  if (n_chunks <= 0 || n_queries <= 0) return hipSuccess;
  const bool fp = fp_scale > 0.0f;
  MfmaArgs a{sa, qbytes, qmax, n_queries, 1, fp_scale, 1, sa.cap, kMfmaQueueCap};
  const bool compact = sa.idx.layout == kLayoutCompact;
  switch (sa.idx.w16) {
    case 1: return launch_mfma_w<1>(a, compact, fp, n_queries, n_chunks, s);
    case 6: return launch_mfma_w<6>(a, compact, fp, n_queries, n_chunks, s);
    case 8: return launch_mfma_w<8>(a, compact, fp, n_queries, n_chunks, s);
    case 12: return launch_mfma_w<12>(a, compact, fp, n_queries, n_chunks, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bbq
