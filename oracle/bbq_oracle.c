/*
 * bbq_oracle.c - CPU restatement of the reference's scoring + top-k search path.
 * TEST INFRASTRUCTURE ONLY (see bbq_oracle.h).  Parity status: PINNED by tests/golden.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * Citations are relative to /root/reference/.
 */
#include "bbq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------ JS number helpers */

/* Math.min / Math.max: NaN-propagating, -0 < +0 */
static double js_min(double a, double b) {
  if (a != a || b != b) return NAN;
  if (a == 0.0 && b == 0.0) return (signbit(a) || signbit(b)) ? -0.0 : 0.0;
  return a < b ? a : b;
}
static double js_max(double a, double b) {
  if (a != a || b != b) return NAN;
  if (a == 0.0 && b == 0.0) return (signbit(a) && signbit(b)) ? -0.0 : 0.0;
  return a > b ? a : b;
}
/* src/utils.ts:79-81 */
static double js_clamp(double x, double lo, double hi) { return js_min(js_max(x, lo), hi); }
/* Math.round: nearest, ties toward +inf, exact for 0.49999999999999994 */
static double js_round(double x) {
  if (x != x || isinf(x)) return x;
  double r = floor(x);
  if (x - r >= 0.5) r += 1.0;
  return r;
}
/* store into a Uint8Array element: ToUint8 (NaN/inf -> 0, else trunc modulo 256) */
static uint8_t js_to_uint8(double x) {
  if (x != x || isinf(x)) return 0;
  double t = trunc(x);
  double m = fmod(t, 256.0);
  if (m < 0) m += 256.0;
  return (uint8_t)m;
}

/* ------------------------------------------------------------------ vector helpers */

/* src/vectorOperations.ts:11-34 */
void orc_normalize(const float *v, int dim, float *out) {
  double norm = 0;
  for (int i = 0; i < dim; i++) norm += (double)v[i] * (double)v[i];
  norm = sqrt(norm);
  if (norm == 0) { for (int i = 0; i < dim; i++) out[i] = 0.0f; return; }
  for (int i = 0; i < dim; i++) out[i] = (float)((double)v[i] / norm);
}

/* src/vectorOperations.ts:126-163: Float32Array accumulator => round to f32 after every += and after /= */
void orc_centroid(const float *base, int64_t n, int dim, float *centroid) {
  for (int i = 0; i < dim; i++) centroid[i] = base[i];
  for (int64_t j = 1; j < n; j++) {
    const float *v = base + j * (int64_t)dim;
    for (int i = 0; i < dim; i++) centroid[i] = (float)((double)centroid[i] + (double)v[i]);
  }
  for (int i = 0; i < dim; i++) centroid[i] = (float)((double)centroid[i] / (double)n);
}

/* src/vectorOperations.ts:171-185 */
double orc_dot_f32(const float *a, const float *b, int dim) {
  double s = 0;
  for (int i = 0; i < dim; i++) s += (double)a[i] * (double)b[i];
  return s;
}

/* src/vectorSimilarity.ts:73-101 */
double orc_cosine_similarity(const float *a, const float *b, int dim) {
  double dp = 0, na = 0, nb = 0;
  for (int i = 0; i < dim; i++) {
    double av = a[i], bv = b[i];
    dp += av * bv; na += av * av; nb += bv * bv;
  }
  if (na == 0 || nb == 0) return 0;
  return dp / (sqrt(na) * sqrt(nb));
}

/* ------------------------------------------------------------------ quantizer */

/* src/constants.ts:38-47 */
static const double MINIMUM_MSE_GRID[8] = {0.798, 1.493, 2.051, 2.514, 2.916, 3.278, 3.611, 3.922};

/* src/optimizedScalarQuantizer.ts:373-407 */
static double compute_loss(const float *w, int dim, double a, double b, int points, double norm2, double lambda) {
  double step = (b - a) / (double)(points - 1);
  double step_inv = 1.0 / step;
  double xe = 0.0, e = 0.0;
  for (int i = 0; i < dim; i++) {
    double xi = w[i];
    double clamped = js_clamp(xi, a, b);
    double k = js_round((clamped - a) * step_inv);
    double xiq = a + step * k;
    xe += xi * (xi - xiq);
    e += (xi - xiq) * (xi - xiq);
  }
  return (1.0 - lambda) * xe * xe / norm2 + lambda * e;
}

/* src/optimizedScalarQuantizer.ts:280-353 */
static void optimize_intervals(double iv[2], const float *w, int dim, double norm2, int points, double lambda, int iters) {
  double initial_loss = compute_loss(w, dim, iv[0], iv[1], points, norm2, lambda);
  double scale = (1.0 - lambda) / norm2;
  if (!isfinite(scale)) return;
  for (int iter = 0; iter < iters; iter++) {
    double a = iv[0], b = iv[1];
    double step_inv = (double)(points - 1) / (b - a);
    double daa = 0, dab = 0, dbb = 0, dax = 0, dbx = 0;
    for (int i = 0; i < dim; i++) {
      double xi = w[i];
      double clamped = js_clamp(xi, a, b);
      double k = js_round((clamped - a) * step_inv);
      double s = k / (double)(points - 1);
      daa += (1.0 - s) * (1.0 - s);
      dab += (1.0 - s) * s;
      dbb += s * s;
      dax += xi * (1.0 - s);
      dbx += xi * s;
    }
    double m0 = scale * dax * dax + lambda * daa;
    double m1 = scale * dax * dbx + lambda * dab;
    double m2 = scale * dbx * dbx + lambda * dbb;
    double det = m0 * m2 - m1 * m1;
    if (fabs(det) < 1e-12) return;                                   /* isNearZero, constants.ts:74 */
    double a_opt = (m2 * dax - m1 * dbx) / det;
    double b_opt = (m0 * dbx - m1 * dax) / det;
    if (fabs(iv[0] - a_opt) < 1e-8 && fabs(iv[1] - b_opt) < 1e-8) return;   /* isNearEqual, constants.ts:76 */
    double new_loss = compute_loss(w, dim, a_opt, b_opt, points, norm2, lambda);
    if (new_loss > initial_loss) return;
    iv[0] = a_opt; iv[1] = b_opt; initial_loss = new_loss;
  }
}

/* src/optimizedScalarQuantizer.ts:108-227 */
void orc_scalar_quantize(const float *vec, int dim, int bits, const float *centroid, int sim,
                         double lambda, int iters, uint8_t *dest, double corr[4]) {
  float *w = (float *)malloc(sizeof(float) * (size_t)(dim > 0 ? dim : 1));
  /* :155-164 centroid dot on the UNcentred input */
  double centroid_dot = 0;
  if (sim != ORC_EUCLIDEAN)
    for (int i = 0; i < dim; i++) centroid_dot += (double)vec[i] * (double)centroid[i];
  /* :167-178 centre; min/max over the f64 differences, working vector stored as f32 */
  double mn = DBL_MAX, mx = -DBL_MAX;
  for (int i = 0; i < dim; i++) {
    double c = (double)vec[i] - (double)centroid[i];
    w[i] = (float)c;
    mn = js_min(mn, c); mx = js_max(mx, c);
  }
  /* :181-183, src/utils.ts:25-68 */
  double sum = 0;
  for (int i = 0; i < dim; i++) sum += (double)w[i];
  double mean = sum / (double)dim;
  double var = 0;
  for (int i = 0; i < dim; i++) { double d = (double)w[i] - mean; var += d * d; }
  double std = sqrt(var / (double)dim);
  double n2 = 0;
  for (int i = 0; i < dim; i++) n2 += (double)w[i] * (double)w[i];
  double norm2 = sqrt(n2);
  /* :245-265 */
  double g = MINIMUM_MSE_GRID[bits - 1];
  double iv[2];
  iv[0] = js_clamp(-g * std + mean, mn, mx);
  iv[1] = js_clamp(g * std + mean, mn, mx);
  optimize_intervals(iv, w, dim, norm2, 1 << bits, lambda, iters);
  /* :192-216 */
  double a = iv[0], b = iv[1];
  int points = 1 << bits, n_steps = points - 1;
  double step = n_steps > 0 ? (b - a) / (double)n_steps : 0;
  double step_inv = step > 0 ? 1 / step : 0;
  double qsum = 0;
  for (int i = 0; i < dim; i++) {
    double xi = w[i];
    double clamped = js_clamp(xi, a, b);
    if (bits == 1) {
      double threshold = (a + b) / 2;
      int qv = clamped >= threshold ? 1 : 0;
      dest[i] = (uint8_t)qv;
      qsum += qv;
    } else {
      double assignment = js_round((clamped - a) * step_inv);
      dest[i] = js_to_uint8(js_min(assignment, (double)n_steps));
      qsum += assignment;
    }
  }
  corr[0] = iv[0]; corr[1] = iv[1];
  corr[2] = (sim == ORC_EUCLIDEAN) ? norm2 : centroid_dot;          /* :219 */
  corr[3] = qsum;
  free(w);
}

/* src/optimizedScalarQuantizer.ts:420-446: MSB-first, last partial byte zero-padded in the low bits */
int orc_pack_binary(const uint8_t *bits, int dim, uint8_t *packed) {
  for (int i = 0; i < dim;) {
    int result = 0;
    for (int j = 7; j >= 0 && i < dim; j--) {
      if (bits[i] != 0 && bits[i] != 1) return -1;
      result |= (bits[i] & 1) << j;
      i++;
    }
    packed[(i - 1) / 8] = (uint8_t)result;
  }
  return 0;
}

/* src/binaryQuantizationFormat.ts:165-263 */
static void build_index_common(const float *base, int64_t n, int dim, int sim, int index_bits, double lambda, int iters,
                               uint8_t *codes, double *corr, float *centroid) {
  const float *proc = base;
  float *norm = NULL;
  if (sim == ORC_COSINE) {                                           /* :174-176 */
    norm = (float *)malloc(sizeof(float) * (size_t)n * (size_t)dim);
    for (int64_t i = 0; i < n; i++) orc_normalize(base + i * dim, dim, norm + i * dim);
    proc = norm;
  }
  orc_centroid(proc, n, dim, centroid);                              /* :214 */
  int pb = (dim + 7) / 8;
  uint8_t *tmp = (uint8_t *)malloc((size_t)(dim > 0 ? dim : 1));
  for (int64_t i = 0; i < n; i++) {                                  /* :221-249 */
    orc_scalar_quantize(proc + i * dim, dim, index_bits, centroid, sim, lambda, iters, tmp, corr + 4 * i);
    if (index_bits == 1) orc_pack_binary(tmp, dim, codes + i * pb);
    else memcpy(codes + i * (int64_t)dim, tmp, (size_t)dim);
  }
  free(tmp);
  free(norm);
}
void orc_build_index(const float *base, int64_t n, int dim, int sim, double lambda, int iters,
                     uint8_t *codes, double *corr, float *centroid) {
  build_index_common(base, n, dim, sim, 1, lambda, iters, codes, corr, centroid);
}
void orc_build_index_unpacked(const float *base, int64_t n, int dim, int sim, int index_bits, double lambda, int iters,
                              uint8_t *codes, double *corr, float *centroid) {
  build_index_common(base, n, dim, sim, index_bits, lambda, iters, codes, corr, centroid);
}

/* src/binaryQuantizationFormat.ts:337-339 then :279-293: COSINE normalises twice */
void orc_quantize_query(const float *query, int dim, const float *centroid, int sim, int qb,
                        double lambda, int iters, uint8_t *qquant, double qcorr[4]) {
  float *p = (float *)malloc(sizeof(float) * (size_t)(dim > 0 ? dim : 1));
  if (sim == ORC_COSINE) {
    float *t = (float *)malloc(sizeof(float) * (size_t)(dim > 0 ? dim : 1));
    orc_normalize(query, dim, t);
    orc_normalize(t, dim, p);
    free(t);
  } else {
    memcpy(p, query, sizeof(float) * (size_t)dim);
  }
  orc_scalar_quantize(p, dim, qb, centroid, sim, lambda, iters, qquant, qcorr);
  free(p);
}

/* ------------------------------------------------------------------ integer dot products */

/* src/utils/computeBatchFourBitDotProductDirectPacked.ts:10-53 (one row) */
int32_t orc_qcdist_unpacked_query(const uint8_t *q, const uint8_t *row, int dim) {
  int32_t dot = 0;
  int main_bytes = dim / 8;
  for (int j = 0; j < main_bytes; j++) {
    int pv = row[j];
    const uint8_t *qq = q + j * 8;
    dot += qq[0] * ((pv >> 7) & 1);
    dot += qq[1] * ((pv >> 6) & 1);
    dot += qq[2] * ((pv >> 5) & 1);
    dot += qq[3] * ((pv >> 4) & 1);
    dot += qq[4] * ((pv >> 3) & 1);
    dot += qq[5] * ((pv >> 2) & 1);
    dot += qq[6] * ((pv >> 1) & 1);
    dot += qq[7] * (pv & 1);
  }
  int rem = main_bytes * 8;
  if (rem < dim) {
    int last = row[main_bytes];
    for (int d = rem; d < dim; d++) dot += q[d] * ((last >> (7 - (d % 8))) & 1);
  }
  return dot;
}

/* src/utils/bitcount.ts:7-15 */
static uint32_t bitcount32(uint32_t n) {
  n = n - ((n >> 1) & 0x55555555u);
  n = (n & 0x33333333u) + ((n >> 2) & 0x33333333u);
  n = (n + (n >> 4)) & 0x0F0F0F0Fu;
  n = n + (n >> 8);
  n = n + (n >> 16);
  return n & 0x3F;
}
/* src/batchDotProduct.ts:22-49 (one row): AND then popcount, byte by byte */
int32_t orc_qcdist_packed_query(const uint8_t *qp, const uint8_t *row, int packed_bytes) {
  int32_t dot = 0;
  for (int i = 0; i < packed_bytes; i++) dot += (int32_t)bitcount32((uint32_t)(qp[i] & row[i]));
  return dot;
}
/* src/bitwiseDotProduct.ts:14-30 */
int32_t orc_dot_u8(const uint8_t *q, const uint8_t *d, int dim) {
  int32_t s = 0;
  for (int i = 0; i < dim; i++) s += (int32_t)q[i] * (int32_t)d[i];
  return s;
}

/* ------------------------------------------------------------------ score */

/* src/batchDotProduct.ts:478-541 (one_bit) and :554-617 (every qb != 1) */
double orc_score(int32_t qcdist, const double q[4], const double x[4], int dim, double cdp, int sim, int one_bit) {
  const double FBS = 1.0 / 15.0;                                     /* src/constants.ts:20 */
  double x1 = x[3], ax = x[0], lx = x[1] - ax;
  double ay = q[0], y1 = q[3];
  double ly = one_bit ? (q[1] - ay) : (q[1] - ay) * FBS;
  double score = ax * ay * (double)dim + ay * lx * x1 + ax * ly * y1 + lx * ly * (double)qcdist;
  if (sim == ORC_EUCLIDEAN) {
    double e = q[2] + x[2] - 2 * score;
    return js_max(1 / (1 + e), 0);
  }
  double t;
  if (one_bit) t = score + (q[2] + x[2] - cdp);                      /* `score += a + b - c`, :517-519/:524-526 */
  else t = score + q[2] + x[2] - cdp;                                /* :591-593 */
  if (sim == ORC_COSINE) return js_max((1 + t) / 2, 0);
  if (one_bit) return t < 0 ? 1 / (1 - t) : t + 1;                   /* :527-533 */
  return t < 0 ? 1 / (1 - t / FBS) : t / FBS + 1;                    /* :597-603 */
}

/* src/binaryQuantizedScorer.ts:315-400 */
void orc_score_all(const uint8_t *codes, const double *corr, int64_t n, int dim,
                   const uint8_t *qquant, const double qcorr[4], int qb, int sim, double cdp,
                   int32_t *qcdist, double *score64, float *score32) {
  int pb = (dim + 7) / 8;
  uint8_t *qp = NULL;
  if (qb == 1) { qp = (uint8_t *)calloc((size_t)(pb > 0 ? pb : 1), 1); orc_pack_binary(qquant, dim, qp); }   /* :333-335 */
  for (int64_t i = 0; i < n; i++) {
    const uint8_t *row = codes + i * pb;
    int32_t d = (qb == 1) ? orc_qcdist_packed_query(qp, row, pb) : orc_qcdist_unpacked_query(qquant, row, dim);
    double s = orc_score(d, qcorr, corr + 4 * i, dim, cdp, sim, qb == 1);
    if (qcdist) qcdist[i] = d;
    if (score64) score64[i] = s;
    if (score32) score32[i] = (float)s;                              /* binaryQuantizationFormat.ts:353,378 */
  }
  free(qp);
}

/* ------------------------------------------------------------------ multi-bit index (indexBits > 1) */

/* src/binaryQuantizedScorer.ts:108-160 (computeOneBitSimilarityScore) and :171-214 (computeFourBitSimilarityScore): the formulas
 * of the PER-ROW scorer, which is what answers for indexBits > 1 (the batch scorer throws on unpacked rows and
 * computeBatchQuantizedScores falls back, :403-419).  They differ from the batch forms above for 4-bit queries: MIP goes through
 * scaleMaxInnerProductScore (src/utils.ts:171-176) WITHOUT the division by FOUR_BIT_SCALE (:207-209). */
double orc_score_single_row(int32_t qcdist, const double q[4], const double x[4], int dim, double cdp, int sim, int one_bit) {
  const double FBS = 1.0 / 15.0;
  double x1 = x[3], ax = x[0], lx = x[1] - ax;
  double ay = q[0], y1 = q[3];
  double ly = one_bit ? (q[1] - ay) : (q[1] - ay) * FBS;                       /* :122 / :186 */
  double score = ax * ay * (double)dim + ay * lx * x1 + ax * ly * y1 + lx * ly * (double)qcdist;   /* :126-129 / :190 */
  if (sim == ORC_EUCLIDEAN) {
    double e = q[2] + x[2] - 2 * score;                                        /* :134-137 / :194-197 */
    return js_max(1 / (1 + e), 0);
  }
  double t;
  if (one_bit) t = score + (q[2] + x[2] - cdp);                                /* `score += ...`, :141-143 / :148-150 */
  else t = score + q[2] + x[2] - cdp;                                          /* :201-203 */
  if (sim == ORC_COSINE) return js_max((1 + t) / 2, 0);
  return t < 0 ? 1 / (1 - t) : t + 1;                                          /* scaleMaxInnerProductScore */
}

/* computeBatchQuantizedScores (src/binaryQuantizedScorer.ts:315-420) as it behaves for indexBits > 1, rows = unpacked bytes
 * (src/binaryQuantizationFormat.ts:241-245):
 *  - createDirectPackedBuffer allocates ceil(dim/8) bytes per row and set()s dim bytes per row (src/batchDotProduct.ts:425-433):
 *    a RangeError for every dim > 1 -> catch -> per-row computeQuantizedScore (:403-419)
 *  - per row: queryBits 1 -> one-bit formula with centroidDP = centroid.centroid (:238-244); queryBits 4 -> four-bit formula with
 *    centroidDP = 0 because searchNearestNeighbors passes no original query (:290); anything else throws (:96) -> returns -1
 *  - dim == 1 is the one width where nothing throws: the batch kernels read the unpacked byte as a packed one (bit 7) and the
 *    batch formulas apply
 * qcDist = computeQuantizedDotProduct over the unpacked bytes (src/bitwiseDotProduct.ts:14-30). */
int orc_score_all_multibit(const uint8_t *codes, const double *corr, int64_t n, int dim,
                           const uint8_t *qquant, const double qcorr[4], int qb, int sim, double cdp_centroid,
                           int32_t *qcdist, double *score64, float *score32) {
  if (dim == 1) {
    for (int64_t i = 0; i < n; i++) {
      int32_t d = (qb == 1) ? (int32_t)bitcount32((uint32_t)(((qquant[0] & 1) << 7) & codes[i])) : (int32_t)qquant[0] * ((codes[i] >> 7) & 1);
      double s = orc_score(d, qcorr, corr + 4 * i, dim, cdp_centroid, sim, qb == 1);
      if (qcdist) qcdist[i] = d;
      if (score64) score64[i] = s;
      if (score32) score32[i] = (float)s;
    }
    return 0;
  }
  if (qb != 1 && qb != 4) return -1;                                           /* :95-97 */
  for (int64_t i = 0; i < n; i++) {
    int32_t d = orc_dot_u8(qquant, codes + i * dim, dim);
    double s = orc_score_single_row(d, qcorr, corr + 4 * i, dim, qb == 1 ? cdp_centroid : 0.0, sim, qb == 1);
    if (qcdist) qcdist[i] = d;
    if (score64) score64[i] = s;
    if (score32) score32[i] = (float)s;
  }
  return 0;
}

/* NOT a reference behaviour: the reference throws for queryBits other than 1 and 4 on a multi-bit index (:95-97).  libbbq documents
 * that it scores such queries with the per-row scorer's 4-bit form (centroidDP = 0) over computeQuantizedDotProduct - the integer
 * part is the reference's definition for any widths (src/bitwiseDotProduct.ts:14-30), the float part is "parity unpinned".  This is
 * that definition, used to check the library and as the CPU baseline of BASELINE config 5 (queryBits 8 / indexBits 2). */
void orc_score_all_multibit_ext(const uint8_t *codes, const double *corr, int64_t n, int dim,
                                const uint8_t *qquant, const double qcorr[4], int qb, int sim, double cdp_centroid,
                                int32_t *qcdist, double *score64, float *score32) {
  for (int64_t i = 0; i < n; i++) {
    int32_t d = orc_dot_u8(qquant, codes + i * dim, dim);
    double s = orc_score_single_row(d, qcorr, corr + 4 * i, dim, qb == 1 ? cdp_centroid : 0.0, sim, qb == 1);
    if (qcdist) qcdist[i] = d;
    if (score64) score64[i] = s;
    if (score32) score32[i] = (float)s;
  }
}

/* ------------------------------------------------------------------ MinHeap (src/minHeap.ts:9-130) */

typedef struct { double score; int32_t index; } heap_item;
typedef struct { heap_item *a; int64_t len; } min_heap;

static void heap_bubble_up(min_heap *h) {                            /* :70-81 */
  int64_t index = h->len - 1;
  while (index > 0) {
    int64_t parent = (index - 1) / 2;
    double cmp = h->a[index].score - h->a[parent].score;
    if (cmp >= 0) break;                                             /* NaN: not >= 0 -> swaps, as in JS */
    heap_item t = h->a[index]; h->a[index] = h->a[parent]; h->a[parent] = t;
    index = parent;
  }
}
static void heap_bubble_down(min_heap *h) {                          /* :86-116 */
  int64_t index = 0;
  for (;;) {
    int64_t smallest = index, l = 2 * index + 1, r = 2 * index + 2;
    if (l < h->len && (h->a[l].score - h->a[smallest].score) < 0) smallest = l;
    if (r < h->len && (h->a[r].score - h->a[smallest].score) < 0) smallest = r;
    if (smallest == index) break;
    heap_item t = h->a[index]; h->a[index] = h->a[smallest]; h->a[smallest] = t;
    index = smallest;
  }
}
static void heap_push(min_heap *h, heap_item it) { h->a[h->len++] = it; heap_bubble_up(h); }     /* :45-48 */
static heap_item heap_pop(min_heap *h) {                             /* :53-65 */
  heap_item mn = h->a[0];
  heap_item last = h->a[--h->len];
  if (h->len > 0) { h->a[0] = last; heap_bubble_down(h); }
  return mn;
}

/* src/binaryQuantizationFormat.ts:383-411 */
int64_t orc_heap_topk(const float *scores, int64_t n, int64_t k, int32_t *out_idx, float *out_score) {
  int64_t k2 = k < n ? k : n;
  if (k2 <= 0) return 0;
  min_heap h; h.a = (heap_item *)malloc(sizeof(heap_item) * (size_t)(k2 + 1)); h.len = 0;
  for (int64_t i = 0; i < n; i++) {
    double cur = (double)scores[i];
    if (h.len < k2) { heap_item it = {cur, (int32_t)i}; heap_push(&h, it); }
    else if (cur > h.a[0].score) { heap_pop(&h); heap_item it = {cur, (int32_t)i}; heap_push(&h, it); }
  }
  int64_t cnt = h.len;
  for (int64_t j = cnt - 1; j >= 0; j--) {                           /* pop ascending, reversed */
    heap_item it = heap_pop(&h);
    out_idx[j] = it.index; out_score[j] = (float)it.score;
  }
  free(h.a);
  return cnt;
}

/* src/binaryQuantizationFormat.ts:308-412 */
int64_t orc_search(const float *query, int query_dim, const uint8_t *codes, const double *corr, const float *centroid,
                   int64_t n, int dim, int sim, int qb, double lambda, int iters, int64_t k,
                   int32_t *out_idx, float *out_score) {
  if (!query) return -1;
  if (!codes) return -2;
  if (k < 0) return -3;
  if (query_dim != dim) return -4;
  if (k == 0) return 0;
  uint8_t *qq = (uint8_t *)malloc((size_t)(dim > 0 ? dim : 1));
  double qcorr[4];
  orc_quantize_query(query, dim, centroid, sim, qb, lambda, iters, qq, qcorr);
  double cdp = orc_dot_f32(centroid, centroid, dim);                 /* getCentroidDP(undefined), :113-121 */
  float *s32 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  orc_score_all(codes, corr, n, dim, qq, qcorr, qb, sim, cdp, NULL, NULL, s32);
  int64_t cnt = orc_heap_topk(s32, n, k, out_idx, out_score);
  free(s32); free(qq);
  return cnt;
}

/* searchNearestNeighbors on an indexBits > 1 index; -5 where the reference throws '不支持的查询位数' */
int64_t orc_search_multibit(const float *query, int query_dim, const uint8_t *codes, const double *corr, const float *centroid,
                            int64_t n, int dim, int sim, int qb, double lambda, int iters, int64_t k,
                            int32_t *out_idx, float *out_score) {
  if (!query) return -1;
  if (!codes) return -2;
  if (k < 0) return -3;
  if (query_dim != dim) return -4;
  if (k == 0) return 0;
  uint8_t *qq = (uint8_t *)malloc((size_t)(dim > 0 ? dim : 1));
  double qcorr[4];
  orc_quantize_query(query, dim, centroid, sim, qb, lambda, iters, qq, qcorr);
  double cdp = orc_dot_f32(centroid, centroid, dim);
  float *s32 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  int64_t cnt = -5;
  if (orc_score_all_multibit(codes, corr, n, dim, qq, qcorr, qb, sim, cdp, NULL, NULL, s32) == 0)
    cnt = orc_heap_topk(s32, n, k, out_idx, out_score);
  free(s32); free(qq);
  return cnt;
}

/* src/topKSelector.ts:29-79 */
int64_t orc_oversampled_topk(const float *query, const float *base, const uint8_t *codes, const double *corr,
                             const float *centroid, int64_t n, int dim, int sim, int qb, double lambda, int iters,
                             int64_t k, int factor, int32_t *out_idx) {
  int64_t ok = k * factor;
  int64_t cap = ok < n ? ok : n;
  int32_t *cidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cap + 1));
  float *csc = (float *)malloc(sizeof(float) * (size_t)(cap + 1));
  int64_t cnt = orc_search(query, dim, codes, corr, centroid, n, dim, sim, qb, lambda, iters, ok, cidx, csc);
  if (cnt < 0) { free(cidx); free(csc); return cnt; }
  min_heap h; h.a = (heap_item *)malloc(sizeof(heap_item) * (size_t)(k + 1)); h.len = 0;
  for (int64_t i = 0; i < cnt; i++) {
    double ts = orc_cosine_similarity(query, base + (int64_t)cidx[i] * dim, dim);
    heap_item it = {ts, cidx[i]};
    if (h.len < k) heap_push(&h, it);
    else if (h.len > 0 && ts > h.a[0].score) { heap_pop(&h); heap_push(&h, it); }
  }
  int64_t m = h.len;
  heap_item *asc = (heap_item *)malloc(sizeof(heap_item) * (size_t)(m + 1));
  for (int64_t j = 0; j < m; j++) asc[j] = heap_pop(&h);
  /* topK.sort((a,b) => b.trueScore - a.trueScore): V8 >= 7.0 sort is stable; insertion sort keeps that */
  for (int64_t i = 1; i < m; i++) {
    heap_item key = asc[i]; int64_t j = i - 1;
    while (j >= 0 && (key.score - asc[j].score) > 0) { asc[j + 1] = asc[j]; j--; }
    asc[j + 1] = key;
  }
  for (int64_t j = 0; j < m; j++) out_idx[j] = asc[j].index;
  free(asc); free(h.a); free(cidx); free(csc);
  return m;
}

/* src/vectorSimilarity.ts:14-126: EUCLIDEAN 1/(1+sqrt(sum (a-b)^2)), COSINE, MAXIMUM_INNER_PRODUCT; all f64, in index order */
double orc_true_similarity(const float *a, const float *b, int dim, int sim) {
  if (sim == ORC_COSINE) return orc_cosine_similarity(a, b, dim);
  if (sim == ORC_EUCLIDEAN) {
    double sum = 0;
    for (int i = 0; i < dim; i++) { double diff = (double)a[i] - (double)b[i]; sum += diff * diff; }
    return 1.0 / (1.0 + sqrt(sum));
  }
  double dp = 0;
  for (int i = 0; i < dim; i++) dp += (double)a[i] * (double)b[i];
  return dp;
}

/* src/topKSelector.ts:40-78 from the candidates' true scores (candidate order = the oversampled search's result order):
 * min-heap of k on trueScore, pop all, stable sort descending.  out_pos = positions into the candidate list. */
int64_t orc_rerank_select_heap(const double *true_scores, int64_t cnt, int64_t k, int32_t *out_pos) {
  min_heap h; h.a = (heap_item *)malloc(sizeof(heap_item) * (size_t)(k + 1)); h.len = 0;
  for (int64_t i = 0; i < cnt; i++) {
    heap_item it = {true_scores[i], (int32_t)i};
    if (h.len < k) heap_push(&h, it);
    else if (h.len > 0 && it.score > h.a[0].score) { heap_pop(&h); heap_push(&h, it); }
  }
  int64_t m = h.len;
  heap_item *asc = (heap_item *)malloc(sizeof(heap_item) * (size_t)(m + 1));
  for (int64_t j = 0; j < m; j++) asc[j] = heap_pop(&h);
  for (int64_t i = 1; i < m; i++) {   /* stable, like V8's sort with a consistent comparator */
    heap_item key = asc[i]; int64_t j = i - 1;
    while (j >= 0 && (key.score - asc[j].score) > 0) { asc[j + 1] = asc[j]; j--; }
    asc[j + 1] = key;
  }
  for (int64_t j = 0; j < m; j++) out_pos[j] = asc[j].index;
  free(asc); free(h.a);
  return m;
}

/* src/topKSelector.ts:102-115: stable sort of all candidates by trueScore descending, first k */
int64_t orc_rerank_select_sort(const double *true_scores, int64_t cnt, int64_t k, int32_t *out_pos) {
  heap_item *v = (heap_item *)malloc(sizeof(heap_item) * (size_t)(cnt + 1));
  for (int64_t i = 0; i < cnt; i++) { v[i].score = true_scores[i]; v[i].index = (int32_t)i; }
  for (int64_t i = 1; i < cnt; i++) {
    heap_item key = v[i]; int64_t j = i - 1;
    while (j >= 0 && (key.score - v[j].score) > 0) { v[j + 1] = v[j]; j--; }
    v[j + 1] = key;
  }
  int64_t m = cnt < k ? cnt : k;
  if (m < 0) m = 0;
  for (int64_t j = 0; j < m; j++) out_pos[j] = v[j].index;
  free(v);
  return m;
}

/* ------------------------------------------------------------------ synthetic inputs */

void orc_mulberry32_fill(uint32_t seed, float *out, int64_t count) {
  uint32_t a = seed;
  for (int64_t i = 0; i < count; i++) {
    a += 0x6D2B79F5u;
    uint32_t t = (a ^ (a >> 15)) * (1u | a);
    t = (t + ((t ^ (t >> 7)) * (61u | t))) ^ t;
    double u = (double)(t ^ (t >> 14)) / 4294967296.0;
    out[i] = (float)(2 * u - 1);
  }
}
