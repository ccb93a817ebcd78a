// bbq_quantizer.cpp - host-side (multithreaded C++) index build and query quantization for the drop-in API.
//
// Follows OptimizedScalarQuantizer.scalarQuantize (reference src/optimizedScalarQuantizer.ts:108-227) with
// getInitialInterval :245-265, optimizeIntervals :280-353, computeLoss :373-407, packAsBinary :420-446, and
// BinaryQuantizationFormat.quantizeVectors / quantizeQueryVector (src/binaryQuantizationFormat.ts:165-299).
// JavaScript number model: every operation is IEEE binary64 in source order (this file is compiled with
// -ffp-contract=off), values are rounded to f32 exactly where the reference stores into a Float32Array.
// The scan kernel consumes the integer codes, so this must be exact, not approximately right.
#include <float.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include "bbq_internal.h"

namespace {

// Math.min / Math.max / Math.round as V8 implements them
inline double jmin(double a, double b) {
  if (a != a || b != b) return NAN;
  if (a == 0.0 && b == 0.0) return (signbit(a) || signbit(b)) ? -0.0 : 0.0;
  return a < b ? a : b;
}
inline double jmax(double a, double b) {
  if (a != a || b != b) return NAN;
  if (a == 0.0 && b == 0.0) return (signbit(a) && signbit(b)) ? -0.0 : 0.0;
  return a > b ? a : b;
}
inline double jclamp(double x, double lo, double hi) { return jmin(jmax(x, lo), hi); }  // src/utils.ts:79-81
inline double jround(double x) {
  if (x != x || isinf(x)) return x;
  double r = floor(x);
  if (x - r >= 0.5) r += 1.0;
  return r;
}
inline uint8_t to_uint8(double x) {  // element store into a Uint8Array
  if (x != x || isinf(x)) return 0;
  double m = fmod(trunc(x), 256.0);
  if (m < 0) m += 256.0;
  return (uint8_t)m;
}

const double kGrid[8] = {0.798, 1.493, 2.051, 2.514, 2.916, 3.278, 3.611, 3.922};  // src/constants.ts:38-47

struct Quantizer {
  int sim;
  double lambda;
  int iters;
  std::vector<float> w;  // centred working vector (Float32Array in the reference)

  double loss(int dim, double a, double b, int points, double norm2) const {  // :373-407
    const double step = (b - a) / (double)(points - 1);
    const double step_inv = 1.0 / step;
    double xe = 0.0, e = 0.0;
    for (int i = 0; i < dim; ++i) {
      const double xi = w[i];
      const double k = jround((jclamp(xi, a, b) - a) * step_inv);
      const double xiq = a + step * k;
      xe += xi * (xi - xiq);
      e += (xi - xiq) * (xi - xiq);
    }
    return (1.0 - lambda) * xe * xe / norm2 + lambda * e;
  }

  void optimize(double iv[2], int dim, double norm2, int points) const {  // :280-353
    double best = loss(dim, iv[0], iv[1], points, norm2);
    const double scale = (1.0 - lambda) / norm2;
    if (!isfinite(scale)) return;
    for (int it = 0; it < iters; ++it) {
      const double a = iv[0], b = iv[1];
      const double step_inv = (double)(points - 1) / (b - a);
      double daa = 0, dab = 0, dbb = 0, dax = 0, dbx = 0;
      for (int i = 0; i < dim; ++i) {
        const double xi = w[i];
        const double k = jround((jclamp(xi, a, b) - a) * step_inv);
        const double s = k / (double)(points - 1);
        daa += (1.0 - s) * (1.0 - s);
        dab += (1.0 - s) * s;
        dbb += s * s;
        dax += xi * (1.0 - s);
        dbx += xi * s;
      }
      const double m0 = scale * dax * dax + lambda * daa;
      const double m1 = scale * dax * dbx + lambda * dab;
      const double m2 = scale * dbx * dbx + lambda * dbb;
      const double det = m0 * m2 - m1 * m1;
      if (fabs(det) < 1e-12) return;
      const double a_opt = (m2 * dax - m1 * dbx) / det;
      const double b_opt = (m0 * dbx - m1 * dax) / det;
      if (fabs(iv[0] - a_opt) < 1e-8 && fabs(iv[1] - b_opt) < 1e-8) return;
      const double nl = loss(dim, a_opt, b_opt, points, norm2);
      if (nl > best) return;
      iv[0] = a_opt;
      iv[1] = b_opt;
      best = nl;
    }
  }

  // scalarQuantize, :108-227.  dest: one value per dimension.
  void quantize(const float *vec, int dim, int bits, const float *centroid, uint8_t *dest, double corr[4]) {
    if ((int)w.size() < dim) w.resize((size_t)dim);
    double cdot = 0;
    if (sim != BBQ_EUCLIDEAN)
      for (int i = 0; i < dim; ++i) cdot += (double)vec[i] * (double)centroid[i];
    double mn = DBL_MAX, mx = -DBL_MAX;
    for (int i = 0; i < dim; ++i) {
      const double c = (double)vec[i] - (double)centroid[i];
      w[i] = (float)c;
      mn = jmin(mn, c);
      mx = jmax(mx, c);
    }
    double sum = 0;
    for (int i = 0; i < dim; ++i) sum += (double)w[i];
    const double mean = sum / (double)dim;
    double var = 0, n2 = 0;
    for (int i = 0; i < dim; ++i) {
      const double d = (double)w[i] - mean;
      var += d * d;
    }
    for (int i = 0; i < dim; ++i) n2 += (double)w[i] * (double)w[i];
    const double sd = sqrt(var / (double)dim);
    const double norm2 = sqrt(n2);
    const double g = kGrid[bits - 1];
    double iv[2] = {jclamp(-g * sd + mean, mn, mx), jclamp(g * sd + mean, mn, mx)};
    const int points = 1 << bits, n_steps = points - 1;
    optimize(iv, dim, norm2, points);
    const double a = iv[0], b = iv[1];
    const double step = n_steps > 0 ? (b - a) / (double)n_steps : 0;
    const double step_inv = step > 0 ? 1 / step : 0;
    double qsum = 0;
    if (bits == 1) {
      const double thr = (a + b) / 2;
      for (int i = 0; i < dim; ++i) {
        const int qv = jclamp((double)w[i], a, b) >= thr ? 1 : 0;
        dest[i] = (uint8_t)qv;
        qsum += qv;
      }
    } else {
      for (int i = 0; i < dim; ++i) {
        const double as = jround((jclamp((double)w[i], a, b) - a) * step_inv);
        dest[i] = to_uint8(jmin(as, (double)n_steps));
        qsum += as;
      }
    }
    corr[0] = iv[0];
    corr[1] = iv[1];
    corr[2] = sim == BBQ_EUCLIDEAN ? norm2 : cdot;
    corr[3] = qsum;
  }
};

void normalize(const float *v, int dim, float *out) {  // src/vectorOperations.ts:11-34
  double n = 0;
  for (int i = 0; i < dim; ++i) n += (double)v[i] * (double)v[i];
  n = sqrt(n);
  if (n == 0) {
    memset(out, 0, sizeof(float) * (size_t)dim);
    return;
  }
  for (int i = 0; i < dim; ++i) out[i] = (float)((double)v[i] / n);
}

void pack_binary(const uint8_t *bits, int dim, uint8_t *packed) {  // :420-446, MSB first
  const int pb = (dim + 7) / 8;
  memset(packed, 0, (size_t)pb);
  for (int d = 0; d < dim; ++d)
    if (bits[d] & 1) packed[d >> 3] |= (uint8_t)(0x80u >> (d & 7));
}

template <class F>
void parallel_rows(int64_t n, int n_threads, F f) {
  int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n / 64 + 1));
  if (T == 1) {
    f(0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(f, n * t / T, n * (t + 1) / T, t);
  for (auto &x : th) x.join();
}

int check_common(const void *a, const void *b, int32_t dim, int32_t sim, int32_t bits, double lambda, int32_t iters) {
  if (!a || !b) return bbq::fail(BBQ_ERR_INVALID_ARG, "输入向量不能为空");
  if (dim <= 0) return bbq::fail(BBQ_ERR_INVALID_ARG, "dimension must be positive");
  if (sim < 0 || sim > 2) return bbq::fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", sim);
  if (bits < 1 || bits > 8) return bbq::fail(BBQ_ERR_INVALID_ARG, "位数必须在1-8之间");
  if (iters < 0 || lambda != lambda) return bbq::fail(BBQ_ERR_INVALID_ARG, "bad lambda/iters");
  return BBQ_OK;
}

}  // namespace

extern "C" {

double bbq_centroid_dp(const float *c, int32_t dim) {  // src/vectorOperations.ts:171-185
  double s = 0;
  for (int i = 0; i < dim; ++i) s += (double)c[i] * (double)c[i];
  return s;
}

int bbq_quantize_vectors(const float *vectors, int64_t n, int32_t dim, int32_t sim, int32_t index_bits, double lambda,
                         int32_t iters, int32_t n_threads, uint8_t *codes, double *corr, float *centroid, int64_t *bad_row,
                         int32_t *bad_col) {
  bbq::clear_error();
  if (n == 0) return bbq::fail(BBQ_ERR_EMPTY, "向量集合不能为空");  // src/binaryQuantizationFormat.ts:169-171
  if (n < 0) return bbq::fail(BBQ_ERR_INVALID_ARG, "n < 0");
  int rc = check_common(vectors, codes, dim, sim, index_bits, lambda, iters);
  if (rc != BBQ_OK) return rc;
  if (!corr || !centroid) return bbq::fail(BBQ_ERR_INVALID_ARG, "null output");

  // :174-176 normalise (COSINE) into a working copy; :196-211 NaN / Infinity validation on the processed vectors,
  // first offender in row-major order
  std::vector<float> norm;
  const float *proc = vectors;
  if (sim == BBQ_COSINE) {
    norm.resize((size_t)n * (size_t)dim);
    parallel_rows(n, n_threads, [&](int64_t lo, int64_t hi, int) {
      for (int64_t i = lo; i < hi; ++i) normalize(vectors + i * dim, dim, norm.data() + i * dim);
    });
    proc = norm.data();
  }
  std::atomic<int64_t> first_bad(INT64_MAX);
  parallel_rows(n, n_threads, [&](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi && i * dim < first_bad.load(std::memory_order_relaxed); ++i)
      for (int j = 0; j < dim; ++j) {
        const float v = proc[i * dim + j];
        if (v != v || isinf(v)) {
          int64_t pos = i * dim + j, cur = first_bad.load();
          while (pos < cur && !first_bad.compare_exchange_weak(cur, pos)) {}
          return;
        }
      }
  });
  if (first_bad.load() != INT64_MAX) {
    const int64_t pos = first_bad.load(), r = pos / dim;
    const int c = (int)(pos % dim);
    if (bad_row) *bad_row = r;
    if (bad_col) *bad_col = c;
    const float v = proc[pos];
    if (v != v) return bbq::fail(BBQ_ERR_NAN_INPUT, "向量 %lld 位置 %d 包含NaN值", (long long)r, c);
    return bbq::fail(BBQ_ERR_INF_INPUT, "向量 %lld 位置 %d 包含Infinity值", (long long)r, c);
  }

  // :214 computeCentroid (src/vectorOperations.ts:126-163): Float32Array accumulator, rounded after every += and
  // after the final /=.  Sequential over rows per dimension, so dimensions can be split across threads.
  parallel_rows(dim, n_threads, [&](int64_t lo, int64_t hi, int) {
    for (int64_t d = lo; d < hi; ++d) {
      float c = proc[d];
      for (int64_t j = 1; j < n; ++j) c = (float)((double)c + (double)proc[j * dim + d]);
      centroid[d] = (float)((double)c / (double)n);
    }
  });

  const int pb = (dim + 7) / 8;
  parallel_rows(n, n_threads, [&](int64_t lo, int64_t hi, int) {
    Quantizer qz{sim, lambda, iters, {}};
    std::vector<uint8_t> tmp((size_t)dim);
    for (int64_t i = lo; i < hi; ++i) {
      qz.quantize(proc + i * dim, dim, index_bits, centroid, tmp.data(), corr + 4 * i);
      if (index_bits == 1) pack_binary(tmp.data(), dim, codes + i * pb);  // :235-240
      else memcpy(codes + i * (int64_t)dim, tmp.data(), (size_t)dim);    // :241-245
    }
  });
  return BBQ_OK;
}

static int quantize_query_impl(const float *query, int32_t dim, const float *centroid, int32_t sim, int32_t qb, double lambda,
                               int32_t iters, uint8_t *qquant, double *qcorr, int normalisations) {
  bbq::clear_error();
  int rc = check_common(query, centroid, dim, sim, qb, lambda, iters);
  if (rc != BBQ_OK) return rc;
  if (!qquant || !qcorr) return bbq::fail(BBQ_ERR_INVALID_ARG, "null output");
  std::vector<float> p(query, query + dim), t((size_t)dim);
  if (sim == BBQ_COSINE)
    for (int r = 0; r < normalisations; ++r) {
      normalize(p.data(), dim, t.data());
      p.swap(t);
    }
  for (int i = 0; i < dim; ++i) {  // scalarQuantize's own validation, :138-148
    if (p[i] != p[i]) return bbq::fail(BBQ_ERR_NAN_INPUT, "向量位置 %d 包含NaN值", i);
    if (isinf(p[i])) return bbq::fail(BBQ_ERR_INF_INPUT, "向量位置 %d 包含Infinity值", i);
  }
  Quantizer qz{sim, lambda, iters, {}};
  qz.quantize(p.data(), dim, qb, centroid, qquant, qcorr);
  return BBQ_OK;
}

int bbq_quantize_query(const float *query, int32_t dim, const float *centroid, int32_t sim, int32_t qb, double lambda,
                       int32_t iters, uint8_t *qquant, double *qcorr) {
  return quantize_query_impl(query, dim, centroid, sim, qb, lambda, iters, qquant, qcorr, 2);
}

int bbq_quantize_query_vector(const float *query, int32_t dim, const float *centroid, int32_t sim, int32_t qb, double lambda,
                              int32_t iters, uint8_t *qquant, double *qcorr) {
  return quantize_query_impl(query, dim, centroid, sim, qb, lambda, iters, qquant, qcorr, 1);
}

int bbq_quantize_queries(const float *queries, int32_t n, int32_t dim, const float *centroid, int32_t sim, int32_t qb, double lambda,
                         int32_t iters, int32_t n_threads, uint8_t *qquant, double *qcorr, int32_t *bad_query) {
  bbq::clear_error();
  if (bad_query) *bad_query = -1;
  if (n < 0) return bbq::fail(BBQ_ERR_INVALID_ARG, "n < 0");
  if (n == 0) return BBQ_OK;
  int rc = check_common(queries, centroid, dim, sim, qb, lambda, iters);
  if (rc != BBQ_OK) return rc;
  if (!qquant || !qcorr) return bbq::fail(BBQ_ERR_INVALID_ARG, "null output");
  int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  T = std::max(1, std::min(T, n / 4 + 1));  // a 768-d query takes ~50 us: below 4 queries per thread spawning costs more
  std::vector<int> first_bad((size_t)T, -1);
  auto work = [&](int lo, int hi, int t) {
    for (int i = lo; i < hi; ++i)
      if (quantize_query_impl(queries + (size_t)i * dim, dim, centroid, sim, qb, lambda, iters, qquant + (size_t)i * dim, qcorr + (size_t)i * 4, 2) !=
          BBQ_OK) {
        first_bad[(size_t)t] = i;
        return;
      }
  };
  if (T == 1) {
    work(0, n, 0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, (int)((int64_t)n * t / T), (int)((int64_t)n * (t + 1) / T), t);
    for (auto &x : th) x.join();
  }
  int bad = -1;
  for (int t = 0; t < T; ++t)
    if (first_bad[(size_t)t] >= 0 && (bad < 0 || first_bad[(size_t)t] < bad)) bad = first_bad[(size_t)t];
  if (bad < 0) return BBQ_OK;
  if (bad_query) *bad_query = bad;
  // the message lives in the worker's thread-local slot: produce it again on this thread
  return quantize_query_impl(queries + (size_t)bad * dim, dim, centroid, sim, qb, lambda, iters, qquant + (size_t)bad * dim, qcorr + (size_t)bad * 4, 2);
}

}  // extern "C"
