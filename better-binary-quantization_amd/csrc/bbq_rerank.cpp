// bbq_rerank.cpp - oversample + exact rerank: fp32 vectors resident in HBM, true scores on the device (bbq_rerank_kernels.hip),
// the reference's two selectors on the host
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <string>
#include "bbq_host.h"

using namespace bbq;

struct bbq_vectors {
  int device = 0;
  DeviceCtx *ctx = nullptr;
  float *d = nullptr;
  int64_t n = 0;
  int32_t dim = 0;
  // grow-only staging for bbq_rerank_scores
  float *d_q = nullptr;
  int64_t q_cap = 0;
  int64_t *d_off = nullptr;
  int64_t off_cap = 0;
  int32_t *d_rows = nullptr;
  double *d_out = nullptr;
  int64_t cand_cap = 0;
};

namespace {

template <class T>
int grow(T **p, int64_t *cap, int64_t need) {
  if (need <= *cap) return BBQ_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const int64_t c = need + need / 2 + 64;
  HIPCHK(hipMalloc((void **)p, (size_t)c * sizeof(T)));
  *cap = c;
  return BBQ_OK;
}

struct Ranked { double score; int32_t pos; };

// Array.prototype.sort((a, b) => b.trueScore - a.trueScore), src/topKSelector.ts:75,112: stable; an element of the right
// run overtakes one of the left run only when the comparator says so (> 0), whatever it says for NaN
void sort_desc_stable(std::vector<Ranked> &v) {
  const size_t n = v.size();
  std::vector<Ranked> tmp(n);
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      const size_t mid = std::min(lo + w, n), hi = std::min(lo + 2 * w, n);
      size_t i = lo, j = mid, o = lo;
      while (i < mid && j < hi) tmp[o++] = (v[j].score - v[i].score) > 0 ? v[j++] : v[i++];
      while (i < mid) tmp[o++] = v[i++];
      while (j < hi) tmp[o++] = v[j++];
    }
    v.swap(tmp);
  }
}

}  // namespace

extern "C" {

int bbq_vectors_create(const float *vectors, int64_t n, int32_t dim, int32_t device, bbq_vectors **out) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_vectors_create: out is null");
  *out = nullptr;
  if (n < 0 || dim <= 0 || (n > 0 && !vectors)) return fail(BBQ_ERR_INVALID_ARG, "bbq_vectors_create: bad arguments");
  if (n > 0x7fffffffLL) return fail(BBQ_ERR_INVALID_ARG, "bbq_vectors_create: more than 2^31-1 rows");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  if (device < 0 || device >= ndev) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));  // before get_ctx: a new context creates its streams on the current device
  DeviceCtx *ctx = nullptr;
  int rc = get_ctx(device, &ctx);
  if (rc != BBQ_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->mu);
  bbq_vectors *v = new bbq_vectors();
  v->device = device;
  v->ctx = ctx;
  v->n = n;
  v->dim = dim;
  if (n > 0) {
    hipError_t e = hipMalloc((void **)&v->d, (size_t)n * dim * sizeof(float));
    if (e != hipSuccess) {
      delete v;
      return fail(BBQ_ERR_OOM, "bbq_vectors_create: %lld x %d fp32: %s", (long long)n, dim, hipGetErrorString(e));
    }
    const int64_t total = n * dim, piece = 64LL << 20;  // 256 MB pieces keep the runtime's pinned staging bounded
    for (int64_t o = 0; o < total; o += piece) {
      e = hipMemcpy(v->d + o, vectors + o, (size_t)std::min(piece, total - o) * sizeof(float), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)hipFree(v->d);
        delete v;
        return fail(BBQ_ERR_HIP, "bbq_vectors_create: copy: %s", hipGetErrorString(e));
      }
    }
  }
  *out = v;
  return BBQ_OK;
}

void bbq_vectors_destroy(bbq_vectors *v) {
  if (!v) return;
  std::lock_guard<std::mutex> lk(v->ctx->mu);
  (void)hipSetDevice(v->device);
  if (v->d) (void)hipFree(v->d);
  if (v->d_q) (void)hipFree(v->d_q);
  if (v->d_off) (void)hipFree(v->d_off);
  if (v->d_rows) (void)hipFree(v->d_rows);
  if (v->d_out) (void)hipFree(v->d_out);
  delete v;
}

int64_t bbq_vectors_size(const bbq_vectors *v) { return v ? v->n : 0; }
int32_t bbq_vectors_dimension(const bbq_vectors *v) { return v ? v->dim : 0; }

int bbq_rerank_scores(bbq_vectors *v, int32_t n_queries, const float *queries, const int64_t *offsets, const int32_t *rows,
                      int32_t true_sim, double *out_true) {
  clear_error();
  if (!v) return fail(BBQ_ERR_INVALID_ARG, "bbq_rerank_scores: vectors handle is null");
  if (n_queries < 0 || n_queries > 65535) return fail(BBQ_ERR_INVALID_ARG, "bbq_rerank_scores: n_queries out of range");
  if (true_sim < 0 || true_sim > 2) return fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", true_sim);
  if (n_queries == 0) return BBQ_OK;
  if (!queries || !offsets) return fail(BBQ_ERR_INVALID_ARG, "bbq_rerank_scores: null argument");
  if (offsets[0] != 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_rerank_scores: offsets[0] must be 0");
  int64_t max_count = 0;
  for (int32_t q = 0; q < n_queries; ++q) {
    const int64_t c = offsets[q + 1] - offsets[q];
    if (c < 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_rerank_scores: offsets must ascend");
    max_count = std::max(max_count, c);
  }
  const int64_t total = offsets[n_queries];
  if (total == 0) return BBQ_OK;
  if (!rows || !out_true) return fail(BBQ_ERR_INVALID_ARG, "bbq_rerank_scores: null argument");
  for (int64_t i = 0; i < total; ++i)
    if (rows[i] < 0 || rows[i] >= v->n) return fail(BBQ_ERR_INVALID_ARG, "基础向量%d不存在", rows[i]);
  std::lock_guard<std::mutex> lk(v->ctx->mu);
  HIPCHK(hipSetDevice(v->device));
  int rc = grow(&v->d_q, &v->q_cap, (int64_t)n_queries * v->dim);
  if (rc == BBQ_OK) rc = grow(&v->d_off, &v->off_cap, (int64_t)n_queries + 1);
  if (rc == BBQ_OK && total > v->cand_cap) {
    int64_t c1 = v->cand_cap, c2 = v->cand_cap;
    rc = grow(&v->d_rows, &c1, total);
    if (rc == BBQ_OK) rc = grow(&v->d_out, &c2, total);
    v->cand_cap = rc == BBQ_OK ? std::min(c1, c2) : 0;
  }
  if (rc != BBQ_OK) return rc;
  hipStream_t st = v->ctx->aux_stream;
  HIPCHK(hipMemcpyAsync(v->d_q, queries, (size_t)n_queries * v->dim * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(v->d_off, offsets, (size_t)(n_queries + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(v->d_rows, rows, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice, st));
  RerankArgs a{};
  a.vecs = v->d;
  a.n = v->n;
  a.dim = v->dim;
  a.sim = true_sim;
  a.queries = v->d_q;
  a.offsets = v->d_off;
  a.rows = v->d_rows;
  a.out = v->d_out;
  HIPCHK(launch_rerank(a, n_queries, max_count, st));
  HIPCHK(hipMemcpyAsync(out_true, v->d_out, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return BBQ_OK;
}

int bbq_search_rerank_batch(bbq_index *ix, bbq_vectors *v, int32_t n_queries, const float *queries, const uint8_t *qquant,
                            const double *qcorr, int32_t query_bits, int32_t sim, int64_t k, int32_t factor, int32_t selector,
                            int32_t true_sim, int32_t *out_idx, float *out_quantized, double *out_true, int64_t *out_n) {
  clear_error();
  if (!ix || !v) return fail(BBQ_ERR_INVALID_ARG, "bbq_search_rerank_batch: null handle");
  if (k < 0) return fail(BBQ_ERR_INVALID_ARG, "k必须是非负数");
  if (factor < 1) return fail(BBQ_ERR_INVALID_ARG, "bbq_search_rerank_batch: factor must be >= 1");
  if (selector != 0 && selector != 1) return fail(BBQ_ERR_INVALID_ARG, "bbq_search_rerank_batch: selector must be 0 (heap) or 1 (sort)");
  if (v->dim != ix->dim) return fail(BBQ_ERR_DIM_MISMATCH, "bbq_search_rerank_batch: vectors are %d-d, index is %d-d", v->dim, ix->dim);
  if (v->n < ix->n_rows) return fail(BBQ_ERR_INVALID_ARG, "bbq_search_rerank_batch: %lld vectors for %lld index rows", (long long)v->n, (long long)ix->n_rows);
  if (n_queries > 0 && (!out_n || !queries)) return fail(BBQ_ERR_INVALID_ARG, "bbq_search_rerank_batch: null argument");
  if (k > 0 && k * (int64_t)factor / factor != k) return fail(BBQ_ERR_INVALID_ARG, "bbq_search_rerank_batch: k*factor overflows");
  const int64_t kk = k * (int64_t)factor;
  const int64_t kq = std::min<int64_t>(kk, ix->n_rows);  // candidates a query can return
  std::vector<int32_t> cidx((size_t)n_queries * (size_t)std::max<int64_t>(kk, 1));
  std::vector<float> csc(cidx.size());
  std::vector<int64_t> cn((size_t)std::max(n_queries, 1));
  // the search strides its outputs by its k; ask for kk but only kq entries per query can be filled
  int rc = bbq_search_batch(ix, n_queries, qquant, qcorr, query_bits, sim, kk, cidx.data(), csc.data(), cn.data());
  if (rc != BBQ_OK) return rc;
  (void)kq;
  for (int32_t q = 0; q < n_queries; ++q) out_n[q] = 0;
  if (n_queries == 0 || k == 0) return BBQ_OK;
  if (!out_idx || !out_quantized || !out_true) return fail(BBQ_ERR_INVALID_ARG, "output arrays are null");
  std::vector<int64_t> off((size_t)n_queries + 1, 0);
  for (int32_t q = 0; q < n_queries; ++q) off[q + 1] = off[q] + cn[q];
  std::vector<int32_t> rows((size_t)off[n_queries]);
  for (int32_t q = 0; q < n_queries; ++q)
    std::copy(cidx.begin() + (int64_t)q * kk, cidx.begin() + (int64_t)q * kk + cn[q], rows.begin() + off[q]);
  std::vector<double> ts(rows.size());
  rc = bbq_rerank_scores(v, n_queries, queries, off.data(), rows.data(), true_sim, ts.data());
  if (rc != BBQ_OK) return rc;
  std::vector<Ranked> r;
  std::vector<int32_t> tag((size_t)k + 1);
  std::vector<double> tsc((size_t)k + 1);
  for (int32_t q = 0; q < n_queries; ++q) {
    const int64_t cnt = cn[q];
    const double *t = ts.data() + off[q];
    r.clear();
    if (selector == 0) {  // src/topKSelector.ts:40-76
      HeapReplay h(k, INT64_MAX);
      for (int64_t i = 0; i < cnt; ++i) h.offer64(t[i], (int32_t)i);
      const int64_t m = h.drain_ascending(tag.data(), tsc.data());
      for (int64_t j = 0; j < m; ++j) r.push_back(Ranked{tsc[j], tag[j]});
      sort_desc_stable(r);
    } else {  // :102-114
      for (int64_t i = 0; i < cnt; ++i) r.push_back(Ranked{t[i], (int32_t)i});
      sort_desc_stable(r);
      if ((int64_t)r.size() > k) r.resize((size_t)k);
    }
    for (size_t j = 0; j < r.size(); ++j) {
      out_idx[(int64_t)q * k + j] = cidx[(int64_t)q * kk + r[j].pos];
      out_quantized[(int64_t)q * k + j] = csc[(int64_t)q * kk + r[j].pos];
      out_true[(int64_t)q * k + j] = r[j].score;
    }
    out_n[q] = (int64_t)r.size();
  }
  return BBQ_OK;
}

}  // extern "C"
