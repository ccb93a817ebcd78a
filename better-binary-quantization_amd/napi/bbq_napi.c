/*
 * bbq_napi.c - thin Node N-API (v8, plain C, no node-addon-api) binding of libbbq's C ABI (include/bbq.h).
 * The JavaScript host (../js/index.js) keeps the reference's public API and calls these functions; typed arrays
 * are passed zero-copy (napi_get_typedarray_info) and are only valid for the duration of the call, exactly as
 * the C ABI borrows them.  Errors become JS `Error`s carrying bbq_last_error() (the reference's own messages).
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/bbq.h"

#define NAPI_CALL(env, call)                                              \
  do {                                                                    \
    if ((call) != napi_ok) {                                              \
      napi_throw_error((env), NULL, "bbq_napi: N-API call failed: " #call); \
      return NULL;                                                        \
    }                                                                     \
  } while (0)

static napi_value throw_bbq(napi_env env, int rc) {
  char code[16];
  snprintf(code, sizeof code, "BBQ%d", rc);
  const char *m = bbq_last_error();
  napi_throw_error(env, code, (m && *m) ? m : "libbbq error");
  return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "bbq_napi: wrong number of arguments");
    return 0;
  }
  return 1;
}

static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len) {
  bool is = false;
  napi_typedarray_type t;
  napi_value ab;
  size_t off;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is ||
      napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok || t != want) {
    napi_throw_type_error(env, NULL, "bbq_napi: typed array of the wrong kind");
    return 0;
  }
  return 1;
}

static int get_i64(napi_env env, napi_value v, int64_t *out) {
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok) {
    napi_throw_type_error(env, NULL, "bbq_napi: number expected");
    return 0;
  }
  *out = (int64_t)d;
  return 1;
}
static int get_f64(napi_env env, napi_value v, double *out) {
  if (napi_get_value_double(env, v, out) != napi_ok) {
    napi_throw_type_error(env, NULL, "bbq_napi: number expected");
    return 0;
  }
  return 1;
}

static napi_value new_typed(napi_env env, napi_typedarray_type t, size_t count, size_t elem, void **data) {
  napi_value ab, ta;
  if (napi_create_arraybuffer(env, count * elem, data, &ab) != napi_ok) return NULL;
  if (napi_create_typedarray(env, t, count, ab, 0, &ta) != napi_ok) return NULL;
  return ta;
}

static void set_prop(napi_env env, napi_value obj, const char *name, napi_value v) { napi_set_named_property(env, obj, name, v); }

/* deviceCount() */
static napi_value DeviceCount(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r;
  NAPI_CALL(env, napi_create_int32(env, bbq_device_count(), &r));
  return r;
}

/* quantizeVectors(flat Float32Array, n, dim, sim, indexBits, lambda, iters, threads) -> {codes, corr, centroid} */
static napi_value QuantizeVectors(napi_env env, napi_callback_info info) {
  napi_value a[8];
  if (!get_args(env, info, 8, a)) return NULL;
  void *vec; size_t vlen;
  int64_t n, dim, sim, ib, iters, threads; double lambda;
  if (!get_typed(env, a[0], napi_float32_array, &vec, &vlen) || !get_i64(env, a[1], &n) || !get_i64(env, a[2], &dim) ||
      !get_i64(env, a[3], &sim) || !get_i64(env, a[4], &ib) || !get_f64(env, a[5], &lambda) || !get_i64(env, a[6], &iters) ||
      !get_i64(env, a[7], &threads)) return NULL;
  if (n < 0 || dim <= 0 || (size_t)(n * dim) != vlen) { napi_throw_range_error(env, NULL, "bbq_napi: n*dim does not match the array"); return NULL; }
  const size_t rb = ib == 1 ? (size_t)((dim + 7) / 8) : (size_t)dim;
  void *codes, *corr, *cen;
  napi_value tcodes = new_typed(env, napi_uint8_array, (size_t)n * rb, 1, &codes);
  napi_value tcorr = new_typed(env, napi_float64_array, (size_t)n * 4, 8, &corr);
  napi_value tcen = new_typed(env, napi_float32_array, (size_t)dim, 4, &cen);
  if (!tcodes || !tcorr || !tcen) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int rc = bbq_quantize_vectors((const float *)vec, n, (int32_t)dim, (int32_t)sim, (int32_t)ib, lambda, (int32_t)iters, (int32_t)threads,
                                (uint8_t *)codes, (double *)corr, (float *)cen, NULL, NULL);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "codes", tcodes); set_prop(env, o, "corr", tcorr); set_prop(env, o, "centroid", tcen);
  return o;
}

/* quantizeQuery(query Float32Array, centroid Float32Array, sim, queryBits, lambda, iters, searchPath) -> {quantizedQuery, corrections} */
static napi_value QuantizeQuery(napi_env env, napi_callback_info info) {
  napi_value a[7];
  if (!get_args(env, info, 7, a)) return NULL;
  void *q, *c; size_t ql, cl;
  int64_t sim, qb, iters; double lambda; bool sp;
  if (!get_typed(env, a[0], napi_float32_array, &q, &ql) || !get_typed(env, a[1], napi_float32_array, &c, &cl) ||
      !get_i64(env, a[2], &sim) || !get_i64(env, a[3], &qb) || !get_f64(env, a[4], &lambda) || !get_i64(env, a[5], &iters)) return NULL;
  NAPI_CALL(env, napi_get_value_bool(env, a[6], &sp));
  if (ql != cl) { napi_throw_error(env, "BBQ6", "向量和质心维度不匹配"); return NULL; }
  void *qq, *qc;
  napi_value tqq = new_typed(env, napi_uint8_array, ql, 1, &qq);
  napi_value tqc = new_typed(env, napi_float64_array, 4, 8, &qc);
  if (!tqq || !tqc) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int rc = sp ? bbq_quantize_query((const float *)q, (int32_t)ql, (const float *)c, (int32_t)sim, (int32_t)qb, lambda, (int32_t)iters, (uint8_t *)qq, (double *)qc)
              : bbq_quantize_query_vector((const float *)q, (int32_t)ql, (const float *)c, (int32_t)sim, (int32_t)qb, lambda, (int32_t)iters, (uint8_t *)qq, (double *)qc);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "quantizedQuery", tqq); set_prop(env, o, "corrections", tqc);
  return o;
}

/* centroidDP(centroid Float32Array) */
static napi_value CentroidDP(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  void *c; size_t cl;
  if (!get_typed(env, a[0], napi_float32_array, &c, &cl)) return NULL;
  napi_value r;
  NAPI_CALL(env, napi_create_double(env, bbq_centroid_dp((const float *)c, (int32_t)cl), &r));
  return r;
}

static void finalize_index(napi_env env, void *data, void *hint) {
  (void)env; (void)hint;
  bbq_index **box = (bbq_index **)data;
  if (*box) bbq_index_destroy(*box);
  free(box);
}

static bbq_index *unbox(napi_env env, napi_value v) {
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !*(bbq_index **)p) {
    napi_throw_error(env, NULL, "目标向量集合不能为空");
    return NULL;
  }
  return *(bbq_index **)p;
}

/* indexCreate(codes Uint8Array, corr Float64Array, n, dim, indexBits, centroidDP, device) -> external */
static napi_value IndexCreate(napi_env env, napi_callback_info info) {
  napi_value a[7];
  if (!get_args(env, info, 7, a)) return NULL;
  void *codes, *corr; size_t cl, rl;
  int64_t n, dim, ib, dev; double cdp;
  if (!get_typed(env, a[0], napi_uint8_array, &codes, &cl) || !get_typed(env, a[1], napi_float64_array, &corr, &rl) ||
      !get_i64(env, a[2], &n) || !get_i64(env, a[3], &dim) || !get_i64(env, a[4], &ib) || !get_f64(env, a[5], &cdp) ||
      !get_i64(env, a[6], &dev)) return NULL;
  if (n < 0 || dim <= 0 || rl != (size_t)n * 4 || cl < (size_t)n * (size_t)(ib == 1 ? (dim + 7) / 8 : dim)) {
    napi_throw_range_error(env, NULL, "bbq_napi: array sizes do not match n/dim"); return NULL;
  }
  bbq_index **box = (bbq_index **)calloc(1, sizeof *box);
  int rc = bbq_index_create((const uint8_t *)codes, (const double *)corr, n, (int32_t)dim, (int32_t)ib, cdp, (int32_t)dev, box);
  if (rc != BBQ_OK) { free(box); return throw_bbq(env, rc); }
  napi_value ext;
  if (napi_create_external(env, box, finalize_index, NULL, &ext) != napi_ok) { bbq_index_destroy(*box); free(box); napi_throw_error(env, NULL, "bbq_napi: external"); return NULL; }
  return ext;
}

/* indexCreateMulti(codes Uint8Array, corr Float64Array, n, dim, indexBits, centroidDP, devices Int32Array, pilotRows) -> external
 * one index row-sharded over the listed devices behind one handle (bbq_index_create_multi) */
static napi_value IndexCreateMulti(napi_env env, napi_callback_info info) {
  napi_value a[8];
  if (!get_args(env, info, 8, a)) return NULL;
  void *codes, *corr, *devs; size_t cl, rl, dl;
  int64_t n, dim, ib, pilot; double cdp;
  if (!get_typed(env, a[0], napi_uint8_array, &codes, &cl) || !get_typed(env, a[1], napi_float64_array, &corr, &rl) ||
      !get_i64(env, a[2], &n) || !get_i64(env, a[3], &dim) || !get_i64(env, a[4], &ib) || !get_f64(env, a[5], &cdp) ||
      !get_typed(env, a[6], napi_int32_array, &devs, &dl) || !get_i64(env, a[7], &pilot)) return NULL;
  if (n < 0 || dim <= 0 || rl != (size_t)n * 4 || cl < (size_t)n * (size_t)(ib == 1 ? (dim + 7) / 8 : dim) || dl < 1) {
    napi_throw_range_error(env, NULL, "bbq_napi: array sizes do not match n/dim"); return NULL;
  }
  bbq_index **box = (bbq_index **)calloc(1, sizeof *box);
  int rc = bbq_index_create_multi((const uint8_t *)codes, (const double *)corr, n, (int32_t)dim, (int32_t)ib, cdp, (int32_t)dl,
                                  (const int32_t *)devs, pilot, box);
  if (rc != BBQ_OK) { free(box); return throw_bbq(env, rc); }
  napi_value ext;
  if (napi_create_external(env, box, finalize_index, NULL, &ext) != napi_ok) { bbq_index_destroy(*box); free(box); napi_throw_error(env, NULL, "bbq_napi: external"); return NULL; }
  return ext;
}

/* indexBuild(flat Float32Array, n, dim, sim, lambda, iters, device, indexBits) -> {handle, codes, corr, centroid}
 * quantizeVectors on the device (bbq_index_build_bits); the handle is the ready device index */
static napi_value IndexBuild(napi_env env, napi_callback_info info) {
  napi_value a[8];
  if (!get_args(env, info, 8, a)) return NULL;
  void *vec; size_t vlen;
  int64_t n, dim, sim, iters, dev, ib; double lambda;
  if (!get_typed(env, a[0], napi_float32_array, &vec, &vlen) || !get_i64(env, a[1], &n) || !get_i64(env, a[2], &dim) ||
      !get_i64(env, a[3], &sim) || !get_f64(env, a[4], &lambda) || !get_i64(env, a[5], &iters) || !get_i64(env, a[6], &dev) ||
      !get_i64(env, a[7], &ib)) return NULL;
  if (n < 0 || dim <= 0 || (size_t)(n * dim) != vlen) { napi_throw_range_error(env, NULL, "bbq_napi: n*dim does not match the array"); return NULL; }
  void *codes, *corr, *cen;
  napi_value tcodes = new_typed(env, napi_uint8_array, (size_t)n * (size_t)(ib == 1 ? (dim + 7) / 8 : dim), 1, &codes);
  napi_value tcorr = new_typed(env, napi_float64_array, (size_t)n * 4, 8, &corr);
  napi_value tcen = new_typed(env, napi_float32_array, (size_t)dim, 4, &cen);
  if (!tcodes || !tcorr || !tcen) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  bbq_index **box = (bbq_index **)calloc(1, sizeof *box);
  int rc = bbq_index_build_bits((const float *)vec, n, (int32_t)dim, (int32_t)sim, (int32_t)ib, lambda, (int32_t)iters, (int32_t)dev, box,
                                (float *)cen, (uint8_t *)codes, (double *)corr, NULL, NULL);
  if (rc != BBQ_OK) { free(box); return throw_bbq(env, rc); }
  napi_value ext, o;
  if (napi_create_external(env, box, finalize_index, NULL, &ext) != napi_ok) { bbq_index_destroy(*box); free(box); napi_throw_error(env, NULL, "bbq_napi: external"); return NULL; }
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "handle", ext); set_prop(env, o, "codes", tcodes); set_prop(env, o, "corr", tcorr); set_prop(env, o, "centroid", tcen);
  return o;
}

/* indexDestroy(handle) */
static napi_value IndexDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  void *p = NULL;
  if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
    bbq_index **box = (bbq_index **)p;
    if (*box) { bbq_index_destroy(*box); *box = NULL; }
  }
  napi_value u; napi_get_undefined(env, &u); return u;
}

/* searchBatch(handle, nq, qquant Uint8Array[nq*dim], qcorr Float64Array[nq*4], queryBits, sim, k) -> {indices Int32Array[nq*k], scores Float32Array[nq*k], counts Float64Array[nq]} */
static napi_value SearchBatch(napi_env env, napi_callback_info info) {
  napi_value a[7];
  if (!get_args(env, info, 7, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  void *qq, *qc; size_t ql, cl;
  int64_t nq, qb, sim, k;
  if (!get_i64(env, a[1], &nq) || !get_typed(env, a[2], napi_uint8_array, &qq, &ql) || !get_typed(env, a[3], napi_float64_array, &qc, &cl) ||
      !get_i64(env, a[4], &qb) || !get_i64(env, a[5], &sim) || !get_i64(env, a[6], &k)) return NULL;
  if (nq < 0 || ql != (size_t)nq * (size_t)bbq_index_dimension(ix) || cl != (size_t)nq * 4) {
    napi_throw_error(env, "BBQ6", "查询向量维度与目标向量维度不匹配"); return NULL;
  }
  if (k < 0) { napi_throw_error(env, "BBQ7", "k值不能为负数"); return NULL; }
  int64_t keff = k < bbq_index_size(ix) ? k : bbq_index_size(ix);
  void *oi, *os, *on;
  napi_value ti = new_typed(env, napi_int32_array, (size_t)(nq * keff), 4, &oi);
  napi_value ts = new_typed(env, napi_float32_array, (size_t)(nq * keff), 4, &os);
  napi_value tn = new_typed(env, napi_float64_array, (size_t)nq, 8, &on);
  if (!ti || !ts || !tn) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int64_t *cnt = (int64_t *)calloc((size_t)nq + 1, sizeof(int64_t));
  /* outputs are strided by the k passed to the library: pass keff so rows are packed */
  int rc = bbq_search_batch(ix, (int32_t)nq, (const uint8_t *)qq, (const double *)qc, (int32_t)qb, (int32_t)sim, keff, (int32_t *)oi, (float *)os, cnt);
  if (rc != BBQ_OK) { free(cnt); return throw_bbq(env, rc); }
  for (int64_t i = 0; i < nq; ++i) ((double *)on)[i] = (double)cnt[i];
  free(cnt);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "indices", ti); set_prop(env, o, "scores", ts); set_prop(env, o, "counts", tn);
  napi_value kv; napi_create_double(env, (double)keff, &kv); set_prop(env, o, "stride", kv);
  return o;
}

/* searchRawBatch(handle, nq, flat Float32Array[nq*dim] raw queries, centroid Float32Array[dim], sim, queryBits, lambda, iters, threads, k)
 *   -> {indices, scores, counts, stride}   (bbq_search_raw_batch: quantization on host threads pipelined with the sweeps) */
static napi_value SearchRawBatch(napi_env env, napi_callback_info info) {
  napi_value a[10];
  if (!get_args(env, info, 10, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  void *q, *cen; size_t ql, cl;
  int64_t nq, sim, qb, iters, threads, k; double lambda;
  if (!get_i64(env, a[1], &nq) || !get_typed(env, a[2], napi_float32_array, &q, &ql) || !get_typed(env, a[3], napi_float32_array, &cen, &cl) ||
      !get_i64(env, a[4], &sim) || !get_i64(env, a[5], &qb) || !get_f64(env, a[6], &lambda) || !get_i64(env, a[7], &iters) ||
      !get_i64(env, a[8], &threads) || !get_i64(env, a[9], &k)) return NULL;
  if (nq < 0 || cl != (size_t)bbq_index_dimension(ix) || ql != (size_t)nq * cl) {
    napi_throw_error(env, "BBQ6", "查询向量维度与目标向量维度不匹配"); return NULL;
  }
  if (k < 0) { napi_throw_error(env, "BBQ7", "k值不能为负数"); return NULL; }
  int64_t keff = k < bbq_index_size(ix) ? k : bbq_index_size(ix);
  void *oi, *os, *on;
  napi_value ti = new_typed(env, napi_int32_array, (size_t)(nq * keff), 4, &oi);
  napi_value ts = new_typed(env, napi_float32_array, (size_t)(nq * keff), 4, &os);
  napi_value tn = new_typed(env, napi_float64_array, (size_t)nq, 8, &on);
  if (!ti || !ts || !tn) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int64_t *cnt = (int64_t *)calloc((size_t)nq + 1, sizeof(int64_t));
  int rc = bbq_search_raw_batch(ix, (int32_t)nq, (const float *)q, (const float *)cen, (int32_t)sim, (int32_t)qb, lambda, (int32_t)iters,
                                (int32_t)threads, keff, (int32_t *)oi, (float *)os, cnt, NULL, NULL, NULL);
  if (rc != BBQ_OK) { free(cnt); return throw_bbq(env, rc); }
  for (int64_t i = 0; i < nq; ++i) ((double *)on)[i] = (double)cnt[i];
  free(cnt);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "indices", ti); set_prop(env, o, "scores", ts); set_prop(env, o, "counts", tn);
  napi_value kv; napi_create_double(env, (double)keff, &kv); set_prop(env, o, "stride", kv);
  return o;
}

/* searchRawInto(handle, query Float32Array[dim] raw, centroid Float32Array[dim], sim, queryBits, lambda, iters, k, outIndices Int32Array[>= k],
 *               outScores Float32Array[>= k]) -> count
 * The reference's own call shape - ONE synchronous searchNearestNeighbors (src/binaryQuantizationFormat.ts:308-412) - without a typed
 * array or an object created per call: the caller owns and reuses the output arrays.  (Every ArrayBuffer the batch form creates counts
 * as external memory; their collection was the 0.8 ms tail of a 0.2 ms call.) */
static napi_value SearchRawInto(napi_env env, napi_callback_info info) {
  napi_value a[10];
  if (!get_args(env, info, 10, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  void *q, *cen, *oi, *os; size_t ql, cl, il, sl;
  int64_t sim, qb, iters, k; double lambda;
  if (!get_typed(env, a[1], napi_float32_array, &q, &ql) || !get_typed(env, a[2], napi_float32_array, &cen, &cl) ||
      !get_i64(env, a[3], &sim) || !get_i64(env, a[4], &qb) || !get_f64(env, a[5], &lambda) || !get_i64(env, a[6], &iters) ||
      !get_i64(env, a[7], &k) || !get_typed(env, a[8], napi_int32_array, &oi, &il) || !get_typed(env, a[9], napi_float32_array, &os, &sl)) return NULL;
  if (cl != (size_t)bbq_index_dimension(ix) || ql != cl) { napi_throw_error(env, "BBQ6", "查询向量维度与目标向量维度不匹配"); return NULL; }
  if (k < 0) { napi_throw_error(env, "BBQ7", "k值不能为负数"); return NULL; }
  int64_t keff = k < bbq_index_size(ix) ? k : bbq_index_size(ix);
  if (il < (size_t)keff || sl < (size_t)keff) { napi_throw_error(env, NULL, "bbq_napi: output arrays shorter than k"); return NULL; }
  int64_t cnt[2] = {0, 0};
  int rc = bbq_search_raw_batch(ix, 1, (const float *)q, (const float *)cen, (int32_t)sim, (int32_t)qb, lambda, (int32_t)iters, 1, keff,
                                (int32_t *)oi, (float *)os, cnt, NULL, NULL, NULL);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value n;
  NAPI_CALL(env, napi_create_double(env, (double)cnt[0], &n));
  return n;
}

/* scoreRows(handle, qquant, qcorr, queryBits, sim, rowBegin, rowCount) -> {qcDist Int32Array, score64 Float64Array, score32 Float32Array} */
static napi_value ScoreRows(napi_env env, napi_callback_info info) {
  napi_value a[7];
  if (!get_args(env, info, 7, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  void *qq, *qc; size_t ql, cl;
  int64_t qb, sim, rb, rc_;
  if (!get_typed(env, a[1], napi_uint8_array, &qq, &ql) || !get_typed(env, a[2], napi_float64_array, &qc, &cl) ||
      !get_i64(env, a[3], &qb) || !get_i64(env, a[4], &sim) || !get_i64(env, a[5], &rb) || !get_i64(env, a[6], &rc_)) return NULL;
  if (ql != (size_t)bbq_index_dimension(ix) || cl != 4) { napi_throw_error(env, "BBQ6", "查询向量维度与目标向量维度不匹配"); return NULL; }
  if (rc_ < 0) rc_ = 0;
  void *od, *o64, *o32;
  napi_value td = new_typed(env, napi_int32_array, (size_t)rc_, 4, &od);
  napi_value t64 = new_typed(env, napi_float64_array, (size_t)rc_, 8, &o64);
  napi_value t32 = new_typed(env, napi_float32_array, (size_t)rc_, 4, &o32);
  if (!td || !t64 || !t32) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int rc = bbq_score_rows(ix, (const uint8_t *)qq, (const double *)qc, (int32_t)qb, (int32_t)sim, rb, rc_, (int32_t *)od, (double *)o64, (float *)o32);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "qcDist", td); set_prop(env, o, "score64", t64); set_prop(env, o, "score32", t32);
  return o;
}

/* setOption(handle, name, value) */
static napi_value SetOption(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  char name[64]; size_t nl; int64_t v;
  NAPI_CALL(env, napi_get_value_string_utf8(env, a[1], name, sizeof name, &nl));
  if (!get_i64(env, a[2], &v)) return NULL;
  int rc = bbq_set_option(ix, name, v);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value u; napi_get_undefined(env, &u); return u;
}

/* stats(handle) -> {lastScanMs, lastScanBytes, candidates, denseFallbacks} */
static napi_value Stats(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  bbq_stats st;
  int rc = bbq_get_stats(ix, &st);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value o, v;
  NAPI_CALL(env, napi_create_object(env, &o));
  napi_create_double(env, st.last_scan_ms, &v); set_prop(env, o, "lastScanMs", v);
  napi_create_double(env, (double)st.last_scan_bytes, &v); set_prop(env, o, "lastScanBytes", v);
  napi_create_double(env, (double)st.candidates, &v); set_prop(env, o, "candidates", v);
  napi_create_double(env, (double)st.dense_fallbacks, &v); set_prop(env, o, "denseFallbacks", v);
  napi_create_double(env, (double)st.host_replays, &v); set_prop(env, o, "hostReplays", v);
  napi_create_double(env, (double)st.resident_bytes, &v); set_prop(env, o, "residentBytes", v);
  napi_create_double(env, (double)bbq_index_shards(ix), &v); set_prop(env, o, "shards", v);
  napi_create_double(env, (double)bbq_index_bytes_per_row(ix), &v); set_prop(env, o, "bytesPerRow", v);
  return o;
}

/* ---- oversample + exact rerank (bbq_vectors_*, bbq_rerank_scores, bbq_search_rerank_batch) */
static void finalize_vectors(napi_env env, void *data, void *hint) {
  (void)env; (void)hint;
  bbq_vectors **box = (bbq_vectors **)data;
  if (*box) bbq_vectors_destroy(*box);
  free(box);
}

static bbq_vectors *unbox_vectors(napi_env env, napi_value v) {
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !*(bbq_vectors **)p) {
    napi_throw_error(env, NULL, "向量不能为空");
    return NULL;
  }
  return *(bbq_vectors **)p;
}

/* vectorsCreate(flat Float32Array, n, dim, device) -> external */
static napi_value VectorsCreate(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  void *vec; size_t vlen;
  int64_t n, dim, dev;
  if (!get_typed(env, a[0], napi_float32_array, &vec, &vlen) || !get_i64(env, a[1], &n) || !get_i64(env, a[2], &dim) ||
      !get_i64(env, a[3], &dev)) return NULL;
  if (n < 0 || dim <= 0 || (size_t)(n * dim) != vlen) { napi_throw_range_error(env, NULL, "bbq_napi: n*dim does not match the array"); return NULL; }
  bbq_vectors **box = (bbq_vectors **)calloc(1, sizeof *box);
  int rc = bbq_vectors_create((const float *)vec, n, (int32_t)dim, (int32_t)dev, box);
  if (rc != BBQ_OK) { free(box); return throw_bbq(env, rc); }
  napi_value ext;
  if (napi_create_external(env, box, finalize_vectors, NULL, &ext) != napi_ok) { bbq_vectors_destroy(*box); free(box); napi_throw_error(env, NULL, "bbq_napi: external"); return NULL; }
  return ext;
}

/* vectorsDestroy(handle) */
static napi_value VectorsDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  void *p = NULL;
  if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
    bbq_vectors **box = (bbq_vectors **)p;
    if (*box) { bbq_vectors_destroy(*box); *box = NULL; }
  }
  napi_value u; napi_get_undefined(env, &u); return u;
}

/* rerankScores(vectors, nq, queries Float32Array[nq*dim], offsets Float64Array[nq+1], rows Int32Array, trueSim) -> Float64Array */
static napi_value RerankScores(napi_env env, napi_callback_info info) {
  napi_value a[6];
  if (!get_args(env, info, 6, a)) return NULL;
  bbq_vectors *v = unbox_vectors(env, a[0]);
  if (!v) return NULL;
  void *q, *off, *rows; size_t ql, ol, rl;
  int64_t nq, sim;
  if (!get_i64(env, a[1], &nq) || !get_typed(env, a[2], napi_float32_array, &q, &ql) || !get_typed(env, a[3], napi_float64_array, &off, &ol) ||
      !get_typed(env, a[4], napi_int32_array, &rows, &rl) || !get_i64(env, a[5], &sim)) return NULL;
  if (nq < 0 || ql != (size_t)nq * (size_t)bbq_vectors_dimension(v)) { napi_throw_error(env, "BBQ6", "向量维度不匹配"); return NULL; }
  if (ol != (size_t)nq + 1) { napi_throw_range_error(env, NULL, "bbq_napi: offsets must have nq+1 entries"); return NULL; }
  int64_t *o64 = (int64_t *)calloc((size_t)nq + 1, sizeof(int64_t));
  for (int64_t i = 0; i <= nq; ++i) o64[i] = (int64_t)((double *)off)[i];
  if (o64[nq] < 0 || (size_t)o64[nq] != rl) { free(o64); napi_throw_range_error(env, NULL, "bbq_napi: offsets do not match rows"); return NULL; }
  void *out;
  napi_value t = new_typed(env, napi_float64_array, rl, 8, &out);
  if (!t) { free(o64); napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int rc = bbq_rerank_scores(v, (int32_t)nq, (const float *)q, o64, (const int32_t *)rows, (int32_t)sim, (double *)out);
  free(o64);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  return t;
}

/* searchRerankBatch(index, vectors, nq, queries Float32Array, qquant Uint8Array, qcorr Float64Array, queryBits, sim, k, factor, selector, trueSim)
 *   -> {indices Int32Array[nq*k], quantized Float32Array[nq*k], trueScores Float64Array[nq*k], counts Float64Array[nq], stride} */
static napi_value SearchRerankBatch(napi_env env, napi_callback_info info) {
  napi_value a[12];
  if (!get_args(env, info, 12, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  bbq_vectors *v = unbox_vectors(env, a[1]);
  if (!v) return NULL;
  void *q, *qq, *qc; size_t fl, ql, cl;
  int64_t nq, qb, sim, k, factor, selector, tsim;
  if (!get_i64(env, a[2], &nq) || !get_typed(env, a[3], napi_float32_array, &q, &fl) || !get_typed(env, a[4], napi_uint8_array, &qq, &ql) ||
      !get_typed(env, a[5], napi_float64_array, &qc, &cl) || !get_i64(env, a[6], &qb) || !get_i64(env, a[7], &sim) || !get_i64(env, a[8], &k) ||
      !get_i64(env, a[9], &factor) || !get_i64(env, a[10], &selector) || !get_i64(env, a[11], &tsim)) return NULL;
  const size_t dim = (size_t)bbq_index_dimension(ix);
  if (nq < 0 || ql != (size_t)nq * dim || fl != (size_t)nq * dim || cl != (size_t)nq * 4) {
    napi_throw_error(env, "BBQ6", "查询向量维度与目标向量维度不匹配"); return NULL;
  }
  if (k < 0) { napi_throw_error(env, "BBQ7", "k值不能为负数"); return NULL; }
  void *oi, *oq, *ot, *on;
  napi_value ti = new_typed(env, napi_int32_array, (size_t)(nq * k), 4, &oi);
  napi_value tq = new_typed(env, napi_float32_array, (size_t)(nq * k), 4, &oq);
  napi_value tt = new_typed(env, napi_float64_array, (size_t)(nq * k), 8, &ot);
  napi_value tn = new_typed(env, napi_float64_array, (size_t)nq, 8, &on);
  if (!ti || !tq || !tt || !tn) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int64_t *cnt = (int64_t *)calloc((size_t)nq + 1, sizeof(int64_t));
  int rc = bbq_search_rerank_batch(ix, v, (int32_t)nq, (const float *)q, (const uint8_t *)qq, (const double *)qc, (int32_t)qb, (int32_t)sim, k,
                                   (int32_t)factor, (int32_t)selector, (int32_t)tsim, (int32_t *)oi, (float *)oq, (double *)ot, cnt);
  if (rc != BBQ_OK) { free(cnt); return throw_bbq(env, rc); }
  for (int64_t i = 0; i < nq; ++i) ((double *)on)[i] = (double)cnt[i];
  free(cnt);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "indices", ti); set_prop(env, o, "quantized", tq); set_prop(env, o, "trueScores", tt); set_prop(env, o, "counts", tn);
  napi_value kv; napi_create_double(env, (double)k, &kv); set_prop(env, o, "stride", kv);
  return o;
}

/* ---- on-disk format (bbq_index_save / bbq_index_load / bbq_index_export) */
static int get_path(napi_env env, napi_value v, char *buf, size_t cap) {
  size_t len = 0;
  if (napi_get_value_string_utf8(env, v, buf, cap, &len) != napi_ok || len == 0 || len >= cap - 1) {
    napi_throw_type_error(env, NULL, "bbq_napi: path string expected");
    return 0;
  }
  return 1;
}

/* indexSave(handle, prefix, centroid Float32Array, sim) */
static napi_value IndexSave(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  char path[4096]; void *cen; size_t cl; int64_t sim;
  if (!get_path(env, a[1], path, sizeof path) || !get_typed(env, a[2], napi_float32_array, &cen, &cl) || !get_i64(env, a[3], &sim)) return NULL;
  if (cl != (size_t)bbq_index_dimension(ix)) { napi_throw_error(env, "BBQ6", "向量和质心维度不匹配"); return NULL; }
  int rc = bbq_index_save(ix, path, (const float *)cen, (int32_t)sim);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value u; napi_get_undefined(env, &u); return u;
}

/* indexLoad(prefix, device[, devices Int32Array]) -> {handle, centroid Float32Array, n, dim, sim, centroidDP, rowBase, shards}
 * devices given and the files hold a multi-device index (a manifest): its shards go over those devices (bbq_index_load_multi) */
static napi_value IndexLoad(napi_env env, napi_callback_info info) {
  napi_value a[3];
  size_t argc = 3;
  if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 2) { napi_throw_type_error(env, NULL, "bbq_napi: wrong number of arguments"); return NULL; }
  char path[4096]; int64_t dev;
  if (!get_path(env, a[0], path, sizeof path) || !get_i64(env, a[1], &dev)) return NULL;
  void *devs = NULL; size_t ndevs = 0;
  if (argc >= 3) {
    napi_valuetype vt;
    if (napi_typeof(env, a[2], &vt) == napi_ok && vt != napi_undefined && vt != napi_null && !get_typed(env, a[2], napi_int32_array, &devs, &ndevs)) return NULL;
  }
  int64_t n = 0, rb = 0; int32_t dim = 0, sim = 0; double cdp = 0;
  int rc = bbq_index_file_info(path, &n, &dim, &sim, &cdp, &rb);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  void *cen;
  napi_value tcen = new_typed(env, napi_float32_array, (size_t)dim, 4, &cen);
  if (!tcen) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  bbq_index **box = (bbq_index **)calloc(1, sizeof *box);
  const int32_t file_shards = bbq_index_file_shards(path);
  if (devs && ndevs > 0 && file_shards > 1) rc = bbq_index_load_multi(path, (int32_t)ndevs, (const int32_t *)devs, box, (float *)cen);
  else rc = bbq_index_load(path, (int32_t)dev, box, (float *)cen);
  if (rc != BBQ_OK) { free(box); return throw_bbq(env, rc); }
  napi_value ext, o, v;
  if (napi_create_external(env, box, finalize_index, NULL, &ext) != napi_ok) { bbq_index_destroy(*box); free(box); napi_throw_error(env, NULL, "bbq_napi: external"); return NULL; }
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "handle", ext); set_prop(env, o, "centroid", tcen);
  napi_create_double(env, (double)n, &v); set_prop(env, o, "n", v);
  napi_create_double(env, (double)dim, &v); set_prop(env, o, "dim", v);
  napi_create_double(env, (double)sim, &v); set_prop(env, o, "sim", v);
  napi_create_double(env, cdp, &v); set_prop(env, o, "centroidDP", v);
  napi_create_double(env, (double)rb, &v); set_prop(env, o, "rowBase", v);
  napi_create_double(env, (double)file_shards, &v); set_prop(env, o, "shards", v);
  return o;
}

/* indexExport(handle) -> {codes Uint8Array[n*ceil(dim/8)], corr Float64Array[n*4]} */
static napi_value IndexExport(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  bbq_index *ix = unbox(env, a[0]);
  if (!ix) return NULL;
  const size_t dimx = (size_t)bbq_index_dimension(ix);
  const size_t n = (size_t)bbq_index_size(ix), pb = bbq_index_bits(ix) == 1 ? (dimx + 7) / 8 : dimx;
  void *codes, *corr;
  napi_value tcodes = new_typed(env, napi_uint8_array, n * pb, 1, &codes);
  napi_value tcorr = new_typed(env, napi_float64_array, n * 4, 8, &corr);
  if (!tcodes || !tcorr) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int rc = bbq_index_export(ix, (uint8_t *)codes, (double *)corr);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "codes", tcodes); set_prop(env, o, "corr", tcorr);
  return o;
}

/* quantizeQueries(flat Float32Array[n*dim], n, centroid Float32Array[dim], sim, queryBits, lambda, iters, threads)
 *   -> {quantized Uint8Array[n*dim], corrections Float64Array[n*4]}   (bbq_quantize_queries: host threads) */
static napi_value QuantizeQueries(napi_env env, napi_callback_info info) {
  napi_value a[8];
  if (!get_args(env, info, 8, a)) return NULL;
  void *q, *cen; size_t ql, cl;
  int64_t n, sim, qb, iters, threads; double lambda;
  if (!get_typed(env, a[0], napi_float32_array, &q, &ql) || !get_i64(env, a[1], &n) || !get_typed(env, a[2], napi_float32_array, &cen, &cl) ||
      !get_i64(env, a[3], &sim) || !get_i64(env, a[4], &qb) || !get_f64(env, a[5], &lambda) || !get_i64(env, a[6], &iters) ||
      !get_i64(env, a[7], &threads)) return NULL;
  if (n < 0 || cl == 0 || ql != (size_t)n * cl) { napi_throw_error(env, "BBQ6", "查询向量维度与目标向量维度不匹配"); return NULL; }
  void *oq, *oc;
  napi_value tq = new_typed(env, napi_uint8_array, ql, 1, &oq);
  napi_value tc = new_typed(env, napi_float64_array, (size_t)n * 4, 8, &oc);
  if (!tq || !tc) { napi_throw_error(env, NULL, "bbq_napi: allocation failed"); return NULL; }
  int rc = bbq_quantize_queries((const float *)q, (int32_t)n, (int32_t)cl, (const float *)cen, (int32_t)sim, (int32_t)qb, lambda, (int32_t)iters,
                                (int32_t)threads, (uint8_t *)oq, (double *)oc, NULL);
  if (rc != BBQ_OK) return throw_bbq(env, rc);
  napi_value o;
  NAPI_CALL(env, napi_create_object(env, &o));
  set_prop(env, o, "quantized", tq); set_prop(env, o, "corrections", tc);
  return o;
}

static napi_value Init(napi_env env, napi_value exports) {
  napi_property_descriptor d[] = {
      {"deviceCount", NULL, DeviceCount, NULL, NULL, NULL, napi_default, NULL},
      {"quantizeVectors", NULL, QuantizeVectors, NULL, NULL, NULL, napi_default, NULL},
      {"quantizeQuery", NULL, QuantizeQuery, NULL, NULL, NULL, napi_default, NULL},
      {"quantizeQueries", NULL, QuantizeQueries, NULL, NULL, NULL, napi_default, NULL},
      {"centroidDP", NULL, CentroidDP, NULL, NULL, NULL, napi_default, NULL},
      {"indexCreate", NULL, IndexCreate, NULL, NULL, NULL, napi_default, NULL},
      {"indexCreateMulti", NULL, IndexCreateMulti, NULL, NULL, NULL, napi_default, NULL},
      {"indexBuild", NULL, IndexBuild, NULL, NULL, NULL, napi_default, NULL},
      {"indexDestroy", NULL, IndexDestroy, NULL, NULL, NULL, napi_default, NULL},
      {"searchBatch", NULL, SearchBatch, NULL, NULL, NULL, napi_default, NULL},
      {"searchRawBatch", NULL, SearchRawBatch, NULL, NULL, NULL, napi_default, NULL},
      {"searchRawInto", NULL, SearchRawInto, NULL, NULL, NULL, napi_default, NULL},
      {"scoreRows", NULL, ScoreRows, NULL, NULL, NULL, napi_default, NULL},
      {"setOption", NULL, SetOption, NULL, NULL, NULL, napi_default, NULL},
      {"stats", NULL, Stats, NULL, NULL, NULL, napi_default, NULL},
      {"vectorsCreate", NULL, VectorsCreate, NULL, NULL, NULL, napi_default, NULL},
      {"vectorsDestroy", NULL, VectorsDestroy, NULL, NULL, NULL, napi_default, NULL},
      {"rerankScores", NULL, RerankScores, NULL, NULL, NULL, napi_default, NULL},
      {"searchRerankBatch", NULL, SearchRerankBatch, NULL, NULL, NULL, napi_default, NULL},
      {"indexSave", NULL, IndexSave, NULL, NULL, NULL, napi_default, NULL},
      {"indexLoad", NULL, IndexLoad, NULL, NULL, NULL, napi_default, NULL},
      {"indexExport", NULL, IndexExport, NULL, NULL, NULL, napi_default, NULL},
  };
  if (napi_define_properties(env, exports, sizeof d / sizeof d[0], d) != napi_ok) return NULL;
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
