// bbq_build.cpp - quantizeVectors on the device and the index in place (kernels: bbq_build_kernels.hip)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <string>
#include "bbq_host.h"

using namespace bbq;

extern "C" {

int bbq_index_build(const float *vectors, int64_t n, int32_t dim, int32_t sim, double lambda, int32_t iters, int32_t device,
                    bbq_index **out, float *centroid, uint8_t *codes, double *corr, int64_t *bad_row, int32_t *bad_col) {
  return bbq_index_build_bits(vectors, n, dim, sim, 1, lambda, iters, device, out, centroid, codes, corr, bad_row, bad_col);
}

// quantizeVectors on the device + index in place (bbq_build_kernels.hip)
int bbq_index_build_bits(const float *vectors, int64_t n, int32_t dim, int32_t sim, int32_t index_bits, double lambda, int32_t iters,
                         int32_t device, bbq_index **out, float *centroid, uint8_t *codes, double *corr, int64_t *bad_row, int32_t *bad_col) {
  return bbq_index_build_opts(vectors, n, dim, sim, index_bits, lambda, iters, device, nullptr, out, centroid, codes, corr, bad_row, bad_col);
}

int bbq_index_build_opts(const float *vectors, int64_t n, int32_t dim, int32_t sim, int32_t index_bits, double lambda, int32_t iters,
                         int32_t device, const bbq_index_options *opts, bbq_index **out, float *centroid, uint8_t *codes, double *corr,
                         int64_t *bad_row, int32_t *bad_col) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_build: out is null");
  *out = nullptr;
  if (index_bits < 1 || index_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "indexBits必须在1-8之间");
  if (check_options(opts) != BBQ_OK) return BBQ_ERR_INVALID_ARG;
  if (!dim_supported(dim, dim == 1 ? 1 : store_bits_of(index_bits)))
    return fail(BBQ_ERR_UNSUPPORTED, "dimension %d at indexBits %d: the integer dot product would not fit 31 bits", dim, index_bits);
  if (n == 0) return fail(BBQ_ERR_EMPTY, "向量集合不能为空");
  if (n < 0 || dim <= 0 || !vectors || !centroid) return fail(BBQ_ERR_INVALID_ARG, "输入向量不能为空");
  if (sim < 0 || sim > 2) return fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", sim);
  if (iters < 0 || lambda != lambda) return fail(BBQ_ERR_INVALID_ARG, "bad lambda/iters");
  if (n > 0xFFFFFFFFll) return fail(BBQ_ERR_UNSUPPORTED, "more than 2^32 rows");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  if (device < 0 || device >= ndev) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));
  DeviceCtx *ctx = nullptr;
  int rc = get_ctx(device, &ctx);
  if (rc != BBQ_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipStream_t st = ctx->aux_stream;

  const int64_t npad = (n + kTileRows - 1) / kTileRows * kTileRows;
  const int dim4 = (dim + 3) / 4;
  float *d_in = nullptr, *d_vT4 = nullptr, *d_cen = nullptr;
  unsigned long long *d_bad = nullptr;
  double *d_corr = nullptr;
  uint8_t *d_codes = nullptr;
  std::unique_ptr<bbq_index> ix(new bbq_index());
  auto cleanup = [&]() {
    if (d_in) (void)hipFree(d_in);
    if (d_vT4) (void)hipFree(d_vT4);
    if (d_cen) (void)hipFree(d_cen);
    if (d_bad) (void)hipFree(d_bad);
    if (d_corr) (void)hipFree(d_corr);
    if (d_codes) (void)hipFree(d_codes);
  };
#define BCHK(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      cleanup();                                                                                     \
      destroy_unlocked(ix.release());                                                                \
      return fail(BBQ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                      \
    }                                                                                                \
  } while (0)
  BCHK(hipMalloc((void **)&d_in, (size_t)n * dim * 4));
  BCHK(hipMalloc((void **)&d_vT4, (size_t)dim4 * npad * 16));
  BCHK(hipMemcpyAsync(d_in, vectors, (size_t)n * dim * 4, hipMemcpyHostToDevice, st));
  BCHK(launch_build_transpose(d_in, n, dim, npad, d_vT4, st));
  BCHK(hipStreamSynchronize(st));
  BCHK(hipFree(d_in));
  d_in = nullptr;
  if (sim == BBQ_COSINE) BCHK(launch_build_normalize(d_vT4, n, dim, npad, st));  // src/binaryQuantizationFormat.ts:174-176
  // :196-211 NaN / Infinity validation on the processed vectors, first offender in row-major order
  unsigned long long bad = ~0ull;
  BCHK(hipMalloc((void **)&d_bad, 8));
  BCHK(hipMemcpyAsync(d_bad, &bad, 8, hipMemcpyHostToDevice, st));
  BCHK(launch_build_validate(d_vT4, n, dim, npad, d_bad, st));
  BCHK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
  BCHK(hipStreamSynchronize(st));
  if (bad != ~0ull) {
    const int64_t r = (int64_t)(bad / (unsigned long long)dim);
    const int c = (int)(bad % (unsigned long long)dim);
    float v = 0;
    BCHK(hipMemcpy(&v, d_vT4 + ((size_t)(c / 4) * npad + r) * 4 + (c & 3), 4, hipMemcpyDeviceToHost));
    cleanup();
    if (bad_row) *bad_row = r;
    if (bad_col) *bad_col = c;
    if (v != v) return fail(BBQ_ERR_NAN_INPUT, "向量 %lld 位置 %d 包含NaN值", (long long)r, c);
    return fail(BBQ_ERR_INF_INPUT, "向量 %lld 位置 %d 包含Infinity值", (long long)r, c);
  }
  BCHK(hipMalloc((void **)&d_cen, (size_t)dim4 * 16));
  BCHK(launch_build_centroid(d_vT4, n, dim, npad, d_cen, st));  // :214
  BCHK(hipMemcpyAsync(centroid, d_cen, (size_t)dim * 4, hipMemcpyDeviceToHost, st));

  ix->device = device;
  ix->ctx = ctx;
  ix->slots = ctx->slots;
  ix->aux_stream = ctx->aux_stream;
  ix->d_aux_flags = ctx->d_aux_flags;
  ix->dim = dim;
  ix->index_bits = index_bits;
  ix->store_bits = dim == 1 ? 1 : store_bits_of(index_bits);
  ix->pb = row_bytes_of(dim, ix->store_bits);
  ix->w16 = (ix->pb + 15) / 16;
  ix->n_rows = n;
  ix->row_base = 0;
  ix->want_compact = want_compact_of(opts);
  if (index_bits > 1) {
    // more than one bit: the kernel leaves what the reference keeps for such an index - one byte per dimension - and the corrections
    // in device memory; the tile records are built from there exactly as bbq_index_create builds them from host rows
    BCHK(hipMalloc((void **)&d_codes, (size_t)n * dim));
    BCHK(hipMalloc((void **)&d_corr, (size_t)n * 32));
    BCHK(launch_build_quantize_bits(d_vT4, n, dim, npad, d_cen, sim, lambda, iters, index_bits, d_codes, d_corr, st));  // :221-249
    BCHK(hipStreamSynchronize(st));
    BCHK(hipFree(d_vT4));
    d_vT4 = nullptr;
    ix->centroid_dp = bbq_centroid_dp(centroid, dim);
    rc = storage_from_device_rows(ix.get(), ix->main, d_codes, d_corr, n, 0, true);
    if (rc == BBQ_OK && corr && hipMemcpy(corr, d_corr, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BBQ_ERR_HIP, "bbq_index_build: copy of the corrections failed");
    if (rc == BBQ_OK && codes && hipMemcpy(codes, d_codes, (size_t)n * dim, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BBQ_ERR_HIP, "bbq_index_build: copy of the codes failed");
    if (rc == BBQ_OK) rc = ensure_aux_qbuf(ctx, qbuf_bytes_per_query_w(ix->w16));
    cleanup();
    if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
    *out = ix.release();
    return BBQ_OK;
  }
  ix->has_x1 = 0;  // a freshly quantized 1-bit row's component sum IS its popcount
  ix->layout = ix->want_compact ? kLayoutCompact : kLayoutInline;
  ix->tile_stride = tile_stride_of(ix->w16, ix->layout, 0);
  ix->bytes_per_row = ix->tile_stride / kTileRows;
  Storage &sto = ix->main;
  const int64_t n_tiles = npad / kTileRows;
  BCHK(hipMalloc((void **)&sto.d_tiles, (size_t)n_tiles * ix->tile_stride));
  if (ix->layout == kLayoutCompact) BCHK(hipMalloc((void **)&sto.d_exact, (size_t)compact_side_bytes(n_tiles)));
  if (corr) BCHK(hipMalloc((void **)&d_corr, (size_t)n * 32));
  BCHK(launch_build_quantize1(d_vT4, n, dim, npad, d_cen, sim, lambda, iters, sto.d_tiles, sto.d_exact, d_corr, ix->w16, ix->tile_stride,
                              ix->layout, st));  // :221-249
  if (ix->layout == kLayoutCompact) BCHK(launch_tile_add_range(sto.d_exact, n, const_cast<float *>(add_range_of(sto.d_exact, n_tiles)), st));
  if (corr) BCHK(hipMemcpyAsync(corr, d_corr, (size_t)n * 32, hipMemcpyDeviceToHost, st));
  if (codes) {
    BCHK(hipMalloc((void **)&d_codes, (size_t)n * ix->pb));
    BCHK(launch_build_untile(sto.d_tiles, n, ix->pb, ix->w16, ix->tile_stride, d_codes, st));
    BCHK(hipMemcpyAsync(codes, d_codes, (size_t)n * ix->pb, hipMemcpyDeviceToHost, st));
  }
  BCHK(hipStreamSynchronize(st));
#undef BCHK
  cleanup();
  sto.row_id_base = 0;
  sto.view.tiles = sto.d_tiles;
  sto.view.exact = sto.d_exact;
  sto.view.add_range = add_range_of(sto.d_exact, n_tiles);
  sto.view.n_rows = n;
  sto.view.w16 = ix->w16;
  sto.view.tile_stride = ix->tile_stride;
  sto.view.has_x1 = 0;
  sto.view.dim = dim;
  sto.view.layout = ix->layout;
  sto.view.store_bits = 1;
  ix->centroid_dp = bbq_centroid_dp(centroid, dim);  // getCentroidDP(undefined), :113-121
  rc = ensure_aux_qbuf(ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
  *out = ix.release();
  return BBQ_OK;
}

}  // extern "C"
