// bbq_launch.h - host-callable launch wrappers of the kernels in bbq_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "bbq_device.h"

namespace bbq {

// planes: number of query bit-planes (1, 2, 4 or 8); multi-bit index: 4 (query values <= 15) or 8
hipError_t launch_scan(const ScanArgs &a, int planes, bool dense, int n_queries, int n_chunks, hipStream_t s);
// shared sweep: `share` (4 or 8) queries per workgroup reuse every loaded row (sparse segments, fixed-width dims only)
bool shared_sweep_supported(const ScanArgs &a, int share);
hipError_t launch_scan_shared(const ScanArgs &a, int planes, int share, int n_queries, int n_chunks, hipStream_t s);
// shared sweep on the matrix cores: 32 queries per workgroup (bbq_mfma_kernels.hip); query values must be <= 127 (fp_scale > 0: <= 15,
// staged as FP6 times fp_scale against the bit weights for v_mfma_f32_32x32x64_f8f6f4; 0: int8 for v_mfma_i32_32x32x32_i8), every
// query's interval width positive and finite
bool mfma_sweep_supported(const ScanArgs &a);
int64_t mfma_query_bytes_per_group(int w16, bool fp);  // staged operand bytes of 32 queries
int mfma_queries_per_tile_load(const ScanArgs &a, int n_queries, bool fp);  // 32, or 64 where a workgroup serves two groups per tile
hipError_t launch_scan_mfma(const ScanArgs &a, const uint8_t *qbytes, const float *qmax, float fp_scale, int n_queries, int n_chunks, hipStream_t s);
hipError_t launch_finalize(const FinalizeArgs &a, int n_queries, hipStream_t s);
// lists are [nq][list_stride]; a query may hold more than advertised_cap entries (a flood): such queries are only dropped
// (flagged) when the packed buffer cannot take the sum
hipError_t launch_pack(const int32_t *counts, const uint64_t *lists, int64_t list_stride, int64_t advertised_cap, int32_t nq, int64_t *offsets,
                       int32_t *flags_out, int64_t *total_out, uint64_t *packed, int64_t packed_cap, hipStream_t s);
hipError_t launch_retile(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t pb, uint8_t *tiles, int32_t w16,
                         int32_t tile_stride, int32_t has_x1, int32_t layout, double *exact, hipStream_t s);
// multi-bit index (store_bits 2 / 4 / 8): codes are unpacked rows [n][dim]; *bad is raised by a code that is not below 2^index_bits
hipError_t launch_retile_multibit(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t store_bits, int32_t index_bits, uint8_t *tiles,
                                  int32_t w16, int32_t tile_stride, int32_t has_x1, int32_t layout, double *exact, uint32_t *bad, hipStream_t s);
hipError_t launch_check_x1_multibit(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, uint32_t *mismatch, hipStream_t s);
// compact layout: each tile's {min, max} of additionalCorrection (read from exact[]) -> add_range[tile][2]
hipError_t launch_tile_add_range(const double *exact, int64_t n_rows, float *add_range, hipStream_t s);
hipError_t launch_check_x1(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t pb, uint32_t *mismatch,
                           hipStream_t s);

// latency path (bbq_latency_kernels.hip): false when this index shape has no instantiation (the caller takes the general path)
bool latency_path_supported(const IndexView &v, int planes);
hipError_t launch_lat_scan(const LatScanArgs &a, int planes, hipStream_t s);
// pre-sampled threshold: per-wave top keys of the first rows, then theta := the rank-th largest of them (0 when there are fewer)
hipError_t launch_lat_pre(const LatPreArgs &a, int planes, hipStream_t s);
hipError_t launch_lat_select(const uint32_t *pre_keys, int n_keys, int rank, uint32_t *theta, hipStream_t s);

// index build on the device (bbq_build_kernels.hip); vT4 is the [ceil(dim/4)][npad] float4 transposed copy
hipError_t launch_build_transpose(const float *in, int64_t n, int32_t dim, int64_t npad, float *vT4, hipStream_t s);
hipError_t launch_build_normalize(float *vT4, int64_t n, int32_t dim, int64_t npad, hipStream_t s);
hipError_t launch_build_validate(const float *vT4, int64_t n, int32_t dim, int64_t npad, unsigned long long *first_bad, hipStream_t s);
hipError_t launch_build_centroid(const float *vT4, int64_t n, int32_t dim, int64_t npad, float *centroid, hipStream_t s);
hipError_t launch_build_quantize1(const float *vT4, int64_t n, int32_t dim, int64_t npad, const float *centroid, int32_t sim,
                                  double lambda, int32_t iters, uint8_t *tiles, double *exact, double *corr_rm, int32_t w16,
                                  int32_t tile_stride, int32_t layout, hipStream_t s);
// indexBits > 1: unpacked codes [n][dim] (one byte per dimension) + row-major corrections [n][4], both in device memory
hipError_t launch_build_quantize_bits(const float *vT4, int64_t n, int32_t dim, int64_t npad, const float *centroid, int32_t sim,
                                      double lambda, int32_t iters, int32_t bits, uint8_t *codes_rm, double *corr_rm, hipStream_t s);
hipError_t launch_build_untile(const uint8_t *tiles, int64_t n, int32_t pb, int32_t w16, int32_t tile_stride, uint8_t *codes_rm,
                               hipStream_t s);

// exact rerank (bbq_rerank_kernels.hip): one wave per 64 candidates of a query; max_count = longest candidate list
hipError_t launch_rerank(const RerankArgs &a, int n_queries, int64_t max_count, hipStream_t s);

}  // namespace bbq
