/*
 * bbq_oracle.h - CPU restatement of the reference's asymmetric binary-quantized
 * scoring + top-k search path (leolee9086/Better-Binary-Quantization, TypeScript).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call it, and only as the checker / the reported CPU
 * baseline.  The product library (better-binary-quantization_amd/csrc) never includes,
 * links or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against golden
 * vectors produced by running the type-erased reference itself under Node 12 in the
 * build container (oracle/tools/erase_ts.py + oracle/tools/gen_fixtures.js ->
 * the JSON files under tests/golden; tests/test_oracle_golden.py).
 *
 * Number model: all arithmetic IEEE binary64, compiled with
 * -ffp-contract=off (no FMA), f32 rounding exactly where the reference stores into a
 * Float32Array.  Each function cites the reference file:line it follows
 * (paths relative to /root/reference/).
 */
#ifndef BBQ_ORACLE_H
#define BBQ_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/types.ts:9-13 (string enum in the reference; ordinal here) */
enum { ORC_EUCLIDEAN = 0, ORC_COSINE = 1, ORC_MIP = 2 };

/* corrections layout, src/types.ts:18-27: {lowerInterval, upperInterval, additionalCorrection, quantizedComponentSum} */

/* src/vectorOperations.ts:11-34 */
void orc_normalize(const float *v, int dim, float *out);
/* src/vectorOperations.ts:126-163 (f32 accumulator, rounded after every += and the final /=) */
void orc_centroid(const float *base, int64_t n, int dim, float *centroid);
/* src/vectorOperations.ts:171-185 */
double orc_dot_f32(const float *a, const float *b, int dim);
/* src/vectorSimilarity.ts:73-101 */
double orc_cosine_similarity(const float *a, const float *b, int dim);

/* src/optimizedScalarQuantizer.ts:108-227 (+ :245-265, :280-353, :373-407).  dest: one value per dim. */
void orc_scalar_quantize(const float *vec, int dim, int bits, const float *centroid, int sim,
                         double lambda, int iters, uint8_t *dest, double corr[4]);
/* src/optimizedScalarQuantizer.ts:420-446.  returns 0, or -1 if a value is not 0/1 (the reference throws). */
int orc_pack_binary(const uint8_t *bits, int dim, uint8_t *packed);

/* src/binaryQuantizationFormat.ts:165-263 (indexBits == 1): normalise (COSINE), centroid, quantize, pack. */
void orc_build_index(const float *base, int64_t n, int dim, int sim, double lambda, int iters,
                     uint8_t *codes /* n*ceil(dim/8) */, double *corr /* n*4 */, float *centroid /* dim */);
/* same for indexBits > 1: codes are unpacked, one byte per dim (binaryQuantizationFormat.ts:241-245) */
void orc_build_index_unpacked(const float *base, int64_t n, int dim, int sim, int index_bits, double lambda, int iters,
                              uint8_t *codes /* n*dim */, double *corr, float *centroid);

/* query side of src/binaryQuantizationFormat.ts:337-347 + :271-299 (COSINE normalises twice, A.5-1) */
void orc_quantize_query(const float *query, int dim, const float *centroid, int sim, int qb,
                        double lambda, int iters, uint8_t *qquant /* dim */, double qcorr[4]);

/* src/utils/computeBatchFourBitDotProductDirectPacked.ts:10-53 (any qb != 1) */
int32_t orc_qcdist_unpacked_query(const uint8_t *q, const uint8_t *row_packed, int dim);
/* src/batchDotProduct.ts:22-49 + src/utils/bitcount.ts:7-15 (qb == 1; both packed) */
int32_t orc_qcdist_packed_query(const uint8_t *q_packed, const uint8_t *row_packed, int packed_bytes);
/* src/bitwiseDotProduct.ts:14-30 (semantic definition for any qb/ib, unpacked bytes) */
int32_t orc_dot_u8(const uint8_t *q, const uint8_t *d, int dim);

/* src/batchDotProduct.ts:478-541 (one_bit != 0) / :554-617 (one_bit == 0); SURVEY App. A.4 */
double orc_score(int32_t qcdist, const double qcorr[4], const double xcorr[4], int dim, double centroid_dp,
                 int sim, int one_bit);

/* src/binaryQuantizedScorer.ts:315-400 over all rows (batching is irrelevant to the values) */
void orc_score_all(const uint8_t *codes, const double *corr, int64_t n, int dim,
                   const uint8_t *qquant, const double qcorr[4], int qb, int sim, double centroid_dp,
                   int32_t *qcdist /* may be NULL */, double *score64 /* may be NULL */, float *score32 /* may be NULL */);

/* indexBits > 1 (rows unpacked, one byte per dim): the per-row scorer's formulas (src/binaryQuantizedScorer.ts:108-214) ... */
double orc_score_single_row(int32_t qcdist, const double qcorr[4], const double xcorr[4], int dim, double centroid_dp,
                            int sim, int one_bit);
/* ... and computeBatchQuantizedScores as it behaves on such an index (batch path throws -> per-row fallback, :403-419;
 * centroidDP = 0 for 4-bit queries, :290).  Returns -1 where the reference throws (queryBits other than 1 and 4). */
int orc_score_all_multibit(const uint8_t *codes_unpacked, const double *corr, int64_t n, int dim,
                           const uint8_t *qquant, const double qcorr[4], int qb, int sim, double centroid_dp,
                           int32_t *qcdist, double *score64, float *score32);
/* NOT a reference behaviour (it throws): the per-row 4-bit form applied to ANY queryBits on a multi-bit index - the definition of
 * libbbq's documented extension ("parity unpinned" beyond the integer dot) and the CPU baseline of BASELINE config 5 */
void orc_score_all_multibit_ext(const uint8_t *codes_unpacked, const double *corr, int64_t n, int dim,
                                const uint8_t *qquant, const double qcorr[4], int qb, int sim, double centroid_dp,
                                int32_t *qcdist, double *score64, float *score32);
/* orc_search on an indexBits > 1 index (codes unpacked); -5 where the reference throws on the queryBits */
int64_t orc_search_multibit(const float *query, int query_dim, const uint8_t *codes_unpacked, const double *corr,
                            const float *centroid, int64_t n, int dim, int sim, int qb, double lambda, int iters, int64_t k,
                            int32_t *out_idx, float *out_score);

/* src/binaryQuantizationFormat.ts:383-411 with src/minHeap.ts:9-130 restated exactly.
 * returns the number of results written (min(k, n)). */
int64_t orc_heap_topk(const float *scores, int64_t n, int64_t k, int32_t *out_idx, float *out_score);

/* src/binaryQuantizationFormat.ts:308-412 from a float query.  returns count, or <0 on the reference's throw
 * conditions (-1 null query, -2 null index, -3 k<0, -4 dim mismatch) */
int64_t orc_search(const float *query, int query_dim, const uint8_t *codes, const double *corr, const float *centroid,
                   int64_t n, int dim, int sim, int qb, double lambda, int iters, int64_t k,
                   int32_t *out_idx, float *out_score);

/* src/topKSelector.ts:29-79 (oversample, exact cosine rerank, heap, final sort) -> indices */
int64_t orc_oversampled_topk(const float *query, const float *base, const uint8_t *codes, const double *corr,
                             const float *centroid, int64_t n, int dim, int sim, int qb, double lambda, int iters,
                             int64_t k, int factor, int32_t *out_idx);

/* src/vectorSimilarity.ts:14-126 (computeSimilarity: the exact f64 score the rerank recipe uses) */
double orc_true_similarity(const float *a, const float *b, int dim, int sim);
/* src/topKSelector.ts:40-78 / :102-115 from the candidates' true scores, candidate order = the oversampled search's
 * result order.  out_pos = positions into that list, in result order; returns the count. */
int64_t orc_rerank_select_heap(const double *true_scores, int64_t cnt, int64_t k, int32_t *out_pos);
int64_t orc_rerank_select_sort(const double *true_scores, int64_t cnt, int64_t k, int32_t *out_pos);

/* SURVEY 8(d) synthetic input generator: mulberry32(seed), value f32(2u-1), row-major fill */
void orc_mulberry32_fill(uint32_t seed, float *out, int64_t count);

#ifdef __cplusplus
}
#endif
#endif
