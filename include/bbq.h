/*
 * bbq.h - C ABI of libbbq: the MI355X-native (gfx950 HIP) implementation of the
 * asymmetric binary-quantized scoring + top-k search path of
 * leolee9086/Better-Binary-Quantization.
 *
 * The reference has no FFI seam on this path (it is pure TypeScript); the boundary is the
 * internal call searchNearestNeighbors -> computeBatchQuantizedScores -> MinHeap.  Each entry
 * point below names the reference interface it replaces (paths relative to the reference
 * checkout).  The N-API addon (better-binary-quantization_amd/napi/bbq_napi.c) and the
 * ctypes binding (better-binary-quantization_amd/python/bbq_amd/capi.py) bind exactly these
 * symbols; INTEGRATION.md shows the reference-side patch.
 *
 * Conventions: plain pointers and sizes, no framework types.  All input arrays are borrowed for
 * the duration of the call and never retained or written.  Every function returns BBQ_OK (0) or
 * a BBQ_ERR_* code; bbq_last_error() returns a thread-local, human-readable message for the last
 * failure on the calling thread.  There is NO CPU fallback: without a usable HIP device every
 * device entry point fails with BBQ_ERR_NO_DEVICE.
 *
 * Corrections layout (src/types.ts:18-27), 4 doubles per vector:
 *   {lowerInterval, upperInterval, additionalCorrection, quantizedComponentSum}.
 * Packed 1-bit rows (src/optimizedScalarQuantizer.ts:420-446): dim d -> byte d>>3, bit 7-(d&7),
 *   ceil(dim/8) bytes per row.
 */
#ifndef BBQ_H
#define BBQ_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBQ_ABI_VERSION 3  /* 2: multi-bit and multi-device indexes, bbq_stats.host_replays
                              3: creation options (corrections layout), asynchronous shard scans with shard-local answers,
                                 bbq_merge_answers, persistence of shards and multi-device indexes */

/* status codes */
enum {
  BBQ_OK = 0,
  BBQ_ERR_INVALID_ARG = 1,   /* null pointer, bad size, bits out of 1..8, ... */
  BBQ_ERR_NO_DEVICE = 2,     /* no HIP device / runtime unusable: the product path refuses to run */
  BBQ_ERR_HIP = 3,           /* a HIP call failed; message carries hipGetErrorString */
  BBQ_ERR_OOM = 4,
  BBQ_ERR_UNSUPPORTED = 5,   /* e.g. more than 2^32 rows, k out of the sharded range, a file of another format version */
  BBQ_ERR_DIM_MISMATCH = 6,  /* src/binaryQuantizationFormat.ts:327-329 */
  BBQ_ERR_NEGATIVE_K = 7,    /* src/binaryQuantizationFormat.ts:324-326 */
  BBQ_ERR_NAN_INPUT = 8,     /* src/binaryQuantizationFormat.ts:202-204 */
  BBQ_ERR_INF_INPUT = 9,     /* src/binaryQuantizationFormat.ts:205-207 */
  BBQ_ERR_EMPTY = 10         /* src/binaryQuantizationFormat.ts:169-171 */
};

/* src/types.ts:9-13 VectorSimilarityFunction (string enum in the reference) */
enum { BBQ_EUCLIDEAN = 0, BBQ_COSINE = 1, BBQ_MAXIMUM_INNER_PRODUCT = 2 };

typedef struct bbq_index bbq_index; /* opaque: a device-resident index shard */

/* one top-k candidate as the device emits it: (global row << 32) | IEEE-754 bits of the f32 score */
typedef uint64_t bbq_cand;

const char *bbq_last_error(void);
int bbq_abi_version(void);
/* number of usable HIP devices (0 if none); never fails */
int bbq_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Index: replaces BinarizedByteVectorValuesImpl (src/binaryQuantizationFormat.ts:24-126) as the
 * scorer sees it, and createDirectPackedBuffer (src/batchDotProduct.ts:420-436): the packed rows
 * are staged ONCE, re-tiled into the device layout (DESIGN.md "HBM layout"), instead of being
 * gathered into a contiguous buffer for every batch of 1000.
 *
 *   codes        index_bits == 1: [n_rows * ceil(dim/8)] packed 1-bit rows
 *                index_bits  > 1: [n_rows * dim] one byte per dimension, values < 2^index_bits - the shape quantizeVectors
 *                gives such rows (src/binaryQuantizationFormat.ts:241-245).  Stored as 2-bit (index_bits 2), 4-bit (3-4) or
 *                8-bit (5-8) fields and scanned with the packed-nibble / packed-byte dot instructions.  Scores follow what
 *                the reference RETURNS for such an index: its batch scorer throws on unpacked rows and the per-row scorer
 *                answers (src/binaryQuantizedScorer.ts:403-419, :69-301): bitDotProduct = computeQuantizedDotProduct
 *                (src/bitwiseDotProduct.ts:14-30), centroidDP = 0 unless query_bits == 1 (:290), MAXIMUM_INNER_PRODUCT without the
 *                FOUR_BIT_SCALE division (:207-209).  The reference's per-row scorer only knows query_bits 1 and 4 and throws
 *                for the rest (:95-97); this library scores every query_bits 2..8 with the 4-bit form ("parity unpinned"
 *                beyond the integer dot product, which the reference defines for any widths) - the JS host throws like the
 *                reference does.
 *   corr         [n_rows * 4]            corrections as doubles
 *   centroid_dp  getCentroidDP(undefined) = centroid . centroid  (src/binaryQuantizationFormat.ts:113-121)
 *   device       HIP device ordinal
 */
int bbq_index_create(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim,
                     int32_t index_bits, double centroid_dp, int32_t device, bbq_index **out);

/* Creation options (ABI 3): every creator has an _opts twin that takes them; NULL = the defaults, which is what the plain
 * creators pass.  The struct may grow: set `size` to sizeof(bbq_index_options).
 *   corrections   how a row's corrections travel with its codes in HBM (DESIGN.md "HBM layout"):
 *                 BBQ_CORRECTIONS_COMPACT  4 B per row streamed (bf16 lower | bf16 upper) + the exact f64 values in a side array that is
 *                                          read only for rows whose proven score bound beats the threshold (100 B/row at 768-d);
 *                                          falls back to inline when quantizedComponentSum is not the row's popcount / code sum
 *                 BBQ_CORRECTIONS_INLINE   the exact f64 corrections inside the tile record (120 / 128 B/row at 768-d)
 *                 BBQ_CORRECTIONS_DEFAULT  compact, unless the environment variable BBQ_COMPACT_CORRECTIONS=0 overrides it
 *                                          (kept for A/B runs of unchanged callers; an explicit value here always wins)
 * Results never depend on the layout (every fixture runs in both). */
enum { BBQ_CORRECTIONS_DEFAULT = -1, BBQ_CORRECTIONS_INLINE = 0, BBQ_CORRECTIONS_COMPACT = 1 };
typedef struct {
  int32_t size;         /* sizeof(bbq_index_options) */
  int32_t corrections;  /* BBQ_CORRECTIONS_* */
} bbq_index_options;

/* quantizeVectors ON THE DEVICE and the index built in place: BinaryQuantizationFormat.quantizeVectors
 * (src/binaryQuantizationFormat.ts:165-263: normalizeVector for COSINE, NaN/Infinity validation, computeCentroid,
 * scalarQuantize, packAsBinary) for indexBits == 1 (bbq_index_build_bits: any indexBits), bit-exact with the TypeScript path,
 * followed by what bbq_index_create does.  vectors [n*dim] row-major host floats.
 *   centroid [dim]            (host, required)  the f32 centroid; centroid_dp is derived from it
 *   codes [n*ceil(dim/8)], corr [n*4]  (host, each may be NULL) what quantizeVectors returns, for hosts that keep
 *                             the BinarizedByteVectorValues accessors (vectorValue / getCorrectiveTerms)
 * Errors as bbq_quantize_vectors. */
int bbq_index_build(const float *vectors, int64_t n, int32_t dim, int32_t sim, double lambda, int32_t iters,
                    int32_t device, bbq_index **out, float *centroid, uint8_t *codes, double *corr,
                    int64_t *bad_row, int32_t *bad_col);

/* The same for any indexBits (1..8): codes [n*ceil(dim/8)] for index_bits == 1, [n*dim] (one byte per dimension,
 * src/binaryQuantizationFormat.ts:241-245) otherwise. */
int bbq_index_build_bits(const float *vectors, int64_t n, int32_t dim, int32_t sim, int32_t index_bits, double lambda, int32_t iters,
                         int32_t device, bbq_index **out, float *centroid, uint8_t *codes, double *corr,
                         int64_t *bad_row, int32_t *bad_col);
int bbq_index_build_opts(const float *vectors, int64_t n, int32_t dim, int32_t sim, int32_t index_bits, double lambda, int32_t iters,
                         int32_t device, const bbq_index_options *opts, bbq_index **out, float *centroid, uint8_t *codes, double *corr,
                         int64_t *bad_row, int32_t *bad_col);

/* Row-sharded variant for one-process-per-GPU deployments (new; the reference has no distribution).
 *   row_base     global row id of this shard's row 0 (shards are contiguous, ascending)
 *   pilot_codes / pilot_corr / n_pilot
 *                optional replica of n_pilot DISTINCT global rows that all precede this shard (global id <
 *                row_base; n_pilot a multiple of 1024, or all row_base of them): lets every shard derive valid
 *                top-k thresholds without waiting for the shards before it.  The prefix [0, n_pilot) works; an
 *                evenly strided sample of [0, row_base) gives tighter thresholds when rows are stored cluster by
 *                cluster.  Only thresholds are taken from these rows (their ids never appear in a result).
 *                Pass NULL/0 on the shard that owns row 0 (row_base == 0).
 */
int bbq_index_create_shard(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim,
                           int32_t index_bits, double centroid_dp, int64_t row_base,
                           const uint8_t *pilot_codes, const double *pilot_corr, int64_t n_pilot,
                           int32_t device, bbq_index **out);
int bbq_index_create_shard_opts(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim,
                                int32_t index_bits, double centroid_dp, int64_t row_base,
                                const uint8_t *pilot_codes, const double *pilot_corr, int64_t n_pilot,
                                int32_t device, const bbq_index_options *opts, bbq_index **out);
/* ONE index row-sharded over several GPUs of this process, behind the same handle (new; SURVEY 8b/8e: the reference has no
 * distribution).  Every entry point that takes a bbq_index - bbq_search, bbq_search_batch, bbq_score_rows, bbq_index_export,
 * bbq_search_rerank_batch, bbq_set_option, bbq_get_stats - works on it and returns exactly what the single-device index
 * returns: shards are contiguous row blocks (whole 512-row chunks) in ascending order, each later shard carries a pilot
 * replica of the first pilot_rows global rows (0: none, thresholds start inside the shard), one host thread per shard
 * sweeps it and lands its packed candidate list in pinned host memory over that device's own PCIe link, the calling thread
 * replays the reference heap over the lists in shard order (bbq_replay_batch) while the next round of queries is swept.
 *   n_shards     1..64
 *   devices      [n_shards] HIP device ordinal of every shard (NULL: shard s on device s); a device may appear more than once
 *                (several shards on one GPU: how a single-GPU box tests the path)
 * bbq_shard_scan refuses such a handle.  Extra option: round_queries 1..65536 (512), queries per round. */
int bbq_index_create_multi(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits,
                           double centroid_dp, int32_t n_shards, const int32_t *devices, int64_t pilot_rows, bbq_index **out);
int bbq_index_create_multi_opts(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits,
                                double centroid_dp, int32_t n_shards, const int32_t *devices, int64_t pilot_rows,
                                const bbq_index_options *opts, bbq_index **out);
int32_t bbq_index_shards(const bbq_index *idx);    /* 1 for a single-device index */
void bbq_index_destroy(bbq_index *idx);
int64_t bbq_index_size(const bbq_index *idx);      /* BinarizedByteVectorValues.size()      src/types.ts:46 */
int32_t bbq_index_dimension(const bbq_index *idx); /* BinarizedByteVectorValues.dimension() src/types.ts:34 */
int32_t bbq_index_bits(const bbq_index *idx);      /* indexBits the index was created with (1..8) */
/* bytes of HBM one query sweep reads per row: the figure bench.py prices the roofline with */
int32_t bbq_index_bytes_per_row(const bbq_index *idx);

/* ------------------------------------------------------------------------------------------
 * Search: replaces steps 2-4 of BinaryQuantizationFormat.searchNearestNeighbors
 * (src/binaryQuantizationFormat.ts:349-411): score every row
 * (BinaryQuantizedScorer.computeBatchQuantizedScores, src/binaryQuantizedScorer.ts:315-420;
 * computeBatchFourBitDotProductDirectPacked / computeBatchDotProductDirectPacked;
 * computeBatch{FourBit,OneBit}SimilarityScores src/batchDotProduct.ts:478-617), round to f32,
 * MinHeap top-k (src/minHeap.ts), descending output.  Bit-exact incl. the order among equal scores.
 *
 *   qquant   [dim]   quantized query, one value per dimension (Uint8Array from quantizeQueryVector)
 *   qcorr    [4]     query corrections
 *   query_bits       1 selects the 1-bit formulas, anything else the "4-bit" formulas
 *   out_idx/out_score [k]  (only min(k, size) entries are written); *out_n = number written
 * k == 0 -> *out_n = 0.  k < 0 -> BBQ_ERR_NEGATIVE_K.  k > 4096 is answered by the dense path (every f32 score to the
 * host; 16 ms per 10 M-row query instead of 0.15-4 ms).
 * Multi-bit index (index_bits > 1) with query_bits other than 1 and 4: the reference throws (src/binaryQuantizedScorer.ts:95-97); this
 * entry point scores it with the per-row 4-bit form - integer dot product pinned by fixtures, float score PARITY UNPINNED (the JS and
 * Python hosts throw like the reference).
 */
int bbq_search(bbq_index *idx, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim,
               int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n);

/* The same for n_queries independent queries (each does its own sweep of the index; the calls are
 * pipelined on the device).  qquant [n_queries*dim], qcorr [n_queries*4], out_idx/out_score
 * [n_queries*k] (row q at offset q*k), out_n [n_queries]. */
int bbq_search_batch(bbq_index *idx, int32_t n_queries, const uint8_t *qquant, const double *qcorr,
                     int32_t query_bits, int32_t sim, int64_t k, int32_t *out_idx, float *out_score,
                     int64_t *out_n);

/* searchNearestNeighbors from the RAW query (src/binaryQuantizationFormat.ts:337-411) for n_queries queries: normalisation (COSINE) +
 * quantizeQueryVector exactly as bbq_quantize_query does them, then bbq_search_batch - but pipelined: the queries are quantized on
 * n_threads host threads (0: half the cores, at most 16) chunk by chunk while the sub-batches in front are already on the device, so
 * the quantizer (~15 us per 768-d query and core) only stands in front of the first sweep.  Results are bit-identical to
 * bbq_quantize_queries followed by bbq_search_batch.
 *   queries [n_queries*dim] raw fp32, centroid [dim] (getCentroid()), lambda / iters: the quantizer's (0.1 / 5 by default in the reference)
 *   qquant_out [n_queries*dim], qcorr_out [n_queries*4]: optional, the quantized queries (what quantizeQueryVector returns)
 *   *bad_query (may be NULL): the first query the quantizer refused (NaN / Infinity input, ...), with its error as the return value */
int bbq_search_raw_batch(bbq_index *idx, int32_t n_queries, const float *queries, const float *centroid, int32_t sim,
                         int32_t query_bits, double lambda, int32_t iters, int32_t n_threads, int64_t k, int32_t *out_idx,
                         float *out_score, int64_t *out_n, uint8_t *qquant_out, double *qcorr_out, int32_t *bad_query);

/* Per-row results of computeBatchQuantizedScores (src/binaryQuantizedScorer.ts:389-400) for the
 * contiguous ords [row_begin, row_begin+row_count): bitDotProduct (integer qcDist), the f64 score and
 * its f32 rounding (src/binaryQuantizationFormat.ts:353,378).  Any output pointer may be NULL. */
int bbq_score_rows(bbq_index *idx, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim,
                   int64_t row_begin, int64_t row_count, int32_t *out_qcdist, double *out_score64,
                   float *out_score32);

/* ------------------------------------------------------------------------------------------
 * Sharded search (one process per GPU).  bbq_shard_scan sweeps THIS shard for n_queries queries and
 * leaves the shard's candidates - per query a superset of the rows that ever enter the reference's heap,
 * ascending by global row - PACKED in caller-provided DEVICE memory, so the host framework can move them
 * over RCCL; bbq_replay_batch then replays the reference heap
 * (src/binaryQuantizationFormat.ts:383-411) over all shards' lists in shard order.
 *
 *   dev_packed   device pointer, [packed_cap] bbq_cand; query q's entries at [offsets[q], offsets[q+1])
 *   dev_offsets  device pointer, [n_queries + 1] int64
 *   dev_flags    device pointer, [n_queries] int32; non-zero: this shard could not bound the query
 *                (NaN score / more candidates than every buffer holds) - it carries no entries and the query must be
 *                scored densely
 *   *out_total   (host) number of entries written = offsets[n_queries]
 *   k            1..4096
 * The call returns after the device work has completed (buffers are ready for a collective).
 * A query may leave more than bbq_shard_list_cap entries (a flood: rows stored cluster by cluster); as long as the sum over
 * the batch fits packed_cap nothing is dropped, otherwise the flooded queries are flagged.  BBQ_ERR_OOM only if packed_cap is
 * smaller than n_queries x bbq_shard_list_cap and the entries still do not fit.
 */
int bbq_shard_scan(bbq_index *idx, int32_t n_queries, const uint8_t *qquant, const double *qcorr,
                   int32_t query_bits, int32_t sim, int64_t k, void *dev_packed, int64_t packed_cap,
                   void *dev_offsets, void *dev_flags, int64_t *out_total);
/* The same, asynchronous, with SHARD-LOCAL ANSWERS (ABI 3).  bbq_shard_scan_begin enqueues the whole sweep of the batch - and, behind
 * it on the device, the packing - and returns without waiting; the queries are copied before it returns.  A second begin may follow
 * before the first has been waited for (two batches in flight per index: the device never drains between batches); waits are taken in
 * begin order.  bbq_shard_scan_wait blocks until everything the batch wrote is ready and reports the packed total.
 * bbq_shard_scan == begin + wait without answers.
 *
 *   dev_answers     device pointer (NULL: lists only; must be NULL for k > 1024), [n_queries][answers_stride] uint64, answers_stride >= k + 3.
 *                   Per query:
 *       [0]  entries in the packed list | flags << 32
 *       [1]  m = number of answer entries | unproven << 32   (unproven != 0: flags, or more keys than the launch can select among - take the packed list)
 *       [2]  cut: the monotone key (bbq_key_of_score) of the (k+1)-th largest f32 score among ALL rows this shard has seen - its own
 *            and its pilot replica's -, 0 when it has seen at most k rows
 *       [3 .. 3+m)  the shard's OWN rows whose score key is above the cut, (row << 32 | f32 bits), descending by score (m <= k)
 *   The cut never exceeds the key of the global (k+1)-th largest score (it is an order statistic of a subset of the rows), so the
 *   union of all shards' answer entries contains every row above max(cut): bbq_merge_answers selects the global answer from them and
 *   proves it (DESIGN.md "Exact top-k": when no two of the k+1 largest scores compare equal the reference heap returns exactly the
 *   k best rows in descending order, whatever its history); only for a query it cannot prove are the packed lists needed (heap replay).
 *   What travels per shard and query is (k + 3) x 8 bytes instead of the ~1.1 K-entry list.
 */
int bbq_shard_scan_begin(bbq_index *idx, int32_t n_queries, const uint8_t *qquant, const double *qcorr,
                         int32_t query_bits, int32_t sim, int64_t k, void *dev_packed, int64_t packed_cap,
                         void *dev_offsets, void *dev_flags, void *dev_answers, int64_t answers_stride);
int bbq_shard_scan_wait(bbq_index *idx, int64_t *out_total);
/* Host-only: the global answers from the shards' answer blocks.  answers[s] = source s's block ([n_queries][strides[s]] uint64 as above,
 * in HOST memory), sources in any order.  n_total = global number of rows; out_idx / out_score [n_queries * k]; out_n [n_queries].
 * status[q]: 0 = answered (bit-identical to the reference's heap), 1 = not provable from the answers (equal scores in or at the edge of
 * the answer, an unproven source): replay the heap over the packed lists (bbq_replay_batch), 2 = a source flagged the query: it needs
 * the dense path.  out_* of queries with status != 0 are left untouched. */
int bbq_merge_answers(int32_t n_sources, const uint64_t *const *answers, const int64_t *strides, int32_t n_queries, int64_t n_total,
                      int64_t k, int32_t n_threads, int32_t *out_idx, float *out_score, int64_t *out_n, uint8_t *status);
/* the monotone key of an f32 score: larger score <=> larger key (NaN excluded); what cuts and thresholds are expressed in */
uint32_t bbq_key_of_score(float score);
/* planned entries ONE query leaves on this shard, with a 4x margin (use n_queries x this for packed_cap) */
int64_t bbq_shard_list_cap(const bbq_index *idx, int64_t k);

/* Host-only (no device needed): exact replay of the reference heap over candidate lists.
 *   lists[i] / counts[i]  i = 0..n_lists-1, in ascending shard order; entries ascending by row
 *   n_total               global number of rows (k2 = min(k, n_total))
 */
int bbq_replay(int32_t n_lists, const bbq_cand *const *lists, const int64_t *counts, int64_t n_total, int64_t k,
               int32_t *out_idx, float *out_score, int64_t *out_n);
/* The same for n_queries queries at once, n_threads host threads: source s (shard s, ascending shard order)
 * contributes packed[s][offsets[s][q] .. offsets[s][q+1]) to query q.  out_idx/out_score [n_queries*k]. */
int bbq_replay_batch(int32_t n_sources, const bbq_cand *const *packed, const int64_t *const *offsets,
                     int32_t n_queries, int64_t n_total, int64_t k, int32_t n_threads, int32_t *out_idx,
                     float *out_score, int64_t *out_n);

/* ------------------------------------------------------------------------------------------
 * On-disk format.  The reference declares a vector-data file ("veb") and a metadata file ("vemb",
 * src/constants.ts:52-57) with the fields of VectorDataFormat / MetadataFormat (src/types.ts:78-113) but only ever
 * builds them in memory (serializeVectorData, src/binaryQuantizationFormat.ts:483-530).  Here they are real files whose
 * payload IS the device layout, so loading is a straight file -> HBM copy with no re-tiling:
 *   <prefix>.vemb  "BVEC", version, MetadataFormat {fieldNumber, vectorEncodingOrdinal, vectorSimilarityOrdinal,
 *                  dimensions, vectorDataOffset, vectorDataLength, vectorCount, centroidSquareMagnitude}, the tile
 *                  geometry, centroid f32[dimensions], checksums            (DESIGN.md "On-disk format")
 *   <prefix>.veb   the 64-row tile records (packed 1-bit codes + corrections per row = VectorDataFormat's
 *                  binaryValues, lowerInterval, upperInterval, additionalCorrection, quantizedComponentSum),
 *                  then the exact-corrections side array of the compact layout
 * Little-endian.  A row shard is saved WITH its pilot replica (format version 3: three more header words, the replica's tile records
 * behind the shard's own), so a one-process-per-GPU service restarts from files without the host rows.  A multi-device index is saved
 * as one ordinary pair per shard, <prefix>.s000, <prefix>.s001, ..., plus the manifest <prefix>.vemb ("BVEM": shard count, the shards'
 * row ranges, totals, centroid, checksum); bbq_index_load_multi puts it back over several devices, bbq_index_load over one.
 */
int bbq_index_save(bbq_index *idx, const char *path_prefix, const float *centroid, int32_t similarity_ordinal);
/* header of <prefix>.vemb; any output pointer may be NULL */
int bbq_index_file_info(const char *path_prefix, int64_t *n_rows, int32_t *dim, int32_t *similarity_ordinal,
                        double *centroid_dp, int64_t *row_base);
/* centroid_out [dim] (may be NULL).  Fails with BBQ_ERR_INVALID_ARG on a malformed, truncated or corrupted file.  A manifest of a
 * multi-device index loads as a multi-device handle with every shard on `device`. */
int bbq_index_load(const char *path_prefix, int32_t device, bbq_index **out, float *centroid_out);
/* a saved multi-device index over several devices: shard s on devices[s % n_devices] (n_devices 0 / devices NULL: shard s on device
 * s modulo the visible devices).  The shards, their row ranges and pilot replicas are the ones that were saved. */
int bbq_index_load_multi(const char *path_prefix, int32_t n_devices, const int32_t *devices, bbq_index **out, float *centroid_out);
/* shards of a saved index: 1 for an ordinary pair, the manifest's count for a multi-device index, 0 if unreadable */
int32_t bbq_index_file_shards(const char *path_prefix);
/* the rows back in the reference's shape: codes [n*ceil(dim/8)] (multi-bit index: [n*dim]), corr [n*4] (either may be NULL) - what
 * vectorValue(ord) / getCorrectiveTerms(ord) return (src/binaryQuantizationFormat.ts:52-76) */
int bbq_index_export(bbq_index *idx, uint8_t *codes, double *corr);

/* ------------------------------------------------------------------------------------------
 * Oversample + exact rerank (the reference's recall recipe: src/topKSelector.ts:29-115,
 * tests/recall-common.ts:188-213).  The ORIGINAL fp32 vectors stay resident in HBM next to the 1-bit
 * index; true scores are computeSimilarity (src/vectorSimilarity.ts:14-126): f64, accumulated in index
 * order, bit-identical to the reference.
 */
typedef struct bbq_vectors bbq_vectors;
/* vectors [n*dim] row-major (the Float32Array[] the reference's selectors take as `vectors`), copied to the device */
int bbq_vectors_create(const float *vectors, int64_t n, int32_t dim, int32_t device, bbq_vectors **out);
void bbq_vectors_destroy(bbq_vectors *v);
int64_t bbq_vectors_size(const bbq_vectors *v);
int32_t bbq_vectors_dimension(const bbq_vectors *v);
/* computeSimilarity(queries[q], vectors[rows[j]], true_sim) for j in [offsets[q], offsets[q+1]), q < n_queries.
 * queries [n_queries*dim] raw fp32 (NOT normalised, exactly what the caller would hand computeCosineSimilarity);
 * offsets [n_queries+1] ascending from 0; out_true [offsets[n_queries]].  A row outside [0, n) fails with
 * BBQ_ERR_INVALID_ARG (the reference skips / throws on a missing vector, src/topKSelector.ts:44-45). */
int bbq_rerank_scores(bbq_vectors *v, int32_t n_queries, const float *queries, const int64_t *offsets,
                      const int32_t *rows, int32_t true_sim, double *out_true);
/* The whole recipe for n_queries queries: searchNearestNeighbors with k*factor on idx (arguments as
 * bbq_search_batch), true scores of those candidates on v, then the reference's selection:
 *   selector 0  getOversampledTopKWithHeap (src/topKSelector.ts:29-79): MinHeap of k on trueScore, drained, sorted
 *   selector 1  getOversampledTopKWithSort (:92-115): stable sort by trueScore descending, first k
 * true_sim is the similarity of the rerank (the reference's selectors always use COSINE = 1).
 * out_idx / out_quantized / out_true [n_queries*k] (query q at offset q*k), out_n [n_queries].
 * NaN true scores (non-finite inputs) make the reference's final Array.sort order engine-defined; here they sort as
 * "equal to everything", which is what the comparator returns. */
int bbq_search_rerank_batch(bbq_index *idx, bbq_vectors *v, int32_t n_queries, const float *queries,
                            const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim, int64_t k,
                            int32_t factor, int32_t selector, int32_t true_sim, int32_t *out_idx,
                            float *out_quantized, double *out_true, int64_t *out_n);

/* ------------------------------------------------------------------------------------------
 * Host-side quantizer (multithreaded C++): what the JS host calls for quantizeVectors /
 * quantizeQueryVector so that the drop-in API needs no TypeScript arithmetic.
 */
/* BinaryQuantizationFormat.quantizeVectors, src/binaryQuantizationFormat.ts:165-263 (+ normalizeVector,
 * computeCentroid, OptimizedScalarQuantizer.scalarQuantize :108-227, packAsBinary :420-446).
 * vectors [n*dim] row-major.  index_bits == 1: codes [n*ceil(dim/8)] packed; otherwise codes [n*dim] unpacked.
 * Fails with BBQ_ERR_EMPTY / BBQ_ERR_NAN_INPUT / BBQ_ERR_INF_INPUT like the reference throws; for the
 * NaN/Inf cases *bad_row / *bad_col (may be NULL) receive the offending position. */
int bbq_quantize_vectors(const float *vectors, int64_t n, int32_t dim, int32_t sim, int32_t index_bits,
                         double lambda, int32_t iters, int32_t n_threads, uint8_t *codes, double *corr,
                         float *centroid, int64_t *bad_row, int32_t *bad_col);
/* searchNearestNeighbors' query preparation, src/binaryQuantizationFormat.ts:337-347 + :271-299
 * (COSINE: the query is normalised twice) */
int bbq_quantize_query(const float *query, int32_t dim, const float *centroid, int32_t sim, int32_t query_bits,
                       double lambda, int32_t iters, uint8_t *qquant, double *qcorr);
/* bbq_quantize_query for n queries at once on n_threads host threads (0 = all cores): queries [n*dim], qquant [n*dim],
 * qcorr [n*4].  On failure *bad_query (may be NULL) is the first offending query and the error is the one
 * bbq_quantize_query reports for it.  A 768-d query takes ~50 us on one core; batches feed the device at its own rate. */
int bbq_quantize_queries(const float *queries, int32_t n, int32_t dim, const float *centroid, int32_t sim,
                         int32_t query_bits, double lambda, int32_t iters, int32_t n_threads, uint8_t *qquant,
                         double *qcorr, int32_t *bad_query);
/* quantizeQueryVector alone, src/binaryQuantizationFormat.ts:271-299 (normalises once for COSINE) */
int bbq_quantize_query_vector(const float *query, int32_t dim, const float *centroid, int32_t sim,
                              int32_t query_bits, double lambda, int32_t iters, uint8_t *qquant, double *qcorr);
/* computeDotProduct(centroid, centroid), src/vectorOperations.ts:171-185 */
double bbq_centroid_dp(const float *centroid, int32_t dim);

/* ------------------------------------------------------------------------------------------
 * Introspection for bench.py / profiling (not part of the drop-in surface)
 */
typedef struct {
  double last_scan_ms;        /* hipEvent time of the dominant (largest) scan launch of the last batch call */
  int64_t last_scan_rows;     /* rows x queries that launch covered */
  int64_t last_scan_bytes;    /* algorithmic HBM bytes of that launch */
  int64_t candidates;         /* candidates replayed on the host in the last call (all queries) */
  int64_t dense_fallbacks;    /* queries of the last call that took the dense path */
  double total_scan_ms;       /* sum of hipEvent times of every dominant-scan launch since the last reset */
  int64_t total_scan_bytes;   /* and their algorithmic bytes */
  int64_t total_scan_launches;
  int64_t host_replays;       /* queries of the last call whose heap was replayed on the host (equal scores in or at the edge of the
                                 answer, k > 1024, device_select 0); the others were selected and sorted by the last finalize launch */
  int64_t resident_bytes;     /* bytes of ONE query's sweep of the index (all launches of its sub-batch; pilot replica included) that are
                                 loaded with the default cache policy, so that they stay in the device's 256 MiB Infinity Cache from one
                                 query's sweep to the next (option resident_mb, per launch); the rest is streamed with non-temporal loads.
                                 Their re-reads do not reach HBM */
} bbq_stats;
int bbq_get_stats(bbq_index *idx, bbq_stats *out);
int bbq_reset_stats(bbq_index *idx);
/* tuning knobs; returns BBQ_ERR_INVALID_ARG for unknown names or values out of range (DESIGN.md "Knobs"):
 *   batch_queries 0..1024 (0 = by index size: 32 from 6 M rows, 64 from 2.5 M, 128 below; at least 64 with sweep_share 32)   pipeline_slots 1..4 (3)   segment_growth 2..1024 (8)   first_segment_rows 1024..8192 (4096)
 *   resident_mb -1..2^20 (MiB of its row range a sweep launch keeps cache-resident; -1: this index's share of 224 MiB, by size among the
 *     indexes active on its device; 0: stream everything)   resident_interleave 0|1 (1: the resident chunks are spread over the range)
 *   replay_threads 1..256 (half the host cores, at most 16)   flood_rows 0..2^24 (262144)   force_dense 0|1 (0)
 *   sweep_share 1|4|8|32 (1: every query sweeps the index itself; 32: shared sweep on the matrix cores, groups of 32 queries, two groups per load of the rows)
 *   device_select 0|1 (1: for k <= 1024 the device selects and sorts the answer itself whenever no two scores in or at the edge of it
 *   compare equal - then the reference heap provably returns that order - and the host replays the heap only for the rest)
 *   latency_queries 0..1024 (4), latency_growth 2..4096 (64): calls with at most latency_queries queries walk the index in
 *   segments that grow by latency_growth instead of segment_growth (fewer dependent launches, more candidates per query) and their
 *   sweeps append the candidates to the query's list themselves (one atomic per workgroup; append_last 0|1 (1): also the last segment)
 *   latency_fused 0|1 (1): a call with ONE query runs without a copy at either end (query in the kernel arguments of every sweep, answer
 *   polled from mapped host memory); latency_presample 0|1 (1): and, from 262144 rows, with its threshold from per-wave top keys of a
 *   prefix - four launches in all (DESIGN.md "The single-query call") */
int bbq_set_option(bbq_index *idx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
