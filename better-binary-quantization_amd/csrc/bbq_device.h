// bbq_device.h - structs shared by the HIP kernels and the host orchestration (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbq {

constexpr int kTileRows = 64;        // one wavefront = one tile: lane r owns row r of the tile
#ifndef BBQ_CHUNK_ROWS
#define BBQ_CHUNK_ROWS 512
#endif
constexpr int kChunkRows = BBQ_CHUNK_ROWS;  // rows per workgroup (one 64-row tile per wave); candidate slots are per chunk
constexpr int kTilesPerChunk = kChunkRows / kTileRows;
constexpr uint32_t kCountRedirect = 0x80000000u;  // chunk count: the entries live in the overflow area
constexpr uint32_t kFlagOverflow = 1u;   // a candidate slot / list / key buffer overflowed
constexpr uint32_t kFlagNaN = 2u;        // a NaN score was produced: order statistics are meaningless
// Append counters (ScanArgs::append_counts, FinalizeArgs::append_counts): one per query, each in a 128-byte line of its own.  The
// memory side retires returning atomics on ONE line one after the other (scripts/ubench/atomic_lines.hip: 5.8 ns each on 32 adjacent
// words, 0.4 ns on 32 words in separate lines): with the counters of a sub-batch side by side the atomics of a sweep's workgroups
// took longer than the sweep.
constexpr int kAppendStride = 32;        // uint32 words between two queries' counters

// Device layout of one index storage (DESIGN.md "HBM layout").  A tile record holds 64 rows:
//   [w16][64] uint4    16-byte code chunk j of row r at (j*64 + r)*16      -> 1 KiB coalesced per wave load
// followed by the corrections, in one of two layouts:
//  kLayoutInline (exact corrections streamed with the codes, 24 or 32 B/row)
//   [64] double2       {lowerInterval, upperInterval}                      -> 1 KiB
//   [64] double        additionalCorrection                                -> 512 B
//   [64] double        quantizedComponentSum (only if has_x1)              -> 512 B
//  kLayoutCompact (4 B/row streamed; exact corrections in a side array, gathered only for rows whose score BOUND passes
//  the threshold)
//   [64] uint32        bf16(lower) | bf16(upper) << 16 (f32 bits truncated)       -> tiles stay multiples of 128 B
//   side array exact[row] = {lower, upper, additional, 0} as 4 doubles (32 B/row), followed by
//   side array add_range[tile] = {min, max} of the tile's additionalCorrection as f32 (8 B per 64 rows, one broadcast load
//   per wave): the bound takes whichever end makes the score larger; the additive term varies far less inside a tile than
//   scores vary between rows
constexpr int kLayoutInline = 0;
constexpr int kLayoutCompact = 1;
// Multi-bit indexes (indexBits > 1; the reference keeps such rows as one byte per dimension,
// src/binaryQuantizationFormat.ts:241-245): a row is stored as `store_bits`-wide fields, field f of row dword w = dimension
// w * (32 / store_bits) + f at bits [f * store_bits, (f + 1) * store_bits); rows are zero-padded to 16-byte chunks and laid out
// in the same [w16][64] uint4 tile records, followed by the same corrections blocks.
__host__ __device__ inline int store_bits_of(int index_bits) { return index_bits <= 1 ? 1 : index_bits <= 2 ? 2 : index_bits <= 4 ? 4 : 8; }
__host__ __device__ inline int row_bytes_of(int dim, int store_bits) { return (dim * store_bits + 7) / 8; }
// 16-byte units of staged query data per 16-byte chunk of a row.  1-bit rows: one bit-plane per query bit (QB = 1, 2, 4, 8).
// Multi-bit rows: per row dword, the query's low nibbles (and, for query values above 15: QB = 8, its high nibbles) in the order
// the kernel unfolds the fields: store_bits 2 -> {even fields, odd fields}, 4 -> the 8 nibbles, 8 -> the 4 bytes.
__host__ __device__ constexpr int query_units_per_chunk(int qb, int store_bits) {
  return store_bits == 1 ? qb : store_bits == 2 ? (qb > 4 ? 4 : 2) : store_bits == 4 ? (qb > 4 ? 2 : 1) : 1;
}
// bytes of one tile record
__host__ __device__ inline int tile_stride_of(int w16, int layout, int has_x1) {
  return w16 * 1024 + (layout == kLayoutCompact ? 4 * kTileRows : 1536 + (has_x1 ? 512 : 0));
}
struct IndexView {
  const uint8_t *tiles;
  const double *exact;  // kLayoutCompact: [n_rows padded to 64][4]
  const float *add_range;  // kLayoutCompact: [tiles][2] {min, max} of additionalCorrection inside the tile
  int64_t n_rows;       // valid rows in this storage
  int32_t w16;          // 16-byte chunks per row = ceil(ceil(dim/8)/16)
  int32_t tile_stride;  // bytes per tile record
  int32_t has_x1;       // 0: quantizedComponentSum == popcount(row), recomputed on the fly
  int32_t dim;
  int32_t layout;
  int32_t store_bits;   // 1: packed 1-bit rows; 2 / 4 / 8: multi-bit fields (indexBits 2 / 3-4 / 5-8)
  // cache residency of a launch (launch_view(), bbq_core.cpp): the chunks read with the default cache policy - they stay in the 256 MiB
  // Infinity Cache from one query's sweep to the next - the others are streamed with non-temporal loads
  int64_t resident_tiles;  // resident_share < 0: chunks whose first tile is below this
  int64_t resident_share;  // >= 0: chunk c is resident iff (c & 63) < resident_share (the resident chunks spread over the whole sweep)
  int64_t nt_delta;        // always 0.  The streamed loads add it to their address so that the compiler sees two different addresses
                           // in the two branches: it merges loads that differ only in the cache policy into ONE plain load
};

// Per-query uniforms of the score formula (src/batchDotProduct.ts:478-617)
struct QueryParams {
  double ay;     // query lowerInterval
  double ly;     // (upper-lower) [* FOUR_BIT_SCALE for every queryBits != 1]
  double y1;     // query quantizedComponentSum
  double qadd;   // query additionalCorrection
  double cdp;    // centroid . centroid
  double dimd;   // dimension as a double
  int32_t sim;   // BBQ_EUCLIDEAN / BBQ_COSINE / BBQ_MAXIMUM_INNER_PRODUCT
  int32_t one_bit;
  int32_t mip_plain;  // 1: MAXIMUM_INNER_PRODUCT is scaleMaxInnerProductScore(t) without the division by FOUR_BIT_SCALE: the
                      // per-row scorer's form (src/binaryQuantizedScorer.ts:207-209), which answers for multi-bit indexes
  int32_t pad_;
};

struct ScanArgs {
  IndexView idx;
  const uint4 *qplanes;        // [Q][w16][QB] bit-planes of the quantized query, packed like the rows (multi-bit index: [Q][w16*4][QN]
                               // dwords, the query's nibbles / bytes in the field order of the row dwords, see tile_dot_multibit)
  const QueryParams *qparams;  // [Q]
  int64_t chunk_begin;         // first chunk of this launch inside idx
  int64_t row_id_base;         // global row id of idx row 0
  // sparse output (candidates above the per-query threshold)
  const uint32_t *theta;       // [Q] monotone keys; a row is a candidate iff key(score) > theta
  uint32_t *counts;            // [Q][n_chunks]
  uint64_t *entries;           // [Q][n_chunks][cap], ascending by row inside a chunk
  uint32_t *flags;             // [Q]
  int32_t cap;
  int32_t n_chunks;            // chunks in this launch (= gridDim.x)
  // flood tier (may be null): a chunk with more than `cap` candidates parks ALL of them, row-ordered, in a block of the
  // query's overflow area and leaves counts = kCountRedirect | n, entries[0] = block offset (bbq_scan_kernel only)
  uint64_t *ovf;               // [Q][ovf_cap]
  uint32_t *ovf_counts;        // [Q] entries handed out so far
  int32_t ovf_cap;
  // append mode (calls with few queries; null: chunk slots as above): a workgroup reserves room for its candidates right in the
  // query's list with ONE atomic and writes them there, unordered inside the segment - the finalize launch then has nothing to
  // compact (walking 19 K chunk counters cost it 20-30 us of a 250 us call).  The answer is selected on the device and does not
  // care about the order; the rare host replay (equal scores) sorts the list by row first.
  uint64_t *append_lists;      // [Q][append_cap]
  const int32_t *append_base;  // [Q][2]: entries the list holds from the earlier segments (list_counts)
  uint32_t *append_counts;     // [Q][kAppendStride] entries this launch has reserved so far (word 0 of each query's line)
  int64_t append_cap;
  // dense output (every row), indexed by row - chunk_begin*1024
  float *dense_score32;        // [Q][dense_stride] or null
  int32_t *dense_qcdist;       // or null
  double *dense_score64;       // or null
  int64_t dense_stride;
};

struct FinalizeArgs {
  // input: either slots of one sparse launch, or the dense f32 scores of the first segment
  const uint32_t *counts;      // [Q][n_chunks]
  const uint64_t *entries;     // [Q][n_chunks][cap]
  const float *dense_score32;  // [Q][dense_stride]  (dense_rows > 0 selects this input)
  int64_t dense_stride;
  int32_t dense_rows;
  int64_t dense_row_id_base;
  int32_t n_chunks;
  int32_t cap;
  const uint64_t *ovf;         // flood tier of the scan launch (null: none)
  int32_t ovf_cap;
  uint32_t *append_counts;     // [Q][kAppendStride] non-null: the scan launch appended its candidates to the list itself (ScanArgs::append_lists);
                               // counts / entries are unused, the launch only takes the keys from list[base, base + count) and resets the counter
  // candidate list being built, ascending by global row
  uint64_t *lists;             // [Q][list_cap]
  int32_t *list_counts;        // [Q][2] {count, flags}
  int64_t list_cap;
  int32_t emit;                // 0: thresholds only (pilot replica on a shard that does not own those rows)
  // running top-k keys and the threshold for the next segment
  uint32_t *topk_keys;         // [Q][k]
  int32_t *topk_counts;        // [Q]
  uint32_t *theta;             // [Q]
  uint32_t *flags;             // [Q]
  int32_t k;
  int32_t need_theta;          // 0 on the last segment
  // final selection (last segment only; null: none).  The caller runs the whole query with k = k2 + 1, so the running top keys
  // hold the (k2 + 1) largest keys seen: the launch then knows the (k2 + 1)-th largest f32 score of the whole index exactly,
  // gathers the k2 rows above it from the list, sorts them by score and checks that no two of them - and not the boundary -
  // compare equal.  In that case the reference's heap (src/binaryQuantizationFormat.ts:383-411) ends up holding exactly those rows
  // whatever its history was and pops them in ascending score order, so the answer is the descending sort (header slot 1 = {k2, 0}).
  // Any tie, NaN flag, list overflow, a flood beyond the key buffer or k2 > kFinalSelectMax leaves {0, 1} there: the host
  // replays the heap over the list as before.
  // final_out [Q][final_stride]: slot 0 = {list count, flags} (a copy of list_counts), slot 1 = {entries that follow, 1 = replay
  // the list on the host}, then the answer (global row << 32 | f32 score bits) descending by score: ONE device-to-host copy
  // brings everything the host needs
  uint64_t *final_out;
  int32_t final_stride;        // >= final_k + 2 (+ 3 in shard mode)
  int32_t final_k;             // k2 = min(k, rows of the index)
  // latency path: final_out is MAPPED HOST memory; after the answer the launch stores `seq` into *done_flag (system scope) - the host
  // polls it instead of waiting for a copy and an event - and leaves the query's control words clean for the next call (the chain has
  // no host-to-device copy that would reset them)
  uint64_t *done_flag;
  uint64_t seq;
  // Shard mode (bbq_shard_scan_begin's dev_answers, include/bbq.h): this storage is one row shard of a larger index and the running top
  // keys may include rows of a pilot replica, so the launch cannot prove an answer by itself.  It leaves what the merge needs instead:
  // slot 1 = {m, unproven}, slot 2 = the cut = the (k2 + 1)-th largest key over every row this shard has seen (0: it has seen at most
  // k2), then the m <= k2 listed rows above the cut, descending by score.  No tie check here: the merge checks the global answer.
  int32_t final_shard;
};
constexpr int kFinalSelectMax = 1024;  // largest k2 the finalize kernel selects and sorts itself

// Latency path (bbq_latency_kernels.hip + the finalize kernel): ONE query per call - the reference's own call shape - with no copy at
// either end: every sweep takes the query from its KERNEL ARGUMENTS, the last finalize launch writes the answer into mapped host
// memory and raises a sequence word the host is polling.
constexpr int kLatPlaneMax = 96;     // 16-byte blocks of staged query data that fit the kernel arguments (768-d / 1536-d x 4 planes: 24 / 48)
struct LatScanArgs {
  IndexView idx;
  int64_t row_id_base;
  int64_t chunk_begin;
  int32_t n_chunks;
  int32_t first;                     // 1: the dense prefix - threshold 0 (every row is listed), nothing listed before it
  const uint32_t *theta;             // the query's control words (device memory), as ScanArgs has them per query
  uint32_t *flags;
  const int32_t *list_counts;        // {count, flags}: entries the list holds from the earlier segments
  uint32_t *append_count;
  uint64_t *list;
  int64_t list_cap;
  // the query itself, in the arguments of every launch of the chain
  QueryParams p;
  uint4 planes[kLatPlaneMax];
};

// Pre-sampled threshold (the first step of the single-query call on large indexes): the first `rows` rows are scored exactly, every
// wave leaves the `per_wave` largest keys of its 64 rows, and one small launch selects the rank-th largest of all of them - the
// k-th largest of a SUBSET of the rows is a valid lower bound of the k-th largest of the index, and with per_wave = 4 it is the
// prefix's true order statistic unless one wave holds five of its best rows.  ONE sweep over all rows with that threshold then lists
// every row above it (a few thousand) and the final selection works on that list alone.
constexpr int kLatPreKeys = 12288;  // keys the selection launch takes (48 KB of LDS; = kFinalizeKeyCap, what the final selection holds)
struct LatPreArgs {
  IndexView idx;
  int32_t rows;                      // rows [0, rows) are sampled (a multiple of kChunkRows, <= idx.n_rows)
  int32_t per_wave;                  // 1..4 keys per wave
  uint32_t *pre_keys;                // [rows / 64 * per_wave]
  uint32_t *flags;                   // NaN scores raise kFlagNaN
  QueryParams p;
  uint4 planes[kLatPlaneMax];
};

// exact rerank (bbq_rerank_kernels.hip): candidates of query q are rows[offsets[q] .. offsets[q+1])
struct RerankArgs {
  const float *vecs;       // [n][dim] original fp32 rows, row-major
  int64_t n;
  int32_t dim;
  int32_t sim;             // 0 EUCLIDEAN, 1 COSINE, 2 MAXIMUM_INNER_PRODUCT
  const float *queries;    // [Q][dim] raw fp32 queries
  const int64_t *offsets;  // [Q+1]
  const int32_t *rows;     // [offsets[Q]] row numbers, all in [0, n)
  double *out;             // [offsets[Q]] computeSimilarity(query, row)
};

constexpr int kFinalizeThreads = 1024;
constexpr int kFinalizeJobs = 4096;      // non-empty chunks one finalize launch copies entry-parallel (more: thread by thread)
constexpr int kFinalizeKeyCap = 12288;   // LDS key buffer of the finalize kernel (new keys + running top-k)
constexpr int kFinalizeCountCap = 24576; // chunk counters one launch stages in LDS, 16 bits each (12.6 M rows per segment; longer segments read them from memory)
constexpr int kFinalizeLdsBytes = (kFinalizeKeyCap + 3 * kFinalizeJobs + 512 + 16 + 16) * 4 + kFinalizeCountCap * 2;  // 146 KB of the CU's 160 KB
// a finalize launch behind an appending or a dense sweep: keys, histogram, scratch and the 8 KB of the job table the final selection sorts in
constexpr int kFinalizeLdsBytesSmall = (kFinalizeKeyCap + 512 + 16 + 16) * 4 + kFinalSelectMax * 8;

__host__ __device__ inline uint32_t key_of_bits(uint32_t b) { return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }

}  // namespace bbq
