// bbq_host.h - host-side internals of libbbq shared by its translation units: the device-resident index object, the
// per-device context (streams, events, per-slot workspace) and the helpers every entry point needs.
//   bbq_core.cpp     index creation from rows, segment plan, pipelined search, sharded scan, options
//   bbq_build.cpp    quantizeVectors on the device (bbq_index_build)
//   bbq_rerank.cpp   oversample + exact rerank (bbq_vectors_*, bbq_rerank_scores, bbq_search_rerank_batch)
//   bbq_persist.cpp  on-disk format (bbq_index_save / load / file_info / export)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "bbq_internal.h"
#include "bbq_launch.h"

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(BBQ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace bbq {

// device scratch that is released on every exit path
struct DevMem {
  void *p = nullptr;
  DevMem() = default;
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
  ~DevMem() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

constexpr int kMaxSlots = 4;
constexpr int64_t kMaxFastK = 4096;  // beyond this the dense path is used: the finalize kernel selects thresholds among at most 12288 keys (running top-k + new), and with k close to that the thresholds get too weak to pay (k = 4096: 4 ms per 10 M-row query, dense path 16 ms, k = 6000 on the sparse path 23 ms)

struct Storage {
  uint8_t *d_tiles = nullptr;
  double *d_exact = nullptr;  // kLayoutCompact: exact corrections, gathered for the rows whose bound passes; the per-tile
                              // additive-correction ranges (view.add_range) live behind them in the same allocation
  IndexView view{};
  int64_t row_id_base = 0;
  int64_t n_chunks() const { return (view.n_rows + kChunkRows - 1) / kChunkRows; }
};

struct Segment {
  int storage;  // 0 = pilot replica, 1 = main
  int64_t chunk_begin, n_chunks, rows;
  bool dense, emit, need_theta, dominant;
  int cap;
  bool big = false;  // large sweeps of different sub-batches are serialised through an event chain
};

struct Plan {
  int64_t k = -1;
  std::vector<Segment> segs;
  int64_t s0 = 0;
  int64_t list_cap = 0;
  int64_t flood_cap = 0;  // per-query entries of the flood tier (overflow area + list headroom); 0: none
  int64_t max_slots = 0;  // max over sparse segments of n_chunks*cap
  int64_t max_chunks = 0;
  int64_t final_k = 0;    // > 0: the plan runs with k = final_k + 1 and the last finalize launch selects the answer itself
  int growth = 0;         // segment growth the plan was built with
  bool latency = false;   // plan of a call with few queries: latency_growth, candidates appended to the list by the scan launches
};

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr, ev_big = nullptr;
  // capacities the buffers below were allocated for
  int q_cap = 0;
  int64_t qbuf_bytes = 0, chunks_cap = 0, slots_cap = 0, dense_cap = 0, list_cap = 0, k_cap = 0, hprefix = 0, flood_cap = 0;
  // one device block [control words of the sub-batch | staged queries] and its pinned host twin whose control part stays zero: ONE
  // host-to-device copy resets the thresholds / counters and brings the queries (d_theta.. and d_qbuf point into it)
  uint8_t *d_block = nullptr, *h_block = nullptr;
  int64_t ctrl_bytes = 0;
  uint8_t *d_qbuf = nullptr, *h_qbuf = nullptr;
  uint32_t *d_theta = nullptr, *d_flags = nullptr, *d_counts = nullptr, *d_topk = nullptr, *d_append_counts = nullptr;
  bool appended = false;  // the in-flight sub-batch's lists are unordered inside their segments (append mode)
  int32_t *d_topk_counts = nullptr, *d_list_counts = nullptr, *h_list_counts = nullptr;
  uint64_t *d_entries = nullptr, *d_lists = nullptr, *h_lists = nullptr, *d_ovf = nullptr;
  uint32_t *d_ovf_counts = nullptr;
  float *d_dense0 = nullptr;
  // final selection on the device (FinalizeArgs::final_out): the sorted answer + {count, needs-host-replay} per query
  uint64_t *d_final = nullptr, *h_final = nullptr;  // [q_cap][final_stride]: 2 header slots + the answer per query
  int64_t final_stride = 0;
  bool final_used = false;  // the in-flight sub-batch was enqueued with the final selection (its list prefix was NOT copied to the host)
  // in-flight sub-batch: busy = device work enqueued and not yet collected; replaying = host replay jobs outstanding
  bool busy = false;
  bool replaying = false;
  std::atomic<int> pending{0};
  std::vector<std::vector<uint64_t>> tails;
  std::vector<int64_t> host_cnt;  // entries of query i in its h_lists row (the rest, if any, in tails[i])
  std::vector<int> dense_q;
  int nq = 0;
  int64_t q_first = 0;
  bool timed = false;
  int64_t timed_rows = 0, timed_bytes = 0;
  // non-null: the in-flight sub-batch belongs to an asynchronous sharded scan of that index (bbq_shard_scan_begin); nothing is to be
  // collected from it but the timing.  Such slots stay busy ACROSS API calls: every entry point that uses the slots settles them first.
  bbq_index *shard_owner = nullptr;
  // the control words are all zero (the latency chain leaves them so and expects them so; every other use of the slot dirties them
  // and resets them with its own host-to-device copy)
  bool ctrl_clean = false;
};

// Per-device context shared by every index on that device: streams, events and the per-slot workspace are expensive
// to create (~10 ms per index with hipStreamCreate/Destroy) and quickSearch builds a fresh index on every call
// (src/index.ts:109), so they live for the process.  One API call at a time per device (mutex).
struct DeviceCtx {
  int device = 0;
  std::mutex mu;
  bool ready = false;
  Slot slots[kMaxSlots];
  hipStream_t aux_stream = nullptr;   // dense path / bbq_score_rows / index build: never touches an in-flight slot
  uint8_t *d_aux_qbuf = nullptr;
  int64_t aux_qbuf_bytes = 0;
  uint32_t *d_aux_flags = nullptr;
  int last_big_slot = -1;             // slot whose ev_big marks the end of the most recently enqueued big sweep
  // latency path (bbq_latency_kernels.hip): the answer of a single-query call lands in mapped, coherent host memory and the host
  // polls a sequence word behind it: [0] sequence, [8 ..) header + entries
  uint64_t *h_lat = nullptr, *d_lat = nullptr;
  uint64_t lat_seq = 0;
  uint32_t *d_pre_keys = nullptr;     // [kLatPreKeys] per-wave top keys of the pre-sampled threshold
  // the indexes that launched sweeps on this device lately share its Infinity Cache (launch_view, bbq_core.cpp); under `mu`
  struct CacheUser { const void *index; int64_t bytes; uint64_t tick; };
  std::vector<CacheUser> cache_users;

};
constexpr int kLatAnswerOffset = 8;   // words in front of the answer block inside DeviceCtx::h_lat

// per query: bit-planes (up to 8) + int8 values in MFMA fragment order + score uniforms + group maxima
int64_t qbuf_bytes_per_query_w(int w16);
// returns the (lazily created, never destroyed) context of a device; call with hipSetDevice(device) done
int get_ctx(int device, DeviceCtx **out);
int ensure_aux_qbuf(DeviceCtx *c, int64_t bytes);
// bytes of the compact layout's side arrays for n_tiles tiles: exact corrections + add ranges
inline int64_t compact_side_bytes(int64_t n_tiles) { return n_tiles * kTileRows * 32 + n_tiles * 8; }
inline const float *add_range_of(const double *d_exact, int64_t n_tiles) {
  return d_exact ? reinterpret_cast<const float *>(d_exact + n_tiles * kTileRows * 4) : nullptr;
}

}  // namespace bbq

namespace bbq { struct MultiState; }

struct bbq_index {
  bbq::MultiState *multi = nullptr;  // non-null: this handle is a row-sharded index over several devices (bbq_multi.cpp); it owns no
                                     // storage itself and every entry point dispatches to the shards
  int device = 0;
  bbq::DeviceCtx *ctx = nullptr;
  bbq::Slot *slots = nullptr;  // = ctx->slots
  int32_t dim = 0, pb = 0, w16 = 0, tile_stride = 0, has_x1 = 0, bytes_per_row = 0, layout = 0, want_compact = 1;
  int32_t index_bits = 1, store_bits = 1;  // pb = stored bytes per row = ceil(dim * store_bits / 8)
  int64_t n_rows = 0, row_base = 0;
  double centroid_dp = 0;
  bool has_pilot = false;
  bbq::Storage pilot, main;
  bbq::Plan plan;
  hipStream_t aux_stream = nullptr;  // = ctx->aux_stream
  uint8_t *d_aux_qbuf = nullptr;     // = ctx->d_aux_qbuf
  uint32_t *d_aux_flags = nullptr;
  float *d_dense_all = nullptr;
  int64_t dense_all_cap = 0;
  // bbq_shard_scan_begin / _wait: per-query lists before packing, two sets (two batches may be in flight) and their tickets
  struct ShardSet {
    uint64_t *d_lists = nullptr;
    int32_t *d_counts = nullptr;  // [q_cap][2] + the packed total (int64) behind them
    int64_t q_cap = 0, list_cap = 0;
    hipEvent_t done = nullptr;    // recorded behind the packing of the batch
    int64_t *h_total = nullptr;   // pinned
    int64_t packed_cap = 0;
    bool in_flight = false;
  } shard_set[2];
  int64_t shard_begun = 0, shard_waited = 0;  // batches begun / waited for: ticket t uses set t & 1
  // options
  int opt_batch = 0 /* 0: by index size, effective_batch() */, opt_slots = 3, opt_growth = 8, opt_force_dense = 0, opt_share = 1, opt_device_select = 1;
  // a call with few queries is latency-bound: every segment costs a dependent scan + finalize launch pair (~15-20 us), so such calls
  // walk the index in fewer, faster-growing segments (more candidates per query - the device selects the answer itself anyway)
  int opt_latency_queries = 4, opt_latency_growth = 64;
  int opt_latency_append = 1;  // 0: calls with few queries keep the chunk slots (and the finalize launches their compaction)
  int64_t sweep_resident_acc = 0;  // cache-resident bytes of the launches of one sweep of the index, summed by launch_view()
  int opt_resident_interleave = 1;  // the resident chunks of a launch are spread over its range (of every 64 chunks the first n) instead of being its head
  int opt_resident_mb = -1;  // MiB of its row range that ONE sweep launch loads with the default cache policy, so that they stay in the Infinity
                             // Cache from one query's sweep to the next (launch_view(), bbq_core.cpp); -1: this index's share of kResidentAutoBytes
  int opt_latency_presample = 1;  // ... and on large indexes get their threshold from per-wave top keys of a prefix (two small launches) instead of two scan / finalize pairs
  int opt_latency_fused = 1;  // single-query calls take the three-launch latency path (bbq_latency_kernels.hip) when the index shape has one
  int opt_append_last = 1;  // append mode also for the last (largest) segment: its finalize launch gets cheaper, its sweep slower (one
                            // atomic per workgroup with candidates); measured at 10 M x 768: 0.250 ms per call with, 0.263 without
  // host threads replaying the heaps of one sub-batch: half the cores, at most 8 (a batch of 32 answers 1.4x sooner than with 1)
  int opt_replay_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
  int64_t opt_s0 = 4096;
  // flood tier: candidates one query may pile up beyond the planned list (rows stored cluster by cluster make the
  // query's own cluster beat a threshold that was derived from other clusters) before it has to take the dense path
  int64_t opt_flood = 262144;
  bbq_stats stats{};
};

namespace bbq {
// corrections layout asked for at creation: an explicit option wins, the environment variable only speaks for callers that passed none
inline int want_compact_of(const bbq_index_options *opts) {
  if (opts && opts->size >= (int32_t)sizeof(bbq_index_options) && opts->corrections != BBQ_CORRECTIONS_DEFAULT)
    return opts->corrections == BBQ_CORRECTIONS_INLINE ? 0 : 1;
  const char *e = getenv("BBQ_COMPACT_CORRECTIONS");
  return (e && e[0] == '0') ? 0 : 1;
}
inline int check_options(const bbq_index_options *opts) {
  if (!opts) return BBQ_OK;
  if (opts->size < (int32_t)sizeof(bbq_index_options)) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_options.size is %d, this library expects at least %d", opts->size, (int)sizeof(bbq_index_options));
  if (opts->corrections < BBQ_CORRECTIONS_DEFAULT || opts->corrections > BBQ_CORRECTIONS_COMPACT) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_options.corrections out of range: %d", opts->corrections);
  return BBQ_OK;
}
// the integer dot product of one row must fit 31 bits: dim * 255 for packed 1-bit rows (query values <= 255), dim * 255 * 255 for multi-bit fields
inline bool dim_supported(int64_t dim, int store_bits) { return dim * 255 * (store_bits > 1 ? 255 : 1) <= 0x7fffffffll; }
// frees what the index owns; the device context (streams, workspace) stays.  Call with the context mutex held.
void destroy_unlocked(bbq_index *ix);
// rows already in device memory (codes in the caller's shape: packed bits, or one byte per dimension for a multi-bit index;
// corrections [n][4]) -> tile records of `st`, deciding the index's layout on the way.  Context mutex held by the caller.
int storage_from_device_rows(bbq_index *ix, Storage &st, const uint8_t *d_codes, const double *d_corr, int64_t n_rows, int64_t row_id_base,
                             bool check_x1);
// waits for the slots an asynchronous sharded scan has left busy on this device (all, or only `owner`'s) and books their timing.
// Context mutex held by the caller.
int settle_shard_slots(DeviceCtx *ctx, bbq_index *owner);
// every f32 score of one query on this (single-device) index, to host memory [n_rows]
int dense_scores_host(bbq_index *ix, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim, float *out);
// multi-device index (bbq_multi.cpp): what the entry points of a handle with ix->multi != nullptr dispatch to
void multi_destroy(bbq_index *ix);
int multi_search_batch(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim,
                       int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n);
int multi_score_rows(bbq_index *ix, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim, int64_t row_begin,
                     int64_t row_count, int32_t *out_qcdist, double *out_score64, float *out_score32);
int multi_export(bbq_index *ix, uint8_t *codes, double *corr);
int multi_set_option(bbq_index *ix, const char *name, int64_t v);
int multi_get_stats(bbq_index *ix, bbq_stats *out);
// persistence of a multi-device index: every shard as an ordinary file pair + a manifest (bbq_persist.cpp / bbq_multi.cpp)
int multi_save(bbq_index *ix, const char *prefix, const float *centroid, int32_t sim);
int multi_assemble(bbq_index *const *shards, const int32_t *devices, int32_t n_shards, int32_t dim, int32_t index_bits, int64_t n_rows,
                   double centroid_dp, bbq_index **out);
int write_manifest(const char *prefix, int32_t n_shards, const int64_t *bounds, int32_t dim, int32_t index_bits, int32_t sim, int64_t n_rows,
                   double centroid_dp, int64_t pilot_rows, const float *centroid);
std::string shard_file_prefix(const char *prefix, int s);
int multi_reset_stats(bbq_index *ix);
}  // namespace bbq
