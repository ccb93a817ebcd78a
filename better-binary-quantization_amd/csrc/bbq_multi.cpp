// bbq_multi.cpp - ONE index row-sharded over several GPUs of this process (bbq_index_create_multi, include/bbq.h).
//
// The north star shards the index by row across the GPUs of a node.  For hosts that are one process - the reference's API is
// one synchronous call per query from one Node.js thread - the sharding lives here, behind the same bbq_index handle, so that
// bbq_search / bbq_search_batch / bbq_score_rows / bbq_search_rerank_batch use every device without the caller knowing:
//
//   shard s = rows [r_s, r_{s+1}) on device d_s (contiguous, ascending: global row order is what the exact heap replay needs),
//             every shard but the first with a PILOT REPLICA of the global prefix so that it derives valid thresholds without
//             waiting for the shards before it (bbq_index_create_shard)
//   search  = per round of queries: one worker thread per shard enqueues the sweep of its shard (bbq_shard_scan_begin: the shard's
//             own top rows above its cut - the SHARD-LOCAL ANSWER, k + 3 words per query - and the packed candidate lists stay in
//             that device's memory) and lands the answers in pinned host memory over ITS OWN PCIe link; the calling thread merges
//             the shards' answers (bbq_merge_answers: an S-way merge of k entries each plus the proof that the reference heap
//             returns exactly that) while the workers' next round is already running on the devices (two buffer sets, the sweep of
//             round r + 1 is enqueued before round r is waited for).  Only a query whose answer has equal scores in or at the edge
//             of it needs its lists: they are fetched for that round and the heap is replayed (bbq_replay), as every query was
//             before ABI 3.
// There is no device-to-device step: the merge is host work of microseconds per query, so a gather to one GPU (RCCL / xGMI) would
// only add a hop before the same D2H copy; the one-process-per-GPU deployment (torch.distributed over RCCL) is
// python/bbq_amd/distributed.py.
// A query some shard cannot bound (NaN scores, a flood beyond every buffer) or k > 4096 is scored densely on every shard and
// replayed row by row: exact, slow, rare.
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "bbq_host.h"

using namespace bbq;

namespace bbq {

// a persistent host thread per shard: spawning threads per call would cost more than a small sweep
class ShardWorker {
 public:
  ShardWorker() : th_([this] { run(); }) {}
  ~ShardWorker() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  void post(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.push_back(std::move(f));
    }
    cv_.notify_one();
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        f = std::move(q_.front());
        q_.pop_front();
      }
      f();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
  bool stop_ = false;
  std::thread th_;
};

struct ShardBuf {  // one round's output of one shard: answers + packed lists on the device, answers landed in pinned host memory
  uint64_t *d_packed = nullptr, *h_packed = nullptr;
  int64_t *d_offsets = nullptr, *h_offsets = nullptr;
  int32_t *d_flags = nullptr, *h_flags = nullptr;
  uint64_t *d_answers = nullptr, *h_answers = nullptr;
  int64_t answers_stride = 0;
  hipEvent_t landed = nullptr;  // behind the copy of the answers
  int64_t packed_cap = 0, h_packed_cap = 0;
  int32_t q_cap = 0;
  int64_t total = 0;
  bool lists_on_host = false;
  std::vector<int64_t> h_list_at;   // per query of the round: offset of its fetched list inside h_packed, -1 if it was not fetched
  int rc = BBQ_OK;
  std::string err;
};

struct MultiShard {
  bbq_index *ix = nullptr;
  int device = 0;
  int64_t r0 = 0, r1 = 0;
  ShardBuf buf[2];
  std::unique_ptr<ShardWorker> worker;
};

struct MultiState {
  std::vector<MultiShard> shards;
  std::mutex mu;  // one call at a time per multi-device index
  int round_queries = 512;
  // completion of posted jobs
  std::mutex done_mu;
  std::condition_variable done_cv;
  std::vector<int> done_count;  // per round: shards finished
};

}  // namespace bbq

namespace {

void free_buf(ShardBuf &b) {
  if (b.d_packed) (void)hipFree(b.d_packed);
  if (b.d_offsets) (void)hipFree(b.d_offsets);
  if (b.d_flags) (void)hipFree(b.d_flags);
  if (b.d_answers) (void)hipFree(b.d_answers);
  if (b.h_packed) (void)hipHostFree(b.h_packed);
  if (b.h_offsets) (void)hipHostFree(b.h_offsets);
  if (b.h_answers) (void)hipHostFree(b.h_answers);
  if (b.h_flags) (void)hipHostFree(b.h_flags);
  if (b.landed) (void)hipEventDestroy(b.landed);
  b = ShardBuf();
}

int ensure_buf(MultiShard &sh, ShardBuf &b, int32_t nq, int64_t k) {
  const int64_t want = std::max<int64_t>(bbq_shard_list_cap(sh.ix, k), 1024) * nq;
  const int64_t stride = k + 3;
  if (b.q_cap >= nq && b.packed_cap >= want && b.answers_stride >= stride) return BBQ_OK;
  HIPCHK(hipSetDevice(sh.device));
  free_buf(b);
  HIPCHK(hipMalloc((void **)&b.d_packed, (size_t)want * 8));
  HIPCHK(hipMalloc((void **)&b.d_offsets, (size_t)(nq + 1) * 8));
  HIPCHK(hipMalloc((void **)&b.d_flags, (size_t)nq * 4));
  HIPCHK(hipMalloc((void **)&b.d_answers, (size_t)nq * (size_t)stride * 8));
  HIPCHK(hipHostMalloc((void **)&b.h_offsets, (size_t)(nq + 1) * 8, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&b.h_answers, (size_t)nq * (size_t)stride * 8, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&b.h_flags, (size_t)nq * 4, hipHostMallocDefault));
  HIPCHK(hipEventCreateWithFlags(&b.landed, hipEventDisableTiming));
  b.packed_cap = want;
  b.q_cap = nq;
  b.answers_stride = stride;
  return BBQ_OK;
}

// one shard's part of one round, first half: enqueue the sweep and, behind it, the copy of the answers to the host.  Returns at once.
int shard_begin(MultiShard &sh, ShardBuf &b, int32_t nq, const uint8_t *qq, const double *qc, int32_t query_bits, int32_t sim, int64_t k,
                bool answers) {
  int rc = ensure_buf(sh, b, nq, k);
  if (rc != BBQ_OK) return rc;
  b.total = 0;
  b.lists_on_host = false;
  rc = bbq_shard_scan_begin(sh.ix, nq, qq, qc, query_bits, sim, k, b.d_packed, b.packed_cap, b.d_offsets, b.d_flags, answers ? b.d_answers : nullptr,
                            b.answers_stride);
  if (rc != BBQ_OK) return rc;
  HIPCHK(hipSetDevice(sh.device));
  hipStream_t st = sh.ix->aux_stream;  // the stream the packing of this batch was enqueued on
  if (answers) HIPCHK(hipMemcpyAsync(b.h_answers, b.d_answers, (size_t)nq * (size_t)b.answers_stride * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(b.landed, st));
  return BBQ_OK;
}

// second half: wait for the sweep and the copy
int shard_finish(MultiShard &sh, ShardBuf &b) {
  int rc = bbq_shard_scan_wait(sh.ix, &b.total);
  if (rc != BBQ_OK) return rc;
  HIPCHK(hipSetDevice(sh.device));
  HIPCHK(hipEventSynchronize(b.landed));
  return BBQ_OK;
}

// the packed lists of the queries that need their heap replayed (status != 0), to host memory: the offsets first, then one small
// copy per such query - a round's whole packed buffer is ~9 KB per query, and nearly every round of a few hundred queries holds one
// with equal scores in its answer
int shard_fetch_lists(MultiShard &sh, ShardBuf &b, int32_t nq, const std::vector<uint8_t> &status, bool all) {
  HIPCHK(hipSetDevice(sh.device));
  // the round's packing has completed (bbq_shard_scan_wait).  The null stream: the shard's own streams are non-blocking, and its
  // auxiliary stream may already hold the NEXT round's packing, which waits for that round's sweeps
  hipStream_t st = nullptr;
  HIPCHK(hipMemcpyAsync(b.h_offsets, b.d_offsets, (size_t)(nq + 1) * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(b.h_flags, b.d_flags, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int64_t need = 0;
  for (int32_t q = 0; q < nq; ++q)
    if (all || status[(size_t)q] != 0) need += b.h_offsets[q + 1] - b.h_offsets[q];
  if (b.h_packed_cap < std::max<int64_t>(need, 1)) {
    if (b.h_packed) HIPCHK(hipHostFree(b.h_packed));
    b.h_packed = nullptr;
    b.h_packed_cap = std::max<int64_t>(need, 1024) * 5 / 4;
    HIPCHK(hipHostMalloc((void **)&b.h_packed, (size_t)b.h_packed_cap * 8, hipHostMallocDefault));
  }
  // h_list_at[q] = where query q's entries sit in h_packed (-1: not fetched)
  b.h_list_at.assign((size_t)nq, -1);
  int64_t at = 0;
  for (int32_t q = 0; q < nq; ++q) {
    if (!(all || status[(size_t)q] != 0)) continue;
    const int64_t cnt = b.h_offsets[q + 1] - b.h_offsets[q];
    b.h_list_at[(size_t)q] = at;
    if (cnt > 0) HIPCHK(hipMemcpyAsync(b.h_packed + at, b.d_packed + b.h_offsets[q], (size_t)cnt * 8, hipMemcpyDeviceToHost, st));
    at += cnt;
  }
  HIPCHK(hipStreamSynchronize(st));
  b.lists_on_host = true;
  return BBQ_OK;
}

// a query no sparse list can answer: every shard scores all its rows, the reference loop runs over all of them
int dense_query(bbq_index *ix, const uint8_t *qq, const double *qc, int32_t query_bits, int32_t sim, int64_t k, int32_t *out_idx,
                float *out_score, int64_t *out_n) {
  MultiState *ms = ix->multi;
  std::vector<float> all((size_t)std::max<int64_t>(ix->n_rows, 1));
  for (MultiShard &sh : ms->shards) {
    int rc = dense_scores_host(sh.ix, qq, qc, query_bits, sim, all.data() + sh.r0);
    if (rc != BBQ_OK) return rc;
  }
  HeapReplay hr(k, ix->n_rows);
  for (int64_t i = 0; i < ix->n_rows; ++i) hr.offer(all[(size_t)i], (int32_t)i);
  *out_n = hr.finish(out_idx, out_score);
  return BBQ_OK;
}

}  // namespace

namespace bbq {

void multi_destroy(bbq_index *ix) {
  MultiState *ms = ix->multi;
  if (!ms) return;
  for (MultiShard &sh : ms->shards) {
    sh.worker.reset();  // joins the thread
    (void)hipSetDevice(sh.device);
    free_buf(sh.buf[0]);
    free_buf(sh.buf[1]);
    if (sh.ix) bbq_index_destroy(sh.ix);
  }
  delete ms;
  ix->multi = nullptr;
  delete ix;
}

int multi_search_batch(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim,
                       int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  const int64_t keff = std::min<int64_t>(k, ix->n_rows);  // thresholds are order statistics of rank min(k, N) (bbq_search_batch)
  const int dim = ix->dim;
  if (keff > kMaxFastK || ix->opt_force_dense) {
    for (int32_t q = 0; q < n_queries; ++q) {
      int rc = dense_query(ix, qquant + (size_t)q * dim, qcorr + (size_t)q * 4, query_bits, sim, k, out_idx + (int64_t)q * k,
                           out_score + (int64_t)q * k, out_n + q);
      if (rc != BBQ_OK) return rc;
    }
    return BBQ_OK;
  }
  const int S = (int)ms->shards.size();
  const int R = std::max(1, ms->round_queries);
  const int64_t rounds = ((int64_t)n_queries + R - 1) / R;
  const bool answers = keff <= kFinalSelectMax && ix->opt_device_select;
  {
    std::lock_guard<std::mutex> dl(ms->done_mu);
    ms->done_count.assign((size_t)rounds, 0);
  }
  // every shard's worker is a FIFO: begin(r) enqueues round r's sweep and returns, finish(r) waits for it.  Posted as
  // begin(0) begin(1) finish(0) finish(1) | begin(2) finish(2) | ... so that a device always has the next round queued behind the one
  // being waited for
  auto post_begin = [&](int64_t r) {
    const int32_t q0 = (int32_t)(r * R), nq = (int32_t)std::min<int64_t>(R, n_queries - r * R);
    for (int s = 0; s < S; ++s) {
      MultiShard *sh = &ms->shards[(size_t)s];
      ShardBuf *b = &sh->buf[r & 1];
      sh->worker->post([sh, b, nq, q0, qquant, qcorr, query_bits, sim, keff, dim, answers] {
        b->rc = shard_begin(*sh, *b, nq, qquant + (size_t)q0 * dim, qcorr + (size_t)q0 * 4, query_bits, sim, keff, answers);
        if (b->rc != BBQ_OK) b->err = bbq_last_error();
      });
    }
  };
  auto post_finish = [&](int64_t r) {
    for (int s = 0; s < S; ++s) {
      MultiShard *sh = &ms->shards[(size_t)s];
      ShardBuf *b = &sh->buf[r & 1];
      sh->worker->post([ms, sh, b, r] {
        if (b->rc == BBQ_OK) {
          b->rc = shard_finish(*sh, *b);
          if (b->rc != BBQ_OK) b->err = bbq_last_error();
        }
        {
          std::lock_guard<std::mutex> dl(ms->done_mu);
          ms->done_count[(size_t)r] += 1;
        }
        ms->done_cv.notify_all();
      });
    }
  };
  auto wait_round = [&](int64_t r) {
    std::unique_lock<std::mutex> dl(ms->done_mu);
    ms->done_cv.wait(dl, [&] { return ms->done_count[(size_t)r] == S; });
  };
  // the lists of round r's unproven queries to the host, on THIS thread: the shards' workers are FIFOs that already hold finish(r + 1),
  // which blocks until round r + 1 has been swept - a fetch posted behind it would stall this round's merge (and with it begin(r + 2))
  // for a whole round.  Null-stream copies out of buffers nothing writes before begin(r + 2).
  auto fetch_lists = [&](int64_t r, int32_t nq, const std::vector<uint8_t> *need) -> int {
    for (int s = 0; s < S; ++s) {
      MultiShard &sh = ms->shards[(size_t)s];
      ShardBuf &b = sh.buf[r & 1];
      if (b.lists_on_host) continue;
      int rc = shard_fetch_lists(sh, b, nq, *need, !answers);
      if (rc != BBQ_OK) return rc;
    }
    return BBQ_OK;
  };
  const int64_t ahead = std::min<int64_t>(rounds, 2);
  for (int64_t r = 0; r < ahead; ++r) post_begin(r);
  for (int64_t r = 0; r < ahead; ++r) post_finish(r);
  int64_t posted = ahead;
  int rc_all = BBQ_OK;
  std::string err_all;
  std::vector<const uint64_t *> blocks((size_t)S);
  std::vector<int64_t> strides((size_t)S);
  std::vector<const bbq_cand *> lists((size_t)S);
  std::vector<int64_t> counts((size_t)S);
  std::vector<uint8_t> status;
  for (int64_t r = 0; r < rounds; ++r) {
    wait_round(r);
    const int32_t q0 = (int32_t)(r * R), nq = (int32_t)std::min<int64_t>(R, n_queries - r * R);
    if (rc_all == BBQ_OK) {
      for (int s = 0; s < S; ++s) {
        ShardBuf &b = ms->shards[(size_t)s].buf[r & 1];
        if (b.rc != BBQ_OK && rc_all == BBQ_OK) { rc_all = b.rc; err_all = b.err; }
        blocks[(size_t)s] = b.h_answers;
        strides[(size_t)s] = b.answers_stride;
      }
    }
    if (rc_all == BBQ_OK) {
      status.assign((size_t)nq, 1);
      if (answers) {
        rc_all = bbq_merge_answers(S, blocks.data(), strides.data(), nq, ix->n_rows, k, ix->opt_replay_threads, out_idx + (int64_t)q0 * k,
                                   out_score + (int64_t)q0 * k, out_n + q0, status.data());
        if (rc_all != BBQ_OK) err_all = bbq_last_error();
      }
      bool need_lists = false;
      for (int32_t q = 0; q < nq; ++q) need_lists = need_lists || status[(size_t)q] != 0;
      if (rc_all == BBQ_OK && need_lists) {
        rc_all = fetch_lists(r, nq, &status);
        if (rc_all != BBQ_OK) err_all = bbq_last_error();
      }
      for (int32_t q = 0; q < nq && rc_all == BBQ_OK; ++q) {
        if (status[(size_t)q] == 0) {  // answered from the shards' answers: the entries that were merged count as its candidates
          for (int s = 0; s < S; ++s) ix->stats.candidates += (int64_t)(uint32_t)blocks[(size_t)s][(size_t)q * (size_t)strides[(size_t)s] + 1];
          continue;
        }
        const int64_t qi = q0 + q;
        bool flagged = status[(size_t)q] == 2;
        int64_t cand = 0;
        for (int s = 0; s < S; ++s) {
          const ShardBuf &b = ms->shards[(size_t)s].buf[r & 1];
          lists[(size_t)s] = b.h_packed + std::max<int64_t>(b.h_list_at[(size_t)q], 0);
          counts[(size_t)s] = b.h_offsets[q + 1] - b.h_offsets[q];
          cand += counts[(size_t)s];
        }
        if (!answers)  // no answer blocks in this call (k > 1024, device_select 0): the shards' flags came with the lists
          for (int s = 0; s < S; ++s) flagged = flagged || ms->shards[(size_t)s].buf[r & 1].h_flags[q] != 0;
        if (!flagged) {
          ix->stats.host_replays += 1;
          ix->stats.candidates += cand;
          rc_all = bbq_replay(S, lists.data(), counts.data(), ix->n_rows, k, out_idx + qi * k, out_score + qi * k, out_n + qi);
          if (rc_all != BBQ_OK) err_all = bbq_last_error();
          continue;
        }
        rc_all = dense_query(ix, qquant + (size_t)qi * dim, qcorr + (size_t)qi * 4, query_bits, sim, k, out_idx + qi * k, out_score + qi * k, out_n + qi);
        if (rc_all != BBQ_OK) err_all = bbq_last_error();
        ix->stats.dense_fallbacks += 1;
        ix->stats.candidates += ix->n_rows;
      }
    }
    // the buffers of round r are free again: round r + 2 may start (every posted round is waited for, also after an error)
    if (posted < rounds) {
      post_begin(posted);
      post_finish(posted);
      ++posted;
    }
  }
  if (rc_all != BBQ_OK) return fail(rc_all, "%s", err_all.c_str());
  return BBQ_OK;
}

int multi_score_rows(bbq_index *ix, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim, int64_t row_begin,
                     int64_t row_count, int32_t *out_qcdist, double *out_score64, float *out_score32) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  for (MultiShard &sh : ms->shards) {
    const int64_t lo = std::max(row_begin, sh.r0), hi = std::min(row_begin + row_count, sh.r1);
    if (lo >= hi) continue;
    const int64_t o = lo - row_begin;
    int rc = bbq_score_rows(sh.ix, qquant, qcorr, query_bits, sim, lo - sh.r0, hi - lo, out_qcdist ? out_qcdist + o : nullptr,
                            out_score64 ? out_score64 + o : nullptr, out_score32 ? out_score32 + o : nullptr);
    if (rc != BBQ_OK) return rc;
  }
  return BBQ_OK;
}

int multi_export(bbq_index *ix, uint8_t *codes, double *corr) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  const int64_t row_bytes = ix->store_bits == 1 ? ix->pb : ix->dim;
  for (MultiShard &sh : ms->shards) {
    int rc = bbq_index_export(sh.ix, codes ? codes + sh.r0 * row_bytes : nullptr, corr ? corr + sh.r0 * 4 : nullptr);
    if (rc != BBQ_OK) return rc;
  }
  return BBQ_OK;
}

int multi_set_option(bbq_index *ix, const char *name, int64_t v) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  if (std::string(name) == "round_queries") {
    if (v < 1 || v > 65536) return fail(BBQ_ERR_INVALID_ARG, "bbq_set_option: round_queries out of range");
    ms->round_queries = (int)v;
    return BBQ_OK;
  }
  const std::string n(name);
  if (n == "device_select" && v != 0 && v != 1)  // the shards refuse it too; checked here so that no shard is changed first
    return fail(BBQ_ERR_INVALID_ARG, "bbq_set_option: unknown option or value out of range: %s=%lld", name, (long long)v);
  for (MultiShard &sh : ms->shards) {
    int rc = bbq_set_option(sh.ix, name, v);
    if (rc != BBQ_OK) return rc;
  }
  if (n == "replay_threads") ix->opt_replay_threads = (int)v;
  if (n == "force_dense") ix->opt_force_dense = (int)v;
  if (n == "device_select") ix->opt_device_select = (int)v;  // 0: no shard-local answers, every query replays its lists (the ABI-2 path)
  return BBQ_OK;
}

// the handle's own fields from its shards (every shard decides has_x1 - and with it a fallback to the inline layout - on its own
// rows: the handle reports the widest)
static void adopt_shard_geometry(bbq_index *ix, MultiState *ms) {
  ix->bytes_per_row = 0;
  ix->has_x1 = 0;
  for (const MultiShard &sh : ms->shards) {
    if (sh.ix->bytes_per_row >= ix->bytes_per_row) {
      ix->bytes_per_row = sh.ix->bytes_per_row;
      ix->layout = sh.ix->layout;
      ix->tile_stride = sh.ix->tile_stride;
    }
    ix->has_x1 = ix->has_x1 || sh.ix->has_x1;
  }
}

// a multi-device handle over shard indexes that already exist (loaded from files); takes ownership of them on success
int multi_assemble(bbq_index *const *shards, const int32_t *devices, int32_t n_shards, int32_t dim, int32_t index_bits, int64_t n_rows,
                   double centroid_dp, bbq_index **out) {
  if (n_shards < 1 || n_shards > 64 || !shards || !out) return fail(BBQ_ERR_INVALID_ARG, "multi_assemble: bad arguments");
  std::unique_ptr<bbq_index> ix(new bbq_index());
  ix->dim = dim;
  ix->index_bits = index_bits;
  ix->store_bits = dim == 1 ? 1 : store_bits_of(index_bits);
  ix->pb = row_bytes_of(dim, ix->store_bits);
  ix->w16 = (ix->pb + 15) / 16;
  ix->n_rows = n_rows;
  ix->centroid_dp = centroid_dp;
  ix->device = devices[0];
  std::unique_ptr<MultiState> ms(new MultiState());
  for (int s = 0; s < n_shards; ++s) {
    MultiShard sh;
    sh.ix = shards[s];
    sh.device = devices[s];
    sh.r0 = shards[s]->row_base;
    sh.r1 = sh.r0 + shards[s]->n_rows;
    sh.worker.reset(new ShardWorker());
    ms->shards.push_back(std::move(sh));
  }
  adopt_shard_geometry(ix.get(), ms.get());
  ix->multi = ms.release();
  *out = ix.release();
  return BBQ_OK;
}

// every shard as an ordinary file pair <prefix>.s<NNN> (with its pilot replica) + the manifest <prefix>.vemb
int multi_save(bbq_index *ix, const char *prefix, const float *centroid, int32_t sim) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  std::vector<int64_t> bounds;
  int64_t pilot = 0;
  for (size_t s = 0; s < ms->shards.size(); ++s) {
    MultiShard &sh = ms->shards[s];
    int rc = bbq_index_save(sh.ix, shard_file_prefix(prefix, (int)s).c_str(), centroid, sim);
    if (rc != BBQ_OK) return rc;
    bounds.push_back(sh.r0);
    bounds.push_back(sh.r1 - sh.r0);
    if (sh.ix->has_pilot) pilot = std::max<int64_t>(pilot, sh.ix->pilot.view.n_rows);
  }
  return write_manifest(prefix, (int32_t)ms->shards.size(), bounds.data(), ix->dim, ix->index_bits, sim, ix->n_rows, ix->centroid_dp, pilot, centroid);
}

int multi_get_stats(bbq_index *ix, bbq_stats *out) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  bbq_stats agg = ix->stats;  // candidates / dense_fallbacks / host_replays of the merged answers
  // the shards sweep side by side: the time of a sweep is the slowest shard's, the bytes are everybody's
  for (MultiShard &sh : ms->shards) {
    bbq_stats s{};
    if (bbq_get_stats(sh.ix, &s) != BBQ_OK) continue;
    agg.last_scan_ms = std::max(agg.last_scan_ms, s.last_scan_ms);
    agg.last_scan_rows += s.last_scan_rows;
    agg.last_scan_bytes += s.last_scan_bytes;
    agg.total_scan_ms = std::max(agg.total_scan_ms, s.total_scan_ms);
    agg.total_scan_bytes += s.total_scan_bytes;
    agg.total_scan_launches = std::max(agg.total_scan_launches, s.total_scan_launches);
    agg.resident_bytes += s.resident_bytes;
  }
  *out = agg;
  return BBQ_OK;
}

int multi_reset_stats(bbq_index *ix) {
  MultiState *ms = ix->multi;
  std::lock_guard<std::mutex> lk(ms->mu);
  ix->stats = bbq_stats{};
  for (MultiShard &sh : ms->shards) (void)bbq_reset_stats(sh.ix);
  return BBQ_OK;
}

}  // namespace bbq

extern "C" {

int bbq_index_create_multi(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits, double centroid_dp,
                           int32_t n_shards, const int32_t *devices, int64_t pilot_rows, bbq_index **out) {
  return bbq_index_create_multi_opts(codes, corr, n_rows, dim, index_bits, centroid_dp, n_shards, devices, pilot_rows, nullptr, out);
}

int bbq_index_create_multi_opts(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits, double centroid_dp,
                                int32_t n_shards, const int32_t *devices, int64_t pilot_rows, const bbq_index_options *opts, bbq_index **out) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create_multi: out is null");
  *out = nullptr;
  if (n_rows < 0 || dim <= 0 || pilot_rows < 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create_multi: bad size");
  if (n_rows > 0 && (!codes || !corr)) return fail(BBQ_ERR_INVALID_ARG, "目标向量集合不能为空");
  if (index_bits < 1 || index_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "indexBits必须在1-8之间");
  if (n_shards < 1 || n_shards > 64) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create_multi: n_shards must be in 1..64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  for (int s = 0; s < n_shards; ++s) {
    const int d = devices ? devices[s] : s;
    if (d < 0 || d >= ndev) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range (0..%d)", d, ndev - 1);
  }
  if (n_rows > 0xFFFFFFFFll) return fail(BBQ_ERR_UNSUPPORTED, "more than 2^32 rows");
  if (check_options(opts) != BBQ_OK) return BBQ_ERR_INVALID_ARG;
  // one layout decision for all shards: every shard is created with the explicit value (a shard whose sums are not the popcounts
  // still falls back to inline on its own; bytes_per_row then reports the largest)
  bbq_index_options shard_opts{(int32_t)sizeof(bbq_index_options), want_compact_of(opts) ? BBQ_CORRECTIONS_COMPACT : BBQ_CORRECTIONS_INLINE};

  std::unique_ptr<bbq_index> ix(new bbq_index());
  ix->dim = dim;
  ix->index_bits = index_bits;
  ix->store_bits = dim == 1 ? 1 : store_bits_of(index_bits);
  ix->pb = row_bytes_of(dim, ix->store_bits);
  ix->w16 = (ix->pb + 15) / 16;
  ix->n_rows = n_rows;
  ix->centroid_dp = centroid_dp;
  ix->device = devices ? devices[0] : 0;
  std::unique_ptr<MultiState> ms(new MultiState());
  // contiguous shards of whole 512-row chunks (the last one takes the remainder); shards that would be empty are not created
  const int64_t row_bytes = ix->store_bits == 1 ? ix->pb : dim;  // bytes per row as the caller hands them over
  const int64_t per = std::max<int64_t>(kChunkRows, ((n_rows + n_shards - 1) / n_shards + kChunkRows - 1) / kChunkRows * kChunkRows);
  auto bail = [&](int code) {
    ix->multi = ms.release();
    multi_destroy(ix.release());
    return code;
  };
  for (int s = 0; s < n_shards; ++s) {
    const int64_t r0 = std::min<int64_t>((int64_t)s * per, n_rows), r1 = s == n_shards - 1 ? n_rows : std::min<int64_t>(r0 + per, n_rows);
    if (r1 <= r0 && !(s == 0)) continue;  // shard 0 exists even for an empty index
    MultiShard sh;
    sh.device = devices ? devices[s] : s;
    sh.r0 = r0;
    sh.r1 = r1;
    int64_t P = 0;
    if (r0 > 0) P = pilot_rows >= r0 ? r0 : pilot_rows / kChunkRows * kChunkRows;
    int rc = bbq_index_create_shard_opts(codes ? codes + r0 * row_bytes : nullptr, corr ? corr + r0 * 4 : nullptr, r1 - r0, dim, index_bits, centroid_dp,
                                         r0, P > 0 ? codes : nullptr, P > 0 ? corr : nullptr, P, sh.device, &shard_opts, &sh.ix);
    if (rc != BBQ_OK) return bail(rc);
    sh.worker.reset(new ShardWorker());
    ms->shards.push_back(std::move(sh));
  }
  adopt_shard_geometry(ix.get(), ms.get());
  ix->multi = ms.release();
  *out = ix.release();
  return BBQ_OK;
}

int32_t bbq_index_shards(const bbq_index *ix) { return !ix ? 0 : ix->multi ? (int32_t)ix->multi->shards.size() : 1; }

}  // extern "C"
