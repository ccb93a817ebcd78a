// bbq_core.cpp - device-resident index shard, segment plan, pipelined search, C ABI (see include/bbq.h).
//
// Search of one query = a short sequence of launches over row SEGMENTS of the index:
//   segment 0   rows [0, s0)           dense: every f32 score is written; the finalize kernel lists all of
//                                      them (the reference heap is still filling / changing fast here) and
//                                      selects theta_1 = k-th largest key
//   segment j   rows [b_j, b_{j+1})    sparse: rows with key > theta_j go to per-chunk candidate slots; the
//                                      finalize kernel compacts them into the list and selects theta_{j+1}
// theta_j only depends on rows before b_j, so it is a lower bound of the reference heap's minimum while the
// reference walks segment j: rows at or below it can never enter the heap (bbq_replay.cpp).
// Queries are processed in sub-batches (grid.y = queries, each query sweeps the index on its own), and
// sub-batches are pipelined over NSLOT streams so the host replay of one overlaps the scan of the next.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <atomic>
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

// per query: bit-planes (up to 8) + int8 values in MFMA fragment order + score uniforms + group maxima
int64_t qbuf_bytes_per_query_w(int w16) { return (int64_t)w16 * 8 * 16 + (int64_t)w16 * 128 + (int64_t)sizeof(QueryParams) + 16; }

}  // namespace bbq

namespace {

// Host worker pool for the heap replays: persistent threads (spawning per sub-batch cost more than the replay itself
// once sweeps are shared), fed while the device already works on the next sub-batches.
class ReplayPool {
 public:
  static ReplayPool &get() {
    static ReplayPool *p = new ReplayPool();  // intentionally never destroyed: workers may outlive static destructors
    return *p;
  }
  void ensure(int n) {
    std::lock_guard<std::mutex> lk(m_);
    while ((int)th_.size() < n) th_.emplace_back([this] { run(); });
  }
  void submit(std::function<void()> f) {
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
        cv_.wait(lk, [this] { return !q_.empty(); });
        f = std::move(q_.front());
        q_.pop_front();
      }
      f();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
  std::vector<std::thread> th_;
};

}  // namespace

namespace bbq {

std::mutex g_ctx_mu;
DeviceCtx *g_ctx[64] = {nullptr};

// returns the (lazily created, never destroyed) context of a device; call with hipSetDevice(device) done
int get_ctx(int device, DeviceCtx **out) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (device < 0 || device >= 64) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range", device);
  if (!g_ctx[device]) {
    DeviceCtx *c = new DeviceCtx();
    c->device = device;
    for (int i = 0; i < kMaxSlots; ++i) {
      HIPCHK(hipStreamCreateWithFlags(&c->slots[i].stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreate(&c->slots[i].ev0));
      HIPCHK(hipEventCreate(&c->slots[i].ev1));
      HIPCHK(hipEventCreateWithFlags(&c->slots[i].ev_done, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->slots[i].ev_big, hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void **)&c->d_aux_flags, 4));
    HIPCHK(hipMemset(c->d_aux_flags, 0, 4));
    HIPCHK(hipHostMalloc((void **)&c->h_lat, (size_t)(kLatAnswerOffset + kFinalSelectMax + 8) * 8, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_lat, 0, (size_t)(kLatAnswerOffset + kFinalSelectMax + 8) * 8);
    HIPCHK(hipHostGetDevicePointer((void **)&c->d_lat, c->h_lat, 0));
    HIPCHK(hipMalloc((void **)&c->d_pre_keys, (size_t)kLatPreKeys * 4));
    c->ready = true;
    g_ctx[device] = c;
  }
  *out = g_ctx[device];
  return BBQ_OK;
}

int ensure_aux_qbuf(DeviceCtx *c, int64_t bytes) {
  if (c->aux_qbuf_bytes >= bytes) return BBQ_OK;
  if (c->d_aux_qbuf) HIPCHK(hipFree(c->d_aux_qbuf));
  c->d_aux_qbuf = nullptr;
  HIPCHK(hipMalloc((void **)&c->d_aux_qbuf, (size_t)bytes));
  c->aux_qbuf_bytes = bytes;
  return BBQ_OK;
}

}  // namespace bbq

namespace {

// ------------------------------------------------------------------------------------------------ storage

int make_storage(bbq_index *ix, Storage &st, const uint8_t *codes, const double *corr, int64_t n_rows, int64_t row_id_base,
                 bool check_x1) {
  const int64_t pb = ix->store_bits > 1 ? ix->dim : ix->pb;  // bytes per row as the caller hands them over (multi-bit: one byte per dimension)
  DevMem m_codes, m_corr;
  hipStream_t s = ix->aux_stream;
  if (n_rows > 0) {
    HIPCHK(m_codes.alloc((size_t)(n_rows * pb)));
    HIPCHK(m_corr.alloc((size_t)n_rows * 32));
    HIPCHK(hipMemcpyAsync(m_codes.p, codes, (size_t)(n_rows * pb), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m_corr.p, corr, (size_t)n_rows * 32, hipMemcpyHostToDevice, s));
  }
  return storage_from_device_rows(ix, st, m_codes.as<uint8_t>(), m_corr.as<double>(), n_rows, row_id_base, check_x1);  // synchronises before the scratch rows go
}

}  // namespace

namespace bbq {

// rows already in device memory (codes in the caller's shape, corrections [n][4]) -> tile records of `st`; decides the layout of
// the index on the way (check_x1).  Returns after the device work has completed.
int storage_from_device_rows(bbq_index *ix, Storage &st, const uint8_t *d_codes, const double *d_corr, int64_t n_rows, int64_t row_id_base,
                             bool check_x1) {
  const int64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  const bool multibit = ix->store_bits > 1;
  const int64_t pb = multibit ? ix->dim : ix->pb;
  DevMem m_mis;
  hipStream_t s = ix->aux_stream;
  if (check_x1) {
    // quantizedComponentSum of a 1-bit row is its popcount (src/optimizedScalarQuantizer.ts:204-209); if that
    // holds for every row the 8 bytes need not be stored or read.  Decided once per index, over all storages.
    uint32_t mis = 0;
    HIPCHK(m_mis.alloc(4));
    uint32_t *d_mis = m_mis.as<uint32_t>();
    HIPCHK(hipMemsetAsync(d_mis, 0, 4, s));
    if (multibit) HIPCHK(launch_check_x1_multibit(d_codes, d_corr, n_rows, ix->dim, d_mis, s));
    else HIPCHK(launch_check_x1(d_codes, d_corr, n_rows, (int32_t)pb, d_mis, s));
    HIPCHK(hipMemcpyAsync(&mis, d_mis, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (mis) ix->has_x1 = 1;
  }
  // compact corrections (4 B/row streamed + exact and add-range side arrays) need the implicit component sum; otherwise inline
  ix->layout = (ix->want_compact && !ix->has_x1) ? kLayoutCompact : kLayoutInline;
  ix->tile_stride = tile_stride_of(ix->w16, ix->layout, ix->has_x1);
  ix->bytes_per_row = ix->tile_stride / kTileRows;
  st.row_id_base = row_id_base;
  st.view.n_rows = n_rows;
  st.view.w16 = ix->w16;
  st.view.tile_stride = ix->tile_stride;
  st.view.has_x1 = ix->has_x1;
  st.view.dim = ix->dim;
  st.view.layout = ix->layout;
  st.view.store_bits = ix->store_bits;
  if (n_tiles > 0) {
    HIPCHK(hipMalloc((void **)&st.d_tiles, (size_t)(n_tiles * ix->tile_stride)));
    if (ix->layout == kLayoutCompact) HIPCHK(hipMalloc((void **)&st.d_exact, (size_t)compact_side_bytes(n_tiles)));
    if (multibit) {
      uint32_t bad = 0;
      if (!m_mis.p) HIPCHK(m_mis.alloc(4));
      HIPCHK(hipMemsetAsync(m_mis.p, 0, 4, s));
      HIPCHK(launch_retile_multibit(d_codes, d_corr, n_rows, ix->dim, ix->store_bits, ix->index_bits, st.d_tiles, ix->w16, ix->tile_stride, ix->has_x1, ix->layout,
                                    st.d_exact, m_mis.as<uint32_t>(), s));
      HIPCHK(hipMemcpyAsync(&bad, m_mis.p, 4, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      if (bad) return fail(BBQ_ERR_INVALID_ARG, "indexBits=%d: a quantized value is not below %d", ix->index_bits, 1 << ix->index_bits);
    } else {
      HIPCHK(launch_retile(d_codes, d_corr, n_rows, (int32_t)pb, st.d_tiles, ix->w16, ix->tile_stride, ix->has_x1, ix->layout, st.d_exact, s));
    }
    if (ix->layout == kLayoutCompact) HIPCHK(launch_tile_add_range(st.d_exact, n_rows, const_cast<float *>(add_range_of(st.d_exact, n_tiles)), s));
    HIPCHK(hipStreamSynchronize(s));
  }
  st.view.exact = st.d_exact;
  st.view.add_range = add_range_of(st.d_exact, n_tiles);
  st.view.tiles = st.d_tiles;
  HIPCHK(hipStreamSynchronize(s));  // the scratch rows are released on return
  return BBQ_OK;
}

}  // namespace bbq

namespace {

// The view a launch gets: the stored view + which chunks it loads cache-resident.  The indexes that have launched sweeps on the device
// lately (kCacheWindow) share its 256 MiB Infinity Cache in proportion to their sizes (resident_mb >= 0: that many MiB per launch,
// whatever else is there).  Called with the device context locked.
constexpr int64_t kResidentAutoBytes = 224ll << 20;  // per launch; measured: 192 / 224 / 240 MiB within noise of each other at 10 M x 768, a resident set of 248 MiB gains nothing
constexpr uint64_t kCacheWindow = 100'000'000;  // ns: an index that has launched nothing for 0.1 s is not competing for the cache
static int64_t cache_sharers_bytes(bbq_index *ix, int64_t own) {
  DeviceCtx *c = ix->ctx;
  if (!c) return own;
  const uint64_t now = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  int64_t all = 0;
  bool found = false;
  for (size_t i = 0; i < c->cache_users.size();) {
    DeviceCtx::CacheUser &u = c->cache_users[i];
    if (u.index == ix) { u.bytes = own; u.tick = now; found = true; }
    if (now - u.tick > kCacheWindow) { c->cache_users.erase(c->cache_users.begin() + (long)i); continue; }
    all += u.bytes;
    ++i;
  }
  if (!found) { c->cache_users.push_back({ix, own, now}); all += own; }
  return all;
}
// One launch sweeps chunks [chunk_begin, chunk_begin + n_chunks) of `sto` once per query of its sub-batch, back to back: what it can
// keep in the cache is a part of ITS range (the launches of a sub-batch run one after the other, each over its own rows).
static IndexView launch_view(bbq_index *ix, const Storage &sto, int64_t chunk_begin = 0, int64_t n_chunks = -1) {
  IndexView v = sto.view;
  const int64_t all_chunks = sto.n_chunks();
  if (n_chunks < 0) n_chunks = all_chunks - chunk_begin;
  const int64_t chunk_bytes = (int64_t)kTilesPerChunk * v.tile_stride;
  const int64_t own = ix->main.n_chunks() * (int64_t)kTilesPerChunk * ix->main.view.tile_stride;
  const int64_t all = std::max<int64_t>(1, cache_sharers_bytes(ix, own));
  const int64_t budget = ix->opt_resident_mb >= 0 ? ((int64_t)ix->opt_resident_mb << 20)
                                            : (int64_t)((double)kResidentAutoBytes * ((double)own / (double)all));
  int64_t fit = std::min(n_chunks, budget / std::max<int64_t>(1, chunk_bytes));  // chunks of this launch's range that stay resident
  // an index only a little larger than the budget: its launches are short and overlap (the small ones of the next sub-batch run beside
  // the large one), so their resident sets must fit TOGETHER - the same share of every launch's range
  if (own > budget && own <= budget + budget / 4 && ix->opt_resident_mb < 0) fit = std::min(n_chunks, (int64_t)((double)n_chunks * (double)budget / (double)own));
  int64_t resident_chunks;
  if (fit >= n_chunks) {  // everything this launch reads
    v.resident_share = -1;
    v.resident_tiles = (chunk_begin + n_chunks) * kTilesPerChunk;
    resident_chunks = n_chunks;
  } else if (ix->opt_resident_interleave && n_chunks >= 64) {  // of every 64 chunks the first `share`: cache and HBM deliver side by side
    v.resident_share = fit * 64 / n_chunks;  // rounded down: never more than the budget
    v.resident_tiles = 0;
    resident_chunks = n_chunks * v.resident_share / 64;
  } else {  // the first chunks of the range
    v.resident_share = -1;
    v.resident_tiles = (chunk_begin + fit) * kTilesPerChunk;
    resident_chunks = fit;
  }
  ix->sweep_resident_acc += resident_chunks * chunk_bytes;  // the caller books it per sweep of the index (bbq_stats.resident_bytes)
  return v;
}

// queries per launch sequence (sub-batch).  The largest sweep of a sub-batch should run for about a millisecond: shorter ones pay the
// device's dependent-launch gaps and their own ramp (1.25 M rows x 2048 queries: 52.5 K q/s with 32 per sub-batch, 56.5 K with 64,
// 57 K with 96-128; at 10 M rows 32 is as good as 64 and needs half the workspace).  A call should also be cut into at least four
// sub-batches where it can: the first sub-batch's small segments run alone on the device and only the later ones hide theirs behind
// another sub-batch's large sweep (1 M rows x 256 queries per call: 0.849 of the roofline end to end with 2 x 128, 0.855 with 4 x 64)
int effective_batch(const bbq_index *ix, int64_t n_queries = 0) {
  if (ix->opt_batch > 0) return ix->opt_batch;
  const int64_t rows = ix->main.view.n_rows;
  int q = rows >= 6000000 ? 32 : rows >= 2500000 ? 64 : 128;
  while (n_queries > 0 && q > 32 && n_queries < 4 * (int64_t)q) q >>= 1;
  // the sweep on the matrix cores serves two groups of 32 queries per tile load (bbq_mfma_kernels.hip): 64 queries per launch chain
  if (ix->opt_share == 32 && q < 64 && (n_queries == 0 || n_queries > 32)) q = 64;
  return q;
}

// ------------------------------------------------------------------------------------------------ plan

int cap_for(int64_t k, int64_t rows_before) {
  const double lam = (double)k * kChunkRows / (double)std::max<int64_t>(rows_before, 1);
  int64_t c = (int64_t)ceil(lam + 8.0 * sqrt(lam) + 16.0);
  c = (c + 7) / 8 * 8;
  return (int)std::min<int64_t>(std::max<int64_t>(c, 16), kChunkRows);
}

// have_theta: a threshold derived from earlier rows (the pilot replica) already exists when this storage starts;
// otherwise the storage's own first rows form the dense segment (for a shard without a replica that gives
// thresholds from the shard's local prefix: weaker than global ones, still valid)
void add_storage_segments(const bbq_index *ix, Plan &p, int storage, const Storage &st, int64_t rows_before, bool have_theta,
                          bool emit, double &expected) {
  const int64_t R = st.view.n_rows;
  if (R <= 0) return;
  int64_t b = 0;
  if (!have_theta) {
    rows_before = 0;
    const int64_t rows = std::min(p.s0, R);
    p.segs.push_back(Segment{storage, 0, (rows + kChunkRows - 1) / kChunkRows, rows, true, emit, true, false, 0});
    expected += (double)rows;
    b = rows;
  }
  while (b < R) {
    // the threshold of a segment comes from every row seen before it (pilot replica + this storage's earlier
    // segments); the next boundary multiplies that count by `growth`, so every segment emits ~k*(growth-1) candidates
    const int64_t before = rows_before + b;
    int64_t e = R;
    const int64_t nb = (before * p.growth - rows_before) / kChunkRows * kChunkRows;
    if (before > 0 && nb > b && nb <= R / 2) e = nb;
    const int cap = cap_for(p.k, before);
    const int64_t rows = e - b;
    p.segs.push_back(Segment{storage, b / kChunkRows, (rows + kChunkRows - 1) / kChunkRows, rows, false, emit, true, false, cap});
    expected += (double)p.k * (double)rows / (double)before;
    b = e;
  }
}

// k: the rank the device selects thresholds with; final_k > 0: k == final_k + 1 and the last finalize launch selects the answer
void build_plan(bbq_index *ix, int64_t k, int64_t final_k = 0, bool latency = false) {
  Plan &p = ix->plan;
  const int growth = latency ? ix->opt_latency_growth : ix->opt_growth;
  if (p.k == k && p.final_k == final_k && p.growth == growth && p.latency == latency) return;
  p = Plan();
  p.k = k;
  p.final_k = final_k;
  p.growth = growth;
  p.latency = latency;
  p.s0 = std::max<int64_t>(ix->opt_s0, (4 * k + kChunkRows - 1) / kChunkRows * kChunkRows);
  p.s0 = std::min<int64_t>(p.s0, 8192);
  double expected_emit = 0, dummy = 0;
  if (ix->has_pilot) {
    add_storage_segments(ix, p, 0, ix->pilot, 0, false, false, dummy);
    add_storage_segments(ix, p, 1, ix->main, ix->pilot.view.n_rows, true, true, expected_emit);
  } else {
    add_storage_segments(ix, p, 1, ix->main, 0, false, true, expected_emit);
  }
  if (!p.segs.empty()) p.segs.back().need_theta = false;
  int64_t best = -1;
  for (size_t i = 0; i < p.segs.size(); ++i) {
    Segment &s = p.segs[i];
    if (!s.dense) {
      p.max_slots = std::max(p.max_slots, s.n_chunks * (int64_t)s.cap);
      p.max_chunks = std::max(p.max_chunks, s.n_chunks);
    }
    if (best < 0 || s.rows > p.segs[best].rows) best = (int64_t)i;
  }
  if (best >= 0) p.segs[best].dominant = true;
  for (Segment &sg : p.segs) sg.big = best >= 0 && sg.rows * 32 >= p.segs[best].rows && sg.rows >= 65536;
  // list capacity: everything dense + 4x the expected sparse candidates + slack
  double sparse = 0;
  int64_t dense_rows = 0;
  for (const Segment &s : p.segs)
    if (s.emit && s.dense) dense_rows += s.rows;
  sparse = expected_emit - (double)dense_rows;
  if (sparse < 0) sparse = 0;
  p.list_cap = dense_rows + (int64_t)(4.0 * sparse) + 4096;
  p.list_cap = (p.list_cap + 1023) / 1024 * 1024;
  // per query; bounded so that the overflow areas of one pipeline slot stay within 512 MB however many queries a sub-batch has
  p.flood_cap = std::min<int64_t>(ix->opt_flood, (ix->main.view.n_rows + 1023) / 1024 * 1024);
  if (p.flood_cap > 0)
    p.flood_cap = std::min<int64_t>(p.flood_cap, std::max<int64_t>(16384, ((64ll << 20) / std::max(32, effective_batch(ix))) / 1024 * 1024));
}

// ------------------------------------------------------------------------------------------------ slots

void free_slot_buffers(Slot &s) {
  if (s.d_block) (void)hipFree(s.d_block);
  if (s.h_block) (void)hipHostFree(s.h_block);
  s.d_block = s.h_block = nullptr;
  if (s.d_counts) (void)hipFree(s.d_counts);
  if (s.d_topk) (void)hipFree(s.d_topk);
  if (s.h_list_counts) (void)hipHostFree(s.h_list_counts);
  if (s.d_entries) (void)hipFree(s.d_entries);
  if (s.d_lists) (void)hipFree(s.d_lists);
  if (s.h_lists) (void)hipHostFree(s.h_lists);
  if (s.d_dense0) (void)hipFree(s.d_dense0);
  if (s.d_ovf) (void)hipFree(s.d_ovf);
  if (s.d_final) (void)hipFree(s.d_final);
  if (s.h_final) (void)hipHostFree(s.h_final);
  s.d_final = s.h_final = nullptr;
  s.final_stride = 0;
  s.d_ovf = nullptr;
  s.d_ovf_counts = nullptr;
  s.d_qbuf = s.h_qbuf = nullptr;
  s.d_theta = s.d_flags = s.d_counts = s.d_topk = nullptr;
  s.d_topk_counts = s.d_list_counts = s.h_list_counts = nullptr;
  s.d_entries = s.d_lists = s.h_lists = nullptr;
  s.d_dense0 = nullptr;
  s.q_cap = 0;
}

int64_t qbuf_bytes_per_query(const bbq_index *ix) { return qbuf_bytes_per_query_w(ix->w16); }

int ensure_slot(bbq_index *ix, Slot &s, int nq, bool own_lists) {
  const Plan &p = ix->plan;
  const int64_t qb = qbuf_bytes_per_query(ix);
  const int64_t hprefix = std::min<int64_t>(p.list_cap, 16384);
  const bool ok = s.q_cap >= nq && s.qbuf_bytes >= qb && s.chunks_cap >= p.max_chunks && s.slots_cap >= p.max_slots &&
                  s.dense_cap >= p.s0 && s.list_cap >= p.list_cap + (own_lists ? p.flood_cap : 0) && s.k_cap >= p.k && s.hprefix >= hprefix &&
                  s.flood_cap >= p.flood_cap && s.final_stride >= p.final_k + 2 &&
                  (!own_lists || s.d_lists != nullptr);
  if (ok) return BBQ_OK;
  // grow-only in every dimension: a process that alternates between call shapes (single queries use another segment plan than
  // batches, k varies) would otherwise free and allocate the workspace at every switch - each time sized for the plan at hand and
  // therefore too small for the other one (milliseconds per switch)
  const int old_q = s.q_cap;
  const int64_t old_final = s.final_stride;
  const bool had_lists = s.d_lists != nullptr;
  free_slot_buffers(s);
  own_lists = own_lists || had_lists;
  const int Q = std::max((std::max(nq, effective_batch(ix)) + 31) / 32 * 32, old_q);  // multiple of 32: the MFMA query layout is per group of 32
  s.qbuf_bytes = std::max(s.qbuf_bytes, qb);
  s.chunks_cap = std::max<int64_t>(s.chunks_cap, std::max<int64_t>(p.max_chunks, 1));
  s.slots_cap = std::max<int64_t>(s.slots_cap, std::max<int64_t>(p.max_slots, 1));
  s.dense_cap = std::max<int64_t>(s.dense_cap, p.s0);
  s.flood_cap = std::max<int64_t>(s.flood_cap, p.flood_cap);
  s.list_cap = std::max<int64_t>(s.list_cap, p.list_cap + (own_lists ? s.flood_cap : 0));  // own lists can take a flood; external ones are the caller's size
  s.k_cap = std::max<int64_t>(s.k_cap, std::max<int64_t>(p.k, 1));
  s.hprefix = std::max<int64_t>(s.hprefix, hprefix);
  // theta | flags | topk_counts | list_counts | ovf_counts | append_counts live in one control block in front of the staged queries: the copy that
  // brings a sub-batch's queries also resets them (the host twin's control part stays zero)
  s.ctrl_bytes = ((int64_t)Q * (24 + 4 * kAppendStride) + 255) / 256 * 256;  // the append counters sit one per 128-byte line (kAppendStride)
  HIPCHK(hipMalloc((void **)&s.d_block, (size_t)(s.ctrl_bytes + Q * s.qbuf_bytes)));
  s.ctrl_clean = false;
  HIPCHK(hipHostMalloc((void **)&s.h_block, (size_t)(s.ctrl_bytes + Q * s.qbuf_bytes), hipHostMallocDefault));
  memset(s.h_block, 0, (size_t)s.ctrl_bytes);
  s.d_qbuf = s.d_block + s.ctrl_bytes;
  s.h_qbuf = s.h_block + s.ctrl_bytes;
  s.d_theta = reinterpret_cast<uint32_t *>(s.d_block);
  s.d_flags = s.d_theta + Q;
  s.d_topk_counts = reinterpret_cast<int32_t *>(s.d_theta + 2 * (size_t)Q);
  s.d_list_counts = reinterpret_cast<int32_t *>(s.d_theta + 3 * (size_t)Q);
  s.d_ovf_counts = s.d_theta + 5 * (size_t)Q;
  s.d_append_counts = s.d_theta + 6 * (size_t)Q;
  if (s.flood_cap > 0) HIPCHK(hipMalloc((void **)&s.d_ovf, (size_t)(Q * s.flood_cap) * 8));
  HIPCHK(hipMalloc((void **)&s.d_counts, (size_t)(Q * s.chunks_cap) * 4));
  HIPCHK(hipMalloc((void **)&s.d_topk, (size_t)(Q * s.k_cap) * 4));
  HIPCHK(hipHostMalloc((void **)&s.h_list_counts, (size_t)Q * 8, hipHostMallocDefault));
  HIPCHK(hipMalloc((void **)&s.d_entries, (size_t)(Q * s.slots_cap) * 8));
  HIPCHK(hipMalloc((void **)&s.d_dense0, (size_t)(Q * s.dense_cap) * 4));
  if (own_lists) {
    HIPCHK(hipMalloc((void **)&s.d_lists, (size_t)(Q * s.list_cap) * 8));
    HIPCHK(hipHostMalloc((void **)&s.h_lists, (size_t)(Q * s.hprefix) * 8, hipHostMallocDefault));
  }
  s.final_stride = std::max<int64_t>(old_final, std::max<int64_t>(p.final_k, 126) + 2);
  HIPCHK(hipMalloc((void **)&s.d_final, (size_t)(Q * s.final_stride) * 8));
  HIPCHK(hipHostMalloc((void **)&s.h_final, (size_t)(Q * s.final_stride) * 8, hipHostMallocDefault));
  s.q_cap = Q;
  return BBQ_OK;
}

// ------------------------------------------------------------------------------------------------ query prep

struct PreparedQueries {
  int planes = 4;
  int one_bit = 0;
};

int max_value(const uint8_t *q, int64_t count) {
  uint8_t m = 0;
  for (int64_t i = 0; i < count; ++i) m = std::max(m, q[i]);
  return m;
}

int planes_for(const uint8_t *q, int64_t count) {
  uint8_t m = 0;
  for (int64_t i = 0; i < count; ++i) m |= q[i];
  if (m <= 1) return 1;
  if (m <= 3) return 2;
  if (m <= 15) return 4;
  return 8;
}

// bytes of staged query data per query (bit-planes, or the nibble / byte dwords of a multi-bit index)
int64_t query_data_bytes(const bbq_index *ix, int planes) { return (int64_t)ix->w16 * query_units_per_chunk(planes, ix->store_bits) * 16; }

// kernel variant for a call: 1-bit index -> number of bit-planes the query values need; multi-bit index -> 4 (values <= 15: low
// nibbles only) or 8
int planes_of_call(const bbq_index *ix, const uint8_t *q, int64_t count, int one_bit) {
  if (ix->store_bits == 1) return one_bit ? 1 : planes_for(q, count);
  if (ix->store_bits == 8) return 8;
  return max_value(q, count) <= 15 ? 4 : 8;
}

// writes the query data and the score uniforms of one query into the staging buffer.
// 1-bit index: bit-planes ([j][p] 16-byte blocks, packed like the rows: dim d -> byte d>>3, bit 7-(d&7)).
// multi-bit index: per row dword w the dwords the kernel multiplies its unfolded fields with (dot_chunk_multibit):
//   store_bits 2: {lo nibbles of dims 16w+0,2,..,14 | lo nibbles of dims 16w+1,3,..,15 [| hi nibbles of the same, planes == 8]}
//   store_bits 4: {lo nibbles of dims 8w..8w+7 [| hi nibbles]}          store_bits 8: {bytes of dims 4w..4w+3}
void fill_query(const bbq_index *ix, uint8_t *planes_dst, QueryParams *pp, const uint8_t *q, const double *qc, int planes,
                int one_bit, int sim) {
  memset(planes_dst, 0, (size_t)query_data_bytes(ix, planes));
  if (ix->store_bits == 1) {
    // eight dimensions (one byte of every plane) at a time: bit p of the eight query bytes, gathered MSB-first by one multiply -
    // source bit 8i (dimension 8*byte + i) goes to bit 63 - i of the product, no two partial products share a position
    const int full = ix->dim >> 3;
    for (int byte = 0; byte < full; ++byte) {
      uint64_t x;
      memcpy(&x, q + (size_t)byte * 8, 8);
      if (!x) continue;
      const int j = byte >> 4, b = byte & 15;
      for (int p = 0; p < planes; ++p)
        planes_dst[((size_t)j * planes + p) * 16 + b] = (uint8_t)((((x >> p) & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);
    }
    for (int d = full * 8; d < ix->dim; ++d) {  // the last, partial byte
      const uint8_t v = q[d];
      if (!v) continue;
      const int byte = d >> 3, j = byte >> 4, b = byte & 15;
      const uint8_t bit = (uint8_t)(0x80u >> (d & 7));
      for (int p = 0; p < planes; ++p)
        if ((v >> p) & 1) planes_dst[((size_t)j * planes + p) * 16 + b] |= bit;
    }
  } else {
    const int sb = ix->store_bits, per = 32 / sb, qn = query_units_per_chunk(planes, sb);
    uint32_t *dst = reinterpret_cast<uint32_t *>(planes_dst);
    for (int d = 0; d < ix->dim; ++d) {
      const uint32_t v = q[d];
      if (!v) continue;
      const int w = d / per, f = d % per;
      uint32_t *qw = dst + (size_t)w * qn;
      if (sb == 2) {
        const int half = f & 1, nib = f >> 1;
        qw[half] |= (v & 15u) << (4 * nib);
        if (planes > 4) qw[2 + half] |= (v >> 4) << (4 * nib);
      } else if (sb == 4) {
        qw[0] |= (v & 15u) << (4 * f);
        if (planes > 4) qw[1] |= (v >> 4) << (4 * f);
      } else {
        qw[0] |= v << (8 * f);
      }
    }
  }
  const double FBS = 1.0 / 15.0;  // src/constants.ts:20
  pp->ay = qc[0];
  pp->ly = one_bit ? (qc[1] - qc[0]) : (qc[1] - qc[0]) * FBS;  // src/batchDotProduct.ts:498 / :574
  pp->y1 = qc[3];
  pp->qadd = qc[2];
  // multi-bit index: the reference's batch scorer throws on unpacked rows and its per-row scorer answers
  // (src/binaryQuantizedScorer.ts:403-419): centroidDP is 0 for every query width but 1 (searchNearestNeighbors passes no
  // original query, :290) and MAXIMUM_INNER_PRODUCT is not divided by FOUR_BIT_SCALE (:207-209)
  const bool per_row_form = ix->store_bits > 1;
  pp->cdp = (per_row_form && !one_bit) ? 0.0 : ix->centroid_dp;
  pp->dimd = (double)ix->dim;
  pp->sim = sim;
  pp->one_bit = one_bit;
  pp->mip_plain = per_row_form ? 1 : 0;
  pp->pad_ = 0;
}

// MFMA shared sweep, int8 form (query values up to 127): the int8 query values in the order the code bits fall out of the packed
// words.  For 32-dim word g, half h, dword c, byte i the kernel extracts bit p = 4h + c + 8i of the little-endian word, which is row
// byte 4g + (p >> 3), bit (p & 7), i.e. dimension 32g + 8*(p >> 3) + 7 - (p & 7) (MSB-first packing,
// src/optimizedScalarQuantizer.ts:420-446).  Layout: [group][g][h][n][16 B], n = query in its group of 32.
void fill_query_mfma(const bbq_index *ix, uint8_t *dst, int q_in_batch, const uint8_t *q) {
  const int words = ix->w16 * 4, group = q_in_batch / 32, n = q_in_batch % 32;
  uint8_t *gb = dst + (size_t)group * mfma_query_bytes_per_group(ix->w16, false);
  for (int g = 0; g < words; ++g)
    for (int h = 0; h < 2; ++h) {
      uint8_t *o = gb + (((size_t)g * 2 + h) * 32 + n) * 16;
      for (int cc = 0; cc < 4; ++cc)
        for (int i = 0; i < 4; ++i) {
          const int p = 4 * h + cc + 8 * i;
          const int d = 32 * g + 8 * (p >> 3) + 7 - (p & 7);
          o[4 * cc + i] = d < ix->dim ? q[d] : 0;
        }
    }
}

// FP form (query values <= 15): v_mfma_f32_32x32x64_f8f6f4 with the rows as FP4 and the queries as FP6 (e2m3).  Step g covers the
// code words 2g (lower half-wave) and 2g + 1 (upper); element i of a lane is bit p = 4 (i & 7) + (i >> 3) of its word - the kernel
// masks bit c = i >> 3 of every nibble where it stands, an FP4 number of 0.5, 1.0, 2.0 and (bit 3, shifted down) 0.5 - and the query
// value carries 1/2, 1/4, 1/8, 1/2 against it: every product is q / 4 (q / 2 or q for smaller query values, see below).  e2m3 holds q / 2, q / 4 and q / 8 exactly for q <= 15
// (exponent 0: m / 8; exponent e: (1 + m / 8) 2^(e-1)).  A lane's 32 six-bit codes are 24 bytes: the first 16 in [g][h][n][16 B],
// the last 8 in [g][h][n][8 B] behind all of them (two aligned LDS reads per lane and step).
static uint32_t fp6_code_of_eighths(int eighths) {
  if (eighths < 8) return (uint32_t)eighths;                   // exponent field 0: m / 8
  int e = 1;
  while (eighths >= (8 << e)) ++e;                             // 2^(e-1) <= value < 2^e
  return ((uint32_t)e << 3) | (uint32_t)((eighths >> (e - 1)) - 8);   // exact: the low e - 1 bits of eighths are zero for q <= 15
}

void fill_query_mfma_fp(const bbq_index *ix, uint8_t *dst, int q_in_batch, const uint8_t *q, int scale8) {
  const int steps = ix->w16 * 2, group = q_in_batch / 32, n = q_in_batch % 32;
  uint8_t *gb = dst + (size_t)group * mfma_query_bytes_per_group(ix->w16, true);
  uint8_t *gb2 = gb + (size_t)steps * 2 * 32 * 16;
  // the value x 8: q/2, q/4, q/8, q/2 at scale8 = 2 (products q/4: values up to 15); twice that for values up to 7 (scale8 = 4,
  // products q/2), four times for values up to 3 (scale8 = 8, products q): the finest grain e2m3's range (7.5) allows
  uint8_t lut[4][16];
  for (int cls = 0; cls < 4; ++cls)
    for (int v = 0; v < 16; ++v) lut[cls][v] = (uint8_t)fp6_code_of_eighths(v * (cls == 1 ? 2 : cls == 2 ? 1 : 4) * (scale8 / 2));
  // element i = 8 cls + j of a lane is bit p = 4 j + cls of its word: row byte j >> 1, bit 4 (j & 1) + cls, i.e. dimension
  // 8 (j >> 1) + 7 - 4 (j & 1) - cls of the word's 32.  The eight six-bit codes of a class are 48 bits: bytes [6 cls, 6 cls + 6)
  static const int off[8] = {7, 3, 15, 11, 23, 19, 31, 27};
  for (int g = 0; g < steps; ++g)
    for (int h = 0; h < 2; ++h) {
      const int base = 32 * (2 * g + h);
      uint8_t bits[24];
      for (int cls = 0; cls < 4; ++cls) {
        uint64_t v48 = 0;
        if (base + 32 <= ix->dim) {
          for (int j = 0; j < 8; ++j) v48 |= (uint64_t)lut[cls][q[base + off[j] - cls] & 15] << (6 * j);
        } else {
          for (int j = 0; j < 8; ++j) {
            const int d = base + off[j] - cls;
            v48 |= (uint64_t)lut[cls][d < ix->dim ? (q[d] & 15) : 0] << (6 * j);
          }
        }
        for (int b = 0; b < 6; ++b) bits[6 * cls + b] = (uint8_t)(v48 >> (8 * b));
      }
      memcpy(gb + (((size_t)g * 2 + h) * 32 + n) * 16, bits, 16);
      memcpy(gb2 + (((size_t)g * 2 + h) * 32 + n) * 8, bits + 16, 8);
    }
}

// The matrix-core sweep tests "score > threshold" as an inequality on the integer dot product (bbq_mfma_kernels.hip), which divides by
// the query's interval width: it takes queries with a positive, finite width and finite corrections; any other sub-batch sweeps on
// the vector ALUs.
bool mfma_query_ok(const QueryParams &p) {
  return p.ly > 0.0 && p.ly < 1e30 && 1.0 / p.ly < 1e30 && fabs(p.ay) < 1e30 && fabs(p.y1) < 1e30 && fabs(p.qadd) < 1e30 && fabs(p.cdp) < 1e30;
}

int validate_query_args(const bbq_index *ix, int32_t nq, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                        int32_t sim, int64_t k, bool values_pending = false) {
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "目标向量集合不能为空");
  if (nq < 0) return fail(BBQ_ERR_INVALID_ARG, "n_queries < 0");
  if (nq > 0 && (!qquant || !qcorr)) return fail(BBQ_ERR_INVALID_ARG, "查询向量不能为空");
  if (k < 0) return fail(BBQ_ERR_NEGATIVE_K, "k值不能为负数");
  if (query_bits < 1 || query_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "queryBits必须在1-8之间");
  if (sim < 0 || sim > 2) return fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", sim);
  if (query_bits == 1 && !values_pending)  // (values_pending: the library's own quantizer is still producing them)
    for (int64_t i = 0; i < (int64_t)nq * ix->dim; ++i)
      if (qquant[i] > 1) return fail(BBQ_ERR_INVALID_ARG, "1位量化值必须为0或1");
  return BBQ_OK;
}

// Raw queries of bbq_search_raw_batch being quantized on host threads, chunk by chunk, while the sub-batches in front are already
// on the device: the quantizer (~15 us per 768-d query and core) never stands in front of a sweep except for the first chunk.
struct RawFeed {
  const float *queries = nullptr, *centroid = nullptr;
  int32_t n = 0, dim = 0, sim = 0, qb = 0, iters = 0, chunk = 8;  // small chunks: the first sub-batch waits for its own queries only (8 x ~15 us)
  double lambda = 0;
  uint8_t *qq = nullptr;
  double *qc = nullptr;
  std::unique_ptr<std::atomic<int>[]> ready;  // per chunk: 0 pending, 1 done, 2 failed
  std::atomic<int> next{0};
  std::vector<std::thread> threads;
  int n_chunks() const { return (n + chunk - 1) / chunk; }
  void start(int n_threads) {
    ready.reset(new std::atomic<int>[(size_t)n_chunks()]);
    for (int i = 0; i < n_chunks(); ++i) ready[(size_t)i].store(0);
    const int T = std::max(1, std::min(n_threads, n_chunks()));
    auto work = [this] {
      for (;;) {
        const int ci = next.fetch_add(1);
        if (ci >= n_chunks()) return;
        int state = 1;
        for (int i = ci * chunk; i < std::min(n, (ci + 1) * chunk); ++i)
          if (bbq_quantize_query(queries + (size_t)i * dim, dim, centroid, sim, qb, lambda, iters, qq + (size_t)i * dim, qc + (size_t)i * 4) != BBQ_OK) {
            state = 2;
            break;
          }
        ready[(size_t)ci].store(state, std::memory_order_release);
      }
    };
    // one chunk (the reference's call shape: ONE query per call): on the calling thread.  Creating a thread per call cost more than the
    // quantization (~15 us) and, now and then, a millisecond: the 0.8 ms p99 of a 0.2 ms call through the N-API host
    if (n_chunks() <= 1) { work(); return; }
    for (int t = 0; t < T; ++t) threads.emplace_back(work);
  }
  // blocks until queries [first, first + count) are quantized; on a failed query returns its index through *bad
  int wait(int64_t first, int count, int32_t *bad) {
    for (int ci = (int)(first / chunk); ci <= (int)((first + count - 1) / chunk); ++ci) {
      int st;
      while ((st = ready[(size_t)ci].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      if (st == 2) {
        // which query, and its message on THIS thread (the worker's is thread-local)
        for (int i = ci * chunk; i < std::min(n, (ci + 1) * chunk); ++i) {
          const int rc = bbq_quantize_query(queries + (size_t)i * dim, dim, centroid, sim, qb, lambda, iters, qq + (size_t)i * dim, qc + (size_t)i * 4);
          if (rc != BBQ_OK) { if (bad) *bad = i; return rc; }
        }
        return fail(BBQ_ERR_INVALID_ARG, "query quantization failed");
      }
    }
    return BBQ_OK;
  }
  void join() {
    next.store(1 << 30);
    for (auto &t : threads) t.join();
    threads.clear();
  }
  ~RawFeed() { join(); }
};

// ------------------------------------------------------------------------------------------------ enqueue / complete

struct BatchCtx {
  bbq_index *ix;
  const uint8_t *qquant;
  const double *qcorr;
  int planes, one_bit, sim;
  int64_t k;
  int maxq = 255;  // largest quantized query value of the call (the MFMA sweep needs <= 127)
};

// outputs of a sharded scan: the per-query lists live in the index's own buffers and (optionally) the shard-local answers go straight
// into the caller's device memory (bbq_shard_scan_begin)
struct ExtOut {
  uint64_t *lists = nullptr;     // [nq of the batch][list_cap], this sub-batch's first row
  int64_t list_cap = 0;
  int32_t *counts = nullptr;     // [nq][2]
  uint64_t *answers = nullptr;   // this sub-batch's first row of the caller's [n_queries][answers_stride], or null
  int64_t answers_stride = 0;
};

int enqueue_subbatch(const BatchCtx &c, Slot &s, int64_t q_first, int nq, const ExtOut *ext) {
  bbq_index *ix = c.ix;
  uint64_t *d_lists_ext = ext ? ext->lists : nullptr;
  const int64_t list_cap_ext = ext ? ext->list_cap : 0;
  int32_t *d_counts_ext = ext ? ext->counts : nullptr;
  const Plan &p = ix->plan;
  const int64_t qb = query_data_bytes(ix, c.planes);
  uint8_t *hp = s.h_qbuf;
  QueryParams *hq = reinterpret_cast<QueryParams *>(s.h_qbuf + (size_t)nq * qb);
  for (int i = 0; i < nq; ++i)
    fill_query(ix, hp + (size_t)i * qb, hq + i, c.qquant + (size_t)(q_first + i) * ix->dim, c.qcorr + (size_t)(q_first + i) * 4,
               c.planes, c.one_bit, c.sim);
  size_t bytes = (size_t)nq * qb + (size_t)nq * sizeof(QueryParams);
  bool use_mfma = ix->opt_share == 32 && c.maxq <= 127 && ix->store_bits == 1;
  for (int i = 0; i < nq && use_mfma; ++i) use_mfma = mfma_query_ok(hq[i]);
  const bool mfma_fp = c.maxq <= 15;  // queryBits <= 4: rows as FP4, queries as FP6 - 64 dimensions per MFMA in the time the int8 form takes for 32
  const int mfma_scale8 = c.maxq <= 3 ? 8 : c.maxq <= 7 ? 4 : 2;   // products of scale8 / 8 * q: the accumulator's quarter-unit grain is 1, 1/2 or 1/4 of a qcDist unit
  size_t off_qbytes = 0, off_qmax = 0;
  if (use_mfma) {  // second copy of the queries as MFMA operands in fragment order + per-group maxima for the rows' magnitude budget
    const int groups = (nq + 31) / 32;
    off_qbytes = (bytes + 15) / 16 * 16;
    const size_t qbytes_len = (size_t)groups * (size_t)mfma_query_bytes_per_group(ix->w16, mfma_fp);
    off_qmax = off_qbytes + qbytes_len;
    bytes = off_qmax + (size_t)groups * 16;
    memset(s.h_qbuf + off_qbytes, 0, qbytes_len);
    float *qm = reinterpret_cast<float *>(s.h_qbuf + off_qmax);
    for (int gidx = 0; gidx < groups; ++gidx) qm[4 * gidx] = qm[4 * gidx + 1] = qm[4 * gidx + 2] = qm[4 * gidx + 3] = 0.f;
    for (int i = 0; i < nq; ++i) {
      if (mfma_fp) fill_query_mfma_fp(ix, s.h_qbuf + off_qbytes, i, c.qquant + (size_t)(q_first + i) * ix->dim, mfma_scale8);
      else fill_query_mfma(ix, s.h_qbuf + off_qbytes, i, c.qquant + (size_t)(q_first + i) * ix->dim);
      float *m = qm + 4 * (i / 32);
      // upper bounds (rounded up) of the group's |ay / ly|, |y1|, 1 / (cs * ly) and of the sum of a query's values (the largest
      // qcDist there can be, whatever y1 the caller passed)
      double qsum = 0;
      const uint8_t *qv = c.qquant + (size_t)(q_first + i) * ix->dim;
      for (int d = 0; d < ix->dim; ++d) qsum += qv[d];
      m[0] = std::max(m[0], (float)(fabs(hq[i].ay / hq[i].ly) * 1.000001));
      m[1] = std::max(m[1], (float)(fabs(hq[i].y1) * 1.000001));
      m[2] = std::max(m[2], (float)(1.0 / ((c.sim == 0 ? 2.0 : 1.0) * hq[i].ly) * 1.000001));
      m[3] = std::max(m[3], (float)(qsum * 1.000001));
    }
  }
  hipStream_t st = s.stream;
  HIPCHK(hipMemcpyAsync(s.d_block, s.h_block, (size_t)s.ctrl_bytes + bytes, hipMemcpyHostToDevice, st));  // control words := 0, queries
  s.ctrl_clean = false;
  uint64_t *d_lists = d_lists_ext ? d_lists_ext : s.d_lists;
  const int64_t list_cap = d_lists_ext ? list_cap_ext : s.list_cap;
  int32_t *d_list_counts = d_counts_ext ? d_counts_ext : s.d_list_counts;
  if (d_counts_ext) HIPCHK(hipMemsetAsync(d_counts_ext, 0, (size_t)nq * 8, st));
  // a shard without rows (and without a pilot replica) launches nothing that would write its answer blocks: all-zero blocks say
  // "nothing listed, no cut, no flags", which is what bbq_merge_answers expects of such a shard
  if (ext && ext->answers && p.segs.empty())
    HIPCHK(hipMemsetAsync(ext->answers, 0, ((size_t)(nq - 1) * (size_t)ext->answers_stride + (size_t)c.k + 2) * 8, st));

  const bool use_final = !d_lists_ext && p.final_k > 0 && !p.segs.empty();
  // few queries: the sparse launches append their candidates to the list themselves (ScanArgs::append_lists)
  const bool append = use_final && p.latency && ix->opt_latency_append && !ix->has_pilot && ix->opt_share == 1;
  s.appended = append || (use_mfma && !d_lists_ext);
  s.timed = false;
  ix->sweep_resident_acc = 0;
  for (const Segment &g : p.segs) {
    const Storage &sto = g.storage == 0 ? ix->pilot : ix->main;
    ScanArgs a{};
    a.idx = launch_view(ix, sto, g.chunk_begin, g.n_chunks);
    ix->stats.resident_bytes = ix->sweep_resident_acc;  // of one query's sweep over all segments (complete after the last one)
    a.qplanes = reinterpret_cast<const uint4 *>(s.d_qbuf);
    a.qparams = reinterpret_cast<const QueryParams *>(s.d_qbuf + (size_t)nq * qb);
    a.chunk_begin = g.chunk_begin;
    a.row_id_base = sto.row_id_base;
    a.theta = s.d_theta;
    a.counts = s.d_counts;
    a.entries = s.d_entries;
    a.flags = s.d_flags;
    a.cap = g.cap;
    a.n_chunks = (int32_t)g.n_chunks;
    // the slot's area may be larger than this plan asks for (slots are shared and grow-only): the plan's size governs
    a.ovf = p.flood_cap > 0 ? s.d_ovf : nullptr;
    a.ovf_counts = s.d_ovf_counts;
    a.ovf_cap = (int32_t)p.flood_cap;
    a.dense_score32 = g.dense ? s.d_dense0 : nullptr;
    a.dense_stride = s.dense_cap;
    // the matrix-core shared sweep appends as well whenever the lists are this slot's own: its waves then never wait for each other
    const bool mfma_append = use_mfma && !g.dense && !d_lists_ext;
    const bool append_here = (append && !g.dense && (ix->opt_append_last || &g != &p.segs.back())) || mfma_append;
    if (append_here) {
      a.append_lists = d_lists;
      a.append_base = d_list_counts;
      a.append_counts = s.d_append_counts;
      a.append_cap = list_cap;
    }
    const int my_slot = (int)(&s - ix->slots);
    // one big sweep at a time on the device - for the memory-bound sweeps, which share the Infinity Cache's budget.  The chains of the
    // matrix-core sweep are issue-bound and run side by side: serialised, the finalize launch and the launch gaps between a chain's two
    // large sweeps were idle time (128 -> 138 K q/s at 10 M x 768)
    const bool mfma_here = use_mfma && !g.dense && mfma_sweep_supported(a);
    if (g.big && !mfma_here && ix->ctx->last_big_slot >= 0 && ix->ctx->last_big_slot != my_slot)
      HIPCHK(hipStreamWaitEvent(st, ix->slots[ix->ctx->last_big_slot].ev_big, 0));
    if (g.dominant) HIPCHK(hipEventRecord(s.ev0, st));
    if (mfma_here)
      HIPCHK(launch_scan_mfma(a, s.d_qbuf + off_qbytes, reinterpret_cast<const float *>(s.d_qbuf + off_qmax), mfma_fp ? mfma_scale8 / 8.0f : 0.0f, nq, (int)g.n_chunks, st));
    else if (!g.dense && ix->opt_share > 1 && shared_sweep_supported(a, ix->opt_share))
      HIPCHK(launch_scan_shared(a, c.planes, ix->opt_share, nq, (int)g.n_chunks, st));
    else
      HIPCHK(launch_scan(a, c.planes, g.dense, nq, (int)g.n_chunks, st));
    if (g.big && !mfma_here) {
      HIPCHK(hipEventRecord(s.ev_big, st));
      ix->ctx->last_big_slot = my_slot;
    }
    if (g.dominant) {
      HIPCHK(hipEventRecord(s.ev1, st));
      s.timed = true;
      s.timed_rows = g.rows * nq;
      // a shared sweep reads each row once for `share` queries
      const int share = mfma_here ? mfma_queries_per_tile_load(a, nq, mfma_fp) : (!g.dense && ix->opt_share > 1 && shared_sweep_supported(a, ix->opt_share)) ? ix->opt_share : 1;
      // the matrix-core sweep reads the codes and the EXACT corrections (compact layout: 24 of the side array's 32 B per row instead of
      // the tile's 4-byte word), once per 32 queries
      const int64_t row_bytes = !mfma_here ? (int64_t)ix->bytes_per_row
                                           : sto.view.layout == kLayoutCompact ? (int64_t)sto.view.w16 * 16 + 24 : (int64_t)sto.view.tile_stride / kTileRows;
      s.timed_bytes = g.rows * ((nq + share - 1) / share) * row_bytes;
    }
    FinalizeArgs f{};
    f.counts = s.d_counts;
    f.entries = s.d_entries;
    f.dense_score32 = s.d_dense0;
    f.dense_stride = s.dense_cap;
    f.dense_rows = g.dense ? (int32_t)g.rows : 0;
    f.dense_row_id_base = sto.row_id_base + g.chunk_begin * kChunkRows;
    f.n_chunks = (int32_t)g.n_chunks;
    f.cap = g.cap;
    f.ovf = a.ovf;
    f.ovf_cap = a.ovf_cap;
    f.append_counts = append_here ? s.d_append_counts : nullptr;
    f.lists = d_lists;
    f.list_counts = d_list_counts;
    f.list_cap = list_cap;
    f.emit = g.emit ? 1 : 0;
    f.topk_keys = s.d_topk;
    f.topk_counts = s.d_topk_counts;
    f.theta = s.d_theta;
    f.flags = s.d_flags;
    f.k = (int32_t)c.k;
    f.need_theta = g.need_theta ? 1 : 0;
    if (use_final && &g == &p.segs.back()) {
      f.final_out = s.d_final;
      f.final_stride = (int32_t)s.final_stride;
      f.final_k = (int32_t)p.final_k;
    } else if (ext && ext->answers && p.final_k > 0 && &g == &p.segs.back()) {
      f.final_out = ext->answers;
      f.final_stride = (int32_t)ext->answers_stride;
      f.final_k = (int32_t)p.final_k;
      f.final_shard = 1;
    }
    HIPCHK(launch_finalize(f, nq, st));
  }
  s.final_used = use_final;
  if (!d_lists_ext) {
    if (use_final) {
      // the answer itself (header + k entries per query, one copy) instead of the candidate list: the list is fetched only for a
      // query whose answer the device could not prove (ties), see begin_replay
      if (nq == 1 || p.final_k + 2 == s.final_stride)
        HIPCHK(hipMemcpyAsync(s.h_final, s.d_final, ((size_t)(nq - 1) * s.final_stride + (size_t)p.final_k + 2) * 8, hipMemcpyDeviceToHost, st));
      else
        HIPCHK(hipMemcpy2DAsync(s.h_final, (size_t)s.final_stride * 8, s.d_final, (size_t)s.final_stride * 8, (size_t)(p.final_k + 2) * 8, (size_t)nq,
                                hipMemcpyDeviceToHost, st));
    } else {
      HIPCHK(hipMemcpyAsync(s.h_list_counts, s.d_list_counts, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpy2DAsync(s.h_lists, (size_t)s.hprefix * 8, s.d_lists, (size_t)s.list_cap * 8, (size_t)s.hprefix * 8, (size_t)nq,
                              hipMemcpyDeviceToHost, st));
    }
  }
  HIPCHK(hipEventRecord(s.ev_done, st));
  s.busy = true;
  s.nq = nq;
  s.q_first = q_first;
  return BBQ_OK;
}

void account_timing(bbq_index *ix, Slot &s) {
  if (!s.timed) return;
  float ms = 0;
  if (hipEventElapsedTime(&ms, s.ev0, s.ev1) == hipSuccess) {
    ix->stats.last_scan_ms = ms;
    ix->stats.last_scan_rows = s.timed_rows;
    ix->stats.last_scan_bytes = s.timed_bytes;
    ix->stats.total_scan_ms += ms;
    ix->stats.total_scan_bytes += s.timed_bytes;
    ix->stats.total_scan_launches += 1;
  }
}

// every f32 score of one query to the host (out [n_rows] of this index)
int dense_scores_one(const BatchCtx &c, int64_t qi, float *out) {
  bbq_index *ix = c.ix;
  const int64_t n = ix->main.view.n_rows;
  const int64_t chunks = ix->main.n_chunks();
  if (ix->dense_all_cap < n) {
    if (ix->d_dense_all) HIPCHK(hipFree(ix->d_dense_all));
    ix->d_dense_all = nullptr;
    HIPCHK(hipMalloc((void **)&ix->d_dense_all, (size_t)std::max<int64_t>(n, 1) * 4));
    ix->dense_all_cap = n;
  }
  int rc_aux = ensure_aux_qbuf(ix->ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc_aux != BBQ_OK) return rc_aux;
  const int64_t qb = query_data_bytes(ix, c.planes);
  std::vector<uint8_t> hb((size_t)qb + sizeof(QueryParams));
  fill_query(ix, hb.data(), reinterpret_cast<QueryParams *>(hb.data() + qb), c.qquant + (size_t)qi * ix->dim, c.qcorr + (size_t)qi * 4,
             c.planes, c.one_bit, c.sim);
  hipStream_t st = ix->aux_stream;
  HIPCHK(hipMemcpyAsync(ix->ctx->d_aux_qbuf, hb.data(), hb.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  ScanArgs a{};
  a.idx = launch_view(ix, ix->main);
  a.qplanes = reinterpret_cast<const uint4 *>(ix->ctx->d_aux_qbuf);
  a.qparams = reinterpret_cast<const QueryParams *>(ix->ctx->d_aux_qbuf + qb);
  a.chunk_begin = 0;
  a.row_id_base = ix->main.row_id_base;
  a.flags = ix->d_aux_flags;
  a.dense_score32 = ix->d_dense_all;
  a.dense_stride = n;
  // gridDim.x is limited to 2^31-1: fine for any index that fits in HBM
  HIPCHK(launch_scan(a, c.planes, true, 1, (int)chunks, st));
  HIPCHK(hipMemcpyAsync(out, ix->d_dense_all, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return BBQ_OK;
}

// dense path for one query: every f32 score to the host, full replay of the reference loop
int dense_search_one(const BatchCtx &c, int64_t qi, int32_t *out_idx, float *out_score, int64_t *out_n) {
  bbq_index *ix = c.ix;
  const int64_t n = ix->main.view.n_rows;
  std::vector<float> h((size_t)std::max<int64_t>(n, 1));
  int rc = dense_scores_one(c, qi, h.data());
  if (rc != BBQ_OK) return rc;
  HeapReplay hr(c.k, n);
  for (int64_t i = 0; i < n; ++i) hr.offer(h[(size_t)i], (int32_t)(ix->main.row_id_base + i));
  *out_n = hr.finish(out_idx, out_score);
  ix->stats.dense_fallbacks += 1;
  ix->stats.candidates += n;
  return BBQ_OK;
}

// device work of the slot's sub-batch is done: collect it and start the heap replays (on the pool when replay_threads > 1)
int begin_replay(const BatchCtx &c, Slot &s, int32_t *out_idx, float *out_score, int64_t *out_n) {
  bbq_index *ix = c.ix;
  HIPCHK(hipEventSynchronize(s.ev_done));
  s.busy = false;
  account_timing(ix, s);
  const int nq = s.nq;
  s.dense_q.clear();
  s.tails.assign((size_t)nq, std::vector<uint64_t>());
  const bool fin = s.final_used;
  if (fin)  // the header slots of the answer block carry what the two small copies used to bring
    for (int i = 0; i < nq; ++i) {
      const uint64_t *hdr = s.h_final + (size_t)i * s.final_stride;
      s.h_list_counts[2 * i] = (int32_t)(uint32_t)hdr[0];
      s.h_list_counts[2 * i + 1] = (int32_t)(uint32_t)(hdr[0] >> 32);
    }
  auto final_cnt = [&s](int i) { return (int32_t)(uint32_t)s.h_final[(size_t)i * s.final_stride + 1]; };
  auto final_replay = [&s](int i) { return (uint32_t)(s.h_final[(size_t)i * s.final_stride + 1] >> 32) != 0; };
  // list entries of query i that sit in the pinned h_lists row: the prefix the enqueue-time copy brought over, or - when the device was
  // to answer and could not (equal scores) - the whole list, fetched below with ONE batch of asynchronous copies on the slot's stream
  // (a synchronous pageable copy per query on the null stream stalled every other slot: duplicated vectors make such queries common)
  s.host_cnt.assign((size_t)nq, 0);
  int n_replay = 0;
  struct CountReplays {  // on every exit path
    bbq_index *ix; int &n;
    ~CountReplays() { ix->stats.host_replays += n; }
  } count_replays{ix, n_replay};
  bool fetched = false;
  for (int i = 0; i < nq; ++i) {
    const int32_t cnt = s.h_list_counts[2 * i], flags = s.h_list_counts[2 * i + 1];
    if (flags != 0) { s.dense_q.push_back(i); continue; }
    if (fin && !final_replay(i)) continue;  // answered on the device
    ++n_replay;
    int64_t have = fin ? 0 : std::min<int64_t>(cnt, s.hprefix);
    if (fin && cnt > 0) {
      have = std::min<int64_t>(cnt, s.hprefix);
      HIPCHK(hipMemcpyAsync(s.h_lists + (size_t)i * s.hprefix, s.d_lists + (size_t)i * s.list_cap, (size_t)have * 8, hipMemcpyDeviceToHost, s.stream));
      fetched = true;
    }
    s.host_cnt[(size_t)i] = have;
    if (cnt > have) {  // rare (a flood): the rest of a list longer than the pinned row
      s.tails[(size_t)i].resize((size_t)(cnt - have));
      HIPCHK(hipMemcpyAsync(s.tails[(size_t)i].data(), s.d_lists + (size_t)i * s.list_cap + have, (size_t)(cnt - have) * 8, hipMemcpyDeviceToHost, s.stream));
      fetched = true;
    }
  }
  if (fetched) HIPCHK(hipStreamSynchronize(s.stream));
  if (s.appended)  // append mode leaves the entries of a segment in arrival order: the reference loop wants them by row (row << 32 | score bits)
    for (int i = 0; i < nq; ++i) {
      if (s.host_cnt[(size_t)i] == 0 && s.tails[(size_t)i].empty()) continue;
      uint64_t *l = s.h_lists + (size_t)i * s.hprefix;
      std::vector<uint64_t> &t = s.tails[(size_t)i];
      if (t.empty()) {
        std::sort(l, l + s.host_cnt[(size_t)i]);
      } else {  // list longer than the pinned row: sort the whole of it in the tail vector
        t.insert(t.begin(), l, l + s.host_cnt[(size_t)i]);
        s.host_cnt[(size_t)i] = 0;
        std::sort(t.begin(), t.end());
      }
    }
  const int64_t k = c.k, n_total = ix->main.row_id_base + ix->main.view.n_rows;
  Slot *sp = &s;
  if (fin) {  // queries the last finalize launch answered: the sorted rows are the result
    for (int i = 0; i < nq; ++i) {
      if (s.h_list_counts[2 * i + 1] != 0 || final_replay(i)) continue;
      const int64_t qi = s.q_first + i;
      const int32_t m = final_cnt(i);
      const uint64_t *fo = s.h_final + (size_t)i * s.final_stride + 2;
      for (int32_t j = 0; j < m; ++j) {
        const uint32_t bits = (uint32_t)fo[j];
        out_idx[qi * k + j] = (int32_t)(uint32_t)(fo[j] >> 32);
        memcpy(&out_score[qi * k + j], &bits, 4);
      }
      out_n[qi] = m;
    }
  }
  auto replay_range = [sp, k, n_total, out_idx, out_score, out_n, fin](int lo, int hi) {
    Slot &s = *sp;
    for (int i = lo; i < hi; ++i) {
      const int32_t cnt = s.h_list_counts[2 * i], flags = s.h_list_counts[2 * i + 1];
      if (flags != 0) continue;
      if (fin && (uint32_t)(s.h_final[(size_t)i * s.final_stride + 1] >> 32) == 0) continue;
      HeapReplay hr(k, n_total);
      const uint64_t *l = s.h_lists + (size_t)i * s.hprefix;
      const int64_t head = s.host_cnt[(size_t)i];
      (void)cnt;
      for (int64_t j = 0; j < head; ++j) {
        const uint32_t bits = (uint32_t)l[j];
        float sc;
        memcpy(&sc, &bits, 4);
        hr.offer(sc, (int32_t)(uint32_t)(l[j] >> 32));
      }
      for (uint64_t e : s.tails[(size_t)i]) {
        const uint32_t bits = (uint32_t)e;
        float sc;
        memcpy(&sc, &bits, 4);
        hr.offer(sc, (int32_t)(uint32_t)(e >> 32));
      }
      const int64_t qi = s.q_first + i;
      out_n[qi] = hr.finish(out_idx + qi * k, out_score + qi * k);
    }
  };
  const int T = std::min(ix->opt_replay_threads, n_replay);
  s.replaying = true;
  if (n_replay == 0) {
    // nothing to replay
  } else if (T <= 1) {
    replay_range(0, nq);
  } else {
    ReplayPool &pool = ReplayPool::get();
    pool.ensure(ix->opt_replay_threads);
    const int jobs = std::min(nq, T * 2);  // a few more jobs than threads: uneven lists balance out
    s.pending.store(jobs);
    for (int t = 0; t < jobs; ++t) {
      const int lo = (int)((int64_t)nq * t / jobs), hi = (int)((int64_t)nq * (t + 1) / jobs);
      pool.submit([sp, replay_range, lo, hi] {
        replay_range(lo, hi);
        sp->pending.fetch_sub(1, std::memory_order_release);
      });
    }
  }
  return BBQ_OK;
}

// waits for the slot's replays, then serves the queries the device could not bound (dense path)
int finish_replay(const BatchCtx &c, Slot &s, int32_t *out_idx, float *out_score, int64_t *out_n) {
  bbq_index *ix = c.ix;
  while (s.pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  s.replaying = false;
  const int64_t k = c.k;
  for (int i = 0; i < s.nq; ++i)
    if (s.h_list_counts[2 * i + 1] == 0) ix->stats.candidates += s.h_list_counts[2 * i];
  if (s.dense_q.empty()) return BBQ_OK;
  // queries the device could not bound.  The shared sweeps have no flood tier: a query whose candidate slots overflowed
  // there first gets one sweep of its own (slot s is free again at this point) before it pays for the dense path.
  const std::vector<int> flagged = s.dense_q;
  std::vector<uint32_t> why;
  for (int i : flagged) why.push_back((uint32_t)s.h_list_counts[2 * i + 1]);
  const int64_t first = s.q_first;
  s.dense_q.clear();
  for (size_t j = 0; j < flagged.size(); ++j) {
    const int64_t qi = first + flagged[j];
    if (ix->opt_share > 1 && ix->plan.flood_cap > 0 && why[j] == kFlagOverflow) {
      BatchCtx cs = c;
      cs.k = ix->plan.k;  // the rank the device runs this call with
      const int share = ix->opt_share;
      ix->opt_share = 1;
      int rc = enqueue_subbatch(cs, s, qi, 1, nullptr);
      ix->opt_share = share;
      if (rc == BBQ_OK) rc = begin_replay(c, s, out_idx, out_score, out_n);
      if (rc != BBQ_OK) return rc;
      while (s.pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
      s.replaying = false;
      const bool solved = s.dense_q.empty();
      s.dense_q.clear();
      if (solved) {
        ix->stats.candidates += s.h_list_counts[0];
        continue;
      }
    }
    int rc = dense_search_one(c, qi, out_idx + qi * k, out_score + qi * k, out_n + qi);
    if (rc != BBQ_OK) return rc;
  }
  return BBQ_OK;
}

// brings a slot back to "free": collect + replay + wait, whatever is still outstanding
int reclaim_slot(const BatchCtx &c, Slot &s, int32_t *out_idx, float *out_score, int64_t *out_n) {
  int rc = BBQ_OK;
  if (s.busy) rc = begin_replay(c, s, out_idx, out_score, out_n);
  if (rc == BBQ_OK && s.replaying) rc = finish_replay(c, s, out_idx, out_score, out_n);
  return rc;
}

int drain(bbq_index *ix) {
  for (int i = 0; i < kMaxSlots; ++i)
    if (ix->slots[i].stream) HIPCHK(hipStreamSynchronize(ix->slots[i].stream));
  return BBQ_OK;
}

// ------------------------------------------------------------------------------------------------ single-query latency path

// waits for the sequence word the last finalize launch of a latency chain raises in mapped host memory (polling: no event, no copy)
int wait_latency_answer(DeviceCtx *ctx, Slot &s, uint64_t seq) {
  volatile uint64_t *flag = ctx->h_lat;
  for (int64_t spin = 0; spin < (1ll << 31); ++spin) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return BBQ_OK;
    if ((spin & 0xffff) == 0xffff && hipEventQuery(s.ev_done) != hipErrorNotReady) break;  // the launch chain is over (or failed)
    __builtin_ia32_pause();
  }
  const hipError_t e = hipEventSynchronize(s.ev_done);
  if (e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return BBQ_OK;
  s.ctrl_clean = false;  // whatever the chain left behind
  if (e != hipSuccess) return fail(BBQ_ERR_HIP, "latency path: %s", hipGetErrorString(e));
  return fail(BBQ_ERR_HIP, "latency path: the device finished without an answer");
}

// The single-query call on a large index: threshold from a pre-sampled prefix (bbq_lat_pre_kernel + bbq_lat_select_kernel: two small
// launches), ONE sweep over all rows with it, final selection on the list alone - four launches where the segmented chain has six,
// and nothing in front of the large sweep but the two small ones.  The list is every row above the threshold, not a heap history: a
// query whose answer the device cannot prove (equal scores, NaN, more candidates than the selection holds) is handed to the
// segmented chain (*done = false), which replays it exactly.
int search_latency_presampled(const BatchCtx &c, const BatchCtx &cs, int32_t *out_idx, float *out_score, int64_t *out_n, bool *done) {
  bbq_index *ix = c.ix;
  const Plan &p = ix->plan;
  *done = false;
  const int64_t N = ix->main.view.n_rows, k2 = p.final_k;
  if (!ix->opt_latency_presample || !ix->opt_latency_fused || !ix->opt_latency_append || ix->has_pilot || ix->opt_share != 1 || !p.latency ||
      k2 < 1 || k2 > kFinalSelectMax || N < 262144 || !latency_path_supported(ix->main.view, cs.planes))
    return BBQ_OK;
  // sample enough rows for ~6000 candidates in the sweep (the selection holds kFinalizeKeyCap of them)
  int64_t P = ((k2 + 2) * N / 6000 + kChunkRows - 1) / kChunkRows * kChunkRows;
  P = std::max<int64_t>(P, 8192);
  if (P > N / 4) return BBQ_OK;
  // Keys per wave of the sample: ONE (the wave's maximum) when the sample has at least eight times as many waves as the rank asks for -
  // two of the rank's best rows then rarely share a wave, the threshold is all but the prefix's true order statistic, and the selection
  // launch has a quarter of the keys to go through (10 M rows, k = 100: 2 656 instead of 10 624 keys, select 11.6 -> 6.6 us, pre-sample
  // 9.4 -> 8.0 us); four otherwise.  Either way the threshold is an order statistic of a SUBSET of the rows: a valid lower bound.
  int per_wave = (P / kTileRows >= 8 * (k2 + 2)) ? 1 : 4;
  if (P / kTileRows * per_wave > kLatPreKeys) per_wave = 1;
  const int64_t n_keys = P / kTileRows * per_wave;
  if (n_keys > kLatPreKeys || n_keys < k2 + 2) return BBQ_OK;
  Slot &s = ix->slots[0];
  int rc = ensure_slot(ix, s, 1, true);
  if (rc != BBQ_OK) return rc;
  DeviceCtx *ctx = ix->ctx;
  hipStream_t st = s.stream;
  if (!s.ctrl_clean) {
    HIPCHK(hipMemsetAsync(s.d_block, 0, (size_t)s.ctrl_bytes, st));
    s.ctrl_clean = true;
  }
  LatScanArgs a{};
  ix->sweep_resident_acc = 0;
  a.idx = launch_view(ix, ix->main);
  ix->stats.resident_bytes = ix->sweep_resident_acc;  // the one sweep over all rows
  a.row_id_base = ix->main.row_id_base;
  a.theta = s.d_theta;
  a.flags = s.d_flags;
  a.list_counts = s.d_list_counts;
  a.append_count = s.d_append_counts;
  a.list = s.d_lists;
  a.list_cap = s.list_cap;
  fill_query(ix, reinterpret_cast<uint8_t *>(a.planes), &a.p, c.qquant, c.qcorr, cs.planes, cs.one_bit, cs.sim);
  LatPreArgs pre{};
  pre.idx = launch_view(ix, ix->main);
  pre.rows = (int32_t)P;
  pre.per_wave = per_wave;
  pre.pre_keys = ctx->d_pre_keys;
  pre.flags = s.d_flags;
  pre.p = a.p;
  memcpy(pre.planes, a.planes, sizeof pre.planes);
  HIPCHK(launch_lat_pre(pre, cs.planes, st));
  // rank k2 + 2: the sweep must list at least k2 + 1 rows for the selection to see the boundary of the answer
  HIPCHK(launch_lat_select(ctx->d_pre_keys, (int)n_keys, (int)(k2 + 2), s.d_theta, st));
  a.chunk_begin = 0;
  a.n_chunks = (int32_t)ix->main.n_chunks();
  a.first = 0;
  HIPCHK(launch_lat_scan(a, cs.planes, st));
  const uint64_t seq = ++ctx->lat_seq;
  FinalizeArgs f{};
  f.append_counts = s.d_append_counts;
  f.lists = s.d_lists;
  f.list_counts = s.d_list_counts;
  f.list_cap = s.list_cap;
  f.emit = 1;
  f.topk_keys = s.d_topk;
  f.topk_counts = s.d_topk_counts;
  f.theta = s.d_theta;
  f.flags = s.d_flags;
  f.k = (int32_t)cs.k;
  f.final_out = ctx->d_lat + kLatAnswerOffset;
  f.final_stride = kFinalSelectMax + 2;
  f.final_k = (int32_t)k2;
  f.done_flag = ctx->d_lat;
  f.seq = seq;
  HIPCHK(launch_finalize(f, 1, st));
  HIPCHK(hipEventRecord(s.ev_done, st));
  rc = wait_latency_answer(ctx, s, seq);
  if (rc != BBQ_OK) return rc;
  const uint64_t *hdr = ctx->h_lat + kLatAnswerOffset;
  const uint32_t listed = (uint32_t)hdr[0], flags = (uint32_t)(hdr[0] >> 32), m = (uint32_t)hdr[1], replay = (uint32_t)(hdr[1] >> 32);
  s.timed = false;
  // The list holds the rows ABOVE the sampled threshold only, so the selection's "take every listed row" case (total <= k2) proves
  // nothing here: equal keys at ranks k2+1 / k2+2 of the sample can leave fewer than k2 rows above it (N >= 262144 > k2, so a
  // complete answer has exactly k2 entries).  Anything else goes to the segmented chain.
  if (flags != 0 || replay != 0 || m != (uint32_t)k2) return BBQ_OK;
  for (uint32_t j = 0; j < m; ++j) {
    const uint64_t e = hdr[2 + j];
    const uint32_t bits = (uint32_t)e;
    out_idx[j] = (int32_t)(uint32_t)(e >> 32);
    memcpy(&out_score[j], &bits, 4);
  }
  out_n[0] = m;
  ix->stats.candidates += listed;
  *done = true;
  return BBQ_OK;
}

// one query, no copies: every sweep takes the query from its kernel arguments (bbq_latency_kernels.hip), the last finalize launch
// writes the answer to mapped host memory and raises the sequence word this thread polls.  Returns BBQ_OK with *done = false when the
// call has to take the general path (index shape without an instantiation).
int search_latency_chain(const BatchCtx &c, const BatchCtx &cs, int32_t *out_idx, float *out_score, int64_t *out_n, bool *done) {
  bbq_index *ix = c.ix;
  const Plan &p = ix->plan;
  *done = false;
  if (!ix->opt_latency_fused || !ix->opt_latency_append || !ix->opt_append_last || ix->has_pilot || ix->opt_share != 1 || !p.latency ||
      p.final_k < 1 || p.final_k > kFinalSelectMax || p.segs.empty() || !p.segs[0].dense || !latency_path_supported(ix->main.view, cs.planes))
    return BBQ_OK;
  for (size_t i = 1; i < p.segs.size(); ++i)
    if (p.segs[i].dense || p.segs[i].storage != 1) return BBQ_OK;
  Slot &s = ix->slots[0];
  int rc = ensure_slot(ix, s, 1, true);
  if (rc != BBQ_OK) return rc;
  DeviceCtx *ctx = ix->ctx;
  hipStream_t st = s.stream;
  if (!s.ctrl_clean) {  // the slot's last user was not this chain
    HIPCHK(hipMemsetAsync(s.d_block, 0, (size_t)s.ctrl_bytes, st));
    s.ctrl_clean = true;
  }
  LatScanArgs a{};
  a.idx = launch_view(ix, ix->main);
  a.row_id_base = ix->main.row_id_base;
  a.theta = s.d_theta;
  a.flags = s.d_flags;
  a.list_counts = s.d_list_counts;
  a.append_count = s.d_append_counts;
  a.list = s.d_lists;
  a.list_cap = s.list_cap;
  fill_query(ix, reinterpret_cast<uint8_t *>(a.planes), &a.p, c.qquant, c.qcorr, cs.planes, cs.one_bit, cs.sim);
  const uint64_t seq = ++ctx->lat_seq;
  for (size_t i = 0; i < p.segs.size(); ++i) {
    const Segment &g = p.segs[i];
    a.chunk_begin = g.chunk_begin;
    a.n_chunks = (int32_t)g.n_chunks;
    a.first = i == 0 ? 1 : 0;
    HIPCHK(launch_lat_scan(a, cs.planes, st));
    FinalizeArgs f{};
    f.append_counts = s.d_append_counts;
    f.lists = s.d_lists;
    f.list_counts = s.d_list_counts;
    f.list_cap = s.list_cap;
    f.emit = 1;
    f.topk_keys = s.d_topk;
    f.topk_counts = s.d_topk_counts;
    f.theta = s.d_theta;
    f.flags = s.d_flags;
    f.k = (int32_t)cs.k;
    f.need_theta = i + 1 < p.segs.size() ? 1 : 0;
    if (i + 1 == p.segs.size()) {
      f.final_out = ctx->d_lat + kLatAnswerOffset;
      f.final_stride = kFinalSelectMax + 2;
      f.final_k = (int32_t)p.final_k;
      f.done_flag = ctx->d_lat;
      f.seq = seq;
    }
    HIPCHK(launch_finalize(f, 1, st));
  }
  HIPCHK(hipEventRecord(s.ev_done, st));
  rc = wait_latency_answer(ctx, s, seq);
  if (rc != BBQ_OK) return rc;
  const uint64_t *hdr = ctx->h_lat + kLatAnswerOffset;
  const uint32_t listed = (uint32_t)hdr[0], flags = (uint32_t)(hdr[0] >> 32), m = (uint32_t)hdr[1], replay = (uint32_t)(hdr[1] >> 32);
  s.timed = false;
  if (flags == 0 && replay == 0) {  // answered on the device
    const int64_t k = c.k;
    for (uint32_t j = 0; j < m; ++j) {
      const uint64_t e = hdr[2 + j];
      const uint32_t bits = (uint32_t)e;
      out_idx[j] = (int32_t)(uint32_t)(e >> 32);
      memcpy(&out_score[j], &bits, 4);
    }
    (void)k;
    out_n[0] = m;
    ix->stats.candidates += listed;
    *done = true;
    return BBQ_OK;
  }
  // equal scores in or at the edge of the answer (or a flagged query): hand over to the general path's collection - the list on the
  // device is complete and it is this slot's
  s.h_final[0] = (uint64_t)listed | ((uint64_t)flags << 32);
  s.h_final[1] = (uint64_t)1 << 32;
  s.busy = true;
  s.nq = 1;
  s.q_first = 0;
  s.final_used = true;
  s.appended = true;
  rc = begin_replay(c, s, out_idx, out_score, out_n);
  if (rc == BBQ_OK) rc = finish_replay(c, s, out_idx, out_score, out_n);
  if (rc != BBQ_OK) return rc;
  *done = true;
  return BBQ_OK;
}

}  // namespace

namespace bbq {

int settle_shard_slots(DeviceCtx *ctx, bbq_index *owner) {
  for (int i = 0; i < kMaxSlots; ++i) {
    Slot &s = ctx->slots[i];
    if (!s.busy || !s.shard_owner || (owner && s.shard_owner != owner)) continue;
    HIPCHK(hipEventSynchronize(s.ev_done));
    s.busy = false;
    account_timing(s.shard_owner, s);
    s.shard_owner = nullptr;
  }
  return BBQ_OK;
}

// every f32 score of one query on this index (shard), to host memory: the dense path of a multi-device index
int dense_scores_host(bbq_index *ix, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim, float *out) {
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  if (ix->n_rows == 0) return BBQ_OK;
  BatchCtx c{ix, qquant, qcorr, planes_of_call(ix, qquant, ix->dim, query_bits == 1), query_bits == 1 ? 1 : 0, sim, 0};
  ix->stats.dense_fallbacks += 1;
  return dense_scores_one(c, 0, out);
}

// frees what the index owns; the device context (streams, workspace) stays
void destroy_unlocked(bbq_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  if (ix->pilot.d_tiles) (void)hipFree(ix->pilot.d_tiles);
  if (ix->main.d_tiles) (void)hipFree(ix->main.d_tiles);
  if (ix->pilot.d_exact) (void)hipFree(ix->pilot.d_exact);
  if (ix->main.d_exact) (void)hipFree(ix->main.d_exact);
  if (ix->d_dense_all) (void)hipFree(ix->d_dense_all);
  if (ix->ctx)
    for (size_t i = 0; i < ix->ctx->cache_users.size(); ++i)
      if (ix->ctx->cache_users[i].index == ix) { ix->ctx->cache_users.erase(ix->ctx->cache_users.begin() + (long)i); break; }
  if (ix->ctx) (void)settle_shard_slots(ix->ctx, ix);  // sub-batches of an asynchronous scan that was never waited for
  for (bbq_index::ShardSet &set : ix->shard_set) {
    if (set.done) { (void)hipEventSynchronize(set.done); (void)hipEventDestroy(set.done); }
    if (set.h_total) (void)hipHostFree(set.h_total);
    if (set.d_lists) (void)hipFree(set.d_lists);
    if (set.d_counts) (void)hipFree(set.d_counts);
  }
  delete ix;
}

}  // namespace bbq

// ================================================================================================ C ABI

extern "C" {

int bbq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int bbq_index_create_shard(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits,
                           double centroid_dp, int64_t row_base, const uint8_t *pilot_codes, const double *pilot_corr,
                           int64_t n_pilot, int32_t device, bbq_index **out) {
  return bbq_index_create_shard_opts(codes, corr, n_rows, dim, index_bits, centroid_dp, row_base, pilot_codes, pilot_corr, n_pilot, device, nullptr, out);
}

int bbq_index_create_shard_opts(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits,
                                double centroid_dp, int64_t row_base, const uint8_t *pilot_codes, const double *pilot_corr,
                                int64_t n_pilot, int32_t device, const bbq_index_options *opts, bbq_index **out) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create: out is null");
  *out = nullptr;
  if (n_rows < 0 || dim <= 0 || row_base < 0 || n_pilot < 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create: bad size");
  if (n_rows > 0 && (!codes || !corr)) return fail(BBQ_ERR_INVALID_ARG, "目标向量集合不能为空");
  if (index_bits < 1 || index_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "indexBits必须在1-8之间");
  if (check_options(opts) != BBQ_OK) return BBQ_ERR_INVALID_ARG;
  if (!dim_supported(dim, dim == 1 ? 1 : store_bits_of(index_bits)))
    return fail(BBQ_ERR_UNSUPPORTED, "dimension %d at indexBits %d: the integer dot product would not fit 31 bits", dim, index_bits);
  if (n_pilot > 0 && (!pilot_codes || !pilot_corr)) return fail(BBQ_ERR_INVALID_ARG, "pilot arrays are null");
  if (n_pilot > 0 && row_base == 0) return fail(BBQ_ERR_INVALID_ARG, "the shard that owns row 0 takes no pilot replica");
  if (n_pilot > 0 && n_pilot > row_base) return fail(BBQ_ERR_INVALID_ARG, "pilot rows must precede the shard (n_pilot <= row_base)");
  if (n_pilot > 0 && n_pilot != row_base && n_pilot % kChunkRows != 0)
    return fail(BBQ_ERR_INVALID_ARG, "n_pilot must be a multiple of %d", kChunkRows);
  if (row_base + n_rows > 0xFFFFFFFFll) return fail(BBQ_ERR_UNSUPPORTED, "more than 2^32 rows");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  if (device < 0 || device >= ndev) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));

  std::unique_ptr<bbq_index> ix(new bbq_index());
  ix->device = device;
  ix->dim = dim;
  ix->index_bits = index_bits;
  // a multi-bit index of dimension 1 is the one shape the reference's BATCH scorer accepts (the unpacked byte is read as a
  // packed row, src/batchDotProduct.ts:425-433): it is stored and scored as the packed 1-bit row it is taken for
  ix->store_bits = dim == 1 ? 1 : store_bits_of(index_bits);
  ix->pb = row_bytes_of(dim, ix->store_bits);
  ix->w16 = (ix->pb + 15) / 16;
  ix->n_rows = n_rows;
  ix->row_base = row_base;
  ix->centroid_dp = centroid_dp;
  ix->has_pilot = n_pilot > 0;
  ix->want_compact = want_compact_of(opts);
  DeviceCtx *ctx = nullptr;
  int rc0 = get_ctx(device, &ctx);
  if (rc0 != BBQ_OK) return rc0;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ix->ctx = ctx;
  ix->slots = ctx->slots;
  ix->aux_stream = ctx->aux_stream;
  ix->d_aux_flags = ctx->d_aux_flags;
  rc0 = ensure_aux_qbuf(ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc0 != BBQ_OK) return rc0;
  int rc;
  if (ix->has_pilot) {
    rc = make_storage(ix.get(), ix->pilot, pilot_codes, pilot_corr, n_pilot, 0, true);
    if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
    const int had = ix->has_x1;
    rc = make_storage(ix.get(), ix->main, codes, corr, n_rows, row_base, true);
    if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
    if (ix->has_x1 != had) {  // main needs explicit sums but pilot was built without: rebuild the pilot
      if (ix->pilot.d_tiles) (void)hipFree(ix->pilot.d_tiles);
      if (ix->pilot.d_exact) (void)hipFree(ix->pilot.d_exact);
      ix->pilot.d_tiles = nullptr;
      ix->pilot.d_exact = nullptr;
      rc = make_storage(ix.get(), ix->pilot, pilot_codes, pilot_corr, n_pilot, 0, false);
      if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
    }
  } else {
    rc = make_storage(ix.get(), ix->main, codes, corr, n_rows, row_base, true);
    if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
  }
  *out = ix.release();
  return BBQ_OK;
}

int bbq_index_create(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits,
                     double centroid_dp, int32_t device, bbq_index **out) {
  return bbq_index_create_shard(codes, corr, n_rows, dim, index_bits, centroid_dp, 0, nullptr, nullptr, 0, device, out);
}

void bbq_index_destroy(bbq_index *ix) {
  if (!ix) return;
  if (ix->multi) { multi_destroy(ix); return; }
  if (ix->ctx) {
    std::lock_guard<std::mutex> lk(ix->ctx->mu);
    destroy_unlocked(ix);
  } else {
    destroy_unlocked(ix);
  }
}

int64_t bbq_index_size(const bbq_index *ix) { return ix ? ix->n_rows : 0; }
int32_t bbq_index_dimension(const bbq_index *ix) { return ix ? ix->dim : 0; }
int32_t bbq_index_bytes_per_row(const bbq_index *ix) { return ix ? ix->bytes_per_row : 0; }
int32_t bbq_index_bits(const bbq_index *ix) { return ix ? ix->index_bits : 0; }

}  // extern "C"

// bbq_search_batch, and - with `feed` - bbq_search_raw_batch: the quantized queries of a sub-batch are waited for right before it is
// enqueued
static int search_batch_impl(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                             int32_t sim, int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n, RawFeed *feed, int32_t *bad_query) {
  int rc = validate_query_args(ix, n_queries, qquant, qcorr, query_bits, sim, k, feed != nullptr);
  if (rc != BBQ_OK) return rc;
  if (n_queries > 0 && !out_n) return fail(BBQ_ERR_INVALID_ARG, "out_n is null");
  for (int32_t i = 0; i < n_queries; ++i) out_n[i] = 0;
  if (k == 0 || n_queries == 0) return BBQ_OK;  // src/binaryQuantizationFormat.ts:332-334
  if (!out_idx || !out_score) return fail(BBQ_ERR_INVALID_ARG, "output arrays are null");
  if (ix->multi) {
    ix->stats.candidates = 0;
    ix->stats.dense_fallbacks = 0;
    ix->stats.host_replays = 0;
    if (ix->n_rows == 0) return BBQ_OK;
    return multi_search_batch(ix, n_queries, qquant, qcorr, query_bits, sim, k, out_idx, out_score, out_n);
  }
  if (ix->has_pilot || ix->row_base != 0)
    return fail(BBQ_ERR_INVALID_ARG, "bbq_search on a non-root shard: use bbq_shard_scan + bbq_replay");
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  rc = settle_shard_slots(ix->ctx, nullptr);  // an asynchronous sharded scan on this device may have left slots busy
  if (rc != BBQ_OK) return rc;
  ix->stats.candidates = 0;
  ix->stats.dense_fallbacks = 0;
  ix->stats.host_replays = 0;
  if (ix->n_rows == 0) return BBQ_OK;

  BatchCtx c{ix, qquant, qcorr, 0, query_bits == 1 ? 1 : 0, sim, k};
  if (feed) {  // the values are still being produced: the kernel variant follows from the bit width they are quantized to
    const int pq = query_bits <= 1 ? 1 : query_bits <= 2 ? 2 : query_bits <= 4 ? 4 : 8;
    c.planes = ix->store_bits == 1 ? pq : ix->store_bits == 8 ? 8 : (query_bits <= 4 ? 4 : 8);
    c.maxq = (1 << query_bits) - 1;
  } else {
    c.planes = planes_of_call(ix, qquant, (int64_t)n_queries * ix->dim, query_bits == 1);
    c.maxq = c.planes <= 4 ? 15 : max_value(qquant, (int64_t)n_queries * ix->dim);
  }
  const int64_t keff = std::min<int64_t>(k, ix->n_rows);
  c.k = k;
  if (keff > kMaxFastK || ix->opt_force_dense) {
    for (int32_t i = 0; i < n_queries; ++i) {
      if (feed && (rc = feed->wait(i, 1, bad_query)) != BBQ_OK) return rc;
      rc = dense_search_one(c, i, out_idx + (int64_t)i * k, out_score + (int64_t)i * k, out_n + i);
      if (rc != BBQ_OK) return rc;
    }
    return BBQ_OK;
  }
  // thresholds are order statistics of rank k2 = min(k, N): selecting with a larger k would be wrong.
  // cs drives the device (k2); c (the caller's k) strides the outputs and sizes the replayed heap.
  BatchCtx cs = c;
  // up to kFinalSelectMax the device runs with rank keff + 1 and the last finalize launch selects and sorts the answer itself
  // (FinalizeArgs::final_out); the host replays the heap only for queries with equal scores in or at the edge of their answer
  const int64_t final_k = (keff <= kFinalSelectMax && ix->opt_device_select) ? keff : 0;
  cs.k = final_k > 0 ? keff + 1 : keff;
  build_plan(ix, cs.k, final_k, final_k > 0 && n_queries <= ix->opt_latency_queries);
  if (n_queries == 1 && ix->plan.latency && !feed) {
    bool done = false;
    rc = search_latency_presampled(c, cs, out_idx, out_score, out_n, &done);
    if (rc != BBQ_OK || done) return rc;
    rc = search_latency_chain(c, cs, out_idx, out_score, out_n, &done);
    if (rc != BBQ_OK || done) return rc;
  }
  const int Q = effective_batch(ix, n_queries);
  const int nslots = std::min(std::max(1, ix->opt_slots), kMaxSlots);
  const int64_t nsub = ((int64_t)n_queries + Q - 1) / Q;
  auto fail_out = [&](int code) {
    for (int i = 0; i < kMaxSlots; ++i)  // never leave pool jobs pointing at a caller's buffers
      while (ix->slots[i].pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    drain(ix);
    for (int i = 0; i < kMaxSlots; ++i) ix->slots[i].busy = ix->slots[i].replaying = false;
    return code;
  };
  for (int64_t i = 0; i < nsub; ++i) {
    Slot &s = ix->slots[i % nslots];
    rc = reclaim_slot(c, s, out_idx, out_score, out_n);
    if (rc != BBQ_OK) return fail_out(rc);
    const int nq = (int)std::min<int64_t>(Q, n_queries - i * Q);
    rc = ensure_slot(ix, s, nq, true);
    if (rc != BBQ_OK) return fail_out(rc);
    if (feed && (rc = feed->wait(i * Q, nq, bad_query)) != BBQ_OK) return fail_out(rc);
    rc = enqueue_subbatch(cs, s, i * Q, nq, nullptr);
    if (rc != BBQ_OK) return fail_out(rc);
    // hand finished sub-batches to the replay workers as early as possible (their slot is needed again soon)
    for (int j = 0; j < nslots; ++j) {
      Slot &t = ix->slots[j];
      if (&t != &s && t.busy && hipEventQuery(t.ev_done) == hipSuccess) {
        rc = begin_replay(c, t, out_idx, out_score, out_n);
        if (rc != BBQ_OK) return fail_out(rc);
      }
    }
  }
  for (int64_t i = std::max<int64_t>(0, nsub - nslots); i < nsub; ++i) {  // oldest first
    rc = reclaim_slot(c, ix->slots[i % nslots], out_idx, out_score, out_n);
    if (rc != BBQ_OK) return fail_out(rc);
  }
  return BBQ_OK;
}

extern "C" {

int bbq_search_batch(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                     int32_t sim, int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n) {
  clear_error();
  return search_batch_impl(ix, n_queries, qquant, qcorr, query_bits, sim, k, out_idx, out_score, out_n, nullptr, nullptr);
}

int bbq_search_raw_batch(bbq_index *ix, int32_t n_queries, const float *queries, const float *centroid, int32_t sim, int32_t query_bits,
                         double lambda, int32_t iters, int32_t n_threads, int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n,
                         uint8_t *qquant_out, double *qcorr_out, int32_t *bad_query) {
  clear_error();
  if (bad_query) *bad_query = -1;
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "目标向量集合不能为空");
  if (n_queries < 0) return fail(BBQ_ERR_INVALID_ARG, "n_queries < 0");
  if (n_queries > 0 && (!queries || !centroid)) return fail(BBQ_ERR_INVALID_ARG, "查询向量不能为空");
  if (k < 0) return fail(BBQ_ERR_NEGATIVE_K, "k值不能为负数");
  if (query_bits < 1 || query_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "queryBits必须在1-8之间");
  if (n_queries == 0) return BBQ_OK;
  const int dim = ix->dim;
  std::vector<uint8_t> own_q;
  std::vector<double> own_c;
  uint8_t *qq = qquant_out;
  double *qc = qcorr_out;
  if (!qq) { own_q.resize((size_t)n_queries * dim); qq = own_q.data(); }
  if (!qc) { own_c.resize((size_t)n_queries * 4); qc = own_c.data(); }
  const int T = n_threads > 0 ? n_threads : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
  // few queries, k == 0 (the quantizer's errors still surface, as in the reference's order of checks) or a multi-device handle
  // (its rounds take whole arrays): quantize first, then search
  if (n_queries <= 64 || k == 0 || ix->multi) {
    int rc = bbq_quantize_queries(queries, n_queries, dim, centroid, sim, query_bits, lambda, iters, T, qq, qc, bad_query);
    if (rc != BBQ_OK) return rc;
    return search_batch_impl(ix, n_queries, qq, qc, query_bits, sim, k, out_idx, out_score, out_n, nullptr, nullptr);
  }
  RawFeed feed;
  feed.queries = queries; feed.centroid = centroid; feed.n = n_queries; feed.dim = dim; feed.sim = sim; feed.qb = query_bits;
  feed.lambda = lambda; feed.iters = iters; feed.qq = qq; feed.qc = qc;
  // the argument checks of the quantizer, once, on this thread (the workers would only report "failed")
  {
    int rc = bbq_quantize_query(queries, dim, centroid, sim, query_bits, lambda, iters, qq, qc);
    if (rc != BBQ_OK) { if (bad_query) *bad_query = 0; return rc; }
  }
  feed.start(T);
  int rc = search_batch_impl(ix, n_queries, qq, qc, query_bits, sim, k, out_idx, out_score, out_n, &feed, bad_query);
  feed.join();
  return rc;
}

int bbq_search(bbq_index *ix, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim, int64_t k,
               int32_t *out_idx, float *out_score, int64_t *out_n) {
  return bbq_search_batch(ix, 1, qquant, qcorr, query_bits, sim, k, out_idx, out_score, out_n);
}

int bbq_score_rows(bbq_index *ix, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim,
                   int64_t row_begin, int64_t row_count, int32_t *out_qcdist, double *out_score64, float *out_score32) {
  clear_error();
  int rc = validate_query_args(ix, 1, qquant, qcorr, query_bits, sim, 0);
  if (rc != BBQ_OK) return rc;
  if (row_begin < 0 || row_count < 0 || row_begin + row_count > ix->n_rows)
    return fail(BBQ_ERR_INVALID_ARG, "向量索引 %lld 不存在", (long long)(row_begin + row_count - 1));
  if (row_count == 0) return BBQ_OK;
  if (ix->multi) return multi_score_rows(ix, qquant, qcorr, query_bits, sim, row_begin, row_count, out_qcdist, out_score64, out_score32);
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  BatchCtx c{ix, qquant, qcorr, planes_of_call(ix, qquant, ix->dim, query_bits == 1), query_bits == 1 ? 1 : 0, sim, 0};
  rc = ensure_aux_qbuf(ix->ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc != BBQ_OK) return rc;
  const int64_t qb = query_data_bytes(ix, c.planes);
  std::vector<uint8_t> hb((size_t)qb + sizeof(QueryParams));
  fill_query(ix, hb.data(), reinterpret_cast<QueryParams *>(hb.data() + qb), qquant, qcorr, c.planes, c.one_bit, sim);
  hipStream_t st = ix->aux_stream;
  HIPCHK(hipMemcpyAsync(ix->ctx->d_aux_qbuf, hb.data(), hb.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t piece_chunks = 1024;  // 1M rows per piece
  const int64_t c_first = row_begin / kChunkRows, c_last = (row_begin + row_count + kChunkRows - 1) / kChunkRows;
  const int64_t piece_rows = std::min(piece_chunks, c_last - c_first) * kChunkRows;
  DevMem m32, mqc, m64;
  HIPCHK(m32.alloc((size_t)piece_rows * 4));
  HIPCHK(mqc.alloc((size_t)piece_rows * 4));
  HIPCHK(m64.alloc((size_t)piece_rows * 8));
  float *d32 = m32.as<float>();
  int32_t *dqc = mqc.as<int32_t>();
  double *d64 = m64.as<double>();
  std::vector<float> h32((size_t)piece_rows);
  std::vector<int32_t> hqc((size_t)piece_rows);
  std::vector<double> h64((size_t)piece_rows);
  rc = BBQ_OK;
  for (int64_t cb = c_first; cb < c_last && rc == BBQ_OK; cb += piece_chunks) {
    const int64_t nc = std::min(piece_chunks, c_last - cb);
    ScanArgs a{};
    a.idx = launch_view(ix, ix->main);
    a.qplanes = reinterpret_cast<const uint4 *>(ix->ctx->d_aux_qbuf);
    a.qparams = reinterpret_cast<const QueryParams *>(ix->ctx->d_aux_qbuf + qb);
    a.chunk_begin = cb;
    a.row_id_base = ix->main.row_id_base;
    a.flags = ix->d_aux_flags;
    a.dense_score32 = d32;
    a.dense_qcdist = dqc;
    a.dense_score64 = d64;
    a.dense_stride = piece_rows;
    hipError_t e = launch_scan(a, c.planes, true, 1, (int)nc, st);
    const int64_t r0 = cb * kChunkRows, r1 = std::min((cb + nc) * kChunkRows, ix->main.view.n_rows);
    if (e == hipSuccess) e = hipMemcpyAsync(h32.data(), d32, (size_t)(r1 - r0) * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hqc.data(), dqc, (size_t)(r1 - r0) * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h64.data(), d64, (size_t)(r1 - r0) * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { rc = fail(BBQ_ERR_HIP, "bbq_score_rows: %s", hipGetErrorString(e)); break; }
    const int64_t lo = std::max(r0, row_begin), hi = std::min(r1, row_begin + row_count);
    for (int64_t r = lo; r < hi; ++r) {
      if (out_score32) out_score32[r - row_begin] = h32[(size_t)(r - r0)];
      if (out_qcdist) out_qcdist[r - row_begin] = hqc[(size_t)(r - r0)];
      if (out_score64) out_score64[r - row_begin] = h64[(size_t)(r - r0)];
    }
  }
  return rc;
}

int64_t bbq_shard_list_cap(const bbq_index *cix, int64_t k) {
  if (!cix || k <= 0 || cix->multi) return 0;
  bbq_index *ix = const_cast<bbq_index *>(cix);
  // the scan runs with rank k + 1 whenever it can leave shard-local answers (k <= kFinalSelectMax): size for that plan
  const int64_t keff = std::min<int64_t>(k, kMaxFastK);
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  if (keff <= kFinalSelectMax) build_plan(ix, keff + 1, keff);
  else build_plan(ix, keff);
  return ix->plan.list_cap;
}

int bbq_shard_scan_begin(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits, int32_t sim,
                         int64_t k, void *dev_packed, int64_t packed_cap, void *dev_offsets, void *dev_flags, void *dev_answers,
                         int64_t answers_stride) {
  clear_error();
  int rc = validate_query_args(ix, n_queries, qquant, qcorr, query_bits, sim, k);
  if (rc != BBQ_OK) return rc;
  if (n_queries <= 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan: n_queries must be positive");
  if (!dev_packed || !dev_offsets || !dev_flags || packed_cap <= 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan: null output buffers");
  if (k == 0 || k > kMaxFastK) return fail(BBQ_ERR_UNSUPPORTED, "bbq_shard_scan: k must be in 1..%lld", (long long)kMaxFastK);
  if (dev_answers && answers_stride < k + 3) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan: answers_stride must be at least k + 3");
  if (dev_answers && k > kFinalSelectMax)
    return fail(BBQ_ERR_UNSUPPORTED, "bbq_shard_scan: shard-local answers exist for k <= %d (pass dev_answers = NULL and merge the lists)", kFinalSelectMax);
  if (ix->multi) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan: the handle is a multi-device index (it shards by itself)");
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  if (ix->shard_begun - ix->shard_waited >= 2) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan_begin: two batches are already in flight on this index (wait for one first)");
  BatchCtx c{ix, qquant, qcorr, planes_of_call(ix, qquant, (int64_t)n_queries * ix->dim, query_bits == 1), query_bits == 1 ? 1 : 0, sim, k};
  // with answers the shard runs with rank k + 1, like the single index does: its last finalize launch then knows the (k + 1)-th largest
  // key of everything it has seen (the cut) and the rows above it.  Lists for rank k + 1 are supersets of the lists for rank k.
  const bool answers = dev_answers != nullptr;
  if (answers) { c.k = k + 1; build_plan(ix, k + 1, k); }
  else build_plan(ix, k);
  bbq_index::ShardSet &set = ix->shard_set[ix->shard_begun & 1];
  if (!set.done) {
    HIPCHK(hipEventCreateWithFlags(&set.done, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void **)&set.h_total, 8, hipHostMallocDefault));
  }
  // per-query lists with room for a flood (rows stored cluster by cluster); what travels is packed, so the headroom costs
  // device memory only
  const int64_t list_cap = ix->plan.list_cap + std::min<int64_t>(ix->plan.flood_cap, 65536);
  if (set.q_cap < n_queries || set.list_cap < list_cap) {  // per-query lists the finalize kernels build (the set is idle: its last batch was waited for)
    if (set.d_lists) HIPCHK(hipFree(set.d_lists));
    if (set.d_counts) HIPCHK(hipFree(set.d_counts));
    set.d_lists = nullptr;
    set.d_counts = nullptr;
    HIPCHK(hipMalloc((void **)&set.d_lists, (size_t)n_queries * (size_t)list_cap * 8));
    HIPCHK(hipMalloc((void **)&set.d_counts, (size_t)n_queries * 8 + 16));
    set.q_cap = n_queries;
    set.list_cap = list_cap;
  }
  const int Q = effective_batch(ix, n_queries);
  const int nslots = std::min(std::max(1, ix->opt_slots), kMaxSlots);
  const int64_t nsub = ((int64_t)n_queries + Q - 1) / Q;
  auto bail = [&](int code) {
    (void)settle_shard_slots(ix->ctx, nullptr);
    drain(ix);
    return code;
  };
  // slots other indexes (or the previous batch of this one) have left busy are retired one by one as they are needed: the device
  // keeps working on them while this batch is being enqueued behind
  for (int64_t i = 0; i < nsub; ++i) {
    Slot &s = ix->slots[i % nslots];
    if (s.busy) {
      const hipError_t e = hipEventSynchronize(s.ev_done);
      if (e != hipSuccess) return bail(fail(BBQ_ERR_HIP, "bbq_shard_scan_begin: %s", hipGetErrorString(e)));
      s.busy = false;
      account_timing(s.shard_owner ? s.shard_owner : ix, s);
      s.shard_owner = nullptr;
    }
    const int nq = (int)std::min<int64_t>(Q, n_queries - i * Q);
    rc = ensure_slot(ix, s, nq, false);
    if (rc != BBQ_OK) return bail(rc);
    ExtOut ext;
    ext.lists = set.d_lists + (size_t)(i * Q) * list_cap;
    ext.list_cap = list_cap;
    ext.counts = set.d_counts + (size_t)(i * Q) * 2;
    if (answers) {
      ext.answers = reinterpret_cast<uint64_t *>(dev_answers) + (size_t)(i * Q) * (size_t)answers_stride;
      ext.answers_stride = answers_stride;
    }
    rc = enqueue_subbatch(c, s, i * Q, nq, &ext);
    if (rc != BBQ_OK) return bail(rc);
    s.shard_owner = ix;
  }
  // the packing runs on the auxiliary stream behind the last sub-batch of every slot this batch has used
  hipStream_t aux = ix->aux_stream;
  for (int j = 0; j < nslots; ++j)
    if (ix->slots[j].busy && ix->slots[j].shard_owner == ix) HIPCHK(hipStreamWaitEvent(aux, ix->slots[j].ev_done, 0));
  // pack: [nq][list_cap] -> contiguous entries + offsets, what the host framework sends over RCCL
  int64_t *d_total = reinterpret_cast<int64_t *>(set.d_counts + (size_t)n_queries * 2);
  d_total = reinterpret_cast<int64_t *>(((uintptr_t)d_total + 7) & ~(uintptr_t)7);
  HIPCHK(launch_pack(set.d_counts, set.d_lists, list_cap, ix->plan.list_cap, n_queries, reinterpret_cast<int64_t *>(dev_offsets),
                     reinterpret_cast<int32_t *>(dev_flags), d_total, reinterpret_cast<uint64_t *>(dev_packed), packed_cap, aux));
  HIPCHK(hipMemcpyAsync(set.h_total, d_total, 8, hipMemcpyDeviceToHost, aux));
  HIPCHK(hipEventRecord(set.done, aux));
  set.packed_cap = packed_cap;
  set.in_flight = true;
  ix->shard_begun += 1;
  return BBQ_OK;
}

int bbq_shard_scan_wait(bbq_index *ix, int64_t *out_total) {
  clear_error();
  if (!ix || ix->multi || !ix->ctx) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan_wait: not a shard handle");
  if (out_total) *out_total = 0;
  hipEvent_t ev = nullptr;
  bbq_index::ShardSet *set = nullptr;
  {
    std::lock_guard<std::mutex> lk(ix->ctx->mu);
    if (ix->shard_begun == ix->shard_waited) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan_wait: no batch in flight");
    set = &ix->shard_set[ix->shard_waited & 1];
    ev = set->done;
  }
  // outside the device mutex: the next batch is being enqueued by another thread meanwhile
  hipError_t e = hipEventSynchronize(ev);
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  set->in_flight = false;
  ix->shard_waited += 1;
  if (e != hipSuccess) return fail(BBQ_ERR_HIP, "bbq_shard_scan_wait: %s", hipGetErrorString(e));
  const int64_t total = *set->h_total;
  if (out_total) *out_total = total;
  if (total > set->packed_cap) return fail(BBQ_ERR_OOM, "bbq_shard_scan: %lld candidates do not fit packed_cap %lld", (long long)total, (long long)set->packed_cap);
  return BBQ_OK;
}

int bbq_shard_scan(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                   int32_t sim, int64_t k, void *dev_packed, int64_t packed_cap, void *dev_offsets, void *dev_flags,
                   int64_t *out_total) {
  if (out_total) *out_total = 0;
  if (n_queries == 0) {
    clear_error();
    return validate_query_args(ix, n_queries, qquant, qcorr, query_bits, sim, k);
  }
  if (!out_total) return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan: null output buffers");
  int rc = bbq_shard_scan_begin(ix, n_queries, qquant, qcorr, query_bits, sim, k, dev_packed, packed_cap, dev_offsets, dev_flags, nullptr, 0);
  if (rc != BBQ_OK) return rc;
  rc = bbq_shard_scan_wait(ix, out_total);
  if (rc != BBQ_OK) return rc;
  std::lock_guard<std::mutex> lk(ix->ctx->mu);  // the synchronous form leaves nothing in flight: timings are booked when it returns
  return settle_shard_slots(ix->ctx, ix);
}

int bbq_get_stats(bbq_index *ix, bbq_stats *out) {
  if (!ix || !out) return fail(BBQ_ERR_INVALID_ARG, "bbq_get_stats: null");
  if (ix->multi) return multi_get_stats(ix, out);
  if (ix->ctx) {
    std::lock_guard<std::mutex> lk(ix->ctx->mu);
    int prev = -1;  // a getter must not change the calling thread's current device
    (void)hipGetDevice(&prev);
    HIPCHK(hipSetDevice(ix->device));
    int rc = settle_shard_slots(ix->ctx, ix);  // timings of an asynchronous scan are booked when its slots are retired
    if (prev >= 0 && prev != ix->device) (void)hipSetDevice(prev);
    if (rc != BBQ_OK) return rc;
    *out = ix->stats;
    return BBQ_OK;
  }
  *out = ix->stats;
  return BBQ_OK;
}
int bbq_reset_stats(bbq_index *ix) {
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "bbq_reset_stats: null");
  if (ix->multi) return multi_reset_stats(ix);
  ix->stats = bbq_stats{};
  return BBQ_OK;
}

int bbq_set_option(bbq_index *ix, const char *name, int64_t v) {
  if (!ix || !name) return fail(BBQ_ERR_INVALID_ARG, "bbq_set_option: null");
  if (ix->multi) return multi_set_option(ix, name, v);
  const std::string n(name);
  if (n == "batch_queries" && v >= 0 && v <= 1024) ix->opt_batch = (int)v;  // 0: by index size
  else if (n == "pipeline_slots" && v >= 1 && v <= kMaxSlots) ix->opt_slots = (int)v;
  else if (n == "segment_growth" && v >= 2 && v <= 1024) { ix->opt_growth = (int)v; ix->plan.k = -1; }
  else if (n == "first_segment_rows" && v >= 1024 && v <= 8192 && v % kChunkRows == 0) { ix->opt_s0 = v; ix->plan.k = -1; }
  else if (n == "resident_interleave" && (v == 0 || v == 1)) ix->opt_resident_interleave = (int)v;
  else if (n == "resident_mb" && v >= -1 && v <= 1 << 20) ix->opt_resident_mb = (int)v;
  else if (n == "replay_threads" && v >= 1 && v <= 256) ix->opt_replay_threads = (int)v;
  else if (n == "force_dense" && (v == 0 || v == 1)) ix->opt_force_dense = (int)v;
  else if (n == "device_select" && (v == 0 || v == 1)) ix->opt_device_select = (int)v;
  else if (n == "latency_queries" && v >= 0 && v <= 1024) ix->opt_latency_queries = (int)v;
  else if (n == "append_last" && (v == 0 || v == 1)) ix->opt_append_last = (int)v;
  else if (n == "latency_append" && (v == 0 || v == 1)) ix->opt_latency_append = (int)v;
  else if (n == "latency_fused" && (v == 0 || v == 1)) ix->opt_latency_fused = (int)v;
  else if (n == "latency_presample" && (v == 0 || v == 1)) ix->opt_latency_presample = (int)v;
  else if (n == "latency_growth" && v >= 2 && v <= 4096) ix->opt_latency_growth = (int)v;
  else if (n == "sweep_share" && (v == 1 || v == 4 || v == 8 || v == 32)) ix->opt_share = (int)v;
  else if (n == "flood_rows" && v >= 0 && v <= (1 << 24)) ix->opt_flood = (v + 1023) / 1024 * 1024;
  else return fail(BBQ_ERR_INVALID_ARG, "bbq_set_option: unknown option or value out of range: %s=%lld", name, (long long)v);
  ix->plan.k = -1;  // workspace is grow-only and re-checked by ensure_slot on the next call
  return BBQ_OK;
}

}  // extern "C"
