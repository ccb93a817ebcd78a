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
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "bbq_internal.h"
#include "bbq_launch.h"

using namespace bbq;

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(BBQ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

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
constexpr int64_t kMaxFastK = 2048;  // beyond this the dense path is used (finalize LDS key buffer)

struct Storage {
  uint8_t *d_tiles = nullptr;
  double *d_exact = nullptr;  // kLayoutCompact: exact corrections, gathered for the rows whose bound passes
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
};

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr, ev_big = nullptr;
  // capacities the buffers below were allocated for
  int q_cap = 0;
  int64_t qbuf_bytes = 0, chunks_cap = 0, slots_cap = 0, dense_cap = 0, list_cap = 0, k_cap = 0, hprefix = 0, flood_cap = 0;
  uint8_t *d_qbuf = nullptr, *h_qbuf = nullptr;
  uint32_t *d_theta = nullptr, *d_flags = nullptr, *d_counts = nullptr, *d_topk = nullptr;
  int32_t *d_topk_counts = nullptr, *d_list_counts = nullptr, *h_list_counts = nullptr;
  uint64_t *d_entries = nullptr, *d_lists = nullptr, *h_lists = nullptr, *d_ovf = nullptr;
  uint32_t *d_ovf_counts = nullptr;
  float *d_dense0 = nullptr;
  // in-flight sub-batch: busy = device work enqueued and not yet collected; replaying = host replay jobs outstanding
  bool busy = false;
  bool replaying = false;
  std::atomic<int> pending{0};
  std::vector<std::vector<uint64_t>> tails;
  std::vector<int> dense_q;
  int nq = 0;
  int64_t q_first = 0;
  bool timed = false;
  int64_t timed_rows = 0, timed_bytes = 0;
};

}  // namespace

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
};

struct bbq_index {
  int device = 0;
  DeviceCtx *ctx = nullptr;
  Slot *slots = nullptr;  // = ctx->slots
  int32_t dim = 0, pb = 0, w16 = 0, tile_stride = 0, has_x1 = 0, bytes_per_row = 0, layout = 0, want_compact = 1;
  int64_t n_rows = 0, row_base = 0;
  double centroid_dp = 0;
  bool has_pilot = false;
  Storage pilot, main;
  Plan plan;
  hipStream_t aux_stream = nullptr;  // = ctx->aux_stream
  uint8_t *d_aux_qbuf = nullptr;     // = ctx->d_aux_qbuf
  uint32_t *d_aux_flags = nullptr;
  float *d_dense_all = nullptr;
  int64_t dense_all_cap = 0;
  // bbq_shard_scan: per-query lists before packing
  uint64_t *d_shard_lists = nullptr;
  int32_t *d_shard_counts = nullptr;
  int64_t shard_q_cap = 0, shard_list_cap = 0;
  // options
  int opt_batch = 32, opt_slots = 2, opt_growth = 8, opt_force_dense = 0, opt_share = 1;
  // host threads replaying the heaps of one sub-batch: half the cores, at most 8 (a batch of 32 answers 1.4x sooner than with 1)
  int opt_replay_threads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
  int64_t opt_s0 = 4096;
  // flood tier: candidates one query may pile up beyond the planned list (rows stored cluster by cluster make the
  // query's own cluster beat a threshold that was derived from other clusters) before it has to take the dense path
  int64_t opt_flood = 262144;
  bbq_stats stats{};
};

namespace {

// per query: bit-planes (up to 8) + int8 values in MFMA fragment order + score uniforms + group maxima
int64_t qbuf_bytes_per_query_w(int w16) { return (int64_t)w16 * 8 * 16 + (int64_t)w16 * 128 + (int64_t)sizeof(QueryParams) + 16; }

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

// ------------------------------------------------------------------------------------------------ storage

int make_storage(bbq_index *ix, Storage &st, const uint8_t *codes, const double *corr, int64_t n_rows, int64_t row_id_base,
                 bool check_x1) {
  const int64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  const int64_t pb = ix->pb;
  DevMem m_codes, m_corr, m_mis;
  hipStream_t s = ix->aux_stream;
  if (n_rows > 0) {
    HIPCHK(m_codes.alloc((size_t)(n_rows * pb)));
    HIPCHK(m_corr.alloc((size_t)n_rows * 32));
    HIPCHK(hipMemcpyAsync(m_codes.p, codes, (size_t)(n_rows * pb), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m_corr.p, corr, (size_t)n_rows * 32, hipMemcpyHostToDevice, s));
  }
  uint8_t *d_codes = m_codes.as<uint8_t>();
  double *d_corr = m_corr.as<double>();
  if (check_x1) {
    // quantizedComponentSum of a 1-bit row is its popcount (src/optimizedScalarQuantizer.ts:204-209); if that
    // holds for every row the 8 bytes need not be stored or read.  Decided once per index, over all storages.
    uint32_t mis = 0;
    HIPCHK(m_mis.alloc(4));
    uint32_t *d_mis = m_mis.as<uint32_t>();
    HIPCHK(hipMemsetAsync(d_mis, 0, 4, s));
    HIPCHK(launch_check_x1(d_codes, d_corr, n_rows, (int32_t)pb, d_mis, s));
    HIPCHK(hipMemcpyAsync(&mis, d_mis, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (mis) ix->has_x1 = 1;
  }
  // compact corrections (8 B/row streamed + exact side array) need the implicit component sum; otherwise inline
  ix->layout = (ix->want_compact && !ix->has_x1) ? kLayoutCompact : kLayoutInline;
  ix->tile_stride = ix->w16 * 1024 + (ix->layout == kLayoutCompact ? 512 : 1536 + (ix->has_x1 ? 512 : 0));
  ix->bytes_per_row = ix->tile_stride / kTileRows;
  st.row_id_base = row_id_base;
  st.view.n_rows = n_rows;
  st.view.w16 = ix->w16;
  st.view.tile_stride = ix->tile_stride;
  st.view.has_x1 = ix->has_x1;
  st.view.dim = ix->dim;
  st.view.layout = ix->layout;
  if (n_tiles > 0) {
    HIPCHK(hipMalloc((void **)&st.d_tiles, (size_t)(n_tiles * ix->tile_stride)));
    if (ix->layout == kLayoutCompact) HIPCHK(hipMalloc((void **)&st.d_exact, (size_t)(n_tiles * kTileRows) * 32));
    HIPCHK(launch_retile(d_codes, d_corr, n_rows, (int32_t)pb, st.d_tiles, ix->w16, ix->tile_stride, ix->has_x1, ix->layout, st.d_exact, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  st.view.exact = st.d_exact;
  st.view.tiles = st.d_tiles;
  HIPCHK(hipStreamSynchronize(s));  // the scratch rows are released on return
  return BBQ_OK;
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
    const int64_t nb = (before * ix->opt_growth - rows_before) / kChunkRows * kChunkRows;
    if (before > 0 && nb > b && nb <= R / 2) e = nb;
    const int cap = cap_for(p.k, before);
    const int64_t rows = e - b;
    p.segs.push_back(Segment{storage, b / kChunkRows, (rows + kChunkRows - 1) / kChunkRows, rows, false, emit, true, false, cap});
    expected += (double)p.k * (double)rows / (double)before;
    b = e;
  }
}

void build_plan(bbq_index *ix, int64_t k) {
  Plan &p = ix->plan;
  if (p.k == k) return;
  p = Plan();
  p.k = k;
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
  p.flood_cap = std::min<int64_t>(ix->opt_flood, (ix->main.view.n_rows + 1023) / 1024 * 1024);
}

// ------------------------------------------------------------------------------------------------ slots

void free_slot_buffers(Slot &s) {
  if (s.d_qbuf) (void)hipFree(s.d_qbuf);
  if (s.h_qbuf) (void)hipHostFree(s.h_qbuf);
  if (s.d_theta) (void)hipFree(s.d_theta);
  if (s.d_counts) (void)hipFree(s.d_counts);
  if (s.d_topk) (void)hipFree(s.d_topk);
  if (s.h_list_counts) (void)hipHostFree(s.h_list_counts);
  if (s.d_entries) (void)hipFree(s.d_entries);
  if (s.d_lists) (void)hipFree(s.d_lists);
  if (s.h_lists) (void)hipHostFree(s.h_lists);
  if (s.d_dense0) (void)hipFree(s.d_dense0);
  if (s.d_ovf) (void)hipFree(s.d_ovf);
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
                  s.flood_cap >= p.flood_cap &&
                  (!own_lists || s.d_lists != nullptr);
  if (ok) return BBQ_OK;
  free_slot_buffers(s);
  const int Q = (std::max(nq, ix->opt_batch) + 31) / 32 * 32;  // multiple of 32: the MFMA query layout is per group of 32
  s.qbuf_bytes = qb;
  s.chunks_cap = std::max<int64_t>(p.max_chunks, 1);
  s.slots_cap = std::max<int64_t>(p.max_slots, 1);
  s.dense_cap = p.s0;
  s.list_cap = p.list_cap + (own_lists ? p.flood_cap : 0);  // own lists can take a flood; external ones are the caller's size
  s.flood_cap = p.flood_cap;
  s.k_cap = std::max<int64_t>(p.k, 1);
  s.hprefix = hprefix;
  HIPCHK(hipMalloc((void **)&s.d_qbuf, (size_t)(Q * qb)));
  HIPCHK(hipHostMalloc((void **)&s.h_qbuf, (size_t)(Q * qb), hipHostMallocDefault));
  // theta | flags | topk_counts | list_counts | ovf_counts live in one control block so that one memset resets a sub-batch
  HIPCHK(hipMalloc((void **)&s.d_theta, (size_t)Q * 24));
  s.d_flags = s.d_theta + Q;
  s.d_topk_counts = reinterpret_cast<int32_t *>(s.d_theta + 2 * (size_t)Q);
  s.d_list_counts = reinterpret_cast<int32_t *>(s.d_theta + 3 * (size_t)Q);
  s.d_ovf_counts = s.d_theta + 5 * (size_t)Q;
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

// writes the bit-planes ([j][p] 16-byte blocks, packed like the rows: dim d -> byte d>>3, bit 7-(d&7)) and the
// score uniforms of one query into the staging buffer
void fill_query(const bbq_index *ix, uint8_t *planes_dst, QueryParams *pp, const uint8_t *q, const double *qc, int planes,
                int one_bit, int sim) {
  const int w16 = ix->w16;
  memset(planes_dst, 0, (size_t)w16 * planes * 16);
  for (int d = 0; d < ix->dim; ++d) {
    const uint8_t v = q[d];
    if (!v) continue;
    const int byte = d >> 3, j = byte >> 4, b = byte & 15;
    const uint8_t bit = (uint8_t)(0x80u >> (d & 7));
    for (int p = 0; p < planes; ++p)
      if ((v >> p) & 1) planes_dst[((size_t)j * planes + p) * 16 + b] |= bit;
  }
  const double FBS = 1.0 / 15.0;  // src/constants.ts:20
  pp->ay = qc[0];
  pp->ly = one_bit ? (qc[1] - qc[0]) : (qc[1] - qc[0]) * FBS;  // src/batchDotProduct.ts:498 / :574
  pp->y1 = qc[3];
  pp->qadd = qc[2];
  pp->cdp = ix->centroid_dp;
  pp->dimd = (double)ix->dim;
  pp->sim = sim;
  pp->one_bit = one_bit;
}

// MFMA shared sweep: the int8 query values in the order the code bits fall out of the packed words.  For 32-dim word
// g, half h, dword c, byte i the kernel extracts bit p = 4h + c + 8i of the little-endian word ((w >> (4h + c)) &
// 0x01010101), which is row byte 4g + (p >> 3), bit (p & 7), i.e. dimension 32g + 8*(p >> 3) + 7 - (p & 7)
// (MSB-first packing, src/optimizedScalarQuantizer.ts:420-446).  Layout: [group][g][h][n][16 B], n = query in its group of 32.
void fill_query_mfma(const bbq_index *ix, uint8_t *dst, int q_in_batch, const uint8_t *q) {
  const int words = ix->w16 * 4, group = q_in_batch / 32, n = q_in_batch % 32;
  uint8_t *gb = dst + (size_t)group * words * 2 * 32 * 16;
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

int validate_query_args(const bbq_index *ix, int32_t nq, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                        int32_t sim, int64_t k) {
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "目标向量集合不能为空");
  if (nq < 0) return fail(BBQ_ERR_INVALID_ARG, "n_queries < 0");
  if (nq > 0 && (!qquant || !qcorr)) return fail(BBQ_ERR_INVALID_ARG, "查询向量不能为空");
  if (k < 0) return fail(BBQ_ERR_NEGATIVE_K, "k值不能为负数");
  if (query_bits < 1 || query_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "queryBits必须在1-8之间");
  if (sim < 0 || sim > 2) return fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", sim);
  if (query_bits == 1)
    for (int64_t i = 0; i < (int64_t)nq * ix->dim; ++i)
      if (qquant[i] > 1) return fail(BBQ_ERR_INVALID_ARG, "1位量化值必须为0或1");
  return BBQ_OK;
}

// ------------------------------------------------------------------------------------------------ enqueue / complete

struct BatchCtx {
  bbq_index *ix;
  const uint8_t *qquant;
  const double *qcorr;
  int planes, one_bit, sim;
  int64_t k;
  int maxq = 255;  // largest quantized query value of the call (the MFMA sweep needs <= 127)
};

int enqueue_subbatch(const BatchCtx &c, Slot &s, int64_t q_first, int nq, uint64_t *d_lists_ext, int64_t list_cap_ext,
                     int32_t *d_counts_ext) {
  bbq_index *ix = c.ix;
  const Plan &p = ix->plan;
  const int64_t qb = (int64_t)ix->w16 * c.planes * 16;
  uint8_t *hp = s.h_qbuf;
  QueryParams *hq = reinterpret_cast<QueryParams *>(s.h_qbuf + (size_t)nq * qb);
  for (int i = 0; i < nq; ++i)
    fill_query(ix, hp + (size_t)i * qb, hq + i, c.qquant + (size_t)(q_first + i) * ix->dim, c.qcorr + (size_t)(q_first + i) * 4,
               c.planes, c.one_bit, c.sim);
  size_t bytes = (size_t)nq * qb + (size_t)nq * sizeof(QueryParams);
  const bool use_mfma = ix->opt_share == 32 && c.maxq <= 127;
  size_t off_qbytes = 0, off_qmax = 0;
  if (use_mfma) {  // second copy of the queries as int8 values in MFMA fragment order + per-group maxima for the pre-filter slack
    const int groups = (nq + 31) / 32;
    off_qbytes = (bytes + 15) / 16 * 16;
    const size_t qbytes_len = (size_t)groups * 32 * ix->w16 * 128;
    off_qmax = off_qbytes + qbytes_len;
    bytes = off_qmax + (size_t)groups * 16;
    memset(s.h_qbuf + off_qbytes, 0, qbytes_len);
    float *qm = reinterpret_cast<float *>(s.h_qbuf + off_qmax);
    for (int gidx = 0; gidx < groups; ++gidx) qm[4 * gidx] = qm[4 * gidx + 1] = qm[4 * gidx + 2] = qm[4 * gidx + 3] = 0.f;
    for (int i = 0; i < nq; ++i) {
      fill_query_mfma(ix, s.h_qbuf + off_qbytes, i, c.qquant + (size_t)(q_first + i) * ix->dim);
      float *m = qm + 4 * (i / 32);
      // upper bounds (rounded up) of the group's |ay|, |ly|, y1, |qadd - cdp|
      m[0] = std::max(m[0], (float)(fabs(hq[i].ay) * 1.000001));
      m[1] = std::max(m[1], (float)(fabs(hq[i].ly) * 1.000001));
      m[2] = std::max(m[2], (float)(fabs(hq[i].y1) * 1.000001));
      m[3] = std::max(m[3], (float)(fabs(hq[i].qadd - hq[i].cdp) * 1.000001));
    }
  }
  hipStream_t st = s.stream;
  HIPCHK(hipMemcpyAsync(s.d_qbuf, s.h_qbuf, bytes, hipMemcpyHostToDevice, st));
  uint64_t *d_lists = d_lists_ext ? d_lists_ext : s.d_lists;
  const int64_t list_cap = d_lists_ext ? list_cap_ext : s.list_cap;
  int32_t *d_list_counts = d_counts_ext ? d_counts_ext : s.d_list_counts;
  HIPCHK(hipMemsetAsync(s.d_theta, 0, (size_t)s.q_cap * 24, st));
  if (d_counts_ext) HIPCHK(hipMemsetAsync(d_counts_ext, 0, (size_t)nq * 8, st));

  s.timed = false;
  for (const Segment &g : p.segs) {
    const Storage &sto = g.storage == 0 ? ix->pilot : ix->main;
    ScanArgs a{};
    a.idx = sto.view;
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
    const int my_slot = (int)(&s - ix->slots);
    if (g.big && ix->ctx->last_big_slot >= 0 && ix->ctx->last_big_slot != my_slot)
      HIPCHK(hipStreamWaitEvent(st, ix->slots[ix->ctx->last_big_slot].ev_big, 0));  // one big sweep at a time on the device
    if (g.dominant) HIPCHK(hipEventRecord(s.ev0, st));
    const bool mfma_here = use_mfma && !g.dense && mfma_sweep_supported(a);
    if (mfma_here)
      HIPCHK(launch_scan_mfma(a, s.d_qbuf + off_qbytes, reinterpret_cast<const float *>(s.d_qbuf + off_qmax), nq, (int)g.n_chunks, st));
    else if (!g.dense && ix->opt_share > 1 && shared_sweep_supported(a, ix->opt_share))
      HIPCHK(launch_scan_shared(a, c.planes, ix->opt_share, nq, (int)g.n_chunks, st));
    else
      HIPCHK(launch_scan(a, c.planes, g.dense, nq, (int)g.n_chunks, st));
    if (g.big) {
      HIPCHK(hipEventRecord(s.ev_big, st));
      ix->ctx->last_big_slot = my_slot;
    }
    if (g.dominant) {
      HIPCHK(hipEventRecord(s.ev1, st));
      s.timed = true;
      s.timed_rows = g.rows * nq;
      // a shared sweep reads each row once for `share` queries
      const int share = mfma_here ? 32 : (!g.dense && ix->opt_share > 1 && shared_sweep_supported(a, ix->opt_share)) ? ix->opt_share : 1;
      s.timed_bytes = g.rows * ((nq + share - 1) / share) * (int64_t)ix->bytes_per_row;
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
    HIPCHK(launch_finalize(f, nq, st));
  }
  if (!d_lists_ext) {
    HIPCHK(hipMemcpyAsync(s.h_list_counts, s.d_list_counts, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpy2DAsync(s.h_lists, (size_t)s.hprefix * 8, s.d_lists, (size_t)s.list_cap * 8, (size_t)s.hprefix * 8, (size_t)nq,
                            hipMemcpyDeviceToHost, st));
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

// dense path for one query: every f32 score to the host, full replay of the reference loop
int dense_search_one(const BatchCtx &c, int64_t qi, int32_t *out_idx, float *out_score, int64_t *out_n) {
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
  const int64_t qb = (int64_t)ix->w16 * c.planes * 16;
  std::vector<uint8_t> hb((size_t)qb + sizeof(QueryParams));
  fill_query(ix, hb.data(), reinterpret_cast<QueryParams *>(hb.data() + qb), c.qquant + (size_t)qi * ix->dim, c.qcorr + (size_t)qi * 4,
             c.planes, c.one_bit, c.sim);
  hipStream_t st = ix->aux_stream;
  HIPCHK(hipMemcpyAsync(ix->ctx->d_aux_qbuf, hb.data(), hb.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  ScanArgs a{};
  a.idx = ix->main.view;
  a.qplanes = reinterpret_cast<const uint4 *>(ix->ctx->d_aux_qbuf);
  a.qparams = reinterpret_cast<const QueryParams *>(ix->ctx->d_aux_qbuf + qb);
  a.chunk_begin = 0;
  a.row_id_base = ix->main.row_id_base;
  a.flags = ix->d_aux_flags;
  a.dense_score32 = ix->d_dense_all;
  a.dense_stride = n;
  // gridDim.x is limited to 2^31-1: fine for any index that fits in HBM
  HIPCHK(launch_scan(a, c.planes, true, 1, (int)chunks, st));
  std::vector<float> h((size_t)std::max<int64_t>(n, 1));
  HIPCHK(hipMemcpyAsync(h.data(), ix->d_dense_all, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
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
  for (int i = 0; i < nq; ++i) {
    const int32_t cnt = s.h_list_counts[2 * i], flags = s.h_list_counts[2 * i + 1];
    if (flags != 0) { s.dense_q.push_back(i); continue; }
    if (cnt > s.hprefix) {  // rare: fetch what the prefix copy did not cover
      s.tails[(size_t)i].resize((size_t)(cnt - s.hprefix));
      HIPCHK(hipMemcpy(s.tails[(size_t)i].data(), s.d_lists + (size_t)i * s.list_cap + s.hprefix, (size_t)(cnt - s.hprefix) * 8,
                       hipMemcpyDeviceToHost));
    }
  }
  const int64_t k = c.k, n_total = ix->main.row_id_base + ix->main.view.n_rows;
  Slot *sp = &s;
  auto replay_range = [sp, k, n_total, out_idx, out_score, out_n](int lo, int hi) {
    Slot &s = *sp;
    for (int i = lo; i < hi; ++i) {
      const int32_t cnt = s.h_list_counts[2 * i], flags = s.h_list_counts[2 * i + 1];
      if (flags != 0) continue;
      HeapReplay hr(k, n_total);
      const uint64_t *l = s.h_lists + (size_t)i * s.hprefix;
      const int64_t head = std::min<int64_t>(cnt, s.hprefix);
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
  const int T = std::min(ix->opt_replay_threads, nq);
  s.replaying = true;
  if (T <= 1) {
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
      cs.k = std::min<int64_t>(c.k, ix->n_rows);
      const int share = ix->opt_share;
      ix->opt_share = 1;
      int rc = enqueue_subbatch(cs, s, qi, 1, nullptr, 0, nullptr);
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

// frees what the index owns; the device context (streams, workspace) stays
void destroy_unlocked(bbq_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  if (ix->pilot.d_tiles) (void)hipFree(ix->pilot.d_tiles);
  if (ix->main.d_tiles) (void)hipFree(ix->main.d_tiles);
  if (ix->pilot.d_exact) (void)hipFree(ix->pilot.d_exact);
  if (ix->main.d_exact) (void)hipFree(ix->main.d_exact);
  if (ix->d_dense_all) (void)hipFree(ix->d_dense_all);
  if (ix->d_shard_lists) (void)hipFree(ix->d_shard_lists);
  if (ix->d_shard_counts) (void)hipFree(ix->d_shard_counts);
  delete ix;
}

}  // namespace

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
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create: out is null");
  *out = nullptr;
  if (n_rows < 0 || dim <= 0 || row_base < 0 || n_pilot < 0) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_create: bad size");
  if (n_rows > 0 && (!codes || !corr)) return fail(BBQ_ERR_INVALID_ARG, "目标向量集合不能为空");
  if (index_bits < 1 || index_bits > 8) return fail(BBQ_ERR_INVALID_ARG, "indexBits必须在1-8之间");
  if (index_bits != 1)
    return fail(BBQ_ERR_UNSUPPORTED,
                "indexBits=%d: the reference has no working batch scorer for multi-bit indexes (SURVEY A.7); only indexBits=1 is scored",
                index_bits);
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
  ix->pb = (dim + 7) / 8;
  ix->w16 = (ix->pb + 15) / 16;
  ix->n_rows = n_rows;
  ix->row_base = row_base;
  ix->centroid_dp = centroid_dp;
  ix->has_pilot = n_pilot > 0;
  {
    const char *e = getenv("BBQ_COMPACT_CORRECTIONS");  // 0: stream the exact f64 corrections (120 B/row at 768-d) instead
    ix->want_compact = (e && e[0] == '0') ? 0 : 1;
  }
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

// quantizeVectors on the device + index in place (bbq_build_kernels.hip)
int bbq_index_build(const float *vectors, int64_t n, int32_t dim, int32_t sim, double lambda, int32_t iters, int32_t device,
                    bbq_index **out, float *centroid, uint8_t *codes, double *corr, int64_t *bad_row, int32_t *bad_col) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_build: out is null");
  *out = nullptr;
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
  ix->pb = (dim + 7) / 8;
  ix->w16 = (ix->pb + 15) / 16;
  ix->n_rows = n;
  ix->row_base = 0;
  {
    const char *e = getenv("BBQ_COMPACT_CORRECTIONS");
    ix->want_compact = (e && e[0] == '0') ? 0 : 1;
  }
  ix->has_x1 = 0;  // a freshly quantized 1-bit row's component sum IS its popcount
  ix->layout = ix->want_compact ? kLayoutCompact : kLayoutInline;
  ix->tile_stride = ix->w16 * 1024 + (ix->layout == kLayoutCompact ? 512 : 1536);
  ix->bytes_per_row = ix->tile_stride / kTileRows;
  Storage &sto = ix->main;
  const int64_t n_tiles = npad / kTileRows;
  BCHK(hipMalloc((void **)&sto.d_tiles, (size_t)n_tiles * ix->tile_stride));
  if (ix->layout == kLayoutCompact) BCHK(hipMalloc((void **)&sto.d_exact, (size_t)npad * 32));
  if (corr) BCHK(hipMalloc((void **)&d_corr, (size_t)n * 32));
  BCHK(launch_build_quantize1(d_vT4, n, dim, npad, d_cen, sim, lambda, iters, sto.d_tiles, sto.d_exact, d_corr, ix->w16, ix->tile_stride,
                              ix->layout, st));  // :221-249
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
  sto.view.n_rows = n;
  sto.view.w16 = ix->w16;
  sto.view.tile_stride = ix->tile_stride;
  sto.view.has_x1 = 0;
  sto.view.dim = dim;
  sto.view.layout = ix->layout;
  ix->centroid_dp = bbq_centroid_dp(centroid, dim);  // getCentroidDP(undefined), :113-121
  rc = ensure_aux_qbuf(ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc != BBQ_OK) { destroy_unlocked(ix.release()); return rc; }
  *out = ix.release();
  return BBQ_OK;
}

int bbq_index_create(const uint8_t *codes, const double *corr, int64_t n_rows, int32_t dim, int32_t index_bits,
                     double centroid_dp, int32_t device, bbq_index **out) {
  return bbq_index_create_shard(codes, corr, n_rows, dim, index_bits, centroid_dp, 0, nullptr, nullptr, 0, device, out);
}

void bbq_index_destroy(bbq_index *ix) {
  if (!ix) return;
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

int bbq_search_batch(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                     int32_t sim, int64_t k, int32_t *out_idx, float *out_score, int64_t *out_n) {
  clear_error();
  int rc = validate_query_args(ix, n_queries, qquant, qcorr, query_bits, sim, k);
  if (rc != BBQ_OK) return rc;
  if (n_queries > 0 && !out_n) return fail(BBQ_ERR_INVALID_ARG, "out_n is null");
  for (int32_t i = 0; i < n_queries; ++i) out_n[i] = 0;
  if (k == 0 || n_queries == 0) return BBQ_OK;  // src/binaryQuantizationFormat.ts:332-334
  if (!out_idx || !out_score) return fail(BBQ_ERR_INVALID_ARG, "output arrays are null");
  if (ix->has_pilot || ix->row_base != 0)
    return fail(BBQ_ERR_INVALID_ARG, "bbq_search on a non-root shard: use bbq_shard_scan + bbq_replay");
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  ix->stats.candidates = 0;
  ix->stats.dense_fallbacks = 0;
  if (ix->n_rows == 0) return BBQ_OK;

  BatchCtx c{ix, qquant, qcorr, planes_for(qquant, (int64_t)n_queries * ix->dim), query_bits == 1 ? 1 : 0, sim, k};
  if (c.one_bit) c.planes = 1;
  c.maxq = c.planes <= 4 ? 15 : max_value(qquant, (int64_t)n_queries * ix->dim);
  const int64_t keff = std::min<int64_t>(k, ix->n_rows);
  c.k = k;
  if (keff > kMaxFastK || ix->opt_force_dense) {
    for (int32_t i = 0; i < n_queries; ++i) {
      rc = dense_search_one(c, i, out_idx + (int64_t)i * k, out_score + (int64_t)i * k, out_n + i);
      if (rc != BBQ_OK) return rc;
    }
    return BBQ_OK;
  }
  // thresholds are order statistics of rank k2 = min(k, N): selecting with a larger k would be wrong.
  // cs drives the device (k2); c (the caller's k) strides the outputs and sizes the replayed heap.
  BatchCtx cs = c;
  cs.k = keff;
  build_plan(ix, keff);
  const int Q = std::max(1, ix->opt_batch);
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
    rc = enqueue_subbatch(cs, s, i * Q, nq, nullptr, 0, nullptr);
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
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  BatchCtx c{ix, qquant, qcorr, planes_for(qquant, ix->dim), query_bits == 1 ? 1 : 0, sim, 0};
  if (c.one_bit) c.planes = 1;
  rc = ensure_aux_qbuf(ix->ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc != BBQ_OK) return rc;
  const int64_t qb = (int64_t)ix->w16 * c.planes * 16;
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
    a.idx = ix->main.view;
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
  if (!cix || k <= 0) return 0;
  bbq_index *ix = const_cast<bbq_index *>(cix);
  const int64_t keff = std::min<int64_t>(k, kMaxFastK);
  build_plan(ix, keff);
  return ix->plan.list_cap;
}

int bbq_shard_scan(bbq_index *ix, int32_t n_queries, const uint8_t *qquant, const double *qcorr, int32_t query_bits,
                   int32_t sim, int64_t k, void *dev_packed, int64_t packed_cap, void *dev_offsets, void *dev_flags,
                   int64_t *out_total) {
  clear_error();
  int rc = validate_query_args(ix, n_queries, qquant, qcorr, query_bits, sim, k);
  if (rc != BBQ_OK) return rc;
  if (out_total) *out_total = 0;
  if (n_queries == 0) return BBQ_OK;
  if (!dev_packed || !dev_offsets || !dev_flags || !out_total || packed_cap <= 0)
    return fail(BBQ_ERR_INVALID_ARG, "bbq_shard_scan: null output buffers");
  if (k == 0 || k > kMaxFastK) return fail(BBQ_ERR_UNSUPPORTED, "bbq_shard_scan: k must be in 1..%lld", (long long)kMaxFastK);
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  BatchCtx c{ix, qquant, qcorr, planes_for(qquant, (int64_t)n_queries * ix->dim), query_bits == 1 ? 1 : 0, sim, k};
  if (c.one_bit) c.planes = 1;
  build_plan(ix, k);
  // per-query lists with room for a flood (rows stored cluster by cluster); what travels is packed, so the headroom costs
  // device memory only
  const int64_t list_cap = ix->plan.list_cap + std::min<int64_t>(ix->plan.flood_cap, 65536);
  if (ix->shard_q_cap < n_queries || ix->shard_list_cap < list_cap) {  // per-query lists the finalize kernels build
    if (ix->d_shard_lists) HIPCHK(hipFree(ix->d_shard_lists));
    if (ix->d_shard_counts) HIPCHK(hipFree(ix->d_shard_counts));
    ix->d_shard_lists = nullptr;
    ix->d_shard_counts = nullptr;
    HIPCHK(hipMalloc((void **)&ix->d_shard_lists, (size_t)n_queries * (size_t)list_cap * 8));
    HIPCHK(hipMalloc((void **)&ix->d_shard_counts, (size_t)n_queries * 8 + 16));
    ix->shard_q_cap = n_queries;
    ix->shard_list_cap = list_cap;
  }
  const int Q = std::max(1, ix->opt_batch);
  const int nslots = std::min(std::max(1, ix->opt_slots), kMaxSlots);
  const int64_t nsub = ((int64_t)n_queries + Q - 1) / Q;
  auto retire = [&](Slot &s) -> int {
    HIPCHK(hipEventSynchronize(s.ev_done));
    s.busy = false;
    account_timing(ix, s);
    return BBQ_OK;
  };
  for (int64_t i = 0; i < nsub; ++i) {
    Slot &s = ix->slots[i % nslots];
    if (s.busy && (rc = retire(s)) != BBQ_OK) { drain(ix); return rc; }
    const int nq = (int)std::min<int64_t>(Q, n_queries - i * Q);
    rc = ensure_slot(ix, s, nq, false);
    if (rc != BBQ_OK) { drain(ix); return rc; }
    rc = enqueue_subbatch(c, s, i * Q, nq, ix->d_shard_lists + (size_t)(i * Q) * list_cap, list_cap, ix->d_shard_counts + (size_t)(i * Q) * 2);
    if (rc != BBQ_OK) { drain(ix); return rc; }
  }
  for (int i = 0; i < nslots; ++i)
    if (ix->slots[i].busy && (rc = retire(ix->slots[i])) != BBQ_OK) { drain(ix); return rc; }
  // pack: [nq][list_cap] -> contiguous entries + offsets, what the host framework sends over RCCL
  int64_t *d_total = reinterpret_cast<int64_t *>(ix->d_shard_counts + (size_t)n_queries * 2);
  d_total = reinterpret_cast<int64_t *>(((uintptr_t)d_total + 7) & ~(uintptr_t)7);
  HIPCHK(launch_pack(ix->d_shard_counts, ix->d_shard_lists, list_cap, ix->plan.list_cap, n_queries, reinterpret_cast<int64_t *>(dev_offsets),
                     reinterpret_cast<int32_t *>(dev_flags), d_total, reinterpret_cast<uint64_t *>(dev_packed), packed_cap, ix->aux_stream));
  int64_t total = 0;
  HIPCHK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ix->aux_stream));
  HIPCHK(hipStreamSynchronize(ix->aux_stream));
  *out_total = total;
  if (total > packed_cap) return fail(BBQ_ERR_OOM, "bbq_shard_scan: %lld candidates do not fit packed_cap %lld", (long long)total, (long long)packed_cap);
  return BBQ_OK;
}

int bbq_get_stats(bbq_index *ix, bbq_stats *out) {
  if (!ix || !out) return fail(BBQ_ERR_INVALID_ARG, "bbq_get_stats: null");
  *out = ix->stats;
  return BBQ_OK;
}
int bbq_reset_stats(bbq_index *ix) {
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "bbq_reset_stats: null");
  ix->stats = bbq_stats{};
  return BBQ_OK;
}

int bbq_set_option(bbq_index *ix, const char *name, int64_t v) {
  if (!ix || !name) return fail(BBQ_ERR_INVALID_ARG, "bbq_set_option: null");
  const std::string n(name);
  if (n == "batch_queries" && v >= 1 && v <= 1024) ix->opt_batch = (int)v;
  else if (n == "pipeline_slots" && v >= 1 && v <= kMaxSlots) ix->opt_slots = (int)v;
  else if (n == "segment_growth" && v >= 2 && v <= 1024) { ix->opt_growth = (int)v; ix->plan.k = -1; }
  else if (n == "first_segment_rows" && v >= 1024 && v <= 8192 && v % kChunkRows == 0) { ix->opt_s0 = v; ix->plan.k = -1; }
  else if (n == "replay_threads" && v >= 1 && v <= 256) ix->opt_replay_threads = (int)v;
  else if (n == "force_dense" && (v == 0 || v == 1)) ix->opt_force_dense = (int)v;
  else if (n == "sweep_share" && (v == 1 || v == 4 || v == 8 || v == 32)) ix->opt_share = (int)v;
  else if (n == "flood_rows" && v >= 0 && v <= (1 << 24)) ix->opt_flood = (v + 1023) / 1024 * 1024;
  else return fail(BBQ_ERR_INVALID_ARG, "bbq_set_option: unknown option or value out of range: %s=%lld", name, (long long)v);
  ix->plan.k = -1;  // workspace is grow-only and re-checked by ensure_slot on the next call
  return BBQ_OK;
}

}  // extern "C"

/* ------------------------------------------------------------------ oversample + exact rerank */

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
  DeviceCtx *ctx = nullptr;
  int rc = get_ctx(device, &ctx);
  if (rc != BBQ_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(hipSetDevice(device));
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

/* ------------------------------------------------------------------ on-disk format (veb / vemb) */

namespace {

#pragma pack(push, 1)
struct MetaHeader {
  char magic[4];  // "BVEC" (COMPONENT_NAMES.BINARIZED_VECTOR, src/constants.ts:62-65)
  uint32_t version;
  // MetadataFormat, src/types.ts:92-113
  int32_t fieldNumber, vectorEncodingOrdinal, vectorSimilarityOrdinal, dimensions;
  int64_t vectorDataOffset, vectorDataLength, vectorCount;
  double centroidSquareMagnitude;
  // geometry of the tile records in the vector-data file (= the device layout, bbq_device.h)
  int32_t indexBits, layout, w16, tileStride, hasX1, tileRows;
  int64_t tilesBytes, exactBytes, rowBase;
};
#pragma pack(pop)
static_assert(sizeof(MetaHeader) == 104, "vemb header layout");
constexpr uint32_t kFileVersion = 1;

// FNV-1a over little-endian 64-bit words (tail zero-padded): one multiply per 8 bytes keeps up with the disk
uint64_t fnv64_words(const void *data, size_t n, uint64_t h) {
  const uint8_t *p = (const uint8_t *)data;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
  }
  if (i < n) {
    uint64_t w = 0;
    memcpy(&w, p + i, n - i);
    h = (h ^ w) * 0x100000001b3ull;
  }
  return h;
}
constexpr uint64_t kFnvSeed = 0xcbf29ce484222325ull;

int32_t expected_tile_stride(int32_t w16, int32_t layout, int32_t has_x1) {
  return w16 * 1024 + (layout == kLayoutCompact ? 512 : 1536 + (has_x1 ? 512 : 0));
}

struct FileCloser {
  FILE *f;
  ~FileCloser() { if (f) fclose(f); }
};

int read_meta(const char *prefix, MetaHeader *h, std::vector<float> *centroid, uint64_t *data_sum) {
  const std::string path = std::string(prefix) + ".vemb";
  FileCloser fc{fopen(path.c_str(), "rb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot open %s", path.c_str());
  if (fread(h, sizeof *h, 1, fc.f) != 1) return fail(BBQ_ERR_INVALID_ARG, "%s: truncated header", path.c_str());
  if (memcmp(h->magic, "BVEC", 4) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: not a BVEC metadata file", path.c_str());
  if (h->version != kFileVersion) return fail(BBQ_ERR_UNSUPPORTED, "%s: format version %u (this build reads %u)", path.c_str(), h->version, kFileVersion);
  if (h->dimensions <= 0 || h->dimensions > (1 << 24) || h->vectorCount < 0 || h->rowBase < 0 || h->indexBits != 1 || h->tileRows != kTileRows ||
      (h->layout != kLayoutCompact && h->layout != kLayoutInline) || (h->hasX1 != 0 && h->hasX1 != 1) ||
      h->vectorSimilarityOrdinal < 0 || h->vectorSimilarityOrdinal > 2)
    return fail(BBQ_ERR_INVALID_ARG, "%s: header fields out of range", path.c_str());
  const int32_t pb = (h->dimensions + 7) / 8;
  const int64_t n_tiles = (h->vectorCount + kTileRows - 1) / kTileRows;
  if (h->w16 != (pb + 15) / 16 || h->tileStride != expected_tile_stride(h->w16, h->layout, h->hasX1) ||
      (h->layout == kLayoutCompact && h->hasX1) || h->tilesBytes != n_tiles * h->tileStride ||
      h->exactBytes != (h->layout == kLayoutCompact ? n_tiles * kTileRows * 32 : 0) ||
      h->vectorDataLength != h->tilesBytes + h->exactBytes || h->vectorDataOffset < 0)
    return fail(BBQ_ERR_INVALID_ARG, "%s: tile geometry does not match dimensions/vectorCount", path.c_str());
  std::vector<float> cen((size_t)h->dimensions);
  uint64_t sums[2];
  if (fread(cen.data(), 4, cen.size(), fc.f) != cen.size() || fread(sums, 8, 2, fc.f) != 2)
    return fail(BBQ_ERR_INVALID_ARG, "%s: truncated", path.c_str());
  uint64_t m = fnv64_words(h, sizeof *h, kFnvSeed);
  m = fnv64_words(cen.data(), cen.size() * 4, m);
  m = fnv64_words(&sums[0], 8, m);
  if (m != sums[1]) return fail(BBQ_ERR_INVALID_ARG, "%s: metadata checksum mismatch", path.c_str());
  if (centroid) centroid->swap(cen);
  if (data_sum) *data_sum = sums[0];
  return BBQ_OK;
}

}  // namespace

extern "C" {

int bbq_index_save(bbq_index *ix, const char *prefix, const float *centroid, int32_t sim) {
  clear_error();
  if (!ix || !prefix || !centroid) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_save: null argument");
  if (sim < 0 || sim > 2) return fail(BBQ_ERR_INVALID_ARG, "不支持的相似性函数: %d", sim);
  if (ix->has_pilot) return fail(BBQ_ERR_UNSUPPORTED, "bbq_index_save: a shard with a pilot replica cannot be saved");
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  const int64_t n_tiles = (ix->n_rows + kTileRows - 1) / kTileRows;
  MetaHeader h{};
  memcpy(h.magic, "BVEC", 4);
  h.version = kFileVersion;
  h.vectorSimilarityOrdinal = sim;
  h.dimensions = ix->dim;
  h.vectorCount = ix->n_rows;
  h.centroidSquareMagnitude = ix->centroid_dp;
  h.indexBits = 1;
  h.layout = ix->layout;
  h.w16 = ix->w16;
  h.tileStride = ix->tile_stride;
  h.hasX1 = ix->has_x1;
  h.tileRows = kTileRows;
  h.tilesBytes = n_tiles * ix->tile_stride;
  h.exactBytes = ix->layout == kLayoutCompact ? n_tiles * kTileRows * 32 : 0;
  h.rowBase = ix->row_base;
  h.vectorDataOffset = 0;
  h.vectorDataLength = h.tilesBytes + h.exactBytes;
  const std::string dpath = std::string(prefix) + ".veb", mpath = std::string(prefix) + ".vemb";
  uint64_t dsum = kFnvSeed;
  {
    FileCloser fc{fopen(dpath.c_str(), "wb")};
    if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot create %s", dpath.c_str());
    const size_t piece = 64u << 20;  // multiple of 8: the checksum words never straddle pieces
    std::vector<uint8_t> buf(piece);
    const uint8_t *src[2] = {ix->main.d_tiles, (const uint8_t *)ix->main.d_exact};
    const int64_t len[2] = {h.tilesBytes, h.exactBytes};
    for (int part = 0; part < 2; ++part) {
      for (int64_t o = 0; o < len[part]; o += (int64_t)piece) {
        const size_t m = (size_t)std::min<int64_t>((int64_t)piece, len[part] - o);
        HIPCHK(hipMemcpy(buf.data(), src[part] + o, m, hipMemcpyDeviceToHost));
        dsum = fnv64_words(buf.data(), m, dsum);
        if (fwrite(buf.data(), 1, m, fc.f) != m) return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", dpath.c_str());
      }
    }
    if (fflush(fc.f) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", dpath.c_str());
  }
  FileCloser fc{fopen(mpath.c_str(), "wb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot create %s", mpath.c_str());
  uint64_t m = fnv64_words(&h, sizeof h, kFnvSeed);
  m = fnv64_words(centroid, (size_t)ix->dim * 4, m);
  m = fnv64_words(&dsum, 8, m);
  const uint64_t sums[2] = {dsum, m};
  if (fwrite(&h, sizeof h, 1, fc.f) != 1 || fwrite(centroid, 4, (size_t)ix->dim, fc.f) != (size_t)ix->dim || fwrite(sums, 8, 2, fc.f) != 2 ||
      fflush(fc.f) != 0)
    return fail(BBQ_ERR_INVALID_ARG, "%s: write failed", mpath.c_str());
  return BBQ_OK;
}

int bbq_index_file_info(const char *prefix, int64_t *n_rows, int32_t *dim, int32_t *sim, double *cdp, int64_t *row_base) {
  clear_error();
  if (!prefix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_file_info: null path");
  MetaHeader h;
  int rc = read_meta(prefix, &h, nullptr, nullptr);
  if (rc != BBQ_OK) return rc;
  if (n_rows) *n_rows = h.vectorCount;
  if (dim) *dim = h.dimensions;
  if (sim) *sim = h.vectorSimilarityOrdinal;
  if (cdp) *cdp = h.centroidSquareMagnitude;
  if (row_base) *row_base = h.rowBase;
  return BBQ_OK;
}

int bbq_index_load(const char *prefix, int32_t device, bbq_index **out, float *centroid_out) {
  clear_error();
  if (!out) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load: out is null");
  *out = nullptr;
  if (!prefix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_load: null path");
  MetaHeader h;
  std::vector<float> cen;
  uint64_t want_sum = 0;
  int rc = read_meta(prefix, &h, &cen, &want_sum);
  if (rc != BBQ_OK) return rc;
  if (h.rowBase + h.vectorCount > 0xFFFFFFFFll) return fail(BBQ_ERR_UNSUPPORTED, "more than 2^32 rows");
  const std::string dpath = std::string(prefix) + ".veb";
  FileCloser fc{fopen(dpath.c_str(), "rb")};
  if (!fc.f) return fail(BBQ_ERR_INVALID_ARG, "cannot open %s", dpath.c_str());
  if (fseeko(fc.f, 0, SEEK_END) != 0 || ftello(fc.f) < h.vectorDataOffset + h.vectorDataLength)
    return fail(BBQ_ERR_INVALID_ARG, "%s: shorter than vectorDataOffset + vectorDataLength", dpath.c_str());
  if (fseeko(fc.f, h.vectorDataOffset, SEEK_SET) != 0) return fail(BBQ_ERR_INVALID_ARG, "%s: seek failed", dpath.c_str());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BBQ_ERR_NO_DEVICE, "no HIP device available: libbbq has no CPU fallback (hipGetDeviceCount found %d)", ndev);
  if (device < 0 || device >= ndev) return fail(BBQ_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));
  DeviceCtx *ctx = nullptr;
  rc = get_ctx(device, &ctx);
  if (rc != BBQ_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->mu);
  std::unique_ptr<bbq_index> ix(new bbq_index());
  ix->device = device;
  ix->dim = h.dimensions;
  ix->pb = (h.dimensions + 7) / 8;
  ix->w16 = h.w16;
  ix->n_rows = h.vectorCount;
  ix->row_base = h.rowBase;
  ix->centroid_dp = h.centroidSquareMagnitude;
  ix->has_pilot = false;
  ix->want_compact = h.layout == kLayoutCompact;
  ix->layout = h.layout;
  ix->has_x1 = h.hasX1;
  ix->tile_stride = h.tileStride;
  ix->bytes_per_row = h.tileStride / kTileRows;
  ix->ctx = ctx;
  ix->slots = ctx->slots;
  ix->aux_stream = ctx->aux_stream;
  ix->d_aux_flags = ctx->d_aux_flags;
  rc = ensure_aux_qbuf(ctx, qbuf_bytes_per_query_w(ix->w16));
  if (rc != BBQ_OK) return rc;
  Storage &st = ix->main;
  auto bail = [&](int code) {
    destroy_unlocked(ix.release());
    return code;
  };
  if (h.tilesBytes > 0 && hipMalloc((void **)&st.d_tiles, (size_t)h.tilesBytes) != hipSuccess)
    return bail(fail(BBQ_ERR_OOM, "bbq_index_load: %lld bytes of tiles", (long long)h.tilesBytes));
  if (h.exactBytes > 0 && hipMalloc((void **)&st.d_exact, (size_t)h.exactBytes) != hipSuccess)
    return bail(fail(BBQ_ERR_OOM, "bbq_index_load: %lld bytes of exact corrections", (long long)h.exactBytes));
  const size_t piece = 64u << 20;
  std::vector<uint8_t> buf(piece);
  uint8_t *dst[2] = {st.d_tiles, (uint8_t *)st.d_exact};
  const int64_t len[2] = {h.tilesBytes, h.exactBytes};
  uint64_t dsum = kFnvSeed;
  for (int part = 0; part < 2; ++part) {
    for (int64_t o = 0; o < len[part]; o += (int64_t)piece) {
      const size_t m = (size_t)std::min<int64_t>((int64_t)piece, len[part] - o);
      if (fread(buf.data(), 1, m, fc.f) != m) return bail(fail(BBQ_ERR_INVALID_ARG, "%s: read failed", dpath.c_str()));
      dsum = fnv64_words(buf.data(), m, dsum);
      if (hipMemcpy(dst[part] + o, buf.data(), m, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(BBQ_ERR_HIP, "bbq_index_load: copy to the device failed"));
    }
  }
  if (dsum != want_sum) return bail(fail(BBQ_ERR_INVALID_ARG, "%s: vector data checksum mismatch", dpath.c_str()));
  st.row_id_base = h.rowBase;
  st.view.n_rows = h.vectorCount;
  st.view.w16 = h.w16;
  st.view.tile_stride = h.tileStride;
  st.view.has_x1 = h.hasX1;
  st.view.dim = h.dimensions;
  st.view.layout = h.layout;
  st.view.tiles = st.d_tiles;
  st.view.exact = st.d_exact;
  if (centroid_out) memcpy(centroid_out, cen.data(), cen.size() * 4);
  *out = ix.release();
  return BBQ_OK;
}

int bbq_index_export(bbq_index *ix, uint8_t *codes, double *corr) {
  clear_error();
  if (!ix) return fail(BBQ_ERR_INVALID_ARG, "bbq_index_export: null handle");
  if (ix->n_rows == 0 || (!codes && !corr)) return BBQ_OK;
  std::lock_guard<std::mutex> lk(ix->ctx->mu);
  HIPCHK(hipSetDevice(ix->device));
  const int64_t n_tiles = (ix->n_rows + kTileRows - 1) / kTileRows;
  const int64_t group = std::max<int64_t>(1, (64ll << 20) / ix->tile_stride);  // tiles per piece
  std::vector<uint8_t> buf((size_t)(group * ix->tile_stride));
  std::vector<double> ex;
  if (corr && ix->layout == kLayoutCompact) ex.resize((size_t)(group * kTileRows * 4));
  const int pb = ix->pb, w16 = ix->w16;
  for (int64_t t0 = 0; t0 < n_tiles; t0 += group) {
    const int64_t nt = std::min(group, n_tiles - t0);
    HIPCHK(hipMemcpy(buf.data(), ix->main.d_tiles + t0 * ix->tile_stride, (size_t)(nt * ix->tile_stride), hipMemcpyDeviceToHost));
    if (!ex.empty())
      HIPCHK(hipMemcpy(ex.data(), ix->main.d_exact + t0 * kTileRows * 4, (size_t)(nt * kTileRows) * 32, hipMemcpyDeviceToHost));
    for (int64_t t = 0; t < nt; ++t) {
      const uint8_t *tp = buf.data() + t * ix->tile_stride;
      const uint8_t *cr = tp + (size_t)w16 * (kTileRows * 16);
      for (int r = 0; r < kTileRows; ++r) {
        const int64_t row = (t0 + t) * kTileRows + r;
        if (row >= ix->n_rows) break;
        int ones = 0;
        for (int b = 0; b < pb; ++b) {
          const uint8_t v = tp[((size_t)(b >> 4) * kTileRows + r) * 16 + (b & 15)];
          if (codes) codes[row * pb + b] = v;
          ones += __builtin_popcount(v);
        }
        if (!corr) continue;
        double *c = corr + row * 4;
        if (ix->layout == kLayoutCompact) {
          const double *e = ex.data() + ((size_t)t * kTileRows + r) * 4;
          c[0] = e[0]; c[1] = e[1]; c[2] = e[2];
          c[3] = (double)ones;  // compact layout is only chosen when every sum equals the popcount
        } else {
          memcpy(c, cr + r * 16, 16);
          memcpy(c + 2, cr + 1024 + r * 8, 8);
          if (ix->has_x1) memcpy(c + 3, cr + 1536 + r * 8, 8);
          else c[3] = (double)ones;
        }
      }
    }
  }
  return BBQ_OK;
}

}  // extern "C"
