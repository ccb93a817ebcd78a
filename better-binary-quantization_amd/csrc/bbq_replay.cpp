// bbq_replay.cpp - host-side exact merge: replays the reference's MinHeap over the device's candidates.
//
// Why a replay and not "sort by score": the reference keeps the top-k in a binary min-heap
// (src/minHeap.ts:9-130) that compares f32-rounded scores with a strict `>` on insertion
// (src/binaryQuantizationFormat.ts:394) and a `>= 0` stop in bubble-up / strict `<` in bubble-down, so which
// of several equal-scored rows survives, and the order equal scores are returned in, depend on the heap's
// whole mutation history.  A row can change the heap only if its score exceeds the heap minimum at that
// moment = the k-th largest score among the rows before it; the device emits a superset of those rows in
// row order (thresholds are lower bounds of that k-th largest), and replaying the identical heap over that
// superset reproduces the identical mutation sequence.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <vector>
#include "bbq_internal.h"

namespace bbq {

static thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
void clear_error() { g_err.clear(); }
const char *last_error_cstr() { return g_err.c_str(); }

HeapReplay::HeapReplay(int64_t k, int64_t n_total) : k2_(k < n_total ? k : n_total) {
  if (k2_ < 0) k2_ = 0;
  heap_.reserve((size_t)k2_ + 1);
}

// MinHeap.push + _bubbleUp (src/minHeap.ts:45-48,70-81); compareFn = a.score - b.score
void HeapReplay::push(double s, int32_t row) {
  heap_.push_back(Item{s, row});
  size_t index = heap_.size() - 1;
  while (index > 0) {
    const size_t parent = (index - 1) / 2;
    const double cmp = heap_[index].score - heap_[parent].score;
    if (cmp >= 0) break;  // a NaN difference is not >= 0: the reference swaps, so do we
    const Item t = heap_[index];
    heap_[index] = heap_[parent];
    heap_[parent] = t;
    index = parent;
  }
  ++mutations_;
}

// MinHeap.pop + _bubbleDown (src/minHeap.ts:53-65,86-116)
HeapReplay::Item HeapReplay::pop() {
  const Item mn = heap_[0];
  const Item last = heap_.back();
  heap_.pop_back();
  const size_t len = heap_.size();
  if (len > 0) {
    heap_[0] = last;
    size_t index = 0;
    for (;;) {
      size_t smallest = index;
      const size_t l = 2 * index + 1, r = 2 * index + 2;
      if (l < len && (heap_[l].score - heap_[smallest].score) < 0) smallest = l;
      if (r < len && (heap_[r].score - heap_[smallest].score) < 0) smallest = r;
      if (smallest == index) break;
      const Item t = heap_[index];
      heap_[index] = heap_[smallest];
      heap_[smallest] = t;
      index = smallest;
    }
  }
  return mn;
}

int64_t HeapReplay::finish(int32_t *out_idx, float *out_score) {
  const int64_t n = (int64_t)heap_.size();
  for (int64_t j = n - 1; j >= 0; --j) {
    const Item it = pop();
    out_idx[j] = it.index;
    out_score[j] = (float)it.score;
  }
  return n;
}

int64_t HeapReplay::drain_ascending(int32_t *out_tag, double *out_score) {
  const int64_t n = (int64_t)heap_.size();
  for (int64_t j = 0; j < n; ++j) {
    const Item it = pop();
    out_tag[j] = it.index;
    out_score[j] = it.score;
  }
  return n;
}

}  // namespace bbq

extern "C" {

const char *bbq_last_error(void) { return bbq::last_error_cstr(); }
int bbq_abi_version(void) { return BBQ_ABI_VERSION; }

int bbq_replay(int32_t n_lists, const bbq_cand *const *lists, const int64_t *counts, int64_t n_total, int64_t k,
               int32_t *out_idx, float *out_score, int64_t *out_n) {
  bbq::clear_error();
  if (n_lists < 0 || (n_lists > 0 && (!lists || !counts)) || !out_n) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_replay: null argument");
  if (k < 0) return bbq::fail(BBQ_ERR_NEGATIVE_K, "k值不能为负数");
  if (k > 0 && (!out_idx || !out_score)) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_replay: null output");
  *out_n = 0;
  if (k == 0) return BBQ_OK;
  bbq::HeapReplay h(k, n_total);
  uint32_t prev = 0;
  bool first = true;
  for (int32_t i = 0; i < n_lists; ++i) {
    const bbq_cand *l = lists[i];
    for (int64_t j = 0; j < counts[i]; ++j) {
      const uint32_t row = (uint32_t)(l[j] >> 32);
      if (!first && row <= prev) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_replay: candidates not in ascending row order");
      prev = row;
      first = false;
      const uint32_t bits = (uint32_t)l[j];
      float s;
      memcpy(&s, &bits, 4);
      h.offer(s, (int32_t)row);
    }
  }
  *out_n = h.finish(out_idx, out_score);
  return BBQ_OK;
}


int bbq_replay_batch(int32_t n_sources, const bbq_cand *const *packed, const int64_t *const *offsets, int32_t n_queries,
                     int64_t n_total, int64_t k, int32_t n_threads, int32_t *out_idx, float *out_score, int64_t *out_n) {
  bbq::clear_error();
  if (n_sources < 0 || n_queries < 0 || (n_sources > 0 && (!packed || !offsets)) || (n_queries > 0 && !out_n))
    return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_replay_batch: null argument");
  if (k < 0) return bbq::fail(BBQ_ERR_NEGATIVE_K, "k值不能为负数");
  for (int32_t q = 0; q < n_queries; ++q) out_n[q] = 0;
  if (k == 0 || n_queries == 0) return BBQ_OK;
  if (!out_idx || !out_score) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_replay_batch: null output");
  std::vector<int> bad((size_t)n_queries, 0);
  auto work = [&](int32_t lo, int32_t hi) {
    for (int32_t q = lo; q < hi; ++q) {
      bbq::HeapReplay h(k, n_total);
      uint32_t prev = 0;
      bool first = true;
      for (int32_t s = 0; s < n_sources && !bad[(size_t)q]; ++s) {
        const bbq_cand *l = packed[s] + offsets[s][q];
        const int64_t cnt = offsets[s][q + 1] - offsets[s][q];
        for (int64_t j = 0; j < cnt; ++j) {
          const uint32_t row = (uint32_t)(l[j] >> 32);
          if (!first && row <= prev) { bad[(size_t)q] = 1; break; }
          prev = row;
          first = false;
          const uint32_t bits = (uint32_t)l[j];
          float sc;
          memcpy(&sc, &bits, 4);
          h.offer(sc, (int32_t)row);
        }
      }
      out_n[q] = h.finish(out_idx + (int64_t)q * k, out_score + (int64_t)q * k);
    }
  };
  int T = n_threads > 0 ? n_threads : 1;
  if (T > n_queries) T = n_queries;
  if (T <= 1) {
    work(0, n_queries);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, (int32_t)((int64_t)n_queries * t / T), (int32_t)((int64_t)n_queries * (t + 1) / T));
    for (auto &x : th) x.join();
  }
  for (int32_t q = 0; q < n_queries; ++q)
    if (bad[(size_t)q]) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_replay_batch: candidates of query %d not in ascending row order", q);
  return BBQ_OK;
}

}  // extern "C"
