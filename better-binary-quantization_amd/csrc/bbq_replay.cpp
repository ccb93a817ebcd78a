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
#include <algorithm>
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


static inline uint32_t key_of_score_bits(uint32_t b) { return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
static inline float score_of_key(uint32_t key) {
  const uint32_t b = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

uint32_t bbq_key_of_score(float score) {
  uint32_t b;
  memcpy(&b, &score, 4);
  return key_of_score_bits(b);
}

// The global answer from the shards' answer blocks (include/bbq.h, bbq_shard_scan_begin).  Every source s lists its own rows above
// its cut c_s, and c_s never exceeds the key G of the global (k2 + 1)-th largest score; with C = max c_s the union E of the listed rows
// above C is therefore EVERY row above C.  |E| >= k2 + 1: the k2 + 1 largest of E are the k2 + 1 largest of the index; |E| == k2: then
// G == C and E is exactly the rows above the boundary; fewer: the boundary value repeats inside the answer.  Whenever the k2 best
// scores and the boundary are pairwise different as floats the reference heap returns exactly those rows, descending (DESIGN.md
// "Exact top-k"); otherwise its history decides and the caller replays it over the lists (status 1).
int bbq_merge_answers(int32_t n_sources, const uint64_t *const *answers, const int64_t *strides, int32_t n_queries, int64_t n_total,
                      int64_t k, int32_t n_threads, int32_t *out_idx, float *out_score, int64_t *out_n, uint8_t *status) {
  bbq::clear_error();
  if (n_sources <= 0 || n_sources > 4096 || !answers || !strides || n_queries < 0 || (n_queries > 0 && (!out_n || !status)))
    return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_merge_answers: null argument");
  if (k < 0) return bbq::fail(BBQ_ERR_NEGATIVE_K, "k值不能为负数");
  if (k > 0 && n_queries > 0 && (!out_idx || !out_score)) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_merge_answers: null output");
  for (int32_t s = 0; s < n_sources; ++s)
    if (!answers[s] || strides[s] < 3) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_merge_answers: source %d has no block", s);
  const int64_t k2 = std::min<int64_t>(k, n_total);
  std::vector<int> bad((size_t)std::max(n_queries, 1), 0);
  auto work = [&](int32_t lo, int32_t hi) {
    std::vector<const uint64_t *> head((size_t)n_sources), end((size_t)n_sources);
    std::vector<uint64_t> sel;
    sel.reserve((size_t)k2 + 1);
    for (int32_t q = lo; q < hi; ++q) {
      uint32_t cut = 0, flags = 0, unproven = 0;
      for (int32_t s = 0; s < n_sources; ++s) {
        const uint64_t *b = answers[s] + (size_t)q * (size_t)strides[s];
        flags |= (uint32_t)(b[0] >> 32);
        unproven |= (uint32_t)(b[1] >> 32);
        const uint64_t m = (uint32_t)b[1];
        if ((int64_t)m + 3 > strides[s]) { bad[(size_t)q] = 1; unproven = 1; }
        cut = std::max(cut, (uint32_t)b[2]);
        head[(size_t)s] = b + 3;
        end[(size_t)s] = b + 3 + ((int64_t)m + 3 > strides[s] ? 0 : m);
      }
      if (flags) { status[q] = 2; continue; }
      if (unproven || k2 == 0) { status[q] = k2 == 0 ? 0 : 1; if (k2 == 0) out_n[q] = 0; continue; }
      // the k2 + 1 largest entries above the cut, descending: S-way merge of descending lists
      sel.clear();
      while ((int64_t)sel.size() < k2 + 1) {
        int best = -1;
        uint32_t best_key = 0;
        for (int32_t s = 0; s < n_sources; ++s) {
          if (head[(size_t)s] == end[(size_t)s]) continue;
          const uint32_t key = key_of_score_bits((uint32_t)*head[(size_t)s]);
          if (best < 0 || key > best_key) { best = s; best_key = key; }
        }
        if (best < 0 || best_key <= cut) break;
        sel.push_back(*head[(size_t)best]++);
      }
      const int64_t have = (int64_t)sel.size();
      bool ok = have >= k2;
      auto fscore = [](uint64_t e) { const uint32_t b = (uint32_t)e; float f; memcpy(&f, &b, 4); return f; };
      for (int64_t j = 0; ok && j + 1 < have; ++j) ok = fscore(sel[(size_t)j]) != fscore(sel[(size_t)j + 1]);  // sorted: equal scores are neighbours (+0 / -0 too)
      if (ok && have == k2 && cut != 0u) ok = fscore(sel[(size_t)k2 - 1]) != score_of_key(cut);       // the boundary is the cut itself
      if (!ok) { status[q] = 1; continue; }
      for (int64_t j = 0; j < k2; ++j) {
        out_idx[(int64_t)q * k + j] = (int32_t)(uint32_t)(sel[(size_t)j] >> 32);
        out_score[(int64_t)q * k + j] = fscore(sel[(size_t)j]);
      }
      out_n[q] = k2;
      status[q] = 0;
    }
  };
  int T = n_threads > 0 ? n_threads : 1;
  if (T > n_queries / 64) T = std::max(1, n_queries / 64);  // a query merges in a few microseconds: threads only pay for large batches
  if (T <= 1) {
    work(0, n_queries);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, (int32_t)((int64_t)n_queries * t / T), (int32_t)((int64_t)n_queries * (t + 1) / T));
    for (auto &x : th) x.join();
  }
  for (int32_t q = 0; q < n_queries; ++q)
    if (bad[(size_t)q]) return bbq::fail(BBQ_ERR_INVALID_ARG, "bbq_merge_answers: query %d: an answer block claims more entries than its stride holds", q);
  return BBQ_OK;
}

}  // extern "C"
