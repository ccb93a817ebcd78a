// bbq_internal.h - host-side internals of libbbq shared between translation units
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/bbq.h"

namespace bbq {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// Exact replay of the reference's top-k loop (src/binaryQuantizationFormat.ts:383-411, src/minHeap.ts:9-130).
// Feed rows in ascending global row order; rows that are skipped must be rows that cannot change the
// reference heap (DESIGN.md "Exact top-k").
class HeapReplay {
 public:
  HeapReplay(int64_t k, int64_t n_total);
  inline void offer(float score, int32_t row) {
    const double s = (double)score;
    if ((int64_t)heap_.size() < k2_) {
      push(s, row);
    } else if (k2_ > 0 && s > heap_[0].score) {
      pop();
      push(s, row);
    }
  }
  // pops everything (ascending) and writes it reversed = descending, like topKResults.reverse()
  int64_t finish(int32_t *out_idx, float *out_score);
  // the same heap keyed by an f64 (the rerank selector's trueScore, src/topKSelector.ts:40-66); drain pops ascending
  inline void offer64(double s, int32_t tag) {
    if ((int64_t)heap_.size() < k2_) {
      push(s, tag);
    } else if (k2_ > 0 && s > heap_[0].score) {
      pop();
      push(s, tag);
    }
  }
  int64_t drain_ascending(int32_t *out_tag, double *out_score);
  int64_t mutations() const { return mutations_; }

 private:
  struct Item { double score; int32_t index; };
  void push(double s, int32_t row);
  Item pop();
  std::vector<Item> heap_;
  int64_t k2_;
  int64_t mutations_ = 0;
};

}  // namespace bbq
