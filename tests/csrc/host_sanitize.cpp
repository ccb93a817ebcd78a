// Exercises the host-only part of libbbq (quantizer, heap replay, error plumbing: bbq_quantizer.cpp + bbq_replay.cpp, no HIP) through
// the C ABI; built by tests/test_sanitizers_cpu.py with -fsanitize=address,undefined and with -fsanitize=thread.  GPU sanitizers are
// not available on this pool, so this is where memory and thread errors of the host code are looked for.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/bbq.h"

static uint32_t rng_state = 12345;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "check failed at line %d: %s (%s)\n", __LINE__, #cond, bbq_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  const int dims[] = {1, 7, 64, 100, 129};
  const int bits[] = {1, 2, 4, 8};
  for (int dim : dims)
    for (int ib : bits)
      for (int sim = 0; sim < 3; ++sim) {
        const int64_t n = 257;
        std::vector<float> v((size_t)n * dim);
        for (float &x : v) x = frand();
        const int64_t rb = ib == 1 ? (dim + 7) / 8 : dim;
        std::vector<uint8_t> codes((size_t)(n * rb));
        std::vector<double> corr((size_t)n * 4);
        std::vector<float> cen((size_t)dim);
        int64_t bad_row = -1;
        int32_t bad_col = -1;
        CHECK(bbq_quantize_vectors(v.data(), n, dim, sim, ib, 0.1, 5, 4, codes.data(), corr.data(), cen.data(), &bad_row, &bad_col) == BBQ_OK);
        if (ib > 1)
          for (uint8_t c : codes) CHECK(c < (1u << ib));
        // queries: one by one and as a batch on threads, identical
        const int nq = 9;
        std::vector<float> q((size_t)nq * dim);
        for (float &x : q) x = frand();
        std::vector<uint8_t> qq((size_t)nq * dim), qq1((size_t)dim);
        std::vector<double> qc((size_t)nq * 4), qc1(4);
        int32_t bad_q = -1;
        CHECK(bbq_quantize_queries(q.data(), nq, dim, cen.data(), sim, 4, 0.1, 5, 3, qq.data(), qc.data(), &bad_q) == BBQ_OK);
        for (int i = 0; i < nq; ++i) {
          CHECK(bbq_quantize_query(q.data() + (size_t)i * dim, dim, cen.data(), sim, 4, 0.1, 5, qq1.data(), qc1.data()) == BBQ_OK);
          CHECK(memcmp(qq1.data(), qq.data() + (size_t)i * dim, (size_t)dim) == 0);
          CHECK(memcmp(qc1.data(), qc.data() + (size_t)i * 4, 32) == 0);
        }
        (void)bbq_centroid_dp(cen.data(), dim);
      }
  {  // the reference's input errors
    std::vector<float> v = {1.f, NAN, 3.f, 4.f};
    std::vector<uint8_t> codes(8);
    std::vector<double> corr(8);
    std::vector<float> cen(2);
    int64_t bad_row = -1;
    int32_t bad_col = -1;
    CHECK(bbq_quantize_vectors(v.data(), 2, 2, 0, 1, 0.1, 5, 2, codes.data(), corr.data(), cen.data(), &bad_row, &bad_col) == BBQ_ERR_NAN_INPUT);
    CHECK(bad_row == 0 && bad_col == 1);
    CHECK(bbq_quantize_vectors(v.data(), 0, 2, 0, 1, 0.1, 5, 2, codes.data(), corr.data(), cen.data(), nullptr, nullptr) == BBQ_ERR_EMPTY);
    CHECK(strlen(bbq_last_error()) > 0);
  }
  {  // heap replay: several sources, many queries, host threads; ties; k = 0, k > rows
    const int S = 3, Q = 50;
    const int64_t per = 400, n_total = S * per;
    std::vector<std::vector<bbq_cand>> packed((size_t)S);
    std::vector<std::vector<int64_t>> offsets((size_t)S);
    for (int s = 0; s < S; ++s) {
      offsets[(size_t)s].push_back(0);
      for (int qi = 0; qi < Q; ++qi) {
        for (int64_t r = 0; r < per; ++r) {
          if (((r * 7 + qi) % 3) == 0) continue;  // ragged lists
          const float sc = (float)((r * 31 + qi * 17 + s) % 23) * 0.125f;  // few distinct scores: ties everywhere
          uint32_t b;
          memcpy(&b, &sc, 4);
          packed[(size_t)s].push_back(((uint64_t)(uint32_t)(s * per + r) << 32) | b);
        }
        offsets[(size_t)s].push_back((int64_t)packed[(size_t)s].size());
      }
    }
    const bbq_cand *pp[S];
    const int64_t *po[S];
    for (int s = 0; s < S; ++s) { pp[s] = packed[(size_t)s].data(); po[s] = offsets[(size_t)s].data(); }
    for (int64_t k : {(int64_t)0, (int64_t)1, (int64_t)37, (int64_t)5000}) {
      std::vector<int32_t> idx((size_t)(Q * (k > 0 ? k : 1))), idx1((size_t)(k > 0 ? k : 1));
      std::vector<float> sc((size_t)(Q * (k > 0 ? k : 1))), sc1((size_t)(k > 0 ? k : 1));
      std::vector<int64_t> cnt((size_t)Q);
      CHECK(bbq_replay_batch(S, pp, po, Q, n_total, k, 4, idx.data(), sc.data(), cnt.data()) == BBQ_OK);
      for (int qi = 0; qi < Q; qi += 7) {  // the single-query entry point agrees
        const bbq_cand *l[S];
        int64_t c[S];
        for (int s = 0; s < S; ++s) { l[s] = pp[s] + po[s][qi]; c[s] = po[s][qi + 1] - po[s][qi]; }
        int64_t n1 = -1;
        CHECK(bbq_replay(S, l, c, n_total, k, idx1.data(), sc1.data(), &n1) == BBQ_OK);
        CHECK(n1 == cnt[(size_t)qi]);
        CHECK(k == 0 || memcmp(idx1.data(), idx.data() + (size_t)qi * k, (size_t)n1 * 4) == 0);
      }
    }
    int64_t n1 = 0;
    CHECK(bbq_replay(1, pp, nullptr, n_total, 5, nullptr, nullptr, &n1) == BBQ_ERR_INVALID_ARG);
    CHECK(bbq_replay_batch(S, pp, po, Q, n_total, -1, 2, nullptr, nullptr, nullptr) != BBQ_OK);
  }
  printf("host sanitize ok\n");
  return 0;
}
