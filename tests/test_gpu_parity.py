"""GPU parity: the HIP path, called through the C ABI, against the pinned oracle and the golden vectors.
Bit-exact: integer qcDist, f64 score, f32 score, top-k indices AND their order (ties included)."""
import os

import numpy as np
import pytest

import orclib as O
from bbqlib import bbq_amd as B

pytestmark = pytest.mark.gpu

CASES = [n for n in O.golden_names() if not n.startswith(("intdot_", "api_", "rerank_"))]
FULL = [n for n in CASES if O.load_golden(n)["full"]]
HASHED = [n for n in CASES if not O.load_golden(n)["full"]]


def canon64(a):
    a = np.array(a, np.float64)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def canon32(a):
    a = np.array(a, np.float32)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint32)


def _index_from_case(g):
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    # the product's own host quantizer builds the index; it must reproduce the reference's bytes
    codes, corr, cen = B.quantize_vectors(base, sim, g["ib"], g["lambda"], g["iters"])
    assert O.sha(codes) == g["codes_sha256"]
    if not np.isnan(corr).any():
        assert O.sha(corr) == g["corr_sha256"]
    assert O.sha(cen) == O.sha(O.dec(g["centroid_f32"], "<f4"))
    cdp = B.centroid_dp(cen)
    assert np.float64(cdp).view(np.uint64) == O.dec(g["centroid_dp_f64"], "<f8").view(np.uint64)[0]
    return sim, base, queries, codes, corr, cen, cdp


def _make_index(codes, corr, dim, cdp, compact=True, **kw):
    """compact=False streams the exact f64 corrections (inline layout); True (default) the 8-byte compact ones"""
    return B.Index(codes, corr, dim, cdp, corrections="compact" if compact else "inline", **kw)


def _stored_row_bytes(g):
    """bytes of one row in HBM, padded to 16: packed bits, or 2 / 4 / 8-bit fields for indexBits 2 / 3-4 / 5-8 (dim 1 excepted)"""
    ib, dim = g["ib"], g["dim"]
    sb = 1 if (ib == 1 or dim == 1) else 2 if ib == 2 else 4 if ib <= 4 else 8
    return ((dim * sb + 7) // 8 + 15) // 16 * 16


def _check(name, options=None, compact=True):
    g = O.load_golden(name)
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    ib = g["ib"]
    ix = _make_index(codes, corr, g["dim"], cdp, compact, index_bits=ib)
    implicit_sum = ib == 1 or (g["dim"] > 1 and (corr[:, 3] == codes.sum(axis=1)).all())   # else the sums are stored: 8 B more, inline
    assert ix.bytes_per_row == _stored_row_bytes(g) + ((4 if compact else 24) if implicit_sum else 32)
    for k_, v_ in (options or {}).items():
        ix.set_option(k_, v_)
    try:
        for qi, rec in enumerate(g["queries"]):
            qq, qc = B.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
            np.testing.assert_array_equal(qq, O.dec(rec["qquant_u8"], "u1"))
            np.testing.assert_array_equal(canon64(qc), canon64(O.dec(rec["qcorr_f64"], "<f8")))
            d, s64, s32 = ix.score_rows(qq, qc, g["qb"], sim)
            if "per_row_error" in rec:
                # the reference throws for this queryBits on a multi-bit index; the integer dot product is still defined
                # (src/bitwiseDotProduct.ts:14-30) and the API mirror throws the reference's message
                np.testing.assert_array_equal(d, [O.lib().orc_dot_u8(O.u8p(qq), O.u8p(codes[i]), g["dim"]) for i in range(g["n"])])
                f = B.createBinaryQuantizationFormat({"queryBits": g["qb"], "indexBits": ib, "quantizer": {
                    "similarityFunction": g["sim"], "lambda": g["lambda"], "iters": g["iters"]}})
                with pytest.raises(Exception) as e:
                    f.searchNearestNeighbors(queries[qi], f.quantizeVectors(list(base))["quantizedVectors"], 5)
                assert str(e.value) == rec["per_row_error"]
                continue
            assert O.sha(d) == rec["qcdist_sha256"], "integer qcDist"
            if g["full"]:
                np.testing.assert_array_equal(d, O.dec(rec["qcdist_i32"], "<i4"))
                np.testing.assert_array_equal(canon64(s64), canon64(O.dec(rec["score_f64"], "<f8")))
            if not np.isnan(s64).any():
                assert O.sha(s64) == rec["score_sha256"], "f64 score"
                assert O.sha(s32) == rec["score_f32_sha256"], "f32 score"
            for tk in rec["topk"]:
                idx, sc = ix.search(qq, qc, g["qb"], sim, tk["k"])
                np.testing.assert_array_equal(idx, O.dec(tk["idx_i32"], "<i4"), err_msg="%s q%d k=%d" % (name, qi, tk["k"]))
                np.testing.assert_array_equal(canon32(sc), canon32(O.dec(tk["score_f32"], "<f4")))
    finally:
        ix.close()


@pytest.mark.parametrize("name", FULL)
def test_golden_full(name):
    _check(name)


@pytest.mark.parametrize("name", FULL + HASHED)
def test_golden_inline_exact_corrections_layout(name):
    _check(name, compact=False)


@pytest.mark.parametrize("name", HASHED)
def test_golden_ties_and_big(name):
    _check(name)


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("name", ["ties_cos_qb4", "ties_16d_qb1", "big_20000x128_cos", "ties_max_qb4", "ties_euc_qb4", "ties_cos_qb1",
                                  "big_20000x1024_cos", "big_20000x1024_euc_qb8", "ib2_ties_cos_qb4", "ib2_big_20000x128_euc"])
def test_golden_many_small_segments(name, compact):
    """force the multi-segment sparse path on small indexes: 1024-row first segment, x2 growth"""
    _check(name, {"first_segment_rows": 1024, "segment_growth": 2}, compact=compact)


@pytest.mark.parametrize("name", ["ties_euc_qb4", "c1_1000x128_cos_qb4", "big_20000x128_cos"])
def test_golden_dense_path(name):
    _check(name, {"force_dense": 1})


def _oracle_topk(codes, corr, dim, qq, qc, qb, sim, cdp, k):
    _, _, s32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp)
    return O.heap_topk(s32, k)


def test_batch_pipeline_matches_oracle():
    """more queries than one sub-batch, several pipeline slots; every query equals the oracle's replayed heap"""
    rng = np.random.default_rng(5)
    n, dim, nq, k = 30000, 256, 41, 100
    base = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    sim = 1
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    qs = [B.quantize_query(q, cen, sim, 4) for q in queries]
    qq = np.stack([a for a, _ in qs])
    qc = np.stack([b for _, b in qs])
    ix = B.Index(codes, corr, dim, cdp)
    try:
        for opts in ({"batch_queries": 8, "pipeline_slots": 3, "first_segment_rows": 2048, "segment_growth": 4},
                     {"batch_queries": 16, "pipeline_slots": 2, "replay_threads": 4}):
            for o, v in opts.items():
                ix.set_option(o, v)
            idx, sc, cnt = ix.search_batch(qq, qc, 4, sim, k)
            assert (cnt == k).all()
            for i in range(nq):
                oi, os_ = _oracle_topk(codes, corr, dim, qq[i], qc[i], 4, sim, cdp, k)
                np.testing.assert_array_equal(idx[i], oi)
                np.testing.assert_array_equal(sc[i].view(np.uint32), os_.view(np.uint32))
        st = ix.stats()
        assert st["dense_fallbacks"] == 0 and st["candidates"] > 0
    finally:
        ix.close()


@pytest.mark.parametrize("k", [1, 2048, 2049, 3000, 4096, 4097, 25000, 70000])
def test_k_edge_cases(k):
    """large k on the sparse path (up to 4096), k beyond it (dense replay), k > N (clamped like src/binaryQuantizationFormat.ts:385)"""
    rng = np.random.default_rng(6)
    n, dim = 50000, 64
    base = rng.standard_normal((n, dim)).astype(np.float32)
    sim = 0
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    qq, qc = B.quantize_query(rng.standard_normal(dim).astype(np.float32), cen, sim, 4)
    ix = B.Index(codes, corr, dim, cdp)
    try:
        idx, sc = ix.search(qq, qc, 4, sim, k)
        oi, os_ = _oracle_topk(codes, corr, dim, qq, qc, 4, sim, cdp, k)
        assert len(idx) == min(k, n)
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(sc.view(np.uint32), os_.view(np.uint32))
        assert ix.stats()["dense_fallbacks"] == (1 if k > 4096 else 0)
        if 1 < k <= 4096:   # the shared sweeps take the same k
            for share in (8, 32):
                ix.set_option("sweep_share", share)
                idx2, sc2 = ix.search(qq, qc, 4, sim, k)
                np.testing.assert_array_equal(idx2, oi)
                np.testing.assert_array_equal(sc2.view(np.uint32), os_.view(np.uint32))
    finally:
        ix.close()


def test_adversarial_increasing_scores_falls_back_and_stays_exact():
    """rows sorted by ascending score: every row enters the reference heap.  The flood tier carries all of them to the
    host replay; without it the candidate slots overflow and the dense path takes over - exact either way"""
    rng = np.random.default_rng(7)
    n, dim, k = 40000, 64, 50
    base = rng.standard_normal((n, dim)).astype(np.float32)
    sim = 1
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    qq, qc = B.quantize_query(rng.standard_normal(dim).astype(np.float32), cen, sim, 4)
    _, _, s32 = O.score_all(codes, corr, dim, qq, qc, 4, sim, cdp)
    order = np.argsort(s32, kind="stable")
    codes, corr = codes[order].copy(), corr[order].copy()
    ix = B.Index(codes, corr, dim, cdp)
    try:
        oi, os_ = _oracle_topk(codes, corr, dim, qq, qc, 4, sim, cdp, k)
        ix.set_option("latency_queries", 0)     # the chunk-slot path (calls with few queries append to the list instead: below)
        for flood_rows, dense in ((262144, 0), (0, 1)):
            ix.set_option("flood_rows", flood_rows)
            idx, sc = ix.search(qq, qc, 4, sim, k)
            np.testing.assert_array_equal(idx, oi)
            np.testing.assert_array_equal(sc.view(np.uint32), os_.view(np.uint32))
            assert ix.stats()["dense_fallbacks"] == dense
        ix.set_option("latency_queries", 4)     # append mode: every row is a candidate, the list takes them or the query goes dense
        for flood_rows in (262144, 0):
            ix.set_option("flood_rows", flood_rows)
            idx, sc = ix.search(qq, qc, 4, sim, k)
            np.testing.assert_array_equal(idx, oi)
            np.testing.assert_array_equal(sc.view(np.uint32), os_.view(np.uint32))
    finally:
        ix.close()


def test_explicit_component_sum_layout():
    """quantizedComponentSum that is NOT the row popcount (hand-edited corrections) must be honoured (4-double layout)"""
    g = O.load_golden("m_64d_cos_qb4")
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    corr = corr.copy()
    corr[::3, 3] += 2.0
    ix = B.Index(codes, corr, g["dim"], cdp)
    try:
        assert ix.bytes_per_row == 16 + 32
        qq, qc = B.quantize_query(queries[0], cen, sim, 4)
        d, s64, s32 = ix.score_rows(qq, qc, 4, sim)
        od, os64, os32 = O.score_all(codes, corr, g["dim"], qq, qc, 4, sim, cdp)
        np.testing.assert_array_equal(d, od)
        np.testing.assert_array_equal(s64.view(np.uint64), os64.view(np.uint64))
        idx, sc = ix.search(qq, qc, 4, sim, 10)
        oi, _ = O.heap_topk(os32, 10)
        np.testing.assert_array_equal(idx, oi)
    finally:
        ix.close()
    # the same through many small segments and through the shared-sweep kernels (explicit sums must reach their pre-filters)
    rng = np.random.default_rng(21)
    pick = rng.integers(0, codes.shape[0], 20000)
    codes2, corr2 = codes[pick].copy(), corr[pick].copy()
    od, os64, os32 = O.score_all(codes2, corr2, g["dim"], qq, qc, 4, sim, cdp)
    oi, osc = O.heap_topk(os32, 25)
    ix = B.Index(codes2, corr2, g["dim"], cdp)
    try:
        assert ix.bytes_per_row == 16 + 32
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        for share in (1, 8, 32):
            ix.set_option("sweep_share", share)
            idx, sc = ix.search(qq, qc, 4, sim, 25)
            np.testing.assert_array_equal(idx, oi, err_msg="share %d" % share)
            np.testing.assert_array_equal(sc.view(np.uint32), osc.view(np.uint32))
    finally:
        ix.close()


def test_empty_and_argument_errors():
    g = O.load_golden("edge_n1")
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    ix = B.Index(codes, corr, g["dim"], cdp)
    try:
        qq, qc = B.quantize_query(queries[0], cen, sim, 4)
        idx, sc = ix.search(qq, qc, 4, sim, 0)
        assert len(idx) == 0
        with pytest.raises(B.BBQError) as e:
            ix.search(qq, qc, 4, sim, -1)
        assert e.value.code == 7 and "k值不能为负数" in str(e.value)
        with pytest.raises(B.BBQError):
            ix.search(qq[:4], qc, 4, sim, 1)
        with pytest.raises(B.BBQError):
            ix.score_rows(qq, qc, 4, sim, 0, 5)
    finally:
        ix.close()
    with pytest.raises(B.BBQError) as e:     # a code that does not fit its 2-bit field
        B.Index(np.full((4, 8), 4, np.uint8), np.zeros((4, 4)), 8, 0.0, index_bits=2)
    assert e.value.code == 1
    with pytest.raises(B.BBQError) as e:
        B.Index(np.zeros((4, 8), np.uint8), np.zeros((4, 4)), 8, 0.0, index_bits=9)
    assert e.value.code == 1 and "indexBits必须在1-8之间" in str(e.value)
    empty = B.Index(np.zeros((0, 1), np.uint8), np.zeros((0, 4)), 8, 0.0)
    idx, sc = empty.search(np.zeros(8, np.uint8), np.zeros(4), 4, 0, 5)
    assert len(idx) == 0
    empty.close()


def test_python_api_mirror_quicksearch_c1():
    """BASELINE config 1 through the mirrored public API (src/index.ts:95-111): SURVEY App. C known answer"""
    g = O.load_golden("c1_1000x128_cos_qb4")
    base, queries = O.golden_inputs(g)
    res = B.quickSearch(queries[0], list(base), 10)
    assert [r["index"] for r in res] == [438, 839, 190, 656, 637, 545, 630, 174, 862, 42]
    assert res[0]["score"] == 0.6655263304710388


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("qb", [1, 4, 8])
def test_bound_filter_never_drops_a_candidate(sim, qb):
    """compact corrections: the score upper bound must never reject a row the exact score would admit.
    Hostile magnitudes (huge / tiny / zero / negative corrections), many sparse segments, both layouts must agree with
    the oracle bit for bit."""
    rng = np.random.default_rng(100 + 10 * sim + qb)
    n, dim, k = 40000, 96, 64
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    corr = np.zeros((n, 4))
    scale = 10.0 ** rng.uniform(-6, 3, n)                        # six decades of magnitude, row by row
    corr[:, 0] = -scale * rng.uniform(0.1, 1.0, n)
    corr[:, 1] = scale * rng.uniform(0.1, 1.0, n)
    corr[:, 2] = rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 2, n)
    if sim == 0:
        corr[:, 2] = np.abs(corr[:, 2])                          # EUCLIDEAN additional correction is a norm
    corr[::97, 0] = 0.0
    corr[::89, 1] = 0.0
    corr[::83, 2] = 0.0
    corr[:, 3] = pop
    qq = rng.integers(0, 1 << qb, dim).astype(np.uint8)
    qc = np.array([-0.7, 0.9, 0.01 if sim else 1.3, float(qq.sum())])
    cdp = 0.02
    od, os64, os32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp)
    assert not np.isnan(os32).any()
    for compact in (True, False):
        ix = _make_index(codes, corr, dim, cdp, compact)
        try:
            ix.set_option("first_segment_rows", 1024)
            ix.set_option("segment_growth", 2)
            for kk in (1, k, 700):
                idx, sc = ix.search(qq, qc, qb, sim, kk)
                oi, osc = O.heap_topk(os32, kk)
                np.testing.assert_array_equal(idx, oi)
                np.testing.assert_array_equal(sc.view(np.uint32), osc.view(np.uint32))
            assert ix.stats()["dense_fallbacks"] == 0
            d, s64, s32 = ix.score_rows(qq, qc, qb, sim)
            np.testing.assert_array_equal(s64.view(np.uint64), os64.view(np.uint64))
        finally:
            ix.close()


def test_bound_filter_nonfinite_corrections_take_the_exact_path():
    """inf / NaN / overflowing corrections cannot be bounded: such rows must reach the exact path (and a NaN score
    must flag the dense fallback), never be silently dropped"""
    g = O.load_golden("m_64d_cos_qb4")
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    n = 30000
    rng = np.random.default_rng(9)
    pick = rng.integers(0, codes.shape[0], n)
    codes, corr = codes[pick].copy(), corr[pick].copy()
    corr[5000, 1] = 1e300          # overflows f32 -> no finite bound
    corr[12345, 2] = 3.0e38        # f32-representable, huge
    corr[20000, 0] = -1e-320       # f64 subnormal
    qq, qc = B.quantize_query(queries[0], cen, sim, 4)
    od, os64, os32 = O.score_all(codes, corr, g["dim"], qq, qc, 4, sim, cdp)
    ix = _make_index(codes, corr, g["dim"], cdp, True)
    try:
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        idx, sc = ix.search(qq, qc, 4, sim, 20)
        oi, osc = O.heap_topk(os32, 20)
        np.testing.assert_array_equal(idx, oi)
        np.testing.assert_array_equal(canon32(sc), canon32(osc))
        assert 5000 in idx.tolist() or 12345 in idx.tolist()
    finally:
        ix.close()
    corr[25000, 2] = np.nan         # NaN score deep inside a sparse segment
    od, os64, os32 = O.score_all(codes, corr, g["dim"], qq, qc, 4, sim, cdp)
    ix = _make_index(codes, corr, g["dim"], cdp, True)
    try:
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        idx, sc = ix.search(qq, qc, 4, sim, 20)
        oi, osc = O.heap_topk(os32, 20)
        np.testing.assert_array_equal(idx, oi)
        assert ix.stats()["dense_fallbacks"] == 1
    finally:
        ix.close()


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("qb,dim", [(1, 128), (2, 768), (3, 128), (4, 768), (4, 1024), (7, 128), (4, 1536)])
def test_matrix_core_sweep_hostile_rows_and_queries(sim, qb, dim):
    """the shared sweep on the matrix cores (sweep_share 32: the threshold on the integer dot product as the MFMA's start value, FP6 x FP4
    operands for query values <= 15, int8 above) on rows its threshold cannot bound - zero and NEGATIVE interval widths, widths lost in
    f32, huge / tiny / non-finite corrections - and on queries it must refuse (zero width: the sub-batch sweeps on the vector ALUs):
    every answer is the oracle's, for both corrections layouts, 70 queries per call (three groups, the last one partial)."""
    rng = np.random.default_rng(1000 * sim + 10 * qb + dim)
    n, k, nq = 60000, 50, 70
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    corr = np.zeros((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = np.abs(rng.standard_normal(n)) * 1e-2 if sim == 0 else 1e-3 * rng.standard_normal(n)
    hostile = rng.choice(n, 6000, replace=False)
    scale = 10.0 ** rng.uniform(-6, 3, len(hostile))
    corr[hostile, 0] = -scale * rng.uniform(0.1, 1.0, len(hostile))
    corr[hostile, 1] = scale * rng.uniform(0.1, 1.0, len(hostile))
    corr[hostile[:500], 1] = corr[hostile[:500], 0]                                   # zero width
    corr[hostile[500:1000], 1] = corr[hostile[500:1000], 0] - 0.01                    # negative width
    corr[hostile[1000:1500], 1] = corr[hostile[1000:1500], 0] * (1 - 2.0 ** -30)      # a width lost in f32
    corr[hostile[1500:1600], 1] = 1e300                                               # overflows f32
    corr[hostile[1600:1700], 2] = 3.0e38
    corr[hostile[1700:1800], 0] = -1e-320                                             # f64 subnormal
    corr[hostile[1800:2300], 2] = rng.standard_normal(500) * 1e4
    corr[:, 3] = pop
    cdp = 0.02
    qq = rng.integers(0, 1 << qb, size=(nq, dim), dtype=np.uint8)
    qc = np.empty((nq, 4))
    qc[:, 0] = -0.15 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 1] = 0.148 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 2] = 0.7 * rng.random(nq) if sim == 0 else -0.0028 * rng.random(nq)
    qc[:, 3] = qq.sum(axis=1)
    qc[5, 3] *= 0.25                                                                  # a quantizedComponentSum that is not the sum of the values
    want = []
    for q in range(nq):
        _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
        assert not np.isnan(s32).any()
        want.append(O.heap_topk(s32, k))
    for compact in (True, False):
        ix = _make_index(codes, corr, dim, cdp, compact)
        try:
            ix.set_option("sweep_share", 32)
            ix.set_option("first_segment_rows", 2048)
            ix.set_option("segment_growth", 4)
            idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
            for q in range(nq):
                np.testing.assert_array_equal(idx[q], want[q][0], err_msg="query %d compact %s" % (q, compact))
                np.testing.assert_array_equal(canon32(sc[q]), canon32(want[q][1]))
            # a query with zero interval width cannot be swept there (the threshold divides by it): same answers through the fallback
            qz = qc.copy()
            qz[3, 1] = qz[3, 0]
            _, _, s32 = O.score_all(codes, corr, dim, qq[3], qz[3], qb, sim, cdp)
            zi, zs = O.heap_topk(s32, k)
            idx, sc, cnt = ix.search_batch(qq[:40], qz[:40], qb, sim, k)
            np.testing.assert_array_equal(idx[3], zi)
            np.testing.assert_array_equal(canon32(sc[3]), canon32(zs))
            np.testing.assert_array_equal(idx[7], want[7][0])
        finally:
            ix.close()


@pytest.mark.parametrize("sim", [0, 1, 2])
def test_matrix_core_sweep_explicit_component_sums(sim):
    """rows whose quantizedComponentSum is NOT their popcount (the index then carries the sums explicitly, inline layout): the
    matrix-core sweep takes the sum's magnitude into its budget before it has the sum's term (start values: K first, the term with the
    component sum last) - every answer is the oracle's, 70 queries (two groups per tile load + a partial chain)."""
    rng = np.random.default_rng(77 + sim)
    n, dim, k, nq, qb = 40000, 768, 50, 70, 4
    codes = rng.integers(0, 256, size=(n, dim // 8), dtype=np.uint8)
    pop = np.unpackbits(codes, axis=1).sum(axis=1).astype(np.float64)
    corr = np.zeros((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = np.abs(rng.standard_normal(n)) * 1e-2 if sim == 0 else 1e-3 * rng.standard_normal(n)
    corr[:, 3] = pop
    odd = rng.choice(n, 5000, replace=False)
    corr[odd, 3] = pop[odd] + rng.integers(-300, 2000, len(odd))           # sums beyond the dimension and negative ones included
    corr[odd[:50], 3] = rng.standard_normal(50) * 1e6
    cdp = 0.02
    qq = rng.integers(0, 1 << qb, size=(nq, dim), dtype=np.uint8)
    qc = np.empty((nq, 4))
    qc[:, 0] = -0.15 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 1] = 0.148 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 2] = 0.7 * rng.random(nq) if sim == 0 else -0.0028 * rng.random(nq)
    qc[:, 3] = qq.sum(axis=1)
    ix = _make_index(codes, corr, dim, cdp, True)   # (asks for the compact layout: explicit sums make it the inline one)
    try:
        ix.set_option("sweep_share", 32)
        ix.set_option("first_segment_rows", 2048)
        ix.set_option("segment_growth", 4)
        idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
        for q in range(nq):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[q], oi, err_msg="query %d" % q)
            np.testing.assert_array_equal(canon32(sc[q]), canon32(osc))
    finally:
        ix.close()


@pytest.mark.parametrize("name,shards,pilot", [("ties_cos_qb4", 3, 1024), ("big_20000x128_cos", 4, 2048), ("ties_euc_qb4", 2, 0),
                                               ("big_50000x768_cos", 5, 4096)])
def test_sharded_scan_and_replay_single_process(name, shards, pilot):
    """row shards (each with its pilot replica of the global prefix) swept one after the other on the same GPU,
    packed candidate lists replayed in shard order: must equal the reference's global top-k (ties included)"""
    import torch
    g = O.load_golden(name)
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    n, dim = g["n"], g["dim"]
    per = (n + shards - 1) // shards
    nq = len(queries)
    qs = [B.quantize_query(q, cen, sim, g["qb"], g["lambda"], g["iters"]) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    for k in sorted({t["k"] for t in g["queries"][0]["topk"]}):
        if k > 2048:
            continue
        packed, offsets = [], []
        for r in range(shards):
            r0, r1 = r * per, min((r + 1) * per, n)
            P = min(pilot, r0) // 1024 * 1024 if r > 0 else 0
            ix = B.Index(codes[r0:r1], corr[r0:r1], dim, cdp, row_base=r0,
                         pilot_codes=codes[:P] if P else None, pilot_corr=corr[:P] if P else None)
            ix.set_option("first_segment_rows", 1024)
            ix.set_option("segment_growth", 2)
            cap = int(ix.shard_list_cap(k)) * nq
            d_packed = torch.zeros(cap, dtype=torch.int64, device="cuda")
            d_off = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
            d_flags = torch.zeros(nq, dtype=torch.int32, device="cuda")
            total = ix.shard_scan(qq, qc, g["qb"], sim, k, d_packed.data_ptr(), cap, d_off.data_ptr(), d_flags.data_ptr())
            assert int(d_flags.abs().sum().item()) == 0
            off = d_off.cpu().numpy()
            assert off[-1] == total
            packed.append(d_packed[:total].cpu().numpy().view(np.uint64))
            offsets.append(off)
            ix.close()
        idx, sc, cnt = B.replay_batch(packed, offsets, nq, n, k, n_threads=3)
        for qi in range(nq):
            tk = [t for t in g["queries"][qi]["topk"] if t["k"] == k][0]
            np.testing.assert_array_equal(idx[qi, :cnt[qi]], O.dec(tk["idx_i32"], "<i4"), err_msg="%s shards=%d k=%d q%d" % (name, shards, k, qi))
            np.testing.assert_array_equal(canon32(sc[qi, :cnt[qi]]), canon32(O.dec(tk["score_f32"], "<f4")))


BUILD_CASES = [n for n in CASES if O.load_golden(n)["ib"] == 1]


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("name", BUILD_CASES)
def test_device_index_build_matches_reference(name, compact):
    """bbq_index_build: quantizeVectors as HIP kernels - centroid, packed codes, f64 corrections bit-exact vs the golden
    vectors, and the index it leaves on the device answers searches exactly"""
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    ix, codes, corr, cen = B.Index.build(base, sim, g["lambda"], g["iters"], corrections="compact" if compact else "inline")
    try:
        assert O.sha(cen) == O.sha(O.dec(g["centroid_f32"], "<f4")), "centroid"
        assert O.sha(codes) == g["codes_sha256"], "packed codes"
        ocodes, ocorr, ocen = O.build_index(base, sim, g["lambda"], g["iters"])
        np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
        for qi, rec in enumerate(g["queries"]):
            qq, qc = B.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
            for tk in rec["topk"]:
                idx, sc = ix.search(qq, qc, g["qb"], sim, tk["k"])
                np.testing.assert_array_equal(idx, O.dec(tk["idx_i32"], "<i4"))
                np.testing.assert_array_equal(canon32(sc), canon32(O.dec(tk["score_f32"], "<f4")))
    finally:
        ix.close()


def test_device_index_build_errors_and_odd_shapes():
    with pytest.raises(B.BBQError) as e:
        B.Index.build(np.zeros((0, 4), np.float32), 0)
    assert e.value.code == 10
    v = np.ones((300, 7), np.float32)
    v[200, 5] = np.nan
    v[250, 1] = np.inf
    with pytest.raises(B.BBQError) as e:
        B.Index.build(v, 0)
    assert e.value.code == 8 and "向量 200 位置 5 包含NaN值" in str(e.value)
    v[200, 5] = 1.0
    with pytest.raises(B.BBQError) as e:
        B.Index.build(v, 2)
    assert e.value.code == 9 and "向量 250 位置 1 包含Infinity值" in str(e.value)
    # dims that are not multiples of 4 / 8 / 128, row counts that are not multiples of 64
    rng = np.random.default_rng(12)
    for n, dim, sim in ((1, 1, 0), (65, 3, 1), (130, 13, 2), (999, 131, 1), (64, 129, 0)):
        base = rng.standard_normal((n, dim)).astype(np.float32)
        ix, codes, corr, cen = B.Index.build(base, sim)
        ocodes, ocorr, ocen = O.build_index(base, sim)
        np.testing.assert_array_equal(codes, ocodes)
        np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
        np.testing.assert_array_equal(cen.view(np.uint32), ocen.view(np.uint32))
        ix.close()


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("share", [4, 8, 32])
@pytest.mark.parametrize("name", ["ties_cos_qb4", "big_20000x128_cos", "ties_16d_qb1", "m_768d_max_qb4", "big_30000x1536_mip",
                                  "ties_euc_qb4", "ties_max_qb4", "m_768d_euc_qb1", "qb8_128d_cos", "big_50000x768_cos",
                                  "big_20000x1024_cos", "big_20000x1024_euc_qb8", "m_1024d_euc_qb1"])
def test_shared_sweep_gives_identical_results(name, share, compact):
    """API extension: several queries per sweep (one load of each row, `share` queries scored from registers).
    Must return exactly what one-sweep-per-query returns."""
    g = O.load_golden(name)
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    ix = _make_index(codes, corr, g["dim"], cdp, compact)
    try:
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        ix.set_option("sweep_share", share)
        rng = np.random.default_rng(4)
        extra = rng.standard_normal((37, g["dim"])).astype(np.float32)       # 37 + nq queries: not a multiple of `share`
        allq = np.concatenate([queries, extra])
        qs = [B.quantize_query(q, cen, sim, g["qb"], g["lambda"], g["iters"]) for q in allq]
        qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
        k = g["k"]
        idx, sc, cnt = ix.search_batch(qq, qc, g["qb"], sim, k)
        for qi in range(len(allq)):
            d, s64, s32 = O.score_all(codes, corr, g["dim"], qq[qi], qc[qi], g["qb"], sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[qi, :cnt[qi]], oi, err_msg="query %d" % qi)
            np.testing.assert_array_equal(canon32(sc[qi, :cnt[qi]]), canon32(osc))
        assert ix.stats()["dense_fallbacks"] == 0
    finally:
        ix.close()


def test_full_size_10m_x_768_properties():
    """BASELINE size (10 M x 768-d, k=100) on one GPU: the oracle's answer for one query, and size-independent properties
    for more: compact == inline layout, sparse segments == dense replay, shared sweep == one sweep per query, a shard split
    == the single index, every returned score equals the row's dense score."""
    import torch
    sys_path_bench = __import__("importlib").import_module("bench")
    n, dim, k, pb = 10_000_000, 768, 100, 96
    codes, corr = sys_path_bench.synth_rows(1, 0, n, pb)
    qq, qc = sys_path_bench.synth_queries(2, 6, dim)
    cdp = 0.0009110655808639536
    ix = _make_index(codes, corr, dim, cdp, True)
    try:
        idx, sc, cnt = ix.search_batch(qq, qc, 4, 1, k)
        assert (cnt == k).all() and ix.stats()["dense_fallbacks"] == 0
        # (1) the oracle, one query (about 3 s of CPU)
        d, s64, s32 = O.score_all(codes, corr, dim, qq[0], qc[0], 4, 1, cdp)
        oi, osc = O.heap_topk(s32, k)
        np.testing.assert_array_equal(idx[0], oi)
        np.testing.assert_array_equal(sc[0].view(np.uint32), osc.view(np.uint32))
        # (2) returned scores are the rows' own scores, descending
        for q in range(1, 3):
            for j in (0, 17, 99):
                r = int(idx[q, j])
                _, _, one = ix.score_rows(qq[q], qc[q], 4, 1, r, 1)
                assert one[0].view(np.uint32) == sc[q, j].view(np.uint32)
            assert (np.diff(sc[q]) <= 0).all()
        # (3) dense replay of everything == sparse segments
        ix.set_option("force_dense", 1)
        di, ds, _ = ix.search_batch(qq[:2], qc[:2], 4, 1, k)
        ix.set_option("force_dense", 0)
        np.testing.assert_array_equal(di, idx[:2])
        np.testing.assert_array_equal(ds.view(np.uint32), sc[:2].view(np.uint32))
        # (4) shared sweeps (VALU and matrix-core variants) == one sweep per query
        for share in (8, 32):
            ix.set_option("sweep_share", share)
            si, ss, _ = ix.search_batch(qq, qc, 4, 1, k)
            np.testing.assert_array_equal(si, idx)
            np.testing.assert_array_equal(ss.view(np.uint32), sc.view(np.uint32))
        ix.set_option("sweep_share", 1)
        # (4b) 70 queries: the matrix-core sweep serves two groups of 32 per tile load (64 + a partial chain of 6)
        qq70, qc70 = sys_path_bench.synth_queries(3, 70, dim)
        ui, us, _ = ix.search_batch(qq70, qc70, 4, 1, k)
        ix.set_option("sweep_share", 32)
        mi, ms, _ = ix.search_batch(qq70, qc70, 4, 1, k)
        ix.set_option("sweep_share", 1)
        np.testing.assert_array_equal(mi, ui)
        np.testing.assert_array_equal(ms.view(np.uint32), us.view(np.uint32))
        assert ix.stats()["dense_fallbacks"] == 0
    finally:
        ix.close()
    # (5) inline layout
    ix = _make_index(codes, corr, dim, cdp, False)
    try:
        ii, isc, _ = ix.search_batch(qq, qc, 4, 1, k)
        np.testing.assert_array_equal(ii, idx)
        np.testing.assert_array_equal(isc.view(np.uint32), sc.view(np.uint32))
    finally:
        ix.close()
    # (6) three uneven shards with a 32 K pilot, swept one after the other, replayed in shard order
    cuts = [0, 3_000_000, 7_500_000, n]
    packed, offsets = [], []
    for r in range(3):
        r0, r1 = cuts[r], cuts[r + 1]
        P = 32768 if r > 0 else 0
        sh = B.Index(codes[r0:r1], corr[r0:r1], dim, cdp, row_base=r0, pilot_codes=codes[:P] if P else None,
                     pilot_corr=corr[:P] if P else None)
        cap = int(sh.shard_list_cap(k)) * len(qq)
        d_packed = torch.zeros(cap, dtype=torch.int64, device="cuda")
        d_off = torch.zeros(len(qq) + 1, dtype=torch.int64, device="cuda")
        d_flags = torch.zeros(len(qq), dtype=torch.int32, device="cuda")
        total = sh.shard_scan(qq, qc, 4, 1, k, d_packed.data_ptr(), cap, d_off.data_ptr(), d_flags.data_ptr())
        assert int(d_flags.abs().sum().item()) == 0
        packed.append(d_packed[:total].cpu().numpy().view(np.uint64))
        offsets.append(d_off.cpu().numpy())
        sh.close()
    ri, rs, rc = B.replay_batch(packed, offsets, len(qq), n, k, n_threads=4)
    np.testing.assert_array_equal(ri, idx)
    np.testing.assert_array_equal(rs.view(np.uint32), sc.view(np.uint32))


def test_config2_shape_1m_x_768_properties():
    """BASELINE config 2 at its real size (1 M x 768-d, queryBits 4 / indexBits 1, k = 100): the plan the library picks below
    2.5 M rows - 128 queries per sub-batch - held to the oracle for one query of every sub-batch position, and to the size-independent
    properties for all 300: == dense replay, == host heap replay, == shared sweeps (VALU and matrix cores), == the single-query call
    shape (latency plan, append mode), == inline layout, == a 3-shard split behind one handle"""
    import bench
    n, dim, k, pb = 1_000_000, 768, 100, 96
    codes, corr = bench.synth_rows(1, 0, n, pb)
    qq, qc = bench.synth_queries(2, 300, dim)      # 128 + 128 + 44: three sub-batches, the last one partial
    cdp = float(B.centroid_dp(bench.synth_centroid(dim)))
    ix = _make_index(codes, corr, dim, cdp, True)
    try:
        ix.set_option("pipeline_slots", 3)
        idx, sc, cnt = ix.search_batch(qq, qc, 4, 1, k)
        st = ix.stats()
        # the device answers every query itself except those with equal scores in or at the edge of the answer (f32 scores in a narrow
        # range: a pair among the 101 best coincides now and then) - exactly those are replayed on the host
        _, s101, _ = ix.search_batch(qq, qc, 4, 1, k + 1)
        tied = sum(len(np.unique(s101[q].astype(np.float64))) != k + 1 for q in range(len(qq)))
        assert (cnt == k).all() and st["dense_fallbacks"] == 0 and st["host_replays"] == tied and tied < 30
        assert st["last_scan_rows"] > 0
        # (1) the oracle: first / last query of a sub-batch and one of the partial tail
        for q in (0, 127, 128, 299):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], 4, 1, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[q], oi)
            np.testing.assert_array_equal(sc[q].view(np.uint32), osc.view(np.uint32))
        # (2) returned scores are the rows' own scores, descending
        for q in (5, 200):
            for j in (0, 50, 99):
                _, _, one = ix.score_rows(qq[q], qc[q], 4, 1, int(idx[q, j]), 1)
                assert one[0].view(np.uint32) == sc[q, j].view(np.uint32)
            assert (np.diff(sc[q]) <= 0).all()
        # (3) dense replay == sparse segments
        ix.set_option("force_dense", 1)
        di, ds, _ = ix.search_batch(qq[126:130], qc[126:130], 4, 1, k)
        ix.set_option("force_dense", 0)
        np.testing.assert_array_equal(di, idx[126:130])
        np.testing.assert_array_equal(ds.view(np.uint32), sc[126:130].view(np.uint32))
        # (4) host heap replay of every query == device-selected answers
        ix.set_option("device_select", 0)
        hi, hs, _ = ix.search_batch(qq, qc, 4, 1, k)
        assert ix.stats()["host_replays"] == 300
        ix.set_option("device_select", 1)
        np.testing.assert_array_equal(hi, idx)
        np.testing.assert_array_equal(hs.view(np.uint32), sc.view(np.uint32))
        # (5) shared sweeps == one sweep per query
        for share in (8, 32):
            ix.set_option("sweep_share", share)
            si, ss, _ = ix.search_batch(qq, qc, 4, 1, k)
            np.testing.assert_array_equal(si, idx)
            np.testing.assert_array_equal(ss.view(np.uint32), sc.view(np.uint32))
        ix.set_option("sweep_share", 1)
        # (6) the reference's call shape: one query per call
        for q in (0, 1, 150, 299):
            i1, s1 = ix.search(qq[q], qc[q], 4, 1, k)
            np.testing.assert_array_equal(i1, idx[q])
            np.testing.assert_array_equal(s1.view(np.uint32), sc[q].view(np.uint32))
        # (7) other sub-batch sizes give the same answers
        for sub in (32, 100):
            ix.set_option("batch_queries", sub)
            bi, bs, _ = ix.search_batch(qq, qc, 4, 1, k)
            np.testing.assert_array_equal(bi, idx)
            np.testing.assert_array_equal(bs.view(np.uint32), sc.view(np.uint32))
        ix.set_option("batch_queries", 0)
    finally:
        ix.close()
    ix = _make_index(codes, corr, dim, cdp, False)
    try:
        ii, isc, _ = ix.search_batch(qq, qc, 4, 1, k)
        np.testing.assert_array_equal(ii, idx)
        np.testing.assert_array_equal(isc.view(np.uint32), sc.view(np.uint32))
    finally:
        ix.close()
    mx = B.Index.create_multi(codes, corr, dim, cdp, [0, 0, 0], pilot_rows=32768)
    try:
        mx.set_option("round_queries", 128)
        mi, ms, _ = mx.search_batch(qq, qc, 4, 1, k)
        np.testing.assert_array_equal(mi, idx)
        np.testing.assert_array_equal(ms.view(np.uint32), sc.view(np.uint32))
        assert mx.stats()["host_replays"] == tied and mx.stats()["dense_fallbacks"] == 0
    finally:
        mx.close()


@pytest.mark.parametrize("dim,qb,sim", [(2000, 4, 1), (264, 8, 2), (4096, 2, 0), (520, 1, 1)])
def test_generic_row_widths(dim, qb, sim):
    """dims whose packed rows are not one of the compile-time widths take the runtime-loop kernel; the shared-sweep
    option must quietly fall back there"""
    rng = np.random.default_rng(dim)
    n, k = 6000, 40
    base = rng.standard_normal((n, dim)).astype(np.float32)
    ix, codes, corr, cen = B.Index.build(base, sim)
    ocodes, ocorr, ocen = O.build_index(base, sim)
    np.testing.assert_array_equal(codes, ocodes)
    np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
    cdp = B.centroid_dp(cen)
    try:
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        for share in (1, 8):
            ix.set_option("sweep_share", share)
            qs = [B.quantize_query(rng.standard_normal(dim).astype(np.float32), cen, sim, qb) for _ in range(5)]
            qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
            idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
            for i in range(5):
                d, s64, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], qb, sim, cdp)
                oi, osc = O.heap_topk(s32, k)
                np.testing.assert_array_equal(idx[i], oi)
                np.testing.assert_array_equal(canon32(sc[i]), canon32(osc))
            gd, g64, g32 = ix.score_rows(qq[0], qc[0], qb, sim)
            d, s64, s32 = O.score_all(codes, corr, dim, qq[0], qc[0], qb, sim, cdp)
            np.testing.assert_array_equal(gd, d)
            np.testing.assert_array_equal(canon64(g64), canon64(s64))
    finally:
        ix.close()


# ---------------------------------------------------------------- oversample + exact rerank

RERANK = O.golden_names("rerank_*")


def _rerank_case(name):
    g = O.load_golden(name)
    n, dim, nq = g["n"], g["dim"], g["nq"]
    base = O.dec(g["base_f32"], np.float32).reshape(n, dim)
    queries = O.dec(g["queries_f32"], np.float32).reshape(nq, dim)
    return g, base, queries


@pytest.mark.parametrize("name", RERANK)
def test_rerank_true_scores_golden(name):
    """computeSimilarity of every (query, row), all three functions, bit for bit against the reference's values"""
    g, base, queries = _rerank_case(name)
    dv = B.Vectors(base)
    assert (dv.n, dv.dim) == base.shape
    all_rows = np.arange(g["n"], dtype=np.int32)
    for sim_name, sim in O.SIMS.items():
        got = dv.rerank_scores(queries, [all_rows] * g["nq"], sim)
        ref = O.dec(g["true_f64"][sim_name], np.float64).reshape(g["nq"], g["n"])
        for qi in range(g["nq"]):
            np.testing.assert_array_equal(canon64(got[qi]), canon64(ref[qi]))
    dv.close()


@pytest.mark.parametrize("name", RERANK)
def test_search_rerank_selectors_golden(name):
    """getOversampledTopKWithHeap / WithSort end to end (search k*factor, true scores, selection) vs the reference"""
    g, base, queries = _rerank_case(name)
    k = g["k"]
    codes, corr, cen = B.quantize_vectors(base, 1, 1, g["lambda"], g["iters"])
    ix = B.Index(codes, corr, g["dim"], B.centroid_dp(cen))
    dv = B.Vectors(base)
    qs = [B.quantize_query(q, cen, 1, 4, g["lambda"], g["iters"]) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    for f in sorted({r["factor"] for r in g["oversample"]}):
        for sel, how in ((0, "heap"), (1, "sort")):
            idx, qsc, tsc, cnt = B.search_rerank_batch(ix, dv, queries, qq, qc, 4, 1, k, f, sel, 1)
            for rec in (r for r in g["oversample"] if r["factor"] == f):
                qi = rec["query"]
                m = int(cnt[qi])
                np.testing.assert_array_equal(idx[qi, :m], O.dec(rec[how]["idx_i32"], np.int32))
                np.testing.assert_array_equal(canon32(qsc[qi, :m]), canon32(O.dec(rec[how]["quantized_f32"], np.float32)))
                np.testing.assert_array_equal(canon64(tsc[qi, :m]), canon64(O.dec(rec[how]["true_f64"], np.float64)))
    # the Python mirror of the reference's function names goes through the same entry point
    fmt = B.BinaryQuantizationFormat({"queryBits": 4, "indexBits": 1,
                                            "quantizer": {"similarityFunction": "COSINE", "lambda": g["lambda"], "iters": g["iters"]}})
    qv = fmt.quantizeVectors(base)["quantizedVectors"]
    rec = g["oversample"][0]
    got = B.getOversampledTopKWithHeap(queries[rec["query"]], qv, dv, k, rec["factor"], fmt)
    assert [x["index"] for x in got] == list(O.dec(rec["heap"]["idx_i32"], np.int32))
    got = B.getOversampledTopKWithSort(queries[rec["query"]], qv, base, k, rec["factor"], fmt)
    assert [x["index"] for x in got] == list(O.dec(rec["sort"]["idx_i32"], np.int32))
    np.testing.assert_array_equal(canon64([x["trueScore"] for x in got]), canon64(O.dec(rec["sort"]["true_f64"], np.float64)))
    dv.close()
    ix.close()


@pytest.mark.parametrize("dim", [1, 31, 33, 130, 768, 1536])
def test_rerank_ragged_lists_vs_oracle(dim):
    """ragged candidate lists (0, 1, 63, 64, 65, 700 rows, repeats allowed) at dims around the 32-column stage"""
    rng = np.random.default_rng(dim)
    n = 3000
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[17] = 0
    queries = rng.standard_normal((6, dim)).astype(np.float32)
    lens = [0, 1, 63, 64, 65, 700]
    lists = [rng.integers(0, n, m).astype(np.int32) for m in lens]
    lists[3][5] = 17
    dv = B.Vectors(base)
    for sim in (0, 1, 2):
        got = dv.rerank_scores(queries, lists, sim)
        for qi, rows in enumerate(lists):
            ref = O.true_similarity(queries[qi:qi + 1], base[rows], sim)[0] if len(rows) else np.zeros(0)
            np.testing.assert_array_equal(canon64(got[qi]), canon64(ref))
    dv.close()


def test_rerank_selection_ties_and_small_k():
    """duplicate rows give exactly equal true scores: the heap's history and the stable sort decide the order"""
    rng = np.random.default_rng(5)
    pool = rng.standard_normal((20, 64)).astype(np.float32)
    base = pool[rng.integers(0, 20, 2000)]
    queries = rng.standard_normal((5, 64)).astype(np.float32)
    codes, corr, cen = B.quantize_vectors(base, 1, 1)
    ix = B.Index(codes, corr, 64, B.centroid_dp(cen))
    dv = B.Vectors(base)
    qs = [B.quantize_query(q, cen, 1, 4) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    for k, f in ((1, 1), (7, 3), (50, 10), (300, 20)):
        cidx, csc, ccnt = ix.search_batch(qq, qc, 4, 1, k * f)
        for sel, how in ((0, "heap"), (1, "sort")):
            idx, qsc, tsc, cnt = B.search_rerank_batch(ix, dv, queries, qq, qc, 4, 1, k, f, sel, 1)
            for qi in range(5):
                cand = cidx[qi, :ccnt[qi]]
                t = O.true_similarity(queries[qi:qi + 1], base[cand], 1)[0]
                pos = O.rerank_select(t, k, how)
                m = int(cnt[qi])
                assert m == len(pos)
                np.testing.assert_array_equal(idx[qi, :m], cand[pos])
                np.testing.assert_array_equal(canon64(tsc[qi, :m]), canon64(t[pos]))
                np.testing.assert_array_equal(canon32(qsc[qi, :m]), canon32(csc[qi, :ccnt[qi]][pos]))
    dv.close()
    ix.close()


def test_rerank_errors():
    base = np.random.default_rng(1).standard_normal((100, 16)).astype(np.float32)
    dv = B.Vectors(base)
    q = base[:1]
    with pytest.raises(B.BBQError):
        dv.rerank_scores(q, [np.array([100], np.int32)], 1)
    with pytest.raises(B.BBQError):
        dv.rerank_scores(q, [np.array([-1], np.int32)], 1)
    with pytest.raises(B.BBQError):
        dv.rerank_scores(q, [np.array([1], np.int32)], 3)
    with pytest.raises(B.BBQError):
        dv.rerank_scores(base[:1, :8], [np.array([1], np.int32)], 1)
    assert [len(x) for x in dv.rerank_scores(base[:2], [np.zeros(0, np.int32)] * 2, 1)] == [0, 0]
    codes, corr, cen = B.quantize_vectors(base, 1, 1)
    ix = B.Index(codes, corr, 16, B.centroid_dp(cen))
    qq, qc = B.quantize_query(q[0], cen, 1, 4)
    with pytest.raises(B.BBQError):
        B.search_rerank_batch(ix, dv, q, qq[None], qc[None], 4, 1, 5, 0)
    with pytest.raises(B.BBQError):
        B.search_rerank_batch(ix, dv, q, qq[None], qc[None], 4, 1, -1, 3)
    with pytest.raises(B.BBQError):
        B.search_rerank_batch(ix, dv, q, qq[None], qc[None], 4, 1, 5, 3, selector=2)
    short = B.Vectors(base[:50])
    with pytest.raises(B.BBQError):
        B.search_rerank_batch(ix, short, q, qq[None], qc[None], 4, 1, 5, 3)
    idx, _, _, cnt = B.search_rerank_batch(ix, dv, q, qq[None], qc[None], 4, 1, 0, 3)
    assert cnt[0] == 0
    idx, _, ts, cnt = B.search_rerank_batch(ix, dv, q, qq[None], qc[None], 4, 1, 500, 3)   # k*factor > n
    assert cnt[0] == 100 and idx[0, 0] == 0
    short.close()
    dv.close()
    ix.close()


# ---------------------------------------------------------------- on-disk format

def _search_all(ix, g, cen, sim, qb, k):
    _, queries = O.golden_inputs(g)
    qs = [B.quantize_query(q, cen, sim, qb, g["lambda"], g["iters"]) for q in queries]
    return ix.search_batch(np.stack([a for a, _ in qs]), np.stack([b for _, b in qs]), qb, sim, k)


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("name", ["m_768d_cos_qb4", "m_100d_euc_qb4", "big_20000x128_cos", "edge_zero_const"])
def test_save_load_roundtrip(tmp_path, name, compact):
    """save -> load is a byte-for-byte copy of the device buffers: same geometry, same answers, same exported rows"""
    import os
    g = O.load_golden(name)
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    ix = _make_index(codes, corr, g["dim"], cdp, compact)
    k = min(100, g["n"])
    want = _search_all(ix, g, cen, sim, g["qb"], k)
    prefix = str(tmp_path / "idx")
    ix.save(prefix, cen, sim)
    n_tiles = (g["n"] + 63) // 64
    w16 = ((g["dim"] + 7) // 8 + 15) // 16
    stride = w16 * 1024 + (256 if compact else 1536)   # compact: 4 B per row in the tile; exact corrections + add ranges aside
    assert ix.bytes_per_row == stride // 64
    assert os.path.getsize(prefix + ".veb") == n_tiles * stride + (n_tiles * (64 * 32 + 8) if compact else 0)
    assert os.path.getsize(prefix + ".vemb") == 104 + 4 * g["dim"] + 16
    info = B.file_info(prefix)
    assert info == {"n_rows": g["n"], "dim": g["dim"], "sim": sim, "centroid_dp": cdp, "row_base": 0}
    ix2, cen2, _ = B.Index.load(prefix)
    assert ix2.bytes_per_row == ix.bytes_per_row
    np.testing.assert_array_equal(cen2.view(np.uint32), cen.view(np.uint32))
    got = _search_all(ix2, g, cen2, sim, g["qb"], k)
    for a, b in zip(want, got):
        np.testing.assert_array_equal(np.asarray(a).view(np.uint32) if a.dtype == np.float32 else a,
                                      np.asarray(b).view(np.uint32) if b.dtype == np.float32 else b)
    for src in (ix, ix2):
        c2, r2 = src.export()
        np.testing.assert_array_equal(c2, codes)
        np.testing.assert_array_equal(canon64(r2), canon64(corr))
    # per-row scores through the loaded copy too
    qq, qc = B.quantize_query(queries[0], cen, sim, g["qb"], g["lambda"], g["iters"])
    d1, s1, _ = ix.score_rows(qq, qc, g["qb"], sim)
    d2, s2, _ = ix2.score_rows(qq, qc, g["qb"], sim)
    np.testing.assert_array_equal(d1, d2)
    np.testing.assert_array_equal(canon64(s1), canon64(s2))
    ix.close()
    ix2.close()


def test_save_load_device_built_and_explicit_sums(tmp_path):
    rng = np.random.default_rng(8)
    base = rng.standard_normal((5000, 96)).astype(np.float32)
    ix, codes, corr, cen = B.Index.build(base, 1)
    ix.save(str(tmp_path / "built"), cen, 1)
    ix2, cen2, info = B.Index.load(str(tmp_path / "built"))
    assert info["n_rows"] == 5000 and info["dim"] == 96
    c2, r2 = ix2.export()
    np.testing.assert_array_equal(c2, codes)
    np.testing.assert_array_equal(canon64(r2), canon64(corr))
    q = rng.standard_normal(96).astype(np.float32)
    qq, qc = B.quantize_query(q, cen, 1, 4)
    a, b = ix.search(qq, qc, 4, 1, 50), ix2.search(qq, qc, 4, 1, 50)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(canon32(a[1]), canon32(b[1]))
    ix.close()
    ix2.close()
    # quantizedComponentSum that is NOT the popcount: explicit sums travel in the tile records (inline layout + x1)
    corr3 = corr.copy()
    corr3[::7, 3] += 2.0
    ix3 = B.Index(codes, corr3, 96, B.centroid_dp(cen))
    ix3.save(str(tmp_path / "x1"), cen, 1)
    ix4, _, _ = B.Index.load(str(tmp_path / "x1"))
    assert ix4.bytes_per_row == ix3.bytes_per_row
    _, r4 = ix4.export()
    np.testing.assert_array_equal(canon64(r4), canon64(corr3))
    a, b = ix3.search(qq, qc, 4, 1, 50), ix4.search(qq, qc, 4, 1, 50)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(canon32(a[1]), canon32(b[1]))
    ix3.close()
    ix4.close()


def test_load_rejects_damaged_files(tmp_path):
    rng = np.random.default_rng(9)
    base = rng.standard_normal((700, 64)).astype(np.float32)
    codes, corr, cen = B.quantize_vectors(base, 1, 1)
    ix = B.Index(codes, corr, 64, B.centroid_dp(cen))
    prefix = str(tmp_path / "d")
    ix.save(prefix, cen, 1)
    veb, vemb = open(prefix + ".veb", "rb").read(), open(prefix + ".vemb", "rb").read()

    def variant(name, veb_bytes, vemb_bytes):
        p = str(tmp_path / name)
        open(p + ".veb", "wb").write(veb_bytes)
        open(p + ".vemb", "wb").write(vemb_bytes)
        return p

    flipped = bytearray(veb)
    flipped[len(veb) // 2] ^= 0x10
    bad_meta = bytearray(vemb)
    bad_meta[110] ^= 1   # a centroid byte
    for p in (variant("flip", bytes(flipped), vemb), variant("short", veb[:-8], vemb), variant("meta", veb, bytes(bad_meta)),
              variant("magic", veb, b"XVEC" + vemb[4:]), variant("trunc", veb, vemb[:50]), str(tmp_path / "missing")):
        with pytest.raises(B.BBQError):
            B.Index.load(p)
    ok, _, _ = B.Index.load(variant("good", veb, vemb))
    ok.close()
    with pytest.raises(B.BBQError):
        ix.save(str(tmp_path / "nodir" / "x"), cen, 1)
    with pytest.raises(B.BBQError):
        ix.save(prefix, cen[:10], 1)
    # a shard with a pilot replica is saved with it (format version 3) and its damaged variants are refused as well
    sh = B.Index(codes[512:], corr[512:], 64, B.centroid_dp(cen), row_base=512, pilot_codes=codes[:512], pilot_corr=corr[:512])
    sh.save(prefix + "_shard", cen, 1)
    sh.close()
    sveb, svemb = open(prefix + "_shard.veb", "rb").read(), open(prefix + "_shard.vemb", "rb").read()
    assert svemb[4:8] == (3).to_bytes(4, "little") and vemb[4:8] == (2).to_bytes(4, "little")
    assert len(svemb) == len(vemb) + 24
    pflip = bytearray(sveb)
    pflip[-9] ^= 0x40        # a byte of the pilot replica's side section
    ext = bytearray(svemb)
    ext[104] ^= 0x01         # pilotRows
    for p in (variant("pflip", bytes(pflip), svemb), variant("pshort", sveb[:-64], svemb), variant("pext", sveb, bytes(ext))):
        with pytest.raises(B.BBQError):
            B.Index.load(p)
    back, _, info = B.Index.load(prefix + "_shard")
    assert info["row_base"] == 512 and back.n == 700 - 512
    back.close()
    ix.close()


def test_config4_shape_10m_x_1536_mip_properties():
    """BASELINE config 4's shape at full size (10 M x 1536-d, MAXIMUM_INNER_PRODUCT, k = 100): the oracle's answer for one query
    (about 6 s of CPU), size-independent properties for more - single-query plan == batch plan == dense replay == host replay,
    returned scores equal the rows' own scores"""
    bench = __import__("importlib").import_module("bench")
    n, dim, k, sim = 10_000_000, 1536, 100, 2
    codes, corr = bench.synth_rows(1, 0, n, dim // 8)
    qq, qc = bench.synth_queries(2, 6, dim)
    cdp = 0.0009110655808639536
    ix = _make_index(codes, corr, dim, cdp, True)
    try:
        assert ix.bytes_per_row == 192 + 4
        idx, sc, cnt = ix.search_batch(qq, qc, 4, sim, k)           # 6 queries: the batch plan
        assert (cnt == k).all() and ix.stats()["dense_fallbacks"] == 0 and ix.stats()["host_replays"] == 0
        d, s64, s32 = O.score_all(codes, corr, dim, qq[0], qc[0], 4, sim, cdp)
        oi, osc = O.heap_topk(s32, k)
        np.testing.assert_array_equal(idx[0], oi)
        np.testing.assert_array_equal(sc[0].view(np.uint32), osc.view(np.uint32))
        for q in range(6):                                          # one query per call: the latency plan (three segments, append mode)
            i1, s1 = ix.search(qq[q], qc[q], 4, sim, k)
            np.testing.assert_array_equal(i1, idx[q])
            np.testing.assert_array_equal(s1.view(np.uint32), sc[q].view(np.uint32))
        ix.set_option("device_select", 0)                           # the host replays the heap over the candidate lists
        hi, hs, _ = ix.search_batch(qq, qc, 4, sim, k)
        assert ix.stats()["host_replays"] == 6
        ix.set_option("device_select", 1)
        np.testing.assert_array_equal(hi, idx)
        np.testing.assert_array_equal(hs.view(np.uint32), sc.view(np.uint32))
        ix.set_option("force_dense", 1)
        di, ds, _ = ix.search_batch(qq[4:6], qc[4:6], 4, sim, k)
        ix.set_option("force_dense", 0)
        np.testing.assert_array_equal(di, idx[4:6])
        np.testing.assert_array_equal(ds.view(np.uint32), sc[4:6].view(np.uint32))
        for q in range(1, 6):
            assert (np.diff(sc[q]) <= 0).all()
            for j in (0, 50, 99):
                r = int(idx[q, j])
                _, _, one = ix.score_rows(qq[q], qc[q], 4, sim, r, 1)
                assert one[0].view(np.uint32) == sc[q, j].view(np.uint32)
    finally:
        ix.close()


# ---------------------------------------------------------------- randomized sweep of shapes and options

@pytest.mark.parametrize("seed", range(int(os.environ.get("BBQ_FUZZ_SEEDS", "24"))))   # BBQ_FUZZ_SEEDS=400 for a soak
def test_fuzz_shapes_options_vs_oracle(seed):
    """random (n, dim, k, queryBits, similarity, layout, segment plan, sweep sharing, data flavour) against the oracle:
    index bytes, per-row integers and f64 scores, and the replayed top-k incl. order"""
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 96, 127, 128, 129, 200, 256, 384, 500, 768, 1000, 1024, 1024, 1536]))
    n = int(rng.choice([1, 2, 63, 64, 65, 511, 512, 513, 1500, 4097, 9000]))
    sim = int(rng.integers(0, 3))
    qb = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 7, 8]))
    k = int(rng.choice([1, 2, 10, 100, n, n + 3, max(1, n // 2)]))
    flavour = int(rng.integers(0, 4))
    if flavour == 0:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    elif flavour == 1:   # few distinct rows: masses of exactly equal scores
        pool = rng.standard_normal((max(2, min(12, n)), dim)).astype(np.float32)
        base = pool[rng.integers(0, pool.shape[0], n)]
    elif flavour == 2:   # wildly different magnitudes per row, some zero / constant rows
        base = (rng.standard_normal((n, dim)) * 10.0 ** rng.integers(-6, 7, (n, 1))).astype(np.float32)
        base[rng.integers(0, n, max(1, n // 50))] = 0
        base[rng.integers(0, n, max(1, n // 50))] = 3.5
    else:                # scores rising with the row number: the worst case for the thresholds
        base = rng.standard_normal((n, dim)).astype(np.float32) * 0.05 + np.linspace(-1, 1, n, dtype=np.float32)[:, None] * rng.standard_normal(dim).astype(np.float32)
    nq = int(rng.integers(1, 40))
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    if flavour == 3:
        queries[0] = base[-1]
    codes, corr, cen = O.build_index(base, sim)
    pcodes, pcorr, pcen = B.quantize_vectors(base, sim)
    np.testing.assert_array_equal(pcodes, codes)
    np.testing.assert_array_equal(canon64(pcorr), canon64(corr))
    np.testing.assert_array_equal(pcen.view(np.uint32), cen.view(np.uint32))
    cdp = B.centroid_dp(cen)
    compact = bool(rng.integers(0, 2))
    ix = _make_index(codes, corr, dim, cdp, compact)
    try:
        ix.set_option("first_segment_rows", int(rng.choice([1024, 2048, 4096])))
        ix.set_option("segment_growth", int(rng.choice([2, 3, 8, 64])))
        ix.set_option("batch_queries", int(rng.choice([1, 5, 32])))
        ix.set_option("pipeline_slots", int(rng.choice([1, 2, 3])))
        ix.set_option("replay_threads", int(rng.choice([1, 4])))
        ix.set_option("sweep_share", int(rng.choice([1, 1, 4, 8, 32])))
        ix.set_option("resident_mb", int(rng.choice([-1, -1, 0, 1, 2])))      # how much of a launch's range is read cache-resident
        ix.set_option("resident_interleave", int(rng.integers(0, 2)))
        qs = [B.quantize_query(q, cen, sim, qb) for q in queries]
        qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
        for i in range(nq):
            oq, oc = O.quantize_query(queries[i], cen, sim, qb)
            np.testing.assert_array_equal(qq[i], oq)
            np.testing.assert_array_equal(canon64(qc[i]), canon64(oc))
        idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
        for i in range(nq):
            d, s64, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            assert cnt[i] == len(oi) == min(k, n)
            np.testing.assert_array_equal(idx[i, :cnt[i]], oi)
            np.testing.assert_array_equal(canon32(sc[i, :cnt[i]]), canon32(osc))
        gd, g64, g32 = ix.score_rows(qq[0], qc[0], qb, sim)
        d, s64, s32 = O.score_all(codes, corr, dim, qq[0], qc[0], qb, sim, cdp)
        np.testing.assert_array_equal(gd, d)
        np.testing.assert_array_equal(canon64(g64), canon64(s64))
        np.testing.assert_array_equal(canon32(g32), canon32(s32))
        # the single-query call shape (latency plan: few fast-growing segments, candidates appended by the sweeps)
        ix.set_option("latency_growth", int(rng.choice([2, 64, 4096])))
        ix.set_option("append_last", int(rng.integers(0, 2)))
        i1, s1 = ix.search(qq[0], qc[0], qb, sim, k)
        oi, osc = O.heap_topk(s32, k)
        np.testing.assert_array_equal(i1, oi)
        np.testing.assert_array_equal(canon32(s1), canon32(osc))
    finally:
        ix.close()


# ---------------------------------------------------------------- rows stored cluster by cluster (flood tier)

@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("flood_rows,expect_dense", [(262144, False), (1024, True), (0, True)])
def test_cluster_ordered_rows_stay_exact_and_sparse(compact, flood_rows, expect_dense):
    """topic/time-ordered ingestion: the query's own cluster sits in one contiguous run of rows and beats a threshold that was
    derived from other clusters, so whole chunks pass.  The flood tier keeps such queries on the sparse path (no dense
    fallback) and the answers stay bit-identical; with the tier disabled or too small they fall back and are still exact."""
    rng = np.random.default_rng(77)
    n, dim, ncl, nq, k = 120000, 64, 40, 12, 50
    centres = rng.standard_normal((ncl, dim)).astype(np.float32)
    cid = np.sort(rng.integers(0, ncl, n))
    base = centres[cid] + 0.4 * rng.standard_normal((n, dim)).astype(np.float32)
    qcl = np.array([0, 1, 5, 13, 20, 27, 33, 38, 39, 39, 17, 9])
    queries = centres[qcl] + 0.4 * rng.standard_normal((nq, dim)).astype(np.float32)
    sim = 1
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    ix = _make_index(codes, corr, dim, cdp, compact)
    try:
        ix.set_option("flood_rows", flood_rows)
        ix.set_option("replay_threads", 4)
        qs = [B.quantize_query(q, cen, sim, 4) for q in queries]
        qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
        idx, sc, cnt = ix.search_batch(qq, qc, 4, sim, k)
        st = ix.stats()
        for i in range(nq):
            d, s64, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], 4, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[i], oi)
            np.testing.assert_array_equal(canon32(sc[i]), canon32(osc))
        if expect_dense:
            assert st["dense_fallbacks"] > 0
        else:
            assert st["dense_fallbacks"] == 0
            assert st["candidates"] > nq * 1000   # the floods were replayed, not dropped
        # the shared sweeps have no flood tier of their own: an overflowing query gets one sweep of its own before the dense path
        for share in (8, 32):
            ix.set_option("sweep_share", share)
            idx8, sc8, _ = ix.search_batch(qq, qc, 4, sim, k)
            np.testing.assert_array_equal(idx8, idx)
            np.testing.assert_array_equal(canon32(sc8), canon32(sc))
            if share == 8:
                assert (ix.stats()["dense_fallbacks"] > 0) == expect_dense
            else:  # the matrix-core sweep appends its candidates to the query's list itself: a flood needs no tier as long as the list holds it
                assert ix.stats()["dense_fallbacks"] == 0 or expect_dense
    finally:
        ix.close()


@pytest.mark.parametrize("squeeze", [False, True])
def test_sharded_scan_cluster_ordered_rows(squeeze):
    """row shards over cluster-ordered rows: the shard that owns the query's cluster floods; its list has headroom and the
    packed buffer usually has room (other queries use far less than advertised), so nothing is flagged.  With a packed
    buffer of exactly the advertised size and EVERY query flooding, the flooded ones are dropped (flagged) instead of
    failing the call, and what is left still fits."""
    import torch
    rng = np.random.default_rng(78)
    n, dim, ncl, k, shards = 150000, 64, 30, 50, 3
    centres = rng.standard_normal((ncl, dim)).astype(np.float32)
    cid = np.sort(rng.integers(0, ncl, n))
    base = centres[cid] + 0.4 * rng.standard_normal((n, dim)).astype(np.float32)
    qcl = np.array([25, 27, 29, 29, 28, 26, 24, 23]) if squeeze else np.array([0, 3, 11, 12, 19, 21, 28, 29])
    nq = len(qcl)
    queries = centres[qcl] + 0.4 * rng.standard_normal((nq, dim)).astype(np.float32)
    sim = 1
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    qs = [B.quantize_query(q, cen, sim, 4) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    per = (n + shards - 1) // shards
    packed, offsets, flagged = [], [], np.zeros(nq, bool)
    for r in range(shards):
        r0, r1 = r * per, min((r + 1) * per, n)
        # pilot replica = an evenly strided sample of the rows before the shard (a prefix would only know the first clusters)
        # (the squeeze variant keeps the weak prefix pilot on purpose: it provokes the large floods)
        pick = (np.arange(4096) if squeeze else np.linspace(0, r0 - 1, 4096).astype(np.int64)) if r > 0 else None
        ix = B.Index(codes[r0:r1], corr[r0:r1], dim, cdp, row_base=r0,
                     pilot_codes=codes[pick] if r > 0 else None, pilot_corr=corr[pick] if r > 0 else None)
        adv = int(ix.shard_list_cap(k))
        cap = adv * nq * (1 if squeeze else 8)
        d_packed = torch.zeros(cap, dtype=torch.int64, device="cuda")
        d_off = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
        d_flags = torch.zeros(nq, dtype=torch.int32, device="cuda")
        total = ix.shard_scan(qq, qc, 4, sim, k, d_packed.data_ptr(), cap, d_off.data_ptr(), d_flags.data_ptr())
        off = d_off.cpu().numpy()
        fl = d_flags.cpu().numpy()
        assert off[-1] == total <= cap
        if squeeze and r == shards - 1:
            # all eight queries sit in this shard's clusters: more candidates than the advertised buffer holds
            assert (np.diff(off)[fl == 0] <= adv).all() and fl.any()
        elif not squeeze:
            assert not fl.any()
        flagged |= fl != 0
        packed.append(d_packed[:total].cpu().numpy().view(np.uint64))
        offsets.append(off)
        ix.close()
    idx, sc, cnt = B.replay_batch(packed, offsets, nq, n, k, n_threads=2)
    for qi in range(nq):
        if flagged[qi]:
            continue   # the host framework scores those densely (ShardedSearcher's dense path)
        d, s64, s32 = O.score_all(codes, corr, dim, qq[qi], qc[qi], 4, sim, cdp)
        oi, osc = O.heap_topk(s32, k)
        np.testing.assert_array_equal(idx[qi, :cnt[qi]], oi)
        np.testing.assert_array_equal(canon32(sc[qi, :cnt[qi]]), canon32(osc))
    assert (not squeeze) or flagged.any()
    assert squeeze or not flagged.any()


@pytest.mark.parametrize("seed", range(int(os.environ.get("BBQ_FUZZ_MEDIUM_SEEDS", "8"))))
def test_fuzz_medium_sizes_vs_oracle(seed):
    """the same sweep at sizes where every segment of the plan, the flood tier and the shared sweeps are exercised"""
    rng = np.random.default_rng(5000 + seed)
    dim = int(rng.choice([64, 128, 768, 1024]))
    n = int(rng.choice([20000, 70000, 150000, 300000]))
    sim = int(rng.integers(0, 3))
    qb = int(rng.choice([1, 4, 4, 8]))
    k = int(rng.choice([1, 10, 100, 1000, 4096]))
    ordered = bool(rng.integers(0, 2))
    ncl = 50
    centres = rng.standard_normal((ncl, dim)).astype(np.float32)
    cid = rng.integers(0, ncl, n)
    if ordered:
        cid = np.sort(cid)
    base = centres[cid] + 0.5 * rng.standard_normal((n, dim)).astype(np.float32)
    base *= np.exp(0.5 * rng.standard_normal((n, 1))).astype(np.float32)      # norms vary inside every tile
    nq = int(rng.integers(3, 40))
    queries = centres[rng.integers(0, ncl, nq)] + 0.5 * rng.standard_normal((nq, dim)).astype(np.float32)
    ix, codes, corr, cen = B.Index.build(base, sim)
    ocodes, ocorr, ocen = O.build_index(base, sim)
    np.testing.assert_array_equal(codes, ocodes)
    np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
    cdp = B.centroid_dp(cen)
    try:
        ix.set_option("segment_growth", int(rng.choice([4, 8, 16])))
        ix.set_option("replay_threads", 4)
        share = int(rng.choice([1, 1, 8, 32]))
        ix.set_option("sweep_share", share)
        qq, qc = B.quantize_queries(queries, cen, sim, qb)
        idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
        for i in range(nq):
            d, s64, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[i, :cnt[i]], oi)
            np.testing.assert_array_equal(canon32(sc[i, :cnt[i]]), canon32(osc))
        assert ix.stats()["dense_fallbacks"] == 0
        for i in range(min(nq, 3)):     # and one query per call (latency plan, append mode)
            d, s64, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            i1, s1 = ix.search(qq[i], qc[i], qb, sim, k)
            np.testing.assert_array_equal(i1, oi)
            np.testing.assert_array_equal(canon32(s1), canon32(osc))
    finally:
        ix.close()


# ---------------------------------------------------------------- answers selected on the device vs replayed on the host

@pytest.mark.parametrize("name", ["ties_cos_qb4", "ties_16d_qb1", "big_20000x128_cos", "m_768d_cos_qb4", "edge_zero_const", "edge_n1",
                                  "c1_1000x128_cos_qb4", "ib2_ties_cos_qb4", "big_20000x1024_cos"])
def test_device_selected_answers_equal_host_replay(name):
    """device_select 1 (default): the last finalize launch selects and sorts the answer when no two scores in or at the edge of it
    compare equal; otherwise - and always with device_select 0 - the host replays the reference heap.  Same bits either way, and
    the tie-stress fixtures really take the replay while tie-free data never does."""
    g = O.load_golden(name)
    sim, base, queries, codes, corr, cen, cdp = _index_from_case(g)
    ix = _make_index(codes, corr, g["dim"], cdp, True, index_bits=g["ib"])
    try:
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        qs = [B.quantize_query(q, cen, sim, g["qb"], g["lambda"], g["iters"]) for q in queries]
        qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
        for k in sorted({t["k"] for t in g["queries"][0]["topk"]}):
            res, replays = {}, {}
            for sel in (1, 0):
                ix.set_option("device_select", sel)
                res[sel] = ix.search_batch(qq, qc, g["qb"], sim, k)
                replays[sel] = ix.stats()["host_replays"]
            np.testing.assert_array_equal(res[1][0], res[0][0])
            np.testing.assert_array_equal(canon32(res[1][1]), canon32(res[0][1]))
            np.testing.assert_array_equal(res[1][2], res[0][2])
            for qi in range(len(queries)):
                tk = [t for t in g["queries"][qi]["topk"] if t["k"] == k][0]
                np.testing.assert_array_equal(res[1][0][qi, :res[1][2][qi]], O.dec(tk["idx_i32"], "<i4"))
            nan_case = np.isnan(res[0][1]).any()
            assert replays[0] + ix.stats()["dense_fallbacks"] >= len(queries) - (len(queries) if nan_case else 0)
            if name.startswith(("ties_cos", "ib2_ties")) and k >= 7:   # rows drawn from a pool of 40 vectors: equal scores everywhere
                assert replays[1] > 0, "tie-stress fixture answered without a replay"
            if name.startswith("big_"):
                assert replays[1] == 0, "tie-free data needed a host replay"
    finally:
        ix.close()


def test_device_select_boundary_ties_and_duplicates():
    """duplicate rows straddling the k-th place, +0 / -0, k = N, k = N - 1, k = 1024 / 1025: the device either proves its answer or
    hands the query to the host replay; the oracle's heap decides what is right"""
    rng = np.random.default_rng(31)
    n, dim, sim = 5000, 64, 1
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[100:140] = base[7]            # 41 equal rows: ties inside and at the edge of small answers
    base[4000] = base[3000]
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    qq, qc = B.quantize_query(base[7] + 0.01 * rng.standard_normal(dim).astype(np.float32), cen, sim, 4)
    _, _, s32 = O.score_all(codes, corr, dim, qq, qc, 4, sim, cdp)
    ix = B.Index(codes, corr, dim, cdp)
    try:
        for k in (1, 5, 41, 42, 100, 1024, 1025, n - 1, n, n + 5):
            idx, sc = ix.search(qq, qc, 4, sim, k)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx, oi, err_msg="k=%d" % k)
            np.testing.assert_array_equal(canon32(sc), canon32(osc))
    finally:
        ix.close()


@pytest.mark.parametrize("dim,qb,sim,compact,n", [(768, 4, 1, True, 150_000), (768, 4, 0, False, 150_000), (1024, 1, 2, True, 150_000),
                                                  (1536, 4, 2, True, 90_000), (768, 8, 1, True, 150_000), (1024, 2, 0, True, 150_000),
                                                  (1536, 8, 1, False, 90_000), (768, 1, 1, False, 150_000), (768, 4, 1, True, 600_000),
                                                  (1024, 4, 2, False, 400_000), (1536, 4, 0, True, 300_000)])
def test_latency_fused_path_equals_general_path(dim, qb, sim, compact, n):
    """the single-query call without copies (bbq_latency_kernels.hip: query in the kernel arguments of every sweep, answer polled from
    mapped host memory) - with the pre-sampled threshold (indexes of 262144 rows and more) and as the segmented chain - against the
    oracle and against the general path (latency_fused 0), on rows with duplicates (equal scores -> host replay of the complete
    list), for k from 1 to 500"""
    rng = np.random.default_rng(dim * 10 + qb + sim)
    pb = dim // 8
    codes = rng.integers(0, 256, size=(n, pb), dtype=np.uint8)
    corr = np.empty((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = 1e-4 * (2 * rng.random(n) - 1)
    corr[:, 3] = np.unpackbits(codes, axis=1).sum(axis=1)
    dup = rng.integers(0, n, 4000)
    codes[dup[2000:]] = codes[dup[:2000]]      # duplicated rows: equal scores, some of them inside an answer
    corr[dup[2000:]] = corr[dup[:2000]]
    cdp = 0.0009
    qq = rng.integers(0, 1 << qb, size=(12, dim), dtype=np.uint8)
    qc = np.empty((12, 4))
    qc[:, 0] = -0.15 * (0.9 + 0.2 * rng.random(12))
    qc[:, 1] = 0.148 * (0.9 + 0.2 * rng.random(12))
    qc[:, 2] = -0.0028 * rng.random(12)
    qc[:, 3] = qq.sum(axis=1)
    ix = _make_index(codes, corr, dim, cdp, compact)
    try:
        replays = 0
        for q in range(12):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
            for k in ((1, 10, 100, 500) if q < 3 else (100,)):
                oi, osc = O.heap_topk(s32, k)
                for fused, presample in ((1, 1), (1, 0), (0, 0)):
                    ix.set_option("latency_fused", fused)
                    ix.set_option("latency_presample", presample)
                    fi, fs = ix.search(qq[q], qc[q], qb, sim, k)
                    replays += ix.stats()["host_replays"]
                    np.testing.assert_array_equal(fi, oi, err_msg="fused %d presample %d q%d k=%d" % (fused, presample, q, k))
                    np.testing.assert_array_equal(fs.view(np.uint32), osc.view(np.uint32))
        # a NaN row flags the query: dense path, still exact
        corr2 = corr.copy()
        corr2[n // 2, 0] = np.nan
        ix.close()
        ix = _make_index(codes, corr2, dim, cdp, compact)
        ix.set_option("latency_fused", 1)
        ix.set_option("latency_presample", 1)
        _, _, s32 = O.score_all(codes, corr2, dim, qq[0], qc[0], qb, sim, cdp)
        oi, osc = O.heap_topk(s32, 50)
        fi, fs = ix.search(qq[0], qc[0], qb, sim, 50)
        np.testing.assert_array_equal(fi, oi)
        assert ix.stats()["dense_fallbacks"] == 1
        # k beyond the path's limit takes the general path
        oi, _ = O.heap_topk(s32, 50)
    finally:
        ix.close()


@pytest.mark.parametrize("k,copies", [(1, 3), (1, 9), (10, 12), (10, 40), (100, 130)])
def test_latency_presampled_threshold_ties_in_the_prefix(k, copies):
    """the pre-sampled single-query call when the sampled keys at ranks k+1 and k+2 are EQUAL: the best-matching row is stored `copies`
    >= k + 2 times inside the sampled prefix, so the threshold is that row's own key, no row lies strictly above it and the sweep
    lists fewer than k rows.  The reference always returns min(k, size) entries (src/binaryQuantizationFormat.ts:385): the call has to
    notice that its short list proves nothing and hand the query to the segmented chain."""
    rng = np.random.default_rng(1000 * k + copies)
    n, dim, qb, sim = 300_000, 768, 4, 1
    pb = dim // 8
    codes = rng.integers(0, 256, size=(n, pb), dtype=np.uint8)
    corr = np.empty((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = 1e-4 * (2 * rng.random(n) - 1)
    cdp = 0.0009
    qq = rng.integers(0, 1 << qb, size=dim, dtype=np.uint8)
    qc = np.array([-0.15, 0.148, -0.0014, float(qq.sum())])
    # the row that matches the query best: its own bit pattern (every set query bit counted), large corrections
    best = np.packbits((qq >= 8).astype(np.uint8)).astype(np.uint8)
    at = np.sort(rng.choice(4000, size=copies, replace=False))       # all inside the first 8192 rows = inside every prefix
    codes[at] = best
    corr[at, 0] = -0.06
    corr[at, 1] = 0.06
    corr[at, 2] = 1e-4
    corr[:, 3] = np.unpackbits(codes, axis=1).sum(axis=1)
    _, _, s32 = O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp)
    assert (s32 == s32[at[0]]).sum() == copies and s32.max() == s32[at[0]], "the planted rows are the strict maximum, all equal"
    oi, osc = O.heap_topk(s32, k)
    assert len(oi) == k
    for compact in (True, False):
        ix = _make_index(codes, corr, dim, cdp, compact)
        try:
            for fused, presample in ((1, 1), (1, 0), (0, 0)):
                ix.set_option("latency_fused", fused)
                ix.set_option("latency_presample", presample)
                fi, fs = ix.search(qq, qc, qb, sim, k)
                assert len(fi) == k, "fused %d presample %d compact %s: %d results for k = %d" % (fused, presample, compact, len(fi), k)
                np.testing.assert_array_equal(fi, oi, err_msg="fused %d presample %d compact %s" % (fused, presample, compact))
                np.testing.assert_array_equal(fs.view(np.uint32), osc.view(np.uint32))
        finally:
            ix.close()


@pytest.mark.parametrize("sim,qb,nq", [(1, 4, 200), (0, 4, 70), (2, 1, 130), (1, 8, 65), (1, 4, 5)])
def test_search_raw_batch_equals_quantize_then_search(sim, qb, nq):
    """bbq_search_raw_batch (quantization on host threads pipelined with the sweeps) == bbq_quantize_queries + bbq_search_batch, bit for
    bit, quantized queries included; a query the quantizer refuses is reported with its index; k = 0 and the multi-device handle work"""
    rng = np.random.default_rng(sim * 100 + qb)
    n, dim, k = 40_000, 96, 30
    base = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    ix, codes, corr, cen = B.Index.build(base, sim)
    try:
        ix.set_option("batch_queries", 32)        # several sub-batches: the later ones are enqueued while their queries are still being quantized
        qq, qc = B.quantize_queries(queries, cen, sim, qb)
        want = ix.search_batch(qq, qc, qb, sim, k)
        for threads in (1, 7):
            idx, sc, cnt, rq, rc = ix.search_raw_batch(queries, cen, sim, qb, k, n_threads=threads, want_quantized=True)
            np.testing.assert_array_equal(rq, qq)
            np.testing.assert_array_equal(rc.view(np.uint64), qc.view(np.uint64))
            np.testing.assert_array_equal(idx, want[0])
            np.testing.assert_array_equal(sc.view(np.uint32), want[1].view(np.uint32))
            np.testing.assert_array_equal(cnt, want[2])
        _, _, s32 = O.score_all(codes, corr, dim, qq[0], qc[0], qb, sim, B.centroid_dp(cen))
        np.testing.assert_array_equal(want[0][0], O.heap_topk(s32, k)[0])
        idx0, _, cnt0 = ix.search_raw_batch(queries, cen, sim, qb, 0)
        assert idx0.shape == (nq, 0) and (cnt0 == 0).all()
        bad = queries.copy()
        bad[nq - 2, 5] = np.nan
        with pytest.raises(B.BBQError) as e:
            ix.search_raw_batch(bad, cen, sim, qb, k, n_threads=3)
        assert e.value.code == B.capi.ERR_NAN_INPUT
        # the index is usable afterwards
        idx, sc, cnt = ix.search_raw_batch(queries, cen, sim, qb, k)
        np.testing.assert_array_equal(idx, want[0])
    finally:
        ix.close()
    mx = B.Index.create_multi(codes, corr, dim, B.centroid_dp(cen), [0, 0], pilot_rows=1024)
    try:
        idx, sc, cnt = mx.search_raw_batch(queries, cen, sim, qb, k)
        np.testing.assert_array_equal(idx, want[0])
        np.testing.assert_array_equal(sc.view(np.uint32), want[1].view(np.uint32))
    finally:
        mx.close()


@pytest.mark.parametrize("dim,qb,sim,compact,ib", [(768, 4, 1, True, 1), (1024, 1, 2, False, 1), (1536, 4, 0, True, 1), (200, 4, 1, True, 1),
                                                   (1024, 8, 1, True, 2), (256, 4, 1, True, 4)])
def test_cache_resident_chunks_change_no_answer(dim, qb, sim, compact, ib):
    """the sweeps load a prefix of the index with the default cache policy (it stays in the Infinity Cache from one query's sweep to the
    next) and stream the rest (option resident_mb per sweep launch, IndexView::resident_share / resident_tiles): every split - nothing,
    a few chunks, a part, everything, the automatic share; spread over the launch's range or at its head - gives the oracle's answer for batches, single queries and the shared sweep; bbq_stats.resident_bytes
    reports the split"""
    rng = np.random.default_rng(dim + qb + sim)
    n, k, nq = 70_000, 50, 9
    if ib == 1:
        pb = (dim + 7) // 8
        codes = rng.integers(0, 256, size=(n, pb), dtype=np.uint8)
        if dim % 8:
            codes[:, -1] &= (0xFF << (8 - dim % 8)) & 0xFF
        x1 = np.unpackbits(codes, axis=1).sum(axis=1)
    else:
        codes = rng.integers(0, 1 << ib, size=(n, dim), dtype=np.uint8)
        x1 = codes.sum(axis=1)
    corr = np.empty((n, 4))
    corr[:, 0] = -0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 1] = 0.04 * (0.9 + 0.2 * rng.random(n))
    corr[:, 2] = 1e-4 * (2 * rng.random(n) - 1)
    corr[:, 3] = x1
    cdp = 0.0009
    qq = rng.integers(0, 1 << qb, size=(nq, dim), dtype=np.uint8)
    qc = np.empty((nq, 4))
    qc[:, 0] = -0.15 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 1] = 0.148 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 2] = -0.0028 * rng.random(nq)
    qc[:, 3] = qq.sum(axis=1)
    want = []
    for q in range(nq):
        if ib == 1:
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
        elif qb in (1, 4):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp, ib=ib)
        else:   # the reference throws for this queryBits on a multi-bit index: libbbq's documented extension
            _, _, s32 = O.score_all_multibit_ext(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
        want.append(O.heap_topk(s32, k))
    ix = _make_index(codes, corr, dim, cdp, compact, index_bits=ib)
    try:
        row_bytes = ix.bytes_per_row
        index_bytes = (n + 511) // 512 * 512 * row_bytes     # in whole chunks of 512 rows, as the library counts
        seen = set()
        for mb, spread in ((0, 1), (1, 1), (1, 0), (3, 1), (3, 0), (1 << 20, 1), (-1, 0), (-1, 1)):
            ix.set_option("resident_mb", mb)
            ix.set_option("resident_interleave", spread)    # the resident chunks spread over a launch's range / at its head
            idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
            for q in range(nq):
                np.testing.assert_array_equal(idx[q], want[q][0])
                np.testing.assert_array_equal(sc[q].view(np.uint32), want[q][1].view(np.uint32))
            rb = ix.stats()["resident_bytes"]
            seen.add(rb)
            assert 0 <= rb <= index_bytes and rb % (512 * row_bytes) == 0     # whole chunks of 512 rows
            if mb == 0:
                assert rb == 0
            if mb in (1, 3):
                assert 0 < rb <= 8 * (mb << 20)                               # per launch: a sweep has a handful of launches
            si, ss = ix.search(qq[0], qc[0], qb, sim, k)        # the single-query paths
            np.testing.assert_array_equal(si, want[0][0])
            np.testing.assert_array_equal(ss.view(np.uint32), want[0][1].view(np.uint32))
        assert len(seen) >= 3
        assert ix.stats()["resident_bytes"] > 0                  # the automatic share of a small index: all of its whole chunks
        if ib == 1 and qb == 4 and dim % 128 == 0:
            ix.set_option("resident_mb", 2)
            ix.set_option("sweep_share", 32)
            idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
            for q in range(nq):
                np.testing.assert_array_equal(idx[q], want[q][0])
            ix.set_option("sweep_share", 1)
    finally:
        ix.close()
