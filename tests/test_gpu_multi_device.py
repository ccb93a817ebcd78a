"""GPU: one index row-sharded over several devices behind ONE handle (bbq_index_create_multi) must answer exactly like the
single-device index - checked against the reference's fixtures and the oracle.  A single-GPU box maps every shard onto device 0
(the code path is the same: one worker thread, one shard index, one pilot replica, one packed list per shard)."""
import numpy as np
import pytest

import orclib as O
from bbqlib import bbq_amd as B

pytestmark = pytest.mark.gpu


def canon32(a):
    a = np.array(a, np.float32)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint32)


def canon64(a):
    a = np.array(a, np.float64)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def _case(name):
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    codes, corr, cen = B.quantize_vectors(base, sim, g["ib"], g["lambda"], g["iters"])
    assert O.sha(codes) == g["codes_sha256"]
    return g, sim, base, queries, codes, corr, cen, B.centroid_dp(cen)


@pytest.mark.parametrize("name,shards,pilot", [("ties_cos_qb4", 3, 1024), ("big_20000x128_cos", 4, 2048), ("ties_euc_qb4", 2, 0),
                                               ("big_50000x768_cos", 5, 4096), ("c1_1000x128_cos_qb4", 2, 512), ("m_100d_max_qb1", 8, 32768),
                                               ("ib2_big_20000x128_euc", 3, 1024), ("big_20000x1024_cos", 2, 1024), ("edge_n1", 4, 0),
                                               ("ties_16d_qb1", 7, 512)])
def test_multi_device_index_equals_the_reference(name, shards, pilot):
    g, sim, base, queries, codes, corr, cen, cdp = _case(name)
    ix = B.Index.create_multi(codes, corr, g["dim"], cdp, [0] * shards, index_bits=g["ib"], pilot_rows=pilot)
    try:
        assert 1 <= ix.shards <= shards
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        for qi, rec in enumerate(g["queries"]):
            qq, qc = B.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
            d, s64, s32 = ix.score_rows(qq, qc, g["qb"], sim)
            assert O.sha(d) == rec["qcdist_sha256"]
            if not np.isnan(s64).any():
                assert O.sha(s64) == rec["score_sha256"] and O.sha(s32) == rec["score_f32_sha256"]
            for tk in rec["topk"]:
                idx, sc = ix.search(qq, qc, g["qb"], sim, tk["k"])
                np.testing.assert_array_equal(idx, O.dec(tk["idx_i32"], "<i4"), err_msg="%s q%d k=%d" % (name, qi, tk["k"]))
                np.testing.assert_array_equal(canon32(sc), canon32(O.dec(tk["score_f32"], "<f4")))
        c2, r2 = ix.export()
        np.testing.assert_array_equal(c2, codes)
        np.testing.assert_array_equal(canon64(r2), canon64(corr))
    finally:
        ix.close()


def test_multi_device_rounds_dense_fallbacks_and_large_k():
    """more queries than one round (double-buffered rounds), a NaN row (every shard scores densely for the queries it reaches),
    k beyond the sparse path, k > N; stats are the merged ones"""
    rng = np.random.default_rng(17)
    n, dim, nq, k, sim = 30000, 96, 41, 100, 1
    base = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    codes, corr, cen = B.quantize_vectors(base, sim)
    cdp = B.centroid_dp(cen)
    qq, qc = B.quantize_queries(queries, cen, sim, 4)
    want = []
    for i in range(nq):
        _, _, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], 4, sim, cdp)
        want.append(s32)
    ix = B.Index.create_multi(codes, corr, dim, cdp, [0, 0, 0], pilot_rows=2048)
    single = B.Index(codes, corr, dim, cdp)
    try:
        for rq in (7, 512):
            ix.set_option("round_queries", rq)
            ix.set_option("replay_threads", 3)
            idx, sc, cnt = ix.search_batch(qq, qc, 4, sim, k)
            sidx, ssc, _ = single.search_batch(qq, qc, 4, sim, k)
            np.testing.assert_array_equal(idx, sidx)
            np.testing.assert_array_equal(canon32(sc), canon32(ssc))
            for i in range(nq):
                oi, osc = O.heap_topk(want[i], k)
                np.testing.assert_array_equal(idx[i, :cnt[i]], oi)
            st = ix.stats()
            assert st["dense_fallbacks"] == 0 and st["candidates"] > nq * k and st["total_scan_launches"] > 0
        for kk in (5000, n + 10):
            idx, sc = ix.search(qq[0], qc[0], 4, sim, kk)
            oi, osc = O.heap_topk(want[0], kk)
            np.testing.assert_array_equal(idx, oi)
            np.testing.assert_array_equal(canon32(sc), canon32(osc))
        ix.set_option("force_dense", 1)
        idx, sc = ix.search(qq[1], qc[1], 4, sim, k)
        np.testing.assert_array_equal(idx, O.heap_topk(want[1], k)[0])
        ix.set_option("force_dense", 0)
    finally:
        ix.close()
        single.close()
    corr2 = corr.copy()
    corr2[25000, 2] = np.nan   # a NaN score in the last shard: flagged there, the query goes dense everywhere
    ix = B.Index.create_multi(codes, corr2, dim, cdp, [0, 0, 0], pilot_rows=2048)
    try:
        idx, sc, cnt = ix.search_batch(qq[:5], qc[:5], 4, sim, k)
        for i in range(5):
            _, _, s32 = O.score_all(codes, corr2, dim, qq[i], qc[i], 4, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[i, :cnt[i]], oi)
            np.testing.assert_array_equal(canon32(sc[i, :cnt[i]]), canon32(osc))
        assert ix.stats()["dense_fallbacks"] == 5
    finally:
        ix.close()


def test_multi_device_rerank_recipe_and_errors():
    g = O.load_golden("rerank_100d")
    n, dim, nq, k = g["n"], g["dim"], g["nq"], g["k"]
    base = O.dec(g["base_f32"], np.float32).reshape(n, dim)
    queries = O.dec(g["queries_f32"], np.float32).reshape(nq, dim)
    codes, corr, cen = B.quantize_vectors(base, 1, 1, g["lambda"], g["iters"])
    ix = B.Index.create_multi(codes, corr, dim, B.centroid_dp(cen), [0, 0], pilot_rows=512)
    dv = B.Vectors(base)
    qq, qc = B.quantize_queries(queries, cen, 1, 4, g["lambda"], g["iters"])
    try:
        for f in sorted({r["factor"] for r in g["oversample"]}):
            idx, qsc, tsc, cnt = B.search_rerank_batch(ix, dv, queries, qq, qc, 4, 1, k, f, 0, 1)
            for rec in (r for r in g["oversample"] if r["factor"] == f):
                qi, m = rec["query"], int(cnt[rec["query"]])
                np.testing.assert_array_equal(idx[qi, :m], O.dec(rec["heap"]["idx_i32"], np.int32))
                np.testing.assert_array_equal(canon64(tsc[qi, :m]), canon64(O.dec(rec["heap"]["true_f64"], np.float64)))
        with pytest.raises(B.BBQError):
            ix.save("/tmp/never", cen, 1)
        with pytest.raises(B.BBQError):
            ix.set_option("round_queries", 0)
    finally:
        dv.close()
        ix.close()
    with pytest.raises(B.BBQError):
        B.Index.create_multi(codes, corr, dim, 0.0, [0, 99])
    with pytest.raises(B.BBQError):
        B.Index.create_multi(codes, corr, dim, 0.0, [])
    with pytest.raises(B.BBQError):
        B.Vectors(base, device=99)


def test_multi_device_full_size_10m():
    """the headline size through the multi-device handle (4 shards on the one GPU): same answers as the single-device index"""
    import bench
    n, dim, k, pb = 10_000_000, 768, 100, 96
    codes, corr = bench.synth_rows(1, 0, n, pb)
    qq, qc = bench.synth_queries(2, 40, dim)
    cdp = 0.0009110655808639536
    single = B.Index(codes, corr, dim, cdp)
    want = single.search_batch(qq, qc, 4, 1, k)
    single.close()
    ix = B.Index.create_multi(codes, corr, dim, cdp, [0, 0, 0, 0])
    try:
        assert ix.shards == 4
        ix.set_option("round_queries", 16)
        got = ix.search_batch(qq, qc, 4, 1, k)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(canon32(got[1]), canon32(want[1]))
        assert ix.stats()["dense_fallbacks"] == 0
    finally:
        ix.close()
