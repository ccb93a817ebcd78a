"""GPU parity of multi-bit indexes (indexBits > 1; BASELINE config 5 = queryBits 8 / indexBits 2 at 1024-d): the integer dot product
against the reference's computeQuantizedDotProduct fixtures, the scores and top-k against what the reference RETURNS for such
an index (per-row fallback, src/binaryQuantizedScorer.ts:403-419) where it returns anything (queryBits 1 and 4), and against
the oracle's restatement of the same formula where the reference throws (every other queryBits: "parity unpinned" beyond the
integer dot product)."""
import os

import numpy as np
import pytest

import orclib as O
from bbqlib import bbq_amd as B

pytestmark = pytest.mark.gpu


def canon64(a):
    a = np.array(a, np.float64)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def canon32(a):
    a = np.array(a, np.float32)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint32)


def oracle_scores(codes, corr, dim, qq, qc, qb, sim, cdp):
    """what the library documents for ANY queryBits on a multi-bit index: the reference's per-row formulas (1-bit form for
    queryBits 1, 4-bit form with centroidDP = 0 otherwise) over computeQuantizedDotProduct"""
    return O.score_all_multibit_ext(codes, corr, dim, qq, qc, qb, sim, cdp)


@pytest.mark.parametrize("name", O.golden_names("intdot_*"))
def test_integer_dot_fixtures_through_score_rows(name):
    """bitDotProduct of bbq_score_rows == computeQuantizedDotProduct(query, row) of the reference, for queryBits 4 and 8 on
    2-, 4- and 8-bit indexes (incl. 1024-d queryBits 8 / indexBits 2: BASELINE config 5's shape)"""
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    n, dim, ib = g["n"], g["dim"], g["ib"]
    codes = O.dec(g["codes_unpacked_u8"], "u1").reshape(n, dim)
    corr = O.dec(g["corr_f64"], "<f8").reshape(n, 4)
    cen = O.dec(g["centroid_f32"], "<f4")
    cdp = B.centroid_dp(cen)
    queries = O.mulberry32(g["gen"]["query_seed"], g["nq"] * dim).reshape(g["nq"], dim)
    for compact in (True, False):
        ix = B.Index(codes, corr, dim, cdp, index_bits=ib, corrections="compact" if compact else "inline")
        try:
            for qi, rec in enumerate(g["queries"]):
                qq, qc = B.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
                np.testing.assert_array_equal(qq, O.dec(rec["qquant_u8"], "u1"))
                d, s64, s32 = ix.score_rows(qq, qc, g["qb"], sim)
                np.testing.assert_array_equal(d, O.dec(rec["qcdist_i32"], "<i4"))
                od, os64, os32 = oracle_scores(codes, corr, dim, qq, qc, g["qb"], sim, cdp)
                np.testing.assert_array_equal(canon64(s64), canon64(os64))
                idx, sc = ix.search(qq, qc, g["qb"], sim, 10)
                oi, osc = O.heap_topk(os32, 10)
                np.testing.assert_array_equal(idx, oi)
                np.testing.assert_array_equal(canon32(sc), canon32(osc))
            c2, r2 = ix.export()
            np.testing.assert_array_equal(c2, codes)
            np.testing.assert_array_equal(canon64(r2), canon64(corr))
        finally:
            ix.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("BBQ_FUZZ_MULTIBIT_SEEDS", "16"))))   # BBQ_FUZZ_MULTIBIT_SEEDS=300 for a soak
def test_multibit_randomized_vs_oracle(seed):
    """random (n, dim, indexBits, queryBits, similarity, layout, segment plan): index bytes from the product's quantizer equal the
    oracle's, integers / f64 / f32 scores and the replayed top-k equal the oracle's multi-bit path"""
    rng = np.random.default_rng(7000 + seed)
    dim = int(rng.choice([2, 7, 16, 17, 63, 64, 65, 100, 128, 384, 768, 1024]))
    n = int(rng.choice([1, 63, 64, 65, 513, 1500, 4097, 9000]))
    ib = int(rng.choice([2, 2, 2, 3, 4, 5, 8]))
    qb = int(rng.choice([1, 4, 4, 8, 8, 2, 6]))
    sim = int(rng.integers(0, 3))
    k = int(rng.choice([1, 10, 100, n, n + 2]))
    flavour = int(rng.integers(0, 3))
    if flavour == 0:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    elif flavour == 1:
        pool = rng.standard_normal((max(2, min(12, n)), dim)).astype(np.float32)
        base = pool[rng.integers(0, pool.shape[0], n)]
    else:
        base = (rng.standard_normal((n, dim)) * 10.0 ** rng.integers(-4, 5, (n, 1))).astype(np.float32)
    nq = int(rng.integers(1, 12))
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    ocodes, ocorr, ocen = O.build_index(base, sim, ib=ib)
    codes, corr, cen = B.quantize_vectors(base, sim, ib)
    np.testing.assert_array_equal(codes, ocodes)
    np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
    cdp = B.centroid_dp(cen)
    ix = B.Index(codes, corr, dim, cdp, index_bits=ib, corrections=int(rng.integers(0, 2)))
    try:
        ix.set_option("first_segment_rows", int(rng.choice([1024, 4096])))
        ix.set_option("segment_growth", int(rng.choice([2, 8])))
        ix.set_option("batch_queries", int(rng.choice([1, 5, 32])))
        ix.set_option("sweep_share", int(rng.choice([1, 8, 32])))      # shared sweeps do not exist for multi-bit rows: must fall back quietly
        qq, qc = B.quantize_queries(queries, cen, sim, qb)
        idx, sc, cnt = ix.search_batch(qq, qc, qb, sim, k)
        for i in range(nq):
            if qb in (1, 4):
                d, s64, s32 = O.score_all(codes, corr, dim, qq[i], qc[i], qb, sim, cdp, ib)     # the pinned restatement
            else:
                d, s64, s32 = oracle_scores(codes, corr, dim, qq[i], qc[i], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(idx[i, :cnt[i]], oi)
            np.testing.assert_array_equal(canon32(sc[i, :cnt[i]]), canon32(osc))
            if i == 0:
                gd, g64, g32 = ix.score_rows(qq[0], qc[0], qb, sim)
                np.testing.assert_array_equal(gd, d)
                np.testing.assert_array_equal(canon64(g64), canon64(s64))
                np.testing.assert_array_equal(canon32(g32), canon32(s32))
    finally:
        ix.close()


def test_multibit_explicit_component_sums_and_persistence(tmp_path):
    """quantizedComponentSum that is not the sum of the codes is honoured (explicit sums in the tile records); save / load /
    export round-trip a 2-bit index"""
    rng = np.random.default_rng(3)
    n, dim, ib, sim = 3000, 200, 2, 1
    base = rng.standard_normal((n, dim)).astype(np.float32)
    codes, corr, cen = B.quantize_vectors(base, sim, ib)
    cdp = B.centroid_dp(cen)
    q = rng.standard_normal(dim).astype(np.float32)
    qq, qc = B.quantize_query(q, cen, sim, 4)
    ix = B.Index(codes, corr, dim, cdp, index_bits=ib)
    assert ix.bytes_per_row == 64 + 4           # 200 dims x 2 bits = 50 B -> 64, + 4 B compact corrections
    ix.save(str(tmp_path / "mb"), cen, sim)
    ix2, cen2, info = B.Index.load(str(tmp_path / "mb"))
    assert ix2.index_bits == 2 and ix2.bytes_per_row == ix.bytes_per_row
    a, b = ix.search(qq, qc, 4, sim, 50), ix2.search(qq, qc, 4, sim, 50)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(canon32(a[1]), canon32(b[1]))
    c2, r2 = ix2.export()
    np.testing.assert_array_equal(c2, codes)
    np.testing.assert_array_equal(canon64(r2), canon64(corr))
    ix.close()
    ix2.close()
    corr3 = corr.copy()
    corr3[::5, 3] += 3.0
    ix3 = B.Index(codes, corr3, dim, cdp, index_bits=ib)
    assert ix3.bytes_per_row == 64 + 32
    d, s64, s32 = ix3.score_rows(qq, qc, 4, sim)
    od, os64, os32 = O.score_all(codes, corr3, dim, qq, qc, 4, sim, cdp, ib)
    np.testing.assert_array_equal(d, od)
    np.testing.assert_array_equal(canon64(s64), canon64(os64))
    idx, sc = ix3.search(qq, qc, 4, sim, 30)
    oi, osc = O.heap_topk(os32, 30)
    np.testing.assert_array_equal(idx, oi)
    ix3.close()


@pytest.mark.parametrize("shards,pilot", [(3, 1024), (2, 0)])
def test_multibit_sharded_scan(shards, pilot):
    """row shards of a 2-bit index (pilot replica, packed lists, shard-ordered replay) == the reference's global top-k"""
    import torch
    g = O.load_golden("ib2_big_20000x128_euc")
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    codes, corr, cen = B.quantize_vectors(base, sim, 2, g["lambda"], g["iters"])
    assert O.sha(codes) == g["codes_sha256"]
    n, dim, k = g["n"], g["dim"], 100
    cdp = B.centroid_dp(cen)
    qq, qc = B.quantize_queries(queries, cen, sim, g["qb"])
    nq = len(queries)
    per = (n + shards - 1) // shards
    packed, offsets = [], []
    for r in range(shards):
        r0, r1 = r * per, min((r + 1) * per, n)
        P = min(pilot, r0) // 1024 * 1024 if r > 0 else 0
        ix = B.Index(codes[r0:r1], corr[r0:r1], dim, cdp, index_bits=2, row_base=r0,
                     pilot_codes=codes[:P] if P else None, pilot_corr=corr[:P] if P else None)
        ix.set_option("first_segment_rows", 1024)
        ix.set_option("segment_growth", 2)
        cap = int(ix.shard_list_cap(k)) * nq
        d_packed = torch.zeros(cap, dtype=torch.int64, device="cuda")
        d_off = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
        d_flags = torch.zeros(nq, dtype=torch.int32, device="cuda")
        total = ix.shard_scan(qq, qc, g["qb"], sim, k, d_packed.data_ptr(), cap, d_off.data_ptr(), d_flags.data_ptr())
        assert int(d_flags.abs().sum().item()) == 0
        packed.append(d_packed[:total].cpu().numpy().view(np.uint64))
        offsets.append(d_off.cpu().numpy())
        ix.close()
    idx, sc, cnt = B.replay_batch(packed, offsets, nq, n, k, n_threads=2)
    for qi in range(nq):
        tk = [t for t in g["queries"][qi]["topk"] if t["k"] == k][0]
        np.testing.assert_array_equal(idx[qi, :cnt[qi]], O.dec(tk["idx_i32"], "<i4"))
        np.testing.assert_array_equal(canon32(sc[qi, :cnt[qi]]), canon32(O.dec(tk["score_f32"], "<f4")))


def test_config5_shape_1m_x_1024_qb8_ib2_properties():
    """BASELINE config 5's shape at full size (1 M x 1024-d, queryBits 8 / indexBits 2, k = 100): the oracle's formula for one
    query (integer dot pinned, float score unpinned - the reference throws), size-independent properties for more"""
    import bench
    n, dim, k = 1_000_000, 1024, 100
    codes, corr = bench.synth_rows_multibit(1, 0, n, dim, 2)
    qq, qc = bench.synth_queries(2, 4, dim, 8)
    cdp = 0.0009110655808639536
    ix = B.Index(codes, corr, dim, cdp, index_bits=2)
    try:
        assert ix.bytes_per_row == 256 + 4
        idx, sc, cnt = ix.search_batch(qq, qc, 8, 1, k)
        assert (cnt == k).all() and ix.stats()["dense_fallbacks"] == 0
        # (1) numpy restatement of the integer dot + the oracle's per-row formula, one query
        d = (codes.astype(np.int32) @ qq[0].astype(np.int32)).astype(np.int32)
        gd, g64, g32 = ix.score_rows(qq[0], qc[0], 8, 1)
        np.testing.assert_array_equal(gd, d)
        L = O.lib()
        pick = np.concatenate([np.arange(0, n, 997), idx[0]])
        for r in pick:
            want = L.orc_score_single_row(int(d[r]), O.f64p(qc[0]), O.f64p(corr[r]), dim, 0.0, 1, 0)
            assert np.float64(want).view(np.uint64) == g64[r].view(np.uint64)
        oi, osc = O.heap_topk(g32, k)
        np.testing.assert_array_equal(idx[0], oi)
        np.testing.assert_array_equal(canon32(sc[0]), canon32(osc))
        # (2) dense replay == sparse segments; returned scores descending and equal to the rows' own scores
        ix.set_option("force_dense", 1)
        di, ds, _ = ix.search_batch(qq[:2], qc[:2], 8, 1, k)
        ix.set_option("force_dense", 0)
        np.testing.assert_array_equal(di, idx[:2])
        np.testing.assert_array_equal(canon32(ds), canon32(sc[:2]))
        for q in range(1, 4):
            assert (np.diff(sc[q]) <= 0).all()
            r = int(idx[q, 37])
            _, _, one = ix.score_rows(qq[q], qc[q], 8, 1, r, 1)
            assert one[0].view(np.uint32) == sc[q, 37].view(np.uint32)
    finally:
        ix.close()


MB_CASES = [n for n in O.golden_names("ib*") if "throws" not in n]


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("name", MB_CASES)
def test_device_build_multibit_matches_reference(name, compact):
    """bbq_index_build_bits: quantizeVectors for indexBits > 1 as HIP kernels - centroid, the one-byte-per-dimension codes and the f64
    corrections bit-exact vs the golden vectors, and the index it leaves on the device answers like the reference"""
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    ix, codes, corr, cen = B.Index.build(base, sim, g["lambda"], g["iters"], index_bits=g["ib"], corrections="compact" if compact else "inline")
    try:
        assert ix.index_bits == g["ib"]
        assert O.sha(cen) == O.sha(O.dec(g["centroid_f32"], "<f4")), "centroid"
        assert O.sha(codes) == g["codes_sha256"], "codes"
        ocodes, ocorr, ocen = O.build_index(base, sim, g["lambda"], g["iters"], g["ib"])
        np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
        for qi, rec in enumerate(g["queries"]):
            qq, qc = B.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
            for tk in rec["topk"]:
                idx, sc = ix.search(qq, qc, g["qb"], sim, tk["k"])
                np.testing.assert_array_equal(idx, O.dec(tk["idx_i32"], "<i4"))
                np.testing.assert_array_equal(canon32(sc), canon32(O.dec(tk["score_f32"], "<f4")))
        c2, r2 = ix.export()
        np.testing.assert_array_equal(c2, codes)
    finally:
        ix.close()


def test_device_build_multibit_odd_shapes_and_degenerate_rows():
    rng = np.random.default_rng(44)
    for n, dim, sim, ib in ((1, 1, 0, 2), (65, 3, 1, 3), (130, 13, 2, 8), (999, 131, 1, 2), (64, 129, 0, 4), (300, 1024, 1, 2), (257, 100, 2, 5)):
        base = rng.standard_normal((n, dim)).astype(np.float32)
        if n > 10:
            base[3] = 0          # zero row: degenerate interval
            base[7] = 2.5        # constant row
            base[9] *= 1e20
        ix, codes, corr, cen = B.Index.build(base, sim, index_bits=ib)
        ocodes, ocorr, ocen = O.build_index(base, sim, ib=ib)
        np.testing.assert_array_equal(codes, ocodes)
        np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
        np.testing.assert_array_equal(cen.view(np.uint32), ocen.view(np.uint32))
        hcodes, hcorr, hcen = B.quantize_vectors(base, sim, ib)     # the host quantizer agrees too
        np.testing.assert_array_equal(hcodes, codes)
        np.testing.assert_array_equal(canon64(hcorr), canon64(corr))
        ix.close()
    v = np.ones((300, 7), np.float32)
    v[200, 5] = np.nan
    with pytest.raises(B.BBQError) as e:
        B.Index.build(v, 0, index_bits=2)
    assert e.value.code == 8 and "向量 200 位置 5 包含NaN值" in str(e.value)
    with pytest.raises(B.BBQError):
        B.Index.build(np.ones((3, 7), np.float32), 0, index_bits=9)
