"""Pins the CPU oracle (oracle/bbq_oracle.c) bit-for-bit against golden vectors produced by RUNNING the
reference (type-erased TypeScript under Node 12, oracle/tools/gen_fixtures.js).  CPU only."""
import os

import numpy as np
import pytest

import orclib as O



def b64(a):
    """f64 bit patterns with every NaN canonicalised (JS has a single NaN value; its sign/payload in a
    Float64Array is implementation-defined, so only NaN-ness is pinned)."""
    a = np.array(a, np.float64)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


def b32(a):
    a = np.array(a, np.float32)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint32)


def same_or_sha(arr, sha, full_b64, dt):
    """sha256 pin, except when NaNs are present (then compare canonicalised bits of the full array)"""
    if np.issubdtype(arr.dtype, np.floating) and np.isnan(arr).any():
        assert full_b64 is not None, "NaN in a hashes-only fixture"
        ref = O.dec(full_b64, dt)
        (np.testing.assert_array_equal)(b64(arr) if arr.dtype == np.float64 else b32(arr),
                                        b64(ref) if arr.dtype == np.float64 else b32(ref))
    else:
        assert O.sha(arr) == sha


CASES = [n for n in O.golden_names() if not n.startswith(("intdot_", "api_", "rerank_"))]
SMALL = [n for n in CASES if "big_" not in n]


def _check_case(name):
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    dim = g["dim"]
    ib = g["ib"]
    codes, corr, cen = O.build_index(base, sim, g["lambda"], g["iters"], ib)
    # index build: centroid, codes, corrections bit-exact
    assert O.sha(cen) == O.sha(O.dec(g["centroid_f32"], "<f4")), "centroid"
    pb = g["row_bytes"]
    assert codes.shape[1] == pb
    keep = min(g["n"], 4)
    np.testing.assert_array_equal(codes[:keep].ravel(), O.dec(g["head_codes_u8"], "u1"))
    np.testing.assert_array_equal(b64(corr[:keep].ravel()), b64(O.dec(g["head_corr_f64"], "<f8")))
    assert O.sha(codes) == g["codes_sha256"], "codes"
    same_or_sha(corr.ravel(), g["corr_sha256"], g.get("corr_f64"), "<f8")
    cdp = O.centroid_dp(cen)
    assert np.float64(cdp).view(np.uint64) == O.dec(g["centroid_dp_f64"], "<f8").view(np.uint64)[0]
    for qi, rec in enumerate(g["queries"]):
        qq, qc = O.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
        np.testing.assert_array_equal(qq, O.dec(rec["qquant_u8"], "u1"))
        np.testing.assert_array_equal(b64(qc), b64(O.dec(rec["qcorr_f64"], "<f8")))
        if "per_row_error" in rec:      # the reference throws here (indexBits > 1 with a queryBits its fallback does not know)
            assert ib > 1 and g["qb"] not in (1, 4) and "不支持的查询位数" in rec["per_row_error"]
            with pytest.raises(O.ReferenceThrows):
                O.score_all(codes, corr, dim, qq, qc, g["qb"], sim, cdp, ib)
            for tk in rec["topk"]:
                assert tk["error"] == rec["per_row_error"]
                assert O.search(queries[qi], codes, corr, cen, sim, g["qb"], tk["k"], g["lambda"], g["iters"], ib)[0] == -5
            continue
        d, s64, s32 = O.score_all(codes, corr, dim, qq, qc, g["qb"], sim, cdp, ib)
        assert O.sha(d) == rec["qcdist_sha256"], "integer qcDist"
        same_or_sha(s64, rec["score_sha256"], rec.get("score_f64"), "<f8")
        if not np.isnan(s32).any():
            assert O.sha(s32) == rec["score_f32_sha256"], "f32 scores"
        if g["full"]:
            np.testing.assert_array_equal(d, O.dec(rec["qcdist_i32"], "<i4"))
            np.testing.assert_array_equal(b64(s64), b64(O.dec(rec["score_f64"], "<f8")))
        for tk in rec["topk"]:
            assert "error" not in tk
            idx, sc = O.search(queries[qi], codes, corr, cen, sim, g["qb"], tk["k"], g["lambda"], g["iters"], ib)
            np.testing.assert_array_equal(idx, O.dec(tk["idx_i32"], "<i4"), err_msg="top-k indices k=%d" % tk["k"])
            np.testing.assert_array_equal(b32(sc), b32(O.dec(tk["score_f32"], "<f4")))
            # and the heap alone, fed the f32 scores
            idx2, sc2 = O.heap_topk(s32, tk["k"])
            np.testing.assert_array_equal(idx2, idx)
        if "oversample" in rec:
            m = min(g["k"], g["n"])
            out = np.zeros(m + 1, np.int32)
            cnt = O.lib().orc_oversampled_topk(O.f32p(queries[qi]), O.f32p(base), O.u8p(codes), O.f64p(corr), O.f32p(cen),
                                               g["n"], dim, sim, g["qb"], g["lambda"], g["iters"], g["k"],
                                               rec["oversample"]["factor"], O.i32p(out))
            assert list(out[:cnt]) == rec["oversample"]["idx"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_reference_small(name):
    _check_case(name)


@pytest.mark.parametrize("name", [n for n in CASES if "big_" in n])
def test_oracle_matches_reference_big(name):
    _check_case(name)


def test_ties_fixtures_really_have_ties():
    """the tie-stress fixtures must exercise equal f32 scores inside / at the edge of the top-k"""
    g = O.load_golden("ties_cos_qb4")
    tk = [t for t in g["queries"][0]["topk"] if t["k"] == 100][0]
    sc = O.dec(tk["score_f32"], "<f4")
    assert len(np.unique(sc)) < len(sc)


@pytest.mark.parametrize("name", O.golden_names("intdot_*"))
def test_integer_dot_multibit_index(name):
    """ib >= 2: the reference's float score is unpinned (search throws / falls back);
    the integer dot product is pinned against computeQuantizedDotProduct (src/bitwiseDotProduct.ts:14-30)."""
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    n, dim = g["n"], g["dim"]
    base = O.mulberry32(g["gen"]["base_seed"], n * dim).reshape(n, dim)
    queries = O.mulberry32(g["gen"]["query_seed"], g["nq"] * dim).reshape(g["nq"], dim)
    codes, corr, cen = O.build_index_unpacked(base, sim, g["ib"], g["lambda"], g["iters"])
    np.testing.assert_array_equal(codes.ravel(), O.dec(g["codes_unpacked_u8"], "u1"))
    np.testing.assert_array_equal(corr.ravel().view(np.uint64), O.dec(g["corr_f64"], "<f8").view(np.uint64))
    for qi, rec in enumerate(g["queries"]):
        qq, _ = O.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
        np.testing.assert_array_equal(qq, O.dec(rec["qquant_u8"], "u1"))
        d = np.array([O.lib().orc_dot_u8(O.u8p(qq), O.u8p(codes[i]), dim) for i in range(n)], np.int32)
        np.testing.assert_array_equal(d, O.dec(rec["qcdist_i32"], "<i4"))


def test_known_answers_from_reference_unit_tests():
    """rust-wasm known answers quoted in SURVEY section 4 / App. C (also true of the TS path)."""
    L = O.lib()
    q = np.arange(1, 9, dtype=np.uint8)
    assert L.orc_qcdist_unpacked_query(O.u8p(q), O.u8p(np.array([0xFF], np.uint8)), 8) == 36
    assert L.orc_qcdist_unpacked_query(O.u8p(q), O.u8p(np.array([0x00], np.uint8)), 8) == 0
    packed = np.zeros(1, np.uint8)
    assert L.orc_pack_binary(O.u8p(np.array([1, 0, 1, 0, 1, 0, 1, 0], np.uint8)), 8, O.u8p(packed)) == 0
    assert packed[0] == 0b10101010
    qp = np.array([0xFF], np.uint8)
    got = [L.orc_qcdist_packed_query(O.u8p(qp), O.u8p(np.array([b], np.uint8)), 1) for b in (0xFF, 0x00, 0xF0)]
    assert got == [8, 0, 4]
    assert L.orc_dot_u8(O.u8p(np.array([1, 2, 3, 4], np.uint8)), O.u8p(np.array([5, 6, 7, 8], np.uint8)), 4) == 70
    assert L.orc_dot_u8(O.u8p(np.array([15, 14, 13, 12], np.uint8)), O.u8p(np.array([1, 1, 0, 1], np.uint8)), 4) == 41
    assert L.orc_pack_binary(O.u8p(np.array([2], np.uint8)), 1, O.u8p(packed)) == -1
    # tests/computeCentroid-correctness.test.ts:64-83
    cen = np.zeros(3, np.float32)
    base = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32)
    L.orc_centroid(O.f32p(base), 3, 3, O.f32p(cen))
    assert list(cen) == [4, 5, 6]


def test_search_argument_errors():
    g = O.load_golden("edge_n1")
    base, queries = O.golden_inputs(g)
    codes, corr, cen = O.build_index(base, 0)
    assert O.search(queries[0], codes, corr, cen, 0, 4, -1)[0] == -3
    assert O.search(queries[0][:4].copy(), codes, corr, cen, 0, 4, 3)[0] == -4
    idx, sc = O.search(queries[0], codes, corr, cen, 0, 4, 0)
    assert len(idx) == 0


def test_reference_recall_thresholds_closed_form():
    """tests/recall.test.ts:88-165,387-508 thresholds on the reference's closed-form dataset (recall@10 vs exact cosine)."""
    for name, thr in (("closed_100x128_qb1", 0.70), ("closed_100x128_qb4", 0.60)):
        g = O.load_golden(name)
        base, queries = O.golden_inputs(g)
        bn = base / np.linalg.norm(base, axis=1, keepdims=True)
        rec = []
        for qi in range(g["nq"]):
            q = queries[qi] / np.linalg.norm(queries[qi])
            truth = set(np.argsort(-(bn @ q), kind="stable")[:10].tolist())
            got = set(O.dec(g["queries"][qi]["topk"][0]["idx_i32"], "<i4").tolist())
            rec.append(len(truth & got) / 10.0)
        assert np.mean(rec) >= thr


# ---------------------------------------------------------------- exact rerank

@pytest.mark.parametrize("name", O.golden_names("rerank_*"))
def test_true_similarity_and_rerank_selectors(name):
    """computeSimilarity (all three functions, every (query,row)) and both oversample selectors of the reference"""
    g = O.load_golden(name)
    n, dim, nq, k = g["n"], g["dim"], g["nq"], g["k"]
    base = O.dec(g["base_f32"], np.float32).reshape(n, dim)
    queries = O.dec(g["queries_f32"], np.float32).reshape(nq, dim)
    true = {}
    for sim_name, sim in O.SIMS.items():
        true[sim] = O.true_similarity(queries, base, sim)
        np.testing.assert_array_equal(b64(true[sim].ravel()), b64(O.dec(g["true_f64"][sim_name], np.float64)))
    codes, corr, cen = O.build_index(base, 1, g["lambda"], g["iters"])
    for rec in g["oversample"]:
        qi, f = rec["query"], rec["factor"]
        cand, qsc = O.search(queries[qi], codes, corr, cen, 1, 4, k * f, g["lambda"], g["iters"])
        t = true[1][qi][cand]
        for how in ("heap", "sort"):
            pos = O.rerank_select(t, k, how)
            np.testing.assert_array_equal(cand[pos], O.dec(rec[how]["idx_i32"], np.int32))
            np.testing.assert_array_equal(b32(qsc[pos]), b32(O.dec(rec[how]["quantized_f32"], np.float32)))
            np.testing.assert_array_equal(b64(t[pos]), b64(O.dec(rec[how]["true_f64"], np.float64)))


def test_cpu_baseline_js_crosscheck_fixture():
    """the JS restatement that bench.py times as cpu_baseline_js was run next to the type-erased reference on the same host
    (oracle/tools/crosscheck_js_baseline.js, build container only): their per-row costs lie in the same band"""
    import json
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_cpu_baseline_crosscheck.json")))
    assert len(d["runs"]) >= 2
    ref = [x for r in d["runs"] for x in r["reference_passes_us_per_row"]]
    mine = [x for r in d["runs"] for x in r["restatement_passes_us_per_row"]]
    assert all(r["rows"] == 50000 and r["dim"] == 768 and r["k"] == 100 for r in d["runs"])
    # the bands overlap and the medians over all passes agree within 35 % (both programs are bimodal under V8: see the fixture's note)
    assert max(min(ref), min(mine)) <= min(max(ref), max(mine))
    assert 0.65 <= sorted(mine)[len(mine) // 2] / sorted(ref)[len(ref) // 2] <= 1.35
