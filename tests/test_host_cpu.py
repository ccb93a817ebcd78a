"""CPU: the product's host-side code (C++ quantizer, heap replay) against the golden vectors and the oracle."""
import numpy as np
import pytest

import orclib as O
from bbqlib import bbq_amd as B, capi

CASES = [n for n in O.golden_names() if not n.startswith(("intdot_", "api_", "rerank_", "big_5", "big_3"))]


def canon64(a):
    a = np.array(a, np.float64)
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


@pytest.mark.parametrize("name", CASES)
def test_product_quantizer_matches_reference(name):
    g = O.load_golden(name)
    sim = O.SIMS[g["sim"]]
    base, queries = O.golden_inputs(g)
    codes, corr, cen = B.quantize_vectors(base, sim, g["ib"], g["lambda"], g["iters"], n_threads=3)
    assert O.sha(codes) == g["codes_sha256"]
    ocodes, ocorr, ocen = O.build_index(base, sim, g["lambda"], g["iters"], g["ib"])
    np.testing.assert_array_equal(canon64(corr), canon64(ocorr))
    assert O.sha(cen) == O.sha(O.dec(g["centroid_f32"], "<f4"))
    for qi, rec in enumerate(g["queries"]):
        qq, qc = B.quantize_query(queries[qi], cen, sim, g["qb"], g["lambda"], g["iters"])
        np.testing.assert_array_equal(qq, O.dec(rec["qquant_u8"], "u1"))
        np.testing.assert_array_equal(canon64(qc), canon64(O.dec(rec["qcorr_f64"], "<f8")))


@pytest.mark.parametrize("name", O.golden_names("intdot_*"))
def test_product_quantizer_multibit_index(name):
    g = O.load_golden(name)
    n, dim = g["n"], g["dim"]
    base = O.mulberry32(g["gen"]["base_seed"], n * dim).reshape(n, dim)
    codes, corr, cen = B.quantize_vectors(base, O.SIMS[g["sim"]], g["ib"], g["lambda"], g["iters"])
    np.testing.assert_array_equal(codes.ravel(), O.dec(g["codes_unpacked_u8"], "u1"))
    np.testing.assert_array_equal(corr.ravel().view(np.uint64), O.dec(g["corr_f64"], "<f8").view(np.uint64))


def test_quantizer_errors_like_the_reference():
    with pytest.raises(B.BBQError) as e:
        B.quantize_vectors(np.zeros((0, 4), np.float32), 0)
    assert e.value.code == capi.ERR_EMPTY and "向量集合不能为空" in str(e.value)
    v = np.ones((3, 4), np.float32)
    v[2, 1] = np.nan
    with pytest.raises(B.BBQError) as e:
        B.quantize_vectors(v, 0)
    assert e.value.code == capi.ERR_NAN_INPUT and "向量 2 位置 1 包含NaN值" in str(e.value)
    v[2, 1] = np.inf
    v[1, 3] = -np.inf
    with pytest.raises(B.BBQError) as e:
        B.quantize_vectors(v, 2)
    assert e.value.code == capi.ERR_INF_INPUT and "向量 1 位置 3 包含Infinity值" in str(e.value)


def _entries(rows, s32):
    return (rows.astype(np.uint64) << np.uint64(32)) | s32[rows].view(np.uint32).astype(np.uint64)


def _superset(s32, k, seg_bounds):
    """rows the device would emit: everything in the first segment, then rows above the k-th largest of the prefix
    that ends where their segment starts (keys compared as the kernels do)"""
    b = s32.view(np.uint32).astype(np.int64)
    key = np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)
    keep = np.zeros(len(s32), bool)
    keep[:seg_bounds[0]] = True
    for a, e in zip(seg_bounds[:-1], seg_bounds[1:]):
        if a >= k:
            th = np.sort(key[:a])[-k]
            keep[a:e] = key[a:e] > th
        else:
            keep[a:e] = True
    return np.nonzero(keep)[0]


@pytest.mark.parametrize("seed", range(6))
def test_replay_over_candidate_superset_equals_reference_heap(seed):
    """size-independent property: replaying the heap over ANY superset (in row order) of the rows above the
    prefix thresholds reproduces the reference's top-k exactly, ties included"""
    rng = np.random.default_rng(seed)
    n, k = 20000, [1, 7, 50, 100, 300, 1000][seed]
    levels = [40, 300, 5000, 20000, 8, 100][seed]  # few distinct values => heavy ties
    s32 = (rng.integers(0, levels, n).astype(np.float32) / np.float32(levels)).astype(np.float32)
    oi, osc = O.heap_topk(s32, k)
    bounds = [min(max(1024, 4 * k), n)]
    while bounds[-1] < n:
        bounds.append(min(bounds[-1] * 4, n))
    rows = _superset(s32, k, bounds)
    assert len(rows) < n
    idx, sc = B.replay([_entries(rows, s32)], n, k)
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(sc.view(np.uint32), osc.view(np.uint32))
    # a looser superset (extra random rows) changes nothing
    extra = np.union1d(rows, rng.integers(0, n, 500))
    idx2, _ = B.replay([_entries(extra, s32)], n, k)
    np.testing.assert_array_equal(idx2, oi)
    # split over several shard lists
    cut = [0, len(rows) // 3, 2 * len(rows) // 3, len(rows)]
    idx3, _ = B.replay([_entries(rows[a:b], s32) for a, b in zip(cut[:-1], cut[1:])], n, k)
    np.testing.assert_array_equal(idx3, oi)


def test_replay_edge_cases():
    s32 = np.array([0.5, 0.25, 0.5, 0.75], np.float32)
    rows = np.arange(4)
    idx, sc = B.replay([_entries(rows, s32)], 4, 10)          # k > N
    oi, osc = O.heap_topk(s32, 10)
    np.testing.assert_array_equal(idx, oi)
    idx, sc = B.replay([_entries(rows, s32)], 4, 0)           # k == 0
    assert len(idx) == 0
    idx, sc = B.replay([], 0, 5)                              # nothing
    assert len(idx) == 0
    with pytest.raises(B.BBQError):
        B.replay([_entries(rows[::-1].copy(), s32)], 4, 2)    # not ascending
    with pytest.raises(B.BBQError) as e:
        B.replay([_entries(rows, s32)], 4, -1)
    assert e.value.code == capi.ERR_NEGATIVE_K
    # NaN scores follow the reference's comparator (a - b >= 0 is false for NaN)
    s = np.array([0.1, np.nan, 0.3, 0.2, np.nan, 0.9], np.float32)
    idx, sc = B.replay([_entries(np.arange(6), s)], 6, 3)
    oi, osc = O.heap_topk(s, 3)
    np.testing.assert_array_equal(idx, oi)


def test_vemb_header_spec(tmp_path):
    """the metadata file layout as DESIGN.md documents it, parsed by bbq_index_file_info (no device needed):
    104-byte little-endian header, centroid f32[dim], data checksum, metadata checksum (FNV-1a over 64-bit words)"""
    import struct
    from bbqlib import bbq_amd as B

    def fnv(data, h):
        data = data + b"\0" * (-len(data) % 8)
        for (w,) in struct.iter_unpack("<Q", data):
            h = ((h ^ w) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return h

    dim, n = 100, 1000
    pb, w16 = (dim + 7) // 8, ((dim + 7) // 8 + 15) // 16
    n_tiles = (n + 63) // 64
    stride = w16 * 1024 + 256                                        # compact layout: 4 B per row
    tiles, exact = n_tiles * stride, n_tiles * 64 * 32 + n_tiles * 8  # side section: exact corrections + {min, max} of add per tile
    hdr = struct.pack("<4sI4i3qd6i3q", b"BVEC", 2, 0, 0, 1, dim, 0, tiles + exact, n, 0.125, 1, 1, w16, stride, 0, 64, tiles, exact, 0)
    assert len(hdr) == 104
    cen = np.arange(dim, dtype=np.float32).tobytes()
    dsum = struct.pack("<Q", 0x1234)
    msum = fnv(dsum, fnv(cen, fnv(hdr, 0xcbf29ce484222325)))
    p = str(tmp_path / "h")
    open(p + ".vemb", "wb").write(hdr + cen + dsum + struct.pack("<Q", msum))
    assert B.file_info(p) == {"n_rows": n, "dim": dim, "sim": 1, "centroid_dp": 0.125, "row_base": 0}
    open(p + ".vemb", "wb").write(hdr + cen + dsum + struct.pack("<Q", msum ^ 1))
    with pytest.raises(B.BBQError):
        B.file_info(p)
    bad = struct.pack("<4sI4i3qd6i3q", b"BVEC", 2, 0, 0, 1, dim, 0, tiles + exact, n, 0.125, 1, 1, w16, stride + 16, 0, 64, tiles, exact, 0)
    open(p + ".vemb", "wb").write(bad + cen + dsum + struct.pack("<Q", fnv(dsum, fnv(cen, fnv(bad, 0xcbf29ce484222325)))))
    with pytest.raises(B.BBQError):
        B.file_info(p)
    with pytest.raises(B.BBQError):
        B.file_info(str(tmp_path / "absent"))


def test_vemb_shard_and_manifest_spec(tmp_path):
    """format version 3 (a row shard with its pilot replica: three more words behind the 104-byte header) and the manifest of a
    multi-device index ("BVEM"), as DESIGN.md documents them, parsed without a device"""
    import struct
    from bbqlib import bbq_amd as B

    def fnv(data, h):
        data = data + b"\0" * (-len(data) % 8)
        for (w,) in struct.iter_unpack("<Q", data):
            h = ((h ^ w) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return h

    seed = 0xcbf29ce484222325
    dim, n, row_base, pilot = 100, 1000, 4096, 1024
    w16 = ((dim + 7) // 8 + 15) // 16
    stride = w16 * 1024 + 256
    nt, npt = (n + 63) // 64, (pilot + 63) // 64
    tiles, exact = nt * stride, nt * 64 * 32 + nt * 8
    ptiles, pexact = npt * stride, npt * 64 * 32 + npt * 8
    hdr = struct.pack("<4sI4i3qd6i3q", b"BVEC", 3, 0, 0, 1, dim, 0, tiles + exact + ptiles + pexact, n, 0.125, 1, 1, w16, stride, 0, 64, tiles, exact, row_base)
    ext = struct.pack("<3q", pilot, ptiles, pexact)
    cen = np.arange(dim, dtype=np.float32).tobytes()
    dsum = struct.pack("<Q", 0x77)
    p = str(tmp_path / "shard")
    open(p + ".vemb", "wb").write(hdr + ext + cen + dsum + struct.pack("<Q", fnv(dsum, fnv(cen, fnv(ext, fnv(hdr, seed))))))
    assert B.file_info(p) == {"n_rows": n, "dim": dim, "sim": 1, "centroid_dp": 0.125, "row_base": row_base}
    assert B.file_shards(p) == 1
    bad_ext = struct.pack("<3q", row_base + 64, ptiles, pexact)   # more pilot rows than precede the shard
    open(p + ".vemb", "wb").write(hdr + bad_ext + cen + dsum + struct.pack("<Q", fnv(dsum, fnv(cen, fnv(bad_ext, fnv(hdr, seed))))))
    with pytest.raises(B.BBQError):
        B.file_info(p)
    # manifest: header (48 B), {rowBase, rows} per shard, centroid, checksum
    bounds = struct.pack("<4q", 0, 512, 512, 488)
    mh = struct.pack("<4sI4iqdq", b"BVEM", 1, 2, dim, 1, 1, n, 0.125, pilot)
    assert len(mh) == 48
    m = str(tmp_path / "multi")
    open(m + ".vemb", "wb").write(mh + bounds + cen + struct.pack("<Q", fnv(cen, fnv(bounds, fnv(mh, seed)))))
    assert B.file_info(m) == {"n_rows": n, "dim": dim, "sim": 1, "centroid_dp": 0.125, "row_base": 0}
    assert B.file_shards(m) == 2
    gap = struct.pack("<4q", 0, 512, 600, 400)                    # shards must be contiguous
    open(m + ".vemb", "wb").write(mh + gap + cen + struct.pack("<Q", fnv(cen, fnv(gap, fnv(mh, seed)))))
    with pytest.raises(B.BBQError):
        B.file_info(m)
    assert B.file_shards(m) == 0
    assert B.file_shards(str(tmp_path / "absent")) == 0


@pytest.mark.parametrize("sim,qb", [(0, 4), (1, 4), (2, 1), (1, 8)])
def test_quantize_queries_batch_equals_one_by_one(sim, qb):
    """bbq_quantize_queries (host threads) = bbq_quantize_query per query, bit for bit; the first bad query is reported"""
    rng = np.random.default_rng(17)
    dim, n = 200, 37
    cen = (0.05 * rng.standard_normal(dim)).astype(np.float32)
    qs = rng.standard_normal((n, dim)).astype(np.float32)
    qs[5] = 0
    for threads in (1, 3, 0):
        qq, qc = B.quantize_queries(qs, cen, sim, qb, n_threads=threads)
        for i in range(n):
            a, b = B.quantize_query(qs[i], cen, sim, qb)
            np.testing.assert_array_equal(qq[i], a)
            np.testing.assert_array_equal(canon64(qc[i]), canon64(b))
    assert B.quantize_queries(qs[:0], cen, sim, qb)[0].shape == (0, dim)
    bad = qs.copy()
    bad[30, 7] = np.inf
    bad[12, 3] = np.nan
    with pytest.raises(B.BBQError) as e:
        B.quantize_queries(bad, cen, sim, qb, n_threads=4)
    with pytest.raises(B.BBQError) as e1:
        B.quantize_query(bad[12], cen, sim, qb)
    assert e.value.code == e1.value.code and str(e.value) == str(e1.value)
    with pytest.raises(B.BBQError):
        B.quantize_queries(qs[:, :10], cen, sim, qb)


@pytest.mark.parametrize("seed", range(12))
def test_heap_order_is_the_descending_sort_when_the_top_scores_differ(seed):
    """the fact the device-side answers rest on: if the k+1 largest f32 scores are pairwise different, the
    reference's heap returns the k best rows sorted by score, whatever the arrival order and however many ties lie below;
    checked here with the product's replay of the reference heap (bbq_replay) and with the oracle's"""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([5, 64, 1000, 20000]))
    k = int(rng.choice([1, 3, 10, 100]))
    k2 = min(k, n)
    # few distinct values below the top, strictly separated values on top, shuffled arrival order
    s = rng.integers(0, 7, n).astype(np.float32) * np.float32(0.125)
    top = rng.choice(n, min(k2 + 1, n), replace=False)
    s[top] = np.float32(10.0) + np.arange(len(top), dtype=np.float32) * np.float32(0.5)
    want = np.argsort(-s, kind="stable")[:k2]
    ent = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | s.view(np.uint32).astype(np.uint64)
    idx, sc = B.replay([ent], n, k)
    np.testing.assert_array_equal(idx, want)
    np.testing.assert_array_equal(sc, s[want])
    oi, osc = O.heap_topk(s, k)
    np.testing.assert_array_equal(oi, want)
    # and the converse matters too: with a tie inside the answer the heap's history decides, so the sort is NOT generally the answer
    if n >= 64 and k2 >= 3:
        t = s.copy()
        t[top[:2]] = t[top[0]]
        oi2, _ = O.heap_topk(t, k)
        idx2, _ = B.replay([(np.arange(n, dtype=np.uint64) << np.uint64(32)) | t.view(np.uint32).astype(np.uint64)], n, k)
        np.testing.assert_array_equal(idx2, oi2)     # the replay follows the reference either way
