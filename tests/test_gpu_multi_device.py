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
            assert st["dense_fallbacks"] == 0 and st["candidates"] >= nq * k and st["total_scan_launches"] > 0
            assert st["host_replays"] == 0   # gaussian data has no equal scores in an answer: merged from the shard-local answers alone
        # the ABI-2 path (every query replays its lists) is still there and agrees
        ix.set_option("device_select", 0)
        idx0, sc0, _ = ix.search_batch(qq, qc, 4, sim, k)
        assert ix.stats()["host_replays"] == nq
        ix.set_option("device_select", 1)
        np.testing.assert_array_equal(idx0, idx)
        np.testing.assert_array_equal(canon32(sc0), canon32(sc))
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


@pytest.mark.parametrize("shards", [4, 8])
def test_multi_device_full_size_10m(shards):
    """the headline size through the multi-device handle (4 shards, and the 8 shards of the --gpus 8 target, all on the one GPU): same
    answers as the single-device index"""
    import bench
    n, dim, k, pb = 10_000_000, 768, 100, 96
    codes, corr = bench.synth_rows(1, 0, n, pb)
    qq, qc = bench.synth_queries(2, 40, dim)
    cdp = 0.0009110655808639536
    single = B.Index(codes, corr, dim, cdp)
    want = single.search_batch(qq, qc, 4, 1, k)
    # queries whose answer has equal scores in or at the edge of it: only those may need the heap replayed (the synthetic scores are
    # f32 values in a narrow range, a pair among the 101 best coincides now and then)
    _, s101, _ = single.search_batch(qq, qc, 4, 1, k + 1)
    tied = sum(len(np.unique(s101[q].astype(np.float64))) != k + 1 for q in range(len(qq)))
    single.close()
    ix = B.Index.create_multi(codes, corr, dim, cdp, [0] * shards)
    try:
        assert ix.shards == shards
        ix.set_option("round_queries", 16)
        got = ix.search_batch(qq, qc, 4, 1, k)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(canon32(got[1]), canon32(want[1]))
        st = ix.stats()
        assert st["dense_fallbacks"] == 0 and st["host_replays"] == tied
    finally:
        ix.close()


def _key_of(s32):
    b = s32.view(np.uint32).astype(np.int64)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


@pytest.mark.parametrize("name,pilot", [("big_20000x128_cos", 2048), ("ties_cos_qb4", 1024), ("big_50000x768_cos", 0)])
def test_shard_answers_block_and_async_scans(name, pilot):
    """bbq_shard_scan_begin / _wait on one shard: (a) the answer block is what include/bbq.h says - the cut is the (k+1)-th largest key
    over the shard's own rows and its pilot replica, the entries are the shard's own rows above it, descending; (b) two batches in
    flight give what two synchronous scans give; (c) merged with the root shard's block the answers are the reference's"""
    import torch
    g, sim, base, queries, codes, corr, cen, cdp = _case(name)
    n, dim, qb = g["n"], g["dim"], g["qb"]
    k = 10
    half = (n // 2) // 512 * 512
    P = min(pilot, half) // 512 * 512
    root = B.Index(codes[:half], corr[:half], dim, cdp)
    sh = B.Index(codes[half:], corr[half:], dim, cdp, row_base=half, pilot_codes=codes[:P] if P else None, pilot_corr=corr[:P] if P else None)
    qs = [B.quantize_query(q, cen, sim, qb, g["lambda"], g["iters"]) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    nq, stride = len(qq), k + 3
    try:
        for ix in (root, sh):
            ix.set_option("first_segment_rows", 1024)
            ix.set_option("segment_growth", 2)
            ix.set_option("batch_queries", 2)   # several sub-batches per batch: slots are reused while a batch is in flight

        def bufs(ix):
            cap = int(ix.shard_list_cap(k)) * nq
            return {"cap": cap, "packed": torch.zeros(cap, dtype=torch.int64, device="cuda"), "off": torch.zeros(nq + 1, dtype=torch.int64, device="cuda"),
                    "flags": torch.zeros(nq, dtype=torch.int32, device="cuda"), "ans": torch.zeros(nq * stride, dtype=torch.int64, device="cuda")}

        def begin(ix, b, q0, q1):
            ix.shard_scan_begin(qq[q0:q1], qc[q0:q1], qb, sim, k, b["packed"].data_ptr(), b["cap"], b["off"].data_ptr(), b["flags"].data_ptr(),
                                b["ans"].data_ptr(), stride)

        b1, b2, b3 = bufs(sh), bufs(sh), bufs(sh)
        assert nq >= 2
        m = nq // 2
        begin(sh, b1, 0, m)          # two batches in flight on one index
        begin(sh, b2, m, nq)
        with pytest.raises(B.BBQError):
            begin(sh, b3, 0, 1)      # a third one is refused
        t1 = sh.shard_scan_wait()
        t2 = sh.shard_scan_wait()
        with pytest.raises(B.BBQError):
            sh.shard_scan_wait()     # nothing in flight
        begin(sh, b3, 0, nq)         # the same queries in one batch
        t3 = sh.shard_scan_wait()
        assert t1 + t2 == t3
        a12 = np.concatenate([b1["ans"].cpu().numpy().view(np.uint64).reshape(nq, stride)[:m],
                              b2["ans"].cpu().numpy().view(np.uint64).reshape(nq, stride)[:nq - m]])
        a3 = b3["ans"].cpu().numpy().view(np.uint64).reshape(nq, stride)
        np.testing.assert_array_equal(a12, a3)
        p12 = np.concatenate([b1["packed"][:t1].cpu().numpy(), b2["packed"][:t2].cpu().numpy()])
        np.testing.assert_array_equal(p12, b3["packed"][:t3].cpu().numpy())
        # (a) the block against its definition
        for q in range(nq):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
            if np.isnan(s32).any():
                continue
            keys = _key_of(s32)
            seen = np.unique(np.concatenate([np.arange(0, P), np.arange(half, n)]))
            ks = np.sort(keys[seen])[::-1]
            cut = int(ks[k]) if len(ks) >= k + 1 else 0
            own = np.arange(half, n)
            own = own[keys[own] > cut]
            blk = a3[q]
            assert int(blk[0] >> np.uint64(32)) == 0 and int(blk[1] >> np.uint64(32)) == 0
            assert int(blk[2]) == cut
            mm = int(blk[1] & np.uint64(0xffffffff))
            assert mm == len(own)
            rows = (blk[3:3 + mm] >> np.uint64(32)).astype(np.int64)
            assert sorted(rows.tolist()) == sorted(own.tolist())
            got_keys = _key_of((blk[3:3 + mm] & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32))
            assert (np.diff(got_keys) <= 0).all()
            np.testing.assert_array_equal((blk[3:3 + mm] & np.uint64(0xffffffff)).astype(np.uint32), s32[rows].view(np.uint32))
        # (c) merged with the root shard
        br = bufs(root)
        begin(root, br, 0, nq)
        tr = root.shard_scan_wait()
        ar = br["ans"].cpu().numpy().view(np.uint64).reshape(nq, stride)
        idx, sc, cnt, status = B.merge_answers([a3, ar], nq, n, k, 2)
        lists = [br["packed"][:tr].cpu().numpy().view(np.uint64), b3["packed"][:t3].cpu().numpy().view(np.uint64)]
        offs = [br["off"].cpu().numpy(), b3["off"].cpu().numpy()]
        ri, rs, rc = B.replay_batch(lists, offs, nq, n, k, 2)
        for q in range(nq):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            np.testing.assert_array_equal(ri[q, :rc[q]], oi)                       # the lists (rank k + 1 supersets) replay to the reference's answer
            np.testing.assert_array_equal(canon32(rs[q, :rc[q]]), canon32(osc))
            if status[q] == 0:
                np.testing.assert_array_equal(idx[q, :cnt[q]], oi)
                np.testing.assert_array_equal(canon32(sc[q, :cnt[q]]), canon32(osc))
            else:
                top = np.sort(s32)[::-1][:k + 1].astype(np.float64)
                assert len(np.unique(top)) < len(top)                              # only equal scores may send a query to the replay
        if name.startswith("big_"):
            assert (status == 0).all()
        st = sh.stats()
        assert st["total_scan_launches"] > 0
    finally:
        root.close()
        sh.close()


def test_shard_scan_begin_argument_errors():
    rng = np.random.default_rng(3)
    n, dim = 4096, 64
    base = rng.standard_normal((n, dim)).astype(np.float32)
    codes, corr, cen = B.quantize_vectors(base, 1)
    ix = B.Index(codes, corr, dim, B.centroid_dp(cen))
    import torch
    qq, qc = B.quantize_queries(rng.standard_normal((2, dim)).astype(np.float32), cen, 1, 4)
    d = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    try:
        with pytest.raises(B.BBQError):   # stride too small for k
            ix.shard_scan_begin(qq, qc, 4, 1, 10, d.data_ptr(), 1 << 15, d.data_ptr(), d.data_ptr(), d.data_ptr(), 12)
        with pytest.raises(B.BBQError):   # answers beyond what the finalize launch selects
            ix.shard_scan_begin(qq, qc, 4, 1, 2000, d.data_ptr(), 1 << 15, d.data_ptr(), d.data_ptr(), d.data_ptr(), 2003)
        with pytest.raises(B.BBQError):
            ix.shard_scan_wait()
        # an index destroyed with a batch in flight settles it first
        p, o, f = torch.zeros(1 << 16, dtype=torch.int64, device="cuda"), torch.zeros(3, dtype=torch.int64, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda")
        ix.shard_scan_begin(qq, qc, 4, 1, 10, p.data_ptr(), 1 << 16, o.data_ptr(), f.data_ptr(), None, 0)
    finally:
        ix.close()
    # another index of the same device is searched while a shard's batch is still in flight: the entry point settles the busy slots first
    sh = B.Index(codes[2048:], corr[2048:], dim, B.centroid_dp(cen), row_base=2048, pilot_codes=codes[:1024], pilot_corr=corr[:1024])
    other = B.Index(codes, corr, dim, B.centroid_dp(cen))
    try:
        p, o, f = torch.zeros(1 << 16, dtype=torch.int64, device="cuda"), torch.zeros(3, dtype=torch.int64, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda")
        a = torch.zeros(2 * 13, dtype=torch.int64, device="cuda")
        sh.shard_scan_begin(qq, qc, 4, 1, 10, p.data_ptr(), 1 << 16, o.data_ptr(), f.data_ptr(), a.data_ptr(), 13)
        idx, sc = other.search(qq[1], qc[1], 4, 1, 10)
        _, _, s32 = O.score_all(codes, corr, dim, qq[1], qc[1], 4, 1, B.centroid_dp(cen))
        np.testing.assert_array_equal(idx, O.heap_topk(s32, 10)[0])
        total = sh.shard_scan_wait()
        assert total == int(o.cpu().numpy()[2]) and int(f.abs().sum().item()) == 0
        blk = a.cpu().numpy().view(np.uint64).reshape(2, 13)
        assert (blk[:, 1] >> np.uint64(32) == 0).all()          # proven shard-local answers
        assert sh.stats()["total_scan_launches"] > 0            # timings were booked when the slots were settled
    finally:
        sh.close()
        other.close()
    # the device context is intact afterwards
    ix = B.Index(codes, corr, dim, B.centroid_dp(cen))
    try:
        idx, sc = ix.search(qq[0], qc[0], 4, 1, 10)
        _, _, s32 = O.score_all(codes, corr, dim, qq[0], qc[0], 4, 1, B.centroid_dp(cen))
        np.testing.assert_array_equal(idx, O.heap_topk(s32, 10)[0])
    finally:
        ix.close()


@pytest.mark.parametrize("name,shards,pilot", [("big_20000x128_cos", 3, 2048), ("ties_cos_qb4", 2, 512), ("ib2_big_20000x128_euc", 4, 1024)])
def test_multi_device_save_load_roundtrip(tmp_path, name, shards, pilot):
    """a multi-device index is saved as one pair per shard (pilot replica included) + a manifest and comes back - over several
    devices (bbq_index_load_multi) or over one (bbq_index_load) - answering exactly like before; no host rows are needed"""
    g, sim, base, queries, codes, corr, cen, cdp = _case(name)
    dim, qb, k = g["dim"], g["qb"], 10
    qs = [B.quantize_query(q, cen, sim, qb, g["lambda"], g["iters"]) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    ix = B.Index.create_multi(codes, corr, dim, cdp, [0] * shards, index_bits=g["ib"], pilot_rows=pilot)
    prefix = str(tmp_path / "multi")
    try:
        want = ix.search_batch(qq, qc, qb, sim, k)
        ix.save(prefix, cen, sim)
        n_shards = ix.shards
    finally:
        ix.close()
    assert B.file_shards(prefix) == n_shards
    info = B.file_info(prefix)
    assert info["n_rows"] == g["n"] and info["dim"] == dim and info["sim"] == sim and info["row_base"] == 0
    assert np.float64(info["centroid_dp"]).view(np.uint64) == np.float64(cdp).view(np.uint64)
    for loader in (lambda: B.Index.load_multi(prefix, [0] * n_shards), lambda: B.Index.load_multi(prefix, [0]), lambda: B.Index.load_multi(prefix),
                   lambda: B.Index.load(prefix, 0)):
        lx, lcen, _ = loader()
        try:
            assert lx.shards == n_shards and lx.index_bits == g["ib"]
            np.testing.assert_array_equal(lcen, cen)
            got = lx.search_batch(qq, qc, qb, sim, k)
            np.testing.assert_array_equal(got[0], want[0])
            np.testing.assert_array_equal(canon32(got[1]), canon32(want[1]))
            np.testing.assert_array_equal(got[2], want[2])
            c2, r2 = lx.export()
            np.testing.assert_array_equal(c2, codes)
            np.testing.assert_array_equal(canon64(r2), canon64(corr))
        finally:
            lx.close()
    # a shard file is an ordinary (version 3) pair: it loads as the shard it was, pilot replica included
    if n_shards > 1 and pilot > 0:
        import torch
        sh, _, sinfo = B.Index.load(prefix + ".s001", 0)
        try:
            assert sinfo["row_base"] > 0 and sinfo["n_rows"] == sh.n
            r0 = sinfo["row_base"]
            P = min(pilot, r0) // 512 * 512 if pilot < r0 else r0
            ref = B.Index(codes[r0:r0 + sh.n], corr[r0:r0 + sh.n], dim, cdp, index_bits=g["ib"], row_base=r0,
                          pilot_codes=codes[:P] if P else None, pilot_corr=corr[:P] if P else None)
            outs = []
            for h in (sh, ref):
                cap = int(h.shard_list_cap(k)) * len(qq)
                d_p, d_o, d_f = torch.zeros(cap, dtype=torch.int64, device="cuda"), torch.zeros(len(qq) + 1, dtype=torch.int64, device="cuda"), torch.zeros(len(qq), dtype=torch.int32, device="cuda")
                total = h.shard_scan(qq, qc, qb, sim, k, d_p.data_ptr(), cap, d_o.data_ptr(), d_f.data_ptr())
                outs.append((d_p[:total].cpu().numpy(), d_o.cpu().numpy(), d_f.cpu().numpy()))
            ref.close()
            for a, b in zip(outs[0], outs[1]):
                np.testing.assert_array_equal(a, b)   # same thresholds from the same pilot rows: the same candidate lists
            with pytest.raises(B.BBQError):
                sh.search(qq[0], qc[0], qb, sim, k)   # a non-root shard does not answer searches by itself
        finally:
            sh.close()
    # damage: a flipped byte in the manifest, a missing shard file, a shard that does not match the manifest
    raw = bytearray(open(prefix + ".vemb", "rb").read())
    raw[20] ^= 1
    open(prefix + ".bad.vemb", "wb").write(bytes(raw))
    with pytest.raises(B.BBQError):
        B.file_info(prefix + ".bad")
    import os
    import shutil
    os.rename(prefix + ".s000.veb", prefix + ".s000.veb.away")
    with pytest.raises(B.BBQError):
        B.Index.load_multi(prefix, [0])
    os.rename(prefix + ".s000.veb.away", prefix + ".s000.veb")
    if n_shards > 1:
        for ext in (".veb", ".vemb"):
            shutil.copy(prefix + ".s001" + ext, prefix + ".s000" + ext)
        with pytest.raises(B.BBQError):
            B.Index.load_multi(prefix, [0])


@pytest.mark.gpu
def test_shard_without_rows_leaves_empty_answer_blocks():
    """a shard handle with no rows and no pilot replica (a rank beyond the end of a small index) launches nothing that would write its
    answer blocks: they must come back as "nothing listed, no cut, no flags" - not as whatever the caller's buffer held - and merge
    with the root shard's blocks into the reference's answers (advisor, round 3)"""
    import torch
    g, sim, base, queries, codes, corr, cen, cdp = _case("big_20000x128_cos")
    n, dim, qb, k = g["n"], g["dim"], g["qb"], 10
    root = B.Index(codes, corr, dim, cdp)
    empty = B.Index(codes[:0], corr[:0], dim, cdp, row_base=n)
    qs = [B.quantize_query(q, cen, sim, qb, g["lambda"], g["iters"]) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    nq, stride = len(qq), k + 3
    try:
        blocks = []
        for ix in (root, empty):
            cap = max(int(ix.shard_list_cap(k)), 1) * nq
            packed = torch.zeros(cap, dtype=torch.int64, device="cuda")
            off = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
            flags = torch.zeros(nq, dtype=torch.int32, device="cuda")
            ans = torch.full((nq * stride,), -1, dtype=torch.int64, device="cuda")      # garbage the scan has to overwrite
            ix.shard_scan_begin(qq, qc, qb, sim, k, packed.data_ptr(), cap, off.data_ptr(), flags.data_ptr(), ans.data_ptr(), stride)
            total = ix.shard_scan_wait()
            blk = ans.cpu().numpy().view(np.uint64).reshape(nq, stride)
            if ix is empty:
                assert total == 0
                assert (blk[:, :3] == 0).all(), "the empty shard's headers are not zero"
                assert int(flags.cpu().numpy().sum()) == 0 and int(off.cpu().numpy()[nq]) == 0
            blocks.append(blk)
        idx, sc, cnt, status = B.merge_answers(blocks, nq, n, k)
        for q in range(nq):
            _, _, s32 = O.score_all(codes, corr, dim, qq[q], qc[q], qb, sim, cdp)
            oi, osc = O.heap_topk(s32, k)
            if status[q] == 0:
                np.testing.assert_array_equal(idx[q], oi)
                np.testing.assert_array_equal(canon32(sc[q]), canon32(osc))
        assert (status == 0).sum() >= nq // 2      # (queries with equal scores in their answer ask for the lists: status 1)
    finally:
        root.close()
        empty.close()
