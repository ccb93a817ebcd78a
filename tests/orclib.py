"""ctypes binding of the CPU oracle (oracle/libbbq_oracle.so) + golden-fixture loader.

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module.
"""
import base64
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
SIMS = {"EUCLIDEAN": 0, "COSINE": 1, "MAXIMUM_INNER_PRODUCT": 2}

_lib = None


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libbbq_oracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("bbq_oracle.c", "bbq_oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        f32p, f64p, u8p, i32p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32))
        L.orc_normalize.argtypes = [f32p, C.c_int, f32p]
        L.orc_centroid.argtypes = [f32p, C.c_int64, C.c_int, f32p]
        L.orc_dot_f32.argtypes = [f32p, f32p, C.c_int]
        L.orc_dot_f32.restype = C.c_double
        L.orc_scalar_quantize.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_double, C.c_int, u8p, f64p]
        L.orc_pack_binary.argtypes = [u8p, C.c_int, u8p]
        L.orc_pack_binary.restype = C.c_int
        L.orc_build_index.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, u8p, f64p, f32p]
        L.orc_build_index_unpacked.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, u8p, f64p, f32p]
        L.orc_quantize_query.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_double, C.c_int, u8p, f64p]
        L.orc_qcdist_unpacked_query.argtypes = [u8p, u8p, C.c_int]
        L.orc_qcdist_unpacked_query.restype = C.c_int32
        L.orc_qcdist_packed_query.argtypes = [u8p, u8p, C.c_int]
        L.orc_qcdist_packed_query.restype = C.c_int32
        L.orc_dot_u8.argtypes = [u8p, u8p, C.c_int]
        L.orc_dot_u8.restype = C.c_int32
        L.orc_score.argtypes = [C.c_int32, f64p, f64p, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_score.restype = C.c_double
        L.orc_score_all.argtypes = [u8p, f64p, C.c_int64, C.c_int, u8p, f64p, C.c_int, C.c_int, C.c_double, i32p, f64p, f32p]
        L.orc_score_single_row.argtypes = [C.c_int32, f64p, f64p, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_score_single_row.restype = C.c_double
        L.orc_score_all_multibit.argtypes = [u8p, f64p, C.c_int64, C.c_int, u8p, f64p, C.c_int, C.c_int, C.c_double, i32p, f64p, f32p]
        L.orc_score_all_multibit.restype = C.c_int
        L.orc_score_all_multibit_ext.argtypes = [u8p, f64p, C.c_int64, C.c_int, u8p, f64p, C.c_int, C.c_int, C.c_double, i32p, f64p, f32p]
        L.orc_score_all_multibit_ext.restype = None
        L.orc_search_multibit.argtypes = [f32p, C.c_int, u8p, f64p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                          C.c_int64, i32p, f32p]
        L.orc_search_multibit.restype = C.c_int64
        L.orc_heap_topk.argtypes = [f32p, C.c_int64, C.c_int64, i32p, f32p]
        L.orc_heap_topk.restype = C.c_int64
        L.orc_search.argtypes = [f32p, C.c_int, u8p, f64p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                 C.c_int64, i32p, f32p]
        L.orc_search.restype = C.c_int64
        L.orc_oversampled_topk.argtypes = [f32p, f32p, u8p, f64p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double,
                                           C.c_int, C.c_int64, C.c_int, i32p]
        L.orc_oversampled_topk.restype = C.c_int64
        L.orc_mulberry32_fill.argtypes = [C.c_uint32, f32p, C.c_int64]
        L.orc_true_similarity.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.orc_true_similarity.restype = C.c_double
        L.orc_rerank_select_heap.argtypes = [f64p, C.c_int64, C.c_int64, i32p]
        L.orc_rerank_select_heap.restype = C.c_int64
        L.orc_rerank_select_sort.argtypes = [f64p, C.c_int64, C.c_int64, i32p]
        L.orc_rerank_select_sort.restype = C.c_int64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def f32p(a):
    return _p(a, C.c_float)


def f64p(a):
    return _p(a, C.c_double)


def u8p(a):
    return _p(a, C.c_uint8)


def i32p(a):
    return _p(a, C.c_int32)


# ---------------------------------------------------------------- numpy-level wrappers

def mulberry32(seed, count):
    out = np.empty(count, np.float32)
    lib().orc_mulberry32_fill(seed, f32p(out), count)
    return out


def build_index(base, sim, lam=0.1, iters=5, ib=1):
    if ib != 1:
        return build_index_unpacked(base, sim, ib, lam, iters)
    base = np.ascontiguousarray(base, np.float32)
    n, dim = base.shape
    pb = (dim + 7) // 8
    codes = np.zeros((n, pb), np.uint8)
    corr = np.zeros((n, 4), np.float64)
    cen = np.zeros(dim, np.float32)
    lib().orc_build_index(f32p(base), n, dim, sim, lam, iters, u8p(codes), f64p(corr), f32p(cen))
    return codes, corr, cen


def build_index_unpacked(base, sim, ib, lam=0.1, iters=5):
    base = np.ascontiguousarray(base, np.float32)
    n, dim = base.shape
    codes = np.zeros((n, dim), np.uint8)
    corr = np.zeros((n, 4), np.float64)
    cen = np.zeros(dim, np.float32)
    lib().orc_build_index_unpacked(f32p(base), n, dim, sim, ib, lam, iters, u8p(codes), f64p(corr), f32p(cen))
    return codes, corr, cen


def quantize_query(query, cen, sim, qb, lam=0.1, iters=5):
    query = np.ascontiguousarray(query, np.float32)
    dim = query.shape[0]
    qq = np.zeros(dim, np.uint8)
    qc = np.zeros(4, np.float64)
    lib().orc_quantize_query(f32p(query), dim, f32p(cen), sim, qb, lam, iters, u8p(qq), f64p(qc))
    return qq, qc


def centroid_dp(cen):
    return lib().orc_dot_f32(f32p(cen), f32p(cen), cen.shape[0])


class ReferenceThrows(Exception):
    """the reference throws on this input (e.g. indexBits > 1 with queryBits other than 1 and 4)"""


def score_all(codes, corr, dim, qq, qc, qb, sim, cdp, ib=1):
    """ib > 1: codes are unpacked [n, dim]; cdp is centroid.centroid either way (the multi-bit path decides what it uses)"""
    n = codes.shape[0]
    d = np.zeros(n, np.int32)
    s64 = np.zeros(n, np.float64)
    s32 = np.zeros(n, np.float32)
    codes = np.ascontiguousarray(codes)
    corr = np.ascontiguousarray(corr)
    if ib != 1:
        assert codes.shape[1] == dim
        rc = lib().orc_score_all_multibit(u8p(codes), f64p(corr), n, dim, u8p(qq), f64p(qc), qb, sim, cdp, i32p(d), f64p(s64), f32p(s32))
        if rc != 0:
            raise ReferenceThrows("不支持的查询位数: %d，只支持1位和4位" % qb)
        return d, s64, s32
    lib().orc_score_all(u8p(codes), f64p(corr), n, dim, u8p(qq), f64p(qc), qb, sim, cdp, i32p(d), f64p(s64), f32p(s32))
    return d, s64, s32


def score_all_multibit_ext(codes, corr, dim, qq, qc, qb, sim, cdp):
    """libbbq's documented extension for queryBits the reference throws on (multi-bit index): per-row 4-bit form, any queryBits"""
    n = codes.shape[0]
    d = np.zeros(n, np.int32)
    s64 = np.zeros(n, np.float64)
    s32 = np.zeros(n, np.float32)
    codes = np.ascontiguousarray(codes)
    corr = np.ascontiguousarray(corr)
    assert codes.shape[1] == dim
    lib().orc_score_all_multibit_ext(u8p(codes), f64p(corr), n, dim, u8p(qq), f64p(qc), qb, sim, cdp, i32p(d), f64p(s64), f32p(s32))
    return d, s64, s32


def heap_topk(s32, k):
    s32 = np.ascontiguousarray(s32, np.float32)
    n = s32.shape[0]
    m = max(min(k, n), 0)
    idx = np.zeros(m + 1, np.int32)
    sc = np.zeros(m + 1, np.float32)
    cnt = lib().orc_heap_topk(f32p(s32), n, k, i32p(idx), f32p(sc))
    return idx[:cnt].copy(), sc[:cnt].copy()


def search(query, codes, corr, cen, sim, qb, k, lam=0.1, iters=5, ib=1):
    query = np.ascontiguousarray(query, np.float32)
    n = codes.shape[0]
    dim = cen.shape[0]
    m = max(min(k, n), 0)
    idx = np.zeros(m + 1, np.int32)
    sc = np.zeros(m + 1, np.float32)
    fn = lib().orc_search if ib == 1 else lib().orc_search_multibit
    cnt = fn(f32p(query), query.shape[0], u8p(codes), f64p(corr), f32p(cen), n, dim, sim, qb, lam, iters, k,
                           i32p(idx), f32p(sc))
    if cnt < 0:
        return cnt, None
    return idx[:cnt].copy(), sc[:cnt].copy()


def true_similarity(queries, base, sim):
    """computeSimilarity for every (query, row): f64 [nq, n]"""
    queries = np.ascontiguousarray(queries, np.float32)
    base = np.ascontiguousarray(base, np.float32)
    nq, dim = queries.shape
    out = np.zeros((nq, base.shape[0]), np.float64)
    L = lib()
    for qi in range(nq):
        for i in range(base.shape[0]):
            out[qi, i] = L.orc_true_similarity(f32p(queries[qi]), f32p(base[i]), dim, sim)
    return out


def rerank_select(true_scores, k, how="heap"):
    """positions into the candidate list, in the order getOversampledTopKWith{Heap,Sort} return them"""
    t = np.ascontiguousarray(true_scores, np.float64)
    out = np.zeros(max(min(k, t.shape[0]), 0) + 1, np.int32)
    fn = lib().orc_rerank_select_heap if how == "heap" else lib().orc_rerank_select_sort
    cnt = fn(f64p(t), t.shape[0], k, i32p(out))
    return out[:cnt].copy()


# ---------------------------------------------------------------- golden fixtures

def _dec(s, dt):
    return np.frombuffer(base64.b64decode(s), dtype=dt).copy()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_names(pattern="*"):
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN_DIR, pattern + ".json")))


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json"), encoding="utf-8") as f:
        return json.load(f)


def golden_inputs(g):
    """(base f32[n,dim], queries f32[nq,dim]) regenerated or decoded exactly as gen_fixtures.js made them."""
    n, dim, nq = g["n"], g["dim"], g["nq"]
    gen = g["gen"]
    if gen["kind"] == "mulberry32":
        base = mulberry32(gen["base_seed"], n * dim).reshape(n, dim)
        queries = mulberry32(gen["query_seed"], nq * dim).reshape(nq, dim)
    elif gen["kind"] == "dup_pool":
        pool = mulberry32(gen["base_seed"], gen["pool"] * dim).reshape(gen["pool"], dim)
        # pick = floor(u * pool) with u from mulberry32(pick_seed); recover u from the f32 generator is lossy,
        # so regenerate u in float64 here (same integer recurrence)
        u = mulberry32_u(gen["pick_seed"], n)
        pick = np.floor(u * gen["pool"]).astype(np.int64)
        base = pool[pick]
        queries = mulberry32(gen["query_seed"], nq * dim).reshape(nq, dim)
    else:
        base = _dec(g["base_f32"], "<f4").reshape(n, dim)
        queries = _dec(g["queries_f32"], "<f4").reshape(nq, dim)
    if "base_sha256" in g:
        assert sha(base) == g["base_sha256"], "input generator drifted from gen_fixtures.js"
        assert sha(queries) == g["queries_sha256"]
    return np.ascontiguousarray(base, np.float32), np.ascontiguousarray(queries, np.float32)


def mulberry32_u(seed, count):
    """the raw uniform doubles of mulberry32 (vectorised: the state is a counter)"""
    i = np.arange(1, count + 1, dtype=np.uint64)
    a = ((np.uint64(seed) + i * np.uint64(0x6D2B79F5)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    with np.errstate(over="ignore"):
        t = (a ^ (a >> np.uint32(15))) * (np.uint32(1) | a)
        t = (t + ((t ^ (t >> np.uint32(7))) * (np.uint32(61) | t))) ^ t
        r = t ^ (t >> np.uint32(14))
    return r.astype(np.float64) / 4294967296.0


def dec(s, dt):
    return _dec(s, dt)
