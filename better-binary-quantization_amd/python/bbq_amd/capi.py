"""ctypes binding of libbbq.so (include/bbq.h).  Plumbing only: every numeric operation happens in the
C++/HIP library.  There is no CPU fallback - device entry points raise BBQError(BBQ_ERR_NO_DEVICE) without a GPU."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))            # better-binary-quantization_amd/
LIB_PATH = os.environ.get("BBQ_LIB") or os.path.join(PKG_ROOT, "lib", "libbbq.so")  # BBQ_LIB: experiments with alternative builds

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED = 0, 1, 2, 3, 4, 5
ERR_DIM_MISMATCH, ERR_NEGATIVE_K, ERR_NAN_INPUT, ERR_INF_INPUT, ERR_EMPTY = 6, 7, 8, 9, 10
EUCLIDEAN, COSINE, MAXIMUM_INNER_PRODUCT = 0, 1, 2
SIMS = {"EUCLIDEAN": 0, "COSINE": 1, "MAXIMUM_INNER_PRODUCT": 2}

# every symbol include/bbq.h declares (tests/test_capi_symbols.py checks the library exports all of them)
SYMBOLS = [
    "bbq_last_error", "bbq_abi_version", "bbq_device_count", "bbq_index_create", "bbq_index_create_shard", "bbq_index_create_multi", "bbq_index_shards", "bbq_index_build", "bbq_index_build_bits",
    "bbq_index_destroy", "bbq_index_size", "bbq_index_dimension", "bbq_index_bytes_per_row", "bbq_index_bits", "bbq_search",
    "bbq_search_batch", "bbq_score_rows", "bbq_shard_scan", "bbq_shard_list_cap", "bbq_replay", "bbq_replay_batch",
    "bbq_quantize_vectors", "bbq_quantize_query", "bbq_quantize_query_vector", "bbq_centroid_dp", "bbq_get_stats",
    "bbq_reset_stats", "bbq_set_option", "bbq_vectors_create", "bbq_vectors_destroy", "bbq_vectors_size",
    "bbq_vectors_dimension", "bbq_rerank_scores", "bbq_search_rerank_batch", "bbq_index_save", "bbq_index_file_info",
    "bbq_index_load", "bbq_index_export", "bbq_quantize_queries",
    "bbq_index_create_shard_opts", "bbq_index_create_multi_opts", "bbq_index_build_opts",
    "bbq_shard_scan_begin", "bbq_shard_scan_wait", "bbq_merge_answers", "bbq_key_of_score", "bbq_index_load_multi", "bbq_index_file_shards",
    "bbq_search_raw_batch",
]


class BBQError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class IndexOptions(C.Structure):
    """bbq_index_options (include/bbq.h)"""
    _fields_ = [("size", C.c_int32), ("corrections", C.c_int32)]


CORRECTIONS_DEFAULT, CORRECTIONS_INLINE, CORRECTIONS_COMPACT = -1, 0, 1


def _opts(corrections):
    """corrections: None (library default) | "inline" | "compact" | a BBQ_CORRECTIONS_* ordinal"""
    if corrections is None:
        return None
    v = {"inline": CORRECTIONS_INLINE, "compact": CORRECTIONS_COMPACT, "default": CORRECTIONS_DEFAULT}.get(corrections, corrections)
    return C.byref(IndexOptions(C.sizeof(IndexOptions), int(v)))


class Stats(C.Structure):
    _fields_ = [("last_scan_ms", C.c_double), ("last_scan_rows", C.c_int64), ("last_scan_bytes", C.c_int64),
                ("candidates", C.c_int64), ("dense_fallbacks", C.c_int64), ("total_scan_ms", C.c_double),
                ("total_scan_bytes", C.c_int64), ("total_scan_launches", C.c_int64), ("host_replays", C.c_int64), ("resident_bytes", C.c_int64)]


_lib = None


def lib():
    """loads libbbq.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()').
    Note: PyTorch-ROCm wheels bundle their own libamdhip64 with the system's soname; a process that uses both must
    import torch BEFORE the first call into libbbq so that one HIP runtime serves both."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BBQError(ERR_NO_DEVICE, "libbbq.so is not built (%s): run __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.bbq_last_error.restype = C.c_char_p
    L.bbq_abi_version.restype = C.c_int
    L.bbq_device_count.restype = C.c_int
    L.bbq_index_create.argtypes = [vp, vp, i64, i32, i32, dbl, i32, C.POINTER(vp)]
    L.bbq_index_create_shard.argtypes = [vp, vp, i64, i32, i32, dbl, i64, vp, vp, i64, i32, C.POINTER(vp)]
    L.bbq_index_create_multi.argtypes = [vp, vp, i64, i32, i32, dbl, i32, vp, i64, C.POINTER(vp)]
    L.bbq_index_create_shard_opts.argtypes = [vp, vp, i64, i32, i32, dbl, i64, vp, vp, i64, i32, vp, C.POINTER(vp)]
    L.bbq_index_create_multi_opts.argtypes = [vp, vp, i64, i32, i32, dbl, i32, vp, i64, vp, C.POINTER(vp)]
    L.bbq_index_build_opts.argtypes = [vp, i64, i32, i32, i32, dbl, i32, i32, vp, C.POINTER(vp), vp, vp, vp, vp, vp]
    L.bbq_index_shards.argtypes = [vp]
    L.bbq_index_shards.restype = i32
    L.bbq_index_build.argtypes = [vp, i64, i32, i32, dbl, i32, i32, C.POINTER(vp), vp, vp, vp, vp, vp]
    L.bbq_index_build_bits.argtypes = [vp, i64, i32, i32, i32, dbl, i32, i32, C.POINTER(vp), vp, vp, vp, vp, vp]
    L.bbq_index_destroy.argtypes = [vp]
    L.bbq_index_destroy.restype = None
    L.bbq_index_size.argtypes = [vp]
    L.bbq_index_size.restype = i64
    L.bbq_index_dimension.argtypes = [vp]
    L.bbq_index_dimension.restype = i32
    L.bbq_index_bytes_per_row.argtypes = [vp]
    L.bbq_index_bytes_per_row.restype = i32
    L.bbq_index_bits.argtypes = [vp]
    L.bbq_index_bits.restype = i32
    L.bbq_search.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp, vp]
    L.bbq_search_batch.argtypes = [vp, i32, vp, vp, i32, i32, i64, vp, vp, vp]
    L.bbq_search_raw_batch.argtypes = [vp, i32, vp, vp, i32, i32, dbl, i32, i32, i64, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.bbq_score_rows.argtypes = [vp, vp, vp, i32, i32, i64, i64, vp, vp, vp]
    L.bbq_shard_scan.argtypes = [vp, i32, vp, vp, i32, i32, i64, vp, i64, vp, vp, C.POINTER(i64)]
    L.bbq_shard_scan_begin.argtypes = [vp, i32, vp, vp, i32, i32, i64, vp, i64, vp, vp, vp, i64]
    L.bbq_shard_scan_wait.argtypes = [vp, C.POINTER(i64)]
    L.bbq_merge_answers.argtypes = [i32, vp, vp, i32, i64, i64, i32, vp, vp, vp, vp]
    L.bbq_key_of_score.argtypes = [C.c_float]
    L.bbq_key_of_score.restype = C.c_uint32
    L.bbq_shard_list_cap.argtypes = [vp, i64]
    L.bbq_shard_list_cap.restype = i64
    L.bbq_replay.argtypes = [i32, vp, vp, i64, i64, vp, vp, vp]
    L.bbq_replay_batch.argtypes = [i32, vp, vp, i32, i64, i64, i32, vp, vp, vp]
    L.bbq_quantize_vectors.argtypes = [vp, i64, i32, i32, i32, dbl, i32, i32, vp, vp, vp, vp, vp]
    L.bbq_quantize_query.argtypes = [vp, i32, vp, i32, i32, dbl, i32, vp, vp]
    L.bbq_quantize_query_vector.argtypes = [vp, i32, vp, i32, i32, dbl, i32, vp, vp]
    L.bbq_quantize_queries.argtypes = [vp, i32, i32, vp, i32, i32, dbl, i32, i32, vp, vp, C.POINTER(i32)]
    L.bbq_centroid_dp.argtypes = [vp, i32]
    L.bbq_centroid_dp.restype = dbl
    L.bbq_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.bbq_reset_stats.argtypes = [vp]
    L.bbq_set_option.argtypes = [vp, C.c_char_p, i64]
    L.bbq_vectors_create.argtypes = [vp, i64, i32, i32, C.POINTER(vp)]
    L.bbq_vectors_destroy.argtypes = [vp]
    L.bbq_vectors_destroy.restype = None
    L.bbq_vectors_size.argtypes = [vp]
    L.bbq_vectors_size.restype = i64
    L.bbq_vectors_dimension.argtypes = [vp]
    L.bbq_vectors_dimension.restype = i32
    L.bbq_rerank_scores.argtypes = [vp, i32, vp, vp, vp, i32, vp]
    L.bbq_search_rerank_batch.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, i64, i32, i32, i32, vp, vp, vp, vp]
    L.bbq_index_save.argtypes = [vp, C.c_char_p, vp, i32]
    L.bbq_index_file_info.argtypes = [C.c_char_p, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(dbl), C.POINTER(i64)]
    L.bbq_index_load.argtypes = [C.c_char_p, i32, C.POINTER(vp), vp]
    L.bbq_index_load_multi.argtypes = [C.c_char_p, i32, vp, C.POINTER(vp), vp]
    L.bbq_index_file_shards.argtypes = [C.c_char_p]
    L.bbq_index_file_shards.restype = i32
    L.bbq_index_export.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _chk(rc):
    if rc != OK:
        raise BBQError(rc, lib().bbq_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data


def device_count():
    return lib().bbq_device_count()


# ------------------------------------------------------------------------------------------------ host quantizer

def quantize_vectors(vectors, sim, index_bits=1, lam=0.1, iters=5, n_threads=0):
    v = np.ascontiguousarray(vectors, np.float32)
    if v.ndim != 2:
        raise BBQError(ERR_INVALID_ARG, "vectors must be [n, dim]")
    n, dim = v.shape
    if n == 0:
        raise BBQError(ERR_EMPTY, "向量集合不能为空")
    pb = (dim + 7) // 8 if index_bits == 1 else dim
    codes = np.zeros((n, pb), np.uint8)
    corr = np.zeros((n, 4), np.float64)
    cen = np.zeros(dim, np.float32)
    _chk(lib().bbq_quantize_vectors(_ptr(v), n, dim, sim, index_bits, lam, iters, n_threads, _ptr(codes), _ptr(corr), _ptr(cen),
                                   None, None))
    return codes, corr, cen


def quantize_query(query, centroid, sim, query_bits=4, lam=0.1, iters=5, search_path=True):
    q = np.ascontiguousarray(query, np.float32)
    cen = np.ascontiguousarray(centroid, np.float32)
    dim = q.shape[0]
    qq = np.zeros(dim, np.uint8)
    qc = np.zeros(4, np.float64)
    fn = lib().bbq_quantize_query if search_path else lib().bbq_quantize_query_vector
    _chk(fn(_ptr(q), dim, _ptr(cen), sim, query_bits, lam, iters, _ptr(qq), _ptr(qc)))
    return qq, qc


def quantize_queries(queries, centroid, sim, query_bits=4, lam=0.1, iters=5, n_threads=0):
    """searchNearestNeighbors' query preparation for a batch, on host threads: (qquant [n, dim], qcorr [n, 4])"""
    q = np.ascontiguousarray(queries, np.float32)
    cen = np.ascontiguousarray(centroid, np.float32)
    if q.ndim != 2 or q.shape[1] != cen.shape[0]:
        raise BBQError(ERR_DIM_MISMATCH, "查询向量维度与目标向量维度不匹配")
    n, dim = q.shape
    qq = np.zeros((n, dim), np.uint8)
    qc = np.zeros((n, 4), np.float64)
    bad = C.c_int32(-1)
    _chk(lib().bbq_quantize_queries(_ptr(q), n, dim, _ptr(cen), sim, query_bits, lam, iters, n_threads, _ptr(qq), _ptr(qc), C.byref(bad)))
    return qq, qc


def centroid_dp(centroid):
    cen = np.ascontiguousarray(centroid, np.float32)
    return lib().bbq_centroid_dp(_ptr(cen), cen.shape[0])


# ------------------------------------------------------------------------------------------------ device index

class Index:
    """a device-resident index shard (bbq_index)"""

    def __init__(self, codes, corr, dim, cdp, device=0, index_bits=1, row_base=0, pilot_codes=None, pilot_corr=None, corrections=None):
        codes = np.ascontiguousarray(codes, np.uint8)
        corr = np.ascontiguousarray(corr, np.float64)
        n = codes.shape[0]
        h = C.c_void_p()
        if pilot_codes is not None:
            pilot_codes = np.ascontiguousarray(pilot_codes, np.uint8)
            pilot_corr = np.ascontiguousarray(pilot_corr, np.float64)
            npilot = pilot_codes.shape[0]
        else:
            npilot = 0
        _chk(lib().bbq_index_create_shard_opts(_ptr(codes), _ptr(corr), n, dim, index_bits, cdp, row_base, _ptr(pilot_codes),
                                              _ptr(pilot_corr), npilot, device, _opts(corrections), C.byref(h)))
        self._h = h
        self.dim = dim
        self.n = n
        self.index_bits = index_bits

    @classmethod
    def create_multi(cls, codes, corr, dim, cdp, devices, index_bits=1, pilot_rows=32768, corrections=None):
        """one index row-sharded over `devices` (a list of HIP ordinals, one shard each; repeats allowed) behind one handle"""
        codes = np.ascontiguousarray(codes, np.uint8)
        corr = np.ascontiguousarray(corr, np.float64)
        dev = np.ascontiguousarray(devices, np.int32)
        h = C.c_void_p()
        _chk(lib().bbq_index_create_multi_opts(_ptr(codes), _ptr(corr), codes.shape[0], dim, index_bits, cdp, len(dev), _ptr(dev), pilot_rows,
                                              _opts(corrections), C.byref(h)))
        self = cls.__new__(cls)
        self._h, self.dim, self.n, self.index_bits = h, dim, codes.shape[0], index_bits
        return self

    @property
    def shards(self):
        return lib().bbq_index_shards(self._h)

    @classmethod
    def build(cls, vectors, sim, lam=0.1, iters=5, device=0, want_host_copy=True, index_bits=1, corrections=None):
        """quantizeVectors on the device (bbq_index_build_bits): returns (index, codes, corr, centroid); codes/corr are None
        unless want_host_copy"""
        v = np.ascontiguousarray(vectors, np.float32)
        if v.ndim != 2:
            raise BBQError(ERR_INVALID_ARG, "vectors must be [n, dim]")
        n, dim = v.shape
        cen = np.zeros(dim, np.float32)
        codes = np.zeros((n, (dim + 7) // 8 if index_bits == 1 else dim), np.uint8) if want_host_copy else None
        corr = np.zeros((n, 4), np.float64) if want_host_copy else None
        h = C.c_void_p()
        _chk(lib().bbq_index_build_opts(_ptr(v), n, dim, sim, index_bits, lam, iters, device, _opts(corrections), C.byref(h), _ptr(cen), _ptr(codes),
                                       _ptr(corr), None, None))
        self = cls.__new__(cls)
        self._h, self.dim, self.n, self.index_bits = h, dim, n, index_bits
        return self, codes, corr, cen

    def save(self, path_prefix, centroid, sim):
        """<prefix>.veb (the device tiles, byte for byte) + <prefix>.vemb (MetadataFormat + geometry + centroid)"""
        cen = np.ascontiguousarray(centroid, np.float32)
        if cen.shape != (self.dim,):
            raise BBQError(ERR_DIM_MISMATCH, "centroid must be [dim]")
        _chk(lib().bbq_index_save(self._h, os.fsencode(path_prefix), _ptr(cen), sim))

    @classmethod
    def load(cls, path_prefix, device=0):
        """returns (index, centroid, info): a straight file -> HBM copy, no re-tiling"""
        info = file_info(path_prefix)
        cen = np.zeros(info["dim"], np.float32)
        h = C.c_void_p()
        _chk(lib().bbq_index_load(os.fsencode(path_prefix), device, C.byref(h), _ptr(cen)))
        self = cls.__new__(cls)
        self._h, self.dim, self.n = h, info["dim"], info["n_rows"]
        self.index_bits = lib().bbq_index_bits(h)
        return self, cen, info

    @classmethod
    def load_multi(cls, path_prefix, devices=None):
        """a saved multi-device index back over `devices` (None: shard s on device s modulo the visible ones): (index, centroid, info)"""
        info = file_info(path_prefix)
        cen = np.zeros(info["dim"], np.float32)
        dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
        h = C.c_void_p()
        _chk(lib().bbq_index_load_multi(os.fsencode(path_prefix), 0 if dev is None else len(dev), _ptr(dev), C.byref(h), _ptr(cen)))
        self = cls.__new__(cls)
        self._h, self.dim, self.n = h, info["dim"], info["n_rows"]
        self.index_bits = lib().bbq_index_bits(h)
        return self, cen, info

    def export(self):
        """(codes [n, ceil(dim/8)], corr [n, 4]) as vectorValue / getCorrectiveTerms would return them"""
        codes = np.zeros((self.n, (self.dim + 7) // 8 if self.index_bits == 1 else self.dim), np.uint8)
        corr = np.zeros((self.n, 4), np.float64)
        _chk(lib().bbq_index_export(self._h, _ptr(codes), _ptr(corr)))
        return codes, corr

    def close(self):
        if self._h:
            lib().bbq_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bytes_per_row(self):
        return lib().bbq_index_bytes_per_row(self._h)

    def set_option(self, name, value):
        _chk(lib().bbq_set_option(self._h, name.encode(), int(value)))

    def stats(self):
        s = Stats()
        _chk(lib().bbq_get_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def reset_stats(self):
        _chk(lib().bbq_reset_stats(self._h))

    def search(self, qquant, qcorr, query_bits, sim, k):
        idx, sc, n = self.search_batch(np.asarray(qquant)[None, :], np.asarray(qcorr)[None, :], query_bits, sim, k)
        return idx[0, :n[0]], sc[0, :n[0]]

    def search_batch(self, qquant, qcorr, query_bits, sim, k):
        qq = np.ascontiguousarray(qquant, np.uint8)
        qc = np.ascontiguousarray(qcorr, np.float64)
        nq = qq.shape[0]
        if nq and qq.shape[1] != self.dim:
            raise BBQError(ERR_DIM_MISMATCH, "查询向量维度与目标向量维度不匹配")
        kk = max(int(k), 0)
        idx = np.zeros((nq, kk), np.int32)
        sc = np.zeros((nq, kk), np.float32)
        cnt = np.zeros(nq, np.int64)
        _chk(lib().bbq_search_batch(self._h, nq, _ptr(qq), _ptr(qc), query_bits, sim, k, _ptr(idx), _ptr(sc), _ptr(cnt)))
        return idx, sc, cnt

    def search_raw_batch(self, queries, centroid, sim, query_bits, k, lam=0.1, iters=5, n_threads=0, want_quantized=False):
        """searchNearestNeighbors from raw fp32 queries [nq, dim]: quantization on host threads pipelined with the sweeps
        (bbq_search_raw_batch).  Returns (idx, score, count[, qquant, qcorr])."""
        q = np.ascontiguousarray(queries, np.float32)
        cen = np.ascontiguousarray(centroid, np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim or cen.shape != (self.dim,):
            raise BBQError(ERR_DIM_MISMATCH, "查询向量维度与目标向量维度不匹配")
        nq = q.shape[0]
        kk = max(int(k), 0)
        idx = np.zeros((nq, kk), np.int32)
        sc = np.zeros((nq, kk), np.float32)
        cnt = np.zeros(nq, np.int64)
        qq = np.zeros((nq, self.dim), np.uint8) if want_quantized else None
        qc = np.zeros((nq, 4), np.float64) if want_quantized else None
        bad = C.c_int32(-1)
        _chk(lib().bbq_search_raw_batch(self._h, nq, _ptr(q), _ptr(cen), sim, query_bits, lam, iters, n_threads, k, _ptr(idx), _ptr(sc), _ptr(cnt),
                                       _ptr(qq), _ptr(qc), C.byref(bad)))
        return (idx, sc, cnt, qq, qc) if want_quantized else (idx, sc, cnt)

    def score_rows(self, qquant, qcorr, query_bits, sim, row_begin=0, row_count=None):
        qq = np.ascontiguousarray(qquant, np.uint8)
        qc = np.ascontiguousarray(qcorr, np.float64)
        if row_count is None:
            row_count = self.n - row_begin
        d = np.zeros(row_count, np.int32)
        s64 = np.zeros(row_count, np.float64)
        s32 = np.zeros(row_count, np.float32)
        _chk(lib().bbq_score_rows(self._h, _ptr(qq), _ptr(qc), query_bits, sim, row_begin, row_count, _ptr(d), _ptr(s64), _ptr(s32)))
        return d, s64, s32

    def shard_scan_begin(self, qquant, qcorr, query_bits, sim, k, dev_packed_ptr, packed_cap, dev_offsets_ptr, dev_flags_ptr,
                         dev_answers_ptr=None, answers_stride=0):
        """enqueue the sweep of one batch (and its packing) and return; shard_scan_wait() later.  Two batches may be in flight."""
        qq = np.ascontiguousarray(qquant, np.uint8)
        qc = np.ascontiguousarray(qcorr, np.float64)
        _chk(lib().bbq_shard_scan_begin(self._h, qq.shape[0], _ptr(qq), _ptr(qc), query_bits, sim, k, dev_packed_ptr, packed_cap,
                                       dev_offsets_ptr, dev_flags_ptr, dev_answers_ptr, answers_stride))

    def shard_scan_wait(self):
        total = C.c_int64(0)
        _chk(lib().bbq_shard_scan_wait(self._h, C.byref(total)))
        return total.value

    def shard_list_cap(self, k):
        return lib().bbq_shard_list_cap(self._h, k)

    def shard_scan(self, qquant, qcorr, query_bits, sim, k, dev_packed_ptr, packed_cap, dev_offsets_ptr, dev_flags_ptr):
        """device pointers in (e.g. torch tensors' data_ptr()); returns the number of packed entries"""
        qq = np.ascontiguousarray(qquant, np.uint8)
        qc = np.ascontiguousarray(qcorr, np.float64)
        total = C.c_int64(0)
        _chk(lib().bbq_shard_scan(self._h, qq.shape[0], _ptr(qq), _ptr(qc), query_bits, sim, k, dev_packed_ptr, packed_cap,
                                 dev_offsets_ptr, dev_flags_ptr, C.byref(total)))
        return total.value


def merge_answers(blocks, n_queries, n_total, k, n_threads=1):
    """blocks[s]: uint64 array [>= n_queries, stride_s] of source s (bbq_shard_scan_begin's answers, in host memory).  Host only.
    Returns (idx [nq,k], score [nq,k], count [nq], status [nq]: 0 answered / 1 replay the lists / 2 dense path)."""
    bl = [np.ascontiguousarray(b, np.uint64) for b in blocks]
    n = len(bl)
    pp = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bl])
    strides = np.array([b.shape[1] for b in bl], np.int64)
    kk = max(int(k), 0)
    idx = np.zeros((n_queries, kk), np.int32)
    sc = np.zeros((n_queries, kk), np.float32)
    cnt = np.zeros(n_queries, np.int64)
    status = np.zeros(n_queries, np.uint8)
    _chk(lib().bbq_merge_answers(n, pp, _ptr(strides), n_queries, n_total, k, n_threads, _ptr(idx), _ptr(sc), _ptr(cnt), _ptr(status)))
    return idx, sc, cnt, status


def key_of_score(score):
    return int(lib().bbq_key_of_score(float(score)))


def file_shards(path_prefix):
    return int(lib().bbq_index_file_shards(os.fsencode(path_prefix)))


def file_info(path_prefix):
    n, dim, sim, cdp, rb = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_double(0), C.c_int64(0)
    _chk(lib().bbq_index_file_info(os.fsencode(path_prefix), C.byref(n), C.byref(dim), C.byref(sim), C.byref(cdp), C.byref(rb)))
    return {"n_rows": n.value, "dim": dim.value, "sim": sim.value, "centroid_dp": cdp.value, "row_base": rb.value}


class Vectors:
    """the original fp32 vectors resident on one GPU (bbq_vectors_*): the `vectors` argument of the reference's
    oversample selectors (src/topKSelector.ts:29-115)"""

    def __init__(self, vectors, device=0):
        v = np.ascontiguousarray(vectors, np.float32)
        if v.ndim != 2:
            raise BBQError(ERR_INVALID_ARG, "vectors must be [n, dim]")
        self.n, self.dim = int(v.shape[0]), int(v.shape[1])
        h = C.c_void_p()
        _chk(lib().bbq_vectors_create(_ptr(v), self.n, self.dim, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().bbq_vectors_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rerank_scores(self, queries, rows_per_query, true_sim=1):
        """computeSimilarity(queries[q], vectors[r]) for r in rows_per_query[q]; returns a list of f64 arrays"""
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise BBQError(ERR_DIM_MISMATCH, "向量维度不匹配")
        lists = [np.ascontiguousarray(r, np.int32) for r in rows_per_query]
        if len(lists) != q.shape[0]:
            raise BBQError(ERR_INVALID_ARG, "one candidate list per query")
        off = np.zeros(len(lists) + 1, np.int64)
        off[1:] = np.cumsum([a.shape[0] for a in lists])
        rows = np.concatenate(lists) if lists else np.zeros(0, np.int32)
        rows = np.ascontiguousarray(rows, np.int32)
        out = np.zeros(int(off[-1]), np.float64)
        _chk(lib().bbq_rerank_scores(self._h, q.shape[0], _ptr(q), _ptr(off), _ptr(rows), true_sim, _ptr(out)))
        return [out[off[i]:off[i + 1]] for i in range(len(lists))]


def search_rerank_batch(index, vectors, queries, qquant, qcorr, query_bits, sim, k, factor, selector=0, true_sim=1):
    """oversample k*factor on `index`, exact true scores on `vectors`, the reference's selector (0 heap, 1 sort).
    Returns (idx [nq,k], quantized f32 [nq,k], true f64 [nq,k], counts)."""
    q = np.ascontiguousarray(queries, np.float32)
    qq = np.ascontiguousarray(qquant, np.uint8)
    qc = np.ascontiguousarray(qcorr, np.float64)
    nq = qq.shape[0]
    if nq and (qq.shape[1] != index.dim or q.shape[1] != index.dim):
        raise BBQError(ERR_DIM_MISMATCH, "查询向量维度与目标向量维度不匹配")
    kk = max(int(k), 0)
    idx = np.zeros((nq, kk), np.int32)
    qs = np.zeros((nq, kk), np.float32)
    ts = np.zeros((nq, kk), np.float64)
    cnt = np.zeros(nq, np.int64)
    _chk(lib().bbq_search_rerank_batch(index._h, vectors._h, nq, _ptr(q), _ptr(qq), _ptr(qc), query_bits, sim, k, factor,
                                       selector, true_sim, _ptr(idx), _ptr(qs), _ptr(ts), _ptr(cnt)))
    return idx, qs, ts, cnt


def replay(lists, n_total, k):
    """lists: sequence of uint64 numpy arrays (ascending shard order).  Host only - needs no device."""
    arrs = [np.ascontiguousarray(a, np.uint64) for a in lists]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
    counts = np.array([a.shape[0] for a in arrs], np.int64)
    kk = max(int(k), 0)
    idx = np.zeros(kk, np.int32)
    sc = np.zeros(kk, np.float32)
    cnt = C.c_int64(0)
    _chk(lib().bbq_replay(n, ptrs, _ptr(counts), n_total, k, _ptr(idx), _ptr(sc), C.byref(cnt)))
    return idx[:cnt.value], sc[:cnt.value]


def replay_batch(packed, offsets, n_queries, n_total, k, n_threads=1):
    """packed[s]: uint64 array of source s (ascending shard order), offsets[s]: int64 [n_queries+1].  Host only."""
    pk = [np.ascontiguousarray(a, np.uint64) for a in packed]
    of = [np.ascontiguousarray(a, np.int64) for a in offsets]
    n = len(pk)
    pp = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in pk])
    po = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in of])
    kk = max(int(k), 0)
    idx = np.zeros((n_queries, kk), np.int32)
    sc = np.zeros((n_queries, kk), np.float32)
    cnt = np.zeros(n_queries, np.int64)
    _chk(lib().bbq_replay_batch(n, pp, po, n_queries, n_total, k, n_threads, _ptr(idx), _ptr(sc), _ptr(cnt)))
    return idx, sc, cnt
