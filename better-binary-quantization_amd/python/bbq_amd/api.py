"""Python mirror of the reference's public surface for the search path (same names, argument meaning and error
behaviour as src/index.ts:47-111 and src/binaryQuantizationFormat.ts:132-412), implemented entirely on libbbq."""
import weakref

import numpy as np

from . import capi


class VectorSimilarityFunction:  # src/types.ts:9-13 (string enum)
    EUCLIDEAN = "EUCLIDEAN"
    COSINE = "COSINE"
    MAXIMUM_INNER_PRODUCT = "MAXIMUM_INNER_PRODUCT"


DEFAULT_CONFIG = {"queryBits": 4, "indexBits": 1,
                  "quantizer": {"similarityFunction": VectorSimilarityFunction.COSINE, "lambda": 0.1, "iters": 5}}


class BinarizedByteVectorValues:
    """src/types.ts:32-49 / BinarizedByteVectorValuesImpl (src/binaryQuantizationFormat.ts:24-126): flat arrays with
    per-row views; the device copy is created lazily on first search and cached on the object."""

    def __init__(self, codes, corr, centroid, index_bits):
        self._codes, self._corr, self._centroid, self._index_bits = codes, corr, centroid, index_bits
        self._device_index = None

    def dimension(self):
        return int(self._centroid.shape[0])

    def size(self):
        return int(self._codes.shape[0])

    def vectorValue(self, ord_):
        if not 0 <= ord_ < self.size():
            raise Exception("向量索引 %d 不存在" % ord_)
        return self._codes[ord_]

    def getCorrectiveTerms(self, ord_):
        if not 0 <= ord_ < self.size():
            raise Exception("修正项索引 %d 不存在" % ord_)
        c = self._corr[ord_]
        return {"lowerInterval": float(c[0]), "upperInterval": float(c[1]), "additionalCorrection": float(c[2]),
                "quantizedComponentSum": float(c[3])}

    def getCentroid(self):
        return self._centroid

    def getCentroidDP(self, queryVector=None):
        if queryVector is not None:
            q = np.asarray(queryVector, np.float32).astype(np.float64)
            return float(np.sum(q * self._centroid.astype(np.float64)))  # unused by the search path
        return capi.centroid_dp(self._centroid)

    def _device(self, device=0):
        if self._device_index is None:
            self._device_index = capi.Index(self._codes, self._corr, self.dimension(), self.getCentroidDP(), device=device,
                                            index_bits=self._index_bits)
        return self._device_index


class BinaryQuantizationFormat:
    def __init__(self, config):
        qb = config.get("queryBits")
        ib = config.get("indexBits")
        if qb is not None and (qb < 1 or qb > 8):
            raise Exception("queryBits必须在1-8之间")
        if ib is not None and (ib < 1 or ib > 8):
            raise Exception("indexBits必须在1-8之间")
        self._config = {"queryBits": 4, "indexBits": 1}
        self._config.update(config)
        qz = config["quantizer"]
        self._sim = qz.get("similarityFunction", VectorSimilarityFunction.EUCLIDEAN)
        self._lambda = qz.get("lambda", 0.1)
        self._iters = qz.get("iters", 5)

    def getConfig(self):
        return self._config

    def quantizeVectors(self, vectors):
        if len(vectors) == 0:
            raise Exception("向量集合不能为空")
        dim = len(vectors[0])
        for i, v in enumerate(vectors):
            if len(v) != dim:
                raise Exception("向量 %d 维度 %d 与第一个向量维度 %d 不匹配" % (i, len(v), dim))
        try:
            if capi.device_count() > 0:
                # quantizeVectors as HIP kernels; the device index is ready when this returns
                ix, codes, corr, cen = capi.Index.build(np.asarray(vectors, np.float32), capi.SIMS[self._sim], self._lambda, self._iters,
                                                        index_bits=self._config["indexBits"])
                values = BinarizedByteVectorValues(codes, corr, cen, self._config["indexBits"])
                values._device_index = ix
            else:
                codes, corr, cen = capi.quantize_vectors(np.asarray(vectors, np.float32), capi.SIMS[self._sim],
                                                         self._config["indexBits"], self._lambda, self._iters)
                values = BinarizedByteVectorValues(codes, corr, cen, self._config["indexBits"])
        except capi.BBQError as e:
            raise Exception(str(e))
        return {"quantizedVectors": values, "queryQuantizer": self}

    def quantizeQueryVector(self, queryVector, centroid):
        qq, qc = capi.quantize_query(queryVector, centroid, capi.SIMS[self._sim], self._config["queryBits"], self._lambda,
                                     self._iters, search_path=False)
        return {"quantizedQuery": qq, "queryCorrections": {"lowerInterval": qc[0], "upperInterval": qc[1],
                                                          "additionalCorrection": qc[2], "quantizedComponentSum": qc[3]}}

    def searchNearestNeighbors(self, queryVector, targetVectors, k):
        if queryVector is None:
            raise Exception("查询向量不能为空")
        if targetVectors is None:
            raise Exception("目标向量集合不能为空")
        if k < 0:
            raise Exception("k值不能为负数")
        if len(queryVector) != targetVectors.dimension():
            raise Exception("查询向量维度与目标向量维度不匹配")
        if k == 0:
            return []
        sim = capi.SIMS[self._sim]
        if targetVectors._index_bits != 1 and targetVectors.dimension() > 1 and self._config["queryBits"] not in (1, 4):
            # the reference's batch scorer throws on a multi-bit index and its per-row fallback only knows 1- and 4-bit queries
            # (src/binaryQuantizedScorer.ts:95-97, :403-419); libbbq itself would score it (4-bit form, parity unpinned)
            raise Exception("不支持的查询位数: %d，只支持1位和4位" % self._config["queryBits"])
        try:
            qq, qc = capi.quantize_query(queryVector, targetVectors.getCentroid(), sim, self._config["queryBits"], self._lambda,
                                         self._iters, search_path=True)
            idx, sc = targetVectors._device().search(qq, qc, self._config["queryBits"], sim, k)
        except capi.BBQError as e:
            raise Exception(str(e))
        return [{"index": int(i), "score": float(s)} for i, s in zip(idx, sc)]


def createBinaryQuantizationFormat(config=DEFAULT_CONFIG):
    return BinaryQuantizationFormat(config)


def quickQuantize(vectors, similarityFunction=VectorSimilarityFunction.COSINE):
    return BinaryQuantizationFormat({"quantizer": {"similarityFunction": similarityFunction, "lambda": 0.1, "iters": 5}}).quantizeVectors(vectors)


def quickSearch(queryVector, targetVectors, k, similarityFunction=VectorSimilarityFunction.COSINE):
    f = BinaryQuantizationFormat({"quantizer": {"similarityFunction": similarityFunction, "lambda": 0.1, "iters": 5}})
    return f.searchNearestNeighbors(queryVector, f.quantizeVectors(targetVectors)["quantizedVectors"], k)


def createDeviceVectors(vectors, device=0):
    """upload the original fp32 vectors once; pass the handle wherever the reference's selectors take `vectors`"""
    return capi.Vectors(np.asarray(vectors, np.float32), device)


def _oversampled(query, quantizedVectors, vectors, k, oversampleFactor, fmt, selector):
    # src/topKSelector.ts:29-115: searchNearestNeighbors(k*factor) -> computeCosineSimilarity per candidate -> select.
    # Search, true scores and selection all happen behind bbq_search_rerank_batch.
    oversampled = k * oversampleFactor
    if oversampled < 0:
        raise Exception("k值不能为负数")
    if len(query) != quantizedVectors.dimension():
        raise Exception("查询向量维度与目标向量维度不匹配")
    if oversampled == 0:
        return []
    owned = not isinstance(vectors, capi.Vectors)
    dv = createDeviceVectors(vectors) if owned else vectors
    try:
        sim = capi.SIMS[fmt._sim]
        q = np.ascontiguousarray(query, np.float32)
        qq, qc = capi.quantize_query(q, quantizedVectors.getCentroid(), sim, fmt._config["queryBits"], fmt._lambda, fmt._iters,
                                     search_path=True)
        idx, qs, ts, cnt = capi.search_rerank_batch(quantizedVectors._device(), dv, q[None, :], qq[None, :], qc[None, :],
                                                    fmt._config["queryBits"], sim, k, oversampleFactor, selector, 1)
    except capi.BBQError as e:
        raise Exception(str(e))
    finally:
        if owned:
            dv.close()
    return [{"index": int(idx[0, j]), "quantizedScore": float(qs[0, j]), "trueScore": float(ts[0, j])} for j in range(int(cnt[0]))]


def getOversampledTopKWithHeap(query, quantizedVectors, vectors, k, oversampleFactor, format):
    return _oversampled(query, quantizedVectors, vectors, k, oversampleFactor, format, 0)


def getOversampledTopKWithSort(query, quantizedVectors, vectors, k, oversampleFactor, format):
    return _oversampled(query, quantizedVectors, vectors, k, oversampleFactor, format, 1)
