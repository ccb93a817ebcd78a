'use strict';
/*
 * Host-side drop-in for the search path of leolee9086/Better-Binary-Quantization on MI355X.
 *
 * Same public surface as the reference's src/index.ts (createBinaryQuantizationFormat, quickQuantize, quickSearch,
 * BinaryQuantizationFormat, VectorSimilarityFunction, DEFAULT_CONFIG, VERSION) with the same argument meaning,
 * defaults and error messages; every numeric operation runs in libbbq (C++ host quantizer + gfx950 HIP kernels)
 * through the N-API addon ../lib/bbq_napi.node.  There is no JavaScript/CPU fallback for the scan: without a HIP
 * device searchNearestNeighbors throws.
 *
 * Plain ES2019 CommonJS (runs on Node >= 12 without a build step); index.d.ts carries the TypeScript surface.
 */
const path = require('path');
const native = require(path.join(__dirname, '..', 'lib', 'bbq_napi.node'));

// src/types.ts:9-13 (string enum)
const VectorSimilarityFunction = Object.freeze({
  EUCLIDEAN: 'EUCLIDEAN', COSINE: 'COSINE', MAXIMUM_INNER_PRODUCT: 'MAXIMUM_INNER_PRODUCT',
});
const SIM_ORDINAL = { EUCLIDEAN: 0, COSINE: 1, MAXIMUM_INNER_PRODUCT: 2 };

// src/constants.ts:9-30
const QUERY_BITS = 4;
const INDEX_BITS = 1;
const FOUR_BIT_SCALE = 1.0 / ((1 << 4) - 1);
const DEFAULT_LAMBDA = 0.1;
const DEFAULT_ITERS = 5;

// src/index.ts:47-55
const DEFAULT_CONFIG = Object.freeze({
  queryBits: 4, indexBits: 1,
  quantizer: Object.freeze({ similarityFunction: VectorSimilarityFunction.COSINE, lambda: 0.1, iters: 5 }),
});

function simOrdinal(sim) {
  const o = SIM_ORDINAL[sim];
  if (o === undefined) throw new Error('不支持的相似性函数: ' + sim);
  return o;
}

function corrObject(f64, off) {
  return {
    lowerInterval: f64[off], upperInterval: f64[off + 1], additionalCorrection: f64[off + 2],
    quantizedComponentSum: f64[off + 3],
  };
}

/**
 * BinarizedByteVectorValues (src/types.ts:32-49; BinarizedByteVectorValuesImpl, src/binaryQuantizationFormat.ts:24-126).
 * One flat Uint8Array / Float64Array instead of per-row objects; vectorValue(ord) is a subarray view, which is
 * API-compatible with the reference's per-row Uint8Array.  The device copy is created on first search and cached.
 */
class BinarizedByteVectorValuesImpl {
  constructor(codes, corr, centroid, indexBits, size) {
    this._codes = codes; this._corr = corr; this._centroid = centroid; this._indexBits = indexBits;
    this._size = size; this._rowBytes = (size > 0 && codes) ? codes.length / size : 0;
    this._device = null;
  }
  dimension() { return this._centroid.length; }
  size() { return this._size; }
  /** host copies of the rows; an index loaded from disk fetches them from the device on first use (bbq_index_export) */
  _host() {
    if (!this._codes && this._device) {
      const r = native.indexExport(this._device);
      this._codes = r.codes; this._corr = r.corr;
    }
    return this;
  }
  vectorValue(ord) {
    if (!(ord >= 0 && ord < this._size)) throw new Error('向量索引 ' + ord + ' 不存在');
    return this._host()._codes.subarray(ord * this._rowBytes, (ord + 1) * this._rowBytes);
  }
  getUnpackedVector(ord) {
    if (!(ord >= 0 && ord < this._size)) throw new Error('未打包向量索引 ' + ord + ' 不存在');
    const dim = this.dimension();
    if (this._indexBits !== 1) return new Uint8Array(this.vectorValue(ord));
    const row = this.vectorValue(ord), out = new Uint8Array(dim);
    for (let d = 0; d < dim; d++) out[d] = (row[d >> 3] >> (7 - (d & 7))) & 1;
    return out;
  }
  clearUnpackedVectorCache() {}
  getCorrectiveTerms(ord) {
    if (!(ord >= 0 && ord < this._size)) throw new Error('修正项索引 ' + ord + ' 不存在');
    return corrObject(this._host()._corr, 4 * ord);
  }
  getCentroidDP(queryVector) {
    if (queryVector) {  // computeDotProduct(queryVector, centroid), src/vectorOperations.ts:171-185
      if (queryVector.length !== this._centroid.length) throw new Error('向量维度不匹配');
      let s = 0;
      for (let i = 0; i < queryVector.length; i++) s += queryVector[i] * this._centroid[i];
      return s;
    }
    return native.centroidDP(this._centroid);
  }
  getCentroid() { return this._centroid; }
  /** device-resident copy (libbbq bbq_index); created lazily, released by dispose() or GC */
  _deviceIndex() {
    if (!this._device) {
      if (!this._codes) throw new Error('目标向量集合不能为空');
      const devices = shardDevices(this._size);
      if (devices.length > 1) {
        // one index row-sharded over several GPUs behind the same handle: every search below uses all of them
        this._device = native.indexCreateMulti(this._codes, this._corr, this._size, this.dimension(), this._indexBits, this.getCentroidDP(),
          Int32Array.from(devices), Number(process.env.BBQ_PILOT_ROWS || 32768));
      } else {
        this._device = native.indexCreate(this._codes, this._corr, this._size, this.dimension(), this._indexBits,
          this.getCentroidDP(), devices[0]);
      }
    }
    return this._device;
  }
  dispose() { if (this._device) { this._host(); native.indexDestroy(this._device); this._device = null; } }
  /** tuning knobs of the device index (libbbq bbq_set_option), e.g. ('sweep_share', 32) for searchNearestNeighborsBatch */
  setDeviceOption(name, value) { native.setOption(this._deviceIndex(), name, value); }
  deviceStats() { return native.stats(this._deviceIndex()); }
}

/**
 * Devices an index of `rows` rows is sharded over.  Sharding is OPT-IN: BBQ_DEVICES = a list of HIP ordinals ("0,1,2,3"; repeats
 * allowed: several shards on one GPU) or a count ("8" = devices 0..7), applied to indexes of at least BBQ_SHARD_MIN_ROWS rows
 * (default 0: every index).  Unset: BBQ_DEVICE (default 0) alone - one GPU answers a single 10 M-row query in 0.2 ms, and the
 * multi-GPU path has not been measured on more than one physical device yet (bench.py reports it when it has).
 */
function shardDevices(rows) {
  const one = [Number(process.env.BBQ_DEVICE || 0)];
  const spec = process.env.BBQ_DEVICES;
  if (spec !== undefined && spec !== '' && rows >= Number(process.env.BBQ_SHARD_MIN_ROWS || 0)) {
    if (spec.indexOf(',') >= 0) return spec.split(',').map(Number);
    const n = Number(spec);
    if (!(n >= 1)) return one;
    if (n === 1) return one;
    const out = []; for (let i = 0; i < n; i++) out.push(i);
    return out;
  }
  return one;
}

/** time the batch call has spent inside the addon since the process started (tests/js/bench_scale.js prices the host's own share with it) */
const hostClock = { insideAddonNs: 0n };

function flatten(vectors, dim) {
  const flat = new Float32Array(vectors.length * dim);
  for (let i = 0; i < vectors.length; i++) flat.set(vectors[i], i * dim);
  return flat;
}

/** the quantizer handle the reference returns as `queryQuantizer` (OptimizedScalarQuantizer) */
const transposeCache = { map: new WeakMap(), hits: 0, misses: 0 };

class OptimizedScalarQuantizer {
  constructor(config) {
    this.lambda = (config.lambda !== undefined && config.lambda !== null) ? config.lambda : DEFAULT_LAMBDA;
    this.iters = (config.iters !== undefined && config.iters !== null) ? config.iters : DEFAULT_ITERS;
    this.similarityFunction = (config.similarityFunction !== undefined && config.similarityFunction !== null)
      ? config.similarityFunction : VectorSimilarityFunction.EUCLIDEAN;
  }
  /** scalarQuantize(vector, destination, bits, centroid), src/optimizedScalarQuantizer.ts:108-227 */
  scalarQuantize(vector, destination, bits, centroid) {
    if (!vector) throw new Error('输入向量不能为空');
    if (!destination) throw new Error('目标数组不能为空');
    if (!centroid) throw new Error('质心向量不能为空');
    if (vector.length !== centroid.length) throw new Error('向量和质心维度不匹配');
    if (destination.length !== vector.length) throw new Error('目标数组长度与向量长度不匹配');
    if (bits < 1 || bits > 8) throw new Error('位数必须在1-8之间');
    // a sim that never normalises gives the bare scalarQuantize; EUCLIDEAN/MIP ordinals keep their correction rule
    const sim = this.similarityFunction === VectorSimilarityFunction.EUCLIDEAN ? 0 : 2;
    const r = native.quantizeQuery(Float32Array.from(vector), centroid, sim, bits, this.lambda, this.iters, false);
    destination.set(r.quantizedQuery);
    return corrObject(r.corrections, 0);
  }
  /** multiScalarQuantize(vector, destinations, bits[], centroid): one scalarQuantize per (destination, bits) pair, src/optimizedScalarQuantizer.ts:67-93 */
  multiScalarQuantize(vector, destinations, bits, centroid) {
    if (destinations.length !== bits.length) throw new Error('目标数组和位数数组长度不匹配');
    const results = [];
    for (let i = 0; i < destinations.length; i++) {
      if (destinations[i] && bits[i] !== undefined) results.push(this.scalarQuantize(vector, destinations[i], bits[i], centroid));
    }
    return results;
  }
  /** discretize(value, bucket): value rounded up to a multiple of bucket, :460-463 */
  static discretize(value, bucket) { return Math.floor((value + (bucket - 1)) / bucket) * bucket; }
  /**
   * transposeHalfByte(q, quantQueryByte): the four bit-planes of a 4-bit query, ONE BYTE PER BIT (plane p of dimension i at
   * i + p * q.length), :476-517.  (The device keeps the planes packed eight dimensions to a byte: bbq_core.cpp fill_query.)
   */
  static transposeHalfByte(q, quantQueryByte) {
    if (!q || !quantQueryByte) throw new Error('输入数组不能为空');
    const n = q.length;
    if (quantQueryByte.length !== n * 4) throw new Error('转置数组长度不正确，期望' + n * 4 + '，实际' + quantQueryByte.length);
    quantQueryByte.fill(0);
    for (let i = 0; i < n; i++) {
      const v = q[i];
      if (v === undefined || v < 0 || v > 15) throw new Error('4位量化值必须在0-15之间');
      for (let p = 0; p < 4; p++) quantQueryByte[i + p * n] = (v >> p) & 1;
    }
  }
  /** transposeHalfByteOptimized(q, quantQueryByte, useCache = true): the same through a cache keyed by the query ARRAY (identity), :528-552 */
  static transposeHalfByteOptimized(q, quantQueryByte, useCache) {
    const cache = useCache === undefined ? true : useCache;
    if (cache) {
      const hit = transposeCache.map.get(q);
      if (hit) { transposeCache.hits++; quantQueryByte.set(hit); return; }
      transposeCache.misses++;
    }
    OptimizedScalarQuantizer.transposeHalfByte(q, quantQueryByte);
    if (cache) transposeCache.map.set(q, new Uint8Array(quantQueryByte));
  }
  /** transposeHalfByteFast(q, quantQueryByte): PACKED planes, eight dimensions per byte, MSB first, plane size = quantQueryByte.length / 4; no validation, :561-592 */
  static transposeHalfByteFast(q, quantQueryByte) {
    quantQueryByte.fill(0);
    const n = q.length, planeSize = quantQueryByte.length / 4;
    for (let g = 0; g * 8 < n; g++) {
      const planes = [0, 0, 0, 0];
      for (let j = 0; j < 8 && g * 8 + j < n; j++) {
        const v = q[g * 8 + j];
        for (let p = 0; p < 4; p++) planes[p] |= ((v >> p) & 1) << (7 - j);
      }
      for (let p = 0; p < 4; p++) quantQueryByte[g + p * planeSize] = planes[p];
    }
  }
  static clearTransposeCache() { transposeCache.map = new WeakMap(); transposeCache.hits = 0; transposeCache.misses = 0; }
  /** {size: 0 (a WeakMap has none), hitRate}, :619-627 */
  static getTransposeCacheStats() {
    const total = transposeCache.hits + transposeCache.misses;
    return { size: 0, hitRate: total > 0 ? transposeCache.hits / total : 0 };
  }
  /**
   * packAsBinary(vector, packed): dimension d -> byte d >> 3, bit 7 - (d & 7)  (the packing of src/optimizedScalarQuantizer.ts:420-446; same
   * errors in the same order: a value other than 0 / 1 is reported before a destination that is too short for its byte)
   */
  static packAsBinary(vector, packed) {
    const n = vector.length;
    for (let byte = 0, d = 0; d < n; byte++) {
      const stop = Math.min(d + 8, n);
      let bits = 0;
      for (let shift = 7; d < stop; d++, shift--) {
        const v = vector[d];
        if (v !== 0 && v !== 1) throw new Error('1位量化值必须为0或1');
        bits |= v << shift;
      }
      if (byte >= packed.length) throw new Error('打包数组长度不足');
      packed[byte] = bits;
    }
  }
}

/** BinaryQuantizedScorer as far as the search path uses it (src/binaryQuantizedScorer.ts:315-420) */
class BinaryQuantizedScorer {
  constructor(similarityFunction) { this.similarityFunction = similarityFunction; }
  /**
   * computeQuantizedScore(quantizedQuery, queryCorrections, targetVectors, targetOrd, queryBits, originalQueryVector?)
   * src/binaryQuantizedScorer.ts:69-301: the single-row path the reference falls back to.  Host arithmetic only (one row);
   * for 4-bit queries it is defined differently from the batch path: centroidDP = query . centroid if the original query is
   * given and 0 otherwise, and MAXIMUM_INNER_PRODUCT is scaled without FOUR_BIT_SCALE.
   */
  computeQuantizedScore(quantizedQuery, queryCorrections, targetVectors, targetOrd, queryBits, originalQueryVector) {
    if (queryBits !== 1 && queryBits !== 4) throw new Error('不支持的查询位数: ' + queryBits + '，只支持1位和4位');
    const qcDist = computeQuantizedDotProduct(quantizedQuery, targetVectors.getUnpackedVector(targetOrd));
    const ic = targetVectors.getCorrectiveTerms(targetOrd), dimension = targetVectors.dimension();
    const x1 = ic.quantizedComponentSum, ax = ic.lowerInterval, lx = ic.upperInterval - ax;
    const ay = queryCorrections.lowerInterval, y1 = queryCorrections.quantizedComponentSum;
    const ly = queryBits === 1 ? queryCorrections.upperInterval - ay : (queryCorrections.upperInterval - ay) * FOUR_BIT_SCALE;
    const centroidDP = queryBits === 1 ? targetVectors.getCentroidDP() : (originalQueryVector ? targetVectors.getCentroidDP(originalQueryVector) : 0);
    let score = ax * ay * dimension + ay * lx * x1 + ax * ly * y1 + lx * ly * qcDist;
    const sim = this.similarityFunction;
    if (sim === VectorSimilarityFunction.EUCLIDEAN) {
      score = queryCorrections.additionalCorrection + ic.additionalCorrection - 2 * score;
      score = Math.max(1 / (1 + score), 0);
    } else if (sim === VectorSimilarityFunction.COSINE || sim === VectorSimilarityFunction.MAXIMUM_INNER_PRODUCT) {
      if (queryBits === 1) score += queryCorrections.additionalCorrection + ic.additionalCorrection - centroidDP;
      else score = score + queryCorrections.additionalCorrection + ic.additionalCorrection - centroidDP;
      score = sim === VectorSimilarityFunction.COSINE ? Math.max((1 + score) / 2, 0) : (score < 0 ? 1 / (1 - score) : score + 1);
    } else throw new Error('不支持的相似性函数: ' + sim);
    return { score: score, bitDotProduct: qcDist, corrections: { query: queryCorrections, index: ic } };
  }

  /**
   * computeBatchQuantizedScores(quantizedQuery, queryCorrections, targetVectors, targetOrds, queryBits, originalQueryVector?)
   * src/binaryQuantizedScorer.ts:315-420.  Scores come from the device (bbq_score_rows).  Two corners follow the reference on the
   * host, row by row:
   *  - originalQueryVector given with a multi-bit query on a 1-bit index: centroidDP = query . centroid instead of centroid . centroid
   *    (:372-381; searchNearestNeighbors never passes it) - the device's integer qcDist, the batch formula restated here;
   *  - a multi-bit index: the reference's batch kernels throw on its unpacked rows and it falls back to computeQuantizedScore per
   *    row with one console.warn per call (:403-419) - the device already returns exactly those per-row values; the warning is
   *    kept, and queryBits other than 1 and 4 throw like the fallback does (:95-97).
   */
  computeBatchQuantizedScores(quantizedQuery, queryCorrections, targetVectors, targetOrds, queryBits, originalQueryVector) {
    if (targetOrds.length === 0) return [];
    const multibit = isMultiBit(targetVectors);
    if (multibit) {
      console.warn('批量计算失败，回退到原始方法:', new RangeError('offset is out of bounds'));
      if (queryBits !== 1 && queryBits !== 4) throw new Error('不支持的查询位数: ' + queryBits + '，只支持1位和4位');
      if (originalQueryVector && queryBits !== 1) {
        const self = this;
        return targetOrds.map(function (ord) {
          return self.computeQuantizedScore(quantizedQuery, queryCorrections, targetVectors, ord, queryBits, originalQueryVector);
        });
      }
    }
    const qc = new Float64Array([queryCorrections.lowerInterval, queryCorrections.upperInterval,
      queryCorrections.additionalCorrection, queryCorrections.quantizedComponentSum]);
    let lo = targetOrds[0], hi = targetOrds[0];
    for (let i = 1; i < targetOrds.length; i++) { if (targetOrds[i] < lo) lo = targetOrds[i]; if (targetOrds[i] > hi) hi = targetOrds[i]; }
    const r = native.scoreRows(targetVectors._deviceIndex(), quantizedQuery, qc, queryBits, simOrdinal(this.similarityFunction), lo, hi - lo + 1);
    const withQuery = !multibit && originalQueryVector && queryBits !== 1;
    const cdp = withQuery ? targetVectors.getCentroidDP(originalQueryVector) : 0, sim = this.similarityFunction, dimension = targetVectors.dimension();
    return targetOrds.map(function (ord) {
      const ic = targetVectors.getCorrectiveTerms(ord);
      let score = r.score64[ord - lo];
      if (withQuery) score = fourBitBatchScore(r.qcDist[ord - lo], queryCorrections, ic, dimension, cdp, sim);
      return { score: score, bitDotProduct: r.qcDist[ord - lo], corrections: { query: queryCorrections, index: ic } };
    });
  }
  /** computeOriginalScore(originalQuery, targetVector, similarityFunction): computeSimilarity on the fp32 vectors (:429-447) */
  computeOriginalScore(originalQuery, targetVector, similarityFunction) {
    if (similarityFunction !== VectorSimilarityFunction.EUCLIDEAN && similarityFunction !== VectorSimilarityFunction.COSINE &&
        similarityFunction !== VectorSimilarityFunction.MAXIMUM_INNER_PRODUCT) throw new Error('不支持的相似性函数: ' + similarityFunction);
    return computeSimilarity(originalQuery, targetVector, similarityFunction);
  }

  /** compareScores(originalScore, quantizedScore) -> {difference, relativeError, correlation}  (:455-477, correlation of two single values :485-513) */
  compareScores(originalScore, quantizedScore) {
    const difference = Math.abs(originalScore - quantizedScore);
    const relativeError = originalScore === 0 ? (quantizedScore === 0 ? 0 : Infinity) : difference / Math.abs(originalScore);
    // the reference "correlates" two single values: 1 when they are equal, 0 when exactly one is zero, otherwise a quotient of
    // deviations from themselves - 0 for finite values (0 / 0 is caught), NaN when a value is not finite (x - x is NaN then)
    let correlation = 0;
    if (originalScore === quantizedScore) correlation = 1;
    else if (originalScore !== 0 && quantizedScore !== 0) {
      const da = originalScore - originalScore, db = quantizedScore - quantizedScore, den = Math.abs(da) * Math.abs(db);
      correlation = den === 0 ? 0 : (da * db) / den;
    }
    return { difference: difference, relativeError: relativeError, correlation: correlation };
  }

  /**
   * computeQuantizationAccuracy(originalScores, quantizedScores) -> {meanError, maxError, minError, stdError, correlation}
   * (what src/binaryQuantizedScorer.ts:524-617 reports; every sum runs left to right over the pairs, so the floats are the reference's).
   * Pairs with an undefined member are skipped by the error statistics and by the sums of the correlation, whose n stays the array length.
   */
  computeQuantizationAccuracy(originalScores, quantizedScores) {
    const n = originalScores.length;
    if (n !== quantizedScores.length) throw new Error('原始分数和量化分数数组长度不匹配');
    const defined = [];
    for (let i = 0; i < n; i++) if (originalScores[i] !== undefined && quantizedScores[i] !== undefined) defined.push(i);
    const abs = defined.map(function (i) { return Math.abs(originalScores[i] - quantizedScores[i]); });
    const total = abs.reduce(function (acc, e) { return acc + e; }, 0);
    const meanError = total / abs.length;
    const spread = abs.reduce(function (acc, e) { return acc + (e - meanError) * (e - meanError); }, 0);
    // Pearson's r from the five running sums
    const m = { x: 0, y: 0, xy: 0, xx: 0, yy: 0 };
    for (const i of defined) {
      const a = originalScores[i], b = quantizedScores[i];
      m.x += a; m.y += b; m.xy += a * b; m.xx += a * a; m.yy += b * b;
    }
    const cov = n * m.xy - m.x * m.y, norm = Math.sqrt((n * m.xx - m.x * m.x) * (n * m.yy - m.y * m.y));
    return {
      meanError: meanError,
      maxError: abs.reduce(function (acc, e) { return Math.max(acc, e); }, 0),
      minError: abs.reduce(function (acc, e) { return Math.min(acc, e); }, Infinity),
      stdError: Math.sqrt(spread / abs.length),
      correlation: norm === 0 ? 0 : cov / norm,
    };
  }

  getSimilarityFunction() { return this.similarityFunction; }
}

/** rows are unpacked bytes and the reference's batch scorer throws on them (dimension 1 is the one width where it does not) */
function isMultiBit(targetVectors) { return targetVectors._indexBits !== 1 && targetVectors.dimension() > 1; }

/** computeBatchFourBitSimilarityScores for one row, src/batchDotProduct.ts:554-617 (used only for the originalQueryVector corner) */
function fourBitBatchScore(qcDist, q, x, dimension, centroidDP, sim) {
  const x1 = x.quantizedComponentSum, ax = x.lowerInterval, lx = x.upperInterval - ax;
  const ay = q.lowerInterval, ly = (q.upperInterval - ay) * FOUR_BIT_SCALE, y1 = q.quantizedComponentSum;
  let score = ax * ay * dimension + ay * lx * x1 + ax * ly * y1 + lx * ly * qcDist;
  if (sim === VectorSimilarityFunction.EUCLIDEAN) {
    score = q.additionalCorrection + x.additionalCorrection - 2 * score;
    return Math.max(1 / (1 + score), 0);
  }
  score = score + q.additionalCorrection + x.additionalCorrection - centroidDP;
  if (sim === VectorSimilarityFunction.COSINE) return Math.max((1 + score) / 2, 0);
  if (sim === VectorSimilarityFunction.MAXIMUM_INNER_PRODUCT) return score < 0 ? 1 / (1 - score / FOUR_BIT_SCALE) : score / FOUR_BIT_SCALE + 1;
  throw new Error('不支持的相似性函数: ' + sim);
}

/** BinaryQuantizationFormat, src/binaryQuantizationFormat.ts:132-412 */
class BinaryQuantizationFormat {
  constructor(config) {
    if (config.queryBits !== undefined && (config.queryBits < 1 || config.queryBits > 8)) throw new Error('queryBits必须在1-8之间');
    if (config.indexBits !== undefined && (config.indexBits < 1 || config.indexBits > 8)) throw new Error('indexBits必须在1-8之间');
    this.config = Object.assign({ queryBits: QUERY_BITS, indexBits: INDEX_BITS }, config);
    this.quantizer = new OptimizedScalarQuantizer(config.quantizer);
    this.scorer = new BinaryQuantizedScorer(config.quantizer.similarityFunction);
  }

  /** quantizeVectors(vectors: Float32Array[]) -> {quantizedVectors, queryQuantizer}  (:165-263) */
  quantizeVectors(vectors) {
    if (vectors.length === 0) throw new Error('向量集合不能为空');
    const first = vectors[0];
    if (!first) throw new Error('第一个向量不能为空');
    const dim = first.length;
    for (let i = 1; i < vectors.length; i++) {
      const v = vectors[i];
      if (!v) throw new Error('向量 ' + i + ' 不能为空');
      if (v.length !== dim) throw new Error('向量 ' + i + ' 维度 ' + v.length + ' 与第一个向量维度 ' + dim + ' 不匹配');
    }
    const q = this.quantizer;
    let values;
    if (process.env.BBQ_HOST_QUANTIZER !== '1' && native.deviceCount() > 0) {
      // quantizeVectors as HIP kernels (any indexBits); the device index is ready when this returns (no second upload on first search)
      const r = native.indexBuild(flatten(vectors, dim), vectors.length, dim, simOrdinal(q.similarityFunction), q.lambda, q.iters,
        Number(process.env.BBQ_DEVICE || 0), this.config.indexBits);
      values = new BinarizedByteVectorValuesImpl(r.codes, r.corr, r.centroid, this.config.indexBits, vectors.length);
      if (shardDevices(vectors.length).length > 1) native.indexDestroy(r.handle);  // the sharded index is created from the rows on first search
      else values._device = r.handle;
    } else {
      const r = native.quantizeVectors(flatten(vectors, dim), vectors.length, dim, simOrdinal(q.similarityFunction),
        this.config.indexBits, q.lambda, q.iters, Number(process.env.BBQ_THREADS || 0));
      values = new BinarizedByteVectorValuesImpl(r.codes, r.corr, r.centroid, this.config.indexBits, vectors.length);
    }
    return { quantizedVectors: values, queryQuantizer: this.quantizer };
  }

  /** quantizeQueryVector(queryVector, centroid) -> {quantizedQuery, queryCorrections}  (:271-299) */
  quantizeQueryVector(queryVector, centroid) {
    const q = this.quantizer;
    const r = native.quantizeQuery(Float32Array.from(queryVector), centroid, simOrdinal(q.similarityFunction), this.config.queryBits,
      q.lambda, q.iters, false);
    return { quantizedQuery: r.quantizedQuery, queryCorrections: corrObject(r.corrections, 0) };
  }

  /** searchNearestNeighbors(queryVector, targetVectors, k) -> Array<{index, score}>  (:308-412) */
  searchNearestNeighbors(queryVector, targetVectors, k) {
    if (!queryVector) throw new Error('查询向量不能为空');
    if (!targetVectors) throw new Error('目标向量集合不能为空');
    if (k < 0) throw new Error('k值不能为负数');
    if (queryVector.length !== targetVectors.dimension()) throw new Error('查询向量维度与目标向量维度不匹配');
    if (k === 0) return [];
    if (isMultiBit(targetVectors) || !native.searchRawInto) return this.searchNearestNeighborsBatch([queryVector], targetVectors, k)[0];
    // the one-query call allocates nothing but its result: the query goes to the addon as it is (or through one staging array kept by
    // this format when it is not a Float32Array), the answer comes back in two arrays kept per format
    const dim = queryVector.length, q = this.quantizer;
    let flat = queryVector;
    if (!(queryVector instanceof Float32Array)) {
      if (!this._stage || this._stage.length !== dim) this._stage = new Float32Array(dim);
      flat = this._stage;
      flat.set(queryVector);
    }
    if (!this._outIdx || this._outIdx.length < k) {
      this._outIdx = new Int32Array(Math.max(k, 128));
      this._outScore = new Float32Array(Math.max(k, 128));
    }
    const tNative = process.hrtime.bigint();
    const n = native.searchRawInto(targetVectors._deviceIndex(), flat, targetVectors.getCentroid(), simOrdinal(q.similarityFunction), this.config.queryBits,
      q.lambda, q.iters, k, this._outIdx, this._outScore);
    hostClock.insideAddonNs += process.hrtime.bigint() - tNative;
    const indices = this._outIdx, scores = this._outScore, res = new Array(n);
    for (let j = 0; j < n; j++) res[j] = { index: indices[j], score: scores[j] };
    return res;
  }

  /** extension (not in the reference): many queries per call, pipelined on the device; each query sweeps the index itself */
  searchNearestNeighborsBatch(queryVectors, targetVectors, k) {
    if (!queryVectors) throw new Error('查询向量不能为空');
    if (!targetVectors) throw new Error('目标向量集合不能为空');
    if (k < 0) throw new Error('k值不能为负数');
    const dim = targetVectors.dimension(), nq = queryVectors.length;
    if (k === 0) return queryVectors.map(function () { return []; });
    const q = this.quantizer, sim = simOrdinal(q.similarityFunction);
    if (isMultiBit(targetVectors) && nq > 0) {
      // the reference answers a multi-bit index through its per-row fallback, warning once per batch of 1000 rows
      // (src/binaryQuantizedScorer.ts:403-405), and throws for queryBits its fallback does not know (:95-97); one warning per call here
      console.warn('批量计算失败，回退到原始方法:', new RangeError('offset is out of bounds'));
      if (this.config.queryBits !== 1 && this.config.queryBits !== 4) throw new Error('不支持的查询位数: ' + this.config.queryBits + '，只支持1位和4位');
    }
    const flat = new Float32Array(nq * dim);
    for (let i = 0; i < nq; i++) {
      const v = queryVectors[i];
      if (!v) throw new Error('查询向量不能为空');
      if (v.length !== dim) throw new Error('查询向量维度与目标向量维度不匹配');
      flat.set(v, i * dim);
    }
    // searchNearestNeighbors normalises for COSINE and quantizeQueryVector normalises again (:337-347, :279-281); both happen behind
    // bbq_search_raw_batch, on host threads, chunk by chunk while the sub-batches in front are already on the device (one 768-d query
    // costs ~15 us on one core, more than its sweep of 1 M rows on the device)
    const tNative = process.hrtime.bigint();
    const r = native.searchRawBatch(targetVectors._deviceIndex(), nq, flat, targetVectors.getCentroid(), sim, this.config.queryBits, q.lambda, q.iters,
      Number(process.env.BBQ_THREADS || 0), k);
    hostClock.insideAddonNs += process.hrtime.bigint() - tNative;
    const out = new Array(nq), indices = r.indices, scores = r.scores, stride = r.stride;
    for (let i = 0; i < nq; i++) {
      const n = r.counts[i], base = i * stride, res = new Array(n);
      for (let j = 0; j < n; j++) res[j] = { index: indices[base + j], score: scores[base + j] };
      out[i] = res;
    }
    return out;
  }

  /**
   * serializeVectorData(vectors) -> {vectorData: VectorDataFormat[], metadata: MetadataFormat}  (:483-530, src/types.ts:78-113).
   * The reference re-packs the already packed rows here and throws for any byte outside {0,1} (SURVEY §5); this returns
   * what it meant to: binaryValues = the packed row, the four corrections, and the metadata record.
   */
  serializeVectorData(vectors) {
    const qv = this.quantizeVectors(vectors).quantizedVectors, centroid = qv.getCentroid();
    const vectorData = [];
    for (let i = 0; i < qv.size(); i++) {
      const c = qv.getCorrectiveTerms(i);
      vectorData.push({ binaryValues: new Uint8Array(qv.vectorValue(i)), lowerInterval: c.lowerInterval, upperInterval: c.upperInterval,
        additionalCorrection: c.additionalCorrection, quantizedComponentSum: c.quantizedComponentSum });
    }
    const metadata = { fieldNumber: 0, vectorEncodingOrdinal: 0, vectorSimilarityOrdinal: 0, dimensions: centroid.length,
      vectorDataOffset: 0, vectorDataLength: 0, vectorCount: qv.size(), centroid: centroid, centroidSquareMagnitude: qv.getCentroidDP() };
    qv.dispose();
    return { vectorData: vectorData, metadata: metadata };
  }

  /** deserializeVectorData(vectorData, metadata) -> BinarizedByteVectorValues, searchable  (:538-566) */
  deserializeVectorData(vectorData, metadata) {
    const dim = metadata.dimensions, pb = Math.ceil(dim / 8), n = vectorData.length;
    if (!metadata.centroid || metadata.centroid.length !== dim) throw new Error('向量和质心维度不匹配');
    const codes = new Uint8Array(n * pb), corr = new Float64Array(n * 4);
    for (let i = 0; i < n; i++) {
      const d = vectorData[i];
      if (!d || !d.binaryValues || d.binaryValues.length !== pb) throw new Error('向量 ' + i + ' 维度不匹配');
      codes.set(d.binaryValues, i * pb);
      corr[4 * i] = d.lowerInterval; corr[4 * i + 1] = d.upperInterval; corr[4 * i + 2] = d.additionalCorrection; corr[4 * i + 3] = d.quantizedComponentSum;
    }
    return new BinarizedByteVectorValuesImpl(codes, corr, Float32Array.from(metadata.centroid), 1, n);
  }

  /**
   * extension: the declared-but-unused file pair of the reference (FILE_EXTENSIONS veb / vemb, src/constants.ts:52-57) as real
   * files: <prefix>.veb holds the device tile records byte for byte, <prefix>.vemb the MetadataFormat fields + centroid.
   */
  saveIndex(quantizedVectors, pathPrefix) {
    if (!quantizedVectors) throw new Error('目标向量集合不能为空');
    // a multi-device index is written as one pair per shard (<prefix>.s000, ...) plus a manifest: loadIndex puts it back over BBQ_DEVICES
    native.indexSave(quantizedVectors._deviceIndex(), String(pathPrefix), quantizedVectors.getCentroid(), simOrdinal(this.quantizer.similarityFunction));
  }
  /** loads <prefix>.veb/.vemb straight into HBM; vectorValue()/getCorrectiveTerms() fetch the rows back lazily */
  loadIndex(pathPrefix) {
    const devices = shardDevices(Infinity);
    const r = native.indexLoad(String(pathPrefix), Number(process.env.BBQ_DEVICE || 0), devices.length > 1 ? Int32Array.from(devices) : undefined);
    if (r.sim !== simOrdinal(this.quantizer.similarityFunction)) {
      native.indexDestroy(r.handle);
      throw new Error('不支持的相似性函数: file was written for ordinal ' + r.sim);
    }
    const values = new BinarizedByteVectorValuesImpl(null, null, r.centroid, 1, r.n);
    values._rowBytes = Math.ceil(r.dim / 8);
    values._device = r.handle;
    return values;
  }

  /**
   * computeQuantizationAccuracy(originalVectors, queryVectors) -> {meanError, maxError, minError, stdError, correlation}  (:420-476).
   * As in the reference every query is scored against vector 0 only: quantized (the single-row scorer, no original query passed, so a
   * 4-bit query's centroid term is 0) against computeSimilarity on the fp32 vectors.
   */
  computeQuantizationAccuracy(originalVectors, queryVectors) {
    if (originalVectors.length === 0) throw new Error('原始向量集合不能为空');
    if (queryVectors.length === 0) throw new Error('查询向量集合不能为空');
    if (originalVectors.length !== queryVectors.length) throw new Error('原始向量集合和查询向量集合长度不匹配');
    const quantizedVectors = this.quantizeVectors(originalVectors).quantizedVectors;
    const originalScores = [], quantizedScores = [];
    for (const queryVector of queryVectors) {
      const q = this.quantizeQueryVector(queryVector, quantizedVectors.getCentroid());
      quantizedScores.push(this.scorer.computeQuantizedScore(q.quantizedQuery, q.queryCorrections, quantizedVectors, 0, this.config.queryBits).score);
      originalScores.push(this.scorer.computeOriginalScore(queryVector, originalVectors[0], this.config.quantizer.similarityFunction));
    }
    return this.scorer.computeQuantizationAccuracy(originalScores, quantizedScores);
  }

  getConfig() { return this.config; }
  getQuantizer() { return this.quantizer; }
  getScorer() { return this.scorer; }
}

// ------------------------------------------------------------------ src/topKSelector.ts (host-side caller of the path)

/** MinHeap with the reference's tie behaviour (src/minHeap.ts:9-130) */
class MinHeap {
  constructor(compareFn) { this.heap = []; this.compareFn = compareFn || function (a, b) { return a - b; }; }
  size() { return this.heap.length; }
  isEmpty() { return this.heap.length === 0; }
  peek() { return this.heap[0]; }
  push(item) {
    const h = this.heap; h.push(item);
    let i = h.length - 1;
    while (i > 0) {
      const p = Math.floor((i - 1) / 2);
      if (this.compareFn(h[i], h[p]) >= 0) break;
      const t = h[i]; h[i] = h[p]; h[p] = t; i = p;
    }
  }
  pop() {
    const h = this.heap;
    if (h.length === 0) return null;
    const min = h[0], last = h.pop();
    if (h.length > 0) {
      h[0] = last;
      let i = 0;
      for (;;) {
        let s = i; const l = 2 * i + 1, r = 2 * i + 2;
        if (l < h.length && this.compareFn(h[l], h[s]) < 0) s = l;
        if (r < h.length && this.compareFn(h[r], h[s]) < 0) s = r;
        if (s === i) break;
        const t = h[i]; h[i] = h[s]; h[s] = t; i = s;
      }
    }
    return min;
  }
  toArray() { return this.heap.slice().sort(this.compareFn); }
  clear() { this.heap = []; }
}

// src/vectorSimilarity.ts:73-101
function computeCosineSimilarity(a, b) {
  if (!a || !b) throw new Error('向量不能为空');
  if (a.length !== b.length) throw new Error('向量维度不匹配');
  let dp = 0, na = 0, nb = 0;
  for (let i = 0; i < a.length; i++) { dp += a[i] * b[i]; na += a[i] * a[i]; nb += b[i] * b[i]; }
  if (na === 0 || nb === 0) return 0;
  return dp / (Math.sqrt(na) * Math.sqrt(nb));
}

// ------------------------------------------------------------------ small host helpers the reference also exports
// (src/vectorOperations.ts, src/vectorSimilarity.ts, src/bitwiseDotProduct.ts): same names, messages and number model
// (Float32Array stores round to f32, everything else is f64)

/** normalizeVector, src/vectorOperations.ts:11-34: f64 norm, f32 result; the zero vector stays zero */
function normalizeVector(vector) {
  let norm = 0;
  for (let i = 0; i < vector.length; i++) norm += vector[i] * vector[i];
  norm = Math.sqrt(norm);
  const out = new Float32Array(vector.length);
  if (norm === 0) return out;
  for (let i = 0; i < vector.length; i++) out[i] = vector[i] / norm;
  return out;
}
/** computeCentroid, src/vectorOperations.ts:126-163: the accumulator is a Float32Array (rounded after every += and the final /=) */
function computeCentroid(vectors) {
  if (vectors.length === 0) throw new Error('向量集合不能为空');
  if (!vectors[0]) throw new Error('第一个向量不能为空');
  const dim = vectors[0].length, centroid = new Float32Array(dim);
  for (let i = 0; i < dim; i++) centroid[i] = vectors[0][i];
  for (let j = 1; j < vectors.length; j++) { const v = vectors[j]; if (v) for (let i = 0; i < dim; i++) if (v[i] !== undefined) centroid[i] += v[i]; }
  for (let i = 0; i < dim; i++) centroid[i] /= vectors.length;
  return centroid;
}
/** computeDotProduct, src/vectorOperations.ts:171-185 */
function computeDotProduct(a, b) {
  if (a.length !== b.length) throw new Error('向量维度不匹配');
  let sum = 0;
  for (let i = 0; i < a.length; i++) sum += a[i] * b[i];
  return sum;
}
/** computeEuclideanDistance / computeEuclideanSimilarity, src/vectorSimilarity.ts:38-70 */
function computeEuclideanDistance(a, b) {
  if (!a || !b) throw new Error('向量不能为空');
  if (a.length !== b.length) throw new Error('向量维度不匹配');
  let sum = 0;
  for (let i = 0; i < a.length; i++) { const d = a[i] - b[i]; sum += d * d; }
  return Math.sqrt(sum);
}
function computeEuclideanSimilarity(a, b) { return 1.0 / (1.0 + computeEuclideanDistance(a, b)); }
/** computeMaximumInnerProduct, src/vectorSimilarity.ts:108-118 */
function computeMaximumInnerProduct(a, b) {
  let dp = 0;
  for (let i = 0; i < a.length; i++) if (b[i] !== undefined) dp += a[i] * b[i];
  return dp;
}
/** computeSimilarity, src/vectorSimilarity.ts:14-30 */
function computeSimilarity(a, b, similarityFunction) {
  switch (similarityFunction) {
    case 'EUCLIDEAN': return computeEuclideanSimilarity(a, b);
    case 'COSINE': return computeCosineSimilarity(a, b);
    case 'MAXIMUM_INNER_PRODUCT': return computeMaximumInnerProduct(a, b);
    default: throw new Error('不支持的相似性函数: ' + similarityFunction);
  }
}
/** computeQuantizedDotProduct (= computeInt4BitDotProduct = computeInt1BitDotProduct), src/bitwiseDotProduct.ts:14-55 */
function computeQuantizedDotProduct(q, d) {
  if (q.length !== d.length) throw new Error('向量长度不匹配：查询向量长度' + q.length + '，索引向量长度' + d.length);
  let sum = 0;
  for (let i = 0; i < q.length; i++) sum += q[i] * d[i];
  return sum;
}

/**
 * The original fp32 vectors resident on the GPU (libbbq bbq_vectors_*).  Pass it wherever the reference's selectors
 * take `vectors: Float32Array[]`: the per-candidate computeCosineSimilarity then runs on the device
 * (bbq_rerank_scores: f64, index order, bit-identical), everything else stays as in the reference.
 */
class DeviceVectors {
  constructor(vectors, device) {
    if (!vectors || vectors.length === 0) throw new Error('向量集合不能为空');
    this.length = vectors.length;
    this.dim = vectors[0].length;
    for (let i = 0; i < vectors.length; i++) if (!vectors[i] || vectors[i].length !== this.dim) throw new Error('向量维度不匹配');
    this._h = native.vectorsCreate(flatten(vectors, this.dim), this.length, this.dim, device === undefined ? 0 : device);
  }
  /** computeSimilarity(query, vectors[rows[j]]) for all j (similarity: a VectorSimilarityFunction; default COSINE) */
  trueScores(query, rows, similarityFunction) {
    if (!query) throw new Error('向量不能为空');
    if (query.length !== this.dim) throw new Error('向量维度不匹配');
    const sim = similarityFunction === undefined ? 1 : simOrdinal(similarityFunction);
    return native.rerankScores(this._handle(), 1, Float32Array.from(query), Float64Array.of(0, rows.length), Int32Array.from(rows), sim);
  }
  _handle() { if (!this._h) throw new Error('向量不能为空'); return this._h; }
  dispose() { if (this._h) { native.vectorsDestroy(this._h); this._h = null; } }
}
function createDeviceVectors(vectors, device) { return new DeviceVectors(vectors, device); }

// candidates of the oversampled search with their true scores; host arrays follow the reference line by line,
// a DeviceVectors handle moves the similarity loop to the GPU (missing vectors cannot occur there: rows < length)
function rerankCandidates(query, results, vectors) {
  if (vectors instanceof DeviceVectors) {
    const rows = new Int32Array(results.length);
    for (let i = 0; i < results.length; i++) rows[i] = results[i].index;
    const ts = results.length ? vectors.trueScores(query, rows) : [];
    return results.map(function (r, i) { return { index: r.index, quantizedScore: r.score, trueScore: ts[i] }; });
  }
  return results.map(function (r) {
    const v = vectors[r.index];
    return v ? { index: r.index, quantizedScore: r.score, trueScore: computeCosineSimilarity(query, v) } : null;
  });
}

/** getOversampledTopKWithHeap, src/topKSelector.ts:29-79 */
function getOversampledTopKWithHeap(query, quantizedVectors, vectors, k, oversampleFactor, format) {
  const results = format.searchNearestNeighbors(query, quantizedVectors, k * oversampleFactor);
  const heap = new MinHeap(function (a, b) { return a.trueScore - b.trueScore; });
  for (const cand of rerankCandidates(query, results, vectors)) {
    if (!cand) continue;
    if (heap.size() < k) heap.push(cand);
    else { const p = heap.peek(); if (p && cand.trueScore > p.trueScore) { heap.pop(); heap.push(cand); } }
  }
  const topK = [];
  while (!heap.isEmpty()) { const it = heap.pop(); if (it) topK.push(it); }
  topK.sort(function (a, b) { return b.trueScore - a.trueScore; });
  return topK;
}

/** getOversampledTopKWithSort, src/topKSelector.ts:92-115 */
function getOversampledTopKWithSort(query, quantizedVectors, vectors, k, oversampleFactor, format) {
  const results = format.searchNearestNeighbors(query, quantizedVectors, k * oversampleFactor);
  const cands = rerankCandidates(query, results, vectors).filter(function (c) { return c !== null; });
  cands.sort(function (a, b) { return b.trueScore - a.trueScore; });
  return cands.slice(0, k);
}

/**
 * extension (not in the reference): the whole recipe for many queries in one native call (bbq_search_rerank_batch):
 * oversampled search, true scores and the selector ('heap' | 'sort') all behind the C ABI.
 */
function getOversampledTopKBatch(queries, quantizedVectors, deviceVectors, k, oversampleFactor, format, selector) {
  if (!(deviceVectors instanceof DeviceVectors)) throw new Error('getOversampledTopKBatch needs createDeviceVectors(vectors)');
  if (k < 0) throw new Error('k值不能为负数');
  const dim = quantizedVectors.dimension(), nq = queries.length;
  if (k === 0) return queries.map(function () { return []; });
  const q = format.getQuantizer(), sim = simOrdinal(q.similarityFunction), qb = format.getConfig().queryBits;
  const flat = new Float32Array(nq * dim);
  for (let i = 0; i < nq; i++) {
    const v = queries[i];
    if (!v) throw new Error('查询向量不能为空');
    if (v.length !== dim) throw new Error('查询向量维度与目标向量维度不匹配');
    flat.set(v, i * dim);
  }
  const qz = native.quantizeQueries(flat, nq, quantizedVectors.getCentroid(), sim, qb, q.lambda, q.iters, Number(process.env.BBQ_THREADS || 0));
  const qq = qz.quantized, qc = qz.corrections;
  const r = native.searchRerankBatch(quantizedVectors._deviceIndex(), deviceVectors._handle(), nq, flat, qq, qc, qb, sim, k,
    oversampleFactor, selector === 'sort' ? 1 : 0, 1);
  const out = [];
  for (let i = 0; i < nq; i++) {
    const res = [], n = r.counts[i], base = i * r.stride;
    for (let j = 0; j < n; j++) res.push({ index: r.indices[base + j], quantizedScore: r.quantized[base + j], trueScore: r.trueScores[base + j] });
    out.push(res);
  }
  return out;
}

// ------------------------------------------------------------------ .fvecs / .ivecs (tests/benchmarks/siftDataLoader.ts:27-127)

/** loadSiftVectors(filePath, maxVectors = 10000) -> {vectors: [{dimension, values}], count, dimension}; little-endian records */
function loadSiftVectors(filePath, maxVectors) {
  const max = maxVectors === undefined ? 10000 : maxVectors;
  try {
    const buffer = require('fs').readFileSync(filePath);
    const dv = new DataView(buffer.buffer, buffer.byteOffset, buffer.byteLength);
    const dimension = dv.getUint32(0, true);
    const total = Math.floor(buffer.length / (dimension + 1) / 4), count = Math.min(max, total);
    const vectors = [];
    for (let i = 0; i < count; i++) {
      const off = i * (dimension + 1) * 4, d = dv.getUint32(off, true);
      if (d !== dimension) throw new Error('向量维度不一致: 期望' + dimension + ', 实际' + d);
      const values = new Float32Array(dimension);
      for (let j = 0; j < dimension; j++) values[j] = dv.getFloat32(off + 4 + j * 4, true);
      vectors.push({ dimension: d, values: values });
    }
    return { vectors: vectors, count: vectors.length, dimension: dimension };
  } catch (error) {
    throw new Error('读取SIFT数据失败: ' + (error instanceof Error ? error.message : String(error)));
  }
}
function loadSiftDataset(datasetDir, fileType, maxVectors) {
  return loadSiftVectors(require('path').join(datasetDir, 'sift_' + (fileType === undefined ? 'base' : fileType) + '.fvecs'), maxVectors);
}
/** loadSiftQueries(datasetDir, maxQueries = 100) -> {queries, groundtruth: number[][]} from sift_query.fvecs + sift_groundtruth.ivecs */
function loadSiftQueries(datasetDir, maxQueries) {
  const max = maxQueries === undefined ? 100 : maxQueries;
  const q = loadSiftDataset(datasetDir, 'query', max);
  const buffer = require('fs').readFileSync(require('path').join(datasetDir, 'sift_groundtruth.ivecs'));
  const dv = new DataView(buffer.buffer, buffer.byteOffset, buffer.byteLength);
  const k = dv.getUint32(0, true), groundtruth = [];
  for (let i = 0; i < Math.min(max, q.count); i++) {
    const off = i * (k * 4 + 4), nb = [];
    for (let j = 0; j < k; j++) nb.push(dv.getUint32(off + 4 + j * 4, true));
    groundtruth.push(nb);
  }
  return { queries: q.vectors, groundtruth: groundtruth };
}

// ------------------------------------------------------------------ src/index.ts:62-111

function createBinaryQuantizationFormat(config) {
  return new BinaryQuantizationFormat(config === undefined ? DEFAULT_CONFIG : config);
}
function quickQuantize(vectors, similarityFunction) {
  const sim = similarityFunction === undefined ? VectorSimilarityFunction.COSINE : similarityFunction;
  return new BinaryQuantizationFormat({ quantizer: { similarityFunction: sim, lambda: 0.1, iters: 5 } }).quantizeVectors(vectors);
}
function quickSearch(queryVector, targetVectors, k, similarityFunction) {
  const sim = similarityFunction === undefined ? VectorSimilarityFunction.COSINE : similarityFunction;
  const format = new BinaryQuantizationFormat({ quantizer: { similarityFunction: sim, lambda: 0.1, iters: 5 } });
  const built = format.quantizeVectors(targetVectors);
  try {
    return format.searchNearestNeighbors(queryVector, built.quantizedVectors, k);
  } finally {
    built.quantizedVectors.dispose();  // quickSearch rebuilds the index on every call (src/index.ts:109): free the device copy now
  }
}

/** computeAccuracy(originalVectors, queryVectors, similarityFunction = COSINE): the accuracy report of a lambda-0.1 / 5-iteration format (src/index.ts:120-134) */
function computeAccuracy(originalVectors, queryVectors, similarityFunction) {
  const format = new BinaryQuantizationFormat({
    quantizer: { similarityFunction: similarityFunction === undefined ? VectorSimilarityFunction.COSINE : similarityFunction, lambda: 0.1, iters: 5 },
  });
  return format.computeQuantizationAccuracy(originalVectors, queryVectors);
}

module.exports = Object.assign({}, require('./helpers'), {
  computeAccuracy,
  VectorSimilarityFunction, DEFAULT_CONFIG, VERSION: '1.0.0',
  QUERY_BITS, INDEX_BITS, FOUR_BIT_SCALE, DEFAULT_LAMBDA, DEFAULT_ITERS,
  BinaryQuantizationFormat, OptimizedScalarQuantizer, BinaryQuantizedScorer, MinHeap,
  createBinaryQuantizationFormat, quickQuantize, quickSearch,
  getOversampledTopKWithHeap, getOversampledTopKWithSort, getOversampledTopKBatch, computeCosineSimilarity,
  DeviceVectors, createDeviceVectors, loadSiftVectors, loadSiftDataset, loadSiftQueries,
  normalizeVector, computeCentroid, computeDotProduct, computeEuclideanDistance, computeEuclideanSimilarity, computeMaximumInnerProduct,
  computeSimilarity, computeQuantizedDotProduct, computeInt4BitDotProduct: computeQuantizedDotProduct, computeInt1BitDotProduct: computeQuantizedDotProduct,
  deviceCount: native.deviceCount,
  _native: native, _hostClock: hostClock,
});
