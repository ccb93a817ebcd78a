'use strict';
// Host-side checks that need no GPU: the addon loads, index build / query quantization reproduce the golden
// vectors through the JS API, and argument errors carry the reference's messages (tests/golden/api_behaviour.json).
const T = require('./common');
const bbq = T.bbq;

// --- error behaviour pinned from the reference
const api = T.loadGolden('api_behaviour');
const byLabel = {}; api.cases.forEach(function (c) { byLabel[c.label] = c; });
function expectThrow(label, fn) {
  let msg = null;
  try { fn(); } catch (e) { msg = String(e.message); }
  T.check(msg === byLabel[label].message, label + ': got ' + JSON.stringify(msg) + ' want ' + JSON.stringify(byLabel[label].message));
}
const F = function (extra) {
  const cfg = { quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } };
  Object.keys(extra || {}).forEach(function (k) { cfg[k] = extra[k]; });
  return new bbq.BinaryQuantizationFormat(cfg);
};
expectThrow('ctor queryBits=0', function () { F({ queryBits: 0 }); });
expectThrow('ctor queryBits=9', function () { F({ queryBits: 9 }); });
expectThrow('ctor indexBits=0', function () { F({ indexBits: 0 }); });
expectThrow('ctor indexBits=9', function () { F({ indexBits: 9 }); });
expectThrow('quantize empty', function () { F().quantizeVectors([]); });
expectThrow('quantize dim mismatch', function () { F().quantizeVectors([new Float32Array(4), new Float32Array(5)]); });
expectThrow('quantize NaN', function () { F({ quantizer: { similarityFunction: 'EUCLIDEAN' } }).quantizeVectors([new Float32Array([1, NaN])]); });
expectThrow('quantize Infinity', function () { F({ quantizer: { similarityFunction: 'EUCLIDEAN' } }).quantizeVectors([new Float32Array([1, Infinity])]); });
const base = T.randMatrix(21, 20, 8), q = T.randMatrix(22, 1, 8)[0];
const f = F(), idx = f.quantizeVectors(base).quantizedVectors;
expectThrow('search null query', function () { f.searchNearestNeighbors(null, idx, 3); });
expectThrow('search null index', function () { f.searchNearestNeighbors(q, null, 3); });
expectThrow('search k<0', function () { f.searchNearestNeighbors(q, idx, -1); });
expectThrow('search dim mismatch', function () { f.searchNearestNeighbors(new Float32Array(7), idx, 3); });
expectThrow('vectorValue out of range', function () { idx.vectorValue(99); });
expectThrow('getCorrectiveTerms out of range', function () { idx.getCorrectiveTerms(99); });
T.check(JSON.stringify(f.searchNearestNeighbors(q, idx, 0)) === '[]', 'k=0 -> []');
T.check(idx.dimension() === 8 && idx.size() === 20, 'dimension/size');
T.check(JSON.stringify(f.getConfig()) === JSON.stringify(byLabel['getConfig'].result), 'getConfig');
T.check(bbq.DEFAULT_CONFIG.queryBits === 4 && bbq.DEFAULT_CONFIG.quantizer.similarityFunction === 'COSINE' && bbq.VERSION === '1.0.0', 'DEFAULT_CONFIG / VERSION');

// --- index build + query quantization against the golden vectors
T.goldenNames().filter(function (n) { return !/^(intdot_|api_|rerank_|big_)/.test(n); }).forEach(function (name) {
  const g = T.loadGolden(name), io = T.inputs(g);
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: g.qb, indexBits: g.ib, quantizer: { similarityFunction: g.sim, lambda: g.lambda, iters: g.iters } });
  const index = fmt.quantizeVectors(io.base).quantizedVectors;
  T.check(T.sha(index._codes) === g.codes_sha256, name + ': packed codes');
  T.check(T.sameBits(index.getCentroid(), T.dec(g.centroid_f32, Float32Array)), name + ': centroid');
  const keep = Math.min(g.n, 4), hc = T.dec(g.head_corr_f64, Float64Array);
  for (let i = 0; i < keep; i++) {
    const t = index.getCorrectiveTerms(i);
    T.check(T.sameBits(new Float64Array([t.lowerInterval, t.upperInterval, t.additionalCorrection, t.quantizedComponentSum]), hc.subarray(4 * i, 4 * i + 4)), name + ': corrections row ' + i);
  }
  T.check(T.sameBits(new Float64Array([index.getCentroidDP()]), T.dec(g.centroid_dp_f64, Float64Array)), name + ': centroidDP');
  T.check(index.vectorValue(0).length === g.row_bytes, name + ': row bytes');
  const un = index.getUnpackedVector(0); let ones = 0; for (let d = 0; d < un.length; d++) ones += un[d];
  T.check(un.length === g.dim && (ones === index.getCorrectiveTerms(0).quantizedComponentSum || g.ib !== 1), name + ': unpacked row');
  for (let qi = 0; qi < g.nq; qi++) {
    // the search path normalises COSINE queries twice; quantizeQueryVector once: feed it the already-normalised query
    let p = io.queries[qi];
    if (g.sim === 'COSINE') {
      let n2 = 0; for (let i = 0; i < p.length; i++) n2 += p[i] * p[i];
      const nm = Math.sqrt(n2), pn = new Float32Array(p.length);
      if (nm !== 0) for (let i = 0; i < p.length; i++) pn[i] = p[i] / nm;
      p = pn;
    }
    const r = fmt.quantizeQueryVector(p, index.getCentroid()), rec = g.queries[qi];
    T.check(T.sameBits(r.quantizedQuery, T.dec(rec.qquant_u8, Uint8Array)), name + ': quantized query ' + qi);
    const c = r.queryCorrections;
    T.check(T.sameBits(new Float64Array([c.lowerInterval, c.upperInterval, c.additionalCorrection, c.quantizedComponentSum]), T.dec(rec.qcorr_f64, Float64Array)), name + ': query corrections ' + qi);
    if (rec.single_row) {   // the reference's single-row scorer, with and without the original query
      const want0 = T.dec(rec.single_row.score_f64, Float64Array), want1 = T.dec(rec.single_row.score_with_query_f64, Float64Array);
      const wantd = T.dec(rec.single_row.dot_i32, Int32Array), got0 = new Float64Array(want0.length), got1 = new Float64Array(want0.length), gotd = new Int32Array(want0.length);
      for (let i = 0; i < want0.length; i++) {
        const a = fmt.getScorer().computeQuantizedScore(r.quantizedQuery, c, index, i, g.qb);
        got0[i] = a.score; gotd[i] = a.bitDotProduct;
        got1[i] = fmt.getScorer().computeQuantizedScore(r.quantizedQuery, c, index, i, g.qb, io.queries[qi]).score;
      }
      T.check(T.sameBits(got0, want0) && T.sameBits(got1, want1) && T.sameBits(gotd, wantd), name + ': computeQuantizedScore (single row) ' + qi);
    }
  }
});
// packAsBinary known answer (rust-wasm/src/optimized_scalar_quantizer.rs:321-327)
const packed = new Uint8Array(1);
bbq.OptimizedScalarQuantizer.packAsBinary(new Uint8Array([1, 0, 1, 0, 1, 0, 1, 0]), packed);
T.check(packed[0] === 0xAA, 'packAsBinary');
// serializeVectorData / deserializeVectorData (src/binaryQuantizationFormat.ts:483-566) with the double-pack bug fixed:
// records carry the packed row + four corrections, the metadata the MetadataFormat fields; the round trip keeps every byte
(function () {
  const g = T.loadGolden('m_100d_cos_qb4'), io = T.inputs(g);
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: g.lambda, iters: g.iters } });
  const ser = fmt.serializeVectorData(io.base);
  T.check(ser.vectorData.length === g.n && ser.metadata.vectorCount === g.n && ser.metadata.dimensions === g.dim, 'serialize: counts');
  T.check(ser.vectorData[0].binaryValues.length === g.row_bytes, 'serialize: binaryValues is the packed row');
  T.check(T.sameBits(ser.metadata.centroid, T.dec(g.centroid_f32, Float32Array)), 'serialize: centroid');
  T.check(T.sameBits(new Float64Array([ser.metadata.centroidSquareMagnitude]), T.dec(g.centroid_dp_f64, Float64Array)), 'serialize: centroidSquareMagnitude');
  const flat = new Uint8Array(g.n * g.row_bytes);
  ser.vectorData.forEach(function (d, i) { flat.set(d.binaryValues, i * g.row_bytes); });
  T.check(T.sha(flat) === g.codes_sha256, 'serialize: rows equal the reference codes');
  const back = fmt.deserializeVectorData(ser.vectorData, ser.metadata);
  T.check(back.size() === g.n && back.dimension() === g.dim && T.sha(back._codes) === g.codes_sha256, 'deserialize: rows');
  const hc = T.dec(g.head_corr_f64, Float64Array), t = back.getCorrectiveTerms(1);
  T.check(T.sameBits(new Float64Array([t.lowerInterval, t.upperInterval, t.additionalCorrection, t.quantizedComponentSum]), hc.subarray(4, 8)), 'deserialize: corrections');
  let threw = false;
  try { fmt.deserializeVectorData(ser.vectorData, Object.assign({}, ser.metadata, { dimensions: g.dim + 8 })); } catch (e) { threw = true; }
  T.check(threw, 'deserialize: dimension mismatch throws');
})();
// .fvecs / .ivecs loaders (tests/benchmarks/siftDataLoader.ts:27-127)
(function () {
  const fs = require('fs'), os = require('os'), path = require('path');
  const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'bbq_sift_'));
  const dim = 7, n = 5, k = 3;
  const wf = function (name, rows) {
    const buf = Buffer.alloc(rows.length * (dim + 1) * 4);
    rows.forEach(function (r, i) { buf.writeUInt32LE(dim, i * (dim + 1) * 4); r.forEach(function (v, j) { buf.writeFloatLE(v, i * (dim + 1) * 4 + 4 + j * 4); }); });
    fs.writeFileSync(path.join(dir, name), buf);
  };
  const rows = []; for (let i = 0; i < n; i++) { const r = []; for (let j = 0; j < dim; j++) r.push(i * 10 + j / 4); rows.push(r); }
  wf('sift_base.fvecs', rows); wf('sift_query.fvecs', rows.slice(0, 2));
  const gt = Buffer.alloc(2 * (k + 1) * 4);
  [[4, 2, 0], [1, 3, 2]].forEach(function (r, i) { gt.writeUInt32LE(k, i * (k + 1) * 4); r.forEach(function (v, j) { gt.writeUInt32LE(v, i * (k + 1) * 4 + 4 + j * 4); }); });
  fs.writeFileSync(path.join(dir, 'sift_groundtruth.ivecs'), gt);
  const d = bbq.loadSiftDataset(dir, 'base', 4);
  T.check(d.count === 4 && d.dimension === dim && d.vectors[3].values[2] === 30.5 && d.vectors[3].dimension === dim, 'loadSiftDataset');
  T.check(bbq.loadSiftVectors(path.join(dir, 'sift_base.fvecs')).count === n, 'loadSiftVectors default cap');
  const q = bbq.loadSiftQueries(dir, 10);
  T.check(q.queries.length === 2 && JSON.stringify(q.groundtruth) === '[[4,2,0],[1,3,2]]', 'loadSiftQueries');
  let msg = '';
  try { bbq.loadSiftVectors(path.join(dir, 'nope.fvecs')); } catch (e) { msg = e.message; }
  T.check(/^读取SIFT数据失败: /.test(msg), 'loadSiftVectors error prefix');
  const bad = Buffer.from(fs.readFileSync(path.join(dir, 'sift_base.fvecs'))); bad.writeUInt32LE(dim + 1, (dim + 1) * 4);
  fs.writeFileSync(path.join(dir, 'bad.fvecs'), bad);
  msg = '';
  try { bbq.loadSiftVectors(path.join(dir, 'bad.fvecs')); } catch (e) { msg = e.message; }
  T.check(msg.indexOf('向量维度不一致') >= 0, 'inconsistent record dimension');
})();
// host helpers the reference exports too: against the rerank fixtures (computeSimilarity of every pair) and known answers
(function () {
  const g = T.loadGolden('rerank_3d');
  const fb = T.dec(g.base_f32, Float32Array), fq = T.dec(g.queries_f32, Float32Array);
  ['EUCLIDEAN', 'COSINE', 'MAXIMUM_INNER_PRODUCT'].forEach(function (sim) {
    const want = T.dec(g.true_f64[sim], Float64Array), got = new Float64Array(g.nq * g.n);
    for (let qi = 0; qi < g.nq; qi++) for (let i = 0; i < g.n; i++)
      got[qi * g.n + i] = bbq.computeSimilarity(fq.subarray(qi * g.dim, (qi + 1) * g.dim), fb.subarray(i * g.dim, (i + 1) * g.dim), sim);
    T.check(T.sameBits(got, want), 'computeSimilarity ' + sim + ' equals the reference bit for bit');
  });
  const c = bbq.computeCentroid([[1, 2, 3], [4, 5, 6], [7, 8, 9]].map(function (r) { return new Float32Array(r); }));
  T.check(c[0] === 4 && c[1] === 5 && c[2] === 6, 'computeCentroid known answer');
  const gm = T.loadGolden('m_64d_cos_qb4'), io = T.inputs(gm);
  T.check(T.sameBits(bbq.computeCentroid(io.base.map(bbq.normalizeVector)), T.dec(gm.centroid_f32, Float32Array)), 'normalizeVector + computeCentroid reproduce the fixture centroid');
  const cen = T.dec(gm.centroid_f32, Float32Array);
  T.check(T.sameBits(new Float64Array([bbq.computeDotProduct(cen, cen)]), T.dec(gm.centroid_dp_f64, Float64Array)), 'computeDotProduct(centroid, centroid) = centroidDP');
  T.check(bbq.computeQuantizedDotProduct(new Uint8Array([1, 2, 3, 4]), new Uint8Array([5, 6, 7, 8])) === 70 &&
    bbq.computeInt4BitDotProduct(new Uint8Array([15, 14, 13, 12]), new Uint8Array([1, 1, 0, 1])) === 41, 'bitwise_dot_product.rs:105-120 known answers');
  T.check(bbq.computeEuclideanDistance(new Float32Array([0, 0]), new Float32Array([3, 4])) === 5 && bbq.computeMaximumInnerProduct(new Float32Array([1, 2]), new Float32Array([3, 4])) === 11, 'distance / inner product known answers');
  let msg = '';
  try { bbq.computeDotProduct(new Float32Array(2), new Float32Array(3)); } catch (e) { msg = e.message; }
  T.check(msg === '向量维度不匹配', 'computeDotProduct dimension message');
})();
// --- the helpers / constants the package root re-exports (src/index.ts:20-37), against values the reference returned
(function () {
  const h = api.helpers, va = Float32Array.from(h.inputs.a), vb = Float32Array.from(h.inputs.b), W = h.values;
  const same = function (got, want, label) { T.check(JSON.stringify(got) === JSON.stringify(want), 'helper ' + label + ': ' + JSON.stringify(got) + ' vs ' + JSON.stringify(want)); };
  same(bbq.computeL2Norm(va), W.computeL2Norm, 'computeL2Norm'); same(bbq.computeMean(va), W.computeMean, 'computeMean');
  same(bbq.computeStd(va, bbq.computeMean(va)), W.computeStd, 'computeStd');
  same([bbq.clamp(5, 0, 1), bbq.clamp(-5, 0, 1), bbq.clamp(0.25, 0, 1), bbq.clamp(NaN, 0, 1)].map(String), W.clamp, 'clamp');
  same([0, 1, 255, 0xF0F0F0F0, -1, 0x80000000].map(function (n) { return bbq.bitCount(n); }), W.bitCount, 'bitCount');
  same(bbq.bitCountBytes(new Uint8Array([0, 255, 170, 1])), W.bitCountBytes, 'bitCountBytes');
  same(bbq.bitCountBytesOptimized(new Uint8Array([0, 255, 170, 1])), W.bitCountBytesOptimized, 'bitCountBytesOptimized');
  same([0, 7, 255, 256 + 3].map(function (n) { return bbq.getBitCount(n); }), W.getBitCount, 'getBitCount');
  same(T.sha(bbq.BIT_COUNT_LOOKUP_TABLE), W.BIT_COUNT_LOOKUP_TABLE_sha256, 'BIT_COUNT_LOOKUP_TABLE');
  same([bbq.isNearZero(1e-9), bbq.isNearZero(1e-7), bbq.isNearZero(0.5, 1)], W.isNearZero, 'isNearZero');
  same([bbq.isNearEqual(1, 1 + 1e-9), bbq.isNearEqual(1, 1.1), bbq.isNearEqual(1, 1.1, 0.5)], W.isNearEqual, 'isNearEqual');
  same([bbq.scaleMaxInnerProductScore(-3), bbq.scaleMaxInnerProductScore(0), bbq.scaleMaxInnerProductScore(2.5)], W.scaleMaxInnerProductScore, 'scaleMaxInnerProductScore');
  same(Array.from(bbq.addVectors(va, vb)), W.addVectors, 'addVectors'); same(Array.from(bbq.subtractVectors(va, vb)), W.subtractVectors, 'subtractVectors');
  same(Array.from(bbq.scaleVector(va, 0.3)), W.scaleVector, 'scaleVector'); same(Array.from(bbq.centerVector(va, vb)), W.centerVector, 'centerVector');
  same(Array.from(bbq.copyVector(va)), W.copyVector, 'copyVector'); same(bbq.computeVectorMagnitude(vb), W.computeVectorMagnitude, 'computeVectorMagnitude');
  same(Array.from(bbq.createZeroVector(3)), W.createZeroVector, 'createZeroVector');
  const rv = bbq.createRandomVector(7, 2, 3);
  T.check(rv.length === W.createRandomVector_length && rv.every(function (x) { return x >= 2 && x <= 3; }), 'helper createRandomVector');
  same(bbq.MINIMUM_MSE_GRID, W.MINIMUM_MSE_GRID, 'MINIMUM_MSE_GRID'); same(bbq.FILE_EXTENSIONS, W.FILE_EXTENSIONS, 'FILE_EXTENSIONS');
  same(bbq.COMPONENT_NAMES, W.COMPONENT_NAMES, 'COMPONENT_NAMES'); same(bbq.NUMERICAL_CONSTANTS, W.NUMERICAL_CONSTANTS, 'NUMERICAL_CONSTANTS');
  same({ QUERY_BITS: bbq.QUERY_BITS, INDEX_BITS: bbq.INDEX_BITS, FOUR_BIT_SCALE: bbq.FOUR_BIT_SCALE, DEFAULT_LAMBDA: bbq.DEFAULT_LAMBDA, DEFAULT_ITERS: bbq.DEFAULT_ITERS }, W.constants, 'constants');
  Object.keys(h.errors).forEach(function (name) {
    let msg = null;
    try { bbq[name](va, new Float32Array(2)); } catch (e) { msg = String(e.message); }
    T.check(msg === h.errors[name], 'helper ' + name + ' dimension error: ' + msg);
  });
})();
// --- computeAccuracy / computeQuantizationAccuracy and the scorer's score-comparison helpers (src/index.ts:120-134,
//     src/binaryQuantizationFormat.ts:420-476, src/binaryQuantizedScorer.ts:429-617) against what the reference returned
(function () {
  const A = api.accuracy, base = T.randMatrix(A.base_seed, A.n, A.dim), queries = T.randMatrix(A.query_seed, A.n, A.dim);
  const same = function (got, want, label) { T.check(JSON.stringify(got) === JSON.stringify(want), 'accuracy ' + label + ': ' + JSON.stringify(got) + ' vs ' + JSON.stringify(want)); };
  Object.keys(A.results).forEach(function (key) {
    const m = /^(.*)_qb(\d)_ib(\d)$/.exec(key);
    const fmt = new bbq.BinaryQuantizationFormat({ queryBits: Number(m[2]), indexBits: Number(m[3]), quantizer: { similarityFunction: m[1], lambda: 0.1, iters: 5 } });
    same(fmt.computeQuantizationAccuracy(base, queries), A.results[key], key);
    if (m[2] === '4' && m[3] === '1') same(bbq.computeAccuracy(base, queries, m[1]), A.results[key], 'computeAccuracy ' + m[1]);
  });
  same(bbq.computeAccuracy(base, queries), A.results.COSINE_qb4_ib1, 'computeAccuracy default similarity');
  const sc = new bbq.BinaryQuantizedScorer('COSINE');
  same(A.scorer.compareScores.map(function (p) {
    const r = sc.compareScores(p.a, p.b);
    return { a: p.a, b: p.b, difference: r.difference, relativeError: String(r.relativeError), correlation: r.correlation };
  }), A.scorer.compareScores, 'compareScores');
  T.check(Number.isNaN(sc.compareScores(Infinity, 1).correlation) && sc.compareScores(NaN, 0).correlation === 0, 'compareScores non-finite corners');
  same(['EUCLIDEAN', 'COSINE', 'MAXIMUM_INNER_PRODUCT'].map(function (sim) { return sc.computeOriginalScore(queries[0], base[0], sim); }), A.scorer.computeOriginalScore, 'computeOriginalScore');
  same(sc.computeQuantizationAccuracy([0.1, 0.5, 0.9, 0.3], [0.12, 0.45, 0.97, 0.3]), A.scorer.computeQuantizationAccuracy, 'scorer.computeQuantizationAccuracy');
  same(sc.computeQuantizationAccuracy([0.5, 0.5], [0.4, 0.6]), A.scorer.constantScores, 'constant scores');
  same(sc.getSimilarityFunction(), A.scorer.getSimilarityFunction, 'getSimilarityFunction');
  const F = function () { return new bbq.BinaryQuantizationFormat({ quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } }); };
  const errs = {
    'empty originals': function () { F().computeQuantizationAccuracy([], queries); },
    'empty queries': function () { F().computeQuantizationAccuracy(base, []); },
    'length mismatch': function () { F().computeQuantizationAccuracy(base, queries.slice(0, 3)); },
    'scores length mismatch': function () { sc.computeQuantizationAccuracy([1, 2], [1]); },
    'bad similarity': function () { sc.computeOriginalScore(queries[0], base[0], 'NOPE'); },
  };
  Object.keys(A.errors).forEach(function (name) {
    let msg = null;
    try { errs[name](); } catch (e) { msg = String(e.message); }
    T.check(msg === A.errors[name], 'accuracy error "' + name + '": ' + msg);
  });
})();

// --- the quantizer's remaining public utilities (src/optimizedScalarQuantizer.ts:67-93, 460-627) against what the reference returned
(function () {
  const Z = api.quantizer_utils, W = Z.values, OSQ = bbq.OptimizedScalarQuantizer, q4 = Uint8Array.from(Z.q4);
  const same = function (got, want, label) { T.check(JSON.stringify(got) === JSON.stringify(want), 'quantizer ' + label + ': ' + JSON.stringify(got) + ' vs ' + JSON.stringify(want)); };
  same([[0, 8], [1, 8], [8, 8], [9, 8], [100, 64], [7.5, 4], [-3, 4]].map(function (p) { return OSQ.discretize(p[0], p[1]); }), W.discretize, 'discretize');
  let o = new Uint8Array(q4.length * 4); OSQ.transposeHalfByte(q4, o); same(Array.from(o), W.transposeHalfByte, 'transposeHalfByte');
  o = new Uint8Array(Math.ceil(q4.length / 8) * 4); OSQ.transposeHalfByteFast(q4, o); same(Array.from(o), W.transposeHalfByteFast, 'transposeHalfByteFast');
  o = new Uint8Array(64); OSQ.transposeHalfByteFast(q4.subarray(0, 16), o); same(Array.from(o), W.transposeHalfByteFast_wide_output, 'transposeHalfByteFast, output wider than needed');
  OSQ.clearTransposeCache();
  const stats = [], first = new Uint8Array(q4.length * 4), o2 = new Uint8Array(q4.length * 4), copy = new Uint8Array(q4);
  stats.push(OSQ.getTransposeCacheStats());
  OSQ.transposeHalfByteOptimized(q4, first); stats.push(OSQ.getTransposeCacheStats());
  OSQ.transposeHalfByteOptimized(q4, o2); stats.push(OSQ.getTransposeCacheStats());
  OSQ.transposeHalfByteOptimized(copy, o2); stats.push(OSQ.getTransposeCacheStats());
  OSQ.transposeHalfByteOptimized(copy, o2, false); stats.push(OSQ.getTransposeCacheStats());
  q4[0] ^= 1; OSQ.transposeHalfByteOptimized(q4, o2); q4[0] ^= 1;
  same(Array.from(o2).join() === Array.from(first).join(), W.cache_stale_hit_equals_first, 'a cache hit returns the planes of the first call');
  same(Array.from(first), W.transposeHalfByte, 'transposeHalfByteOptimized');
  OSQ.clearTransposeCache(); stats.push(OSQ.getTransposeCacheStats());
  same(stats, W.cache, 'transpose cache statistics');
  const M = W.multiScalarQuantize, qn = new OSQ({ similarityFunction: 'EUCLIDEAN', lambda: 0.1, iters: 5 });
  const v = T.randMatrix(M.seeds[0], 1, M.dim)[0], cen = Float32Array.from(T.randMatrix(M.seeds[1], 1, M.dim)[0].map(function (x) { return x * 0.1; }));
  const d = M.bits.map(function () { return new Uint8Array(M.dim); });
  same(qn.multiScalarQuantize(v, d, M.bits, cen), M.results, 'multiScalarQuantize corrections');
  same(d.map(function (x) { return Array.from(x); }), M.destinations, 'multiScalarQuantize codes');
  const errs = {
    'multi length mismatch': function () { qn.multiScalarQuantize(v, d, [1, 4], cen); },
    'transpose null': function () { OSQ.transposeHalfByte(null, new Uint8Array(4)); },
    'transpose length': function () { OSQ.transposeHalfByte(new Uint8Array(3), new Uint8Array(4)); },
    'transpose value': function () { OSQ.transposeHalfByte(new Uint8Array([1, 16]), new Uint8Array(8)); },
  };
  Object.keys(Z.errors).forEach(function (name) {
    let msg = null;
    try { errs[name](); } catch (e) { msg = String(e.message); }
    T.check(msg === Z.errors[name], 'quantizer error "' + name + '": ' + msg);
  });
})();

T.finish('js cpu_checks');
