'use strict';
// The reference's OWN test expectations (SURVEY §4), run through the drop-in JavaScript API on the GPU.  Each block cites the
// reference test it mirrors; data generators are restated from the cited lines (they are closed-form).
const T = require('./common');
const bbq = T.bbq;

function normalizeVector(v) {  // src/vectorOperations.ts:11-34
  let norm = 0;
  for (let i = 0; i < v.length; i++) norm += v[i] * v[i];
  norm = Math.sqrt(norm);
  const out = new Float32Array(v.length);
  if (norm === 0) return out;
  for (let i = 0; i < v.length; i++) out[i] = v[i] / norm;
  return out;
}
function closedForm(n, dim, offset, normalize) {  // tests/recall.test.ts:20-57, tests/recall-common.ts:112-138
  const out = [];
  for (let i = 0; i < n; i++) {
    const v = new Float32Array(dim);
    for (let j = 0; j < dim; j++) { const s = (i + offset) * 1000 + j; v[j] = Math.sin(s) * 0.5 + Math.cos(s * 0.7) * 0.3; }
    out.push(normalize ? normalizeVector(v) : v);
  }
  return out;
}
function trueTopK(query, base, k) {  // tests/recall-common.ts:143-149
  return base.map(function (v, idx) { return { idx: idx, score: bbq.computeCosineSimilarity(query, v) }; })
    .sort(function (a, b) { return b.score - a.score; }).slice(0, k).map(function (x) { return x.idx; });
}
function recall(truth, got) { let hit = 0; for (const i of got) if (truth.includes(i)) hit++; return hit / truth.length; }  // recall-common.ts:161-167
function format(qb, ib, lambda, iters) {
  return new bbq.BinaryQuantizationFormat({ queryBits: qb, indexBits: ib, quantizer: { similarityFunction: 'COSINE', lambda: lambda, iters: iters } });
}
function avgRecall(fmt, base, queries, k, oversample) {
  const index = fmt.quantizeVectors(base).quantizedVectors;
  let total = 0, lengthsOk = true;
  for (const q of queries) {
    const got = oversample
      ? bbq.getOversampledTopKWithHeap(q, index, base, k, oversample, fmt).map(function (x) { return x.index; })
      : fmt.searchNearestNeighbors(q, index, k).map(function (x) { return x.index; });
    lengthsOk = lengthsOk && got.length === k;
    total += recall(trueTopK(q, base, k), got);
  }
  index.dispose();
  return { recall: total / queries.length, lengthsOk: lengthsOk };
}

// ---- tests/recall.test.ts: 100 x 128, 10 queries, k = 10, lambda 0.001, iters 20
(function () {
  const base = closedForm(100, 128, 0, false), queries = closedForm(10, 128, 1000, false);
  const r1 = avgRecall(format(1, 1, 0.001, 20), base, queries, 10);
  T.check(r1.lengthsOk && r1.recall >= 0.70, 'recall.test.ts:88-165  1-bit query / 1-bit index avg recall@10 >= 0.70 (got ' + r1.recall.toFixed(3) + ')');
  const r4 = avgRecall(format(4, 1, 0.001, 20), base, queries, 10);
  T.check(r4.lengthsOk && r4.recall >= 0.60, 'recall.test.ts:387-508 4-bit query / 1-bit index avg recall@10 >= 0.60 (got ' + r4.recall.toFixed(3) + ')');
  const ro = avgRecall(format(4, 1, 0.001, 20), base, queries, 10, 3);
  T.check(ro.lengthsOk && ro.recall >= 0.75, 'recall.test.ts:515-636 3x oversample + exact rerank avg recall@10 >= 0.75 (got ' + ro.recall.toFixed(3) + ')');
  console.log('closed-form 100x128: recall 1-bit ' + r1.recall.toFixed(3) + ', 4-bit ' + r4.recall.toFixed(3) + ', oversample ' + ro.recall.toFixed(3));
})();

// ---- tests/recall.test.ts:299-380: the hand-made 5 x 4 case through the scorer, recall > 0
(function () {
  const base = [[1, 0, 0, 0], [0.9, 0.1, 0, 0], [0.8, 0.2, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]].map(function (r) { return new Float32Array(r); });
  const query = new Float32Array([1, 0, 0, 0]);
  const fmt = format(1, 1, 0.05, 10);
  const index = fmt.quantizeVectors(base).quantizedVectors;
  const qq = fmt.quantizeQueryVector(query, index.getCentroid());
  const scores = fmt.getScorer().computeBatchQuantizedScores(qq.quantizedQuery, qq.queryCorrections, index, [0, 1, 2, 3, 4], 1);
  T.check(scores.length === 5 && scores.every(function (s) { return typeof s.score === 'number'; }), 'recall.test.ts:340-362 five numeric scores');
  const top = scores.map(function (s, idx) { return { idx: idx, score: s.score }; }).sort(function (a, b) { return b.score - a.score; }).slice(0, 3).map(function (x) { return x.idx; });
  T.check(recall(trueTopK(query, base, 3), top) > 0 && top.length === 3, 'recall.test.ts:365-378 recall > 0 on the hand-made case');
  index.dispose();
})();

// ---- tests/recall-all-dimensions.test.ts + recall-common.ts:45-106: 1000 x {384,768,1024,1536} normalised, 20 queries, k = 10
(function () {
  const cfg = { 384: [0.60, 0.75, 0.80], 768: [0.55, 0.70, 0.75], 1024: [0.50, 0.65, 0.70], 1536: [0.45, 0.60, 0.65] };
  const seen = {};
  Object.keys(cfg).forEach(function (d) {
    const dim = Number(d), base = closedForm(1000, dim, 0, true), queries = closedForm(20, dim, 1000, true), th = cfg[d];
    const r1 = avgRecall(format(1, 1, 0.001, 20), base, queries, 10).recall;
    const r4 = avgRecall(format(4, 1, 0.001, 20), base, queries, 10).recall;
    const ro = avgRecall(format(4, 1, 0.001, 20), base, queries, 10, 3).recall;
    seen[d] = [r1, r4, ro];
    T.check(r1 >= th[0], 'recall-all-dimensions.test.ts:48-60 ' + d + 'd 1-bit recall >= ' + th[0] + ' (got ' + r1.toFixed(3) + ')');
    T.check(r4 >= th[1], 'recall-all-dimensions.test.ts:62-74 ' + d + 'd 4-bit recall >= ' + th[1] + ' (got ' + r4.toFixed(3) + ')');
    T.check(ro >= th[2], 'recall-all-dimensions.test.ts:76-88 ' + d + 'd oversample recall >= ' + th[2] + ' (got ' + ro.toFixed(3) + ')');
  });
  console.log('closed-form 1000 x d recall [1-bit, 4-bit, oversample]: ' + JSON.stringify(seen, function (k, v) { return typeof v === 'number' ? Number(v.toFixed(3)) : v; }));
})();

// ---- tests/batch-quantized-scores.test.ts: batch == row-by-row, qcDist == naive dot, empty ords -> []
(function () {
  const base = T.randMatrix(901, 5000, 1024), query = T.randMatrix(902, 1, 1024)[0];
  [1, 4].forEach(function (qb) {
    const fmt = new bbq.BinaryQuantizationFormat({ queryBits: qb, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } });
    const index = fmt.quantizeVectors(base).quantizedVectors;
    const qq = fmt.quantizeQueryVector(normalizeVector(query), index.getCentroid());
    const ords = []; for (let i = 0; i < (qb === 1 ? 5000 : 100); i++) ords.push(i);
    const batch = fmt.getScorer().computeBatchQuantizedScores(qq.quantizedQuery, qq.queryCorrections, index, ords, qb);
    let same = batch.length === ords.length, naive = true;
    for (let i = 0; i < 100; i++) {
      const one = fmt.getScorer().computeBatchQuantizedScores(qq.quantizedQuery, qq.queryCorrections, index, [i], qb)[0];
      same = same && Math.abs(one.score - batch[i].score) <= 1e-10 && one.bitDotProduct === batch[i].bitDotProduct;
      const row = index.getUnpackedVector(i);
      let dot = 0;   // the scorer packs a 1-bit query itself (src/binaryQuantizedScorer.ts:333-335): a plain dot in both cases
      for (let d = 0; d < 1024; d++) dot += qq.quantizedQuery[d] * row[d];
      naive = naive && dot === batch[i].bitDotProduct;
    }
    T.check(same, 'batch-quantized-scores.test.ts:' + (qb === 1 ? '130-145' : '281-294') + ' batch == single-row within 1e-10 (queryBits ' + qb + ')');
    T.check(naive, 'batch-quantized-scores.test.ts:148-154 bitDotProduct == naive per-row dot (queryBits ' + qb + ')');
    T.check(fmt.getScorer().computeBatchQuantizedScores(qq.quantizedQuery, qq.queryCorrections, index, [], qb).length === 0, 'batch-quantized-scores.test.ts:178-194 empty ords -> []');
    index.dispose();
  });
})();

// ---- tests/simple-quantized-query.test.ts:110-111: 5000 x 1024 random, queryBits 4, k = 10, 10 queries x 3
(function () {
  const base = T.randMatrix(911, 5000, 1024), queries = T.randMatrix(912, 10, 1024);
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } });
  const index = fmt.quantizeVectors(base).quantizedVectors;
  let ok = true;
  for (let rep = 0; rep < 3; rep++) {
    const results = queries.map(function (q) { return fmt.searchNearestNeighbors(q, index, 10); });
    ok = ok && results.length === 10 && results[0].length === 10;
  }
  T.check(ok, 'simple-quantized-query.test.ts:110-111 results.length / results[0].length == k');
  index.dispose();
})();

// ---- known answers: computeCentroid-correctness.test.ts:64-83 and the Rust unit tests (SURVEY §4 last row)
(function () {
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'EUCLIDEAN', lambda: 0.1, iters: 5 } });
  const cen = fmt.quantizeVectors([[1, 2, 3], [4, 5, 6], [7, 8, 9]].map(function (r) { return new Float32Array(r); })).quantizedVectors.getCentroid();
  T.check(cen[0] === 4 && cen[1] === 5 && cen[2] === 6, 'computeCentroid-correctness.test.ts:64-83 centroid [4,5,6]');
  // batch_dot_product.rs:137-154: q = [1..8], rows 0xFF, 0x00 -> [36, 0]
  const meta = { fieldNumber: 0, vectorEncodingOrdinal: 0, vectorSimilarityOrdinal: 0, dimensions: 8, vectorDataOffset: 0, vectorDataLength: 0, vectorCount: 2,
    centroid: new Float32Array(8), centroidSquareMagnitude: 0 };
  const rows = [0xFF, 0x00].map(function (b) { return { binaryValues: new Uint8Array([b]), lowerInterval: -1, upperInterval: 1, additionalCorrection: 0, quantizedComponentSum: b ? 8 : 0 }; });
  const index = fmt.deserializeVectorData(rows, meta);
  const corr = { lowerInterval: -1, upperInterval: 1, additionalCorrection: 0, quantizedComponentSum: 36 };
  const r = fmt.getScorer().computeBatchQuantizedScores(new Uint8Array([1, 2, 3, 4, 5, 6, 7, 8]), corr, index, [0, 1], 4);
  T.check(r[0].bitDotProduct === 36 && r[1].bitDotProduct === 0, 'batch_dot_product.rs:137-154 four-bit batch dot [36, 0]');
  const packed = new Uint8Array(1);
  bbq.OptimizedScalarQuantizer.packAsBinary(new Uint8Array([1, 0, 1, 0, 1, 0, 1, 0]), packed);
  T.check(packed[0] === 0b10101010, 'optimized_scalar_quantizer.rs:321-327 packAsBinary');
  index.dispose();
})();

T.finish('js reference_suite');
