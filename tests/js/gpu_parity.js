'use strict';
// GPU parity through the JavaScript drop-in API: top-k lists, per-row scores and quickSearch against the golden vectors.
const T = require('./common');
const bbq = T.bbq;
if (bbq.deviceCount() < 1) { console.error('no HIP device'); process.exit(2); }
// multi-bit indexes warn like the reference's fallback does; counted, not printed
let warnings = 0;
console.warn = function () { warnings++; };

T.goldenNames().filter(function (n) { return !/^(intdot_|api_|rerank_)/.test(n); }).forEach(function (name) {
  if (/^big_(50000|30000)/.test(name)) return;  // covered by pytest; keeps the node run short
  const g = T.loadGolden(name), io = T.inputs(g);
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: g.qb, indexBits: g.ib, quantizer: { similarityFunction: g.sim, lambda: g.lambda, iters: g.iters } });
  const index = fmt.quantizeVectors(io.base).quantizedVectors;
  for (let qi = 0; qi < g.nq; qi++) {
    const rec = g.queries[qi];
    rec.topk.forEach(function (tk) {
      if (tk.error) {   // the reference throws here (multi-bit index, queryBits its per-row fallback does not know)
        let msg = null;
        try { fmt.searchNearestNeighbors(io.queries[qi], index, tk.k); } catch (e) { msg = e.message; }
        T.check(msg === tk.error, name + ' q' + qi + ': throws like the reference (' + msg + ')');
        return;
      }
      const res = fmt.searchNearestNeighbors(io.queries[qi], index, tk.k);
      const wi = T.dec(tk.idx_i32, Int32Array), ws = T.dec(tk.score_f32, Float32Array);
      let ok = res.length === wi.length;
      for (let i = 0; ok && i < res.length; i++) ok = res[i].index === wi[i] && (res[i].score === ws[i] || (res[i].score !== res[i].score && ws[i] !== ws[i]));
      T.check(ok, name + ' q' + qi + ' k=' + tk.k + ': top-k list');
    });
    if (g.full && g.n <= 1000 && !rec.per_row_error) {  // computeBatchQuantizedScores with scattered ords
      const q = fmt.quantizeQueryVector(g.sim === 'COSINE' ? normalise(io.queries[qi]) : io.queries[qi], index.getCentroid());
      const ords = []; for (let i = g.n - 1; i >= 0; i -= 3) ords.push(i);
      const out = fmt.getScorer().computeBatchQuantizedScores(q.quantizedQuery, q.queryCorrections, index, ords, g.qb);
      const wd = T.dec(rec.qcdist_i32, Int32Array), w64 = T.dec(rec.score_f64, Float64Array);
      let ok = out.length === ords.length;
      for (let i = 0; ok && i < ords.length; i++) ok = out[i].bitDotProduct === wd[ords[i]] && (out[i].score === w64[ords[i]] || (out[i].score !== out[i].score && w64[ords[i]] !== w64[ords[i]]));
      T.check(ok, name + ' q' + qi + ': computeBatchQuantizedScores');
    }
    // the optional originalQueryVector argument (centroidDP = query . centroid): batch path on a 1-bit index, per-row fallback on
    // a multi-bit one - both pinned by values the reference returned
    const want6 = rec.batch_with_query ? T.dec(rec.batch_with_query.score_f64, Float64Array)
      : (g.ib !== 1 && g.dim > 1 && g.qb === 4 && rec.single_row) ? T.dec(rec.single_row.score_with_query_f64, Float64Array) : null;
    if (want6) {
      const q = fmt.quantizeQueryVector(g.sim === 'COSINE' ? normalise(io.queries[qi]) : io.queries[qi], index.getCentroid());
      const ords = []; for (let i = 0; i < want6.length; i++) ords.push(i);
      const out = fmt.getScorer().computeBatchQuantizedScores(q.quantizedQuery, q.queryCorrections, index, ords, g.qb, io.queries[qi]);
      T.check(T.sameBits(Float64Array.from(out.map(function (o) { return o.score; })), want6), name + ' q' + qi + ': computeBatchQuantizedScores with originalQueryVector');
    }
  }
  index.dispose();
});
function normalise(v) {
  let n2 = 0; for (let i = 0; i < v.length; i++) n2 += v[i] * v[i];
  const nm = Math.sqrt(n2), out = new Float32Array(v.length);
  if (nm !== 0) for (let i = 0; i < v.length; i++) out[i] = v[i] / nm;
  return out;
}
// BASELINE config 1 through quickSearch (SURVEY App. C)
(function () {
  const base = T.randMatrix(3, 1000, 128), query = T.randMatrix(4, 1, 128)[0];
  const t0 = Date.now();
  const res = bbq.quickSearch(query, base, 10);
  const ms = Date.now() - t0;
  T.check(JSON.stringify(res.map(function (r) { return r.index; })) === JSON.stringify([438, 839, 190, 656, 637, 545, 630, 174, 862, 42]), 'quickSearch C1 indices');
  T.check(res[0].score === 0.6655263304710388, 'quickSearch C1 top score');
  console.log('quickSearch 1000x128 k=10 (quantize + upload + search): ' + ms + ' ms');
  // oversample + rerank (tests/recall.test.ts:515-636) on the reference's closed-form dataset
  const g = T.loadGolden('closed_100x128_qb4'), io = T.inputs(g);
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: 0.001, iters: 20 } });
  const index = fmt.quantizeVectors(io.base).quantizedVectors;
  for (let qi = 0; qi < g.nq; qi++) {
    const got = bbq.getOversampledTopKWithHeap(io.queries[qi], index, io.base, g.k, 3, fmt).map(function (c) { return c.index; });
    T.check(JSON.stringify(got) === JSON.stringify(g.queries[qi].oversample.idx), 'oversampled top-k q' + qi);
    const got2 = bbq.getOversampledTopKWithSort(io.queries[qi], index, io.base, g.k, 3, fmt).map(function (c) { return c.index; });
    T.check(got2.length === g.k, 'oversample with sort length');
  }
  const k50 = fmt.searchNearestNeighbors(io.queries[0], index, 500);
  T.check(k50.length === 100, 'k > N clamps to N');
  const batch = fmt.searchNearestNeighborsBatch(io.queries, index, 10);
  let ok = batch.length === g.nq;
  for (let qi = 0; ok && qi < g.nq; qi++) {
    const wi = T.dec(g.queries[qi].topk[0].idx_i32, Int32Array);
    for (let i = 0; ok && i < 10; i++) ok = batch[qi][i].index === wi[i];
  }
  T.check(ok, 'searchNearestNeighborsBatch equals per-query results');
  index.setDeviceOption('sweep_share', 32);   // shared sweep on the matrix cores: same answers
  const batch32 = fmt.searchNearestNeighborsBatch(io.queries, index, 10);
  T.check(JSON.stringify(batch32) === JSON.stringify(batch), 'shared sweep (32) equals one sweep per query');
  index.setDeviceOption('sweep_share', 1);
})();
// exact rerank on the device (DeviceVectors) against the reference's values: true scores, both selectors, the batch recipe
T.goldenNames().filter(function (n) { return /^rerank_/.test(n); }).forEach(function (name) {
  const g = T.loadGolden(name);
  const fb = T.dec(g.base_f32, Float32Array), fq = T.dec(g.queries_f32, Float32Array);
  const base = [], queries = [];
  for (let i = 0; i < g.n; i++) base.push(fb.slice(i * g.dim, (i + 1) * g.dim));
  for (let i = 0; i < g.nq; i++) queries.push(fq.slice(i * g.dim, (i + 1) * g.dim));
  const dv = bbq.createDeviceVectors(base);
  const all = new Int32Array(g.n); for (let i = 0; i < g.n; i++) all[i] = i;
  ['EUCLIDEAN', 'COSINE', 'MAXIMUM_INNER_PRODUCT'].forEach(function (sim) {
    const want = T.dec(g.true_f64[sim], Float64Array);
    for (let qi = 0; qi < g.nq; qi++)
      T.check(T.sameBits(dv.trueScores(queries[qi], all, sim), want.subarray(qi * g.n, (qi + 1) * g.n)), name + ' true scores ' + sim + ' q' + qi);
  });
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: g.lambda, iters: g.iters } });
  const index = fmt.quantizeVectors(base).quantizedVectors;
  const same = function (got, rec) {
    return T.sameBits(Int32Array.from(got.map(function (c) { return c.index; })), T.dec(rec.idx_i32, Int32Array)) &&
      T.sameBits(Float32Array.from(got.map(function (c) { return c.quantizedScore; })), T.dec(rec.quantized_f32, Float32Array)) &&
      T.sameBits(Float64Array.from(got.map(function (c) { return c.trueScore; })), T.dec(rec.true_f64, Float64Array));
  };
  const factors = Array.from(new Set(g.oversample.map(function (r) { return r.factor; })));
  factors.forEach(function (f) {
    const bh = bbq.getOversampledTopKBatch(queries, index, dv, g.k, f, fmt, 'heap');
    const bs = bbq.getOversampledTopKBatch(queries, index, dv, g.k, f, fmt, 'sort');
    g.oversample.filter(function (r) { return r.factor === f; }).forEach(function (rec) {
      const q = queries[rec.query], tag = name + ' q' + rec.query + ' x' + f;
      T.check(same(bbq.getOversampledTopKWithHeap(q, index, dv, g.k, f, fmt), rec.heap), tag + ' heap selector, device vectors');
      T.check(same(bbq.getOversampledTopKWithSort(q, index, dv, g.k, f, fmt), rec.sort), tag + ' sort selector, device vectors');
      T.check(same(bbq.getOversampledTopKWithHeap(q, index, base, g.k, f, fmt), rec.heap), tag + ' heap selector, host vectors');
      T.check(same(bh[rec.query], rec.heap), tag + ' batch recipe heap');
      T.check(same(bs[rec.query], rec.sort), tag + ' batch recipe sort');
    });
  });
  dv.dispose();
  index.dispose();
});

// on-disk format: saveIndex -> loadIndex keeps every answer and every row; deserialized indexes are searchable
(function () {
  const fs = require('fs'), os = require('os'), path = require('path');
  const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'bbq_idx_'));
  const g = T.loadGolden('m_768d_cos_qb4'), io = T.inputs(g);
  const fmt = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: g.lambda, iters: g.iters } });
  const index = fmt.quantizeVectors(io.base).quantizedVectors;
  const prefix = path.join(dir, 'm768');
  fmt.saveIndex(index, prefix);
  // one pair for a single-device index; a sharded one (BBQ_DEVICES) writes a manifest + one pair per shard
  T.check(fs.existsSync(prefix + '.vemb') && (fs.existsSync(prefix + '.veb') || fs.existsSync(prefix + '.s000.veb')), 'saveIndex writes .veb + .vemb');
  const loaded = fmt.loadIndex(prefix);
  T.check(loaded.size() === g.n && loaded.dimension() === g.dim, 'loadIndex: size/dimension');
  let ok = true;
  for (let qi = 0; qi < g.nq; qi++) {
    const a = fmt.searchNearestNeighbors(io.queries[qi], index, g.k), b = fmt.searchNearestNeighbors(io.queries[qi], loaded, g.k);
    ok = ok && JSON.stringify(a) === JSON.stringify(b) && T.sameBits(Int32Array.from(b.map(function (r) { return r.index; })), T.dec(g.queries[qi].topk.filter(function (t) { return t.k === g.k; })[0].idx_i32, Int32Array));
  }
  T.check(ok, 'loadIndex: same top-k as the index that was saved and as the reference');
  T.check(T.sha(loaded.vectorValue(0)) === T.sha(index.vectorValue(0)) && loaded._codes && T.sha(loaded._codes) === g.codes_sha256, 'loadIndex: rows come back from the device');
  T.check(JSON.stringify(loaded.getCorrectiveTerms(5)) === JSON.stringify(index.getCorrectiveTerms(5)), 'loadIndex: corrections');
  const euc = new bbq.BinaryQuantizationFormat({ quantizer: { similarityFunction: 'EUCLIDEAN' } });
  let threw = false;
  try { euc.loadIndex(prefix); } catch (e) { threw = true; }
  T.check(threw, 'loadIndex refuses a file written for another similarity');
  threw = false;
  try { fmt.loadIndex(path.join(dir, 'absent')); } catch (e) { threw = /cannot open/.test(e.message); }
  T.check(threw, 'loadIndex: missing file');
  const ser = fmt.serializeVectorData(io.base), back = fmt.deserializeVectorData(ser.vectorData, ser.metadata);
  T.check(JSON.stringify(fmt.searchNearestNeighbors(io.queries[0], back, g.k)) === JSON.stringify(fmt.searchNearestNeighbors(io.queries[0], index, g.k)), 'deserialized index searches like the original');
  loaded.dispose(); back.dispose(); index.dispose();
})();

T.check(warnings > 0, 'multi-bit indexes warn like the reference fallback');
// --- computeAccuracy with the device-built index (src/index.ts:120-134): the reference's statistics, bit for bit
(function () {
  const A = T.loadGolden('api_behaviour').accuracy, base = T.randMatrix(A.base_seed, A.n, A.dim), queries = T.randMatrix(A.query_seed, A.n, A.dim);
  Object.keys(A.results).forEach(function (key) {
    const m = /^(.*)_qb(\d)_ib(\d)$/.exec(key);
    const fmt = new bbq.BinaryQuantizationFormat({ queryBits: Number(m[2]), indexBits: Number(m[3]), quantizer: { similarityFunction: m[1], lambda: 0.1, iters: 5 } });
    T.check(JSON.stringify(fmt.computeQuantizationAccuracy(base, queries)) === JSON.stringify(A.results[key]), 'computeQuantizationAccuracy ' + key);
  });
  T.check(JSON.stringify(bbq.computeAccuracy(base, queries, 'EUCLIDEAN')) === JSON.stringify(A.results.EUCLIDEAN_qb4_ib1), 'computeAccuracy EUCLIDEAN');
})();

T.finish('js gpu_parity');
