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
  }
});
// packAsBinary known answer (rust-wasm/src/optimized_scalar_quantizer.rs:321-327)
const packed = new Uint8Array(1);
bbq.OptimizedScalarQuantizer.packAsBinary(new Uint8Array([1, 0, 1, 0, 1, 0, 1, 0]), packed);
T.check(packed[0] === 0xAA, 'packAsBinary');
T.finish('js cpu_checks');
