'use strict';
// shared helpers of the node-side tests (Node >= 12)
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const ROOT = path.join(__dirname, '..', '..');
const bbq = require(path.join(ROOT, 'better-binary-quantization_amd', 'js', 'index.js'));
const GOLDEN = path.join(ROOT, 'tests', 'golden');

function mulberry32(seed) {
  let a = seed | 0;
  return function () {
    a |= 0; a = a + 0x6D2B79F5 | 0;
    let t = Math.imul(a ^ a >>> 15, 1 | a);
    t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t;
    return ((t ^ t >>> 14) >>> 0) / 4294967296;
  };
}
function randMatrix(seed, n, dim) {
  const r = mulberry32(seed), out = [];
  for (let i = 0; i < n; i++) { const v = new Float32Array(dim); for (let j = 0; j < dim; j++) v[j] = 2 * r() - 1; out.push(v); }
  return out;
}
function dec(b64, Ctor) {
  const buf = Buffer.from(b64, 'base64');
  const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
  return new Ctor(ab);
}
function sha(typed) { return crypto.createHash('sha256').update(Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength)).digest('hex'); }
function loadGolden(name) { return JSON.parse(fs.readFileSync(path.join(GOLDEN, name + '.json'), 'utf8')); }
function goldenNames() { return fs.readdirSync(GOLDEN).filter(function (f) { return f.endsWith('.json'); }).map(function (f) { return f.slice(0, -5); }).sort(); }
function inputs(g) {
  const gen = g.gen;
  let base, queries;
  if (gen.kind === 'mulberry32') { base = randMatrix(gen.base_seed, g.n, g.dim); queries = randMatrix(gen.query_seed, g.nq, g.dim); }
  else if (gen.kind === 'dup_pool') {
    const pool = randMatrix(gen.base_seed, gen.pool, g.dim), r = mulberry32(gen.pick_seed);
    base = []; for (let i = 0; i < g.n; i++) base.push(new Float32Array(pool[Math.floor(r() * gen.pool)]));
    queries = randMatrix(gen.query_seed, g.nq, g.dim);
  } else {
    const fb = dec(g.base_f32, Float32Array), fq = dec(g.queries_f32, Float32Array);
    base = []; for (let i = 0; i < g.n; i++) base.push(fb.slice(i * g.dim, (i + 1) * g.dim));
    queries = []; for (let i = 0; i < g.nq; i++) queries.push(fq.slice(i * g.dim, (i + 1) * g.dim));
  }
  return { base: base, queries: queries };
}
let failures = 0, checks = 0;
function check(cond, msg) { checks++; if (!cond) { failures++; console.error('FAIL: ' + msg); } }
function sameBits(a, b) {  // typed arrays equal, NaN == NaN
  if (a.length !== b.length) return false;
  for (let i = 0; i < a.length; i++) if (!(a[i] === b[i] || (a[i] !== a[i] && b[i] !== b[i]))) return false;
  return true;
}
function finish(label) {
  console.log(label + ': ' + checks + ' checks, ' + failures + ' failures');
  process.exit(failures ? 1 : 0);
}
module.exports = { bbq, loadGolden, goldenNames, inputs, dec, sha, check, sameBits, finish, randMatrix };
