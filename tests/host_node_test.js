// Mocha-free restatement of the reference suite's assertions (test/index.js:16-108) against the
// drop-in façade zlib.es_amd/host/zlib.js, plus exact-byte checks against the golden fixtures.
// Run by tests/test_gpu_host_node.py on the GPU box:  node tests/host_node_test.js
'use strict';
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const nodeZlib = require('zlib');
const zlibes = require('../zlib.es_amd/host/zlib.js');

const G = path.join(__dirname, 'golden');
const kat = JSON.parse(fs.readFileSync(path.join(G, 'kat.json'), 'utf8'));
const hex = (u8) => Buffer.from(u8.buffer, u8.byteOffset, u8.length).toString('hex');
const fromHex = (h) => new Uint8Array(Buffer.from(h, 'hex'));
let n = 0;
function it(name, f) { f(); n++; console.log('ok - ' + name); }

const RAW = fromHex(kat.kat.RAW);
const RAW_BIN = new Uint8Array(fs.readFileSync(path.join(G, 'ref_data', 'raw.bin')));
const CMP_BIN = new Uint8Array(fs.readFileSync(path.join(G, 'ref_data', 'compressed.bin')));

// inflate (test/index.js:15-43)
it('inflate UNCOMPRESSED', () => assert.deepStrictEqual(zlibes.inflate(fromHex(kat.kat.UNCOMPRESSED)), RAW));
it('inflate FIXED', () => assert.deepStrictEqual(zlibes.inflate(fromHex(kat.kat.FIXED)), RAW));
it('inflate DYNAMIC', () => assert.deepStrictEqual(zlibes.inflate(fromHex(kat.kat.DYNAMIC)), RAW));
it('inflate binary data', () => assert.ok(Buffer.from(zlibes.inflate(CMP_BIN)).equals(Buffer.from(RAW_BIN))));

// deflate execution + validation (test/index.js:45-108)
it('deflate RAW executes', () => { zlibes.deflate(RAW); });
it('deflate binary data executes', () => { zlibes.deflate(RAW_BIN); });
for (const [name, data] of [['RAW', RAW], ['binary data', RAW_BIN]]) {
  const out = zlibes.deflate(data);
  it('deflate ' + name + ' -> zlib.es inflate', () => assert.ok(Buffer.from(zlibes.inflate(out)).equals(Buffer.from(data))));
  it('deflate ' + name + ' -> node zlib inflate', () => assert.ok(nodeZlib.inflateSync(Buffer.from(out)).equals(Buffer.from(data))));
}
{
  const rep = new Uint8Array(1023);  // "Repeat Length Limit" (test/index.js:88-108)
  for (let i = 0; i < 1023; i++) rep[i] = 48 + (i % 10);
  const out = zlibes.deflate(rep);
  it('repeat limit -> zlib.es inflate', () => assert.deepStrictEqual(zlibes.inflate(out), rep));
  it('repeat limit -> node zlib inflate', () => assert.ok(nodeZlib.inflateSync(Buffer.from(out)).equals(Buffer.from(rep))));
}

// bit-exactness against the reference's own output (fixtures made by tests/golden/make_golden.js)
for (const k of Object.keys(kat.small)) {
  it('deflate bytes == reference: ' + k, () => assert.strictEqual(hex(zlibes.deflate(fromHex(kat.small[k].input))), kat.small[k].deflate));
}
// result shape: fresh exact-size Uint8Array (src/zlib.ts:42)
it('result is an exact-size Uint8Array', () => {
  const o = zlibes.deflate(RAW);
  assert.ok(o instanceof Uint8Array && o.byteOffset === 0 && o.buffer.byteLength === o.length);
});
// Buffer input works like a Uint8Array
it('Buffer input', () => assert.strictEqual(hex(zlibes.deflate(Buffer.from(RAW))), kat.small.RAW.deflate));
// thrown messages (SURVEY §8b)
function throwsMsg(f, msg) { assert.throws(f, (e) => e instanceof Error && e.message === msg); }
// raw forms (src/deflate.ts:14, src/inflate.ts:16)
it('deflateRaw == deflate without wrapper', () => assert.strictEqual(hex(zlibes.deflateRaw(RAW)), kat.small.RAW.deflate.slice(4, -8)));
it('deflateRaw -> node zlib inflateRaw', () => assert.ok(nodeZlib.inflateRawSync(Buffer.from(zlibes.deflateRaw(RAW_BIN))).equals(Buffer.from(RAW_BIN))));
it('inflateRaw(zlib stream, 2)', () => assert.deepStrictEqual(zlibes.inflateRaw(fromHex(kat.kat.DYNAMIC), 2), RAW));
it('inflateRaw(raw)', () => assert.ok(Buffer.from(zlibes.inflateRaw(zlibes.deflateRaw(RAW_BIN))).equals(Buffer.from(RAW_BIN))));
it('deflate(empty) throws', () => throwsMsg(() => zlibes.deflate(new Uint8Array(0)), 'Data is corrupted'));
it('deflate(1 byte) throws', () => throwsMsg(() => zlibes.deflate(new Uint8Array(1)), 'Data is corrupted'));
it('inflate(empty) throws', () => throwsMsg(() => zlibes.inflate(new Uint8Array(0)), 'Not compressed by deflate'));
it('inflate(bad CM) throws', () => throwsMsg(() => zlibes.inflate(new Uint8Array([0x77, 0x9c, 1, 2, 3])), 'Not compressed by deflate'));
it('inflate(BTYPE 3) throws', () => throwsMsg(() => zlibes.inflate(new Uint8Array([0x78, 0x9c, 7, 0, 0, 0])), 'Not supported BTYPE : 3'));
const cases = JSON.parse(fs.readFileSync(path.join(G, 'inflate_cases.json'), 'utf8'));
it('malformed-stream cases (every 7th)', () => {
  for (let i = 0; i < cases.length; i += 7) {
    const c = cases[i];
    let got;
    try { got = {output: hex(zlibes.inflate(fromHex(c.input)))}; } catch (e) { got = {error: e.message}; }
    assert.deepStrictEqual(got, c.error ? {error: c.error} : {output: c.output}, c.name);
  }
});
// Promise-returning forms: same bytes, same messages, several in flight at once
(async () => {
  const outs = await Promise.all([zlibes.deflateAsync(RAW), zlibes.deflateAsync(RAW_BIN), zlibes.inflateAsync(CMP_BIN),
                                  zlibes.inflateAsync(fromHex(kat.kat.DYNAMIC))]);
  it('deflateAsync bytes == deflate', () => assert.strictEqual(hex(outs[0]), hex(zlibes.deflate(RAW))));
  it('deflateAsync binary data == deflate', () => assert.strictEqual(hex(outs[1]), hex(zlibes.deflate(RAW_BIN))));
  it('inflateAsync binary data', () => assert.ok(Buffer.from(outs[2]).equals(Buffer.from(RAW_BIN))));
  it('inflateAsync DYNAMIC', () => assert.deepStrictEqual(outs[3], RAW));
  it('async result is an exact-size Uint8Array', () => assert.ok(outs[0] instanceof Uint8Array && outs[0].byteOffset === 0 && outs[0].buffer.byteLength === outs[0].length));
  let msg = null;
  try { await zlibes.inflateAsync(new Uint8Array([0x78, 0x9c, 7, 0, 0, 0])); } catch (e) { msg = e instanceof Error ? e.message : String(e); }
  it('inflateAsync rejects with the reference message', () => assert.strictEqual(msg, 'Not supported BTYPE : 3'));
  msg = null;
  try { await zlibes.deflateAsync(new Uint8Array(1)); } catch (e) { msg = e.message; }
  it('deflateAsync rejects with the reference message', () => assert.strictEqual(msg, 'Data is corrupted'));
  console.log('all ' + n + ' host checks passed');
})().catch((e) => { console.error(e); process.exit(1); });
