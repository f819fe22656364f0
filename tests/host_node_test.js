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
  // synchronous calls while Promise-returning ones are in flight on worker threads: the library keeps no state
  // between calls (zes_inflate_alloc decodes and copies out under one lock), so they cannot disturb each other
  {
    const pending = [];
    for (let k = 0; k < 6; k++) pending.push(k % 2 ? zlibes.inflateAsync(CMP_BIN) : zlibes.deflateAsync(RAW_BIN));
    const syncOut = [];
    for (let k = 0; k < 6; k++) syncOut.push(k % 2 ? zlibes.inflate(fromHex(kat.kat.DYNAMIC)) : zlibes.deflate(RAW));
    const res = await Promise.all(pending);
    it('sync calls interleaved with pending async ones', () => {
      for (let k = 0; k < 6; k++) {
        if (k % 2) { assert.deepStrictEqual(syncOut[k], RAW); assert.ok(Buffer.from(res[k]).equals(Buffer.from(RAW_BIN))); }
        else { assert.strictEqual(hex(syncOut[k]), kat.small.RAW.deflate); assert.strictEqual(hex(res[k]), hex(outs[1])); }
      }
    });
  }
  // batch forms: element i is what deflate(inputs[i]) / inflate(inputs[i]) returns, or the Error it throws
  {
    const ins = [RAW, RAW_BIN, new Uint8Array(1), fromHex(kat.small.digits1023.input)];
    const b = zlibes.deflateBatch(ins);
    it('deflateBatch', () => {
      assert.strictEqual(hex(b[0]), kat.small.RAW.deflate);
      assert.strictEqual(hex(b[1]), hex(outs[1]));
      assert.ok(b[2] instanceof Error && b[2].message === 'Data is corrupted');
      assert.strictEqual(hex(b[3]), kat.small.digits1023.deflate);
      assert.ok(b[0] instanceof Uint8Array && b[0].byteOffset === 0 && b[0].buffer.byteLength === b[0].length);
    });
    const back = zlibes.inflateBatch([b[0], CMP_BIN, new Uint8Array([0x78, 0x9c, 7, 0, 0, 0]), b[3], fromHex(kat.kat.FIXED)]);
    it('inflateBatch', () => {
      assert.deepStrictEqual(back[0], RAW);
      assert.ok(Buffer.from(back[1]).equals(Buffer.from(RAW_BIN)));
      assert.ok(back[2] instanceof Error && back[2].message === 'Not supported BTYPE : 3');
      assert.strictEqual(hex(back[3]), kat.small.digits1023.input);
      assert.deepStrictEqual(back[4], RAW);
    });
    const ab = await zlibes.deflateBatchAsync(ins);
    const abk = await zlibes.inflateBatchAsync([ab[0], ab[1]]);
    it('batch Async forms', () => {
      assert.strictEqual(hex(ab[1]), hex(outs[1]));
      assert.ok(ab[2] instanceof Error && ab[2].message === 'Data is corrupted');
      assert.ok(Buffer.from(abk[1]).equals(Buffer.from(RAW_BIN)));
    });
    it('batch argument checks', () => assert.throws(() => zlibes.deflateBatch([RAW, 5]), TypeError));
    it('inflateRaw offset checks', () => {
      assert.throws(() => zlibes.inflateRaw(RAW, -1), TypeError);
      assert.throws(() => zlibes.inflateRaw(RAW, 1.5), TypeError);
      assert.throws(() => zlibes.inflateRaw(RAW, NaN), TypeError);
    });
  }
  // pinned arrays are ordinary Uint8Arrays to every entry point
  {
    const p = zlibes.allocPinned(RAW_BIN.length);
    p.set(RAW_BIN);
    it('allocPinned input', () => assert.strictEqual(hex(zlibes.deflate(p)), hex(outs[1])));
  }
  // BASELINE.json configs[3], this GPU's share: 128 x 1 MiB buffers through the batch entry, every result against the
  // reference's own output (tests/golden/batch1m.json; the inputs are written by the pytest wrapper from the same
  // generators the fixture script restates)
  if (process.env.ZES_BATCH1M_INPUT) {
    const crypto = require('crypto');
    const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.length)).digest('hex');
    const gold = JSON.parse(fs.readFileSync(path.join(G, 'batch1m.json'), 'utf8')).slice(0, 128);
    const all = new Uint8Array(fs.readFileSync(process.env.ZES_BATCH1M_INPUT));
    assert.strictEqual(all.length, 128 << 20);
    const ins = gold.map((e, k) => all.subarray(k << 20, (k + 1) << 20));
    const t0 = Date.now();
    const comp = zlibes.deflateBatch(ins);
    const t1 = Date.now();
    const back = await zlibes.inflateBatchAsync(comp);
    const t2 = Date.now();
    it('128 x 1 MiB deflateBatch == reference (' + (t1 - t0) + ' ms), inflateBatchAsync back (' + (t2 - t1) + ' ms)', () => {
      for (let k = 0; k < 128; k++) {
        assert.strictEqual(comp[k].length, gold[k].deflate_len, 'length of buffer ' + k);
        assert.strictEqual(sha(comp[k]), gold[k].deflate_sha256, 'bytes of buffer ' + k);
        assert.ok(Buffer.from(back[k]).equals(Buffer.from(ins[k].buffer, ins[k].byteOffset, ins[k].length)), 'round trip of buffer ' + k);
      }
    });
  }
  // extra: the pooled GPU scratch can be handed back, and the calls go on
  it('trim(), then the same bytes again', () => {
    zlibes.trim();
    assert.strictEqual(hex(zlibes.deflate(RAW_BIN)), hex(outs[1]));
  });
  console.log('all ' + n + ' host checks passed');
})().catch((e) => { console.error(e); process.exit(1); });
