// One Node process, several GPUs (initDevices): the batch forms partition their buffers over every device the library
// drives, single async calls take the devices in turn — results are the reference's bytes, buffer for buffer
// (tests/golden/batch1m.json: BASELINE.json configs[3], made by running the reference).  On a one-GPU box the
// test runs with ZES_OVERSUBSCRIBE=1: three contexts on the same device, the same code paths.
// Run by tests/test_gpu_multidevice.py:  node tests/host_node_multidev_test.js
'use strict';
const assert = require('assert');
const crypto = require('crypto');
const fs = require('fs');
const path = require('path');
const zlibes = require('../zlib.es_amd/host/zlib.js');

const want = parseInt(process.env.ZES_TEST_DEVICES || '3', 10);
const ndev = zlibes.initDevices(want);
assert.strictEqual(ndev, want);
console.log('devices in use: ' + ndev);

const gold = JSON.parse(fs.readFileSync(path.join(__dirname, 'golden', 'batch1m.json'), 'utf8'));
const blob = fs.readFileSync(process.env.ZES_BATCH1M_INPUT);
const count = Math.floor(blob.length / 1048576);
assert.ok(count >= 24);
const inputs = [];
for (let i = 0; i < count; i++) inputs.push(new Uint8Array(blob.buffer, blob.byteOffset + i * 1048576, 1048576));
// ragged extras: sizes the partition has to balance, and two that throw in the reference (n = 1, n = 131073)
inputs.push(inputs[1].subarray(0, 300000), inputs[2].subarray(0, 7), inputs[0].subarray(0, 1), inputs[0].subarray(0, 131073), inputs[4].subarray(0, 900001));
const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.length)).digest('hex');

(async () => {
  const comp = zlibes.deflateBatch(inputs);
  assert.strictEqual(comp.length, inputs.length);
  for (let i = 0; i < count; i++) {
    assert.ok(comp[i] instanceof Uint8Array, 'buffer ' + i);
    assert.strictEqual(comp[i].length, gold[i].deflate_len, 'length of buffer ' + i);
    assert.strictEqual(sha(comp[i]), gold[i].deflate_sha256, 'bytes of buffer ' + i);
  }
  for (let i = count; i < inputs.length; i++) {
    const single = (() => { try { return zlibes.deflate(inputs[i]); } catch (e) { return e; } })();
    if (single instanceof Error) {
      assert.ok(comp[i] instanceof Error && comp[i].message === single.message, 'error of buffer ' + i);
    } else {
      assert.ok(Buffer.from(comp[i]).equals(Buffer.from(single)), 'bytes of extra buffer ' + i);
    }
  }
  console.log('ok - deflateBatch over ' + ndev + ' devices == reference bytes');
  const streams = comp.filter((c) => c instanceof Uint8Array);
  const originals = inputs.filter((_, i) => comp[i] instanceof Uint8Array);
  const back = await zlibes.inflateBatchAsync(streams);
  for (let i = 0; i < streams.length; i++) assert.ok(Buffer.from(back[i]).equals(Buffer.from(originals[i])), 'inflate of buffer ' + i);
  console.log('ok - inflateBatchAsync over ' + ndev + ' devices == inputs');
  // single calls in flight together: the devices in turn
  const many = await Promise.all(inputs.slice(0, 12).map((u) => zlibes.deflateAsync(u)));
  for (let i = 0; i < 12; i++) assert.strictEqual(sha(many[i]), gold[i].deflate_sha256);
  const manyBack = await Promise.all(many.map((u) => zlibes.inflateAsync(u)));
  for (let i = 0; i < 12; i++) assert.ok(Buffer.from(manyBack[i]).equals(Buffer.from(inputs[i])));
  console.log('ok - 12 deflateAsync / inflateAsync calls in flight');
  console.log('multi-device host checks passed');
})().catch((e) => { console.error(e); process.exit(1); });
