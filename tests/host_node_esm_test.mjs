// The ES-module entry (zlib.es_amd/host/zlib.mjs) gives the same functions and bytes as the CommonJS one.
import assert from 'assert';
import { createRequire } from 'module';
import { deflate, inflate, deflateRaw, inflateRaw, deflateAsync, inflateAsync } from '../zlib.es_amd/host/zlib.mjs';
const cjs = createRequire(import.meta.url)('../zlib.es_amd/host/zlib.js');
const data = new Uint8Array(5000);
for (let i = 0; i < data.length; i++) data[i] = (i * 7 + (i >> 5)) & 0xff;
const a = deflate(data), b = cjs.deflate(data);
assert.ok(Buffer.from(a).equals(Buffer.from(b)));
assert.ok(Buffer.from(inflate(a)).equals(Buffer.from(data)));
assert.ok(Buffer.from(inflateRaw(deflateRaw(data))).equals(Buffer.from(data)));
Promise.all([deflateAsync(data), inflateAsync(a)]).then((c) => {  // (Node 12: no top-level await)
  assert.ok(Buffer.from(c[0]).equals(Buffer.from(a)) && Buffer.from(c[1]).equals(Buffer.from(data)));
  console.log('esm entry ok');
}).catch((e) => { console.error(e); process.exit(1); });
