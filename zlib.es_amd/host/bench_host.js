// bench_host.js — the host_api leg of bench.py through Node: deflate(Uint8Array) / inflate(Uint8Array) of the drop-in
// façade (zlib.js over the N-API addon; reference src/zlib.ts:11,25) on the same inputs bench.py's Python rows use.
//   node bench_host.js '{"inputs":[{"name":..,"path":..,"deflate_len":..,"deflate_sha256":..}],"calls":K,"reps":R}'
// Every call returns a fresh exact-size Uint8Array, as the reference does (src/zlib.ts:42): its allocation is inside the
// time.  Rows: the input in an ordinary Uint8Array (pageable) and in an allocPinned() one; each as a tight synchronous loop
// over deflate() / inflate() (the reference's own calling pattern: V8 collects eight to ten results behind in such a loop, so nearly every
// result is fresh memory) and as an awaited loop over deflateAsync() / inflateAsync() (a host that yields to the event loop:
// collected results' page-locked blocks are back in the addon's pool a few calls later).  Prints ONE JSON line.
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const z = require('./zlib.js');

const spec = JSON.parse(process.argv[2]);
const GIB = 1024 * 1024 * 1024;
const now = () => Number(process.hrtime.bigint()) / 1e9;
const median = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];
const spread = (a) => ({ median: +(median(a) * 1e3).toFixed(4), min: +(Math.min(...a) * 1e3).toFixed(4), max: +(Math.max(...a) * 1e3).toFixed(4) });

const rows = {};
let okAll = true;
async function main() {
  for (const inp of spec.inputs) {
    const file = fs.readFileSync(inp.path);
    const src = new Uint8Array(file.buffer, file.byteOffset, file.length);
    for (const pinned of [false, true]) {
      let a = src;
      if (pinned) {
        a = z.allocPinned(src.length);
        a.set(src);
      }
      for (const mode of ['sync', 'async']) {
        const def = mode === 'sync' ? async (x) => z.deflate(x) : (x) => z.deflateAsync(x);
        const inf = mode === 'sync' ? async (x) => z.inflate(x) : (x) => z.inflateAsync(x);
        let comp = await def(a);  // warm-up: pools, staging
        let back = await inf(comp);
        const td = [], ti = [];
        for (let r = 0; r < spec.reps; r++) {
          const t0 = now();
          if (mode === 'sync') for (let k = 0; k < spec.calls; k++) comp = z.deflate(a);
          else for (let k = 0; k < spec.calls; k++) comp = await z.deflateAsync(a);
          const t1 = now();
          if (mode === 'sync') for (let k = 0; k < spec.calls; k++) back = z.inflate(comp);
          else for (let k = 0; k < spec.calls; k++) back = await z.inflateAsync(comp);
          const t2 = now();
          td.push((t1 - t0) / spec.calls);
          ti.push((t2 - t1) / spec.calls);
        }
        let ok = Buffer.compare(Buffer.from(back.buffer, back.byteOffset, back.length), Buffer.from(src.buffer, src.byteOffset, src.length)) === 0;
        let golden = false;
        if (inp.deflate_sha256) {
          const h = crypto.createHash('sha256').update(Buffer.from(comp.buffer, comp.byteOffset, comp.length)).digest('hex');
          ok = ok && comp.length === inp.deflate_len && h === inp.deflate_sha256;
          golden = true;
        }
        ok = ok && comp.byteOffset === 0 && comp.buffer.byteLength === comp.length && back.buffer.byteLength === back.length;  // src/zlib.ts:42
        okAll = okAll && ok;
        rows[inp.name + '_' + (pinned ? 'pinned' : 'pageable') + '_' + mode] = {
          deflate_gibs: +(src.length / median(td) / GIB).toFixed(3), inflate_gibs: +(src.length / median(ti) / GIB).toFixed(3),
          deflate_ms: spread(td), inflate_ms: spread(ti), compressed_bytes: comp.length, verified_bit_exact: ok, golden_sha256_checked: golden,
        };
        comp = back = null;
        await new Promise((res) => setImmediate(res));  // a turn of the event loop between rows
      }
    }
  }
  console.log(JSON.stringify({
    what: 'the TypeScript façade under Node ' + process.version + ': deflate(Uint8Array) / inflate(Uint8Array) in a synchronous loop and deflateAsync / inflateAsync awaited, ' +
          'a fresh exact-length result array per call (src/zlib.ts:42), ' + spec.calls + ' calls per timed loop, median of ' + spec.reps + ' loops',
    rows: rows, verified_bit_exact: okAll,
  }));
}
main().catch((e) => { console.error(e); process.exit(1); });
