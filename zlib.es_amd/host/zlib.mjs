// ES-module entry of the drop-in (the reference ships dist/esm/zlib.js next to dist/cjs/zlib.js, rollup.config.js:3-17):
// the same functions as zlib.js, which holds the only implementation (the N-API addon is a CommonJS native module).
import { createRequire } from 'module';
const z = createRequire(import.meta.url)('./zlib.js');
export const inflate = z.inflate;
export const deflate = z.deflate;
export const inflateRaw = z.inflateRaw;
export const deflateRaw = z.deflateRaw;
export const inflateAsync = z.inflateAsync;
export const deflateAsync = z.deflateAsync;
export const deflateBatch = z.deflateBatch;
export const inflateBatch = z.inflateBatch;
export const deflateBatchAsync = z.deflateBatchAsync;
export const inflateBatchAsync = z.inflateBatchAsync;
export const allocPinned = z.allocPinned;
export const adler32 = z.adler32;
export const init = z.init;
export const initDevices = z.initDevices;
export const trim = z.trim;
