/**
 * zlib.ts — drop-in for zlib.es's public module (reference src/zlib.ts:11,25; types as in
 * dist/tsc/zlib.d.ts:4-5): the same two synchronous functions, the same thrown `Error`
 * messages, results bit-identical to the reference — computed on an AMD MI355X through the
 * N-API addon (zes_napi.cc -> include/zes.h -> HIP kernels).
 *
 * There is no JavaScript fallback: without the addon or a GPU the call throws.
 *
 * zlib.js next to this file is generated from it by strip_types.py (this image has no tsc);
 * keep to erasable syntax: annotations on parameters / return types only.
 */
const addon = require('./build/zes_napi.node');

export function inflate(input: Uint8Array): Uint8Array {
  return addon.inflate(input);
}

export function deflate(input: Uint8Array): Uint8Array {
  return addon.deflate(input);
}

/**
 * The raw forms the wrapper above encloses, for callers that keep DEFLATE inside another container:
 * `deflateRaw` is the reference's internal `deflate(input)` (src/deflate.ts:14), `inflateRaw` its
 * `inflate(input, offset = 0)` (src/inflate.ts:16) — not exported by the reference's package entry.
 */
export function deflateRaw(input: Uint8Array): Uint8Array {
  return addon.deflateRaw(input);
}

export function inflateRaw(input: Uint8Array, offset: number = 0): Uint8Array {
  return addon.inflateRaw(input, offset);
}

/**
 * Promise-returning forms (not in the reference API, SURVEY §8f.4): the same work on a libuv worker thread, so the
 * JS thread stays free while the GPU runs.  Resolve with the same bytes, reject with the same `Error` messages.
 * The input array must not be modified until the promise settles.
 */
export function deflateAsync(input: Uint8Array): Promise<Uint8Array> {
  return addon.deflateAsync(input);
}

export function inflateAsync(input: Uint8Array): Promise<Uint8Array> {
  return addon.inflateAsync(input);
}

/** Extra (not in the reference API): Adler-32 of a buffer, computed on the GPU. */
export function adler32(input: Uint8Array): number {
  return addon.adler32(input);
}

/** Extra: bind this process to a GPU (defaults to device 0 on first use). */
export function init(device: number): void {
  addon.init(device);
}
