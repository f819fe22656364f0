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

type BatchResult = Uint8Array | Error;

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
 * The input array must not be modified, and its ArrayBuffer must not be transferred or detached, until the promise settles.
 */
export function deflateAsync(input: Uint8Array): Promise<Uint8Array> {
  return addon.deflateAsync(input);
}

export function inflateAsync(input: Uint8Array): Promise<Uint8Array> {
  return addon.inflateAsync(input);
}

/**
 * Batch forms (not in the reference API; SURVEY §7 step 3): an array of independent buffers in one call — what a
 * caller's loop over deflate()/inflate() (reference README.md:28-42) becomes when small buffers should share the GPU.
 * Element i of the result is the Uint8Array deflate(inputs[i]) / inflate(inputs[i]) would return, or — instead of a
 * throw — the `Error` it would have thrown (same message).  The Async forms run on a libuv worker thread.
 */
export function deflateBatch(inputs: Uint8Array[]): BatchResult[] {
  return addon.deflateBatch(inputs);
}

export function inflateBatch(inputs: Uint8Array[]): BatchResult[] {
  return addon.inflateBatch(inputs);
}

export function deflateBatchAsync(inputs: Uint8Array[]): Promise<BatchResult[]> {
  return addon.deflateBatchAsync(inputs);
}

export function inflateBatchAsync(inputs: Uint8Array[]): Promise<BatchResult[]> {
  return addon.inflateBatchAsync(inputs);
}

/**
 * Extra: a Uint8Array of n bytes in page-locked memory.  Inputs that live in such an array cross PCIe without the
 * library's staging copy (any Uint8Array is accepted everywhere; this is only faster).
 */
export function allocPinned(n: number): Uint8Array {
  return addon.allocPinned(n);
}

/** Extra (not in the reference API): Adler-32 of a buffer, computed on the GPU. */
export function adler32(input: Uint8Array): number {
  return addon.adler32(input);
}

/** Extra: bind this process to a GPU (defaults to device 0 on first use). */
export function init(device: number): void {
  addon.init(device);
}

/**
 * Extra: drive `n` GPUs from this one process (n omitted or <= 0: every visible one; returns the number in use).  After
 * this the batch forms partition their buffers over all of them by size — a caller's loop over deflate()/inflate()
 * (reference README.md:28-42) spread over the node — and single calls (deflateAsync from several promises at once) take
 * the GPUs in turn.  Results are the same bytes, buffer for buffer.
 */
export function initDevices(n: number = 0): number {
  return addon.initDevices(n);
}

/** Extra: give the library's pooled GPU scratch back to the driver (a long-lived process after one large call). */
export function trim(): void {
  addon.trim();
}
