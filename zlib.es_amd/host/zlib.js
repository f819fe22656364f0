'use strict';
// GENERATED from zlib.ts by strip_types.py — do not edit.
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

function inflate(input) {
  return addon.inflate(input);
}

function deflate(input) {
  return addon.deflate(input);
}

/** Extra (not in the reference API): Adler-32 of a buffer, computed on the GPU. */
function adler32(input) {
  return addon.adler32(input);
}

/** Extra: bind this process to a GPU (defaults to device 0 on first use). */
function init(device) {
  addon.init(device);
}

exports.inflate = inflate;
exports.deflate = deflate;
exports.adler32 = adler32;
exports.init = init;
