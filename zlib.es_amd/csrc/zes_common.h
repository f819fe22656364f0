// zes_common.h — shared definitions of the gfx950 DEFLATE engine (device + host).
//
// Vocabulary: a *buffer* is one deflate()/inflate() argument; a *block* is one
// 131072-byte slice of a buffer (reference src/const.ts:7) — the unit of parallelism, one
// workgroup each; a *token* is a literal or a (length, distance) match; a *slot* is the
// 131072-byte region of the inflate output a block decodes into.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZES_BLK 131072u
#define ZES_WINDOW 32768u
#define ZES_MAXMATCH 258u
#define ZES_HDR_WORDS 256u  // per-block header bit buffer (1 KiB >= 4.5 kbit worst case)

// One entry per buffer of a batch call.
struct ZesBuf {
  uint64_t in_off;    // byte offset of the buffer inside d_in
  uint64_t n;         // input length in bytes
  uint64_t out_off;   // byte offset of the result inside d_out (16-byte aligned)
  uint64_t cap;       // capacity at out_off
  uint32_t first_blk; // global index of the buffer's first block
  uint32_t nblk;
  uint64_t n_read;    // bytes readable from in_off (>= n): the match finder compares up to 258 bytes past a block's end
                      // (SURVEY A.3); larger than n only when the buffer is a block range of a longer input
  uint32_t flags;     // ZES_BUF_*
  uint32_t start_bit;  // ZES_BUF_RANGE: bit of out_off's first byte at which the range's stream starts (0 .. 127)
};
#define ZES_BUF_RANGE 1u     // raw bit stream of a block range: starts at bit 0, no zlib header / trailer; res.out_len = bits
#define ZES_BUF_CONT 4u      // the range continues another one's stream inside the same dword: that dword is not cleared
#define ZES_BUF_NOTFINAL 2u  // the range's last block is not the input's last one: BFINAL stays 0

// Result per buffer, written by the device, read back by the host.
struct ZesRes {
  uint64_t out_len;
  int32_t status;
  uint32_t aux;
};

// Per-block record of the deflate pipeline.
struct ZesBlk {
  uint32_t buf;      // owning buffer
  uint32_t blk;      // index inside the buffer
  uint32_t len;      // bytes in this block (<= 131072)
  uint32_t ntok;     // tokens produced by the greedy parse
  uint32_t hdr_bits; // dynamic-header bits (after the 3 block bits)
  uint32_t bits;     // total bits of the block incl. the 3 block bits and EOB
  uint64_t bit_off;  // absolute bit offset of the block inside the buffer's output
};

// token encoding shared with include/zes.h and the oracle
#define ZES_TOK_MATCH 0x80000000u
__host__ __device__ static inline uint32_t zes_tok_len(uint32_t t) { return ((t >> 16) & 0xffu) + 3u; }
__host__ __device__ static inline uint32_t zes_tok_dist(uint32_t t) { return (t & 0x7fffu) + 1u; }

#ifdef __HIPCC__
// RFC1951 tables (reference src/const.ts:9-35)
__device__ __constant__ static const uint16_t kLenBase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                                               31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ __constant__ static const uint8_t kLenXbits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                                               2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ __constant__ static const uint16_t kDistBase[30] = {1,    2,    3,    4,    5,    7,    9,    13,   17,   25,
                                                                33,   49,   65,   97,   129,  193,  257,  385,  513,  769,
                                                                1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ __constant__ static const uint8_t kDistXbits[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                                                6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ __constant__ static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// length 3..258 -> code index 0..28 (largest i with base[i] <= len; reference src/lz77.ts:97-102)
__device__ static inline uint32_t zes_len_code(uint32_t len) {
  uint32_t y = len - 3u;
  if (y < 8u) return y;
  if (y == 255u) return 28u;
  uint32_t hb = 31u - (uint32_t)__clz((int)y);  // 3..7
  uint32_t e = hb - 2u;
  return 4u + 4u * e + ((y >> e) & 3u);
}
// distance 1..32768 -> code index 0..29 (reference src/lz77.ts:103-108)
__device__ static inline uint32_t zes_dist_code(uint32_t dist) {
  uint32_t x = dist - 1u;
  if (x < 4u) return x;
  uint32_t hb = 31u - (uint32_t)__clz((int)x);
  return 2u * hb + ((x >> (hb - 1u)) & 1u);
}

__device__ static inline uint32_t zes_lane() { return threadIdx.x & 63u; }
__device__ static inline uint64_t zes_lanemask_lt() { return (1ull << zes_lane()) - 1ull; }
#endif
