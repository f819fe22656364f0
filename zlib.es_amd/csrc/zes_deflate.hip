// zes_deflate.hip — gfx950 kernels of the compress direction.
//
// Pipeline per 131072-byte block (one workgroup per block unless noted), bit-exact with the
// reference's src/deflate.ts + src/lz77.ts + src/huffman.ts:
//
//   k_lz_sort    index of the block: positions grouped by exact 3-byte key, ascending inside a
//                group — a stable 3-pass LSD radix sort over the LDS-resident block
//                (replaces generateLZ77IndexMap, src/lz77.ts:11-22)
//   k_lz_match   per position: the match the reference's candidate loop would pick
//                (src/lz77.ts:49-93; a pure function of the position)
//   k_lz_parse   greedy chain p -> p+len | p+1, token compaction, LDS symbol histograms
//                (src/lz77.ts:39-47,95-117; src/deflate.ts:58-77)
//   k_huff       package-merge code lengths by parallel merge ranks, canonical codes,
//                code-length RLE, dynamic header bits, block bit count
//                (src/huffman.ts:55-153; src/deflate.ts:78-181)
//   k_layout     per buffer: exclusive scan of block bit counts, zlib header + Adler trailer
//                (src/deflate.ts:20-38; src/zlib.ts:25-49)
//   k_emit       token bit lengths -> scan -> LDS-staged bit packing -> coalesced stores
//                (src/deflate.ts:183-226; src/utils/BitWriteStream.ts)
//   k_adler      Adler-32 as per-chunk (A, B) partial sums combined by u64 atomics
//                (src/adler32.ts:1-10)
#include "zes_common.h"
#include "zes_kernels.h"

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
// unaligned 32-bit little-endian read from an LDS byte array (4-byte aligned base)
__device__ static inline uint32_t lds_ld32u(const uint8_t* s, uint32_t off) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(s);
  uint32_t i = off >> 2;
  uint32_t lo = w[i], hi = w[i + 1];
  return __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
}

// cooperative copy global -> LDS of `bytes` bytes starting at g (any alignment), zero-padding
// up to `padded` (multiple of 16).  s must be 16-byte aligned.
__device__ static inline void stage_block(uint8_t* s, const uint8_t* g, uint32_t bytes, uint32_t padded) {
  const uint32_t tid = threadIdx.x, nth = blockDim.x;
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  if (mis == 0) {
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4* s4 = reinterpret_cast<uint4*>(s);
    const uint32_t full = bytes >> 4;
    for (uint32_t i = tid; i < full; i += nth) s4[i] = g4[i];
    for (uint32_t i = (full << 4) + tid; i < padded; i += nth) s[i] = i < bytes ? g[i] : (uint8_t)0;
  } else {
    for (uint32_t i = tid; i < padded; i += nth) s[i] = i < bytes ? g[i] : (uint8_t)0;
  }
}

// ------------------------------------------------------------------------------------------
// k_lz_sort: the positions of a block that can take part in a match, stably LSD-radix-sorted by
// their 3 key bytes.
//
// Filter first: position p matters to the match finder only if another position of the block
// has the same 3-byte key (as its candidate, or as the position it is a candidate of).  Every
// key is hashed into 2^19 two-bit counters (128 KiB of LDS: the area that holds the block
// afterwards); a position whose counter stays at one has a key no other position shares and is
// dropped.  Hash collisions only keep too many positions, never too few, so the result is exact.
// Incompressible input keeps about a fifth of its positions; text keeps nearly all (the filter
// then costs a few percent).
//
// Result: idx_a[g][0..ns) sorted positions, ns in idx_a[g][ZES_BLK-1] (cnt <= ZES_BLK-2 leaves
// that slot free).  LDS: the block itself (128 KiB) + per-wave digit counters.
// ------------------------------------------------------------------------------------------
// ZES_DEBUG_PHASES: cycle stamps of workgroup 0..: [g][8], set by the API through zes_sort_set_dbg()
__device__ unsigned long long* g_sort_dbg = nullptr;
void zes_sort_set_dbg(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sort_dbg), &p, sizeof p); }
#define SSTAMP(i)                                                                                        \
  do {                                                                                                   \
    if (g_sort_dbg && threadIdx.x == 0) g_sort_dbg[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); \
  } while (0)

#define SORT_HASH_BITS 19u
#define SORT_OWN 128u  // consecutive positions owned by one thread in the filter phase
#ifndef SORT_ROUNDS
#define SORT_ROUNDS 8u
#endif
#define SORT_TILE (SORT_THREADS * SORT_ROUNDS)
static_assert(SORT_ROUNDS == 4u || SORT_ROUNDS == 8u || SORT_ROUNDS == 16u, "the arrival wait names four, eight or sixteen registers");
struct SortSmem {
  uint8_t in[ZES_BLK + 16];  // filter phase: the counter table; then the staged block
  uint32_t hist[3][256];     // digit histograms of the kept positions, one per pass
  uint32_t base[2][256];     // running output offset per digit (double-buffered across tiles)
  uint16_t whist[2][SORT_WAVES][256];
  uint32_t wsum[SORT_WAVES];
  uint32_t nsat;             // filter: positions that found their key already counted twice
  uint32_t heavy;            // dense blocks: sampled positions whose key's class is a heavy one
};
static_assert((1u << SORT_HASH_BITS) * 2u / 8u <= ZES_BLK, "counter table must fit the block area");
static_assert(SORT_OWN * SORT_THREADS == ZES_BLK, "one thread per 128 positions");

__device__ __forceinline__ static uint32_t sort_hash(uint32_t key24) { return (key24 * 0x9E3779B1u) >> (32u - SORT_HASH_BITS); }
// second, independent hash for the second filter level
__device__ __forceinline__ static uint32_t sort_hash2(uint32_t key24) { return ((key24 ^ (key24 >> 11)) * 0xC2B2AE35u) >> (32u - SORT_HASH_BITS); }

// 16 bytes at offset `off` of the block (zero past its end); the 16-byte path needs an aligned block
__device__ __forceinline__ static uint4 sort_ld16(const uint8_t* __restrict__ src, bool aligned, uint32_t off, uint32_t T) {
  if (aligned && off + 16u <= T) return *reinterpret_cast<const uint4*>(src + off);
  uint32_t t[4] = {0, 0, 0, 0};
  for (uint32_t k = 0; k < 16u; k++)
    if (off + k < T) t[k >> 2] |= (uint32_t)src[off + k] << (8u * (k & 3u));
  return make_uint4(t[0], t[1], t[2], t[3]);
}

__global__ __launch_bounds__(SORT_THREADS) void k_lz_sort(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                          const ZesBlk* __restrict__ blks, uint32_t* idx_a,
                                                          uint32_t* __restrict__ idx_b, uint32_t* inv_all, uint16_t* __restrict__ sd_all,
                                                          uint32_t mode) {
  __shared__ __align__(16) SortSmem S;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const ZesBlk bk = blks[g];
  const ZesBuf bf = bufs[bk.buf];
  const uint32_t T = bk.len;
  const uint8_t* src = d_in + bf.in_off + (uint64_t)bk.blk * ZES_BLK;
  uint32_t* A = idx_a + (uint64_t)g * ZES_BLK;
  uint32_t* B = idx_b + (uint64_t)g * ZES_BLK;
  const uint32_t cnt = T >= 3 ? T - 2 : 0;
  // second launch: only the blocks k_lz_index handed back (they are dense: all positions are sorted)
  const bool redo = (mode & 0xffu) == ZES_SORT_MODE_REDO;
  if (redo && !(A[ZES_BLK - 1] & ZES_SORT_REDO)) return;
  if (cnt == 0) {
    if (tid == 0) A[ZES_BLK - 1] = 0;
    return;
  }

  SSTAMP(0);
  // ---- filter: which positions share their key with another position? ----
  uint32_t* tbl = reinterpret_cast<uint32_t*>(S.in);
  {
    uint4* t4 = reinterpret_cast<uint4*>(S.in);
    for (uint32_t i = tid; i < ZES_BLK / 16; i += SORT_THREADS) t4[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = tid; i < 4 * 3 * 256; i += SORT_THREADS) reinterpret_cast<uint32_t*>(&S.whist[0][0][0])[i] = 0;
    if (tid == 0) {
      S.nsat = 0;
      S.heavy = 0;
    }
  }
  __syncthreads();
  SSTAMP(1);
  const bool aligned = (((uintptr_t)src) & 15u) == 0;
  const uint32_t p0 = tid * SORT_OWN;  // this thread's positions: [p0, p0 + 128) below cnt
  // pass 1: count every key (saturating at two: bit 0 = seen, bit 1 = seen again)
  uint32_t samp[SORT_OWN / 16];  // hash of the first key of each 16-position chunk: the survivor rate is sampled on these
#pragma unroll
  for (uint32_t c = 0; c < SORT_OWN / 16; c++) samp[c] = ~0u;
  if (p0 < cnt && !redo) {
    uint4 cur = sort_ld16(src, aligned, p0, T);
#pragma unroll 1
    for (uint32_t c = 0; c < SORT_OWN / 16; c++) {
      const uint4 nxt = sort_ld16(src, aligned, p0 + 16u * (c + 1u), T);
      const uint32_t w[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) {
        const uint32_t key = __builtin_amdgcn_alignbyte(w[(k >> 2) + 1], w[k >> 2], k & 3u) & 0xffffffu;
        const uint32_t h = sort_hash(key);
        const uint32_t sh = (h & 15u) * 2u;
        // a plain read first: keys that repeat often (text, periodic data) are saturated early and
        // would otherwise hammer one word with same-address atomics
        const bool inr = p0 + 16u * c + k < cnt;
        const bool sat = (tbl[h >> 4] >> sh) & 2u;
        if (k == 0 && inr) {
#pragma unroll
          for (uint32_t cc = 0; cc < SORT_OWN / 16; cc++) samp[cc] = (cc == c) ? h : samp[cc];
        }
        if (inr && !sat) {
          const uint32_t old = atomicOr(&tbl[h >> 4], 1u << sh);
          if (((old >> sh) & 3u) == 1u) atomicOr(&tbl[h >> 4], 2u << sh);
        }
      }
      cur = nxt;
    }
  }
  __syncthreads();
  SSTAMP(2);
  // Sample: one position in sixteen.  When three quarters of them are kept, nearly everything would be
  // (text, periodic data): the second filter pass and the compaction are skipped and all positions are
  // sorted (a position with a unique key simply has no neighbour of its key).
  {
    uint32_t kept = 0, tried = 0;
#pragma unroll
    for (uint32_t c = 0; c < SORT_OWN / 16; c++) {
      const uint32_t h = samp[c];
      if (h != ~0u) {
        tried++;
        kept += (tbl[h >> 4] >> ((h & 15u) * 2u + 1u)) & 1u;
      }
    }
    atomicAdd(&S.nsat, kept | (tried << 16));
  }
  __syncthreads();
  const bool dense = redo || (S.nsat & 0xFFFFu) * 4u >= (S.nsat >> 16) * 3u;
#ifndef NO_LAZY
  if (dense && !redo && (mode & ZES_SORT_USE_INDEX) && inv_all != nullptr) {  // (uniform) k_lz_index builds this block's index inside LDS
    // ... unless much of the block's weight sits in a few heavy keys (text: " th", "he "): k_lz_index would count its classes
    // and hand the block back (a second launch of this kernel; text paid 0.08 ms for this launch's filter pass, 0.04 ms for
    // that count and the second launch's start on top of its sort).  The same verdict from this launch's sample, one
    // position in sixteen: the samples are counted per hash class (2048 of them), a class with more than its share of 512
    // positions is heavy, and with a sixteenth of the positions in such classes the block is sorted HERE, now.  The sample
    // only picks the kernel that builds the index — either builds the same candidate lists — so it need not agree with
    // k_lz_index's own count on a block near the line (that kernel still hands back what it cannot take).
    uint32_t* cc = reinterpret_cast<uint32_t*>(&S.whist[0][0][0]);  // [2048], zeroed at the start
    // (fresh samples, at an offset inside each run of sixteen that differs from run to run: the density test's samples sit
    // on multiples of sixteen, and data whose period is one — the 4 KiB pattern — shows them the same 256 keys over and over)
    {
      uint32_t ky[SORT_OWN / 16];
#pragma unroll
      for (uint32_t c = 0; c < SORT_OWN / 16; c++) {
        const uint32_t sp = min(p0 + 16u * c + (((tid * 8u + c) * 0x9E3779B1u) >> 28), T >= 8u ? T - 8u : 0u);  // (the eight bytes read stay inside the block)
        uint32_t w = 0, w2 = 0;
        if (T >= 8u) {
          __builtin_memcpy(&w, src + (sp & ~3u), 4);
          __builtin_memcpy(&w2, src + (sp & ~3u) + 4u, 4);
        }
        ky[c] = __builtin_amdgcn_alignbyte(w2, w, sp & 3u) & 0xffffffu;
      }
#pragma unroll
      for (uint32_t c = 0; c < SORT_OWN / 16; c++)
        if (p0 + 16u * c < cnt) atomicAdd(&cc[sort_hash(ky[c]) >> (SORT_HASH_BITS - 11u)], 1u);
    }
    __syncthreads();
    {
      const uint32_t thr = max(2u, (32u * cnt + (ZES_BLK - 1u)) >> 17);  // a class's share of samples when it holds 512 positions of a full block
      const uint32_t a = cc[2u * tid], b = cc[2u * tid + 1u];
      const uint32_t w = (a > thr ? a : 0u) + (b > thr ? b : 0u);
      if (w) atomicAdd(&S.heavy, w);
    }
    __syncthreads();
    if (S.heavy * 256u <= cnt) {  // (uniform) samples are a sixteenth of the positions, heavy classes below a sixteenth of those
      if (tid == 0) A[ZES_BLK - 1] = cnt | ZES_SORT_LAZY | ZES_SORT_INDEX;
      return;
    }
    cc[2u * tid] = 0;  // (the dense path's byte histograms expect the area as it was)
    cc[2u * tid + 1u] = 0;
    __syncthreads();
  }
#endif
  // Few kept (incompressible data: a fifth, nearly all of them collisions of the hash, not repeats of a key): a second
  // filter level over the kept ones with another hash leaves ~2 % of the positions, and the three sorting passes —
  // two thirds of this kernel on such data — have a twelfth of the elements.
  const bool two = !redo && (S.nsat & 0xFFFFu) * 5u < (S.nsat >> 16) * 2u;
  uint32_t ns = cnt;
  if (!dense) {
  // pass 2: keep the positions whose counter reached two; 128 flags per thread
  uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
  uint32_t(*sub)[256] = reinterpret_cast<uint32_t(*)[256]>(&S.whist[0][0][0]) + 3u * (lane & 3u);  // 4 x 3 x 256 u32 = 12 KiB
  if (p0 < cnt) {
    uint4 cur = sort_ld16(src, aligned, p0, T);
#pragma unroll 1
    for (uint32_t c = 0; c < SORT_OWN / 16; c++) {
      const uint4 nxt = sort_ld16(src, aligned, p0 + 16u * (c + 1u), T);
      const uint32_t w[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) {
        const uint32_t key = __builtin_amdgcn_alignbyte(w[(k >> 2) + 1], w[k >> 2], k & 3u) & 0xffffffu;
        const uint32_t h = sort_hash(key);
        const uint32_t keep = (tbl[h >> 4] >> ((h & 15u) * 2u + 1u)) & 1u;
        if (keep && p0 + 16u * c + k < cnt) {
          bits |= 1u << k;
          // digit histograms of the kept positions (pass 0 sorts by byte 2, pass 2 by byte 0), four
          // copies picked by lane: neighbouring lanes see the same common bytes at the same time
          if (!two) {
            atomicAdd(&sub[0][(key >> 16) & 255u], 1u);
            atomicAdd(&sub[1][(key >> 8) & 255u], 1u);
            atomicAdd(&sub[2][key & 255u], 1u);
          }
        }
      }
      const uint32_t v = bits << (16u * (c & 1u));
      f0 |= (c >> 1) == 0u ? v : 0u;
      f1 |= (c >> 1) == 1u ? v : 0u;
      f2 |= (c >> 1) == 2u ? v : 0u;
      f3 |= (c >> 1) == 3u ? v : 0u;
      cur = nxt;
    }
  }
  if (two) {  // uniform
    __syncthreads();  // every thread is done with the first level's counters
    {
      uint4* t4 = reinterpret_cast<uint4*>(S.in);
      for (uint32_t i = tid; i < ZES_BLK / 16; i += SORT_THREADS) t4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // level 2, count: the kept positions only, under the second hash
    if (p0 < cnt) {
      uint4 cur = sort_ld16(src, aligned, p0, T);
#pragma unroll 1
      for (uint32_t c = 0; c < SORT_OWN / 16; c++) {
        const uint4 nxt = sort_ld16(src, aligned, p0 + 16u * (c + 1u), T);
        const uint32_t w[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
        const uint32_t fw = (c >> 1) == 0u ? f0 : (c >> 1) == 1u ? f1 : (c >> 1) == 2u ? f2 : f3;
        const uint32_t bits = (fw >> (16u * (c & 1u))) & 0xffffu;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
          const uint32_t key = __builtin_amdgcn_alignbyte(w[(k >> 2) + 1], w[k >> 2], k & 3u) & 0xffffffu;
          const uint32_t h = sort_hash2(key);
          const uint32_t sh = (h & 15u) * 2u;
          if ((bits >> k) & 1u) {
            const uint32_t old = atomicOr(&tbl[h >> 4], 1u << sh);
            if (((old >> sh) & 3u) == 1u) atomicOr(&tbl[h >> 4], 2u << sh);
          }
        }
        cur = nxt;
      }
    }
    __syncthreads();
    // level 2, flags: kept by both levels; their digit histograms
    if (p0 < cnt) {
      uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
      uint4 cur = sort_ld16(src, aligned, p0, T);
#pragma unroll 1
      for (uint32_t c = 0; c < SORT_OWN / 16; c++) {
        const uint4 nxt = sort_ld16(src, aligned, p0 + 16u * (c + 1u), T);
        const uint32_t w[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
        const uint32_t fw = (c >> 1) == 0u ? f0 : (c >> 1) == 1u ? f1 : (c >> 1) == 2u ? f2 : f3;
        const uint32_t bits = (fw >> (16u * (c & 1u))) & 0xffffu;
        uint32_t nb = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
          const uint32_t key = __builtin_amdgcn_alignbyte(w[(k >> 2) + 1], w[k >> 2], k & 3u) & 0xffffffu;
          const uint32_t h = sort_hash2(key);
          const uint32_t keep = (tbl[h >> 4] >> ((h & 15u) * 2u + 1u)) & 1u;
          if (keep && ((bits >> k) & 1u)) {
            nb |= 1u << k;
            atomicAdd(&sub[0][(key >> 16) & 255u], 1u);
            atomicAdd(&sub[1][(key >> 8) & 255u], 1u);
            atomicAdd(&sub[2][key & 255u], 1u);
          }
        }
        const uint32_t v = nb << (16u * (c & 1u));
        g0 |= (c >> 1) == 0u ? v : 0u;
        g1 |= (c >> 1) == 1u ? v : 0u;
        g2 |= (c >> 1) == 2u ? v : 0u;
        g3 |= (c >> 1) == 3u ? v : 0u;
        cur = nxt;
      }
      f0 = g0;
      f1 = g1;
      f2 = g2;
      f3 = g3;
    }
  }
  SSTAMP(3);
  // ordered compaction into B.  Exclusive scan of the per-thread counts; then the flag words and
  // their running totals go to LDS (the counter table is dead by then) so that the list can be
  // written with neighbouring lanes on neighbouring positions: coalesced stores.
  const uint32_t mine = (uint32_t)(__popc(f0) + __popc(f1) + __popc(f2) + __popc(f3));
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d);
    if ((int)lane >= d) incl += t;
  }
  if (lane == 63) S.wsum[wave] = incl;
  __syncthreads();  // also: every thread is done with the counter table
  uint32_t woff = 0;
  ns = 0;
#pragma unroll
  for (uint32_t w = 0; w < SORT_WAVES; w++) {
    const uint32_t v = S.wsum[w];
    woff += w < wave ? v : 0u;
    ns += v;
  }
  uint32_t* fbits = reinterpret_cast<uint32_t*>(S.in);         // [4096] flag words
  uint32_t* fpre = reinterpret_cast<uint32_t*>(S.in) + 4096;   // [4096] kept positions before each word
  if (two) {  // a few positions per thread: straight from the flag words in registers
    uint32_t o = woff + incl - mine;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
      uint32_t f = w == 0u ? f0 : w == 1u ? f1 : w == 2u ? f2 : f3;
      while (f) {
        B[o++] = p0 + 32u * w + (uint32_t)__builtin_ctz(f);
        f &= f - 1u;
      }
    }
  } else {
    uint32_t o = woff + incl - mine;
    fbits[4 * tid + 0] = f0;
    fpre[4 * tid + 0] = o;
    o += (uint32_t)__popc(f0);
    fbits[4 * tid + 1] = f1;
    fpre[4 * tid + 1] = o;
    o += (uint32_t)__popc(f1);
    fbits[4 * tid + 2] = f2;
    fpre[4 * tid + 2] = o;
    o += (uint32_t)__popc(f2);
    fbits[4 * tid + 3] = f3;
    fpre[4 * tid + 3] = o;
  }
  // fold the histogram copies
  if (tid < 3 * 256) {
    const uint32_t* hs = reinterpret_cast<const uint32_t*>(&S.whist[0][0][0]);
    (&S.hist[0][0])[tid] = hs[tid] + hs[768 + tid] + hs[1536 + tid] + hs[2304 + tid];
  }
  __syncthreads();
  if (!two) {
    for (uint32_t p = tid; p < cnt; p += SORT_THREADS) {
      const uint32_t w = fbits[p >> 5];
      if ((w >> (p & 31u)) & 1u) B[fpre[p >> 5] + (uint32_t)__popc(w & ((1u << (p & 31u)) - 1u))] = p;
    }
  }
  }  // !dense
  __syncthreads();  // flag words (or, dense, the counter table) read before the block is staged over them
  SSTAMP(4);
  // Blocks that keep most of their positions (text, periodic data) go to the lazy match finder, which
  // evaluates positions along greedy chains and needs the sorted slot of a position: inv[p].
#ifdef NO_LAZY
  const bool lazy = false;
#else
  const bool lazy = inv_all != nullptr && ns * 2u >= cnt;
#endif
  if (tid == 0) A[ZES_BLK - 1] = ns | (lazy ? ZES_SORT_LAZY : 0u);
  if (ns == 0) return;  // uniform

  stage_block(S.in, src, T, (T + 15u) & ~15u);
  __syncthreads();
  if (dense) {
    // digit histograms of all positions: the block's byte histogram (four copies by lane) minus the few
    // bytes a pass does not see (pass k sorts by the byte at offset 2 - k of positions 0 .. cnt-1)
    uint32_t(*hc)[256] = reinterpret_cast<uint32_t(*)[256]>(&S.whist[0][0][0]);  // zeroed at the start, [4] used
    for (uint32_t i = tid; i < T; i += SORT_THREADS) atomicAdd(&hc[lane & 3u][S.in[i]], 1u);
    __syncthreads();
    if (tid < 256) {
      const uint32_t H = hc[0][tid] + hc[1][tid] + hc[2][tid] + hc[3][tid];
#pragma unroll
      for (uint32_t ps = 0; ps < 3; ps++) {
        const uint32_t off = 2u - ps;
        uint32_t h = H;
        for (uint32_t e = 0; e < off; e++) h -= (S.in[e] == tid);
        for (uint32_t e = off + cnt; e < T; e++) h -= (S.in[e] == tid);
        S.hist[ps][tid] = h;
      }
    }
    __syncthreads();
  }
  SSTAMP(5);  // staged block and the survivor list in B visible to the whole workgroup

  uint32_t bsel = 0;
  for (int pass = 0; pass < 3; pass++) {
    const uint32_t off = 2u - (uint32_t)pass;  // least significant key byte first
    const uint32_t* from = (pass == 1) ? A : B;
    uint32_t* to = (pass == 1) ? B : A;
    {  // the first tile's counter row of this wave (the last pass may have left its offsets there; the dense path its histograms)
      uint32_t* w0 = reinterpret_cast<uint32_t*>(&S.whist[0][wave][0]);
      w0[lane] = 0;
      w0[lane + 64u] = 0;
    }
    // exclusive scan of this pass's digit histogram
    if (wave == 0) {
      uint32_t c[4];
      uint32_t sum = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        c[k] = S.hist[pass][lane * 4 + k];
        sum += c[k];
      }
      uint32_t in2 = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(in2, d);
        if ((int)lane >= d) in2 += t;
      }
      uint32_t run = in2 - sum;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        S.base[bsel][lane * 4 + k] = run;
        run += c[k];
      }
    }
    __syncthreads();
    // Tiles of 8192 elements: wave w takes elements [512 w, 512 w + 512) of the tile in eight rounds of
    // 64, so the two workgroup barriers and the cross-wave scan are paid once per 8192 elements.
    const uint32_t ntiles = (ns + SORT_TILE - 1) / SORT_TILE;
    uint32_t pn[SORT_ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < SORT_ROUNDS; r++) pn[r] = from[min(wave * (SORT_TILE / SORT_WAVES) + lane + 64u * r, ns - 1u)];
    for (uint32_t t = 0; t < ntiles; t++) {
      uint16_t(*wh)[256] = S.whist[t & 1];
      const uint32_t i0 = t * SORT_TILE + wave * (SORT_TILE / SORT_WAVES) + lane;
      uint32_t p[SORT_ROUNDS], d[SORT_ROUNDS], rk[SORT_ROUNDS];
      // The tile's indices were requested before the previous tile's scatter (unconditional loads with
      // clamped addresses).  One wait for all of them here: a load or store under a branch makes the
      // compiler wait for every outstanding memory operation, the scatter stores included, and those
      // would then run one at a time.
#pragma unroll
      for (uint32_t r = 0; r < SORT_ROUNDS; r++) p[r] = (dense && pass == 0) ? min(i0 + 64u * r, ns - 1u) : pn[r];  // dense: the list is 0, 1, 2, ...
      asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
#if SORT_ROUNDS >= 8u
      asm volatile("" : "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
#endif
#if SORT_ROUNDS == 16u
      asm volatile("" : "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]));
      asm volatile("" : "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]));
#endif
#pragma unroll
      for (uint32_t r = 0; r < SORT_ROUNDS; r++) d[r] = (i0 + 64u * r < ns) ? S.in[p[r] + off] : 0u;
      // Rank of an element among the wave's elements of its digit, this round's lower lanes and all earlier rounds: the old
      // value of ONE returning LDS add on the wave's own counter row (two 16-bit counters per word: a wave counts at most
      // 512 per tile).  Lanes of one instruction that meet in a word are served in ascending lane order on this hardware
      // (tools/micro/lds_atomic_order.hip: 0 of 3.3e10 values out of order; tests/test_gpu_hw_props.py checks the box it runs
      // on), and a wave's LDS instructions execute in program order: the ranks are the stable ones.  (Before: eight ballots
      // and sixteen mask updates per round to find the lanes of the same digit.)  The row is this wave's own until the
      // barrier: it was zeroed by this wave, behind its own last read of it (below).
      uint32_t* wrow = reinterpret_cast<uint32_t*>(&wh[wave][0]);
#pragma unroll
      for (uint32_t r = 0; r < SORT_ROUNDS; r++) {
        const bool valid = i0 + 64u * r < ns;
        const uint32_t sh = 16u * (d[r] & 1u);
        const uint32_t old = valid ? atomicAdd(&wrow[d[r] >> 1], 1u << sh) : 0u;
        rk[r] = (old >> sh) & 0xffffu;
      }
      __syncthreads();  // (B) counts visible
      if (tid < 256) {
        uint32_t c[SORT_WAVES];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) c[w] = wh[w][tid];
        uint32_t o = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) {
          wh[w][tid] = (uint16_t)o;  // offset of wave w inside this tile's run of digit tid
          o += c[w];
        }
        S.base[bsel ^ 1][tid] = S.base[bsel][tid] + o;
      }
      __syncthreads();  // (C) offsets visible
      {  // the next tile's row of this wave: read last by the tile before this one (its scan, this wave's scatter)
        uint32_t* wnext = reinterpret_cast<uint32_t*>(&S.whist[(t + 1u) & 1u][wave][0]);
        wnext[lane] = 0;
        wnext[lane + 64u] = 0;
      }
#pragma unroll
      for (uint32_t r = 0; r < SORT_ROUNDS; r++) pn[r] = from[min(i0 + SORT_TILE + 64u * r, ns - 1u)];  // next tile
      // unconditional stores for the same reason (slot ZES_BLK-2 of either array is never used:
      // ns <= ZES_BLK-2, and ns itself lives in slot ZES_BLK-1)
#pragma unroll
      for (uint32_t r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t dst = S.base[bsel][d[r]] + wh[wave][d[r]] + rk[r];
        const bool valid = i0 + 64u * r < ns;
        to[valid ? dst : ZES_BLK - 2u] = p[r];
      }
      bsel ^= 1;
      // no barrier here: the next tile's adds go to whist[other], rows this wave zeroed itself; the scan writes
      // whist[other] and base[other-other] behind barrier (B), which every thread reaches after this scatter
    }
    __syncthreads();
    SSTAMP(6);
  }
  // For the lazy match finder, which walks the candidates of a position most recent first (src/lz77.ts:65):
  //   sd[r]   distance from the position in sorted slot r to the one in slot r-1 when both hold the same key and lie
  //           within 32768 of each other, else 0: the candidates of the position in slot r are reached by subtracting
  //           sd[r], sd[r-1], ... until a 0 — consecutive 2-byte entries, one cache line for 32 candidates;
  //   inv[p]  sorted slot of p (17 bits) | sd of that slot - 1 (15 bits), or ZES_INV_NONE when p has no candidate
  //           (none inside the window: src/lz77.ts:49, or the filter dropped p).
  // sd[] leaves with coalesced stores.  Scattering inv[] straight to memory costs a whole memory transaction per
  // entry, so it is built in LDS in four slices of 32768 positions (the staged block is no longer needed once the
  // same-key flags are taken) and each slice leaves with coalesced 16-byte stores; the sorted list is re-read per
  // slice (it sits in the L2).
  if (lazy) {
    // Round 3: one pass over the sorted list instead of five.  Every slot gives its sd[] entry (coalesced) and the word
    // position-in-slice | slot | has-a-candidate, appended to the bucket of the position's 16384-byte slice (eight LDS
    // cursors; the buckets live in idx_b, free since the last pass); then the buckets come back slice by slice, are
    // scattered into an LDS image of the slice and leave as inv[] with 16-byte stores — the transposition k_lz_index
    // uses.  (Before: the same-key flags in one pass, then four passes over the whole list, one per 32768-position slice
    // of inv[]: 300k of the kernel's 1.2M cycles on text, and 32 of its 51 bytes of traffic per position.)
    // inv[p] holds the slot alone; the match finder takes a position's first distance from sd[] like the others.
    uint32_t* inv = inv_all + (uint64_t)g * ZES_BLK;
    uint16_t* sd = sd_all + (uint64_t)g * ZES_BLK;
    uint32_t* P = B;  // [8][16384]
    uint32_t* pcur = S.wsum;
    static_assert(SORT_WAVES >= 8, "eight bucket cursors live in wsum[]");
    if (tid < 8u) pcur[tid] = 0;
    __syncthreads();
    for (uint32_t rb = tid; rb < ns; rb += 8u * SORT_THREADS) {  // eight slots per thread and step: one memory latency for all
      uint32_t pos[8], prv[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        const uint32_t r = min(rb + k * SORT_THREADS, ns - 1u);
        pos[k] = A[r];
        prv[k] = A[r ? r - 1u : 0u];
      }
      asm volatile("" : "+v"(pos[0]), "+v"(pos[1]), "+v"(pos[2]), "+v"(pos[3]), "+v"(pos[4]), "+v"(pos[5]), "+v"(pos[6]), "+v"(pos[7]));
      asm volatile("" : "+v"(prv[0]), "+v"(prv[1]), "+v"(prv[2]), "+v"(prv[3]), "+v"(prv[4]), "+v"(prv[5]), "+v"(prv[6]), "+v"(prv[7]));
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        const uint32_t r = rb + k * SORT_THREADS;  // (lanes of a wave: 64 consecutive slots)
        const bool valid = r < ns;
        // the key of slot r-1 comes from the lane below; lane 0 reads it
        const uint32_t key = valid ? (lds_ld32u(S.in, pos[k]) & 0xffffffu) : 0u;
        uint32_t kprev = (uint32_t)__shfl_up((int)key, 1);
        if (lane == 0 && valid) kprev = lds_ld32u(S.in, prv[k]) & 0xffffffu;
        const bool sm = valid && r != 0u && key == kprev;  // slot r holds the key of slot r-1
        const uint32_t delta = pos[k] - prv[k];  // > 0: equal keys are in ascending position order
        const bool has = sm && delta <= ZES_WINDOW;
        const uint32_t sl = pos[k] >> 14;
        // One cursor add per slice and round, by the first lane of the slice's lanes (found with three ballots): 64 returning
        // adds on eight cursors were served one lane at a time, ~130k of the kernel's 1.1M cycles on text.
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int bb = 0; bb < 3; bb++) {
          const bool bit = (sl >> bb) & 1u;
          const uint64_t bal = __ballot(bit);
          m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(m & zes_lanemask_lt());
        uint32_t at = 0;
        if (valid && rank == 0u) at = atomicAdd(&pcur[sl], (uint32_t)__popcll(m));
        at = (uint32_t)__shfl((int)at, valid ? (int)__builtin_ctzll(m) : 0) + rank;
        if (valid) {
          sd[r] = (uint16_t)(has ? delta : 0u);  // (lanes of a wave write 64 consecutive entries)
          P[sl * 16384u + at] = (pos[k] & 16383u) | (r << 14) | (has ? 0x80000000u : 0u);
        }
      }
    }
    // the bucket words are in memory before anybody reads them back (a workgroup-scope fence does not wait for stores);
    // the block in S.in is dead from here on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* stage = reinterpret_cast<uint32_t*>(S.in);  // [16384]
    for (uint32_t sl = 0; sl < ZES_BLK / 16384u; sl++) {
      const uint32_t lo = sl * 16384u;
      if (lo >= T) break;  // uniform
      const uint32_t have = pcur[sl];  // (positions the filter dropped have no word: their entries stay "none")
      uint4* st4 = reinterpret_cast<uint4*>(S.in);
      for (uint32_t i = tid; i < 4096u; i += SORT_THREADS) st4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
      __syncthreads();
      {
        uint32_t e[16];  // the slice's bucket: all loads in flight at once
#pragma unroll
        for (uint32_t k = 0; k < 16; k++)
          e[k] = __hip_atomic_load(P + lo + min(k * SORT_THREADS + tid, have ? have - 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (uint32_t k = 0; k < 16; k++)
          if (k * SORT_THREADS + tid < have && (e[k] >> 31)) stage[e[k] & 16383u] = (e[k] >> 14) & 0x1FFFFu;
      }
      // (inv_all may BE idx_a — the library passes one array for both: the sorted list is dead by now, eager blocks have no
      // inv[] — so the block's flag word, slot ZES_BLK-1, travels with the last slice; no position has that number)
      if (tid == 0 && lo + 16384u == ZES_BLK) stage[16383] = ns | ZES_SORT_LAZY;
      __syncthreads();
      uint4* o4 = reinterpret_cast<uint4*>(inv + lo);
      for (uint32_t i = tid; i < 4096u; i += SORT_THREADS) o4[i] = st4[i];
      __syncthreads();
    }
  }
  SSTAMP(7);
}

// ------------------------------------------------------------------------------------------
// k_lz_match: one thread per sorted slot r; candidates are the slots before r with the same key.
// LDS: block + 258-byte halo (compares run past the block end up to the input end),
// bank-swizzled: lanes of a wave handle neighbouring slots = the same 3-byte key, and in
// periodic data those positions are a multiple of the period apart (4096 B for lowent4k) —
// unswizzled, that is a 32-way bank conflict on every read.
// match_out[g][p] = ZES_TOK_MATCH | (len-3)<<16 | (dist-1), or 0 for "literal here".
// ------------------------------------------------------------------------------------------
#define MATCH_IN_DWORDS ((ZES_BLK + 288) / 4)
#define MATCH_SAME 0x80000000u
#define MATCH_WAVES (MATCH_THREADS / 64)
#define MATCH_RING 384u  // sorted-index entries a wave keeps in LDS (6 chunks of 64): a job looks back <= 128 slots
struct MatchSmem {
  uint32_t in[(MATCH_IN_DWORDS + 31) / 32 * 32];  // whole 32-dword rows: the swizzle permutes inside a row
  uint32_t ring[MATCH_WAVES][MATCH_RING];         // position | same-key-as-previous-slot flag (bit 31)
  uint32_t nml;                                   // matches found so far (they are listed in global memory, for k_lz_parse)
};

__device__ __forceinline__ static uint32_t mswz(uint32_t i) { return i ^ ((i >> 5) & 31u) ^ ((i >> 10) & 31u); }
__device__ __forceinline__ static uint32_t m_ld32u(const uint32_t* w, uint32_t off) {  // unaligned LE dword at byte offset
  const uint32_t i = off >> 2;
  return __builtin_amdgcn_alignbyte(w[mswz(i + 1)], w[mswz(i)], off & 3u);
}

// Work is data dependent (0..128 candidates per position, compares of 3..258 bytes), so lanes are
// persistent: each lane runs a small state machine (probe the next candidate | compare 4 more
// bytes) and takes the next sorted slot of its wave's range as soon as its position is finished.
// After staging there is no workgroup barrier: a wave owns a contiguous range of sorted slots and
// keeps the index entries it needs in its own LDS ring.
__global__ __launch_bounds__(MATCH_THREADS) void k_lz_match(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                            const ZesBlk* __restrict__ blks, const uint32_t* __restrict__ idx_a,
                                                            uint32_t* __restrict__ match_out, uint32_t* __restrict__ mlist_all) {
  __shared__ __align__(16) MatchSmem S;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const ZesBlk bk = blks[g];
  const ZesBuf bf = bufs[bk.buf];
  const uint32_t T = bk.len;
  const uint64_t S0 = (uint64_t)bk.blk * ZES_BLK;  // block start inside the buffer
  const uint8_t* src = d_in + bf.in_off + S0;
  const uint64_t remain = bf.n_read - S0;  // bytes from block start to input end
  const uint32_t avail = (uint32_t)(remain < (uint64_t)(T + ZES_MAXMATCH) ? remain : (uint64_t)(T + ZES_MAXMATCH));
  const uint32_t* idx = idx_a + (uint64_t)g * ZES_BLK;
  uint32_t* mo = match_out + (uint64_t)g * ZES_BLK;
  if (idx[ZES_BLK - 1] & ZES_SORT_LAZY) return;  // this block belongs to k_lz_match_lazy
  const uint32_t cnt = idx[ZES_BLK - 1];  // positions k_lz_sort kept (the others share their key with nobody)

  // stage block + halo (zero padded), swizzled
  if ((((uintptr_t)src) & 15u) == 0) {
    const uint4* g4 = reinterpret_cast<const uint4*>(src);
    for (uint32_t i = tid; i < MATCH_IN_DWORDS / 4; i += MATCH_THREADS) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const uint32_t b = i * 16u;
      if (b + 16u <= avail) {
        v = g4[i];
      } else if (b < avail) {
        uint32_t t[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < 16u && b + k < avail; k++) t[k >> 2] |= (uint32_t)src[b + k] << (8u * (k & 3u));
        v = make_uint4(t[0], t[1], t[2], t[3]);
      }
      S.in[mswz(4 * i + 0)] = v.x;
      S.in[mswz(4 * i + 1)] = v.y;
      S.in[mswz(4 * i + 2)] = v.z;
      S.in[mswz(4 * i + 3)] = v.w;
    }
  } else {
    for (uint32_t i = tid; i < MATCH_IN_DWORDS; i += MATCH_THREADS) {
      uint32_t t = 0;
      for (uint32_t k = 0; k < 4u; k++)
        if (4 * i + k < avail) t |= (uint32_t)src[4 * i + k] << (8u * k);
      S.in[mswz(i)] = t;
    }
  }
  // Results are written in sorted-key order, i.e. scattered over the block: every 4-byte store
  // costs a whole memory transaction (measured 8x write amplification).  Most positions have
  // no match, so the block's result words are zero-filled with coalesced 16-byte stores first
  // and only real matches are scattered.  This also covers the last two positions, which are
  // always literals (src/lz77.ts:116-117).
  // The matches are also listed (up to ZES_MLIST_CAP; incompressible data has ~60 a block): k_lz_parse builds the greedy
  // chain of such a block from the list instead of searching all 131072 result words for it, and reads a result word
  // only where a listed match is taken.  A block with no more sorted slots than the list holds cannot overflow it — every
  // match belongs to a slot — so its 512 KiB of result words are not cleared at all (half of this kernel's time on
  // incompressible data, 256 MiB of stores per 64 MiB of input).
  const bool sparse = cnt <= ZES_MLIST_CAP;  // (uniform)
  if (!sparse) {
    uint4* mo4 = reinterpret_cast<uint4*>(mo);
    const uint32_t n4 = (T + 3u) >> 2;
    for (uint32_t i = tid; i < n4; i += MATCH_THREADS) mo4[i] = make_uint4(0, 0, 0, 0);
  }
  uint32_t* gl = mlist_all + (uint64_t)g * ZES_MLIST_WORDS;
  if (tid == 0) S.nml = 0;  // (the count in LDS: a global atomic per match cost 0.02 ms on random64; the entries go straight out)
  __syncthreads();  // from here on every wave works alone, up to the count's way out at the very end

  const uint32_t R = ((cnt + MATCH_WAVES * 64u - 1) / (MATCH_WAVES * 64u)) * 64u;
  const uint32_t r0 = min(cnt, wave * R), r1 = min(cnt, r0 + R);
  if (r0 < r1) {  // (a wave without slots goes straight to the flush below)
  uint32_t* ring = S.ring[wave];
  const uint32_t rb = r0 >= 128u ? r0 - 128u : 0u;  // first slot kept in the ring (multiple of 64)

  uint32_t hi = rb;           // next slot to load into the ring
  uint32_t next = r0;         // next slot to hand out
  uint32_t lastkey = 0;       // key of slot hi-1 (valid when hi > 0)
  if (rb > 0) lastkey = m_ld32u(S.in, idx[rb - 1]) & 0xffffffu;
  uint32_t pf = (hi + lane < r1) ? idx[hi + lane] : 0u;  // prefetched positions of the chunk at `hi`

  // per-lane job state
  uint32_t mode = 0;  // 0 idle, 1 probe, 2 compare
  uint32_t jr = 0, p = 0, rp = 0, curflag = 0, best = 0, bestq = 0, check = 0, pprobe = 0, maxl = 0;
  uint32_t q = 0;

  for (;;) {
    const uint32_t nprobe = (uint32_t)__popcll(__ballot(mode == 1u));
    const uint64_t idle = __ballot(mode == 0u);
    const uint32_t npend = 64u - nprobe - (uint32_t)__popcll(idle);
    // ---- housekeeping (ring refill + slot hand-out), batched: only when >= 8 lanes are idle or
    //      nothing can probe ----
    if (next < r1 && ((uint32_t)__popcll(idle) >= 8u || nprobe == 0u)) {
      if (hi < r1 && next + 64u > hi) {
        // the chunk overwrites slots [hi-384, hi-320): every running job must have r - 128 >= hi - 320
        const bool too_old = mode != 0u && hi >= 192u && jr < hi - 192u;
        if (!__ballot(too_old)) {
          const uint32_t r = hi + lane;
          const bool valid = r < r1;
          const uint32_t pos = pf;
          const uint32_t key = valid ? (m_ld32u(S.in, pos) & 0xffffffu) : 0xFFFFFFFFu;
          uint32_t pk = (uint32_t)__shfl_up((int)key, 1);
          if (lane == 0) pk = lastkey;
          const bool same = valid && r > 0u && key == pk;
          ring[((hi - rb) % MATCH_RING) + lane] = pos | (same ? MATCH_SAME : 0u);
          lastkey = (uint32_t)__shfl((int)key, 63);
          hi += 64u;
          pf = (hi + lane < r1) ? idx[hi + lane] : 0u;
        }
      }
      const uint32_t lim = min(hi, r1);
      if (idle && next < lim) {
        const uint32_t k = (uint32_t)__popcll(idle & zes_lanemask_lt());
        const uint32_t navail = lim - next;
        if (mode == 0u && k < navail) {
          jr = next + k;
          rp = (jr - rb) % MATCH_RING;
          const uint32_t e = ring[rp];
          p = e & 0x7fffffffu;
          curflag = e >> 31;
          best = 0;
          bestq = 0;
          check = 0;
          maxl = min(ZES_MAXMATCH, avail - p);  // = min(258, n - p)
          mode = 1u;
        }
        next += min((uint32_t)__popcll(idle), navail);
      }
    }
    // ---- compares, batched and run to completion: lanes whose candidate passed the probe wait in
    //      mode 2 until 8 of them are pending (or nothing else can run) ----
    if (npend >= 8u || (npend != 0u && nprobe == 0u)) {
      // straight-line select code on purpose: nested if/else becomes exec-mask bookkeeping on the
      // scalar unit, which all 16 waves of the CU share (measured: the kernel was SALU-bound)
      bool cmp = mode == 2u;
      uint32_t L = 3;
      const uint32_t qo = q + 3u, po = p + 3u;
      uint32_t qi = qo >> 2, pi = po >> 2;
      const uint32_t qs = qo & 3u, ps = po & 3u;
      uint32_t qlo = S.in[mswz(qi)], plo = S.in[mswz(pi)];
      while (__ballot(cmp && L < maxl)) {  // 8 bytes per step
        const uint32_t qm = S.in[mswz(qi + 1)], pm = S.in[mswz(pi + 1)];
        const uint32_t qhi = S.in[mswz(qi + 2)], phi = S.in[mswz(pi + 2)];
        const uint32_t x1 = __builtin_amdgcn_alignbyte(qm, qlo, qs) ^ __builtin_amdgcn_alignbyte(pm, plo, ps);
        const uint32_t x2 = __builtin_amdgcn_alignbyte(qhi, qm, qs) ^ __builtin_amdgcn_alignbyte(phi, pm, ps);
        const bool live = cmp && L < maxl;
        const uint32_t f1 = ((uint32_t)__ffs((int)x1) - 1u) >> 3, f2 = ((uint32_t)__ffs((int)x2) - 1u) >> 3;
        const uint32_t add = x1 ? f1 : (x2 ? 4u + f2 : 8u);
        L += live ? add : 0u;
        cmp = cmp && !(live && (x1 | x2));
        qi += 2;
        pi += 2;
        qlo = qhi;
        plo = phi;
      }
      {
        const bool c2 = mode == 2u;
        const uint32_t Lf = min(L, maxl);
        const bool better = c2 && Lf > best;
        best = better ? Lf : best;
        bestq = better ? q : bestq;
        if (__ballot(better)) {
          const uint32_t pw = m_ld32u(S.in, p + (best >= 3u ? best - 3u : 0u));
          pprobe = (better && best < maxl) ? pw : pprobe;
        }
        mode = c2 ? ((better && Lf >= ZES_MAXMATCH) ? 3u : 1u) : mode;
      }
    }
    if (!__ballot(mode != 0u)) {
      if (next >= r1) break;
      continue;  // nothing running: hand out more
    }
    // ---- probe the next candidate (the slot before, while it holds the same key); select code ----
    {
      const bool pr = mode == 1u;
      const bool stop0 = !curflag || check >= 128u || (best >= 8u && check >= 16u);  // src/lz77.ts:66-69
      const uint32_t rp2 = rp ? rp - 1u : MATCH_RING - 1u;
      const uint32_t e = ring[rp2];
      const uint32_t q2 = e & 0x7fffffffu;
      const bool far = (p - q2) > ZES_WINDOW;  // src/lz77.ts:49
      const bool adv = pr && !stop0;
      const bool cand = adv && !far;
      // L > best needs bytes 0..best equal: one dword probe at best-3 rejects most candidates
      const uint32_t pw = m_ld32u(S.in, q2 + (best >= 3u ? best - 3u : 0u));
      const bool skip = (best >= maxl) || (best >= 3u && pw != pprobe);
      rp = adv ? rp2 : rp;
      q = adv ? q2 : q;
      curflag = adv ? (e >> 31) : curflag;
      check += cand ? 1u : 0u;
      mode = (pr && (stop0 || far)) ? 3u : ((cand && !skip) ? 2u : mode);
    }
    if (mode == 3u) {
      if (best >= 3u && p + best + 3u <= T) {  // nowIndex + len <= endIndex = start + T - 3 (src/lz77.ts:95)
        mo[p] = ZES_TOK_MATCH | ((best - 3u) << 16) | (p - bestq - 1u);
        const uint32_t slot = atomicAdd(&S.nml, 1u);
        if (slot < ZES_MLIST_CAP) gl[1u + slot] = p | ((best - 3u) << 17);
      }
      mode = 0u;
    }
  }
  }
  __syncthreads();
  if (tid == 0) gl[0] = S.nml;  // (more than ZES_MLIST_CAP: the list is cut short, and k_lz_parse searches the cleared result words)
}

// ------------------------------------------------------------------------------------------
// k_lz_match_lazy: match finder for blocks that keep most of their positions (text, periodic data).
// The greedy parse only ever asks for the match at the positions of its chain p -> p + len | p + 1
// (about one position in five on text, one in a hundred on long periodic matches), but which positions
// those are is only known once the earlier ones are evaluated.  So:
//   0  the last LAZY_TAIL keyed positions of the block are evaluated up front, one lane each (a match that
//      would run into the block's last three bytes is dropped whole, src/lz77.ts:95: on repetitive data
//      every position there pays a full compare and stays a literal)
//   1  every 128-byte window gets the chain that starts at its first byte, evaluated position by
//      position by one lane (1024 chains in flight); the positions it stands on are kept as a bit mask
//      V1 and the position where it leaves the window as xw[w]
//   2  the chain that really enters a window starts wherever an earlier window's chain left off: from
//      each xw[u] a second chain is followed until it stands on a V1 position (greedy chains on text
//      merge within a few tokens).  By induction the true chain from position 0 is then evaluated
//      everywhere: window 0's own chain up to xw[0], the second chain from there up to a V1 position,
//      that window's own chain up to its xw, and so on.
//   3  only when a second chain gave up after LAZY_MERGE_CAP bytes without meeting a V1 position
//      (periodic data: chains of maximal matches never merge): one wavefront walks the true chain
//      from position 0 and evaluates what nobody has evaluated, the 64 lanes sharing each compare.
// Candidates come from k_lz_sort's tables: inv[p] gives the sorted slot r of p and the distance to its
// nearest candidate; the ones after it follow by subtracting sd[r-1], sd[r-2], ... (0 ends the list) —
// consecutive 2-byte entries, fetched four at a time (one 8-byte load per four candidates, 32 candidates
// per cache line) one group ahead of their use.
// Afterwards every position of the true chain carries its match word (a match, or LAZY_EVAL_LIT);
// the others are zero or whatever a side chain left there, which k_lz_parse never looks at on the
// chain it resolves.
// ------------------------------------------------------------------------------------------
__device__ unsigned long long* g_lazy_dbg = nullptr;  // ZES_DEBUG_PHASES: cycle stamps [g][8]
void zes_lazy_set_dbg(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lazy_dbg), &p, sizeof p); }
#define LSTAMP(i)                                                                                        \
  do {                                                                                                   \
    if (g_lazy_dbg && threadIdx.x == 0) g_lazy_dbg[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); \
  } while (0)
#ifndef LAZY_WIN
#define LAZY_WIN 512u
#endif
#define LAZY_NWIN (ZES_BLK / LAZY_WIN)
#ifndef LAZY_TAIL
#define LAZY_TAIL 272u         // positions at the block end that are evaluated up front, one row each: every position whose match can reach the last three bytes (258 + 3), rounded up
#endif
#define LAZY_EVAL_LIT 1u       // match word of an evaluated position that stays a literal (no match bit: k_lz_parse reads a literal)
#define LAZY_MERGE_CAP 1024u   // a second chain that has met no window chain after this many bytes gives up
#define LAZY_IN_BYTES (MATCH_IN_DWORDS * 4u)
struct LazySmem {
  uint8_t in[LAZY_IN_BYTES];                      // block + halo, zero padded, plain byte order: the compares read 8 bytes at any byte offset (ds_read_b64)
  uint32_t v1[ZES_BLK / 32];                      // positions on the window-start chains
  uint16_t xw[LAZY_NWIN];                         // chain of window w leaves it at 128 w + xw[w]
  uint16_t tail[LAZY_TAIL];                       // hop (1 or match length) of the block's last LAZY_TAIL keyed positions
  uint32_t wq;                                    // next work item to hand out
  uint32_t unmerged;                              // a second chain gave up: the true chain may hold unevaluated positions
  uint32_t ngave;                                 // how many gave up (the probe of a block that may be periodic counts them)
  uint32_t abort3;                                // phase 3 ran out of its budget of single evaluations: the block is not periodic after all
  uint32_t mp[LAZY_NWIN];                         // where the second chain from window w's exit met a window chain (LAZY_NOMERGE: it ran to the block's end)
  uint32_t tfrom[LAZY_NWIN];                      // true chain: the first position of window w's own chain that is on it (LAZY_NOMERGE: none)
  uint8_t titem[LAZY_NWIN];                       // true chain: it leaves window w at that window's exit
  uint16_t tj[LAZY_NWIN];                         // true chain: the window reached from w by 2^round links
  uint8_t tmark[LAZY_NWIN];                       // true chain: window w is on it
};
#define LAZY_NOMERGE 0xFFFFFFFFu
__device__ __forceinline__ static uint64_t lz_ld64(const uint8_t* s, uint32_t off) {  // 8 bytes at any byte offset
  uint64_t v;
  __builtin_memcpy(&v, s + off, 8);
  return v;
}
__device__ __forceinline__ static uint32_t lz_ld32(const uint8_t* s, uint32_t off) {
  uint32_t v;
  __builtin_memcpy(&v, s + off, 4);
  return v;
}

// Match at position p by the whole wavefront (p uniform): the candidates one after the other, every compare
// shared by the 64 lanes (four bytes each).  Candidate order, early exits and tie rule of src/lz77.ts:49-95.
__device__ static uint32_t lazy_wave_eval(const LazySmem& S, uint32_t iv, const uint16_t* __restrict__ sd, uint32_t p, uint32_t T, uint32_t avail) {
  const uint32_t lane = zes_lane();
  const uint32_t maxl = min(ZES_MAXMATCH, avail - p);
  uint32_t best = 0, bestq = 0, check = 0, q = p;
  if (iv == ZES_INV_NONE) return LAZY_EVAL_LIT;
  uint32_t s = iv & 0x1FFFFu;  // slot whose sd entry leads from the current candidate to the next
  uint32_t dq = 0, dqv = sd[s];  // (the entry of p's own slot: the distance to its nearest candidate)
  bool pending = true;  // dqv (requested while the candidate before was compared) still has to be read
  const uint32_t pd = lz_ld32(S.in, p + 4u * lane), pt = lz_ld32(S.in, p + 256u);
  for (;;) {
    if (check >= 128u || (best >= 8u && check >= 16u) || best >= maxl) break;  // :66-69, :89-91 (before the wait: a full match needs no further distance)
    if (pending) dq = (uint32_t)__builtin_amdgcn_readfirstlane((int)dqv);
    if (dq == 0u) break;
    const uint32_t q2 = q - dq;
    if (p - q2 > ZES_WINDOW) break;  // :49
    q = q2;
    check++;
    s--;
    dqv = sd[s];  // (slot 0 has nobody in front of it: sd[0] is 0)
    pending = true;
    const uint32_t x = lz_ld32(S.in, q + 4u * lane) ^ pd;
    const uint64_t mism = __ballot(x != 0u);
    uint32_t L;
    if (mism) {
      const uint32_t fl = (uint32_t)__builtin_ctzll(mism);
      const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)fl);
      L = 4u * fl + ((uint32_t)__builtin_ctz(xf) >> 3);
    } else {
      const uint32_t t = (lz_ld32(S.in, q + 256u) ^ pt) & 0xffffu;  // bytes 256, 257
      L = t ? 256u + ((uint32_t)__builtin_ctz(t) >> 3) : ZES_MAXMATCH;
    }
    L = min(L, maxl);
    if (L > best) {  // strictly longer only: the nearest candidate wins ties (:86-88)
      best = L;
      bestq = q;
    }
  }
  if (best >= 3u && p + best + 3u <= T) return ZES_TOK_MATCH | ((best - 3u) << 16) | (p - bestq - 1u);  // :95
  return LAZY_EVAL_LIT;
}

#define LZ_IDLE 0u   // no chain
#define LZ_PROBE 1u  // looking at the candidates of p
#define LZ_DONE 2u   // the match at p is known
#define LZ_STAND 3u  // the chain stands at p: leave, merge, hop or evaluate
#define LZ_WAIT 4u   // (chain-per-lane form) a position has started: its inv entry is on the way
#define LZ_EVAL 5u   // the row is on a position: rounds of sixteen candidates

// Phases 0-2: one chain per group of 16 lanes (a DPP row), four chains per wavefront, 64 per workgroup.
// A turn of the loop evaluates one position of every chain: its sixteen nearest candidates at once, lane k the
// k-th — their distances are sixteen consecutive 2-byte entries of sd[] (one 32-byte access), the positions a
// prefix sum over the row, every lane's common prefix with p computed side by side, and the reference's
// one-after-the-other rule (src/lz77.ts:65-93) read off a prefix maximum: on text a position has 19 candidates on
// average and 17 of them cannot beat what is already in hand, which a chain-per-lane loop (one candidate per turn)
// finds out one turn at a time.  A position with more than sixteen candidates takes further rounds of sixteen while
// the rule allows (less than 8 bytes in hand after 16 candidates, 128 at most: src/lz77.ts:66-69).
// Control is the same for the sixteen lanes of a row (every lane keeps a copy of the chain's state), so what
// diverges inside a wavefront are only the four chains.
#define LAZY_G 16u
template <int N>
__device__ __forceinline__ static uint32_t row_shr(uint32_t x) {  // lane i of a row gets lane i-N's value, 0 for the first N lanes
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + N, 0xf, 0xf, false);
}

// maximum / minimum over a row of 16 lanes, in every lane of the row: four DPP rotations on the vector unit instead of a lane
// lookup through the LDS crossbar (ds_bpermute: a round trip of ~150 cycles with sixteen wavefronts at it, and the turn had
// eight).  (`old` is the operation's identity, so that the compiler folds the rotation into the max / min instruction itself;
// every lane of a rotation has a source, the value is never used.)
template <int N>
__device__ __forceinline__ static uint32_t row_ror0(uint32_t x) {  // lane i of a row gets lane (i + N) mod 16's value
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x120 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ static uint32_t row_ror1(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x120 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ static uint32_t row_lane(uint32_t x) {  // lane N of the row, in every lane of the row (row_newbcast: gfx90a and later)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + N, 0xf, 0xf, false);
}
__device__ __forceinline__ static uint32_t row_allmax(uint32_t x) {
  x = max(x, row_ror0<8>(x));
  x = max(x, row_ror0<4>(x));
  x = max(x, row_ror0<2>(x));
  x = max(x, row_ror0<1>(x));
  return x;
}
__device__ __forceinline__ static uint32_t row_allmin(uint32_t x) {
  x = min(x, row_ror1<8>(x));
  x = min(x, row_ror1<4>(x));
  x = min(x, row_ror1<2>(x));
  x = min(x, row_ror1<1>(x));
  return x;
}

// The loop's requests to memory.  Where loads and stores are both in flight the compiler waits for a load with
// s_waitcnt vmcnt(0) — they share the counter on gfx9 and it keeps no order between the two kinds — i.e. at EVERY use in
// this loop: each turn sat out the full latency of the request it had made a moment ago, twice (in-kernel laps of round 3:
// 3.4k cycles a turn for ~450 instructions).  So the three requests a turn makes for later turns are issued by hand,
// unconditionally (address 0 when no row asks) and always in the same order — W: a row's next window of inv entries,
// D: the next round's sixteen distances, S: the first sixteen distances of the position a chain will stand on next — and
// are waited for by count: vector memory operations complete in issue order, so "all but the N youngest are done"
// covers a request once N younger ones have been issued; whatever else is issued in between only makes the wait longer,
// never shorter.  The registers the requests land in are touched by nothing but the wait that covers them.
// The requests land in three ACCUMULATION registers (a0: W, a1: D, a2: S — gfx90a and later load straight into them), which
// this kernel has no other use for and the compiler none at all (it takes them for spills and matrix instructions only):
// nothing can copy, reuse or reschedule a register whose data is still on its way — with ordinary registers as
// destinations the register allocator did exactly that (a copy of the pending register at the loop's back edge).  The wait
// that covers a request also moves its data into a vector register.  tools/check_lazy_isa.py looks at the generated code:
// the three registers may appear in these statements only.
#define LZ_REQ(areg, op, base, byteoff) asm volatile(op " " areg ", %0, %1" ::"v"(byteoff), "s"(base) : "memory", areg)

// Which windows a launch of the chains takes (work item k -> window):
#define LZ_MAP_ALL 0u     // every window
#define LZ_MAP_PROBE 1u   // three windows in sixteen (16 g, 16 g + 1, 16 g + 2): the probe of a block that may be periodic
#define LZ_MAP_REST 2u    // the other thirteen
#define LZ_MAP_PROBE2 3u  // the second chains of the probe: from the exits of windows 16 g (they meet V1 inside 16 g + 1, 16 g + 2 or give up)
#define LAZY_PROBE_STRIDE 16u
#define LAZY_PROBE_SPAN 3u
__device__ __forceinline__ static uint32_t lz_map_window(uint32_t map, uint32_t k) {
  if (map == LZ_MAP_PROBE) return (k / LAZY_PROBE_SPAN) * LAZY_PROBE_STRIDE + k % LAZY_PROBE_SPAN;
  if (map == LZ_MAP_REST) return (k / (LAZY_PROBE_STRIDE - LAZY_PROBE_SPAN)) * LAZY_PROBE_STRIDE + LAZY_PROBE_SPAN + k % (LAZY_PROBE_STRIDE - LAZY_PROBE_SPAN);
  if (map == LZ_MAP_PROBE2) return k * LAZY_PROBE_STRIDE;
  return k;
}
__host__ __device__ static inline uint32_t lz_map_count(uint32_t map, uint32_t nwin) {  // (the items are numbered so that k < count names windows below nwin only)
  const uint32_t full = nwin / LAZY_PROBE_STRIDE, part = nwin % LAZY_PROBE_STRIDE;
  if (map == LZ_MAP_PROBE) return full * LAZY_PROBE_SPAN + (part < LAZY_PROBE_SPAN ? part : LAZY_PROBE_SPAN);
  if (map == LZ_MAP_REST) return full * (LAZY_PROBE_STRIDE - LAZY_PROBE_SPAN) + (part > LAZY_PROBE_SPAN ? part - LAZY_PROBE_SPAN : 0u);
  if (map == LZ_MAP_PROBE2) return full + (part ? 1u : 0u);
  return nwin;
}

// GUARDED: every section of a turn sits behind "does any row need it?" — for blocks whose turns mostly need few of them
// (periodic data: runs of literals, one position in 258 behind them); text needs all of them nearly every turn, and a guard
// costs two vector instructions and a branch.
template <uint32_t PHASE, bool GUARDED>
__device__ __forceinline__ static uint32_t lazy_chains(LazySmem& S, const uint32_t* __restrict__ inv, const uint16_t* __restrict__ sd,
                                                       uint32_t* __restrict__ mo, uint32_t T, uint32_t cnt, uint32_t avail, uint32_t tbase,
                                                       uint32_t nitems, uint32_t map = LZ_MAP_ALL) {
  const uint32_t lane = zes_lane(), sub = lane & (LAZY_G - 1u), g0 = lane & ~(LAZY_G - 1u);  // g0: first lane of this row
  uint32_t mode = LZ_IDLE, item = 0, p = 0, wend = 0, cstart = 0;  // the same in the 16 lanes of a row
  uint32_t wcur = 0, wnxt = 0, wpend = 0, wb = 0x40000000u;  // inv entries of positions [wb, wb+16) and [wb+16, wb+32), lane k the k-th
  bool fw = false;                                           // wpend holds this row's next window
  uint32_t sdp = 0, sdp_r = 0;                               // first sixteen distances requested ahead for slot sdp_r
  bool sdp_ok = false;
  // Requests made for a LATER turn stay in registers of their own (spend: the first sixteen distances of the chain's
  // next position; dpend: the next round's sixteen) and are folded into sdp / dnext where that turn first needs them.
  uint32_t spend = 0, dpend = 0;
  bool sfresh = false, dfresh = false;
  // the position a row is on (rounds of sixteen candidates; a row whose position needs another round does not hold up
  // the other three: every turn of the loop is one round for each row that has one to do)
  uint32_t r = 0, maxl = 0, best = 0, bestq = 0, base = 0, lastq = 0, dnext = 0, litrun = 1;
  bool more = false;
  bool drained = false;  // no work items left (uniform)
  uint32_t niter = 0;
#ifdef LAZY_PROF
  unsigned long long lap[6] = {0, 0, 0, 0, 0, 0}, tl = clock64(), nlcp = 0, nrnd = 0;
#define LLAP(i)                              \
  do {                                       \
    const unsigned long long n_ = clock64(); \
    lap[i] += n_ - tl;                       \
    tl = n_;                                 \
  } while (0)
#else
#define LLAP(i)
#endif
  for (;;) {
    niter++;
    // W of the turn before has landed: behind it that turn issued its D and its S
    asm volatile("s_waitcnt vmcnt(2)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(wpend)::"memory");
    wnxt = fw ? wpend : wnxt;
    fw = false;
    // ---- hand out work items to rows without a chain ----
    {
      const uint64_t idle = __ballot(mode == LZ_IDLE) & 0x0001000100010001ull;  // one bit per row
      if (idle && !drained) {
        uint32_t firsti = 0;
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (lane == 0) firsti = atomicAdd(&S.wq, nidle);
        firsti = (uint32_t)__builtin_amdgcn_readfirstlane((int)firsti);
        if (firsti >= nitems) drained = true;
        const uint32_t mine = firsti + (uint32_t)__popcll(idle & ((1ull << g0) - 1ull));
        const bool take = mode == LZ_IDLE && mine < nitems;
        const uint32_t wid = PHASE == 0u ? mine : lz_map_window(map, mine);  // the window (phase 0: the position) the item stands for
        item = take ? wid : item;
        uint32_t np, nw;
        if (PHASE == 0u) {
          np = tbase + mine;
          nw = np + 1u;
        } else if (PHASE == 1u) {
          np = wid * LAZY_WIN;
          nw = min(np + LAZY_WIN, T);
        } else {
          np = wid * LAZY_WIN + S.xw[take ? wid : 0u];
          nw = T;  // (np >= T: the chain ended with the block)
        }
        p = take ? np : p;
        cstart = take ? np : cstart;
        wend = take ? nw : wend;
        mode = take ? LZ_STAND : mode;
      }
    }
    if (!__ballot(mode != LZ_IDLE) && drained) break;
    // ---- chain control: leave the window, merge, step over pre-evaluated or keyless positions, or evaluate ----
    bool ev;
    {
      const bool st = mode == LZ_STAND;
      const bool leave = st && p >= wend;
      bool merged = false, gaveup = false;
      if (PHASE == 2u) {
        const uint32_t pc = min(p, ZES_BLK - 1u);
        const uint32_t vb = (S.v1[pc >> 5] >> (pc & 31u)) & 1u;
        merged = st && !leave && vb != 0u;  // from here on it is a window's own chain
        gaveup = st && !leave && !merged && p - cstart >= LAZY_MERGE_CAP;
        if (gaveup && sub == 0u) {
          S.unmerged = 1u;
          atomicAdd(&S.ngave, 1u);
        }
        if (merged && sub == 0u) S.mp[item] = p;
        if (leave && sub == 0u) S.mp[item] = LAZY_NOMERGE;
      }
      const bool go = st && !leave && !merged && !gaveup;
      if (PHASE == 1u) {
        if (go && sub == 0u) atomicOr(&S.v1[p >> 5], 1u << (p & 31u));
        if (leave && sub == 0u) S.xw[item] = (uint16_t)(p - item * LAZY_WIN);
      }
      const bool keyless = go && p >= cnt;  // the block's last two bytes are always literals (src/lz77.ts:116-117)
      bool intail = false;
      uint32_t hop = 0;
      if (PHASE != 0u) {
        intail = go && !keyless && p >= tbase;  // evaluated in phase 0
        if (__ballot(p >= tbase)) hop = S.tail[intail ? p - tbase : 0u];  // (the block's end only: not an LDS round trip per turn)
      }
      ev = go && !keyless && !intail;
      p += keyless ? 1u : (intail ? hop : 0u);
      mode = (leave || merged || gaveup) ? LZ_IDLE : mode;
    }
    LLAP(0);
    // ---- a position starts (src/lz77.ts:49-95): its sorted slot, the first sixteen distances ----
    bool fin = false;  // the match at p is known
    // (the sections below run in every turn: with four rows a wavefront there is nearly always one that needs them, and every
    // guard "does any row?" costs two vector instructions and a branch)
    uint32_t wreq = 0, iv = ZES_INV_NONE;
    const bool anyev = !GUARDED || __ballot(ev) != 0ull;
    if (anyev) {
      // sorted slot of p from the row's two windows of inv entries (sixteen positions each, lane k the k-th); a chain
      // moves on by a few bytes per position, so the window behind the current one was requested turns ago
      if (__ballot(ev && p - wb >= 2u * LAZY_G)) {  // far jump (or a new work item): both windows from memory
        const bool far = ev && p - wb >= 2u * LAZY_G;
        const uint32_t nb = p & ~(LAZY_G - 1u);
        const uint32_t a0 = inv[far ? min(nb + sub, ZES_BLK - 1u) : 0u], a1 = inv[far ? min(nb + LAZY_G + sub, ZES_BLK - 1u) : 0u];
        wcur = far ? a0 : wcur;
        wnxt = far ? a1 : wnxt;
        wb = far ? nb : wb;
      }
      {
        const bool step = ev && p - wb >= LAZY_G;  // into the window behind: it becomes the current one, the next is requested
        wcur = step ? wnxt : wcur;
        wb += step ? LAZY_G : 0u;
        wreq = step ? min(wb + LAZY_G + sub, ZES_BLK - 1u) : 0u;
        fw = step;  // read at the top of the next turn (this row's request)
      }
      iv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * (g0 + ((p - wb) & (LAZY_G - 1u)))), (int)wcur);  // (a row that is not starting reads some lane of its own)
    }
    LZ_REQ("a0", "global_load_dword", inv, wreq << 2);  // W
    // S and D of the turn before have landed: behind them this turn's W has been issued
    asm volatile("s_waitcnt vmcnt(1)\n\tv_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a1" : "=v"(spend), "=v"(dpend)::"memory");
    LLAP(1);
    uint32_t sreq = 0;  // the slot whose first sixteen distances this row asks for with this turn's S (0: it does not ask)
    bool sask = false;
    if (anyev) {
      const bool has = ev && iv != ZES_INV_NONE;
      const uint32_t rr = iv & 0x1FFFFu;  // sorted slot of p: the distance to candidate k is the sum of sd[r], sd[r-1], ... sd[r-k]
      sdp = sfresh ? spend : sdp;  // what an earlier turn asked for
      sfresh = false;
      // the first sixteen distances were requested when the chain came to stand here — unless the position lay outside the
      // row's windows then, or the window was still on its way (long matches, new work items): those are fetched now
      const bool hit = has && sdp_ok && sdp_r == rr;
      if (__ballot(has && !hit)) {
        const uint32_t t = sd[(has && sub <= rr) ? rr - sub : 0u];
        sdp = (has && !hit) ? t : sdp;
      }
      r = ev ? rr : r;
      maxl = ev ? min(ZES_MAXMATCH, avail - p) : maxl;  // = min(258, n - p)
      best = ev ? 0u : best;
      bestq = ev ? 0u : bestq;
      base = ev ? 0u : base;
      lastq = ev ? p : lastq;
      more = has || (!ev && more);  // (has implies ev; bools are lane masks: logic on them is scalar work, a select is not)
      fin = ev && !has;  // no candidate: a literal
      mode = has ? LZ_EVAL : mode;
      dnext = has ? sdp : dnext;
      dfresh = dfresh && !ev;
      // A literal like that is often one of a run (incompressible stretches, the first period of periodic data): the
      // positions behind it whose inv entries, in the row's current window, say "no candidate" too are settled with it
      // — window chains only: a second chain has to test every position it stands on for a merge.
      litrun = 1u;
      if (PHASE == 1u) {
        const uint32_t off = (ev ? p - wb : 0u) & (LAZY_G - 1u);
        const uint32_t nm = (uint32_t)(__ballot(wcur == ZES_INV_NONE) >> g0) & 0xffffu;  // positions wb + k without a candidate
        const uint32_t inv_run = (uint32_t)__builtin_ctz(~(nm >> off) | 0x10000u);         // how many from p on (>= 1 when p has none)
        const uint32_t lim = min(min(wend, cnt), tbase) - p;                                // stay inside the window, the keyed and the not pre-evaluated positions
        litrun = fin ? max(1u, min(min(inv_run, LAZY_G - off), lim)) : 1u;
      }
    }
    // ---- one round of sixteen candidates for every row that is on a position ----
    const bool was = more;
    const uint32_t k = base + sub;
    uint32_t d = 0, dreq = 0;
    {
      const bool inl = more && k <= r;  // (slot 0 has nobody in front of it: sd[0] is 0)
      dnext = dfresh ? dpend : dnext;   // (a row that has just started a position took its sixteen from sdp: dfresh is off)
      d = inl ? dnext : 0u;
      const uint32_t k2 = k + LAZY_G;
      dreq = (more && k2 <= r) ? r - k2 : 0u;  // next round's sixteen, in flight until the next turn's round
    }
    LZ_REQ("a1", "global_load_ushort", sd, dreq << 1);  // D
    LLAP(2);
    // (periodic data has turn after turn without a round — runs of literals, one position in 258 after that — so this one
    // guard stays; a row has a round to do exactly when its mode says so, and a compare of that is a guard of one vector
    // instruction, where a ballot over the flag takes two)
    if (__ballot(mode == LZ_EVAL)) {
      // candidate positions: prefix sum of the distances over the row
      uint32_t ps = d;
      ps += row_shr<1>(ps);
      ps += row_shr<2>(ps);
      ps += row_shr<4>(ps);
      ps += row_shr<8>(ps);
      const uint32_t qk = lastq - ps;
      // the list ends at the first 0 (another key, or further than 32768 from its neighbour: src/lz77.ts:49 — the
      // window test on the candidate itself comes on top), and at 128 candidates (:66)
      const uint32_t zm = (uint32_t)(__ballot(d == 0u || (p - qk) > ZES_WINDOW || k >= 128u) >> g0) & 0xffffu;
      const uint32_t nv = zm ? (uint32_t)__builtin_ctz(zm) : LAZY_G;  // valid candidates of this round
      const bool v = more && sub < nv;
      // common prefix of the strings at qk and p (first three bytes equal by the key), capped at maxl
      uint32_t L = 3;
      LLAP(3);
      {
        uint32_t qo = (v ? qk : 0u) + 3u, po = p + 3u;
        bool run = v;
        auto step = [&]() {  // 8 bytes: one unaligned read each side
          const uint64_t x = lz_ld64(S.in, qo) ^ lz_ld64(S.in, po);
          const bool live = run && L < maxl;
          const uint32_t add = x ? ((uint32_t)__builtin_ctzll(x) >> 3) : 8u;
          L += live ? add : 0u;
          run = run && !(live && x != 0ull);
          qo += 8u;
          po += 8u;
#ifdef LAZY_PROF
          nlcp++;
#endif
        };
        {
          // The first sixteen bytes of both sides in ONE round trip to the LDS (text settles 99.5 % of its compares in the first
          // eight, but with ~48 candidates a wavefront one of them nearly always needs the second eight).  Equal bytes from the
          // front: per dword of the difference its low zero bytes (a dword without a difference: a huge number), the first
          // dword that has one decides.  The cap at maxl comes behind the compare, so nothing has to stop adding on the way.
          const uint64_t q0 = lz_ld64(S.in, qo), q1 = lz_ld64(S.in, qo + 8u), p0 = lz_ld64(S.in, po), p1 = lz_ld64(S.in, po + 8u);
          const uint64_t x0 = q0 ^ p0, x1 = q1 ^ p1;
          auto zb = [](uint32_t dw) {  // (v_ffbl_b32 of 0 is 0xFFFFFFFF; __ffs / __builtin_ctz make the compiler add a compare and a select)
            uint32_t f;
            asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(dw));
            return f >> 3;
          };
          const uint32_t n16 = min(min(min(zb((uint32_t)x0), zb((uint32_t)(x0 >> 32)) + 4u), zb((uint32_t)x1) + 8u), min(zb((uint32_t)(x1 >> 32)) + 12u, 16u));
          L = 3u + n16;
          run = v && n16 == 16u;
          qo += 16u;
          po += 16u;
#ifdef LAZY_PROF
          nlcp += 2;
#endif
        }
        if (__ballot(run && L < maxl)) {
          // (text hardly ever gets here: two steps settle nearly all of its compares)
          // The nearest candidate of a position's first round still going after sixteen bytes, with the full compare length ahead
          // of it (periodic data: every candidate matches to the end, 32 steps of 8 bytes for sixteen of them): it is tested
          // WHOLE by the row, sixteen bytes a lane.  If it matches at full length the reference takes it and stops looking
          // (src/lz77.ts:86-91: the first candidate examined, and no longer one exists) — whatever the others are.
          const bool wide = ((uint32_t)(__ballot(run && L < maxl && sub == 0u && base == 0u && maxl == ZES_MAXMATCH) >> g0) & 1u) != 0u;  // (the same in a row's lanes)
          if (__ballot(wide)) {
            const uint32_t qn = (uint32_t)__shfl((int)qk, (int)g0);
            const uint32_t qa = (wide ? qn : 0u) + 16u * sub, pa = (wide ? p : 0u) + 16u * sub;
            uint32_t x = 0;
#pragma unroll
            for (uint32_t t = 0; t < 4; t++) x |= lz_ld32(S.in, qa + 4u * t) ^ lz_ld32(S.in, pa + 4u * t);
            if (sub == 0u) x |= (lz_ld32(S.in, qa + 256u) ^ lz_ld32(S.in, pa + 256u)) & 0xffffu;  // bytes 256, 257
            const bool full = wide && ((uint32_t)(__ballot(wide && x != 0u) >> g0) & 0xffffu) == 0u;
            L = (full && sub == 0u) ? ZES_MAXMATCH : L;
            run = run && !full;
          }
          while (__ballot(run && L < maxl)) step();
        }
      }
      LLAP(4);
      L = v ? min(L, maxl) : 0u;
      // what is in hand before candidate k is looked at: the running maximum over the nearer ones
      uint32_t incl = L;
      incl = max(incl, row_shr<1>(incl));
      incl = max(incl, row_shr<2>(incl));
      incl = max(incl, row_shr<4>(incl));
      incl = max(incl, row_shr<8>(incl));
      const uint32_t excl = max(row_shr<1>(incl), best);
      // candidate k is looked at unless 8 bytes are in hand after 16 candidates (:66-69) or a full-length match has
      // ended the scan (:89-91); both tests only ever cut off a tail of the row
      const bool ex = v && (base == 0u || excl < 8u) && excl < maxl;
      const bool allex = row_lane<15>(ex ? 1u : 0u) != 0u;            // all sixteen were there and were looked at (ex is a prefix of the row)
      const uint32_t gmax = row_allmax(ex ? L : 0u);                   // maximum over the candidates looked at
      if (!GUARDED || __ballot(more && gmax > best)) {
        // strictly longer only, and the nearest of the longest: the nearest candidate wins ties (:86-88) — the nearest is the
        // one at the highest position
        const uint32_t qb = row_allmax((ex && L == gmax) ? qk + 1u : 0u) - 1u;
        const bool up = more && gmax > best;
        bestq = up ? qb : bestq;
        best = up ? gmax : best;
      }
      // another round: all sixteen were there and were looked at, fewer than 128 so far, and the rule lets candidate
      // 16 (32, ...) be looked at: less than 8 bytes in hand, no full-length match
      const uint32_t lq = row_lane<15>(qk);  // the sixteenth candidate (only a row that had all sixteen goes on from it)
      lastq = more ? lq : lastq;
      base += more ? LAZY_G : 0u;
      more = more && allex && base < 128u && best < 8u && best < maxl;
      dfresh = more;
      fin = fin || (was && !more);
#ifdef LAZY_PROF
      nrnd++;
#endif
    }
    // ---- a finished evaluation moves its chain on ----
    if (!GUARDED || __ballot(fin)) {
      const bool acc = best >= 3u && p + best + 3u <= T;  // src/lz77.ts:95
      if (fin && sub == 0u) mo[p] = acc ? (ZES_TOK_MATCH | ((best - 3u) << 16) | (p - bestq - 1u)) : LAZY_EVAL_LIT;
      if (PHASE == 0u) {
        if (fin && sub == 0u) S.tail[item] = (uint16_t)(acc ? best : 1u);
      }
      if (PHASE == 1u) {  // a run of literals: their words, and their bits in V1 (the first one's is set already)
        const bool rl = fin && !acc && litrun > 1u;
        if (rl && sub >= 1u && sub < litrun) {
          mo[p + sub] = LAZY_EVAL_LIT;
          atomicOr(&S.v1[(p + sub) >> 5], 1u << ((p + sub) & 31u));
        }
      }
      p += fin ? (acc ? best : litrun) : 0u;
      mode = fin ? LZ_STAND : mode;
      // the first sixteen distances of the position the chain stands on now, for the next turn (when that position
      // is still inside the two windows of inv entries, and if chain control does not move the chain elsewhere)
      {
        const uint32_t off = p - wb;
        const bool in2 = fin && off < 2u * LAZY_G && p < cnt;
        const int from = (int)(4u * (g0 + (off & (LAZY_G - 1u))));
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)wcur);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)wnxt);
        const uint32_t ivn = off < LAZY_G ? e0 : e1;
        const bool want = in2 && ivn != ZES_INV_NONE && !(fw && off >= LAZY_G);  // (a window still on its way: not this time)
        const uint32_t rn = ivn & 0x1FFFFu;
        sask = want || (!fin && sask);  // (want implies fin)
        sreq = fin ? (want ? rn : 0u) : sreq;
        sdp_ok = want || (!fin && sdp_ok);
        sdp_r = fin ? rn : sdp_r;
      }
    }
    // S: the new request overwrites spend in every lane — a row's landed one is folded in first (a row whose request was
    // never folded has moved on without evaluating: that request is dead)
    sdp = sfresh ? spend : sdp;
    sfresh = sask;
    LZ_REQ("a2", "global_load_ushort", sd, ((sask && sub <= sreq) ? sreq - sub : 0u) << 1);
    LLAP(5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last requests have landed before the next phase issues its own)
#ifdef LAZY_PROF
  if (PHASE == 1u && blockIdx.x == 7 && threadIdx.x == 0)
    printf("lazy prof: turns %u rounds %llu | control %llu inv-wait %llu sd-wait %llu prefix %llu lcp %llu (%llu steps) rule+store %llu\n", niter, nrnd,
           lap[0], lap[1], lap[2], lap[3], lap[4], nlcp, lap[5]);
#endif
#undef LLAP
  return niter;
}

__global__ __launch_bounds__(MATCH_THREADS) void k_lz_match_lazy(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                                 const ZesBlk* __restrict__ blks, const uint32_t* __restrict__ idx_a,
                                                                 const uint32_t* __restrict__ inv_all, const uint16_t* __restrict__ sd_all,
                                                                 uint32_t* __restrict__ match_out, uint32_t* __restrict__ tmask_all,
                                                                 uint32_t* __restrict__ mlist_all, const uint32_t* __restrict__ order) {
  __shared__ __align__(16) LazySmem S;
  // (a batch of unlike buffers: the blocks in the order k_lz_order dealt them, heaviest first)
  const uint32_t g = order ? order[blockIdx.x] : blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t flagword = idx_a[(uint64_t)g * ZES_BLK + ZES_BLK - 1];
  if (!(flagword & ZES_SORT_LAZY)) return;  // this block belongs to k_lz_match
  if (threadIdx.x == 0) mlist_all[(uint64_t)g * ZES_MLIST_WORDS] = 0xFFFFFFFFu;  // "no list: a block of this kernel" (k_lz_parse)
  const uint32_t* inv = inv_all + (uint64_t)g * ZES_BLK;
  const uint16_t* sd = sd_all + (uint64_t)g * ZES_BLK;
  const ZesBlk bk = blks[g];
  const ZesBuf bf = bufs[bk.buf];
  const uint32_t T = bk.len;
  const uint64_t S0 = (uint64_t)bk.blk * ZES_BLK;  // block start inside the buffer
  const uint8_t* src = d_in + bf.in_off + S0;
  const uint64_t remain = bf.n_read - S0;  // bytes from block start to input end
  const uint32_t avail = (uint32_t)(remain < (uint64_t)(T + ZES_MAXMATCH) ? remain : (uint64_t)(T + ZES_MAXMATCH));
  uint32_t* mo = match_out + (uint64_t)g * ZES_BLK;
  const uint32_t cnt = T >= 3 ? T - 2 : 0;  // positions that have a 3-byte key
  const uint32_t nwin = (T + LAZY_WIN - 1) / LAZY_WIN;

  LSTAMP(0);
  // stage block + halo (zero padded), plain byte order
  if ((((uintptr_t)src) & 15u) == 0) {
    const uint4* g4 = reinterpret_cast<const uint4*>(src);
    uint4* s4 = reinterpret_cast<uint4*>(S.in);
    for (uint32_t i = tid; i < LAZY_IN_BYTES / 16; i += MATCH_THREADS) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const uint32_t b = i * 16u;
      if (b + 16u <= avail) {
        v = g4[i];
      } else if (b < avail) {
        uint32_t t[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < 16u && b + k < avail; k++) t[k >> 2] |= (uint32_t)src[b + k] << (8u * (k & 3u));
        v = make_uint4(t[0], t[1], t[2], t[3]);
      }
      s4[i] = v;
    }
  } else {
    uint32_t* s1 = reinterpret_cast<uint32_t*>(S.in);
    for (uint32_t i = tid; i < LAZY_IN_BYTES / 4; i += MATCH_THREADS) {
      uint32_t t = 0;
      for (uint32_t k = 0; k < 4u; k++)
        if (4 * i + k < avail) t |= (uint32_t)src[4 * i + k] << (8u * k);
      s1[i] = t;
    }
  }
  // The result words are only ever read where this kernel has written them (the parse reads the chain's positions, the
  // second chains' re-walk the positions those chains evaluated) — except by phase 3's walk, which asks "has anybody
  // evaluated this position?" of words nobody may have touched.  Phase 3 is what periodic data needs (chains of maximal
  // matches never merge), and periodic data is what k_lz_index takes: those blocks clear their 512 KiB of words up front.
  // Any other block (text: 268 MB of stores per 64 MiB of input until round 3) clears them only if a second chain does give
  // up, and then evaluates once more.
  bool cleared = (flagword & ZES_SORT_INDEX) != 0u;  // (uniform)
  const bool guarded = cleared;  // the same kind of block: its turns are mostly runs of literals (see lazy_chains)
  // Periodic data — what k_lz_index's blocks mostly are — gets nothing from its window chains: the true chain stands on
  // p0 + 258 k, the windows start on multiples of 512, the second chains give up, and phase 3 evaluates the whole true
  // chain itself, 64 maximal matches at a time (lowent4k: window and second chains 400k of the kernel's 770k cycles per
  // block).  So such a block is PROBED first: three windows in sixteen get their chains, the second chains from every
  // sixteenth window's exit look for them; when three quarters of those give up the block goes straight to phase 3, with a
  // budget of single evaluations (a block that only looked periodic runs out of it and gets all its chains after all);
  // otherwise the other thirteen windows in sixteen get theirs and every second chain runs, as for any block.
  bool probe = guarded && nwin >= 4u * LAZY_PROBE_STRIDE;
  const uint32_t tbase = cnt > LAZY_TAIL ? cnt - LAZY_TAIL : 0u;  // first pre-evaluated position
  uint32_t* tm = tmask_all + (uint64_t)g * ZES_TMASK_WORDS;
  for (;;) {
    if (cleared) {
      uint4* mo4 = reinterpret_cast<uint4*>(mo);
      const uint32_t n4 = (T + 3u) >> 2;
      for (uint32_t i = tid; i < n4; i += MATCH_THREADS) mo4[i] = make_uint4(0, 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zeros are in memory before another wave writes a result word
    } else if (tid < 2u && cnt + tid < T) {
      mo[cnt + tid] = LAZY_EVAL_LIT;  // the block's last two bytes: on every chain, evaluated by nobody (src/lz77.ts:116-117)
    }
    for (uint32_t i = tid; i < ZES_BLK / 32; i += MATCH_THREADS) S.v1[i] = 0;
    if (tid == 0) {
      S.wq = 0;
      S.unmerged = 0;
      S.ngave = 0;
      S.abort3 = 0;
    }
    __syncthreads();

    LSTAMP(1);
    if (guarded) (void)lazy_chains<0u, true>(S, inv, sd, mo, T, cnt, avail, tbase, cnt - tbase);
    else (void)lazy_chains<0u, false>(S, inv, sd, mo, T, cnt, avail, tbase, cnt - tbase);
    __syncthreads();
    LSTAMP(2);
    uint32_t budget3 = 0xFFFFFFFFu;
    if (probe) {  // (blocks of k_lz_index: the guarded form)
      if (tid == 0) S.wq = 0;
      __syncthreads();
      (void)lazy_chains<1u, true>(S, inv, sd, mo, T, cnt, avail, tbase, lz_map_count(LZ_MAP_PROBE, nwin), LZ_MAP_PROBE);
      __syncthreads();
      if (tid == 0) S.wq = 0;
      __syncthreads();
      const uint32_t n2 = lz_map_count(LZ_MAP_PROBE2, nwin);
      (void)lazy_chains<2u, true>(S, inv, sd, mo, T, cnt, avail, tbase, n2, LZ_MAP_PROBE2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool periodic = S.ngave * 4u >= n2 * 3u;  // (uniform)
      __syncthreads();
      if (periodic) {
        budget3 = 64u + T / 1024u;  // (periodic data: one single evaluation per 64 maximal matches, ~10 a block)
        if (tid == 0) S.unmerged = 1u;
      } else {
        if (tid == 0) {
          S.wq = 0;
          S.unmerged = 0;
        }
        __syncthreads();
        (void)lazy_chains<1u, true>(S, inv, sd, mo, T, cnt, avail, tbase, lz_map_count(LZ_MAP_REST, nwin), LZ_MAP_REST);
        __syncthreads();
        if (tid == 0) S.wq = 0;
        __syncthreads();
        (void)lazy_chains<2u, true>(S, inv, sd, mo, T, cnt, avail, tbase, nwin);  // every second chain (the probe's once more: their windows are all there now)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      LSTAMP(3);
      LSTAMP(4);
    } else {
      if (tid == 0) S.wq = 0;
      __syncthreads();
      const uint32_t niter = guarded ? lazy_chains<1u, true>(S, inv, sd, mo, T, cnt, avail, tbase, nwin) : lazy_chains<1u, false>(S, inv, sd, mo, T, cnt, avail, tbase, nwin);
      if (g_lazy_dbg && tid == 0) g_lazy_dbg[(size_t)blockIdx.x * 8 + 6] = niter;
      __syncthreads();
      LSTAMP(3);
      if (tid == 0) S.wq = 0;
      __syncthreads();
      if (guarded) (void)lazy_chains<2u, true>(S, inv, sd, mo, T, cnt, avail, tbase, nwin);
      else (void)lazy_chains<2u, false>(S, inv, sd, mo, T, cnt, avail, tbase, nwin);
      // (result words another wave of this workgroup reads back — the second chains' re-walk, phase 3 — are in memory)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      LSTAMP(4);
    }
    if (S.unmerged && !cleared) {  // (uniform) phase 3 asks "has anybody evaluated this position?": the words are cleared first, and everything is evaluated once more
      cleared = true;
      __syncthreads();  // everybody has read the flag before it is reset
      continue;
    }
  // ---- the positions of the true chain as a bit mask, for k_lz_parse (which otherwise finds them again: exit maps
  // of all 2048 chunks, region tables, a walk per chunk — 343k of its 480k cycles per block on text).  The chain is
  // window 0's own chain up to its exit, the second chain from there up to where it met a window chain, that window's
  // chain from there to its exit, and so on: the windows' parts are in V1 already. ----
  if (!S.unmerged) {
    for (uint32_t i = tid; i < LAZY_NWIN; i += MATCH_THREADS) {
      S.tfrom[i] = LAZY_NOMERGE;
      S.titem[i] = 0;
    }
    __syncthreads();
    // Which windows the true chain passes through: window 0, then the window its second chain met (mp[0] / LAZY_WIN), and so
    // on — a linked list of up to 256 windows that one lane used to walk (three LDS round trips a link: 77k of this kernel's
    // 2.5M cycles per block of text).  Marked by pointer doubling instead: after round r every window less than 2^(r+1)
    // links from window 0 is marked, and only windows on the chain ever are.
    constexpr uint32_t TJ_END = 0xFFFFu;
    uint32_t mynext = TJ_END, myentry = 0;  // (threads below nwin: the link out of window tid, and where it lands)
    if (tid < LAZY_NWIN) {
      bool leaves = false;
      if (tid < nwin) {
        const uint32_t e = tid * LAZY_WIN + S.xw[tid];
        leaves = e < T;  // (else the chain ended with the block)
        const uint32_t m = S.mp[tid];
        if (leaves && m != LAZY_NOMERGE && m / LAZY_WIN > tid) {  // (else it ran to the block's end; a chain moves forward)
          mynext = m / LAZY_WIN;
          myentry = m;
        }
      }
      S.tj[tid] = (uint16_t)mynext;
      S.tmark[tid] = tid == 0u ? 1 : 0;
      S.titem[tid] = leaves ? 1 : 0;  // (kept only for marked windows, below)
    }
    __syncthreads();
#pragma unroll 1
    for (uint32_t r = 0; r < 8u; r++) {  // 2^8 = LAZY_NWIN
      uint32_t j = TJ_END, jj = TJ_END;
      bool mk = false;
      if (tid < LAZY_NWIN) {
        j = S.tj[tid];
        mk = S.tmark[tid] != 0;
        if (j != TJ_END) jj = S.tj[j];
      }
      __syncthreads();
      if (tid < LAZY_NWIN) {
        if (mk && j != TJ_END) S.tmark[j] = 1;
        S.tj[tid] = (uint16_t)jj;
      }
      __syncthreads();
    }
    static_assert(LAZY_NWIN <= 256u, "eight rounds of pointer doubling");
    if (tid < LAZY_NWIN) {
      const bool on = S.tmark[tid] != 0;
      if (!on) S.titem[tid] = 0;
      if (tid == 0u) S.tfrom[0] = 0;
      if (on && mynext != TJ_END) S.tfrom[mynext] = myentry;  // (a window on the chain has one predecessor on it)
    }
    __syncthreads();
    for (uint32_t i = tid; i < ZES_BLK / 32; i += MATCH_THREADS) {  // a window's chain in front of the meeting point, and windows the chain skips: not on it
      const uint32_t f = S.tfrom[(32u * i) / LAZY_WIN];
      uint32_t keep = 0u;
      if (f != LAZY_NOMERGE) keep = f <= 32u * i ? ~0u : (f >= 32u * i + 32u ? 0u : (~0u << (f - 32u * i)));
      S.v1[i] &= keep;
    }
    __syncthreads();
    if (tid < nwin && S.titem[tid]) {  // the second chains of the true chain: walked again, along the words they left
      uint32_t p = tid * LAZY_WIN + S.xw[tid];
      const uint32_t end = min(S.mp[tid], T);
      while (p < end) {
        atomicOr(&S.v1[p >> 5], 1u << (p & 31u));
        const uint32_t m = __hip_atomic_load(&mo[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (written by another wave of this workgroup)
        p += (m & ZES_TOK_MATCH) ? zes_tok_len(m) : 1u;
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < ZES_BLK / 32; i += MATCH_THREADS) tm[4u + i] = S.v1[i];
  }
  if (tid == 0) tm[0] = S.unmerged ? 0u : 1u;

  // ---- phase 3 (periodic data only): the true chain, by one wavefront (which also notes the chain's positions
  // for k_lz_parse: V1 starts over) ----
  const bool walk3 = S.unmerged != 0u;  // (uniform)
  if (walk3) {
    __syncthreads();
    for (uint32_t i = tid; i < ZES_BLK / 32; i += MATCH_THREADS) S.v1[i] = 0;
    __syncthreads();
  }
  if (walk3 && wave == 0) {
    uint32_t p = 0, nev = 0;  // nev: positions this walk had to evaluate itself, one at a time
    // The words of the chain's positions come 64 at a time (lane k: the word of position cb + k; the chunk behind is
    // requested while this one is walked); runs of evaluated literals are stepped over at once.
    // (Words written by the other waves of this workgroup: read past this CU's L1.)
    uint32_t cb = 0xFFFFFFFFu, cw = 0, nb = 0xFFFFFFFFu, nw = 0;  // current chunk, next sequential chunk
    uint32_t ciw = 0, niw = 0;  // ... and the inv entries of the same positions, in the same batch of loads (a walk through a stretch
                                // nobody has evaluated needs them at every stop: one memory latency per chunk instead of two per stop)
    while (p < cnt) {
      const uint32_t base = p & ~63u;
      if (base != cb) {
        const bool seq = base == nb;
        const uint32_t cw2 = seq ? nw : __hip_atomic_load(&mo[min(base + lane, ZES_BLK - 1u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t ciw2 = seq ? niw : inv[min(base + lane, ZES_BLK - 1u)];
        cb = base;
        nb = base + 64u;
        nw = __hip_atomic_load(&mo[min(nb + lane, ZES_BLK - 1u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        niw = inv[min(nb + lane, ZES_BLK - 1u)];
        cw = cw2;
        ciw = ciw2;
      }
      {
        const uint64_t lit = __ballot(cw == LAZY_EVAL_LIT) >> (p - base);
        if (lit & 1ull) {
          const uint32_t run = (~lit) ? (uint32_t)__builtin_ctzll(~lit) : 64u;
          const uint32_t step = min(min(run, 64u - (p - base)), cnt - p);
          if (lane < step) atomicOr(&S.v1[(p + lane) >> 5], 1u << ((p + lane) & 31u));
          p += step;
          continue;
        }
      }
      uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(p - base));
      if (lane == 0) atomicOr(&S.v1[p >> 5], 1u << (p & 31u));
      if (m == 0u) {  // nobody evaluated this position yet
        if (p >= tbase) {
          m = LAZY_EVAL_LIT;  // (cannot happen: phase 0 evaluated the tail; keeps a corrupted table from hanging the wavefront)
        } else {
          // the inv entries of the 64 positions from p on: a run of positions without any candidate (the first period of
          // periodic data, when no window chain has been there) is settled at once
          const uint32_t po = p - base;  // (lane k of ciw: the inv entry of position base + k)
          const uint64_t none = __ballot(ciw == ZES_INV_NONE && base + lane < tbase) >> po;
          const uint32_t nrun = min((~none) ? (uint32_t)__builtin_ctzll(~none) : 64u, 64u - po);
          if (nrun > 1u) {
            if (lane >= po && lane < po + nrun) {
              mo[base + lane] = LAZY_EVAL_LIT;
              atomicOr(&S.v1[(base + lane) >> 5], 1u << ((base + lane) & 31u));
            }
            p += nrun;  // (the chunk in hand stays: its words of the positions behind p are all it is asked for)
            continue;
          }
          if (++nev > budget3) {  // not the periodic block the probe took it for: the windows get their chains after all
            if (lane == 0) S.abort3 = 1u;
            break;
          }
          const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)ciw, (int)po);
          m = lazy_wave_eval(S, iv, sd, p, T, avail);
          if (lane == 0) mo[p] = m;
          if ((m & ZES_TOK_MATCH) && zes_tok_len(m) == ZES_MAXMATCH) {
            // A maximal match: on periodic data the chain goes on like this, 258 bytes at a time.  The next 64 positions
            // it would stand on are tested side by side, lane j the j-th: does the NEAREST candidate match at full
            // length?  Then the reference takes it and stops looking (src/lz77.ts:89-91), whatever the other candidates
            // are, so the positions up to the first lane that fails are settled: their words are written and the chain
            // moves past all of them at once.
            const uint32_t pj = p + ZES_MAXMATCH * (lane + 1u);
            // (not inside the pre-evaluated tail, where a match that reaches the block's last three bytes is dropped: :95)
            const bool inr = pj < tbase && pj + ZES_MAXMATCH + 3u <= T && pj + ZES_MAXMATCH <= avail;
            const uint32_t ivj = inv[inr ? pj : 0u];
            const bool has = inr && ivj != ZES_INV_NONE;
            const uint32_t dj = sd[has ? (ivj & 0x1FFFFu) : 0u];  // distance to the nearest candidate
            const uint32_t qj = has ? pj - dj : 0u, pp = has ? pj : 0u;
            bool full = has;
            for (uint32_t wv = 0; wv < 32u && __ballot(full); wv += 4u) {  // bytes 0..255, 32 at a time (eight reads in flight)
              uint64_t x = 0;
#pragma unroll
              for (uint32_t t = 0; t < 4u; t++) x |= lz_ld64(S.in, qj + 8u * (wv + t)) ^ lz_ld64(S.in, pp + 8u * (wv + t));
              full = full && x == 0ull;
            }
            full = full && ((lz_ld32(S.in, qj + 256u) ^ lz_ld32(S.in, pp + 256u)) & 0xffffu) == 0u;  // bytes 256, 257
            const uint64_t okm = __ballot(full);
            const uint32_t nk = (~okm) ? (uint32_t)__builtin_ctzll(~okm) : 64u;  // leading lanes that are settled
            if (lane < nk) {
              mo[pj] = ZES_TOK_MATCH | ((ZES_MAXMATCH - 3u) << 16) | (dj - 1u);
              atomicOr(&S.v1[pj >> 5], 1u << (pj & 31u));
            }
            p += ZES_MAXMATCH * nk;
          }
        }
      }
      p += (m & ZES_TOK_MATCH) ? zes_tok_len(m) : 1u;
    }
    // the block's last two bytes are always literals (src/lz77.ts:116-117): the chain ends on them
    if (lane < 2u && cnt + lane < T && cnt + lane >= p) atomicOr(&S.v1[(cnt + lane) >> 5], 1u << ((cnt + lane) & 31u));
  }
  if (walk3) {
    __syncthreads();
    if (S.abort3) {  // (uniform) the probe was wrong about this block: once more, every window with its chain
      probe = false;
      __syncthreads();
      continue;
    }
    for (uint32_t i = tid; i < ZES_BLK / 32; i += MATCH_THREADS) tm[4u + i] = S.v1[i];
    __syncthreads();
    if (tid == 0) tm[0] = 1u;
  }
  break;
  }
  LSTAMP(5);
}

// ------------------------------------------------------------------------------------------
// k_lz_parse: the greedy chain p -> p + len | p + 1 of one block, 1024 threads.
// The chain is sequential, but each hop is <= 258 positions, so:
//   A   every 64-position chunk gets its exit map by in-wave pointer doubling: for each
//       possible entry lane, how far past the chunk the chain lands (u8, 255 = look it up)
//   B   each wave folds its region of 128 chunks backwards (sliding window of 5 chunk maps in
//       registers): entry offset into the region -> offset past the region
//   C   16 region tables are chained from position 0; each wave then walks its own region once,
//       noting the entry lane of every chunk the chain touches
//   D   per touched chunk, the visited lanes are found by binary hop decomposition (again
//       doubling), counted, scanned, and the tokens and both histograms written in parallel.
// ------------------------------------------------------------------------------------------
__device__ unsigned long long* g_parse_dbg = nullptr;  // ZES_DEBUG_PHASES: cycle stamps [g][8]
void zes_parse_set_dbg(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_parse_dbg), &p, sizeof p); }
#define PSTAMP(i)                                                                                          \
  do {                                                                                                     \
    if (g_parse_dbg && threadIdx.x == 0) g_parse_dbg[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); \
  } while (0)
#define PARSE_CHUNKS (ZES_BLK / 64)        // 2048
#define PARSE_REGION 128                    // chunks per wave
#define PARSE_NOENTRY 0xFFu
#define PARSE_BATCH 8u                      // chunks whose global loads are issued together
template <bool SMALL>
struct ParseSmemT;
template <>
struct ParseSmemT<false> {
  union {
    uint8_t xmap[PARSE_CHUNKS][64];         // phase A-C
    struct {
      unsigned long long mask[PARSE_CHUNKS];  // phase D (the exit maps are dead by then)
      uint32_t cpre[PARSE_CHUNKS];
    } d;
  } u;
  uint16_t rtab[PARSE_THREADS / 64][5 * 64];  // region transfer tables (entry offsets 0..319)
  uint8_t centry[PARSE_CHUNKS];
  uint8_t cplain[PARSE_CHUNKS];              // 1: the chunk holds no match word at all (every hop is +1)
  uint32_t rentry[PARSE_THREADS / 64 + 1];
  uint32_t wsum[PARSE_THREADS / 64];
  uint32_t lh[288];
  uint32_t dh[32];
};
// Blocks whose chain arrives as a bit mask or is built from the match list need none of the exit maps: 31 KiB
// instead of 143, and with 64 registers two workgroups share a compute unit (k_lz_parse_small).
template <>
struct ParseSmemT<true> {
  struct {
    struct {
      unsigned long long mask[PARSE_CHUNKS];
      uint32_t cpre[PARSE_CHUNKS];
    } d;
  } u;
  uint32_t srt[ZES_MLIST_WORDS];     // the match list sorted by position
  uint8_t acc[ZES_MLIST_WORDS];      // "taken by the greedy parse"
  uint32_t mstart[ZES_BLK / 32];     // a bit per position: a listed match starts here; in the end: a taken one
  uint8_t cplain[PARSE_CHUNKS];
  uint32_t wsum[PARSE_THREADS / 64];
  uint32_t lh[288];
  uint32_t dh[32];
};

// All eight loaded words are needed "now": one wait for the whole batch here, instead of a full
// vmcnt(0) wait (which also waits for the stores issued in between) at every later use under a branch.
#define PARSE_ARRIVE8(a)                                                                                              \
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
static_assert(PARSE_BATCH == 8u, "PARSE_ARRIVE8 names eight registers");

// exact landing offset past the chunk for entry lane e (slow path behind xmap code 255)
__device__ static uint32_t parse_follow(const uint32_t* mi, uint32_t T, uint32_t chunk, uint32_t e) {
  uint32_t cur = e;
  while (cur < 64u) {
    const uint32_t p = chunk * 64u + cur;
    const uint32_t m = p < T ? mi[p] : 0u;
    cur += (m & ZES_TOK_MATCH) ? zes_tok_len(m) : 1u;
  }
  return cur - 64u;
}

#define PARSE_ARRIVE4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
template <bool SMALL>
__device__ __forceinline__ static void parse_body(ParseSmemT<SMALL>& S, const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                  ZesBlk* __restrict__ blks, const uint32_t* __restrict__ match_in,
                                                  uint32_t* __restrict__ tok_out, uint32_t* __restrict__ hists,
                                                  const uint32_t* __restrict__ tmask_all, const uint32_t* __restrict__ mlist_all) {
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const ZesBlk bk = blks[g];
  const ZesBuf bf = bufs[bk.buf];
  const uint32_t T = bk.len;
  const uint8_t* src = d_in + bf.in_off + (uint64_t)bk.blk * ZES_BLK;
  const uint32_t* mi = match_in + (uint64_t)g * ZES_BLK;
  uint32_t* to = tok_out + (uint64_t)g * ZES_BLK;
  const uint32_t nchunks = (T + 63u) >> 6;
  const uint32_t nregions = (nchunks + PARSE_REGION - 1) / PARSE_REGION;
  // k_lz_match_lazy's blocks come with the positions of the chain as a bit mask (unless its chains did not merge):
  // phases A to D1, which find those positions, are skipped.  Which kernel a block belonged to stands in the head
  // word of its match list (a count from k_lz_match, all ones from k_lz_match_lazy).
  const uint32_t* tmk = tmask_all + (uint64_t)g * ZES_TMASK_WORDS;
  const uint32_t* ml = mlist_all + (uint64_t)g * ZES_MLIST_WORDS;
  const uint32_t nml = ml[0];
  const bool lazyblk = nml == 0xFFFFFFFFu;
  const bool havemask = lazyblk && tmk[0] == 1u;              // (uniform)
  const bool fewmatches = !lazyblk && nml <= ZES_MLIST_CAP;  // (uniform)
  if (SMALL != (havemask || fewmatches)) return;              // the other kernel's block
  for (uint32_t i = tid; i < 288; i += PARSE_THREADS) S.lh[i] = 0;
  if (tid < 32) S.dh[tid] = 0;
  if constexpr (!SMALL)
    for (uint32_t i = tid; i < PARSE_CHUNKS; i += PARSE_THREADS) S.centry[i] = PARSE_NOENTRY;
  PSTAMP(0);
  // ---- A: exit map of every chunk ----
  const uint32_t c_lo = wave * PARSE_REGION, c_hi = min(nchunks, c_lo + PARSE_REGION);
  // k_lz_match's blocks with few matches (incompressible data: ~260): the chain is every position except the insides
  // of the matches the greedy parse takes — a listed match is taken iff it starts at or behind the end of the last
  // one taken (src/lz77.ts:39-47,95): sorted, then one pass in order.  (Phases A to D1 cost 213k cycles a block there.)
  if (havemask) {
    const unsigned long long* t64 = reinterpret_cast<const unsigned long long*>(tmk + 4);
    for (uint32_t c = tid; c < PARSE_CHUNKS; c += PARSE_THREADS) {
      S.u.d.mask[c] = c < nchunks ? t64[c] : 0ull;
      S.cplain[c] = 0;
    }
    __syncthreads();
  } else if (fewmatches) {
    if constexpr (SMALL) {
    // Sorted by position without comparing anything: the listed positions are distinct, so a bit per position and a
    // running count per 32-bit word of that bitmap give every entry its rank (any list length up to ZES_MLIST_CAP).
    uint16_t* wpre = reinterpret_cast<uint16_t*>(S.u.d.cpre);  // [ZES_BLK / 32] listed positions in front of each word (cpre is written in D2)
    for (uint32_t c = tid; c < PARSE_CHUNKS; c += PARSE_THREADS) {
      const uint32_t lo = c * 64u;
      S.u.d.mask[c] = lo + 64u <= T ? ~0ull : (lo < T ? ((1ull << (T - lo)) - 1ull) : 0ull);  // every position, so far
      S.cplain[c] = 1;
    }
    for (uint32_t i = tid; i < ZES_BLK / 32; i += PARSE_THREADS) S.mstart[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nml; i += PARSE_THREADS) {
      const uint32_t pe = ml[1u + i] & 0x1FFFFu;
      atomicOr(&S.mstart[pe >> 5], 1u << (pe & 31u));
    }
    __syncthreads();
    {
      static_assert(ZES_BLK / 32 == 4 * PARSE_THREADS, "four bitmap words per thread");
      uint32_t c4[4], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        c4[k] = (uint32_t)__popc(S.mstart[4u * tid + k]);
        sum += c4[k];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const uint32_t t = __shfl_up(incl, dlt);
        if ((int)lane >= dlt) incl += t;
      }
      if (lane == 63) S.wsum[wave] = incl;
      __syncthreads();
      uint32_t run = incl - sum;
      for (uint32_t w = 0; w < wave; w++) run += S.wsum[w];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        wpre[4u * tid + k] = (uint16_t)run;
        run += c4[k];
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nml; i += PARSE_THREADS) {
      const uint32_t e = ml[1u + i], pe = e & 0x1FFFFu;
      const uint32_t r = (uint32_t)wpre[pe >> 5] + (uint32_t)__popc(S.mstart[pe >> 5] & ((1u << (pe & 31u)) - 1u));
      S.srt[r] = e;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t cur = 0;
      for (uint32_t i = 0; i < nml; i++) {
        const uint32_t e = S.srt[i], p = e & 0x1FFFFu, L = (e >> 17) + 3u;
        const bool take = p >= cur;
        S.acc[i] = take ? 1 : 0;
        cur = take ? p + L : cur;
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nml; i += PARSE_THREADS) {
      const uint32_t e = S.srt[i], p = e & 0x1FFFFu, L = (e >> 17) + 3u;
      if (S.acc[i]) {  // the positions inside a taken match are not on the chain
        S.cplain[p >> 6] = 0;
        for (uint32_t q = p + 1u; q < p + L;) {
          const uint32_t w = q >> 6, b0 = q & 63u, nb = min(64u - b0, p + L - q);
          const unsigned long long bits = (nb >= 64u ? ~0ull : ((1ull << nb) - 1ull)) << b0;
          atomicAnd(&S.u.d.mask[w], ~bits);
          q += nb;
        }
      } else {
        atomicAnd(&S.mstart[p >> 5], ~(1u << (p & 31u)));  // what is left in the bitmap: the starts of the taken matches
      }
    }
    __syncthreads();
    }  // (the big kernel never gets such a block)
  } else if constexpr (!SMALL) {
  // Batches of PARSE_BATCH chunks: all their match words are requested before the first is used (one
  // memory latency per batch instead of per chunk).  A chunk without any match (nearly all of them on
  // incompressible input) needs no pointer doubling: every lane leaves at offset 0.
  for (uint32_t cb = c_lo; cb < c_hi; cb += PARSE_BATCH) {
    uint32_t mw[PARSE_BATCH];
    // (loads are unconditional with clamped addresses: a load under a branch makes the compiler wait
    // for every outstanding memory operation, stores included, before the next one)
#pragma unroll
    for (uint32_t k = 0; k < PARSE_BATCH; k++) {
      const uint32_t p = (cb + k) * 64u + lane;
      mw[k] = mi[min(p, T - 1u)];
    }
    PARSE_ARRIVE8(mw);
#pragma unroll
    for (uint32_t k = 0; k < PARSE_BATCH; k++) {
      const uint32_t c = cb + k;
      if (c >= c_hi) break;
      const uint32_t m = (c * 64u + lane < T) ? mw[k] : 0u;
      const bool plain = __ballot(m & ZES_TOK_MATCH) == 0ull;
      uint32_t v = lane + ((m & ZES_TOK_MATCH) ? zes_tok_len(m) : 1u);
      if (!plain) {
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const uint32_t nv = __shfl(v, (int)(v & 63u));
          if (v < 64u) v = nv;
        }
      }
      S.u.xmap[c][lane] = plain ? (uint8_t)0 : (uint8_t)min(v - 64u, 255u);
      if (lane == 0) S.cplain[c] = plain ? 1 : 0;
    }
  }
  __syncthreads();

  PSTAMP(1);
  // ---- B: fold the region backwards.  E_k[e] = offset past the region end reached from lane e
  //         of chunk c + k; a hop lands at most 4 chunks ahead (overshoot <= 257) ----
  if (wave < nregions) {
    const uint32_t nch = c_hi - c_lo;  // chunks of this region that exist
    // beyond the region's last chunk the "offset past the region" is just the landing offset
    uint32_t E1 = lane, E2 = 64u + lane, E3 = 128u + lane, E4 = 192u + lane, E5 = 256u + lane;
    for (int k = (int)nch - 1; k >= 0; k--) {
      const uint32_t c = c_lo + (uint32_t)k;
      uint32_t E0;
      if (S.cplain[c]) {
        // no match in the chunk: every lane leaves it at offset 0 of the next chunk
        E0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)E1);
      } else {
        uint32_t ov = S.u.xmap[c][lane];
        if (ov == 255u) ov = parse_follow(mi, T, c, lane);
        const uint32_t d = ov >> 6, sl = ov & 63u;
        const uint32_t a1 = __shfl(E1, (int)sl), a2 = __shfl(E2, (int)sl), a3 = __shfl(E3, (int)sl), a4 = __shfl(E4, (int)sl),
                       a5 = __shfl(E5, (int)sl);
        E0 = d == 0 ? a1 : d == 1 ? a2 : d == 2 ? a3 : d == 3 ? a4 : a5;
      }
      E5 = E4;
      E4 = E3;
      E3 = E2;
      E2 = E1;
      E1 = E0;
    }
    // E1..E5 now describe chunks c_lo .. c_lo+4: entry offsets 0..319 into the region
    S.rtab[wave][lane] = (uint16_t)E1;
    S.rtab[wave][64 + lane] = (uint16_t)E2;
    S.rtab[wave][128 + lane] = (uint16_t)E3;
    S.rtab[wave][192 + lane] = (uint16_t)E4;
    S.rtab[wave][256 + lane] = (uint16_t)E5;
  }
  __syncthreads();

  PSTAMP(2);
  // ---- C: chain the regions, then every wave walks its own region from its true entry ----
  if (tid == 0) {
    uint32_t e = 0;
    for (uint32_t r = 0; r < nregions; r++) {
      S.rentry[r] = e;
      e = S.rtab[r][e];
    }
  }
  __syncthreads();
  if (wave < nregions && lane == 0) {
    uint32_t cur = S.rentry[wave];  // offset from the region's first position
    const uint32_t lim = (c_hi - c_lo) * 64u;
    while (cur < lim) {
      const uint32_t c = c_lo + (cur >> 6), e = cur & 63u;
      S.centry[c] = (uint8_t)e;
      uint32_t ov = S.u.xmap[c][e];
      if (ov == 255u) ov = parse_follow(mi, T, c, e);
      cur = ((cur >> 6) + 1u) * 64u + ov;
    }
  }
  __syncthreads();

  PSTAMP(3);
  // ---- D1: visited lanes of every touched chunk: one lane per chunk walks the chunk's chain from its entry lane
  // (at most 64 hops, a dozen on text), 64 chunks of the wave's region side by side; the match words come from
  // global memory, a line of sixteen serving the next hops.  (Round 1 found the visited lanes of one chunk at a
  // time with all 64 lanes, by binary hop decomposition: a dozen cross-lane shuffles per chunk, 287k cycles per
  // block on text against the ~40k of this walk.) ----
  for (uint32_t cb = c_lo; cb < c_hi; cb += 64u) {
    const uint32_t c = cb + lane;
    const bool have = c < c_hi;
    const uint32_t cc = min(c, PARSE_CHUNKS - 1u);
    const uint32_t e = have ? (uint32_t)S.centry[cc] : PARSE_NOENTRY;
    const bool plain = S.cplain[cc] != 0;
    const uint32_t left = (have && cc * 64u < T) ? min(64u, T - cc * 64u) : 0u;  // positions of the chunk that exist
    unsigned long long mk = 0ull;
    if (e != PARSE_NOENTRY && plain)  // no match in the chunk: the chain visits every position from the entry lane on
      mk = (~0ull << e) & (left >= 64u ? ~0ull : ((1ull << left) - 1ull));
    uint32_t cur = (e != PARSE_NOENTRY && !plain) ? e : 64u;
    while (__ballot(cur < left)) {
      const bool act = cur < left;
      const uint32_t m = mi[act ? cc * 64u + cur : 0u];  // unconditional (clamped) load
      mk |= act ? (1ull << cur) : 0ull;
      cur += act ? ((m & ZES_TOK_MATCH) ? zes_tok_len(m) : 1u) : 0u;
    }
    if (have) S.u.d.mask[c] = mk;  // xmap of chunk c is dead: the walk of phase C is over
  }
  __syncthreads();
  // NOTE: mask[] aliases xmap[]; chunk c's mask (8 B at c*8) overlaps xmap rows c/8 — all reads
  // of xmap finished at the barrier above phase D1 only for the walk; D1 itself does not read xmap.

  PSTAMP(4);
  }  // (neither mask nor list: the chain is searched)
  unsigned long long mymask[2] = {0ull, 0ull};  // thread t owns chunks 2t, 2t+1 for the scan
  // ---- D2: token offsets = exclusive scan of the per-chunk counts ----
  uint32_t cnt2[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t c = tid * 2 + k;
    mymask[k] = c < nchunks ? S.u.d.mask[c] : 0ull;
    cnt2[k] = (uint32_t)__popcll(mymask[k]);
  }
  const uint32_t mysum = cnt2[0] + cnt2[1];
  uint32_t incl = mysum;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const uint32_t t = __shfl_up(incl, dlt);
    if ((int)lane >= dlt) incl += t;
  }
  if (lane == 63) S.wsum[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < PARSE_THREADS / 64; w++) {
    const uint32_t sm = S.wsum[w];
    if (w < wave) wbase += sm;
    total += sm;
  }
  const uint32_t ex = wbase + incl - mysum;
  if (tid * 2 < PARSE_CHUNKS) S.u.d.cpre[tid * 2] = ex;
  if (tid * 2 + 1 < PARSE_CHUNKS) S.u.d.cpre[tid * 2 + 1] = ex + cnt2[0];
  __syncthreads();

  PSTAMP(5);
  // ---- D3: tokens + histograms ----
  constexpr uint32_t DB = SMALL ? 4u : PARSE_BATCH;  // chunks per batch of loads
  for (uint32_t cb = c_lo; cb < c_hi; cb += DB) {
    uint32_t mw[DB], by[DB];
    unsigned long long mks[DB];
    bool ism[DB];
#pragma unroll
    for (uint32_t k = 0; k < DB; k++) {  // input bytes of visited positions; match words where the chunk has any
      const uint32_t c = cb + k;
      mks[k] = c < c_hi ? S.u.d.mask[c] : 0ull;
      const bool vis = (mks[k] >> lane) & 1ull;  // implies p < T
      const uint32_t p = c * 64u + lane;
      by[k] = src[vis ? p : min(lane, T - 1u)];  // unconditional loads (see phase A)
      // a block parsed from its match list has result words only where a match was found (k_lz_match did not clear the
      // others): a word is read where a taken match starts, nowhere else
      bool want = vis && !S.cplain[min(c, PARSE_CHUNKS - 1u)];
      if constexpr (SMALL) {
        const uint32_t pw = min(p, ZES_BLK - 1u);
        if (fewmatches) want = want && ((S.mstart[pw >> 5] >> (pw & 31u)) & 1u);
      }
      ism[k] = want;
      mw[k] = mi[want ? p : lane];
    }
    if constexpr (SMALL) {
      PARSE_ARRIVE4(by);
      PARSE_ARRIVE4(mw);
    } else {
      PARSE_ARRIVE8(by);
      PARSE_ARRIVE8(mw);
    }
#pragma unroll
    for (uint32_t k = 0; k < DB; k++) {
      const uint32_t c = cb + k;
      const unsigned long long mk = mks[k];
      if ((mk >> lane) & 1ull) {
        const uint32_t m = ism[k] ? mw[k] : 0u;
        const uint32_t rank = (uint32_t)__popcll(mk & zes_lanemask_lt());
        uint32_t tv;
        if (m & ZES_TOK_MATCH) {
          tv = m;
          atomicAdd(&S.lh[257u + zes_len_code(zes_tok_len(m))], 1u);
          atomicAdd(&S.dh[zes_dist_code(zes_tok_dist(m))], 1u);
        } else {
          tv = by[k];
          atomicAdd(&S.lh[tv], 1u);
        }
        to[S.u.d.cpre[c] + rank] = tv;
      }
    }
  }
  __syncthreads();
  PSTAMP(6);
  if (tid == 0) {
    S.lh[256] = 1;  // EOB (src/deflate.ts:58)
    blks[g].ntok = total;
  }
  __syncthreads();
  uint32_t* hg = hists + (uint64_t)g * 320;
  for (uint32_t i = tid; i < 288; i += PARSE_THREADS) hg[i] = S.lh[i];
  if (tid < 32) hg[288 + tid] = S.dh[tid];
  PSTAMP(7);
}

__global__ __launch_bounds__(PARSE_THREADS) void k_lz_parse(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                            ZesBlk* __restrict__ blks, const uint32_t* __restrict__ match_in,
                                                            uint32_t* __restrict__ tok_out, uint32_t* __restrict__ hists,
                                                            const uint32_t* __restrict__ tmask_all, const uint32_t* __restrict__ mlist_all) {
  __shared__ __align__(16) ParseSmemT<false> S;
  parse_body<false>(S, d_in, bufs, blks, match_in, tok_out, hists, tmask_all, mlist_all);
}
// the blocks with a chain mask or a short match list: two workgroups per compute unit
__global__ __launch_bounds__(PARSE_THREADS, 8) void k_lz_parse_small(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                                     ZesBlk* __restrict__ blks, const uint32_t* __restrict__ match_in,
                                                                     uint32_t* __restrict__ tok_out, uint32_t* __restrict__ hists,
                                                                     const uint32_t* __restrict__ tmask_all,
                                                                     const uint32_t* __restrict__ mlist_all) {
  __shared__ __align__(16) ParseSmemT<true> S;
  parse_body<true>(S, d_in, bufs, blks, match_in, tok_out, hists, tmask_all, mlist_all);
}

// ------------------------------------------------------------------------------------------
// k_huff: per block, 256 threads.
// Package-merge restated as parallel merges: the reference's per-level
// `sort(leaves ++ pairs(prev))` is a stable merge of two sorted runs (leaves first on ties), so
// every item's slot is a binary-search rank; a code length is the number of levels whose
// selected prefix still contains the leaf (the selected items of a level are a prefix of it).
// ------------------------------------------------------------------------------------------
__device__ unsigned long long* g_huff_dbg = nullptr;  // ZES_DEBUG_PHASES: cycle stamps [g][8]
void zes_huff_set_dbg(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_huff_dbg), &p, sizeof p); }
#define USTAMP(i)                                                                                        \
  do {                                                                                                   \
    if (g_huff_dbg && threadIdx.x == 0) g_huff_dbg[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); \
  } while (0)
#define HUFF_THREADS 256
#define PM_MAXN 288
struct HuffSmem {
  uint32_t hist[320];           // working histogram of the current alphabet
  uint32_t lw[PM_MAXN];         // leaf weights, ascending (count, symbol)
  uint16_t lsym[PM_MAXN];       // symbol of sorted leaf i
  uint32_t w[2][2 * PM_MAXN];   // merged weights of the previous / current level
  uint32_t pk[PM_MAXN];         // packages of the current level
  uint16_t leafpos[15][PM_MAXN];
  uint32_t a_k[16];
  uint8_t lens[320];            // [0..288) lit/len, [288..320) dist
  uint8_t clens[32];
  uint16_t codes[320];          // bit-reversed codes, same layout
  uint16_t ccodes[32];
  uint8_t rl_sym[320];
  uint8_t rl_val[320];
  uint32_t nrl;
  uint32_t hdr[ZES_HDR_WORDS];
  uint32_t hdr_bits;
  uint32_t total_bits;
  uint32_t ccount[16], cfirst[16], crun[16];     // canonical codes: symbols per length, first code, symbols seen so far
  uint16_t cwave[HUFF_THREADS / 64][16];         // canonical codes: symbols of each length per wave of the current pass
  uint8_t cl[320];                               // lit/len lengths followed by the distance lengths (src/deflate.ts:81-97)
  uint16_t run_tok[320];                         // run-length coding: tokens before the run that starts at this index
  uint32_t wsum[HUFF_THREADS / 64];
  unsigned long long starts[5];                  // run-length coding: bit i = a run starts at index i
};

__device__ static inline uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t v) {  // #{a[i] < v}
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ static inline uint32_t upper_bound_u32(const uint32_t* a, uint32_t n, uint32_t v) {  // #{a[i] <= v}
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Code lengths of alphabet hist[0..nsym) limited to L, into lens[0..nsym).  All threads call it.
// A level of the package-merge is one round trip: every thread looks up the slots of its items of the level (a leaf
// and a package, two of each for alphabets of more than 256 used symbols) side by side — leaf i goes behind the packages lighter than it, package j behind the leaves not heavier
// (leaves precede packages on ties: src/huffman.ts:79-99) — by fixed-length binary searches whose loads overlap;
// a package's weight is read as the sum of its two items of the level before, so there is no separate pass (and
// barrier) that builds the packages.
__device__ static void pm_lengths(HuffSmem& S, const uint32_t* hist, uint32_t nsym, uint32_t L, uint8_t* lens) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  __syncthreads();
  // rank sort of the used symbols by (count, symbol)  — src/huffman.ts:67,79-85,95-99
  uint32_t n = 0;
  for (uint32_t s0 = 0; s0 < nsym; s0 += HUFF_THREADS) n += (uint32_t)__syncthreads_count(s0 + tid < nsym && hist[s0 + tid] != 0u);
  for (uint32_t s = tid; s < nsym; s += HUFF_THREADS) {
    lens[s] = 0;
    const uint32_t c = hist[s];
    if (c) {
      uint32_t rank = 0;
      for (uint32_t j = 0; j < nsym; j++) {
        const uint32_t cj = hist[j];
        rank += (cj != 0) && (cj < c || (cj == c && j < s));
      }
      S.lw[rank] = c;
      S.lsym[rank] = (uint16_t)s;
    }
  }
  __syncthreads();
  if (n == 0) return;  // empty table
  if (n == 1) {        // src/huffman.ts:71-75
    if (tid == 0) lens[S.lsym[0]] = 1;
    __syncthreads();
    return;
  }
  // level 1: the sorted leaves; odd length drops the last item (src/huffman.ts:100-102)
  for (uint32_t i = tid; i < n; i += HUFF_THREADS) {
    S.w[0][i] = S.lw[i];
    S.leafpos[0][i] = (uint16_t)i;
  }
  uint32_t len_prev = n & ~1u;
  uint32_t sel = 0;
  __syncthreads();
  static_assert(PM_MAXN <= 2 * HUFF_THREADS && PM_MAXN < 512, "two items of either kind per thread, nine search steps");
  const uint32_t st0 = 1u << (31u - (uint32_t)__builtin_clz(n));  // first search step: the largest power of two <= n (>= np)
  const bool wide = n > HUFF_THREADS;                             // (uniform) a second leaf / package per thread
  for (uint32_t k = 1; k < L; k++) {
    const uint32_t np = len_prev >> 1;  // packages of this level: pairs of the level before (src/huffman.ts:87-94); 1 <= np < n
    const uint2* wp2 = reinterpret_cast<const uint2*>(S.w[sel]);
    uint32_t* wc = S.w[sel ^ 1];
    // thread t: leaves t and t + 256, packages t and t + 256 (fixed roles: no selects inside the search)
    const uint32_t i0 = min(tid, n - 1u), i1 = min(tid + HUFF_THREADS, n - 1u);
    const uint32_t j0 = min(tid, np - 1u), j1 = min(tid + HUFF_THREADS, np - 1u);
    const uint32_t lv0 = S.lw[i0], lv1 = S.lw[i1];
    const uint2 pa = wp2[j0], pb = wp2[j1];
    const uint32_t pv0 = pa.x + pa.y, pv1 = pb.x + pb.y;
    uint32_t ll0 = 0, ll1 = 0, pl0 = 0, pl1 = 0;  // leaf: #{packages < lv}; package: #{leaves <= pv}
    if (!wide) {
#pragma unroll 1
      for (uint32_t st = st0; st; st >>= 1) {
        const uint32_t ml = ll0 + st, mp = pl0 + st;
        const uint2 x = wp2[min(ml, np) - 1u];
        const uint32_t y = S.lw[min(mp, n) - 1u];
        ll0 = (ml <= np && x.x + x.y < lv0) ? ml : ll0;
        pl0 = (mp <= n && y <= pv0) ? mp : pl0;
      }
    } else {
#pragma unroll 1
      for (uint32_t st = st0; st; st >>= 1) {
        const uint32_t ml0 = ll0 + st, ml1 = ll1 + st, mp0 = pl0 + st, mp1 = pl1 + st;
        const uint2 x0 = wp2[min(ml0, np) - 1u], x1 = wp2[min(ml1, np) - 1u];
        const uint32_t y0 = S.lw[min(mp0, n) - 1u], y1 = S.lw[min(mp1, n) - 1u];
        ll0 = (ml0 <= np && x0.x + x0.y < lv0) ? ml0 : ll0;
        ll1 = (ml1 <= np && x1.x + x1.y < lv1) ? ml1 : ll1;
        pl0 = (mp0 <= n && y0 <= pv0) ? mp0 : pl0;
        pl1 = (mp1 <= n && y1 <= pv1) ? mp1 : pl1;
      }
    }
    if (tid < n) {
      wc[tid + ll0] = lv0;
      S.leafpos[k][tid] = (uint16_t)(tid + ll0);
    }
    if (tid < np) wc[tid + pl0] = pv0;
    if (wide) {
      if (tid + HUFF_THREADS < n) {
        wc[tid + HUFF_THREADS + ll1] = lv1;
        S.leafpos[k][tid + HUFF_THREADS] = (uint16_t)(tid + HUFF_THREADS + ll1);
      }
      if (tid + HUFF_THREADS < np) wc[tid + HUFF_THREADS + pl1] = pv1;
    }
    len_prev = (n + np) & ~1u;
    sel ^= 1;
    __syncthreads();
  }
  // selected prefix per level, from the last level down (src/huffman.ts:106-115): a = #{i : leafpos[k][i] < m}, counted
  // by the first wavefront (leaf positions ascend with i, so the leaves below m are a prefix)
  if (tid < 64u) {
    uint32_t m = len_prev;
    for (int k = (int)L - 1; k >= 0; k--) {
      uint32_t a = 0;
#pragma unroll
      for (uint32_t c = 0; c < (PM_MAXN + 63) / 64; c++) {
        const uint32_t i = c * 64u + lane;
        a += (uint32_t)__popcll(__ballot(i < n && S.leafpos[k][min(i, n - 1u)] < m));
      }
      if (lane == 0) S.a_k[k] = a;
      m = 2u * (m - a);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += HUFF_THREADS) {
    uint32_t l = 0;
    for (uint32_t k = 0; k < L; k++) l += (S.a_k[k] > i);
    lens[S.lsym[i]] = (uint8_t)l;
  }
  __syncthreads();
}

// canonical codes (src/huffman.ts:117-151), stored bit-reversed for LSB-first packing.  The code of a
// symbol is the first code of its length plus its rank among the symbols of that length: counts per
// length by LDS atomics, ranks from ballots (lower lanes) plus per-wave counts (earlier waves and passes).
__device__ static void canon_codes(HuffSmem& S, const uint8_t* lens, uint32_t nsym, uint16_t* codes) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid < 16) {
    S.ccount[tid] = 0;
    S.crun[tid] = 0;
  }
  __syncthreads();
  for (uint32_t s = tid; s < nsym; s += HUFF_THREADS)
    if (lens[s]) atomicAdd(&S.ccount[lens[s]], 1u);
  __syncthreads();
  if (tid == 0) {
    uint32_t code = 0;
    for (uint32_t l = 1; l < 16; l++) {
      S.cfirst[l] = code;
      code = (code + S.ccount[l]) << 1;
    }
  }
  for (uint32_t s0 = 0; s0 < nsym; s0 += HUFF_THREADS) {  // uniform trip count
    const uint32_t s = s0 + tid;
    const uint32_t l = s < nsym ? lens[s] : 0u;
    uint64_t same = ~0ull;
#pragma unroll
    for (int bt = 0; bt < 4; bt++) {
      const bool bit = (l >> bt) & 1u;
      const uint64_t bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const uint32_t below = (uint32_t)__popcll(same & zes_lanemask_lt());
    if (lane < 16) S.cwave[wave][lane] = 0;
    __syncthreads();  // cfirst / crun of the previous pass visible, cwave cleared
    if (l && below == 0) S.cwave[wave][l] = (uint16_t)__popcll(same);
    __syncthreads();
    uint32_t code = 0;
    if (l) {
      uint32_t rank = S.crun[l] + below;
      for (uint32_t w = 0; w < wave; w++) rank += S.cwave[w][l];
      code = __brev(S.cfirst[l] + rank) >> (32u - l);
    }
    if (s < nsym) codes[s] = (uint16_t)code;
    __syncthreads();
    if (tid < 16) {
      uint32_t add = 0;
      for (uint32_t w = 0; w < HUFF_THREADS / 64; w++) add += S.cwave[w][tid];
      S.crun[tid] += add;
    }
  }
  __syncthreads();
}

// LSB-first bits at a known offset of S.hdr (zeroed beforehand), from any thread
__device__ static inline void hdr_or(HuffSmem& S, uint32_t bitpos, uint32_t value, uint32_t nbits) {
  const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
  atomicOr(&S.hdr[w], value << sh);
  if (sh + nbits > 32u) atomicOr(&S.hdr[w + 1], value >> (32u - sh));
}

__global__ __launch_bounds__(HUFF_THREADS) void k_huff(ZesBlk* __restrict__ blks, const uint32_t* __restrict__ hists,
                                                       uint32_t* __restrict__ codes_out, uint32_t* __restrict__ hdr_out) {
  __shared__ HuffSmem S;
  const uint32_t g = blockIdx.x, tid = threadIdx.x;
  const uint32_t* hg = hists + (uint64_t)g * 320;
  for (uint32_t i = tid; i < 320; i += HUFF_THREADS) S.hist[i] = hg[i];
  for (uint32_t i = tid; i < ZES_HDR_WORDS; i += HUFF_THREADS) S.hdr[i] = 0;
  if (tid == 0) S.total_bits = 0;
  __syncthreads();

  USTAMP(0);
  pm_lengths(S, S.hist, 286, 15, S.lens);             // src/deflate.ts:78
  USTAMP(1);
  pm_lengths(S, S.hist + 288, 30, 15, S.lens + 288);  // src/deflate.ts:79
  USTAMP(2);
  canon_codes(S, S.lens, 286, S.codes);
  canon_codes(S, S.lens + 288, 30, S.codes + 288);
  USTAMP(3);

  // code-length sequence + run-length coding (src/deflate.ts:81-139), one thread per run
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) {
    S.a_k[0] = 256;  // highest used lit/len symbol (the end-of-block code is always used)
    S.a_k[1] = 0;    // highest used distance symbol
    S.nrl = 0;
  }
  __syncthreads();
  if (tid < 29 && S.hist[257 + tid]) atomicMax(&S.a_k[0], 257u + tid);
  if (tid < 30 && S.hist[288 + tid]) atomicMax(&S.a_k[1], tid);
  __syncthreads();
  const uint32_t HLIT = S.a_k[0] + 1u, HDIST = S.a_k[1] + 1u, ncl = HLIT + HDIST;  // <= 316
  for (uint32_t i = tid; i < ncl; i += HUFF_THREADS) S.cl[i] = i < HLIT ? S.lens[i] : S.lens[288 + i - HLIT];
  if (tid < 32) S.hist[tid] = 0;  // reused as the histogram of the run-length symbols
  __syncthreads();
  // a run = maximal stretch of equal lengths (it may cross the lit/len | distance border); the reference
  // cuts it into chunks of at most 138 (zeros) or 6 (other lengths): a chunk of 4 or more becomes one
  // repeat token (zeros) or the length plus a "repeat previous" token, a shorter one plain lengths
  // run starts as bit masks (320 indices = five 64-bit words, index ncl counts as a start)
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const uint32_t i = tid + q * HUFF_THREADS;
    const bool st = i < 320u && (i >= ncl || i == 0 || S.cl[i] != S.cl[i - 1]);
    const uint64_t m = __ballot(st);
    if (lane == 0 && (i >> 6) < 5u) S.starts[i >> 6] = m;
  }
  __syncthreads();
  uint32_t mytok[2] = {0, 0}, myn[2] = {0, 0};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const uint32_t i = tid + q * HUFF_THREADS;
    if (i < ncl && (i == 0 || S.cl[i] != S.cl[i - 1])) {
      const uint32_t v = S.cl[i];
      // next start after i: first set bit above i in the masks
      uint32_t nxt = ncl;
      {
        uint32_t wd = (i + 1u) >> 6;
        uint64_t mm = wd < 5u ? (S.starts[wd] & (~0ull << ((i + 1u) & 63u))) : 0ull;
        while (wd < 5u && mm == 0ull) {
          wd++;
          mm = wd < 5u ? S.starts[wd] : 0ull;
        }
        if (wd < 5u) nxt = min(ncl, wd * 64u + (uint32_t)__builtin_ctzll(mm));
      }
      const uint32_t n = nxt - i;
      const uint32_t cap = v ? 6u : 138u, per = v ? 2u : 1u;
      const uint32_t m = n % cap;
      myn[q] = n;
      mytok[q] = (n / cap) * per + (m >= 4u ? per : m);
    }
  }
  // exclusive scan of the token counts in index order (thread t holds indices t and t + 256)
  uint32_t pre[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    uint32_t incl = mytok[q];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if ((int)lane >= d) incl += t;
    }
    if (lane == 63) S.wsum[wave] = incl;
    __syncthreads();
    uint32_t base = S.nrl;  // tokens of the first 256 indices (0 in the first round)
    for (uint32_t w = 0; w < wave; w++) base += S.wsum[w];
    pre[q] = base + incl - mytok[q];
    __syncthreads();
    if (tid == HUFF_THREADS - 1) S.nrl = pre[q] + mytok[q];
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if (myn[q]) {
      const uint32_t v = S.cl[tid + q * HUFF_THREADS];
      const uint32_t cap = v ? 6u : 138u;
      uint32_t pos = pre[q], rem = myn[q];
      while (rem) {
        const uint32_t r = min(rem, cap);
        if (r >= 4u) {
          if (v == 0) {
            S.rl_sym[pos] = (r >= 11u) ? 18 : 17;
            S.rl_val[pos] = (uint8_t)r;
            pos++;
          } else {
            S.rl_sym[pos] = (uint8_t)v;
            S.rl_val[pos] = 1;
            pos++;
            S.rl_sym[pos] = 16;
            S.rl_val[pos] = (uint8_t)(r - 1u);
            pos++;
          }
        } else {
          for (uint32_t j = 0; j < r; j++) {
            S.rl_sym[pos] = (uint8_t)v;
            S.rl_val[pos] = 1;
            pos++;
          }
        }
        rem -= r;
      }
    }
  }
  __syncthreads();
  const uint32_t nrl = S.nrl;
  for (uint32_t i = tid; i < nrl; i += HUFF_THREADS) atomicAdd(&S.hist[S.rl_sym[i]], 1u);
  __syncthreads();
  USTAMP(4);
  pm_lengths(S, S.hist, 19, 7, S.clens);  // src/deflate.ts:141
  canon_codes(S, S.clens, 19, S.ccodes);
  USTAMP(5);

  // header bits (src/deflate.ts:143-181): fixed fields, the code-length code, then the coded run-length tokens,
  // each token ORed in at its bit offset (exclusive scan of the token bit counts)
  if (tid == 0) S.a_k[2] = 4;  // HCLEN: 1 + last index in transmission order whose symbol is used
  __syncthreads();
  if (tid < 19 && S.clens[kClOrder[tid]]) atomicMax(&S.a_k[2], tid + 1u);
  __syncthreads();
  const uint32_t HCLEN = S.a_k[2];
  if (tid == 0) atomicOr(&S.hdr[0], (HLIT - 257u) | ((HDIST - 1u) << 5) | (((HCLEN - 4u) & 15u) << 10));
  if (tid < HCLEN) {
    uint32_t bp = 14u + 3u * tid;
    hdr_or(S, bp, S.clens[kClOrder[tid]], 3);  // src/deflate.ts:158-165
  }
  uint32_t tv[2] = {0, 0}, tb[2] = {0, 0};
#pragma unroll
  for (int q = 0; q < 2; q++) {  // thread t holds tokens 2t and 2t + 1
    const uint32_t i = 2u * tid + (uint32_t)q;
    if (i < nrl) {
      const uint32_t sy = S.rl_sym[i], cl = S.clens[sy];
      const uint32_t xb = sy == 18u ? 7u : sy == 17u ? 3u : sy == 16u ? 2u : 0u;
      const uint32_t xv = sy == 18u ? S.rl_val[i] - 11u : (sy == 17u || sy == 16u) ? S.rl_val[i] - 3u : 0u;
      tv[q] = (uint32_t)S.ccodes[sy] | (xv << cl);
      tb[q] = cl + xb;
    }
  }
  {
    const uint32_t mine = tb[0] + tb[1];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if ((int)lane >= d) incl += t;
    }
    if (lane == 63) S.wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 14u + 3u * HCLEN, total = 0;
    for (uint32_t w = 0; w < HUFF_THREADS / 64; w++) {
      if (w < wave) base += S.wsum[w];
      total += S.wsum[w];
    }
    uint32_t bp = base + incl - mine;
    if (tb[0]) hdr_or(S, bp, tv[0], tb[0]);
    bp += tb[0];
    if (tb[1]) hdr_or(S, bp, tv[1], tb[1]);
    if (tid == 0) S.hdr_bits = 14u + 3u * HCLEN + total;
  }
  __syncthreads();
  USTAMP(6);
  // block bit count straight from the histograms
  uint32_t part = 0;
  for (uint32_t s = tid; s < 286; s += HUFF_THREADS) {
    uint32_t xb = (s >= 257) ? kLenXbits[s - 257] : 0u;
    part += hg[s] * ((uint32_t)S.lens[s] + xb);
  }
  if (tid < 30) part += hg[288 + tid] * ((uint32_t)S.lens[288 + tid] + kDistXbits[tid]);
  atomicAdd(&S.total_bits, part);
  __syncthreads();
  if (tid == 0) {
    blks[g].hdr_bits = S.hdr_bits;
    blks[g].bits = 3u + S.hdr_bits + S.total_bits;
  }
  uint32_t* cg = codes_out + (uint64_t)g * 320;
  for (uint32_t i = tid; i < 320; i += HUFF_THREADS) cg[i] = (uint32_t)S.codes[i] | ((uint32_t)S.lens[i] << 16);
  uint32_t* hd = hdr_out + (uint64_t)g * ZES_HDR_WORDS;
  for (uint32_t i = tid; i < ZES_HDR_WORDS; i += HUFF_THREADS) hd[i] = S.hdr[i];
  USTAMP(7);
}

// stage-level entry for tests: lengths only
__global__ __launch_bounds__(HUFF_THREADS) void k_huff_lengths_only(const uint32_t* __restrict__ hist, uint32_t nsym, uint32_t L,
                                                                    uint8_t* __restrict__ lens_out) {
  __shared__ HuffSmem S;
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 320; i += HUFF_THREADS) S.hist[i] = i < nsym ? hist[i] : 0u;
  __syncthreads();
  pm_lengths(S, S.hist, nsym, L, S.lens);
  for (uint32_t i = tid; i < nsym; i += HUFF_THREADS) lens_out[i] = S.lens[i];
}

// ------------------------------------------------------------------------------------------
// k_make_blks: the per-block records from the buffer table (a one-buffer call passes its buffer as a
// kernel argument: nothing is uploaded), and the Adler-32 accumulators cleared.
// ------------------------------------------------------------------------------------------
__global__ void k_make_blks(ZesBuf b0, uint32_t nbuf, ZesBuf* __restrict__ bufs, ZesBlk* __restrict__ blks, uint32_t nblk,
                            unsigned long long* __restrict__ adler) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (nbuf == 1u && i == 0) bufs[0] = b0;
  if (i < 2u * nbuf) adler[i] = 0ull;
  if (i >= nblk) return;
  uint32_t lo = 0, hi = nbuf;  // the last buffer whose first block is <= i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((nbuf == 1u ? b0.first_blk : bufs[mid].first_blk) <= i) lo = mid; else hi = mid;
  }
  const uint64_t n = nbuf == 1u ? b0.n : bufs[lo].n;
  const uint32_t fb = nbuf == 1u ? b0.first_blk : bufs[lo].first_blk;
  ZesBlk z;
  z.buf = lo;
  z.blk = i - fb;
  z.len = (uint32_t)min((uint64_t)ZES_BLK, n - (uint64_t)z.blk * ZES_BLK);
  z.ntok = 0;
  z.hdr_bits = 0;
  z.bits = 0;
  z.bit_off = 0;
  blks[i] = z;
}

// ------------------------------------------------------------------------------------------
// k_adler: each workgroup reduces one 64 KiB chunk to (A, B) and adds its closed-form share
//  to two u64 accumulators per buffer: acc[0] += A, acc[1] += B + A * bytesAfter.
// ------------------------------------------------------------------------------------------
// one chunk [c0, c0 + clen) of an n-byte buffer at p0
__device__ __forceinline__ static void adler_chunk(const uint8_t* __restrict__ p0, uint64_t n, uint64_t c0, uint64_t clen,
                                                   unsigned long long* __restrict__ acc) {
  __shared__ uint64_t sa[ADLER_THREADS / 64], sb[ADLER_THREADS / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint8_t* p = p0 + c0;
  // B = sum over j of (clen - j) * b[j]  (j = offset inside the chunk)
  uint64_t A = 0, B = 0;
  const bool aligned = (((uintptr_t)p) & 15u) == 0;
  uint64_t o_first = (uint64_t)tid * 16;
  // Whole aligned rounds first, four 16-byte loads in flight per thread and the byte sums by dot-product instructions
  // (the one-load-at-a-time loop below ran at 1.6 TB/s: a workgroup per 64 KiB leaves four waves on a SIMD, each
  // waiting for its single load).
  if (aligned) {
    constexpr uint64_t ROUND = (uint64_t)ADLER_THREADS * 16 * 4;
    const uint64_t nround = clen / ROUND;
    for (uint64_t r = 0; r < nround; r++) {
      uint4 v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4*>(p + r * ROUND + (uint64_t)q * ADLER_THREADS * 16 + (uint64_t)tid * 16);
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint64_t o = r * ROUND + (uint64_t)q * ADLER_THREADS * 16 + (uint64_t)tid * 16;
        uint32_t a = __builtin_amdgcn_udot4(v[q].x, 0x01010101u, 0u, false);
        a = __builtin_amdgcn_udot4(v[q].y, 0x01010101u, a, false);
        a = __builtin_amdgcn_udot4(v[q].z, 0x01010101u, a, false);
        a = __builtin_amdgcn_udot4(v[q].w, 0x01010101u, a, false);
        uint32_t w = __builtin_amdgcn_udot4(v[q].x, 0x0D0E0F10u, 0u, false);  // byte k of the 16 weighs 16 - k
        w = __builtin_amdgcn_udot4(v[q].y, 0x090A0B0Cu, w, false);
        w = __builtin_amdgcn_udot4(v[q].z, 0x05060708u, w, false);
        w = __builtin_amdgcn_udot4(v[q].w, 0x01020304u, w, false);
        A += a;
        B += (uint64_t)w + (uint64_t)a * (clen - o - 16);
      }
    }
    o_first += nround * ROUND;
  }
  for (uint64_t o = o_first; o < clen; o += (uint64_t)ADLER_THREADS * 16) {
    uint8_t b[16];
    if (aligned && o + 16 <= clen) {
      *reinterpret_cast<uint4*>(b) = *reinterpret_cast<const uint4*>(p + o);
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) b[k] = (o + k < clen) ? p[o + k] : (uint8_t)0;
    }
    uint32_t a = 0, w = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      a += b[k];
      w += (uint32_t)(16 - k) * b[k];
    }
    // (clen - (o+k)) = (clen - o - 16) + (16 - k); bytes past the chunk end are 0 so the
    // (possibly negative) first factor never multiplies a non-zero byte out of range
    A += a;
    B += (uint64_t)w + (uint64_t)a * (clen - o - 16);  // clen - o - 16 may wrap; see below
  }
  // note: when o + 16 > clen the term a * (clen - o - 16) wraps modulo 2^64, and so does the
  // over-count inside w for the padding zeros: together they are still exact modulo 2^64,
  // because (clen - o - 16) + (16 - k) = clen - o - k >= 1 for every real byte.
  for (int d = 32; d >= 1; d >>= 1) {
    A += __shfl_down(A, d);
    B += __shfl_down(B, d);
  }
  if (lane == 0) {
    sa[wave] = A;
    sb[wave] = B;
  }
  __syncthreads();
  if (tid == 0) {
    uint64_t ta = 0, tb = 0;
    for (uint32_t w = 0; w < ADLER_THREADS / 64; w++) {
      ta += sa[w];
      tb += sb[w];
    }
    const uint64_t after = n - c0 - clen;
    ta %= 65521u;
    tb %= 65521u;
    const uint64_t share = (tb + ta * (after % 65521u)) % 65521u;
    atomicAdd(&acc[0], (unsigned long long)ta);
    atomicAdd(&acc[1], (unsigned long long)share);
  }
}

__global__ __launch_bounds__(ADLER_THREADS) void k_adler(const uint8_t* __restrict__ d_in, uint64_t in_off, uint64_t n,
                                                         unsigned long long* __restrict__ acc) {
  const uint64_t c0 = (uint64_t)blockIdx.x * ADLER_CHUNK;
  adler_chunk(d_in + in_off, n, c0, min((uint64_t)ADLER_CHUNK, n - c0), acc);
}

// batch form: one workgroup per deflate block of every buffer (a block is one Adler chunk)
__global__ __launch_bounds__(ADLER_THREADS) void k_adler_blocks(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                                const ZesBlk* __restrict__ blks,
                                                                unsigned long long* __restrict__ acc) {
  const ZesBlk bk = blks[blockIdx.x];
  const ZesBuf bf = bufs[bk.buf];
  adler_chunk(d_in + bf.in_off, bf.n, (uint64_t)bk.blk * ZES_BLK, bk.len, acc + 2 * (size_t)bk.buf);
}

// ------------------------------------------------------------------------------------------
// k_layout: one workgroup per buffer.  Exclusive scan of the block bit counts (blocks are
// bit-concatenated, src/deflate.ts:20-34), final zero pad (:35-37), zlib header and Adler-32
// trailer (src/zlib.ts:28-46).  Boundary dwords of every block are zeroed here because k_emit
// ORs into them atomically.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_layout(uint8_t* __restrict__ d_out, const ZesBuf* __restrict__ bufs,
                                                ZesBlk* __restrict__ blks, const unsigned long long* __restrict__ adler_acc,
                                                ZesRes* __restrict__ res) {
  __shared__ uint64_t s_tot;
  __shared__ uint64_t s_part[256];
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const ZesBuf bf = bufs[b];
  ZesBlk* bk = blks + bf.first_blk;
  uint32_t* out32 = reinterpret_cast<uint32_t*>(d_out + bf.out_off);
  {
    // exclusive scan over the buffer's blocks: contiguous slice per thread, then a serial
    // pass over the 256 slice sums (nblk is a few thousand at most)
    const uint32_t per = (bf.nblk + 255u) / 256u;
    const uint32_t lo = min(bf.nblk, tid * per), hi = min(bf.nblk, lo + per);
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += bk[i].bits;
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint64_t off = (bf.flags & ZES_BUF_RANGE) ? bf.start_bit : 16;  // after the two zlib header bytes
      for (uint32_t t = 0; t < 256; t++) {
        const uint64_t v = s_part[t];
        s_part[t] = off;
        off += v;
      }
      s_tot = off;
    }
    __syncthreads();
    uint64_t off = s_part[tid];
    for (uint32_t i = lo; i < hi; i++) {
      bk[i].bit_off = off;
      off += bk[i].bits;
    }
  }
  __syncthreads();
  const uint64_t tot = s_tot;
  const bool range = (bf.flags & ZES_BUF_RANGE) != 0;
  const uint64_t raw_end = (tot + 7) >> 3;      // byte index just past the padded deflate data
  const uint64_t out_len = raw_end + (range ? 0 : 4);
  // zero every dword touched atomically: block boundary dwords and everything from the last
  // block's final dword through the trailer
  for (uint32_t i = tid; i < bf.nblk; i += 256) {
    const uint64_t s = bk[i].bit_off, e = s + bk[i].bits - 1;
    // (a range that continues another one's stream inside a dword leaves that dword's bits alone)
    if (!(i == 0 && (bf.flags & ZES_BUF_CONT) && (s & 31u))) out32[s >> 5] = 0;
    out32[e >> 5] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    const uint64_t last_dw = (tot - 1) >> 5;
    const uint64_t end_dw = (out_len * 8 - 1) >> 5;
    for (uint64_t w = last_dw + 1; w <= end_dw; w++) out32[w] = 0;
    __threadfence();
    const uint64_t n = bf.n;
    const uint32_t s1 = (uint32_t)((1ull + adler_acc[2 * b + 0]) % 65521ull);
    const uint32_t s2 = (uint32_t)((n % 65521ull + adler_acc[2 * b + 1]) % 65521ull);
    const uint32_t ad = (s2 << 16) | s1;
    if (!range) {
      // zlib header: 0x78 0x9C (src/zlib.ts:29-34)
      atomicOr(&out32[0], 0x9C78u);
      for (int k = 0; k < 4; k++) {  // big-endian trailer (src/zlib.ts:37-40)
        const uint64_t pos = raw_end + k;
        const uint32_t byte = (ad >> (24 - 8 * k)) & 0xffu;
        atomicOr(&out32[pos >> 2], byte << (8 * (pos & 3)));
      }
    }
    res[b].out_len = range ? tot - bf.start_bit : out_len;  // a range reports bits: its seam with the next range is a bit position
    res[b].status = 0;
    res[b].aux = ad;
  }
}

// ------------------------------------------------------------------------------------------
// k_emit: one workgroup per block.  Items (3 block bits, header words, tokens, EOB) get their
// bit offsets from a workgroup scan; bits are OR-ed into an LDS staging window and leave as
// whole dwords.  Only the first and last dword of a block (shared with its neighbours) use
// global atomics.
// ------------------------------------------------------------------------------------------
#ifndef EMIT_ITEMS
#define EMIT_ITEMS 4  // tokens per thread per tile (a multiple of 4: 16-byte loads)
#endif
#define EMIT_TILE (EMIT_THREADS * EMIT_ITEMS)
#define EMIT_STAGE_WORDS (EMIT_TILE * 48 / 32 + 8)
struct EmitSmem {
  uint32_t codes[320];
  uint32_t stage[EMIT_STAGE_WORDS];
  uint32_t wsum[EMIT_THREADS / 64];
  uint32_t carry;
};

__device__ static inline void stage_or(uint32_t* stage, uint32_t rel_bit, uint64_t v, uint32_t nbits) {
  const uint32_t w = rel_bit >> 5, sh = rel_bit & 31u;
  const uint64_t lo = v << sh;
  atomicOr(&stage[w], (uint32_t)lo);
  if (sh + nbits > 32u) atomicOr(&stage[w + 1], (uint32_t)(lo >> 32));
  if (sh + nbits > 64u) atomicOr(&stage[w + 2], (uint32_t)(v >> (64u - sh)));
}

// Workgroup barrier for LDS traffic only: __syncthreads() also waits for every global load and store of the thread
// (vmcnt(0)), which here would hold each tile until its flush stores and the next tile's token loads have landed.
__device__ __forceinline__ static void emit_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(EMIT_THREADS, 8) void k_emit(uint8_t* __restrict__ d_out, const ZesBuf* __restrict__ bufs,
                                                       const ZesBlk* __restrict__ blks, const uint32_t* __restrict__ tok_in,
                                                       const uint32_t* __restrict__ codes_in, const uint32_t* __restrict__ hdr_in) {
  __shared__ EmitSmem S;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const ZesBlk bk = blks[g];
  const ZesBuf bf = bufs[bk.buf];
  uint32_t* out32 = reinterpret_cast<uint32_t*>(d_out + bf.out_off);
  const uint32_t* tk = tok_in + (uint64_t)g * ZES_BLK;
  const uint32_t* hd = hdr_in + (uint64_t)g * ZES_HDR_WORDS;
  for (uint32_t i = tid; i < 320; i += EMIT_THREADS) S.codes[i] = codes_in[(uint64_t)g * 320 + i];
  if (tid == 0) S.carry = 0;
  const uint64_t blk_first_dw = bk.bit_off >> 5;
  const uint64_t blk_last_dw = (bk.bit_off + bk.bits - 1) >> 5;
  const bool is_final = (bk.blk + 1 == bf.nblk) && !(bf.flags & ZES_BUF_NOTFINAL);
  const uint32_t nhdr_items = 1 + ((bk.hdr_bits + 31u) >> 5);  // block bits + header words
  // item space: the first tile holds only the header items, token i is item EMIT_TILE + i, so that a
  // thread's four tokens are one aligned 16-byte load (issued unconditionally: no wait under a branch)
  const uint32_t nitems = EMIT_TILE + bk.ntok + 1;             // + EOB
  static_assert(1 + ZES_HDR_WORDS <= EMIT_TILE, "header items fit the first tile");
  uint64_t cur_bit = bk.bit_off;                               // uniform
  __syncthreads();

  // tokens of the tile in hand (the first tile holds header items only; tile 1 starts at token 0)
  uint4 pre[EMIT_ITEMS / 4];
#pragma unroll
  for (int k4 = 0; k4 < EMIT_ITEMS / 4; k4++) pre[k4] = make_uint4(0, 0, 0, 0);
  for (uint32_t t0 = 0; t0 < nitems; t0 += EMIT_TILE) {
    // 1. each thread builds its (value, nbits) items
    uint64_t val[EMIT_ITEMS];
    uint32_t nb[EMIT_ITEMS];
    uint32_t mysum = 0;
    uint32_t tv4[EMIT_ITEMS];
#pragma unroll
    for (int k4 = 0; k4 < EMIT_ITEMS / 4; k4++) {
      tv4[4 * k4 + 0] = pre[k4].x;
      tv4[4 * k4 + 1] = pre[k4].y;
      tv4[4 * k4 + 2] = pre[k4].z;
      tv4[4 * k4 + 3] = pre[k4].w;
    }
    {  // the next tile's tokens: on their way while this tile is worked on (the barriers below do not wait for them)
      const uint32_t ti0 = t0 + tid * EMIT_ITEMS;  // first token of this thread in tile t0 + EMIT_TILE
      const uint4* qp = reinterpret_cast<const uint4*>(tk + min(ti0, ZES_BLK - EMIT_ITEMS));
#pragma unroll
      for (int k4 = 0; k4 < EMIT_ITEMS / 4; k4++) pre[k4] = qp[k4];
    }
#pragma unroll
    for (int k = 0; k < EMIT_ITEMS; k++) {
      const uint32_t it = t0 + tid * EMIT_ITEMS + k;
      uint64_t v = 0;
      uint32_t b = 0;
      if (it < EMIT_TILE) {
        if (it == 0) {  // BFINAL + BTYPE=2 (src/deflate.ts:21-28)
          v = (is_final ? 1u : 0u) | (2u << 1);
          b = 3;
        } else if (it < nhdr_items) {
          const uint32_t wi = it - 1;
          v = hd[wi];
          b = min(32u, bk.hdr_bits - wi * 32u);
        }
      } else if (it < nitems) {
        if (it == nitems - 1) {  // EOB (src/deflate.ts:222-226)
          const uint32_t c = S.codes[256];
          v = c & 0xffffu;
          b = c >> 16;
        } else {
          const uint32_t tv = tv4[k];
          if (tv & ZES_TOK_MATCH) {  // src/deflate.ts:187-211
            const uint32_t len = zes_tok_len(tv), dist = zes_tok_dist(tv);
            const uint32_t lc = zes_len_code(len), dc = zes_dist_code(dist);
            const uint32_t cl = S.codes[257 + lc], cd = S.codes[288 + dc];
            const uint32_t ll = cl >> 16, dl = cd >> 16;
            const uint32_t lx = kLenXbits[lc], dx = kDistXbits[dc];
            v = (uint64_t)(cl & 0xffffu);
            b = ll;
            v |= (uint64_t)(len - kLenBase[lc]) << b;
            b += lx;
            v |= (uint64_t)(cd & 0xffffu) << b;
            b += dl;
            v |= (uint64_t)(dist - kDistBase[dc]) << b;
            b += dx;
          } else {  // literal (src/deflate.ts:212-219)
            const uint32_t c = S.codes[tv];
            v = c & 0xffffu;
            b = c >> 16;
          }
        }
      }
      val[k] = v;
      nb[k] = b;
      mysum += b;
    }
    // 2. workgroup exclusive scan of mysum
    uint32_t incl = mysum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if ((int)lane >= d) incl += t;
    }
    if (lane == 63) S.wsum[wave] = incl;
    emit_lds_barrier();  // also: previous tile's flush finished reading stage/carry
    uint32_t wbase = 0, tile_bits = 0;
#pragma unroll
    for (uint32_t w = 0; w < EMIT_THREADS / 64; w++) {
      const uint32_t s = S.wsum[w];
      if (w < wave) wbase += s;
      tile_bits += s;
    }
    uint32_t rel = wbase + incl - mysum;  // bit offset of my first item inside the tile
    // 3. zero the staging window, seed it with the carried partial dword
    const uint32_t head = (uint32_t)(cur_bit & 31u);
    const uint32_t nwords = (head + tile_bits + 31u) >> 5;
    for (uint32_t i = tid; i < nwords + 1; i += EMIT_THREADS) S.stage[i] = (i == 0) ? S.carry : 0u;
    emit_lds_barrier();
    // 4. OR the items in.  A thread's items are consecutive in the stream: they are gathered into 64-bit pieces first
    // (seven 9-bit literals make one), so that the staging window sees two or three LDS atomics per piece instead of
    // one or two per item — the atomics, with their bank conflicts, are what this step costs.
    rel += head;
    {
      uint64_t acc = 0;
      uint32_t accn = 0, accpos = rel;
#pragma unroll
      for (int k = 0; k < EMIT_ITEMS; k++) {
        if (accn + nb[k] > 64u) {
          stage_or(S.stage, accpos, acc, accn);
          accpos += accn;
          acc = 0;
          accn = 0;
        }
        acc |= val[k] << accn;  // (an item has at most 48 bits; accn + nb[k] <= 64 here)
        accn += nb[k];
      }
      if (accn) stage_or(S.stage, accpos, acc, accn);
    }
    emit_lds_barrier();
    // 5. flush complete dwords; the trailing partial dword is carried into the next tile
    const uint64_t first_dw = cur_bit >> 5;
    const uint32_t end_bits = head + tile_bits;
    const bool last_tile = (t0 + EMIT_TILE >= nitems);
    const uint32_t ncomplete = last_tile ? nwords : (end_bits >> 5);
    for (uint32_t i = tid; i < ncomplete; i += EMIT_THREADS) {
      const uint64_t gw = first_dw + i;
      const uint32_t w = S.stage[i];
      if (gw == blk_first_dw || gw == blk_last_dw) atomicOr(&out32[gw], w);
      else out32[gw] = w;
    }
    const uint32_t carry_next = (!last_tile && (end_bits & 31u)) ? S.stage[end_bits >> 5] : 0u;
    emit_lds_barrier();
    if (tid == 0) S.carry = carry_next;
    cur_bit += tile_bits;
  }
}

// ------------------------------------------------------------------------------------------
// k_bits_place: ORs an nbits-long bit stream (from bit 0 of src) into dst at bit position pos.  The destination
// was zeroed; interior dwords belong to this piece alone and are stored, the first and the last one are shared
// with the neighbouring pieces (blocks are bit-concatenated, src/deflate.ts:20-37) and are OR-ed atomically.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bits_place(uint32_t* __restrict__ dst, uint64_t pos, const uint32_t* __restrict__ src, uint64_t nbits) {
  if (!nbits) return;
  const uint64_t first = pos >> 5, last = (pos + nbits - 1) >> 5;
  const uint32_t sh = (uint32_t)(pos & 31u);
  const uint64_t nsrc = (nbits + 31) >> 5;  // source dwords holding bits
  for (uint64_t w = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= last; w += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t k = w - first;  // dst dword k takes src bits [32k - sh, 32k - sh + 32)
    uint32_t lo = 0, hi = 0;
    if (k < nsrc) hi = src[k];
    if (k >= 1 && k - 1 < nsrc) lo = src[k - 1];
    uint32_t v = sh ? ((hi << sh) | (lo >> (32u - sh))) : hi;
    // bits of the piece's last source dword beyond nbits are not part of the stream
    const uint64_t bit_lo = w << 5;  // absolute position of this dword's bit 0
    if (bit_lo + 32 > pos + nbits) v &= (uint32_t)((1ull << (pos + nbits - bit_lo)) - 1ull);
    if (bit_lo < pos) v &= ~(uint32_t)((1ull << (pos - bit_lo)) - 1ull);
    if (w == first || w == last) atomicOr(&dst[w], v);
    else dst[w] = v;
  }
}

// small utility kernels ---------------------------------------------------------------------
// The hardware property k_lz_sort's ranks rest on (see its scatter passes): lanes of one wavefront whose returning LDS add
// meets in one word are handed their old values in ascending lane order, and a wave's LDS instructions execute in program
// order.  Sixteen wavefronts, a counter row each (two 16-bit counters per word, as the sort packs them), digit patterns from
// uniform to a single digit, four adds back to back; every value is checked against the rank found with ballots.
// out[0] += values out of order, out[1] += values checked.
__global__ __launch_bounds__(SORT_THREADS) void k_selftest_lds_order(unsigned long long* __restrict__ out, uint32_t iters, uint32_t seed) {
  __shared__ uint32_t row[SORT_WAVES][128];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  unsigned long long bad = 0, n = 0;
  for (uint32_t it = 0; it < iters; it++) {
    row[wave][lane] = 0;
    row[wave][lane + 64u] = 0;
    const uint32_t pat = (it + wave) % 6u;
    uint32_t h = seed ^ (blockIdx.x * 0x9E3779B1u) ^ (it * SORT_THREADS + tid);
    h = (h ^ (h >> 16)) * 0x7feb352du;
    h = (h ^ (h >> 15)) * 0x846ca68bu;
    h ^= h >> 16;
    uint32_t d;
    if (pat == 0) d = h & 255u;
    else if (pat == 1) d = h & 3u;
    else if (pat == 2) d = 77u + (lane >> 6);  // one digit (not a compile-time uniform address)
    else if (pat == 3) d = 32u + ((h >> 3) % 27u);
    else if (pat == 4) d = (h & 1u) + 2u * ((h >> 8) & 7u);  // pairs that share a word
    else d = (lane * 4u + (h & 3u)) & 255u;
    const bool valid = pat != 3u || (h >> 20) % 9u != 0u;
    uint32_t dd[4], rk[4];
#pragma unroll
    for (int r = 0; r < 4; r++) dd[r] = (d + (pat == 0u ? 17u * r : 0u)) & 255u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t sh = 16u * (dd[r] & 1u);
      const uint32_t old = valid ? atomicAdd(&row[wave][dd[r] >> 1], 1u << sh) : 0u;
      rk[r] = (old >> sh) & 0xffffu;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint32_t expect = 0;
      for (int q = 0; q <= r; q++) {  // lanes holding my round-r digit in round q: all of them for q < r, the lower ones for q == r
        uint64_t m = __ballot(valid);
        for (int b = 0; b < 8; b++) {
          const bool bit = (dd[r] >> b) & 1u;
          const uint64_t bal = __ballot((dd[q] >> b) & 1u);
          m &= bit ? bal : ~bal;
        }
        expect += (uint32_t)__popcll(q < r ? m : (m & zes_lanemask_lt()));
      }
      if (valid) {
        n++;
        bad += rk[r] != expect;
      }
    }
  }
  atomicAdd(&out[0], bad);
  atomicAdd(&out[1], n);
}

__global__ void k_zero_u64(unsigned long long* p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0ull;
}
