// zes_index.hip — k_lz_index: the LZ77 index of a *dense* block (text, periodic data: nearly every position shares
// its 3-byte key with another one) built inside LDS.  Replaces, for those blocks, the three global scatter passes of
// k_lz_sort and its four re-reads of the sorted list (src/lz77.ts:11-22: generateLZ77IndexMap).
//
// What the lazy match finder needs from the index (k_lz_match_lazy):
//   sd[r]   u16 per slot: distance from the position in slot r to the one in slot r-1 when both hold the same key and
//           lie within 32768 of each other, else 0 — the candidates of a position, most recent first (src/lz77.ts:65),
//           are consecutive entries;
//   inv[p]  u32 per position: the slot of p, or ZES_INV_NONE when p has no candidate.
// Nothing asks for the slots to be ordered BY key: positions of one key must be neighbours, in ascending order.  So:
//
//   1  every position's key is hashed by a bijection of the 24-bit key space, H = key * odd mod 2^24; the top 4 bits
//      are its *group* (16), the top 11 its *class* (2048 classes, ~64 positions each), the other 13 bits travel with
//      the position in one word: equal upper parts <=> equal keys inside a class;
//   2  sweep 1 over the block (read straight from memory, 16 bytes per lane): class sizes by LDS atomics; a scan gives
//      every class its run of slots.  A block with a class above IDX_BIGCAP words, a group above IDX_GCAP, or a
//      sixteenth of its positions in classes above IDX_REGCAP words (heavy keys: text) goes back to k_lz_sort here
//      (ZES_SORT_REDO: k_lz_sort is the faster one on such data);
//   3  sweep 2, tile by tile (16384 positions): each position's word (class-in-group 7 | rest of H 13 | low 12 bits of the
//      position) is binned in an LDS tile buffer by (group, 4096-byte sub-tile), and every group's bin leaves for the
//      group's part of a global array E as ONE contiguous run; a table of run starts per (group, sub-tile) gives the
//      upper five bits of the positions back;
//   4  group by group: the group's words come back in bulk, each goes to its class's run of a second LDS buffer
//      (cursors from the class sizes), and every class is sorted by ONE wavefront in registers, by the whole word (key
//      part, then position): a bitonic network over 64 M words (M = 1, 2, 4, 8; DPP quad_perm / row_ror, ds_swizzle,
//      ds_bpermute, and register against register above 64), the classes dealt by size category;
//   5  a sorted class gives its sd[] entries (coalesced 2-byte stores) and, per position, the word
//      position-in-slice | slot | has-a-candidate, appended to the bucket of the position's 16384-byte slice (8 LDS
//      cursors, the buckets in global memory);
//   6  classes above IDX_REGCAP words (few, when the block was kept) go back to E and are sorted at the end by radix
//      passes of 6 bits in LDS, four wavefronts with a quarter of the LDS each;
//   7  slice by slice the buckets are read back (all loads of a slice in flight), scattered into an LDS image of the
//      slice and written out as inv[] with coalesced 16-byte stores.
//
// Global traffic per position: 1 (input) + 8 (E) + 8 (buckets) + 2 (sd) + 4 (inv) = 23 bytes, all of it in whole
// lines (k_lz_sort: 51).
#include "zes_common.h"
#include "zes_kernels.h"

#define IDX_WAVES (IDX_THREADS / 64)
#define IDX_CB 11u
#define IDX_NCLASS (1u << IDX_CB)
#define IDX_RSHIFT (24u - IDX_CB)
#define IDX_MUL 0x9E3779u  // odd: key -> key * IDX_MUL mod 2^24 is a bijection
#define IDX_BIGCAP 4096u   // words of the largest class taken (the heavy classes' round: four wavefronts, two buffers of this size each)
#define IDX_NBIGW 4u
#define IDX_SLICE 16384u
#define IDX_NSLICE (ZES_BLK / IDX_SLICE)
#define IDX_MAXBIG 256u  // classes above IDX_REGCAP words: at most 131070 / 513
#define IDX_REGCAP 512u  // words a wavefront sorts in registers
#define IDX_NGROUP 16u   // groups of classes: the top four bits of H
#define IDX_GCLASS (IDX_NCLASS / IDX_NGROUP)
#define IDX_GCAP 16384u  // words of a group the LDS holds
#define IDX_TILE 16384u  // positions per tile of sweep 2 (16 per thread)
#define IDX_SUB 4096u    // sub-tile: the low 12 bits of a position travel in the word
#define IDX_NRUN (ZES_BLK / IDX_SUB)

struct IndexSmem {
  uint32_t scr[2 * IDX_GCAP];  // sweep 1: class counters in [0, 2048 + 64); sweep 2: the tile buffer; groups: [IDX_GCAP, ...) the group's words; heavy classes: scratch; last: a slice of inv
  uint32_t cnt[IDX_NBIGW][5 * 64];         // digit counters of the radix passes (heavy classes), per wavefront
  uint32_t base[IDX_NCLASS + 1];           // first slot of each class
  uint32_t big[IDX_MAXBIG];
  uint32_t runs[IDX_NGROUP][IDX_NRUN + 1];  // group g's words of sub-tile r start runs[g][r] words into the group's part of E
  uint32_t gto[IDX_NGROUP + 1];             // sweep 2, per tile: where group g's words start in the tile buffer
  uint32_t bcnt[IDX_NGROUP * 4], bpos[IDX_NGROUP * 4];
  uint32_t gfill[IDX_NGROUP];
  uint32_t ccur[IDX_GCLASS];                // group stage: cursors of the group's classes
  uint32_t catcnt[2][5], catpos[2][5];  // two sets, used in turn by the groups: a set is cleared while the other one counts
  uint8_t order[IDX_GCLASS];
  uint32_t wsum[IDX_WAVES];
  uint32_t pcur[IDX_NSLICE];
  uint32_t nbig, maxc, nheavy;
};
static_assert(sizeof(IndexSmem) <= 160 * 1024, "LDS");

__device__ __forceinline__ static uint32_t idx_ld_sc1(const uint32_t* p) {  // past this CU's L1: written by another wave of this workgroup
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 20 bytes at offset `off` of the block (zero past its end)
__device__ __forceinline__ static void idx_ld20(const uint8_t* __restrict__ src, bool aligned, uint32_t off, uint32_t T, uint32_t w[5]) {
  if (aligned && off + 20u <= T) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + off);
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
    w[4] = *reinterpret_cast<const uint32_t*>(src + off + 16u);
    return;
  }
#pragma unroll
  for (uint32_t k = 0; k < 5; k++) w[k] = 0;
  for (uint32_t k = 0; k < 20u; k++)
    if (off + k < T) w[k >> 2] |= (uint32_t)src[off + k] << (8u * (k & 3u));
}

template <int CTRL, int BANK>
__device__ __forceinline__ static uint32_t idx_dpp(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, 0xf, BANK, false);
}
// value of lane (lane ^ D)
template <int D>
__device__ __forceinline__ static uint32_t idx_xor_lane(uint32_t x) {
  if (D == 1) return idx_dpp<0xB1, 0xf>(x, x);  // quad_perm [1,0,3,2]
  if (D == 2) return idx_dpp<0x4E, 0xf>(x, x);  // quad_perm [2,3,0,1]
  if (D == 4) {                                  // row_ror:4 for lanes 4-7, 12-15; row_ror:12 for lanes 0-3, 8-11
    const uint32_t t = idx_dpp<0x124, 0xA>(x, x);
    return idx_dpp<0x12C, 0x5>(t, x);
  }
  if (D == 8) return idx_dpp<0x128, 0xf>(x, x);                        // row_ror:8
  if (D == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F);  // bit mode: xor 0x10
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((zes_lane() ^ 32u) << 2), (int)x);
}
template <int K, int D>
__device__ __forceinline__ static uint32_t idx_cmpx(uint32_t x, uint32_t lane) {
  const uint32_t y = idx_xor_lane<D>(x);
  // the lower lane of a pair keeps the smaller word where the run of K lanes is sorted upwards (K = 64: all of them)
  const bool keep_min = ((lane & (uint32_t)D) == 0u) == ((lane & (uint32_t)K) == 0u);
  return keep_min ? min(x, y) : max(x, y);
}
// ascending over the 64 lanes
__device__ __forceinline__ static uint32_t idx_bitonic64(uint32_t x) {
  const uint32_t lane = zes_lane();
  x = idx_cmpx<2, 1>(x, lane);
  x = idx_cmpx<4, 2>(x, lane);
  x = idx_cmpx<4, 1>(x, lane);
  x = idx_cmpx<8, 4>(x, lane);
  x = idx_cmpx<8, 2>(x, lane);
  x = idx_cmpx<8, 1>(x, lane);
  x = idx_cmpx<16, 8>(x, lane);
  x = idx_cmpx<16, 4>(x, lane);
  x = idx_cmpx<16, 2>(x, lane);
  x = idx_cmpx<16, 1>(x, lane);
  x = idx_cmpx<32, 16>(x, lane);
  x = idx_cmpx<32, 8>(x, lane);
  x = idx_cmpx<32, 4>(x, lane);
  x = idx_cmpx<32, 2>(x, lane);
  x = idx_cmpx<32, 1>(x, lane);
  x = idx_cmpx<64, 32>(x, lane);
  x = idx_cmpx<64, 16>(x, lane);
  x = idx_cmpx<64, 8>(x, lane);
  x = idx_cmpx<64, 4>(x, lane);
  x = idx_cmpx<64, 2>(x, lane);
  x = idx_cmpx<64, 1>(x, lane);
  return x;
}

__device__ __forceinline__ static uint32_t idx_xor_lane_d(uint32_t x, int d) {  // (d is a constant wherever this is called: unrolled loops)
  switch (d) {
    case 1: return idx_xor_lane<1>(x);
    case 2: return idx_xor_lane<2>(x);
    case 4: return idx_xor_lane<4>(x);
    case 8: return idx_xor_lane<8>(x);
    case 16: return idx_xor_lane<16>(x);
    default: return idx_xor_lane<32>(x);
  }
}
// 64 * M words, word e = j * 64 + lane in register j of its lane: ascending in e.  Steps over a distance of 64 or more
// are compare-exchanges between two registers of a lane; the others trade with another lane of the same register.
template <int M>
__device__ __forceinline__ static void idx_bitonic(uint32_t (&x)[M]) {
  const uint32_t lane = zes_lane();
#pragma unroll
  for (int k = 2; k <= 64 * M; k <<= 1) {
#pragma unroll
    for (int d = k >> 1; d >= 1; d >>= 1) {
      if (d >= 64) {
        const int dj = d >> 6;
#pragma unroll
        for (int j = 0; j < M; j++) {
          if (j & dj) continue;
          const bool asc = (j & (k >> 6)) == 0;  // (k = 64 M: every run goes upwards)
          const uint32_t lo = min(x[j], x[j | dj]), hi = max(x[j], x[j | dj]);
          x[j] = asc ? lo : hi;
          x[j | dj] = asc ? hi : lo;
        }
      } else {
#pragma unroll
        for (int j = 0; j < M; j++) {
          const uint32_t y = idx_xor_lane_d(x[j], d);
          const bool asc = k < 64 ? ((lane & (uint32_t)k) == 0u) : ((j & (k >> 6)) == 0);
          const bool keep_min = ((lane & (uint32_t)d) == 0u) == asc;
          x[j] = keep_min ? min(x[j], y) : max(x[j], y);
        }
      }
    }
  }
}

// what a sorted class leaves behind, for word x in slot `slot` whose predecessor in the class is `prev` (any word with
// another key part for the first one)
__device__ __forceinline__ static void idx_emit(IndexSmem& S, bool valid, uint32_t x, uint32_t prev, uint32_t slot, uint16_t* __restrict__ sd,
                                                uint32_t* __restrict__ P) {
  const uint32_t pos = x & 0x1FFFFu;
  const uint32_t delta = pos - (prev & 0x1FFFFu);  // > 0: positions ascend inside a key
  const bool has = ((x ^ prev) >> 17) == 0u && delta <= ZES_WINDOW;  // src/lz77.ts:49
  if (valid) {
    sd[slot] = (uint16_t)(has ? delta : 0u);
    const uint32_t sl = pos / IDX_SLICE;
    const uint32_t k = atomicAdd(&S.pcur[sl], 1u);
    P[sl * IDX_SLICE + k] = (pos & (IDX_SLICE - 1u)) | (slot << 14) | (has ? 0x80000000u : 0u);
  }
}

// a class of at most 64 * M words, sorted in registers
template <int M>
__device__ __forceinline__ static void idx_class_regs(IndexSmem& S, uint32_t b0, uint32_t n, const uint32_t* E /* LDS: the class's words */, uint16_t* __restrict__ sd,
                                                      uint32_t* __restrict__ P) {
  const uint32_t lane = zes_lane();
  uint32_t x[M];
#pragma unroll
  for (int j = 0; j < M; j++) x[j] = (uint32_t)j * 64u + lane < n ? E[(uint32_t)j * 64u + lane] : 0xFFFFFFFFu;
  idx_bitonic<M>(x);
#pragma unroll
  for (int j = 0; j < M; j++) {
    // the word in front: the lane below, lane 0 from the last lane of the register before (the class's first word: another key part)
    const uint32_t carry = j ? (uint32_t)__builtin_amdgcn_readlane((int)x[j ? j - 1 : 0], 63) : ~x[0];
    const uint32_t prev = idx_dpp<0x138, 0xf>(carry, x[j]);  // wave_shr:1
    idx_emit(S, (uint32_t)j * 64u + lane < n, x[j], prev, b0 + (uint32_t)j * 64u + lane, sd, P);
  }
}

// a heavy class (above IDX_REGCAP words), by one wavefront: radix passes of 6 bits in LDS.  scratch: 2 * cap words; cntw: 5 * 64 counters
__device__ static void idx_class_heavy(IndexSmem& S, uint32_t b0, uint32_t n, uint32_t* scratch, uint32_t cap, uint32_t* cntw,
                                 const uint32_t* __restrict__ E, uint16_t* __restrict__ sd, uint32_t* __restrict__ P) {
  const uint32_t lane = zes_lane();
  uint32_t* src = scratch;
  uint32_t* dst = scratch + cap;
#pragma unroll
  for (uint32_t d = 0; d < 5; d++) cntw[d * 64u + lane] = 0;
  // load, digit histograms (the multiset of a digit's values does not change from pass to pass), which bits differ at all
  uint32_t diff = 0;
  const uint32_t x0 = idx_ld_sc1(E + b0);
  for (uint32_t i = lane; i < n; i += 64u) {
    const uint32_t x = idx_ld_sc1(E + b0 + i);
    src[i] = x;
    diff |= x ^ x0;
#pragma unroll
    for (uint32_t d = 0; d < 5; d++) atomicAdd(&cntw[d * 64u + ((x >> (6u * d)) & 63u)], 1u);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, o);
  diff = (uint32_t)__builtin_amdgcn_readfirstlane((int)diff);
  for (uint32_t d = 0; d < 5; d++) {
    if (((diff >> (6u * d)) & 63u) == 0u) continue;  // every word has the same digit here
    uint32_t* cw = cntw + d * 64u;
    {  // exclusive scan of the digit counts
      const uint32_t c = cw[lane];
      uint32_t in = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)in, o);
        if ((int)lane >= o) in += t;
      }
      cw[lane] = in - c;
    }
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
      const uint32_t i = i0 + lane;
      const bool valid = i < n;
      const uint32_t x = src[valid ? i : 0u];
      const uint32_t dg = (x >> (6u * d)) & 63u;
      // lanes holding the same digit (stable rank = lower lanes first)
      uint64_t m = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 6; b++) {
        const bool bit = (dg >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const uint32_t rank = (uint32_t)__popcll(m & zes_lanemask_lt());
      uint32_t old = 0;
      if (valid && rank == 0u) old = atomicAdd(&cw[dg], (uint32_t)__popcll(m));
      const uint32_t first = valid ? (uint32_t)__builtin_ctzll(m) : 0u;
      const uint32_t at = (uint32_t)__shfl((int)old, (int)first) + rank;
      if (valid) dst[at] = x;
    }
    uint32_t* t = src;
    src = dst;
    dst = t;
  }
  for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
    const uint32_t i = i0 + lane;
    const bool valid = i < n;
    const uint32_t x = src[valid ? i : 0u];
    const uint32_t prev = (valid && i) ? src[i - 1u] : ~x;
    idx_emit(S, valid, x, prev, b0 + i, sd, P);
  }
}

__global__ __launch_bounds__(IDX_THREADS) void k_lz_index(const uint8_t* __restrict__ d_in, const ZesBuf* __restrict__ bufs,
                                                          const ZesBlk* __restrict__ blks, uint32_t* idx_a,
                                                          uint32_t* __restrict__ idx_b, uint32_t* inv_all, uint16_t* __restrict__ sd_all) {
  __shared__ __align__(16) IndexSmem S;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef IDX_PROF
  unsigned long long tp[8], tpart = 0, tsort = 0, twait = 0;
  int np = 0;
#define ISTAMP() tp[np++] = clock64()
#else
#define ISTAMP()
#endif
  uint32_t* A = idx_a + (uint64_t)g * ZES_BLK;  // [ZES_BLK-1]: the flag word; the rest: the slice buckets
  if (!(A[ZES_BLK - 1] & ZES_SORT_INDEX)) return;  // not a block k_lz_sort left to this kernel
  uint32_t* E = idx_b + (uint64_t)g * ZES_BLK;
  uint32_t* inv = inv_all + (uint64_t)g * ZES_BLK;
  uint16_t* sd = sd_all + (uint64_t)g * ZES_BLK;
  const ZesBlk bk = blks[g];
  const ZesBuf bf = bufs[bk.buf];
  const uint32_t T = bk.len;
  const uint8_t* src = d_in + bf.in_off + (uint64_t)bk.blk * ZES_BLK;
  const uint32_t cnt = T - 2u;  // (a block of this kernel has positions: k_lz_sort saw them)
  const bool aligned = (((uintptr_t)src) & 15u) == 0;
  uint32_t* cur = S.scr;  // [IDX_NCLASS] + 64 words that take the atomics of lanes without a position

  for (uint32_t i = tid; i < IDX_NCLASS + 64u; i += IDX_THREADS) cur[i] = 0;
  if (tid < IDX_NSLICE) S.pcur[tid] = 0;
  if (tid < 10u) (&S.catcnt[0][0])[tid] = (&S.catpos[0][0])[tid] = 0;
  if (tid == 0) {
    S.nbig = 0;
    S.maxc = 0;
    S.nheavy = 0;
  }
  __syncthreads();
  ISTAMP();
  // ---- sweep 1: class sizes ----
#pragma unroll 1
  for (uint32_t ch = 0; ch < ZES_BLK / (16u * IDX_THREADS); ch++) {
    const uint32_t o = (ch * IDX_THREADS + tid) * 16u;
    if (o >= cnt) continue;
    uint32_t w[5];
    idx_ld20(src, aligned, o, T, w);
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      const uint32_t raw = __builtin_amdgcn_alignbyte(w[(k >> 2) + 1], w[k >> 2], k & 3u);
      const uint32_t h = __umul24(raw, IDX_MUL);
      const uint32_t c = (h >> IDX_RSHIFT) & (IDX_NCLASS - 1u);
      atomicAdd(&cur[o + k < cnt ? c : IDX_NCLASS + lane], 1u);
    }
  }
  __syncthreads();
  ISTAMP();
  // ---- first slot of every class ----
  {
    const uint32_t a = cur[2u * tid], b = cur[2u * tid + 1u];
    uint32_t incl = a + b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
      if ((int)lane >= o) incl += t;
    }
    if (lane == 63) S.wsum[wave] = incl;
    atomicMax(&S.maxc, max(a, b));
    // (what counts as heavy for the decision below scales with the block: a 64 KiB buffer's text has the same heavy keys
    // at half the counts — its blocks stayed here and cost more than k_lz_sort takes for them: 2048 x 64 KiB of text)
    const uint32_t hv = max(128u, (uint32_t)(((uint64_t)IDX_REGCAP * cnt + (ZES_BLK - 1u)) >> 17));
    if (a > hv || b > hv) atomicAdd(&S.nheavy, (a > hv ? a : 0u) + (b > hv ? b : 0u));
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (uint32_t w = 0; w < IDX_WAVES; w++) woff += w < wave ? S.wsum[w] : 0u;
    const uint32_t ex = woff + incl - (a + b);
    S.base[2u * tid] = ex;
    S.base[2u * tid + 1u] = ex + a;
    if (tid == IDX_THREADS - 1u) S.base[IDX_NCLASS] = ex + a + b;
    cur[2u * tid] = ex;  // the cursors of sweep 2
    cur[2u * tid + 1u] = ex + a;
  }
  __syncthreads();
  {
    uint32_t gmax = 0;  // the largest group (uniform after the loop)
#pragma unroll
    for (uint32_t q = 0; q < IDX_NGROUP; q++) gmax = max(gmax, S.base[(q + 1u) * IDX_GCLASS] - S.base[q * IDX_GCLASS]);
    // (uniform) a class or a group the LDS cannot hold — or a block with much of its weight in a few heavy keys (text: " th",
    // "he " ... a fifth of the positions sit in classes that need the radix passes, and k_lz_sort is the faster one there):
    // the block goes back to k_lz_sort
    if (S.maxc > IDX_BIGCAP || gmax > IDX_GCAP || S.nheavy * 16u > cnt) {
      if (tid == 0) A[ZES_BLK - 1] = cnt | ZES_SORT_REDO;
      return;
    }
  }
  __syncthreads();  // (cur[] — the tile buffer from here on — has been read by everybody)
  if (tid < IDX_NGROUP * 4u) S.bcnt[tid] = 0;
  if (tid < IDX_NGROUP) S.gfill[tid] = 0;
  for (uint32_t i = tid; i < IDX_NGROUP * (IDX_NRUN + 1u); i += IDX_THREADS) (&S.runs[0][0])[i] = 0;
  __syncthreads();
  ISTAMP();
  // ---- sweep 2: tile by tile, every position's word (class-in-group 7 | rest of H 13 | low 12 bits of the position) goes
  // to the tile buffer, binned by group and sub-tile, and each group's bin leaves for the group's part of E as one run ----
  uint32_t* tbuf = S.scr;
  const uint32_t st = tid >> 8;  // a thread's 16 positions lie in one sub-tile of 4096: the tile's sub-tile (tid * 16) >> 12
#pragma unroll 1
  for (uint32_t ch = 0; ch < ZES_BLK / IDX_TILE; ch++) {
    const uint32_t o = (ch * IDX_THREADS + tid) * 16u;
    if (ch * IDX_TILE >= cnt) break;  // uniform
    uint32_t w[5];
    if (o < cnt) {
      idx_ld20(src, aligned, o, T, w);
    } else {
#pragma unroll
      for (uint32_t k = 0; k < 5; k++) w[k] = 0;
    }
    uint32_t el[16], bn[16];
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      const uint32_t raw = __builtin_amdgcn_alignbyte(w[(k >> 2) + 1], w[k >> 2], k & 3u);
      const uint32_t h = __umul24(raw, IDX_MUL);
      el[k] = (h << 12) | ((o + k) & (IDX_SUB - 1u));  // (the group's four bits fall off the top)
      bn[k] = o + k < cnt ? ((h >> 20) & (IDX_NGROUP - 1u)) * 4u + st : 0xFFFFFFFFu;
      if (bn[k] != 0xFFFFFFFFu) atomicAdd(&S.bcnt[bn[k]], 1u);
    }
    __syncthreads();
    if (wave == 0) {  // bins in order (group, sub-tile): offsets in the tile buffer; the groups' runs in E grow by what the tile brings
      const uint32_t c = S.bcnt[lane];
      uint32_t incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if ((int)lane >= d) incl += t;
      }
      const uint32_t ex = incl - c;
      S.bpos[lane] = ex;
      S.bcnt[lane] = 0;  // for the next tile
      const uint32_t q = lane >> 2;
      const uint32_t gstart = (uint32_t)__shfl((int)ex, (int)(lane & ~3u));  // where the group's four bins start in the tile buffer
      if ((lane & 3u) == 0u) S.gto[q] = ex;
      if (lane == 63) S.gto[IDX_NGROUP] = incl;
      // run (ch * 4 + sub-tile) of group q starts at gfill[q] + (ex - gstart) words into the group's part of E
      const uint32_t fill = S.gfill[q];
      S.runs[q][ch * 4u + (lane & 3u)] = fill + (ex - gstart);
      const uint32_t gtot = (uint32_t)__shfl((int)incl, (int)(lane | 3u)) - gstart;
      if ((lane & 3u) == 3u) {
        S.gfill[q] = fill + gtot;
        S.runs[q][ch * 4u + 4u] = fill + gtot;  // (the end of the tile's last run; the next tile's first overwrites it with the same value)
      }
    }
    __syncthreads();
    {
      uint32_t at[16];
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) at[k] = bn[k] != 0xFFFFFFFFu ? atomicAdd(&S.bpos[bn[k]], 1u) : 0u;
#pragma unroll
      for (uint32_t k = 0; k < 16; k++)
        if (bn[k] != 0xFFFFFFFFu) tbuf[at[k]] = el[k];
    }
    __syncthreads();
#pragma unroll 1
    for (uint32_t q = 0; q < IDX_NGROUP; q++) {  // the group's run of this tile: contiguous in the buffer and in E
      const uint32_t a0 = S.gto[q], a1 = S.gto[q + 1u];
      uint32_t* dst = E + S.base[q * IDX_GCLASS] + S.runs[q][ch * 4u];
      for (uint32_t i = a0 + tid; i < a1; i += IDX_THREADS) dst[i - a0] = tbuf[i];
    }
    __syncthreads();
  }
  // tiles the block does not reach: their runs are empty
  {
    const uint32_t ntile = (cnt + IDX_TILE - 1u) / IDX_TILE;
    for (uint32_t i = tid; i < IDX_NGROUP * (IDX_NRUN + 1u); i += IDX_THREADS) {
      const uint32_t q = i / (IDX_NRUN + 1u), r = i % (IDX_NRUN + 1u);
      if (r > ntile * 4u) S.runs[q][r] = S.gfill[q];
    }
  }
  // the words are in memory (L2) before anybody reads them back: every storing wave waits for its own stores — a
  // workgroup-scope fence does not (it compiles to lgkmcnt only) — and the readers load past the L1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ISTAMP();
  // ---- group by group: the group's words come back into LDS, each to its class's run (cursors from the class sizes of
  // sweep 1), and every class is sorted by one wavefront, in registers ----
  uint32_t* gbuf = S.scr + IDX_GCAP;
  uint32_t cs = 0;  // which set of category counters this group uses (uniform)
#pragma unroll 1
  for (uint32_t q = 0; q < IDX_NGROUP; q++) {
    const uint32_t gb = S.base[q * IDX_GCLASS], gn = S.base[(q + 1u) * IDX_GCLASS] - gb;
    if (gn == 0u) continue;  // uniform
    // The group's classes in the order the wavefronts take them, largest first: a class costs by the register form that
    // sorts it (64, 128, 256 or 512 words: a factor of ten), so the classes are dealt by size category — every wavefront
    // gets its share of each — instead of by number.  (Handing them out one at a time by an LDS counter was tried: the
    // 64-lane atomic that keeps the loop's exit test uniform cost more than the imbalance.)
    uint32_t mycat = 5u;
    if (tid < IDX_GCLASS) {
      const uint32_t b0 = S.base[q * IDX_GCLASS + tid], n = S.base[q * IDX_GCLASS + tid + 1u] - b0;
      S.ccur[tid] = b0 - gb;
      mycat = n == 0u ? 4u : (n > 256u ? 0u : (n > 128u ? 1u : (n > 64u ? 2u : 3u)));
      atomicAdd(&S.catcnt[cs][mycat], 1u);
    }
    __syncthreads();
    // the other set, for the next group: its last readers (the group before: nact, in front of that group's last barrier)
    // are behind the barrier above, its next writers behind this group's two barriers below.  (One set cleared at the
    // loop's end by wave 0 raced with wave 1's adds at the top of the next turn: a lost count, a class sorted twice.)
    if (tid < 5u) S.catcnt[cs ^ 1u][tid] = S.catpos[cs ^ 1u][tid] = 0;
    if (tid < IDX_GCLASS && mycat < 4u) {
      uint32_t at = atomicAdd(&S.catpos[cs][mycat], 1u);
      for (uint32_t k = 0; k < mycat; k++) at += S.catcnt[cs][k];
      S.order[at] = (uint8_t)tid;
    }
#ifdef IDX_PROF
    const unsigned long long tg0 = clock64();
#endif
    for (uint32_t i0 = 0; i0 < gn; i0 += 8u * IDX_THREADS) {  // eight words per thread in flight
      uint32_t e[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        const uint32_t i = i0 + k * IDX_THREADS + tid;
        e[k] = idx_ld_sc1(E + gb + min(i, gn - 1u));
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        const uint32_t i = i0 + k * IDX_THREADS + tid;
        if (i < gn) {
          // which sub-tile's run holds word i: the largest r with runs[q][r] <= i (binary search over the 33 bounds)
          uint32_t r = 0;
#pragma unroll
          for (uint32_t sh = IDX_NRUN / 2u; sh >= 1u; sh >>= 1) r += S.runs[q][r + sh] <= i ? sh : 0u;
          const uint32_t pos = r * IDX_SUB + (e[k] & (IDX_SUB - 1u));
          const uint32_t at = atomicAdd(&S.ccur[e[k] >> 25], 1u);
          gbuf[at] = ((e[k] >> 12) << 17) | pos;  // bits 17-29: the rest of H; 30, 31: the class's low bits (the same in the whole class)
        }
      }
    }
    __syncthreads();
#ifdef IDX_PROF
    const unsigned long long tg1 = clock64();
    tpart += tg1 - tg0;
#endif
    const uint32_t nact = IDX_GCLASS - S.catcnt[cs][4];
    for (uint32_t k = wave; k < nact; k += IDX_WAVES) {
      const uint32_t j = S.order[k];
      const uint32_t c = q * IDX_GCLASS + j;
      const uint32_t b0 = S.base[c], n = S.base[c + 1u] - b0;
      if (n == 0u) continue;
      const uint32_t* cw = gbuf + (b0 - gb);
      if (n <= 64u) {
        idx_class_regs<1>(S, b0, n, cw, sd, A);
      } else if (n <= 128u) {
        idx_class_regs<2>(S, b0, n, cw, sd, A);
      } else if (n <= 256u) {
        idx_class_regs<4>(S, b0, n, cw, sd, A);
      } else if (n <= IDX_REGCAP) {
        idx_class_regs<8>(S, b0, n, cw, sd, A);
      } else {  // a heavy class: its words go back to E, where the class's run is (the group's words have all been read), for the last round
        for (uint32_t i = lane; i < n; i += 64u) E[b0 + i] = cw[i];
        if (lane == 0) S.big[atomicAdd(&S.nbig, 1u)] = c;
      }
    }
#ifdef IDX_PROF
    const unsigned long long tg2 = clock64();
    tsort += tg2 - tg1;
#endif
    __syncthreads();
#ifdef IDX_PROF
    twait += clock64() - tg2;
#endif
    cs ^= 1u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the heavy classes' words)
  __syncthreads();
  ISTAMP();
  if (wave < IDX_NBIGW) {  // the heavy classes: four wavefronts, a quarter of the LDS each, radix passes
    const uint32_t nbig = S.nbig;
    for (uint32_t k = wave; k < nbig; k += IDX_NBIGW) {
      const uint32_t c = S.big[k];
      const uint32_t b0 = S.base[c], n = S.base[c + 1u] - b0;
      idx_class_heavy(S, b0, n, S.scr + wave * (2u * IDX_BIGCAP), IDX_BIGCAP, S.cnt[wave], E, sd, A);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the bucket words, as above)
  __syncthreads();
  ISTAMP();
  // ---- inv[], a slice of 16384 positions at a time: the bucket's words scattered into an LDS image of the slice ----
  uint32_t* stage = S.scr;
  for (uint32_t s = 0; s < IDX_NSLICE; s++) {
    const uint32_t lo = s * IDX_SLICE;
    if (lo >= T) break;  // uniform
    const uint32_t have = min(cnt, lo + IDX_SLICE) > lo ? min(cnt, lo + IDX_SLICE) - lo : 0u;  // positions of the slice that have a key
    for (uint32_t i = tid; i < IDX_SLICE; i += IDX_THREADS)
      if (i >= have) stage[i] = ZES_INV_NONE;
    {
      uint32_t e[IDX_SLICE / IDX_THREADS];  // the slice's bucket: all loads in flight at once
#pragma unroll
      for (uint32_t k = 0; k < IDX_SLICE / IDX_THREADS; k++) e[k] = idx_ld_sc1(A + lo + min(k * IDX_THREADS + tid, have ? have - 1u : 0u));
#pragma unroll
      for (uint32_t k = 0; k < IDX_SLICE / IDX_THREADS; k++)
        if (k * IDX_THREADS + tid < have) stage[e[k] & (IDX_SLICE - 1u)] = (e[k] >> 31) ? ((e[k] >> 14) & 0x1FFFFu) : ZES_INV_NONE;
    }
    // (inv_all may BE idx_a, the library passes one array for both: slice s of inv[] then lands on bucket s, whose words are
    // all in registers or in the image by now; the block's flag word, slot ZES_BLK-1, travels with the last slice)
    if (tid == 0 && lo + IDX_SLICE == ZES_BLK) stage[IDX_SLICE - 1u] = cnt | ZES_SORT_LAZY | ZES_SORT_INDEX;
    __syncthreads();
    const uint4* st4 = reinterpret_cast<const uint4*>(stage);
    uint4* o4 = reinterpret_cast<uint4*>(inv + lo);
    for (uint32_t i = tid; i < IDX_SLICE / 4u; i += IDX_THREADS) o4[i] = st4[i];
    __syncthreads();
  }
  ISTAMP();
#ifdef IDX_PROF
  if (tid == 0 && (g == 7 || g == 300))
    printf("idx prof block %u: count %llu scan %llu scatter %llu classes %llu heavy %llu inv %llu | heavy classes %u\n", g, tp[1] - tp[0], tp[2] - tp[1],
           tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], tp[6] - tp[5], S.nbig);
  if (tid == 0 && g == 7) printf("idx prof: groups: partition %llu sorts (wave 0) %llu wait at the barrier %llu\n", tpart, tsort, twait);
#endif
  if (tid == 0) A[ZES_BLK - 1] = cnt | ZES_SORT_LAZY | ZES_SORT_INDEX;  // (INDEX stays: "structured data, no heavy keys" — k_lz_order deals such blocks last)
}

// ------------------------------------------------------------------------------------------
// k_lz_order: the order in which k_lz_match_lazy's workgroups take the blocks of a batch of unlike buffers.  A text
// block keeps a compute unit four times as long as a periodic one, and workgroups start in index order: with the
// buffers' own order a batch ends on a tail of text blocks (BASELINE configs[3]: 3.06 ms where the work is 2.1 ms).
// Heaviest first: the blocks k_lz_sort indexed itself (heavy keys: text), then the ones k_lz_index took, then the
// blocks the kernel leaves at once.  One workgroup; the order inside a category does not matter.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_lz_order(const uint32_t* __restrict__ idx_a, uint32_t nblk, uint32_t* __restrict__ order) {
  __shared__ uint32_t cnt[3], pos[3];
  const uint32_t tid = threadIdx.x;
  if (tid < 3) cnt[tid] = pos[tid] = 0;
  __syncthreads();
  auto cat = [&](uint32_t b) {
    const uint32_t f = idx_a[(uint64_t)b * ZES_BLK + ZES_BLK - 1];
    return !(f & ZES_SORT_LAZY) ? 2u : ((f & ZES_SORT_INDEX) ? 1u : 0u);
  };
  for (uint32_t b = tid; b < nblk; b += 1024u) atomicAdd(&cnt[cat(b)], 1u);
  __syncthreads();
  const uint32_t base1 = cnt[0], base2 = cnt[0] + cnt[1];
  for (uint32_t b = tid; b < nblk; b += 1024u) {
    const uint32_t c = cat(b);
    order[(c == 0u ? 0u : (c == 1u ? base1 : base2)) + atomicAdd(&pos[c], 1u)] = b;
  }
}
