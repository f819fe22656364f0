// zes_inflate.hip — gfx950 kernels of the decompress direction.
//
// Four tiers, each falling back to the next on anything unusual:
//
//  T1  block-parallel: reference-made streams are chains of BTYPE=2 blocks that each inflate
//      to exactly 131072 bytes without looking behind their own start.
//        k_inf_scan      every bit position is tested for "could start a clean dynamic block"
//        k_inf_verify    survivors get their whole header decoded and Kraft-checked
//        k_inf_ranksort  candidate positions in ascending order
//        k_inf_block_par one workgroup per candidate block (zes_inflate_par.hip)
//        k_inf_chain     walks end-bit -> next start from bit 16; accepts only a gap-free chain
//  T2  segment-parallel, any valid stream (stored / fixed / dynamic blocks, 32 KiB history across
//      blocks): the candidate block starts cut the stream into segments, one wavefront each.
//        k_inf_seg_scan    decodes a segment with its unknown 32 KiB history as marker symbols
//                          (16-bit ring): gives its end bit, its length and the last 32 Ki
//                          symbols as a map "literal | index into the previous window"
//        k_inf_seg_chain   follows the segments from bit 16 to the final block; output offsets
//        k_inf_seg_win_*   resolve the window behind every segment (groups of maps composed in parallel)
//        k_inf_seg_decode  decodes each segment again, now with its window, into its place
//  T3  k_inf_decode: one wavefront walks all blocks of the stream in order.
//  T4  k_inf_exact: single-lane state-for-state restatement of the reference reader and block
//      decoders (src/inflate.ts, src/utils/BitReadStream.ts), resumed at the block where T3 gave
//      up — reproduces the reference's result on malformed streams (which error, or which bytes).
#include "zes_common.h"
#include "zes_kernels.h"

#define ZES_E_NOT_DEFLATE (-1)
#define ZES_E_BTYPE3 (-2)
#define ZES_E_CORRUPT (-3)
#define ZES_E_INSUFFICIENT (-4)
#define ZES_E_LACK (-5)

// ------------------------------------------------------------------------------------------
// bit window helpers on global memory (scan/verify kernels): 64 bits starting at bit position
// ------------------------------------------------------------------------------------------
__device__ static inline uint64_t g_load64_le(const uint8_t* p, uint64_t byte, uint64_t nbytes) {
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint64_t b = byte + k;
    v |= (uint64_t)(b < nbytes ? p[b] : (uint8_t)0) << (8 * k);
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// k_inf_scan: one thread per bit position (LDS-staged chunk).  Keeps positions whose first 17
// bits look like BTYPE=2 with HLIT<=29, HDIST<=29 and whose code-length code is Kraft-complete.
// ------------------------------------------------------------------------------------------
#define SCAN_BYTES INF_SCAN_BYTES  // bytes of the stream per workgroup
#define RANK_LDS 8192u  // candidates k_inf_ranksort keeps in LDS
#define VERIFY_STEPS 32u  // code-length symbols a lane of k_inf_verify decodes before it hands its survivor to a whole wave
#define SCAN_LIST 1024u             // survivors a workgroup can stage (expected: ~0.2 % of 65536 positions = 130; 16 KiB of LDS in all, so eight workgroups share a compute unit)
// Copies whole 128 KiB slots: item i moves slot src_slot[i] of src to slot dst_slot[i] of dst (32 workgroups
// per item, 16 bytes per lane and step).  Used to close the gaps false candidates leave in the output.
__global__ __launch_bounds__(256) void k_inf_move_slots(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                                        const uint32_t* __restrict__ dst_slot, const uint32_t* __restrict__ src_slot,
                                                        uint32_t nitems) {
  const uint32_t item = blockIdx.x >> 5, part = blockIdx.x & 31u;
  if (item >= nitems) return;
  const uint4* s4 = reinterpret_cast<const uint4*>(src + (uint64_t)src_slot[item] * ZES_BLK) + part * 256u;
  uint4* d4 = reinterpret_cast<uint4*>(dst + (uint64_t)dst_slot[item] * ZES_BLK) + part * 256u;
  d4[threadIdx.x] = s4[threadIdx.x];
}

// One-buffer calls: the two-entry buffer table and the zeroed counters come from kernel arguments (a
// host-to-device copy plus a memset cost two trips through the copy engine).
__global__ void k_inf_set_table1(ZesInfBuf b0, ZesInfBuf sentinel, ZesInfBuf* __restrict__ bufs, uint32_t* __restrict__ counters,
                                 uint32_t nwords) {
  if (threadIdx.x == 0) {
    bufs[0] = b0;
    bufs[1] = sentinel;
  }
  if (counters && threadIdx.x < nwords) counters[threadIdx.x] = 0;
}

// The same for a piece of a stream whose place in the output follows from the pieces before it (host inflate, pieces
// enqueued ahead of the host's look at their predecessors' results): block k of the piece goes to slot acc[0] + k, the
// blocks counted so far by k_inf_chain_range.
__global__ void k_inf_set_table_range(ZesInfBuf b0, ZesInfBuf sentinel, ZesInfBuf* __restrict__ bufs, uint32_t* __restrict__ counters,
                                      uint32_t nwords, const unsigned long long* __restrict__ acc, unsigned long long dcap) {
  if (threadIdx.x == 0) {
    const unsigned long long off = acc[0] * ZES_BLK;
    b0.out_off += off < dcap ? off : dcap;
    b0.cap = dcap - (off < dcap ? off : dcap);
    bufs[0] = b0;
    bufs[1] = sentinel;
  }
  if (counters && threadIdx.x < nwords) counters[threadIdx.x] = 0;
}

// first byte of every buffer of a batch (CM nibble check on the host, src/zlib.ts:13)
__global__ void k_inf_first_bytes(const uint8_t* __restrict__ d_in, const uint64_t* __restrict__ offs, uint8_t* __restrict__ out,
                                  uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d_in[offs[i]];
}

// buffer owning workgroup / work item x: the last entry whose first index is <= x
__device__ __forceinline__ static uint32_t buf_of_chunk(const ZesInfBuf* bufs, uint32_t nbuf, uint32_t x) {
  uint32_t lo = 0, hi = nbuf;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (bufs[mid].first_chunk <= x) lo = mid; else hi = mid;
  }
  return lo;
}

#define SCAN_FINAL_ZONE_BITS (160u * 1024u * 8u)
__global__ __launch_bounds__(INF_SCAN_THREADS) void k_inf_scan(const uint8_t* __restrict__ d_in, const ZesInfBuf* __restrict__ bufs,
                                                               uint32_t nbuf, unsigned long long* __restrict__ surv,
                                                               uint32_t surv_cap, uint32_t* __restrict__ counters,
                                                               uint8_t* __restrict__ first_bytes, uint32_t mode,
                                                               const uint8_t* __restrict__ kraft_tab) {
  const uint32_t loose = mode & 1u;  // bit 0: ZES_F_LOOSE_CANDIDATES; bit 1: see below
  // Each thread owns 32 consecutive bit positions at a time.  The fixed-field tests (BTYPE = 2,
  // HLIT <= 29, HDIST <= 29) run on all 32 positions at once as shifted word logic; only the
  // surviving positions (about one in five) pay for the Kraft sum of the code-length code.
  __shared__ __align__(16) uint32_t s[SCAN_BYTES / 4 + 8];
  __shared__ uint32_t s_list[SCAN_LIST];
  __shared__ uint32_t s_cnt, s_base;
  // Kraft contribution of four 3-bit code-length fields at once (units of 2^-7; a field of 0 adds
  // nothing); saturated at 200 so that an over-full group can never sum back to exactly 128
  __shared__ __align__(16) uint8_t s_kraft[4096];
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_cnt = 0;
  // (the table comes ready-made from global memory: filling it in every workgroup was ~40 % of this kernel's instructions)
  static_assert(INF_SCAN_THREADS * 16 == 4096, "one 16-byte load per thread");
  reinterpret_cast<uint4*>(s_kraft)[tid] = reinterpret_cast<const uint4*>(kraft_tab)[tid];
  const uint32_t bi = buf_of_chunk(bufs, nbuf, blockIdx.x);
  const uint64_t in_off = bufs[bi].in_off, c = bufs[bi].c;
  const uint64_t b0 = (uint64_t)(blockIdx.x - bufs[bi].first_chunk) * SCAN_BYTES;
  const uint32_t* in32 = reinterpret_cast<const uint32_t*>(d_in + in_off);
  const uint64_t ndw = (c + 3) >> 2;
  // The chain of blocks this search serves has to start at bit 16 + start_rel with a dynamic block — the reference
  // writes nothing else (src/deflate.ts:28).  A stream that begins with a stored or fixed block (zlib on incompressible
  // data, level 0) is another encoder's: no survivors, the tier declines without the 0.3 ms per 64 MiB of this search.
  // (mode bit 1: the block-parallel tier's own launches ask for this; the tier for other encoders' streams wants every
  // block start of a stream, whatever it begins with.)
  uint32_t hdr2 = 0;
  bool hdr_known = false;
  if ((mode & 2u) && !loose && !(bufs[bi].range_flags & ZES_START_ANY)) {
    const uint64_t hbit = 16ull + bufs[bi].start_rel;
    if (hbit + 3 <= c * 8) {
      const uint64_t hb = hbit >> 3;
      hdr2 = ((uint32_t)d_in[in_off + hb] | ((uint32_t)d_in[in_off + (hb + 1 < c ? hb + 1 : hb)] << 8)) >> (uint32_t)(hbit & 7u);
      hdr_known = true;
    }
  }
  {
    // all loads first (clamped addresses, so they are unconditional and overlap), then the masking
    constexpr uint32_t NLD = (SCAN_BYTES / 4 + 8 + INF_SCAN_THREADS - 1) / INF_SCAN_THREADS;
    uint32_t v[NLD];
#pragma unroll
    for (uint32_t j = 0; j < NLD; j++) {
      const uint64_t gi = (b0 >> 2) + tid + j * INF_SCAN_THREADS;
      v[j] = in32[gi < ndw ? gi : ndw - 1];
    }
#pragma unroll
    for (uint32_t j = 0; j < NLD; j++) {
      const uint32_t i = tid + j * INF_SCAN_THREADS;
      const uint64_t gi = (b0 >> 2) + i;
      uint32_t x = v[j];
      if (gi == ndw - 1 && (c & 3u)) x &= (1u << (8u * (uint32_t)(c & 3u))) - 1u;  // bytes past the end read as 0
      if (gi >= ndw) x = 0;
      if (i < SCAN_BYTES / 4 + 8) s[i] = x;
    }
  }
  __syncthreads();
  // the buffer's first byte rides back with the counters (CM nibble check on the host, src/zlib.ts:13)
  // (low nibble: CM; bits 4-5: BTYPE of the block at bit 16, bit 6: "it is there" — the host skips the stored-blocks
  // attempt for a stream that does not begin with one)
  if (tid == 0 && b0 == 0) first_bytes[bi] = (uint8_t)((s[0] & 0x0fu) | (((s[0] >> 17) & 3u) << 4) | 0x40u);
  if (hdr_known && ((hdr2 >> 1) & 3u) != 2u) return;  // (uniform)
  const uint64_t end_bits = c * 8;
  // Two passes at most.  The first stages the survivors in LDS; a chunk with more of them than the list holds (a
  // periodic stream: the bit pattern of one repeated match passes the test at every repetition) is walked again and
  // writes them straight to their reserved slots — every reserved slot below surv_cap gets written, whatever the data.
  uint32_t gb = 0;
  for (uint32_t pass = 0; pass < 2u; pass++) {
#pragma unroll 1
    for (uint32_t k = 0; k < SCAN_BYTES / 4 / INF_SCAN_THREADS; k++) {
      const uint32_t grp = k * INF_SCAN_THREADS + tid;  // dword of positions inside the chunk
      const uint64_t w0 = (uint64_t)s[grp] | ((uint64_t)s[grp + 1] << 32);
      const uint64_t w1 = (uint64_t)s[grp + 2] | ((uint64_t)s[grp + 3] << 32);
      // bit i of m: position i has bit1 = 0, bit2 = 1 (BTYPE 2), not all of bits 4..7 (HLIT <= 29),
      // not all of bits 9..12 (HDIST <= 29)
      uint32_t m = (uint32_t)(~(w0 >> 1) & (w0 >> 2) & ~((w0 >> 4) & (w0 >> 5) & (w0 >> 6) & (w0 >> 7)) &
                              ~((w0 >> 9) & (w0 >> 10) & (w0 >> 11) & (w0 >> 12)));
      const uint64_t abs_base = b0 * 8 + (uint64_t)grp * 32;
      // Only the last block of a stream has BFINAL set, and a block this tier decodes has at most 144 KiB of
      // compressed data: further from the end than that, a position with bit0 = 1 is not a block start (half of
      // all positions, so half of the survivors the verify kernel would have to decode).
      if (!loose && abs_base + 32 + SCAN_FINAL_ZONE_BITS <= end_bits) m &= (uint32_t)~w0;
      while (m) {
        const uint32_t i = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        const uint64_t abs_bit = abs_base + i;
        if (abs_bit < 16ull + bufs[bi].start_rel || abs_bit + 17 + 12 > end_bits) continue;  // (start_rel: 0 unless the buffer is a piece of a longer stream)
        const uint64_t lo = i ? ((w0 >> i) | (w1 << (64u - i))) : w0;
        const uint64_t hi = w1 >> i;
        const uint32_t ncl = (uint32_t)((lo >> 13) & 15u) + 4u;
        const uint64_t clb = ((lo >> 17) | (hi << 47)) & ((1ull << (3u * ncl)) - 1ull);  // ncl x 3 bits, up to 57
        // The reference sends exactly as many code-length-code lengths as reach its last used symbol
        // (src/deflate.ts:143-148), so the last one is never zero.  T1 only has to find reference-made
        // blocks: a stream from an encoder that pads this list is still decoded, by T2.
        if (((clb >> (3u * ncl - 3u)) & 7ull) == 0ull) continue;
        const uint32_t c_lo = (uint32_t)clb, c_hi = (uint32_t)(clb >> 32);
        const uint32_t kraft = (uint32_t)s_kraft[c_lo & 4095u] + s_kraft[(c_lo >> 12) & 4095u] +
                               s_kraft[((c_lo >> 24) | (c_hi << 8)) & 4095u] + s_kraft[(c_hi >> 4) & 4095u] +
                               s_kraft[(c_hi >> 16) & 4095u];
        if (kraft != 128u) continue;
        const uint32_t slot = atomicAdd(&s_cnt, 1u);  // LDS: one global atomic per workgroup below
        const uint32_t rel16 = (uint32_t)(abs_bit - 16);  // relative to bit 16 (fits u32 for c < 512 MiB)
        if (pass == 0u) {
          if (slot < SCAN_LIST) s_list[slot] = rel16;
        } else if (gb + slot < surv_cap) {
          surv[gb + slot] = ((unsigned long long)bi << 32) | rel16;
        }
      }
    }
    __syncthreads();
    const uint32_t n = s_cnt;
    if (n == 0 || pass == 1u) return;  // (uniform)
    if (tid == 0) s_base = atomicAdd(&counters[0], n);
    __syncthreads();
    gb = s_base;
    if (n <= SCAN_LIST) {
      for (uint32_t i = tid; i < n; i += INF_SCAN_THREADS)
        if (gb + i < surv_cap) surv[gb + i] = ((unsigned long long)bi << 32) | s_list[i];
      return;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// k_inf_verify / k_inf_verify_long: the header test of the block-start search.  A survivor of the scan has its
// code-length sequence decoded and the codes checked: lit/len complete with an end-of-block code; distances complete,
// a single 1-bit code, or absent.  Two launches:
//   k_inf_verify       one LANE per survivor, the first VERIFY_STEPS code-length symbols.  Seven in eight are rejected
//                      by then (a code over-subscribes, or the sequence breaks the reference's run-length rules); the
//                      others go on a list with their state.
//   k_inf_verify_long  one WAVE per listed survivor, to the end: the real headers (~300 symbols) and under-subscribed
//                      garbage.  A lone lane needs ~110 instructions per symbol, 4 cycles each whatever the other 63
//                      lanes do: 300 symbols held its wave for ~90 us.  The wave form decodes the symbol at each of
//                      128 bit offsets at once, walks the chain through them (one lane read per symbol), compacts the
//                      ~22 symbols on the chain into lanes and runs the sequence logic on all of them with wave scans.
// Lane form: the stream bits a lane needs (18 dwords: 32 symbols of at most 14 bits) are copied to LDS in one batch of
// loads when it starts, so a step has no global load in it, and the step has no branch in it.  Per-lane decode table of
// the code-length code in LDS: 128 bytes indexed by the next 7 stream bits read MSB-first (bit-reversed window),
// entry = symbol | length << 5.  In that index space a canonical code is a run of consecutive entries, so the table is
// filled front to back in (length, symbol) order.  Dword j of lane l lives at [j][l].
// ------------------------------------------------------------------------------------------
// dwords of the stream per lane.  VERIFY_STEPS symbols of at most 14 bits + alignment + read-ahead would be 18; the usual
// 32 symbols take ~200 bits, and a lane that runs out reloads (one more load latency for its wave).  What the smaller
// window buys is LDS: 8 KiB of tables + 3 KiB of windows per wave are 14 waves a CU instead of 12 — this kernel is one
// batch's latency (~40 us) per round of resident waves, and the 3171 batches of 64 MiB of incompressible data were
// 3072 + 99: two rounds, 84 us; now one.
#ifndef VERIFY_WIN
#define VERIFY_WIN 12u
#endif
#define VSTATE_WORDS 6u  // per listed survivor: index in surv[], bit position, k, kl, kd, (psym | prev << 5 | eob << 9 | dmaxlen << 10 | nd << 14)

// the reference's run-length coding of the code lengths (src/deflate.ts:100-139) never produces these (another
// encoder's stream that does is decoded by T2):
//  - "repeat previous" (16) directly after anything but a plain non-zero length, or with count 6: a run is cut into
//    chunks of at most 6, the chunk's first length is written out, 16 repeats the other 3..5
//  - a zero run (17) of 3: three zeros are written as three plain zeros
//  - 17 closes its zero run (it codes the remainder 4..10), so a non-zero plain length follows; plain zeros (a
//    remainder below 4) close their run as well, so no 17/18 follows them
__device__ __forceinline__ uint32_t verify_rule_break(uint32_t psym, uint32_t sy, uint32_t xv, uint32_t is16, uint32_t is17, uint32_t is18) {
  const uint32_t p_plain_nz = (uint32_t)(psym - 1u < 15u), s_plain_nz = (uint32_t)(sy - 1u < 15u);
  return (is16 & ((p_plain_nz ^ 1u) | (uint32_t)(xv == 3u))) | (is17 & (uint32_t)(xv == 0u)) | ((uint32_t)(psym == 17u) & (s_plain_nz ^ 1u)) |
         ((uint32_t)(psym == 0u) & (is17 | is18));
}

__global__ __launch_bounds__(64) void k_inf_verify(const uint8_t* __restrict__ d_in, const ZesInfBuf* __restrict__ bufs,
                                                   const unsigned long long* __restrict__ surv, uint32_t surv_cap,
                                                   uint32_t* __restrict__ counters, uint32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ cnt, uint32_t loose, uint32_t* __restrict__ vstate,
                                                   uint32_t vlong_cap, uint32_t vsteps) {
  constexpr uint32_t WIN = VERIFY_WIN;
  __shared__ uint32_t s_lut[32][64];
  __shared__ uint32_t s_win[WIN][64];
  const uint32_t lane = threadIdx.x;
  uint8_t* lut8 = reinterpret_cast<uint8_t*>(&s_lut[0][0]);
  const uint32_t n = min(counters[0], surv_cap);
  vsteps = min(max(vsteps, 1u), VERIFY_STEPS);
  constexpr uint64_t M0 = 0x0049249249249249ull;  // bit 0 of each of the 19 three-bit fields
#pragma unroll 1
  for (uint32_t base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {
    const uint32_t item = base + lane;
    bool have = item < n;
    const uint32_t si = have ? item : base;
    const unsigned long long sv = surv[si];
    const uint32_t mybuf = (uint32_t)(sv >> 32);
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(d_in + bufs[mybuf].in_off);
    const uint32_t lastdw = (uint32_t)((bufs[mybuf].c - 1) >> 2);
    const uint32_t limit = (uint32_t)(bufs[mybuf].c * 8);
    const uint32_t pos0 = (uint32_t)sv + 16u;
    uint32_t HLIT, total, pos;
    uint64_t clb;
    {
      // 17 header bits + up to 57 bits of code-length-code lengths, from four dwords
      const uint32_t di = pos0 >> 5, sh = pos0 & 31u;
      const uint64_t w0 = (uint64_t)in32[min(di, lastdw)] | ((uint64_t)in32[min(di + 1, lastdw)] << 32);
      const uint64_t w1 = (uint64_t)in32[min(di + 2, lastdw)] | ((uint64_t)in32[min(di + 3, lastdw)] << 32);
      const uint64_t lo = sh ? ((w0 >> sh) | (w1 << (64u - sh))) : w0;
      const uint64_t hi = w1 >> sh;
      HLIT = ((uint32_t)(lo >> 3) & 31u) + 257u;
      const uint32_t HDIST = ((uint32_t)(lo >> 8) & 31u) + 1u, HCLEN = ((uint32_t)(lo >> 13) & 15u) + 4u;
      total = HLIT + HDIST;
      clb = ((lo >> 17) | (hi << 47)) & ((1ull << (3u * HCLEN)) - 1ull);
      pos = pos0 + 17u + 3u * HCLEN;
    }
    // the lane's window of the stream, from the dword of the first code-length symbol
    uint32_t wd0 = pos >> 5;
#pragma unroll
    for (uint32_t j0 = 0; j0 < WIN; j0 += 16u) {
      uint32_t t[16];
#pragma unroll
      for (uint32_t j = 0; j < 16u; j++)
        if (j0 + j < WIN) t[j] = in32[min(wd0 + j0 + j, lastdw)];
#pragma unroll
      for (uint32_t j = 0; j < 16u; j++)
        if (j0 + j < WIN) s_win[j0 + j][lane] = t[j];
    }
    // lengths from transmission order into symbol order (3 bits per symbol)
    uint64_t sl = 0;
#pragma unroll
    for (uint32_t q = 0; q < 19; q++) {
      constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};  // src/const.ts:31-35
      sl |= ((clb >> (3u * q)) & 7ull) << (3u * order[q]);
    }
    // fill the table: lengths ascending, symbols ascending (the scan kernel has already checked that the lengths
    // form a complete code, so exactly 128 entries get written)
    uint32_t fill = 0;
    if (have) {
#pragma unroll 1
      for (uint32_t L = 1; L <= 7; L++) {
        const uint64_t x = sl ^ (M0 * L);
        uint64_t z = ~(x | (x >> 1) | (x >> 2)) & M0;  // fields equal to L
        while (z) {
          const uint32_t t = (uint32_t)__builtin_ctzll(z);
          z &= z - 1ull;
          const uint32_t e = ((t * 11u) >> 5) | (L << 5);  // t / 3 for t <= 54
          const uint32_t nn = 128u >> L;
          if (fill + nn <= 128u) {
            if (nn >= 4u) {
              const uint32_t v = e * 0x01010101u;
              for (uint32_t j = 0; j < (nn >> 2); j++) s_lut[(fill >> 2) + j][lane] = v;
            } else {
              for (uint32_t j = 0; j < nn; j++) lut8[(((fill + j) >> 2) * 64u + lane) * 4u + ((fill + j) & 3u)] = (uint8_t)e;
            }
          }
          fill += nn;
        }
      }
    }
    have = have && fill == 128u;  // anything else cannot come from the scan kernel; dropped, not decoded
    uint64_t bb;
    uint32_t nb, wi = 2;
    {
      const uint32_t sh = pos & 31u;
      bb = ((uint64_t)s_win[0][lane] | ((uint64_t)s_win[1][lane] << 32)) >> sh;
      nb = 64u - sh;
    }
    uint32_t k = 0, kl = 0, kd = 0, nd = 0, dmaxlen = 0, prev = 0, psym = 31;  // psym 31: no previous symbol
    uint32_t has_eob = 0;
    uint32_t step = 0;
    const uint32_t strict = loose ? 0u : 1u;  // (ZES_F_LOOSE_CANDIDATES switches the rules off to exercise the false-candidate path)
#pragma unroll 1
    while (__ballot(have)) {
      if (have) {
        // ---- one code-length symbol; no branch in it: a step of a lone lane is a dependent chain, and every exec-mask
        // region the compiler builds for an `if` adds to it (the branchy form took ~1400 cycles per step) ----
        uint32_t ok = (uint32_t)(pos + 14u <= limit);
        {
          const uint32_t need = (uint32_t)(nb <= 32u);
          if (__ballot(need && wi == WIN)) {  // only a lane that found no room on the list gets here: the next window
            if (need && wi == WIN) {
              wd0 += WIN;
              for (uint32_t j = 0; j < WIN; j++) s_win[j][lane] = in32[min(wd0 + j, lastdw)];
              wi = 0;
            }
          }
          const uint32_t wv = s_win[min(wi, WIN - 1u)][lane];  // (read whether needed or not: no branch)
          const uint32_t w = need ? wv : 0u;
          bb |= (uint64_t)w << (nb & 63u);
          nb += need << 5;
          wi += need;
        }
        const uint32_t lo = (uint32_t)bb;
        const uint32_t ix = __brev(lo) >> 25;
        const uint32_t ent = lut8[((ix >> 2) * 64u + lane) * 4u + (ix & 3u)];
        const uint32_t sy = ent & 31u, len = ent >> 5;
        const uint32_t is16 = (uint32_t)(sy == 16u), is17 = (uint32_t)(sy == 17u), is18 = (uint32_t)(sy == 18u);
        const uint32_t run = is16 | is17 | is18;
        const uint32_t xb = is16 * 2u + is17 * 3u + is18 * 7u;
        const uint32_t xv = __builtin_amdgcn_ubfe(lo, len, xb);  // (width 0 gives 0)
        const uint32_t rep = (run ? (is18 ? 11u : 3u) : 1u) + xv;
        const uint32_t val = run ? (is16 ? prev : 0u) : sy;
        ok &= (uint32_t)!(is16 & (uint32_t)(k == 0u));
        ok &= (verify_rule_break(psym, sy, xv, is16, is17, is18) & strict) ^ 1u;
        psym = sy;
        const uint32_t adv = len + xb;
        bb >>= adv;
        nb -= adv;
        pos += adv;
        ok &= (uint32_t)(k + rep <= total);
        {
          // rep entries of length val starting at index k: split at the lit/len | distance border (val = 0 adds nothing)
          const uint32_t c = val ? (32768u >> val) : 0u;
          const uint32_t nl = min(rep, max(HLIT, k) - k);
          const uint32_t ndd = rep - nl;
          kl += nl * c;
          kd += ndd * c;
          const uint32_t dd = val ? ndd : 0u;
          nd += dd;
          dmaxlen = max(dmaxlen, dd ? val : 0u);
          has_eob |= (uint32_t)(val != 0u) & (uint32_t)(k <= 256u) & (uint32_t)(256u < k + nl);
          // garbage headers over-subscribe a code within a few symbols: stop right there
          ok &= (uint32_t)(kl <= 32768u) & (uint32_t)(kd <= 32768u);
        }
        prev = val;
        k += rep;
        const uint32_t fin = (uint32_t)(k >= total);
        const uint32_t good = ok & fin & has_eob & (uint32_t)(kl == 32768u) &
                              ((uint32_t)(kd == 32768u) | (uint32_t)(nd == 0u) | ((uint32_t)(nd == 1u) & (uint32_t)(dmaxlen == 1u)));
        if (__ballot(good)) {  // rare: about one per block of the stream
          if (good) {
            const uint32_t slot = atomicAdd(&cnt[mybuf], 1u);
            if (slot < bufs[mybuf].cand_cap) cand[bufs[mybuf].cand_base + slot] = pos0 - 16u;
          }
        }
        have = (ok & (fin ^ 1u)) != 0u;
      }
      if (++step == vsteps) {
        // still alive: onto the list of the long pass, with the state reached (one atomic per wave)
        const uint64_t alive = __ballot(have);
        if (alive) {
          uint32_t slot0 = 0;
          if (lane == 0) slot0 = atomicAdd(&counters[1], (uint32_t)__popcll(alive));
          slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
          const uint32_t slot = slot0 + (uint32_t)__popcll(alive & zes_lanemask_lt());
          if (have && slot < vlong_cap) {  // (no room — ten times the usual share of long runners: the lane carries on by itself)
            uint32_t* st = vstate + (size_t)slot * VSTATE_WORDS;
            st[0] = si;
            st[1] = pos;
            st[2] = k;
            st[3] = kl;
            st[4] = kd;
            st[5] = psym | (prev << 5) | (has_eob << 9) | (dmaxlen << 10) | (nd << 14);
            have = false;
          }
        }
      }
    }
  }
}

// wave scans over 64 lanes (inclusive), DPP: four steps inside each row of 16, then the row totals carried over
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t vdpp(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, ROWS, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
  x += vdpp<0x111, 0xf>(0u, x);
  x += vdpp<0x112, 0xf>(0u, x);
  x += vdpp<0x114, 0xf>(0u, x);
  x += vdpp<0x118, 0xf>(0u, x);
  x += vdpp<0x142, 0xa>(0u, x);  // row_bcast:15 into rows 1 and 3
  x += vdpp<0x143, 0xc>(0u, x);  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
  x = max(x, vdpp<0x111, 0xf>(0u, x));
  x = max(x, vdpp<0x112, 0xf>(0u, x));
  x = max(x, vdpp<0x114, 0xf>(0u, x));
  x = max(x, vdpp<0x118, 0xf>(0u, x));
  x = max(x, vdpp<0x142, 0xa>(0u, x));
  x = max(x, vdpp<0x143, 0xc>(0u, x));
  return x;
}

__global__ __launch_bounds__(64) void k_inf_verify_long(const uint8_t* __restrict__ d_in, const ZesInfBuf* __restrict__ bufs,
                                                        const unsigned long long* __restrict__ surv, uint32_t surv_cap,
                                                        uint32_t* __restrict__ counters, uint32_t* __restrict__ cand,
                                                        uint32_t* __restrict__ cnt, uint32_t loose, const uint32_t* __restrict__ vstate,
                                                        uint32_t vlong_cap) {
  __shared__ uint8_t s_lut[128];   // 7 stream bits read MSB-first -> symbol | length << 5
  __shared__ uint32_t s_sym[64];   // the symbols on the chain, in order
  const uint32_t lane = threadIdx.x;
  const uint32_t n = min(counters[1], vlong_cap);
  const uint32_t strict = loose ? 0u : 1u;
  constexpr uint64_t M0 = 0x0049249249249249ull;
  auto rl = [](uint32_t v, uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i); };
#pragma unroll 1
  for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
    const uint32_t* st = vstate + (size_t)it * VSTATE_WORDS;
    const uint32_t si = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[0]);
    uint32_t P = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[1]);
    uint32_t c_k = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[2]), c_kl = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[3]);
    uint32_t c_kd = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[4]);
    const uint32_t pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[5]);
    uint32_t c_psym = pk & 31u, c_prev = (pk >> 5) & 15u, c_eob = (pk >> 9) & 1u, c_dmax = (pk >> 10) & 15u, c_nd = pk >> 14;
    const unsigned long long sv0 = surv[si];
    const uint32_t mybuf = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sv0 >> 32));
    const uint32_t pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sv0) + 16u;
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(d_in + bufs[mybuf].in_off);
    const uint32_t lastdw = (uint32_t)((bufs[mybuf].c - 1) >> 2);
    const uint32_t limit = (uint32_t)(bufs[mybuf].c * 8);
    uint32_t d0 = pos0 >> 5;  // first dword of the 64-dword window: lane l holds dword d0 + l
    uint32_t W = in32[min(d0 + lane, lastdw)];
    uint32_t HLIT, total;
    uint64_t sl = 0;
    {
      const uint32_t sh = pos0 & 31u;
      const uint64_t w0 = (uint64_t)rl(W, 0) | ((uint64_t)rl(W, 1) << 32);
      const uint64_t w1 = (uint64_t)rl(W, 2) | ((uint64_t)rl(W, 3) << 32);
      const uint64_t lo = sh ? ((w0 >> sh) | (w1 << (64u - sh))) : w0;
      const uint64_t hi = w1 >> sh;
      HLIT = ((uint32_t)(lo >> 3) & 31u) + 257u;
      const uint32_t HDIST = ((uint32_t)(lo >> 8) & 31u) + 1u, HCLEN = ((uint32_t)(lo >> 13) & 15u) + 4u;
      total = HLIT + HDIST;
      const uint64_t clb = ((lo >> 17) | (hi << 47)) & ((1ull << (3u * HCLEN)) - 1ull);
#pragma unroll
      for (uint32_t q = 0; q < 19; q++) {
        constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};  // src/const.ts:31-35
        sl |= ((clb >> (3u * q)) & 7ull) << (3u * order[q]);
      }
    }
    // the table, two entries per lane: codes in (length, symbol) order are consecutive runs of entries
    uint32_t e_lo = 0, e_hi = 0, fill = 0;
#pragma unroll 1
    for (uint32_t L = 1; L <= 7; L++) {
      const uint64_t x = sl ^ (M0 * L);
      uint64_t z = ~(x | (x >> 1) | (x >> 2)) & M0;
      while (z) {
        const uint32_t t = (uint32_t)__builtin_ctzll(z);
        z &= z - 1ull;
        const uint32_t e = ((t * 11u) >> 5) | (L << 5);
        const uint32_t nn = 128u >> L;
        if (lane >= fill && lane < fill + nn) e_lo = e;
        if (lane + 64u >= fill && lane + 64u < fill + nn) e_hi = e;
        fill += nn;
      }
    }
    __syncthreads();  // (the item before has finished with the table)
    s_lut[lane] = (uint8_t)e_lo;
    s_lut[lane + 64u] = (uint8_t)e_hi;
    __syncthreads();
    bool running = fill == 128u, good = false;
#pragma unroll 1
    while (running) {
      uint32_t i0 = (P >> 5) - d0;
      if (i0 + 5u > 63u) {  // the window is used up: the next one starts at P's dword
        d0 = P >> 5;
        W = in32[min(d0 + lane, lastdw)];
        i0 = 0;
      }
      // the symbol that would start at bit P + lane and the one at P + 64 + lane
      uint32_t p0, p1;
      {
        const uint32_t a = rl(W, i0), b = rl(W, i0 + 1u), c = rl(W, i0 + 2u), d = rl(W, i0 + 3u), e = rl(W, i0 + 4u), f = rl(W, i0 + 5u);
        const uint32_t sh = (P & 31u) + lane;  // 0 .. 94
        const bool s32 = sh < 32u, s64 = sh < 64u;
        const uint32_t x0 = s32 ? a : s64 ? b : c, x1 = s32 ? b : s64 ? c : d;
        const uint32_t y0 = s32 ? c : s64 ? d : e, y1 = s32 ? d : s64 ? e : f;
        const uint32_t v0 = __builtin_amdgcn_alignbit(x1, x0, sh & 31u) & 0x3FFFu;  // the 14 bits at P + lane
        const uint32_t v1 = __builtin_amdgcn_alignbit(y1, y0, sh & 31u) & 0x3FFFu;  // ... at P + 64 + lane
        auto dec = [&](uint32_t v, uint32_t off) {
          const uint32_t ent = s_lut[__brev(v) >> 25];
          const uint32_t sy = ent & 31u, len = ent >> 5;
          const uint32_t xb = (uint32_t)(sy == 16u) * 2u + (uint32_t)(sy == 17u) * 3u + (uint32_t)(sy == 18u) * 7u;
          const uint32_t xv = __builtin_amdgcn_ubfe(v, len, xb);
          return sy | (xv << 5) | ((len + xb) << 12) | (off << 16);
        };
        p0 = dec(v0, lane);
        p1 = dec(v1, lane + 64u);
      }
      // the chain through the 128 offsets (at most 64 symbols a round)
      uint64_t m0 = 0, m1 = 0;
      uint32_t cur = 0, nsym = 0;
#pragma unroll 1
      while (cur < 64u) {
        m0 |= 1ull << cur;
        cur += (rl(p0, cur) >> 12) & 15u;
        nsym++;
      }
#pragma unroll 1
      while (cur < 128u && nsym < 64u) {
        m1 |= 1ull << (cur - 64u);
        cur += (rl(p1, cur - 64u) >> 12) & 15u;
        nsym++;
      }
      // into lanes 0 .. nsym-1, in order
      {
        const uint64_t lt = zes_lanemask_lt();
        const uint32_t r0 = (uint32_t)__popcll(m0 & lt), r1 = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & lt);
        __syncthreads();
        if ((m0 >> lane) & 1ull) s_sym[r0] = p0;
        if ((m1 >> lane) & 1ull) s_sym[r1] = p1;
        __syncthreads();
      }
      const uint32_t t = s_sym[lane];
      const bool valid = lane < nsym;
      const uint32_t sy = valid ? (t & 31u) : 0u, xv = valid ? ((t >> 5) & 127u) : 0u, off = (t >> 16) & 127u;
      const uint32_t is16 = (uint32_t)(sy == 16u), is17 = (uint32_t)(sy == 17u), is18 = (uint32_t)(sy == 18u);
      const uint32_t run = is16 | is17 | is18;
      const uint32_t rep = (run ? (is18 ? 11u : 3u) : 1u) + xv;
      const uint32_t valraw = run ? 0u : sy;
      // "repeat previous" takes the value of the last symbol that is not one (in front of the round: the carry)
      const uint32_t src = wave_scan_max(is16 ? 0u : lane + 1u);
      const uint32_t vsrc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src ? src - 1u : 0u) << 2), (int)valraw);
      const uint32_t val = src ? vsrc : c_prev;
      const uint32_t psym = vdpp<0x138, 0xf>(c_psym, sy);  // wave_shr:1, lane 0 keeps the carry
      const uint32_t kin = wave_scan_add(rep);
      const uint32_t k = c_k + kin - rep;
      const uint32_t c = val ? (32768u >> val) : 0u;
      const uint32_t nl = min(rep, max(HLIT, k) - k);
      const uint32_t ndd = rep - nl;
      const uint32_t kl = c_kl + wave_scan_add(nl * c), kd = c_kd + wave_scan_add(ndd * c);
      const uint32_t dd = val ? ndd : 0u;
      uint32_t ok = (uint32_t)(P + off + 14u <= limit);
      ok &= (uint32_t)!(is16 & (uint32_t)(k == 0u));
      ok &= (verify_rule_break(psym, sy, xv, is16, is17, is18) & strict) ^ 1u;
      ok &= (uint32_t)(k + rep <= total) & (uint32_t)(kl <= 32768u) & (uint32_t)(kd <= 32768u);
      const uint64_t notok = __ballot(valid && !ok), finm = __ballot(valid && k + rep >= total);
      const uint32_t e = notok ? (uint32_t)__builtin_ctzll(notok) : 64u, f = finm ? (uint32_t)__builtin_ctzll(finm) : 64u;
      if (e <= f && e < 64u) {
        running = false;  // rejected (a symbol that breaks a rule at or before the one that completes the sequence)
      } else {
        const uint32_t cut = f < 64u ? f : nsym - 1u;
        const uint32_t nd = c_nd + rl(wave_scan_add(dd), cut);
        const uint32_t dmax = max(c_dmax, rl(wave_scan_max(dd ? val : 0u), cut));
        const uint32_t eob = c_eob | (uint32_t)(__ballot(lane <= cut && val != 0u && k <= 256u && 256u < k + nl) != 0ull);
        const uint32_t kl_c = rl(kl, cut), kd_c = rl(kd, cut);
        if (f < 64u) {
          good = eob && kl_c == 32768u && (kd_c == 32768u || nd == 0u || (nd == 1u && dmax == 1u));
          running = false;
        } else {
          c_k = rl(k + rep, cut);
          c_kl = kl_c;
          c_kd = kd_c;
          c_nd = nd;
          c_dmax = dmax;
          c_eob = eob;
          c_prev = rl(val, cut);
          c_psym = rl(sy, cut);
          P += cur;
        }
      }
    }
    if (good && lane == 0) {
      const uint32_t slot = atomicAdd(&cnt[mybuf], 1u);
      if (slot < bufs[mybuf].cand_cap) cand[bufs[mybuf].cand_base + slot] = pos0 - 16u;
    }
  }
}

// rank sort of each buffer's candidate list (a few to a few thousand entries); one workgroup per buffer
__global__ __launch_bounds__(1024) void k_inf_ranksort(const ZesInfBuf* __restrict__ bufs, const uint32_t* __restrict__ cnt,
                                                      const uint32_t* __restrict__ cand, uint32_t* __restrict__ out, uint32_t max_n) {
  __shared__ uint32_t s_c[RANK_LDS];  // the list itself when it fits (it does: one entry per block of the stream)
  const ZesInfBuf bf = bufs[blockIdx.x];
  const uint32_t n = min(cnt[blockIdx.x], bf.cand_cap);
  if (n > max_n) return;  // (a list this long is thinned instead: the segment-parallel tier's group search)
  const uint32_t* in = cand + bf.cand_base;
  uint32_t* o = out + bf.cand_base;
  const bool lds = n <= RANK_LDS;
  if (lds) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_c[i] = in[i];
    __syncthreads();
  }
  const uint32_t* src = lds ? s_c : in;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t v = src[i];
    uint32_t r = 0;
    uint32_t j = 0;
    for (; j + 16u <= n; j += 16u) {  // sixteen independent reads per step: the loop is bound by LDS latency
      uint32_t u[16];
#pragma unroll
      for (int k = 0; k < 16; k++) u[k] = src[j + k];
#pragma unroll
      for (int k = 0; k < 16; k++) r += (u[k] < v) || (u[k] == v && j + k < i);
    }
    for (; j < n; j++) {
      const uint32_t u = src[j];
      r += (u < v) || (u == v && j < i);
    }
    o[r] = v;
  }
}

// ------------------------------------------------------------------------------------------
// wave decoder
// ------------------------------------------------------------------------------------------
#define LROOT 11u
#define DROOT 9u
#define RING 65536u
#define FLUSH 16384u
#define RING16 33280u  // marker ring (16-bit symbols), full form: window + longest match + slack; not a power of two
// The segment-parallel tier also has a short form of the marker ring: three decoders fit a CU instead of two (the ring is
// what the decoder's LDS goes to), and a match that reaches further back than the ring takes its symbols from the
// segment's symbol store in global memory — or, in front of the segment's first byte, writes the marker values directly.
#ifndef RING16_SHORT
#define RING16_SHORT 20480u
#endif
#define FLUSH_SHORT 4096u  // the short ring flushes to the symbol store this often: everything older than that is in the store

// One wavefront decodes serially; all decoder state is wave-uniform and kept in scalar registers
// (every value that comes out of LDS goes through readfirstlane), so the per-token work is a few
// scalar instructions around one or two LDS table reads.  The compressed input sits in two vector
// registers (lane l = dword l of a 64-dword window and of the next one) and is picked with
// readlane; a window is re-loaded every 2048 bits, one window ahead of its use.
template <uint32_t R16>
struct InfSmemT {
  static constexpr uint32_t kR16 = R16;                                   // symbols of the marker ring
  static constexpr uint32_t kFlush = R16 >= RING16 ? FLUSH : FLUSH_SHORT;  // marker mode: symbols per flush to the store
  static constexpr uint32_t kNear = R16 - ZES_MAXMATCH - 64u;              // a match at most this far back is wholly inside the ring
  uint8_t ring[R16 * 2];  // byte mode uses the first RING bytes (full form only)
  uint32_t lut_l[1u << LROOT];  // wd_entry() words; 0 = code longer than the root, or no code
  uint32_t lut_d[1u << DROOT];
  uint16_t syms_l[288];
  uint16_t syms_d[32];
  uint32_t first_l[16], first_d[16];
  uint16_t cnt_l[16], cnt_d[16], offs_l[16], offs_d[16];
  uint8_t lens[352];  // [0,288) lit/len, [288,320) dist
  uint8_t cl_lut[128];
};
using InfSmem = InfSmemT<RING16>;
using InfSmemShort = InfSmemT<RING16_SHORT>;
static_assert(sizeof(InfSmemShort) * 3 <= 160 * 1024, "at least three short-ring decoders per CU");

struct WaveDec {
  // uniform state (identical in all 64 lanes)
  const uint32_t* in32;  // buffer base (16-byte aligned)
  uint64_t nbytes;
  uint64_t bb;     // bit buffer, next bit at bit 0
  uint32_t nb;     // valid bits in bb
  uint64_t idx;    // next input dword to enter bb; absolute bit position of bb's bit 0 = 32 * idx - nb
  uint8_t* out;    // where output byte 0 goes
  uint64_t cap;    // bytes that may be stored at out
  uint64_t o;      // bytes produced so far
  uint64_t flushed;
  uint64_t ostart; // value of o at the start (0..15: out is the 16-byte aligned base below the first byte)
  uint32_t reach;  // how far back a match may go: min(32768, bytes produced + history preloaded into the ring)
  uint32_t unfl;   // o - flushed
  uint32_t oi;     // marker mode: ring index of output position o
  uint32_t* sym;   // marker mode: where the segment's 16-bit symbols go (pairs), or nullptr
  uint64_t sym_cap;  // symbols that fit there; sym_ovf is set once the segment has outgrown it
  uint32_t sym_ovf;
#ifdef WD_PROFILE
  uint64_t pt[6];  // cycles: lit/len entry, literal path, distance entry, copy, flush test; pt[5] tokens
#endif
  // per lane
  uint32_t vcur, vnxt;  // input dwords (idx & ~63) + lane (end-of-stream zeros applied) and + 64 + lane (as loaded)
};

enum { WD_OK = 0, WD_ANOMALY = 1, WD_NEEDS_HISTORY = 2, WD_FAR_NOSTORE = 3 };  // 3: short marker ring, a far match, and no symbol store to take it from

#define WD_SGPR(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#ifdef WD_PROFILE
__device__ unsigned long long* g_wd_dbg = nullptr;
void zes_wd_set_dbg(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wd_dbg), &p, sizeof p); }
#define WDT(var) const uint64_t var = (uint64_t)__builtin_readcyclecounter()
#define WDACC(i, a, b) d.pt[i] += (b) - (a)
#else
#define WDT(var)
#define WDACC(i, a, b)
#endif
// Placed right behind every lane-dependent branch or loop of the decoder: keeps the join of that branch a block
// of its own.  Without it the join is often folded into a block where the uniform reader state merges (loop
// header, end of an if), the uniformity analysis then calls that state divergent and the whole decoder runs
// on the vector unit.  (Checked with opt -passes='print<uniformity>' on the device IR.)
#define WD_JOIN() asm volatile("")

__device__ __forceinline__ static uint32_t wd_fetch_raw(const WaveDec& d, uint64_t dw) {
  // branch-free on purpose: a divergent branch here joins where the uniform reader state merges, and the
  // compiler then keeps that state in vector registers
  return d.in32[dw * 4 < d.nbytes ? dw : 0];  // the last dword may reach past nbytes inside its own aligned word: masked later
}
__device__ __forceinline__ static uint32_t wd_fetch_fix(const WaveDec& d, uint32_t raw, uint64_t dw) {  // zero beyond the end of the stream
  const uint64_t byte = dw * 4;
  const uint32_t keep = byte + 4 > d.nbytes ? (1u << (8u * ((uint32_t)(d.nbytes - byte) & 3u))) - 1u : 0xFFFFFFFFu;
  return byte < d.nbytes ? raw & keep : 0u;
}
__device__ __forceinline__ static uint64_t wd_pos(const WaveDec& d) { return d.idx * 32 - d.nb; }

__device__ __forceinline__ static void wd_seek(WaveDec& d, uint64_t bit) {
  const uint64_t dw = bit >> 5;
  const uint64_t base = dw & ~63ull;
  d.vcur = wd_fetch_fix(d, wd_fetch_raw(d, base + zes_lane()), base + zes_lane());
  d.vnxt = wd_fetch_raw(d, base + 64 + zes_lane());
  const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)d.vcur, (int)(dw & 63u));
  d.bb = (uint64_t)(w >> (bit & 31u));
  d.nb = 32u - (uint32_t)(bit & 31u);
  d.idx = dw + 1;
  if ((d.idx & 63u) == 0) {
    d.vcur = wd_fetch_fix(d, d.vnxt, d.idx + zes_lane());
    d.vnxt = wd_fetch_raw(d, d.idx + 64 + zes_lane());
  }
}
#define WD_UNLIKELY(x) __builtin_expect(!!(x), 0)
__device__ __forceinline__ static bool wd_refill(WaveDec& d) {  // guarantees nb >= 33; true if a dword was taken in
  const bool take = d.nb <= 32u;
  if (take) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)d.vcur, (int)((uint32_t)d.idx & 63u));
    d.bb |= (uint64_t)w << d.nb;
    d.nb += 32u;
    d.idx++;
    if (WD_UNLIKELY(((uint32_t)d.idx & 63u) == 0)) {  // the load issued one window ago is waited for here
      d.vcur = wd_fetch_fix(d, d.vnxt, d.idx + zes_lane());
      d.vnxt = wd_fetch_raw(d, d.idx + 64 + zes_lane());
    }
  }
  return take;
}
__device__ __forceinline__ static uint32_t wd_take(WaveDec& d, uint32_t k) {  // k <= 16, after wd_refill
  const uint32_t v = (uint32_t)d.bb & ((1u << k) - 1u);
  d.bb >>= k;
  d.nb -= k;
  return v;
}

template <class SM>
__device__ __forceinline__ static void wd_flush_range(SM& S, WaveDec& d, uint64_t from, uint64_t to) {
  // ring -> global for output offsets [from, to); from is a multiple of 16
  const uint32_t lane = zes_lane();
  for (uint64_t off = from + (uint64_t)lane * 16; off < to; off += 64 * 16) {
    const uint32_t r = (uint32_t)(off & (RING - 1));
    if (off + 16 <= to && off + 16 <= d.cap && off >= d.ostart) {
      *reinterpret_cast<uint4*>(d.out + off) = *reinterpret_cast<const uint4*>(&S.ring[r]);
    } else {  // a group shared with the neighbouring segment, or cut by the capacity: own bytes only
      for (uint32_t j = 0; j < 16; j++)
        if (off + j < to && off + j < d.cap && off + j >= d.ostart) d.out[off + j] = S.ring[(r + j) & (RING - 1)];
    }
  }
  WD_JOIN();
}
template <class SM>
__device__ __forceinline__ static void wd_produced(SM& S, WaveDec& d, uint32_t n) {  // n bytes were put into the ring
  d.o += n;
  d.reach = min(d.reach + n, ZES_WINDOW);
  d.unfl += n;
  while (WD_UNLIKELY(d.unfl >= FLUSH)) {
    wd_flush_range(S, d, d.flushed, d.flushed + FLUSH);
    d.flushed += FLUSH;
    d.unfl -= FLUSH;
  }
}

// base value and extra-bit count of length code lc (0..28) and distance code dc (0..29) in closed form
// (the values of src/const.ts:9-35)
__device__ __forceinline__ static void wd_len_code(uint32_t lc, uint32_t* base, uint32_t* xb) {
  const uint32_t e = lc < 8u ? 0u : (lc - 4u) >> 2;
  *base = lc < 8u ? 3u + lc : lc == 28u ? 258u : 3u + ((4u + (lc & 3u)) << e);
  *xb = lc == 28u ? 0u : e;
}
__device__ __forceinline__ static void wd_dist_code(uint32_t dc, uint32_t* base, uint32_t* xb) {
  const uint32_t e = dc < 4u ? 0u : (dc - 2u) >> 1;
  *base = dc < 4u ? 1u + dc : 1u + ((2u + (dc & 1u)) << e);
  *xb = e;
}

// Table entry of symbol s with code length l: everything the symbol loop needs in one word.
//   [4:0]  bits to take in all (code + extra bits)      [8:5] code length
//   [10:9] kind: 0 literal, 1 length / distance, 2 end of block, 3 a symbol that must not occur (286, 287, 30, 31)
//   [31:16] literal byte, or base value of the length / distance
#define WE_LIT 0u
#define WE_BASE 1u
#define WE_EOB 2u
#define WE_BAD 3u
__device__ __forceinline__ static uint32_t wd_entry(uint32_t s, uint32_t l, bool dist) {
  uint32_t kind, val = s, xb = 0;
  if (dist) {
    kind = s < 30u ? WE_BASE : WE_BAD;
    wd_dist_code(s < 30u ? s : 0u, &val, &xb);
  } else if (s < 256u) {
    kind = WE_LIT;
  } else if (s == 256u) {
    kind = WE_EOB;
  } else {
    kind = s < 286u ? WE_BASE : WE_BAD;
    wd_len_code(s < 286u ? s - 257u : 0u, &val, &xb);
  }
  if (kind != WE_BASE) xb = 0;
  return (l + xb) | (l << 5) | (kind << 9) | (val << 16);
}

// canonical tables + root LUT for one alphabet; lens in S.lens[base .. base+nsym).
// Returns false when the length set is over-subscribed.
template <class SM>
__device__ __noinline__ static bool wd_build(SM& S, uint32_t base, uint32_t nsym, uint32_t root, uint32_t* lut, uint16_t* syms,
                                             uint32_t* first, uint16_t* cnt, uint16_t* offs) {
  const uint32_t lane = zes_lane();
  const uint8_t* lens = S.lens + base;
  const bool dist = base != 0;
  for (uint32_t i = lane; i < (1u << root); i += 64) lut[i] = 0;
  // counts per length
  uint32_t c[16];
#pragma unroll
  for (int l = 0; l < 16; l++) c[l] = 0;
  for (uint32_t s0 = 0; s0 < nsym; s0 += 64) {
    const uint32_t s = s0 + lane;
    const uint32_t l = s < nsym ? lens[s] : 0u;
#pragma unroll
    for (int k = 1; k < 16; k++) c[k] += (uint32_t)__popcll(__ballot(l == (uint32_t)k));
  }
  uint32_t code = 0, off = 0, kraft = 0;
  uint32_t fst[16], ofs[16], run[16];
#pragma unroll
  for (int l = 1; l < 16; l++) {
    fst[l] = code;
    ofs[l] = off;
    run[l] = 0;
    code = (code + c[l]) << 1;
    off += c[l];
    kraft += c[l] << (15 - l);
  }
  if (kraft > 32768u) return false;
  if (lane < 16) {
    uint32_t f = 0, o2 = 0, cc = 0;
#pragma unroll
    for (int l = 1; l < 16; l++)
      if ((int)lane == l) {
        f = fst[l];
        o2 = ofs[l];
        cc = c[l];
      }
    first[lane] = f;
    offs[lane] = (uint16_t)o2;
    cnt[lane] = (uint16_t)cc;
  }
  // per symbol: rank inside its length (ballots, symbols ascending), canonical code, LUT fill / sorted-symbol slot
  for (uint32_t s0 = 0; s0 < nsym; s0 += 64) {
    const uint32_t s = s0 + lane;
    const uint32_t l = s < nsym ? lens[s] : 0u;
    uint32_t rank = 0, f = 0, o2 = 0;
#pragma unroll
    for (int k = 1; k < 16; k++) {
      const uint64_t m = __ballot(l == (uint32_t)k);
      if ((int)l == k) {
        rank = run[k] + (uint32_t)__popcll(m & zes_lanemask_lt());
        f = fst[k];
        o2 = ofs[k];
      }
      run[k] += (uint32_t)__popcll(m);
    }
    if (l) {
      syms[o2 + rank] = (uint16_t)s;
      if (l <= root) {
        const uint32_t rev = __brev(f + rank) >> (32u - l);
        const uint32_t ent = wd_entry(s, l, dist);
        for (uint32_t e = rev; e < (1u << root); e += 1u << l) lut[e] = ent;
      }
    }
  }
  return true;
}

// Entry of the next symbol when its code is longer than the root table: canonical walk (reference
// src/inflate.ts:238-252 extends one bit at a time the same way).  Returns 0 if no code matches.
__device__ __noinline__ static uint32_t wd_long(uint64_t bb, uint32_t root, const uint16_t* syms, const uint32_t* first, const uint16_t* cnt,
                                                const uint16_t* offs, bool dist) {
  uint32_t code = __brev((uint32_t)bb & ((1u << root) - 1u)) >> (32u - root);
  for (uint32_t len = root + 1; len <= 15u; len++) {
    code = (code << 1) | (uint32_t)((bb >> (len - 1)) & 1u);
    const uint32_t fl = first[len], cl = cnt[len];
    const uint32_t rel = code - fl;
    if (code >= fl && rel < cl) return wd_entry(syms[offs[len] + rel], len, dist);
  }
  return 0;
}

// i mod dist for i < 512, dist < 512 (overlapping matches): approximate float reciprocal plus one correction either way
__device__ __forceinline__ static uint32_t wd_mod(uint32_t i, uint32_t dist, float rcp) {
  const uint32_t q = (uint32_t)((float)i * rcp);
  int r = (int)i - (int)(q * dist);
  if (r < 0) r += (int)dist;
  if (r >= (int)dist) r -= (int)dist;
  return (uint32_t)r;
}

// Marker mode keeps the decoded symbols: every FLUSH symbols the ring's unflushed part goes to the segment's
// region of the symbol store (k_inf_seg_translate turns it into bytes once the windows are known).
template <class SM>
__device__ __forceinline__ static void wd_mark_flush(SM& S, WaveDec& d, uint32_t count) {  // the oldest `count` unflushed symbols
  const uint16_t* r16 = reinterpret_cast<const uint16_t*>(S.ring);
  constexpr uint32_t R = SM::kR16;
  const uint64_t first = d.o - d.unfl;  // symbol offset inside the segment; a multiple of the flush size
  if (d.sym == nullptr || first + count > d.sym_cap) {
    d.sym_ovf = 1;
  } else {
    uint32_t base = d.oi + R - d.unfl;  // ring index of the first unflushed symbol
    if (base >= R) base -= R;
    uint32_t* dst = d.sym + first / 2;
    for (uint32_t i2 = zes_lane(); i2 < (count + 1) / 2; i2 += 64) {
      uint32_t a = base + 2 * i2, b = a + 1;
      if (a >= R) a -= R;
      if (b >= R) b -= R;
      const uint32_t lo = r16[a], hi = 2 * i2 + 1 < count ? (uint32_t)r16[b] : 0u;
      dst[i2] = lo | (hi << 16);
    }
    WD_JOIN();
  }
  d.unfl -= count;
}
template <class SM>
__device__ __forceinline__ static void wd_mark_produced(SM& S, WaveDec& d, uint32_t n) {  // n symbols entered the ring
  d.o += n;
  d.unfl += n;  // never more than the flush size + 257, so the unflushed symbols are all still in the ring
  while (WD_UNLIKELY(d.unfl >= SM::kFlush)) wd_mark_flush(S, d, SM::kFlush);
}

// Decodes the symbols of one fixed/dynamic block whose tables are built.  Uniform control flow.
// MARK: 16-bit symbols in the marker ring (values >= 256 stand for bytes of the unknown window in
// front of the segment), nothing is stored and every distance is allowed.
template <bool MARK, class SM>
__device__ __forceinline__ static int wd_symbols_serial(SM& S, WaveDec& d) {
  const uint32_t lane = zes_lane();
  const uint64_t limit = d.nbytes * 8;
  // every token refills first, so the reader can run at most 64 bits + one dword ahead of a valid position;
  // the exact test against the end of the data is made at the end of the block
  const uint64_t idx_lim = d.nbytes / 4 + 3;
  uint16_t* r16 = reinterpret_cast<uint16_t*>(S.ring);
  // The two root tables move into vector registers (entry r * 64 + lane in register r): a lookup is then an
  // indexed register move plus a readlane, a handful of cycles, where an LDS read costs the lone wave a
  // round trip of ~200 cycles per symbol.
  uint32_t tl[(1u << LROOT) / 64], td[(1u << DROOT) / 64];
#pragma unroll
  for (uint32_t r = 0; r < (1u << LROOT) / 64; r++) tl[r] = S.lut_l[r * 64 + lane];
#pragma unroll
  for (uint32_t r = 0; r < (1u << DROOT) / 64; r++) td[r] = S.lut_d[r * 64 + lane];
  for (;;) {
    WDT(t0);
    if (WD_UNLIKELY(wd_refill(d) && d.idx > idx_lim)) return WD_ANOMALY;
    const uint32_t il = (uint32_t)d.bb & ((1u << LROOT) - 1u);
    uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)tl[il >> 6], (int)(il & 63u));
    if (WD_UNLIKELY((e & 31u) == 0)) {
      e = WD_SGPR(wd_long(d.bb, LROOT, S.syms_l, S.first_l, S.cnt_l, S.offs_l, false));
      if (e == 0) return WD_ANOMALY;
    }
    const uint32_t kind = (e >> 9) & 3u, tot = e & 31u, cl = (e >> 5) & 15u;
    WDT(t1);
    WDACC(0, t0, t1);
#ifdef WD_PROFILE
    d.pt[5]++;
#endif
    if (kind == WE_LIT) {
      d.bb >>= tot;
      d.nb -= tot;
      if (MARK) {
        if (lane == 0) r16[d.oi] = (uint16_t)(e >> 16);
        WD_JOIN();
        d.oi = d.oi + 1u == SM::kR16 ? 0u : d.oi + 1u;
        wd_mark_produced(S, d, 1);
      } else {
        if (lane == 0) S.ring[d.o & (RING - 1)] = (uint8_t)(e >> 16);
        WD_JOIN();
        wd_produced(S, d, 1);
      }
      WDT(t2);
      WDACC(1, t1, t2);
      continue;
    }
    if (WD_UNLIKELY(kind != WE_BASE)) {
      d.bb >>= tot;
      d.nb -= tot;
      return kind == WE_EOB && wd_pos(d) <= limit ? WD_OK : WD_ANOMALY;
    }
    const uint32_t len = (e >> 16) + (((uint32_t)(d.bb >> cl)) & ((1u << (tot - cl)) - 1u));  // code (<= 15) + extra (<= 5) bits fit one refill
    d.bb >>= tot;
    d.nb -= tot;
    if (WD_UNLIKELY(wd_refill(d) && d.idx > idx_lim)) return WD_ANOMALY;  // (both refills: either one may be the only one that ever takes a dword)
    const uint32_t id = (uint32_t)d.bb & ((1u << DROOT) - 1u);
    uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)td[id >> 6], (int)(id & 63u));
    if (WD_UNLIKELY((e2 & 31u) == 0)) {
      e2 = WD_SGPR(wd_long(d.bb, DROOT, S.syms_d, S.first_d, S.cnt_d, S.offs_d, true));
      if (e2 == 0) return WD_ANOMALY;
    }
    if (WD_UNLIKELY(((e2 >> 9) & 3u) != WE_BASE)) return WD_ANOMALY;
    const uint32_t tot2 = e2 & 31u, cl2 = (e2 >> 5) & 15u;
    const uint32_t dist = (e2 >> 16) + (((uint32_t)(d.bb >> cl2)) & ((1u << (tot2 - cl2)) - 1u));  // <= 15 + 13 bits
    d.bb >>= tot2;
    d.nb -= tot2;
    WDT(t3);
    WDACC(2, t1, t3);
    // lane-parallel copy; overlapping matches read i % dist so every source symbol already exists
    const bool overlap = dist < len;
    if (MARK) {
      constexpr uint32_t R = SM::kR16;
      if (R < RING16 && WD_UNLIKELY(dist > SM::kNear)) {
        // short ring, and the source lies (or may lie) behind it: the symbols come from the segment's store — flushed
        // long ago: the ring keeps far more than one flush interval — or, in front of the segment's first symbol, are
        // the marker values themselves (marker 256 + i stands for byte i of the 32 KiB in front of the segment)
        if (WD_UNLIKELY(d.sym_ovf || d.sym == nullptr)) {
          const int64_t s0 = (int64_t)d.o - (int64_t)dist;
          if (s0 + (int64_t)len > 0) return WD_FAR_NOSTORE;  // (all of it in front of the segment needs no store)
        }
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {  // (dist > len: no overlap)
          const uint32_t i = i0 + lane;
          const int64_t pos = (int64_t)d.o - (int64_t)dist + (int64_t)i;
          uint32_t v = 256u + (uint32_t)((int64_t)ZES_WINDOW + pos);  // pos < 0
          if (pos >= 0 && i < len) {
            const uint32_t w = __hip_atomic_load(&d.sym[(uint64_t)pos >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = (pos & 1) ? w >> 16 : w & 0xffffu;
          }
          uint32_t b = d.oi + i;
          if (b >= R) b -= R;
          if (i < len) r16[b] = (uint16_t)v;
          WD_JOIN();
        }
      } else {
      const uint32_t si = d.oi >= dist ? d.oi - dist : d.oi + R - dist;
      if (!overlap) {  // the common case, kept lean: the wave executes about one instruction per five cycles
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
          const uint32_t i = i0 + lane;
          uint32_t a = si + i, b = d.oi + i;
          if (a >= R) a -= R;
          if (b >= R) b -= R;
          if (i < len) r16[b] = r16[a];
          WD_JOIN();
        }
      } else {
        const float rcp = __builtin_amdgcn_rcpf((float)dist);
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
          const uint32_t i = i0 + lane;
          uint32_t a = si + wd_mod(i, dist, rcp), b = d.oi + i;
          if (a >= R) a -= R;
          if (b >= R) b -= R;
          if (i < len) r16[b] = r16[a];
          WD_JOIN();
        }
      }
      }
      d.oi += len;
      if (d.oi >= R) d.oi -= R;
      wd_mark_produced(S, d, len);
      continue;
    }
    if (WD_UNLIKELY(dist > d.reach)) return WD_NEEDS_HISTORY;  // behind the first output byte of the stream
    const uint32_t srcb = (uint32_t)(d.o - dist), dstb = (uint32_t)d.o;
    if (!overlap) {
      for (uint32_t i0 = 0; i0 < len; i0 += 64) {
        const uint32_t i = i0 + lane;
        if (i < len) S.ring[(dstb + i) & (RING - 1)] = S.ring[(srcb + i) & (RING - 1)];
        WD_JOIN();
      }
    } else {
      const float rcp = __builtin_amdgcn_rcpf((float)dist);
      for (uint32_t i0 = 0; i0 < len; i0 += 64) {
        const uint32_t i = i0 + lane;
        if (i < len) S.ring[(dstb + i) & (RING - 1)] = S.ring[(srcb + wd_mod(i, dist, rcp)) & (RING - 1)];
        WD_JOIN();
      }
    }
    WDT(t4);
    WDACC(3, t3, t4);
    wd_produced(S, d, len);
    WDT(t5);
    WDACC(4, t4, t5);
  }
}

// Decodes the symbols of one fixed/dynamic block whose tables are built — 64 bit positions at a time.
// A token's bits are found serially (where it starts is known only when the one before is decoded), but decoding the
// token that WOULD start at a given bit needs nothing but the bits: every lane decodes the token at its bit of a
// 64-bit window (root tables in LDS: one gather for the literal/length code, one for the distance code; 48 bits of
// lookahead from three dwords of the input window, which sits in two vector registers), and the real chain is then
// followed with lane reads — a few scalar instructions per token where the serial form (wd_symbols_serial, kept for
// the cycle-stamp build) pays two dependent table lookups and the refill logic per token: ~1000 cycles per token for
// a lone wave, five tokens per window on text.
// MARK: 16-bit symbols in the marker ring (values >= 256 stand for bytes of the unknown window in
// front of the segment), nothing is stored and every distance is allowed.
template <bool MARK, class SM>
__device__ __forceinline__ static int wd_symbols(SM& S, WaveDec& d) {
  const uint32_t lane = zes_lane();
  const uint64_t limit = d.nbytes * 8;
  uint16_t* r16 = reinterpret_cast<uint16_t*>(S.ring);
  uint64_t P = wd_pos(d);  // bit position of the next token (uniform)
  // input window of its own: dwords [B, B + 64) in vc (zeros beyond the end applied), [B + 64, B + 128) in vn (as loaded)
  uint64_t B = (P >> 5) & ~63ull;
  uint32_t vc = wd_fetch_fix(d, wd_fetch_raw(d, B + lane), B + lane), vn = wd_fetch_raw(d, B + 64 + lane);
  int rc = -1;
  while (rc < 0) {
    if (WD_UNLIKELY(P >= limit + 64)) {  // (a reader this far past the data has decoded zeros for a while)
      rc = WD_ANOMALY;
      break;
    }
    if (WD_UNLIKELY((P >> 5) - B >= 64)) {  // the window moves on: the load issued one window ago is waited for here
      B += 64;
      vc = wd_fetch_fix(d, vn, B + lane);
      vn = wd_fetch_raw(d, B + 64 + lane);
    }
    // ---- every lane: the token that would start at bit P + lane ----
    uint32_t tk, tv, tdist = 0, tb;
    {
      const uint64_t bpos = P + lane;
      const uint32_t rel = (uint32_t)((bpos >> 5) - B), sh = (uint32_t)bpos & 31u;  // rel <= 65
      uint32_t w[3];
#pragma unroll
      for (uint32_t k = 0; k < 3; k++) {
        const uint32_t q = rel + k;
        const uint32_t a = (uint32_t)__shfl((int)vc, (int)(q & 63u)), bq = (uint32_t)__shfl((int)vn, (int)(q & 63u));
        w[k] = q < 64u ? a : wd_fetch_fix(d, bq, B + q);
      }
      const uint64_t lo = ((uint64_t)w[1] << 32) | w[0];
      const uint64_t v = sh ? (lo >> sh) | ((uint64_t)w[2] << (64u - sh)) : lo;  // 64 bits from bit P + lane on
      uint32_t e = S.lut_l[(uint32_t)v & ((1u << LROOT) - 1u)];
      if (__ballot((e & 31u) == 0u)) {
        if ((e & 31u) == 0u) e = wd_long(v, LROOT, S.syms_l, S.first_l, S.cnt_l, S.offs_l, false);  // 0: no code
        WD_JOIN();
      }
      const uint32_t kind = (e >> 9) & 3u, tot = e & 31u, cl = (e >> 5) & 15u;
      tk = e == 0u ? WE_BAD : kind;
      tb = tot;
      tv = e >> 16;  // literal byte, or base of the length
      if (__ballot(tk == WE_BASE)) {
        const bool ism = tk == WE_BASE;
        tv += ((uint32_t)(v >> cl)) & ((1u << (tot - cl)) - 1u);  // (tot == cl for a literal: adds 0)
        const uint64_t v2 = v >> tot;
        uint32_t e2 = S.lut_d[(uint32_t)v2 & ((1u << DROOT) - 1u)];
        if (__ballot(ism && (e2 & 31u) == 0u)) {
          if (ism && (e2 & 31u) == 0u) e2 = wd_long(v2, DROOT, S.syms_d, S.first_d, S.cnt_d, S.offs_d, true);
          WD_JOIN();
        }
        const uint32_t tot2 = e2 & 31u, cl2 = (e2 >> 5) & 15u;
        const bool bad2 = e2 == 0u || ((e2 >> 9) & 3u) != WE_BASE;
        tdist = (e2 >> 16) + (((uint32_t)(v2 >> cl2)) & ((1u << (tot2 - cl2)) - 1u));  // <= 15 + 13 bits
        tk = (ism && bad2) ? WE_BAD : tk;
        tb += ism ? tot2 : 0u;
        WD_JOIN();
      }
    }
    // ---- the real chain through the window ----
    // (Tried: marking the window's tokens first, then one store for the literals between two matches and the matches in
    // order, ring bookkeeping once per window — 8.05 ms against 7.30 ms for the token-by-token form below on the
    // 64 MiB zlib text stream: the lone wave pays for the extra mask arithmetic more than it saves.)
    uint32_t o = 0;
    while (o < 64u) {
      const uint32_t kind = (uint32_t)__builtin_amdgcn_readlane((int)tk, (int)o);
      const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)tb, (int)o);
      if (WD_UNLIKELY(kind == WE_BAD || P + o + bits > limit)) {  // no such code, or its bits lie behind the data
        rc = WD_ANOMALY;
        break;
      }
      const uint32_t val = (uint32_t)__builtin_amdgcn_readlane((int)tv, (int)o);
      o += bits;
      if (kind == WE_LIT) {
        if (MARK) {
          if (lane == 0) r16[d.oi] = (uint16_t)val;
          WD_JOIN();
          d.oi = d.oi + 1u == SM::kR16 ? 0u : d.oi + 1u;
          wd_mark_produced(S, d, 1);
        } else {
          if (lane == 0) S.ring[d.o & (RING - 1)] = (uint8_t)val;
          WD_JOIN();
          wd_produced(S, d, 1);
        }
        continue;
      }
      if (WD_UNLIKELY(kind == WE_EOB)) {
        rc = WD_OK;
        break;
      }
      const uint32_t len = val;
      const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)tdist, (int)(o - bits));
      // lane-parallel copy; overlapping matches read i % dist so every source symbol already exists
      const bool overlap = dist < len;
      if (MARK) {
        constexpr uint32_t R = SM::kR16;
        if (R < RING16 && WD_UNLIKELY(dist > SM::kNear)) {
          // short ring, and the source lies (or may lie) behind it: the symbols come from the segment's store — flushed
          // long ago: the ring keeps far more than one flush interval — or, in front of the segment's first symbol, are
          // the marker values themselves (marker 256 + i stands for byte i of the 32 KiB in front of the segment)
          if (WD_UNLIKELY(d.sym_ovf || d.sym == nullptr)) {
            const int64_t s0 = (int64_t)d.o - (int64_t)dist;
            if (s0 + (int64_t)len > 0) {  // (all of it in front of the segment needs no store)
              rc = WD_FAR_NOSTORE;
              break;
            }
          }
          for (uint32_t i0 = 0; i0 < len; i0 += 64) {  // (dist > len: no overlap)
            const uint32_t i = i0 + lane;
            const int64_t pos = (int64_t)d.o - (int64_t)dist + (int64_t)i;
            uint32_t sv = 256u + (uint32_t)((int64_t)ZES_WINDOW + pos);  // pos < 0
            if (pos >= 0 && i < len) {
              const uint32_t wd2 = __hip_atomic_load(&d.sym[(uint64_t)pos >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              sv = (pos & 1) ? wd2 >> 16 : wd2 & 0xffffu;
            }
            uint32_t bx = d.oi + i;
            if (bx >= R) bx -= R;
            if (i < len) r16[bx] = (uint16_t)sv;
            WD_JOIN();
          }
        } else {
          const uint32_t si = d.oi >= dist ? d.oi - dist : d.oi + R - dist;
          if (!overlap) {
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
              const uint32_t i = i0 + lane;
              uint32_t ax = si + i, bx = d.oi + i;
              if (ax >= R) ax -= R;
              if (bx >= R) bx -= R;
              if (i < len) r16[bx] = r16[ax];
              WD_JOIN();
            }
          } else {
            const float rcp = __builtin_amdgcn_rcpf((float)dist);
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
              const uint32_t i = i0 + lane;
              uint32_t ax = si + wd_mod(i, dist, rcp), bx = d.oi + i;
              if (ax >= R) ax -= R;
              if (bx >= R) bx -= R;
              if (i < len) r16[bx] = r16[ax];
              WD_JOIN();
            }
          }
        }
        d.oi += len;
        if (d.oi >= R) d.oi -= R;
        wd_mark_produced(S, d, len);
        continue;
      }
      if (WD_UNLIKELY(dist > d.reach)) {  // behind the first output byte of the stream
        rc = WD_NEEDS_HISTORY;
        break;
      }
      const uint32_t srcb = (uint32_t)(d.o - dist), dstb = (uint32_t)d.o;
      if (!overlap) {
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
          const uint32_t i = i0 + lane;
          if (i < len) S.ring[(dstb + i) & (RING - 1)] = S.ring[(srcb + i) & (RING - 1)];
          WD_JOIN();
        }
      } else {
        const float rcp = __builtin_amdgcn_rcpf((float)dist);
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
          const uint32_t i = i0 + lane;
          if (i < len) S.ring[(dstb + i) & (RING - 1)] = S.ring[(srcb + wd_mod(i, dist, rcp)) & (RING - 1)];
          WD_JOIN();
        }
      }
      wd_produced(S, d, len);
    }
    P += o;
  }
  // the reader state of the bit-serial code around this loop (block headers, the stored-block path)
  wd_seek(d, P);
  return rc;
}

// One block starting at the reader's position (BFINAL bit).  *bfinal receives the flag.
template <bool MARK, class SM>
__device__ __forceinline__ static int wd_block(SM& S, WaveDec& d, uint32_t* bfinal) {
  const uint32_t lane = zes_lane();
  wd_refill(d);
  *bfinal = wd_take(d, 1);
  const uint32_t btype = wd_take(d, 2);
  if (btype == 3) return WD_ANOMALY;
  if (btype == 0) {  // stored (src/inflate.ts:42-55)
    const uint64_t bit = (wd_pos(d) + 7) & ~7ull;
    if (bit + 32 > d.nbytes * 8) return WD_ANOMALY;
    wd_seek(d, bit);
    wd_refill(d);
    const uint32_t LEN = wd_take(d, 16);
    wd_refill(d);
    const uint32_t NLEN = wd_take(d, 16);
    if (LEN + NLEN != 65535u) return WD_ANOMALY;
    const uint64_t src = wd_pos(d) >> 3;
    if (src + LEN > d.nbytes) return WD_ANOMALY;
    // (a lone wave moves ~1 GiB/s here whatever the width of its loads — four bytes per lane through a byte funnel
    // was slower than this loop, which the compiler pipelines; long stored runs want a copy kernel of their own)
    const uint8_t* in8 = reinterpret_cast<const uint8_t*>(d.in32);
    if (MARK) {
      uint16_t* r16 = reinterpret_cast<uint16_t*>(S.ring);
      for (uint32_t done = 0; done < LEN;) {
        const uint32_t n = min(LEN - done, SM::kFlush - d.unfl);  // unflushed symbols never exceed the flush size + 258: they stay inside the ring
        for (uint32_t i = lane; i < n; i += 64) {
          uint32_t b = d.oi + i;
          if (b >= SM::kR16) b -= SM::kR16;
          r16[b] = in8[src + done + i];
        }
        WD_JOIN();
        d.oi += n;
        if (d.oi >= SM::kR16) d.oi -= SM::kR16;
        wd_mark_produced(S, d, n);
        done += n;
      }
      wd_seek(d, (src + LEN) * 8);
      return WD_OK;
    }
    for (uint32_t done = 0; done < LEN;) {
      const uint32_t n = min(LEN - done, FLUSH - d.unfl);
      for (uint32_t i = lane; i < n; i += 64) S.ring[(d.o + i) & (RING - 1)] = in8[src + done + i];
      WD_JOIN();
      done += n;
      wd_produced(S, d, n);
    }
    wd_seek(d, (src + LEN) * 8);
    return WD_OK;
  }
  if (btype == 1) {  // fixed (src/huffman.ts:41-53; distance = 5 bits MSB-first, src/inflate.ts:107)
    for (uint32_t i = lane; i < 288; i += 64) S.lens[i] = (uint8_t)(i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8);
    WD_JOIN();
    if (lane < 32) S.lens[288 + lane] = 5;
    WD_JOIN();
  } else {  // dynamic header (src/inflate.ts:120-204)
    wd_refill(d);
    const uint32_t HLIT = wd_take(d, 5) + 257u;
    const uint32_t HDIST = wd_take(d, 5) + 1u;
    const uint32_t HCLEN = wd_take(d, 4) + 4u;
    uint32_t mycl = 0;  // lane s holds the length of code-length symbol s
    for (uint32_t k = 0; k < HCLEN; k++) {
      wd_refill(d);
      const uint32_t v = wd_take(d, 3);
      mycl = lane == kClOrder[k] ? v : mycl;
    }
    // 7-bit LUT of the code-length code
    for (uint32_t i = lane; i < 128; i += 64) S.cl_lut[i] = 0;
    WD_JOIN();
    uint32_t kraft = 0;
    {
      uint32_t code = 0;
      for (uint32_t l = 1; l <= 7; l++) {
        const uint64_t m = __ballot(lane < 19 && mycl == l);
        if (lane < 19 && mycl == l) {
          const uint32_t rank = (uint32_t)__popcll(m & zes_lanemask_lt());
          const uint32_t rev = __brev(code + rank) >> (32u - l);
          for (uint32_t e = rev; e < 128; e += 1u << l) S.cl_lut[e] = (uint8_t)(lane | (l << 5));
        }
        WD_JOIN();
        const uint32_t n = (uint32_t)__popcll(m);
        kraft += n << (7 - l);
        code = (code + n) << 1;
      }
    }
    if (kraft > 128u) return WD_ANOMALY;
    for (uint32_t i = lane; i < 352; i += 64) S.lens[i] = 0;
    WD_JOIN();
    const uint32_t total = HLIT + HDIST;
    uint32_t prev = 0;
    for (uint32_t k = 0; k < total;) {
      wd_refill(d);
      const uint32_t e = WD_SGPR(S.cl_lut[(uint32_t)d.bb & 127u]);
      const uint32_t l = e >> 5, sy = e & 31u;
      if (!l) return WD_ANOMALY;
      wd_take(d, l);
      uint32_t rep = 1, val = sy;
      if (sy == 16) {
        if (k == 0) return WD_ANOMALY;
        rep = 3 + wd_take(d, 2);
        val = prev;
      } else if (sy == 17) {
        rep = 3 + wd_take(d, 3);
        val = 0;
      } else if (sy == 18) {
        rep = 11 + wd_take(d, 7);
        val = 0;
      }
      if (k + rep > total) return WD_ANOMALY;  // the reference spills into the distance table: T4
      if (val && lane < rep) {
        const uint32_t idx = k + lane;
        S.lens[idx < HLIT ? idx : 288 + (idx - HLIT)] = (uint8_t)val;
      }
      WD_JOIN();
      // the reference keeps `codelen` across 17/18 as 0 (src/inflate.ts:172-183)
      prev = val;
      k += rep;
    }
    if (wd_pos(d) > d.nbytes * 8) return WD_ANOMALY;
  }
  // (a call returns in a vector register: back to scalar, or everything behind the branch counts as divergent)
  if (!WD_SGPR(wd_build(S, 0, 288, LROOT, S.lut_l, S.syms_l, S.first_l, S.cnt_l, S.offs_l))) return WD_ANOMALY;
  if (!WD_SGPR(wd_build(S, 288, 32, DROOT, S.lut_d, S.syms_d, S.first_d, S.cnt_d, S.offs_d))) return WD_ANOMALY;
  return wd_symbols<MARK>(S, d);
}


// T3: a single wavefront decodes a whole stream from bit 16; work item = entry of the job table (several
// streams of a batch call decode side by side).
__global__ __launch_bounds__(64) void k_inf_decode(const uint8_t* __restrict__ d_in, uint8_t* __restrict__ d_out,
                                                   const ZesInfBuf* __restrict__ jobs, ZesRes* __restrict__ res_all,
                                                   uint64_t* __restrict__ resume_all) {
  __shared__ __align__(16) InfSmem S;
  const uint32_t lane = threadIdx.x;
  const ZesInfBuf jb = jobs[blockIdx.x];
  const uint64_t in_off = jb.in_off, c = jb.c, out_off = jb.out_off, cap = jb.cap;
  ZesRes* res = res_all + blockIdx.x;
  uint64_t* resume = resume_all + 2 * (size_t)blockIdx.x;
  WaveDec d;
  d.in32 = reinterpret_cast<const uint32_t*>(d_in + in_off);
  d.nbytes = c;
  d.o = 0;
  d.flushed = 0;
  d.out = d_out + out_off;
  d.cap = cap;
  d.ostart = 0;
  d.reach = 0;
  d.unfl = 0;
  d.sym = nullptr;
  d.sym_cap = 0;
  d.sym_ovf = 0;
  d.oi = 0;
#ifdef WD_PROFILE
  for (int i = 0; i < 6; i++) d.pt[i] = 0;
  const uint64_t tk0 = (uint64_t)__builtin_readcyclecounter();
#endif
  wd_seek(d, 16);
  uint32_t bfinal = 0;
  int rc = WD_OK;
  uint64_t blk_bit = 16, blk_out = 0;
  while (!bfinal) {
    blk_bit = wd_pos(d);
    blk_out = d.o;
    rc = wd_block<false>(S, d, &bfinal);
    if (rc != WD_OK) break;
    if (!bfinal && wd_pos(d) >= d.nbytes * 8) {  // stream exhausted without a final block: T4 decides
      rc = WD_ANOMALY;  // resume T4 at the start of the block just decoded so its reader state is exact
      break;
    }
  }
#ifdef WD_PROFILE
  if (g_wd_dbg && lane == 0) {
    for (int i = 0; i < 6; i++) g_wd_dbg[i] = d.pt[i];
    g_wd_dbg[6] = (uint64_t)__builtin_readcyclecounter() - tk0;
  }
#endif
  if (rc == WD_OK) {
    wd_flush_range(S, d, d.flushed, d.o);
    if (lane == 0) {
      res->out_len = d.o;
      res->status = 0;
      res->aux = 3;  // tier
    }
  } else {
    // hand the failing block to the exact decoder: everything before it is valid output
    wd_flush_range(S, d, d.flushed, blk_out);
    if (lane == 0) {
      resume[0] = blk_bit;
      resume[1] = blk_out;
      res->status = 1;  // continue with T4
      res->out_len = blk_out;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Streams of stored blocks only (another encoder on incompressible data, or at level 0): nothing to decode and no
// dynamic header to cut at.  One lane walks the block headers (src/inflate.ts:42-55: byte-aligned LEN / NLEN with
// LEN + NLEN == 65535), then one workgroup per block copies.  Anything else in the chain — another block type, a bad
// NLEN, data that ends early — and the walk reports "not mine": the other tiers take the stream.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_inf_stored_walk(const uint8_t* __restrict__ d_in, uint64_t in_off, uint64_t c, uint64_t cap_entries,
                                                        ZesStoredBlk* __restrict__ list, ZesRes* __restrict__ res) {
  // One wavefront; lane 0 walks.  Where a block starts is known only from the block before, so a step is a trip to
  // memory (~1 us from HBM: 1.2 ms for the 3000 blocks of a 48 MiB zlib stream).  The other lanes shorten the trips:
  // an encoder's stored blocks are all about one size, so while header k+1 is on its way the wave touches the lines
  // around where headers k+2 .. k+5 will be if the sizes repeat (256 bytes each): by the time the walk gets there they
  // sit in the cache.  (Looking 64 blocks ahead and taking the ones whose guess holds was slower: zlib's stored
  // blocks differ by a few bytes.)
  if (blockIdx.x != 0) return;
  const uint32_t lane = threadIdx.x;
  const uint8_t* in = d_in + in_off;
  if (lane == 0) {
    res->status = 1;
    res->out_len = 0;
    res->aux = 0;
  }
  uint64_t pos = 2, dst = 0, n = 0;  // the first block header sits at bit 16, every later one behind a whole byte
  uint32_t sink = 0;
  for (;;) {
    if (pos + 5 > c || n >= cap_entries) break;
    uint32_t h = 0, LEN = 0, NLEN = 0;
    if (lane == 0) {
      h = in[pos];
      LEN = in[pos + 1] | ((uint32_t)in[pos + 2] << 8);
      NLEN = in[pos + 3] | ((uint32_t)in[pos + 4] << 8);
    }
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    LEN = (uint32_t)__builtin_amdgcn_readfirstlane((int)LEN);
    NLEN = (uint32_t)__builtin_amdgcn_readfirstlane((int)NLEN);
    if (((h >> 1) & 3u) || LEN + NLEN != 65535u || pos + 5 + LEN > c) break;  // not a stored block (or the data ends early)
    {
      // lanes 16q .. 16q+15: the 256 bytes around where header k + 2 + q is expected
      const uint64_t guess = pos + (uint64_t)(2u + (lane >> 4)) * (5ull + LEN);
      const uint64_t a = (guess & ~127ull) + 16ull * (lane & 15u) - 64ull;
      if (a + 4 <= c && guess < c) sink += *reinterpret_cast<const uint32_t*>(in + (a & ~3ull));
    }
    if (lane == 0) {
      ZesStoredBlk b;
      b.src = pos + 5;
      b.dst = dst;
      b.len = LEN;
      b.pad = 0;
      list[n] = b;
    }
    n++;
    dst += LEN;
    pos += 5 + (uint64_t)LEN;
    if (h & 1u) {  // BFINAL
      if (lane == 0) {
        res->out_len = dst;
        res->aux = (uint32_t)n;
        res->status = 0;
      }
      break;
    }
  }
  if (sink == 0x9E3779B9u && lane == 63) res->aux ^= 0u;  // (keeps the touches alive)
}

// ------------------------------------------------------------------------------------------
// The same walk in parallel, for streams of up to 128 MiB (round 3).  A stored block's header is recognisable on its
// own — a byte with BTYPE 00 followed by LEN and its complement — so every byte position is tested (k_inf_stored_find:
// per 16 KiB of stream up to STORED_SLOTS hits, in order; one in 2^18 positions passes by chance), and one workgroup
// (k_inf_stored_rank) finds the hits that are on the chain from the first header: next[i] = the hit at position
// P[i] + 5 + LEN[i] (binary search), reachability from hit 0 by pointer doubling, a block's number = reachable hits in
// front of it, its place in the output = P - 2 - 5 * number.  Anything irregular (too many hits, no hit at bit 16, a
// chain that does not end on a final block) and the serial walk decides.
// ------------------------------------------------------------------------------------------
#define STORED_CHUNK 16384u
#define STORED_SLOTS 8u
#define STORED_MAXN 8192u
__global__ __launch_bounds__(256) void k_inf_stored_find(const uint8_t* __restrict__ d_in, uint64_t in_off, uint32_t c,
                                                         uint32_t* __restrict__ slots, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_hit[STORED_CHUNK / 32];
  const uint32_t tid = threadIdx.x;
  const uint8_t* in = d_in + in_off;
  const uint32_t base = blockIdx.x * STORED_CHUNK;
  // 64 positions per thread, from 17 dwords (the stream starts on a 16-byte boundary)
  const uint32_t p0 = base + tid * 64u;
  uint32_t w[18];
#pragma unroll
  for (uint32_t k = 0; k < 18; k++) {
    const uint64_t a = (uint64_t)p0 + 4ull * k;
    w[k] = a + 4 <= c ? *reinterpret_cast<const uint32_t*>(in + a) : 0u;
    if (a < c && a + 4 > c)
      for (uint32_t q = 0; q < 4u && a + q < c; q++) w[k] |= (uint32_t)in[a + q] << (8u * q);
  }
  uint32_t m0 = 0, m1 = 0;
#pragma unroll
  for (uint32_t k = 0; k < 64; k++) {
    const uint32_t i = k >> 2, sh = (k & 3u) * 8u;
    const uint32_t hb = (w[i] >> sh) & 255u;
    const uint32_t k1 = k + 1u;
    const uint32_t v = __builtin_amdgcn_alignbyte(w[(k1 >> 2) + 1u], w[k1 >> 2], k1 & 3u);  // bytes p+1 .. p+4
    const uint32_t LEN = v & 0xFFFFu;
    const bool hit = (hb & 6u) == 0u && ((v >> 16) ^ LEN) == 0xFFFFu && (uint64_t)p0 + k >= 2u && (uint64_t)p0 + k + 5u + LEN <= c;
    if (k < 32u) m0 |= hit ? 1u << k : 0u; else m1 |= hit ? 1u << (k - 32u) : 0u;
  }
  // hits are rare (a true header per stored block, one position in 2^18 by chance): a thread that has any counts the
  // hits of the threads below it and writes its own behind them, in order
  const uint32_t mycnt = (uint32_t)(__popc(m0) + __popc(m1));
  s_hit[tid] = mycnt;
  __syncthreads();
  if (mycnt) {
    uint32_t n = 0;
    for (uint32_t t = 0; t < tid; t++) n += s_hit[t];
    for (uint32_t half = 0; half < 2u; half++) {
      uint32_t m = half ? m1 : m0;
      while (m) {
        if (n < STORED_SLOTS) slots[(size_t)blockIdx.x * STORED_SLOTS + n] = p0 + 32u * half + (uint32_t)__builtin_ctz(m);
        n++;
        m &= m - 1u;
      }
    }
  }
  if (tid == 255) {
    uint32_t n = 0;
    for (uint32_t t = 0; t < 256; t++) n += s_hit[t];
    counts[blockIdx.x] = n;
  }
}

__global__ __launch_bounds__(1024) void k_inf_stored_rank(const uint8_t* __restrict__ d_in, uint64_t in_off, uint32_t c,
                                                          const uint32_t* __restrict__ slots, const uint32_t* __restrict__ counts,
                                                          uint32_t nchunks, uint64_t cap_entries, ZesStoredBlk* __restrict__ list,
                                                          ZesRes* __restrict__ res) {
  __shared__ uint32_t P[STORED_MAXN];      // positions of the hits, ascending
  __shared__ uint16_t L[STORED_MAXN];      // their LEN
  __shared__ uint16_t J[2][STORED_MAXN];   // 2^r-th successor (0xFFFF: none)
  __shared__ uint32_t R[STORED_MAXN / 32], F[STORED_MAXN / 32];  // reachable from hit 0; final block
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_n, s_bad;
  const uint32_t tid = threadIdx.x;
  const uint8_t* in = d_in + in_off;
  if (tid == 0) {
    res->status = 1;
    res->out_len = 0;
    res->aux = 0;
    s_bad = 0;
  }
  // hits per chunk -> offsets (a chunk range per thread)
  const uint32_t per = (nchunks + 1023u) / 1024u;
  const uint32_t c0 = min(nchunks, tid * per), c1 = min(nchunks, c0 + per);
  uint32_t mine = 0, over = 0;
  for (uint32_t k = c0; k < c1; k++) {
    const uint32_t n = counts[k];
    over |= n > STORED_SLOTS ? 1u : 0u;
    mine += min(n, STORED_SLOTS);
  }
  s_scan[tid] = mine;
  __syncthreads();
  if (over) atomicOr(&s_bad, 1u);
  if (tid == 0) {
    uint32_t run = 0;
    for (uint32_t t = 0; t < 1024; t++) {
      const uint32_t v = s_scan[t];
      s_scan[t] = run;
      run += v;
    }
    s_n = run;
  }
  __syncthreads();
  const uint32_t n = s_n;
  if (s_bad || n == 0 || n > STORED_MAXN || (uint64_t)n > cap_entries) return;
  {
    uint32_t o = s_scan[tid];
    for (uint32_t k = c0; k < c1; k++) {
      const uint32_t m = min(counts[k], STORED_SLOTS);
      for (uint32_t q = 0; q < m; q++) {
        const uint32_t p = slots[(size_t)k * STORED_SLOTS + q];
        P[o] = p;
        L[o] = (uint16_t)((uint32_t)in[p + 1] | ((uint32_t)in[p + 2] << 8));
        o++;
      }
    }
  }
  for (uint32_t i = tid; i < STORED_MAXN / 32; i += 1024) {
    R[i] = 0;
    F[i] = 0;
  }
  __syncthreads();
  if (P[0] != 2u) return;  // (uniform) the first block is not a stored one
  // successor of every hit, final flags
  for (uint32_t i = tid; i < n; i += 1024) {
    const uint32_t want = P[i] + 5u + (uint32_t)L[i];
    uint32_t lo = i + 1, hi = n;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (P[mid] < want) lo = mid + 1; else hi = mid;
    }
    J[0][i] = (lo < n && P[lo] == want) ? (uint16_t)lo : (uint16_t)0xFFFFu;
    if (in[P[i]] & 1u) atomicOr(&F[i >> 5], 1u << (i & 31u));
  }
  if (tid == 0) R[0] = 1u;
  __syncthreads();
  // a final block ends the chain: nothing behind it is reached through it
  for (uint32_t i = tid; i < n; i += 1024)
    if ((F[i >> 5] >> (i & 31u)) & 1u) J[0][i] = 0xFFFFu;
  __syncthreads();
  // reachability by doubling: after round r everything within 2^(r+1) - 1 steps of hit 0 is marked
  uint32_t cur = 0;
  for (uint32_t r = 0; (1u << r) < n; r++) {
    for (uint32_t i = tid; i < n; i += 1024) {
      const uint32_t j = J[cur][i];
      if (j != 0xFFFFu && ((R[i >> 5] >> (i & 31u)) & 1u)) atomicOr(&R[j >> 5], 1u << (j & 31u));
      J[cur ^ 1][i] = j != 0xFFFFu ? J[cur][j] : (uint16_t)0xFFFFu;
    }
    cur ^= 1;
    __syncthreads();
  }
  // block numbers = reachable hits in front (eight hits per thread), the list, the result
  const uint32_t i0 = tid * 8u;
  uint32_t cnt = 0;
  for (uint32_t q = 0; q < 8u; q++) {
    const uint32_t i = i0 + q;
    cnt += (i < n && ((R[i >> 5] >> (i & 31u)) & 1u)) ? 1u : 0u;
  }
  s_scan[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (uint32_t t = 0; t < 1024; t++) {
      const uint32_t v = s_scan[t];
      s_scan[t] = run;
      run += v;
    }
    s_n = run;  // blocks on the chain
  }
  __syncthreads();
  const uint32_t nblk = s_n;
  uint32_t rank = s_scan[tid];
  for (uint32_t q = 0; q < 8u; q++) {
    const uint32_t i = i0 + q;
    if (i >= n || !((R[i >> 5] >> (i & 31u)) & 1u)) continue;
    ZesStoredBlk b;
    b.src = (uint64_t)P[i] + 5u;
    b.dst = (uint64_t)P[i] - 2u - 5ull * rank;
    b.len = L[i];
    b.pad = 0;
    list[rank] = b;
    if (rank + 1u == nblk) {  // the last block of the chain must be the stream's final one
      if ((F[i >> 5] >> (i & 31u)) & 1u) {
        res->out_len = b.dst + b.len;
        res->aux = nblk;
        res->status = 0;
      }
    }
    rank++;
  }
}

__global__ __launch_bounds__(256) void k_inf_stored_copy(const uint8_t* __restrict__ d_in, uint64_t in_off, uint8_t* __restrict__ d_out,
                                                         uint64_t out_off, const ZesStoredBlk* __restrict__ list) {
  const ZesStoredBlk b = list[blockIdx.x];
  const uint8_t* src = d_in + in_off + b.src;
  uint8_t* dst = d_out + out_off + b.dst;
  const uint32_t len = b.len, tid = threadIdx.x;
  // head up to the first 4-byte boundary of the destination, aligned dwords (source through a byte funnel), tail
  const uint32_t head = min(len, (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u));
  if (tid < head) dst[tid] = src[tid];
  const uint32_t ndw = (len - head) >> 2;
  const uint8_t* s0 = src + head;
  const uint32_t sh = (uint32_t)((uintptr_t)s0 & 3u);
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s0 - sh);  // aligned view: dword i + 1 may lie behind the block, inside the stream
  uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
  for (uint32_t i = tid; i < ndw; i += 256) d32[i] = sh ? __builtin_amdgcn_alignbyte(s32[i + 1], s32[i], sh) : s32[i];
  const uint32_t done = head + ndw * 4u;
  if (tid < len - done) dst[done + tid] = src[done + tid];
}

// ------------------------------------------------------------------------------------------
// T2, segment-parallel decode of any valid stream.  Work item 0 starts at bit 16, work item
// w > 0 at candidate w-1 (sorted).  A segment runs block after block (any BTYPE) until it lands
// exactly on a candidate position, passes a final block, or fails; false candidates make
// segments nobody chains to.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ static uint32_t seg_ring_idx(uint32_t oi, uint32_t i) {  // full ring: index of position o - 32768 + i
  uint32_t a = oi + (RING16 - ZES_WINDOW) + i;
  if (a >= RING16) a -= RING16;
  if (a >= RING16) a -= RING16;
  return a;
}

// T2 keeps at most one candidate per bucket of the compressed stream (the first one): the list comes out sorted
// without a sort, and the number of segments — each costs a 64 KiB map and a 32 KiB window — stays bounded
// however small the encoder made its blocks.  A dropped candidate only makes a segment longer.
__global__ __launch_bounds__(256) void k_inf_cand_bucket(const uint32_t* __restrict__ cand, const uint32_t* __restrict__ cnt,
                                                         uint32_t cand_cap, uint32_t bucket_bits, uint32_t* __restrict__ bmin) {
  const uint32_t n = min(cnt[0], cand_cap);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t v = cand[i];
    atomicMin(&bmin[min(v / bucket_bits, SEG_BUCKETS - 1u)], v);
  }
}
__global__ __launch_bounds__(1024) void k_inf_cand_compact(const uint32_t* __restrict__ bmin, uint32_t* __restrict__ out,
                                                           uint32_t* __restrict__ out_count) {
  __shared__ uint32_t s_wave[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t PER = SEG_BUCKETS / 1024u;  // <= 4: the ballots below scan three bits of the count
  uint32_t v[PER], have = 0;
#pragma unroll
  for (uint32_t q = 0; q < PER; q++) {
    v[q] = bmin[tid * PER + q];
    have += v[q] != 0xFFFFFFFFu;
  }
  // exclusive prefix of `have` over the workgroup: in-wave by ballots of each count bit, then across waves
  uint32_t pre = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) pre += (uint32_t)__popcll(__ballot((have >> b) & 1u) & zes_lanemask_lt()) << b;
  uint32_t wtot = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) wtot += (uint32_t)__popcll(__ballot((have >> b) & 1u)) << b;
  if (lane == 0) s_wave[wave] = wtot;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (uint32_t k = 0; k < 16; k++) {
    if (k < wave) base += s_wave[k];
    total += s_wave[k];
  }
  uint32_t o = base + pre;
#pragma unroll
  for (uint32_t q = 0; q < PER; q++)
    if (v[q] != 0xFFFFFFFFu) out[o++] = v[q];
  if (tid == 0) *out_count = total;
}

// Launch order of the work items: by compressed span to the next work item, longest first (the span is the only
// estimate of a segment's decode time there is before decoding it).  Rank sort in LDS; identity for huge lists.
#define SEGORDER_LDS 12288u
__global__ __launch_bounds__(1024) void k_inf_seg_order(const ZesSegJob* __restrict__ jobs, const uint32_t* __restrict__ cand_all,
                                                        uint32_t* __restrict__ order_all) {
  // one workgroup per buffer of the group
  const ZesSegJob jb = jobs[blockIdx.x];
  const uint32_t* cand = cand_all + jb.cand_base;
  const uint32_t ncand = jb.ncand;
  const uint64_t c = jb.c;
  uint32_t* order = order_all + jb.work_first;
  __shared__ uint32_t s_span[SEGORDER_LDS];
  const uint32_t nwork = ncand + 1, tid = threadIdx.x;
  if (nwork > SEGORDER_LDS) {
    for (uint32_t w = tid; w < nwork; w += 1024) order[w] = w;
    return;
  }
  const uint32_t endbit = (uint32_t)min(c * 8, (uint64_t)0xFFFFFFFFu);
  for (uint32_t w = tid; w < nwork; w += 1024) {
    const uint32_t st = w ? cand[w - 1] : 0u;
    const uint32_t nx = w < ncand ? cand[w] : endbit;
    s_span[w] = (w > 0 && st + 16u == jb.start0) ? 0u : nx - st;  // the duplicate of work item 0 returns at once
  }
  __syncthreads();
  if (tid == 0 && ncand > 0 && cand[0] + 16u == jb.start0) s_span[0] = ncand > 1 ? cand[1] : endbit;
  __syncthreads();
  for (uint32_t w = tid; w < nwork; w += 1024) {
    const uint32_t v = s_span[w];
    uint32_t r = 0;
    for (uint32_t j = 0; j < nwork; j++) {
      const uint32_t u = s_span[j];
      r += (u > v) || (u == v && j < w);
    }
    order[r] = w;
  }
}

template <class SM>
__device__ __forceinline__ static void seg_scan_body(SM& S, const uint8_t* __restrict__ d_in, const ZesSegJob* __restrict__ jobs, uint32_t njobs,
                                                     const uint32_t* __restrict__ cand_all, ZesSegRes* __restrict__ sres_all,
                                                     uint32_t* __restrict__ maps_all, uint32_t* __restrict__ sym16_all, uint32_t sym_ratio,
                                                     const uint32_t* __restrict__ order, uint32_t* __restrict__ far_nostore,
                                                     const uint32_t* __restrict__ nlive, uint64_t* __restrict__ symoff_all) {
  constexpr uint32_t R = SM::kR16;
  const uint32_t lane = threadIdx.x;
  // (launch behind k_inf_seg_block_par: order[] lists the items that kernel left — numbered over the whole group —
  // and *nlive is how many)
  if (nlive && blockIdx.x >= *nlive) return;
  const uint32_t gid = nlive ? order[blockIdx.x] : blockIdx.x;
  // buffer of this work item: the last one whose first work item is <= gid
  uint32_t bi = 0;
  {
    uint32_t lo = 0, hi = njobs;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (jobs[mid].work_first <= gid) lo = mid; else hi = mid;
    }
    bi = lo;
  }
  const ZesSegJob jb = jobs[bi];
  const uint64_t in_off = jb.in_off, c = jb.c;
  const uint32_t ncand = jb.ncand;
  const uint32_t* cand = cand_all + jb.cand_base;
  ZesSegRes* sres = sres_all + jb.work_first;
  uint32_t* maps = maps_all + (size_t)jb.work_first * (ZES_WINDOW / 2);
  uint32_t* sym16 = sym16_all + jb.sym_base;
  const uint32_t w = nlive ? gid - jb.work_first : order[blockIdx.x];  // (a buffer's part of order[] holds its own work items) longest compressed span first
  ZesSegRes r;
  r.end_bit = 0;
  r.out_len = 0;
  r.flags = 0;
  r.next = 0;
  uint64_t start = jb.start0;
  if (w > 0) {
    const uint32_t c0 = cand[w - 1];
    if (c0 + 16u == jb.start0) {  // the stream start is work item 0 already
      if (lane == 0) sres[w] = r;
      WD_JOIN();
      return;
    }
    start = (uint64_t)c0 + 16;
  }
  WaveDec d;
  d.in32 = reinterpret_cast<const uint32_t*>(d_in + in_off);
  d.nbytes = c;
  d.o = 0;
  d.flushed = 0;
  d.out = nullptr;
  d.cap = 0;
  d.ostart = 0;
  d.reach = 0;
  d.unfl = 0;
  d.oi = R >= RING16 ? ZES_WINDOW : 0u;
  // symbol store: the work item that starts at compressed byte b owns symbols [b * ratio, b' * ratio), b' the start of
  // the next work item (the end of the stream for the last one); a segment that outgrows its share is decoded twice
  {
    uint32_t nx = w;  // candidate that starts the next work item (candidate 0 at bit 16 duplicates work item 0)
    if (w == 0 && ncand > 0 && cand[0] + 16u == jb.start0) nx = 1;
    const uint64_t b0 = (start - 16) / 8, b1 = nx < ncand ? (uint64_t)cand[nx] / 8 : c;
    d.sym = sym_ratio ? sym16 + b0 * sym_ratio / 2 : nullptr;
    if (lane == 0) symoff_all[jb.work_first + w] = jb.sym_base + b0 * sym_ratio / 2;  // (where k_inf_seg_translate finds the symbols)
    d.sym_cap = (b1 - b0) * sym_ratio;
    d.sym_ovf = sym_ratio ? 0u : 1u;
  }
  uint16_t* r16 = reinterpret_cast<uint16_t*>(S.ring);
  // an item the block decoder has decoded up to the end of its block (flags 8: the block behind it is not on the list)
  // is taken over from there: its symbols are in the store, the ring gets the last R of them
  const ZesSegRes prev = sres[w];
  const bool resume = nlive != nullptr && R < RING16 && prev.flags == 8u && d.sym != nullptr && prev.out_len <= d.sym_cap;  // (uniform)
  if (R >= RING16) {
    for (uint32_t i = lane; i < ZES_WINDOW; i += 64) r16[i] = (uint16_t)(256u + i);  // window byte i of the previous segment
  } else if (!resume) {  // short ring: the last R bytes of that window, output position 0 at ring index 0 (= R)
    for (uint32_t i = lane; i < R; i += 64) r16[i] = (uint16_t)(256u + (ZES_WINDOW - R) + i);
  } else {  // ring index of output position p: p mod R
    const uint64_t L = prev.out_len;
    for (uint32_t i = lane; i < R; i += 64) {
      const int64_t pos = (int64_t)L - (int64_t)R + (int64_t)i;  // the R positions in front of L, oldest first
      uint32_t v;
      if (pos < 0) {
        v = 256u + (uint32_t)((int64_t)ZES_WINDOW + pos);
      } else {
        const uint32_t wd = __hip_atomic_load(&d.sym[(uint64_t)pos >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = (pos & 1) ? wd >> 16 : wd & 0xffffu;
      }
      r16[(uint32_t)(((uint64_t)(pos + (int64_t)R * 4)) % R)] = (uint16_t)v;  // (pos >= -R)
    }
    d.o = L;
    d.flushed = L & ~1ull;
    d.unfl = (uint32_t)(L & 1ull);  // symbols are stored in pairs: an odd count's last symbol is flushed again with its successor
    d.oi = (uint32_t)(L % R);
    start = prev.end_bit;
  }
  wd_seek(d, start);
  uint32_t bfinal = 0;
  int rc;
  for (;;) {
    rc = wd_block<true>(S, d, &bfinal);
    if (rc != WD_OK || bfinal) break;
    if (wd_pos(d) >= d.nbytes * 8) {
      rc = WD_ANOMALY;
      break;
    }
    // does the next block start on a candidate?  (uniform binary search)
    const uint64_t want = wd_pos(d) - 16;
    uint32_t lo = 0, hi = ncand;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((uint64_t)cand[mid] < want) lo = mid + 1; else hi = mid;
    }
    if (lo < ncand && (uint64_t)cand[lo] == want) {
      r.next = lo + 1;
      break;
    }
  }
  if (rc != WD_OK) {
    if (rc == WD_FAR_NOSTORE && lane == 0) atomicAdd(far_nostore, 1u);  // the host runs the group again with the full ring
    if (lane == 0) sres[w] = r;
    return;
  }
  // the last 32 Ki symbols of the output so far (for a short segment they start inside the previous window)
  uint32_t* mp = maps + (size_t)w * (ZES_WINDOW / 2);
  if (R >= RING16) {
    for (uint32_t i2 = lane; i2 < ZES_WINDOW / 2; i2 += 64) {
      const uint32_t lo16 = r16[seg_ring_idx(d.oi, 2 * i2)], hi16 = r16[seg_ring_idx(d.oi, 2 * i2 + 1)];
      mp[i2] = lo16 | (hi16 << 16);
    }
    if (d.unfl) wd_mark_flush(S, d, d.unfl);  // the tail of the symbol store
  } else {
    // short ring: the nearest R symbols from the ring, the ones before from the store (everything but the ring's share
    // has been flushed) or, in front of the segment, the markers themselves
    if (d.o > (uint64_t)R && (d.sym_ovf || d.sym == nullptr)) {  // (cannot read back what was never stored)
      if (lane == 0) {
        atomicAdd(far_nostore, 1u);
        sres[w] = r;
      }
      return;
    }
    for (uint32_t i2 = lane; i2 < ZES_WINDOW / 2; i2 += 64) {
      uint32_t v[2];
#pragma unroll
      for (uint32_t h = 0; h < 2; h++) {
        const uint32_t back = ZES_WINDOW - (2 * i2 + h);  // the symbol `back` positions before the end, 1..32768
        if (back <= R) {
          v[h] = r16[d.oi >= back ? d.oi - back : d.oi + R - back];
        } else {
          const int64_t pos = (int64_t)d.o - (int64_t)back;
          if (pos < 0) {
            v[h] = 256u + (uint32_t)((int64_t)ZES_WINDOW + pos);
          } else {
            const uint32_t wd = __hip_atomic_load(&d.sym[(uint64_t)pos >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[h] = (pos & 1) ? wd >> 16 : wd & 0xffffu;
          }
        }
      }
      mp[i2] = v[0] | (v[1] << 16);
    }
    if (d.unfl) wd_mark_flush(S, d, d.unfl);  // the tail of the symbol store
  }
  r.end_bit = wd_pos(d);
  r.out_len = d.o;
  r.flags = 1u | (bfinal ? 2u : 0u) | (d.sym_ovf ? 4u : 0u);
  if (lane == 0) sres[w] = r;
}

__global__ __launch_bounds__(64) void k_inf_seg_scan(const uint8_t* __restrict__ d_in, const ZesSegJob* __restrict__ jobs, uint32_t njobs,
                                                     const uint32_t* __restrict__ cand_all, ZesSegRes* __restrict__ sres_all,
                                                     uint32_t* __restrict__ maps_all, uint32_t* __restrict__ sym16_all, uint32_t sym_ratio,
                                                     const uint32_t* __restrict__ order, uint32_t* __restrict__ far_nostore,
                                                     const uint32_t* __restrict__ nlive, uint64_t* __restrict__ symoff_all) {
  __shared__ __align__(16) InfSmem S;
  seg_scan_body(S, d_in, jobs, njobs, cand_all, sres_all, maps_all, sym16_all, sym_ratio, order, far_nostore, nlive, symoff_all);
}
// the same with the short marker ring: three decoders per CU
__global__ __launch_bounds__(64) void k_inf_seg_scan_short(const uint8_t* __restrict__ d_in, const ZesSegJob* __restrict__ jobs, uint32_t njobs,
                                                           const uint32_t* __restrict__ cand_all, ZesSegRes* __restrict__ sres_all,
                                                           uint32_t* __restrict__ maps_all, uint32_t* __restrict__ sym16_all, uint32_t sym_ratio,
                                                           const uint32_t* __restrict__ order, uint32_t* __restrict__ far_nostore,
                                                           const uint32_t* __restrict__ nlive, uint64_t* __restrict__ symoff_all) {
  __shared__ __align__(16) InfSmemShort S;
  seg_scan_body(S, d_in, jobs, njobs, cand_all, sres_all, maps_all, sym16_all, sym_ratio, order, far_nostore, nlive, symoff_all);
}

// One workgroup per stream: the chain of segments from work item 0 to the final block.
// res->status 0: seg[0..aux) / prefix[] hold the chain, out_len the total; 1: not a clean chain.
#define SEGCHAIN_LDS 8192u  // (T2 has at most SEG_BUCKETS + 1 work items)
__global__ __launch_bounds__(256) void k_inf_seg_chain(const ZesSegJob* __restrict__ jobs, const ZesSegRes* __restrict__ sres_all,
                                                       uint32_t* __restrict__ seg_all, uint64_t* __restrict__ prefix_all,
                                                       ZesRes* __restrict__ res_all, uint32_t* __restrict__ novf_all) {
  // one workgroup per buffer of the group.  A chain that arrives at a work item nobody has decoded (declined by the block
  // decoder: a stored or fixed block, a block behind an unlisted start) reports it: status 1, out_len = item + 1 — the
  // host sends exactly those items to the wave decoder and follows the chains again (out_len 0: the chain is broken).
  const ZesSegJob jb = jobs[blockIdx.x];
  const ZesSegRes* sres = sres_all + jb.work_first;
  const uint32_t nwork = jb.ncand + 1;
  uint32_t* seg = seg_all + jb.work_first;
  uint64_t* prefix = prefix_all + jb.work_first;
  ZesRes* res = res_all + blockIdx.x;
  uint32_t* novf = novf_all + blockIdx.x;
  // The chain as a list ranking instead of one lane's walk (2000 items: 300k cycles, one LDS latency per item).  Every
  // item points at its successor, or at itself where a chain would end on it (1: not decoded, 2: it holds the final
  // block, 3: no successor); the items of the chain from item 0 are marked by pointer doubling (after round r every
  // item less than 2^(r+1) steps from item 0 is marked; what is marked is always on the chain), and since a segment
  // ends behind its start the chain's order is the items' order: the k-th marked item is seg[k].  The result is
  // checked link by link; anything else (more items than the LDS holds, a link that points backwards) and one lane
  // walks the chain as before.
  __shared__ uint16_t s_j[2][SEGCHAIN_LDS];
  __shared__ uint8_t s_m[SEGCHAIN_LDS];  // bit 0: on the chain; bits 1-2: how a chain ends here
  __shared__ unsigned long long s_part[256];
  __shared__ uint32_t s_nseg, s_stuck, s_last, s_bad;
  const uint32_t tid = threadIdx.x;
  const bool lds = nwork <= SEGCHAIN_LDS;
  if (tid == 0) {
    s_nseg = 0;
    s_stuck = 0;
    s_last = 0;
    s_bad = lds ? 0u : 1u;
    res->status = 1;
    res->out_len = 0;
    res->aux = 0;
  }
  __syncthreads();
  if (lds) {
    for (uint32_t i = tid; i < nwork; i += 256) {
      const uint32_t nx = sres[i].next & 0x3FFFFFFFu, fl = sres[i].flags;
      const uint32_t type = !(fl & 1u) ? 1u : (fl & 2u) ? 2u : (nx == 0u || nx >= nwork) ? 3u : 0u;
      s_j[0][i] = (uint16_t)(type ? i : nx);
      s_m[i] = (uint8_t)((type << 1) | (i == 0u ? 1u : 0u));
    }
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t span = 1; span < nwork; span <<= 1) {
      for (uint32_t i = tid; i < nwork; i += 256) {
        const uint32_t j = s_j[cur][i];
        s_j[cur ^ 1u][i] = s_j[cur][j];
        if ((s_m[i] & 1u) && !(s_m[j] & 1u)) s_m[j] |= 1u;  // (every writer of a byte writes the same value)
      }
      cur ^= 1u;
      __syncthreads();
    }
    // the marked items in order -> ord[] (the pointer array that is free now), the last of them -> s_last
    uint16_t* ord = s_j[cur ^ 1u];
    const uint32_t per = (nwork + 255u) / 256u;
    const uint32_t i0 = min(nwork, tid * per), i1 = min(nwork, i0 + per);
    uint32_t mine = 0, lastm = 0;
    for (uint32_t i = i0; i < i1; i++)
      if (s_m[i] & 1u) {
        mine++;
        lastm = i;
      }
    s_part[tid] = mine;
    if (mine) atomicMax(&s_last, lastm);
    __syncthreads();
    if (tid == 0) {
      unsigned long long run = 0;
      for (uint32_t t = 0; t < 256; t++) {
        const unsigned long long v = s_part[t];
        s_part[t] = run;
        run += v;
      }
      s_nseg = (uint32_t)run;  // (for now: every marked item, the one the chain ends on included)
    }
    __syncthreads();
    {
      uint32_t k = (uint32_t)s_part[tid];
      for (uint32_t i = i0; i < i1; i++)
        if (s_m[i] & 1u) ord[k++] = (uint16_t)i;
    }
    __syncthreads();
    const uint32_t nm = s_nseg, last = s_last, ltype = (uint32_t)s_m[last] >> 1;
    // every link: the successor of the k-th is the (k+1)-th, and only the last one ends a chain
    for (uint32_t k = tid; k + 1u < nm; k += 256) {
      const uint32_t i = ord[k];
      if ((s_m[i] >> 1) != 0u || (sres[i].next & 0x3FFFFFFFu) != (uint32_t)ord[k + 1u]) s_bad = 1u;
    }
    if (tid == 0 && (ltype == 0u || nm == 0u || ord[nm - 1u] != last)) s_bad = 1u;
    __syncthreads();
    if (!s_bad) {
      // 2: the chain is whole.  1: it stands in front of an item nobody has decoded.  3: it is broken.
      const uint32_t napp = ltype == 2u ? nm : (ltype == 1u ? nm - 1u : 0u);
      const bool partial = ltype == 1u && (jb.flags & ZES_SEG_PARTIAL) && napp > 0u;  // a piece of a longer stream: the chain as far as it got
      const uint32_t nkeep = (ltype == 2u || partial) ? napp : 0u;
      for (uint32_t k = tid; k < nkeep; k += 256) seg[k] = ord[k];
      if (tid == 0) {
        if (ltype == 1u) {
          res->out_len = (unsigned long long)last + 1ull;  // stuck on an undecoded (or failed) item
          s_stuck = last + 1u;
        }
        s_nseg = nkeep;
      }
    }
    // seg[] is read back below by other lanes than its writers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (s_bad) {  // (uniform)
    if (tid == 0) {
      uint32_t w = 0, k = 0;
      for (;;) {
        const uint32_t nf = (sres[w].next & 0x3FFFFFFFu) | (sres[w].flags << 30);
        if (!(nf >> 30 & 1u) || k >= nwork) {
          if (k < nwork) {
            res->out_len = (unsigned long long)w + 1ull;  // stuck on an undecoded (or failed) item
            s_stuck = w + 1u;
            if ((jb.flags & ZES_SEG_PARTIAL) && k > 0) break;  // a piece of a longer stream: the chain as far as it got
          }
          k = 0;
          break;
        }
        seg[k++] = w;
        if (nf >> 31) break;  // final block inside this segment
        w = nf & 0x3FFFFFFFu;
        if (w == 0 || w >= nwork) {
          k = 0;
          break;
        }
      }
      s_nseg = k;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const uint32_t nseg = s_nseg;
  if (nseg == 0) return;
  // output offsets: chunked sums, scan of the chunk totals, chunked prefixes
  const uint32_t per = (nseg + 255u) / 256u;
  const uint32_t k0 = min(nseg, tid * per), k1 = min(nseg, k0 + per);
  unsigned long long sum = 0;
  uint32_t ovf = 0;
  for (uint32_t k = k0; k < k1; k++) {
    sum += sres[seg[k]].out_len;
    ovf += (sres[seg[k]].flags >> 2) & 1u;
  }
  if (ovf) atomicAdd(novf, ovf);  // segments whose symbols did not fit the store: decoded a second time
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    unsigned long long run = 0;
    for (uint32_t t = 0; t < 256; t++) {
      const unsigned long long v = s_part[t];
      s_part[t] = run;
      run += v;
    }
    res->out_len = run;
    res->aux = nseg;
    res->status = s_stuck ? 3 : 0;  // 3: a partial chain (ZES_SEG_PARTIAL), the item it stands in front of in the second record
    ZesRes* r2 = res_all + gridDim.x + blockIdx.x;  // second record: where the chain ends
    r2->out_len = sres[seg[nseg - 1]].end_bit;
    r2->aux = s_stuck;
    r2->status = (int)((sres[seg[nseg - 1]].flags >> 1) & 1u);  // the stream's final block has passed
  }
  __syncthreads();
  unsigned long long run = s_part[tid];
  for (uint32_t k = k0; k < k1; k++) {
    prefix[k] = run;
    run += sres[seg[k]].out_len;
  }
}

// The 32 KiB window behind each chain segment: wins[k] = the bytes in front of segment k+1 = map k applied to
// window k-1.  Applying maps is associative, so the chain is cut into groups of SEGWIN_GROUP segments:
//   k_inf_seg_win_group  one workgroup per group composes its maps from the identity: per segment the window
//                        as 16-bit symbols over the window in front of the *group* (pw16)
//   k_inf_seg_win_top    one workgroup per buffer walks the groups: the byte window in front of each group (gw)
//   k_inf_seg_win_fin    one workgroup per segment: pw16 through gw -> wins
// (the serial part is groups + group size steps instead of one per segment: 666 segments 1.4 ms -> 0.2 ms)
__global__ __launch_bounds__(1024) void k_inf_seg_win_group(const uint32_t* __restrict__ maps_all, const uint32_t* __restrict__ seg_all,
                                                            const ZesSegJob* __restrict__ jobs, uint32_t* __restrict__ pw16_all) {
  const ZesSegJob jb = jobs[blockIdx.y];
  const uint32_t nseg = jb.nseg, tid = threadIdx.x;
  if (nseg < 2) return;
  const uint32_t k0 = blockIdx.x * SEGWIN_GROUP, k1 = min(k0 + SEGWIN_GROUP, nseg - 1);  // windows k0 .. k1-1
  if (k0 >= k1) return;
  const uint32_t* maps = maps_all + (size_t)jb.work_first * (ZES_WINDOW / 2);
  const uint32_t* seg = seg_all + jb.work_first;
  uint32_t* pw16 = pw16_all + (size_t)jb.work_first * (ZES_WINDOW / 2);
  __shared__ __align__(16) uint16_t W[2][ZES_WINDOW];  // 128 KiB: the composed window, 16-bit symbols
  for (uint32_t i = tid; i < ZES_WINDOW; i += 1024) W[0][i] = (uint16_t)(256u + i);  // identity: byte i of the group's own front window
  for (uint32_t k = k0; k < k1; k++) {
    __syncthreads();
    const uint16_t* Wo = W[(k - k0) & 1];
    uint16_t* Wn = W[((k - k0) & 1) ^ 1];
    const uint4* m = reinterpret_cast<const uint4*>(maps + (size_t)seg[k] * (ZES_WINDOW / 2));
    uint4 cur[4];
#pragma unroll
    for (int j = 0; j < 4; j++) cur[j] = m[j * 1024 + tid];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t wsrc[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
      uint32_t o4[4];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t v = (wsrc[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
        const uint32_t b = v < 256u ? v : (uint32_t)Wo[(v - 256u) & (ZES_WINDOW - 1)];
        o4[q >> 1] = (q & 1) ? (o4[q >> 1] | (b << 16)) : b;
      }
      const uint32_t e = (j * 1024 + tid) * 8;  // entry index
      *reinterpret_cast<uint4*>(&Wn[e]) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
      *reinterpret_cast<uint4*>(&pw16[(size_t)k * (ZES_WINDOW / 2) + e / 2]) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
  }
}

__global__ __launch_bounds__(1024) void k_inf_seg_win_top(const uint32_t* __restrict__ pw16_all, const ZesSegJob* __restrict__ jobs,
                                                          uint8_t* __restrict__ gw_all, const uint8_t* __restrict__ d_out,
                                                          const ZesSegOut* __restrict__ outs) {
  const ZesSegJob jb = jobs[blockIdx.x];
  const uint32_t nseg = jb.nseg, tid = threadIdx.x;
  if (nseg < 2) return;
  const uint32_t ngroups = (nseg - 1 + SEGWIN_GROUP - 1) / SEGWIN_GROUP;
  const uint32_t* pw16 = pw16_all + (size_t)jb.work_first * (ZES_WINDOW / 2);
  uint8_t* gw = gw_all + (size_t)jb.work_first / SEGWIN_GROUP * ZES_WINDOW + (size_t)blockIdx.x * ZES_WINDOW;  // see the host: room per buffer
  __shared__ __align__(16) uint8_t W[2][ZES_WINDOW];
  {
    // in front of the stream: nothing — or, for a later piece of a long stream, the output so far (its last 32 KiB)
    const ZesSegOut ob = outs[blockIdx.x];
    const uint8_t* dst = d_out + ob.out_off;
    for (uint32_t i = tid; i < ZES_WINDOW; i += 1024) W[0][i] = (ZES_WINDOW - i <= ob.hist) ? dst[(int64_t)i - (int64_t)ZES_WINDOW] : (uint8_t)0;
  }
  for (uint32_t gi = 0; gi < ngroups; gi++) {
    __syncthreads();
    const uint8_t* Wo = W[gi & 1];
    uint8_t* Wn = W[(gi & 1) ^ 1];
    // the window in front of group gi goes out; the group's last composed window takes it to the next group
    for (uint32_t i = tid; i < ZES_WINDOW / 16; i += 1024)
      reinterpret_cast<uint4*>(gw + (size_t)gi * ZES_WINDOW)[i] = reinterpret_cast<const uint4*>(Wo)[i];
    if (gi + 1 == ngroups) break;
    const uint32_t klast = (gi + 1) * SEGWIN_GROUP - 1;
    const uint4* m = reinterpret_cast<const uint4*>(pw16 + (size_t)klast * (ZES_WINDOW / 2));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint4 c4 = m[j * 1024 + tid];
      const uint32_t wsrc[4] = {c4.x, c4.y, c4.z, c4.w};
      uint32_t o2[2] = {0, 0};
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t v = (wsrc[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
        const uint32_t b = v < 256u ? v : (uint32_t)Wo[(v - 256u) & (ZES_WINDOW - 1)];
        o2[q >> 2] |= b << ((q & 3) * 8);
      }
      *reinterpret_cast<uint2*>(&Wn[(j * 1024 + tid) * 8]) = make_uint2(o2[0], o2[1]);
    }
  }
}

__global__ __launch_bounds__(1024) void k_inf_seg_win_fin(const uint32_t* __restrict__ pw16_all, const ZesSegJob* __restrict__ jobs,
                                                         const uint8_t* __restrict__ gw_all, uint8_t* __restrict__ wins_all) {
  const ZesSegJob jb = jobs[blockIdx.y];
  const uint32_t nseg = jb.nseg, tid = threadIdx.x, k = blockIdx.x;
  if (nseg < 2 || k + 1 >= nseg) return;
  const uint32_t* pw16 = pw16_all + (size_t)jb.work_first * (ZES_WINDOW / 2);
  const uint8_t* gw = gw_all + (size_t)jb.work_first / SEGWIN_GROUP * ZES_WINDOW + (size_t)blockIdx.y * ZES_WINDOW +
                      (size_t)(k / SEGWIN_GROUP) * ZES_WINDOW;
  uint8_t* wins = wins_all + (size_t)jb.work_first * ZES_WINDOW;
  __shared__ __align__(16) uint8_t W[ZES_WINDOW];
  for (uint32_t i = tid; i < ZES_WINDOW / 16; i += 1024) reinterpret_cast<uint4*>(W)[i] = reinterpret_cast<const uint4*>(gw)[i];
  __syncthreads();
  const uint4* m = reinterpret_cast<const uint4*>(pw16 + (size_t)k * (ZES_WINDOW / 2));
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint4 c4 = m[j * 1024 + tid];
    const uint32_t wsrc[4] = {c4.x, c4.y, c4.z, c4.w};
    uint32_t o2[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t v = (wsrc[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
      const uint32_t b = v < 256u ? v : (uint32_t)W[(v - 256u) & (ZES_WINDOW - 1)];
      o2[q >> 2] |= b << ((q & 3) * 8);
    }
    *reinterpret_cast<uint2*>(&wins[(size_t)k * ZES_WINDOW + (j * 1024 + tid) * 8]) = make_uint2(o2[0], o2[1]);
  }
}

// Chain segment k from the symbol store: bytes as they are, markers through the window in front of the segment.
// A marker that points in front of the first byte of the stream (src/inflate.ts:287-290 would read outside the
// buffer) raises *fail: the serial tiers then reproduce what the reference does.
__global__ __launch_bounds__(256) void k_inf_seg_translate(uint8_t* __restrict__ d_out, const ZesSegJob* __restrict__ jobs,
                                                           const ZesSegOut* __restrict__ outs, const uint32_t* __restrict__ cand_all,
                                                           const ZesSegRes* __restrict__ sres_all, const uint32_t* __restrict__ seg_all,
                                                           const uint64_t* __restrict__ prefix_all, const uint8_t* __restrict__ wins_all,
                                                           const uint32_t* __restrict__ sym16_all, const uint64_t* __restrict__ symoff_all,
                                                           uint32_t* __restrict__ fail_all) {
  // grid: (segments of the longest chain, workgroups sharing a segment, buffers of the group)
  const ZesSegJob jb = jobs[blockIdx.z];
  const ZesSegOut ob = outs[blockIdx.z];
  if (blockIdx.x >= ob.nseg) return;  // (0: this buffer is not translated — not decoded by this tier, or decoded in place)
  const uint64_t out_off = ob.out_off, cap = ob.cap;
  const ZesSegRes* sres = sres_all + jb.work_first;
  const uint32_t* seg = seg_all + jb.work_first;
  const uint64_t* prefix = prefix_all + jb.work_first;
  const uint8_t* wins = wins_all + (size_t)jb.work_first * ZES_WINDOW;
  const uint64_t* symoff = symoff_all + jb.work_first;
  uint32_t* fail = fail_all + blockIdx.z;
  __shared__ __align__(16) uint8_t W[ZES_WINDOW];
  const uint32_t k = blockIdx.x, tid = threadIdx.x;
  const uint32_t w = seg[k];
  const ZesSegRes r = sres[w];
  if (r.flags & 4u) return;  // not in the store: k_inf_seg_decode
  const uint64_t pre = prefix[k];
  if (k > 0) {
    const uint4* wv = reinterpret_cast<const uint4*>(wins + (size_t)(k - 1) * ZES_WINDOW);
    for (uint32_t i = tid; i < ZES_WINDOW / 16; i += 256) reinterpret_cast<uint4*>(W)[i] = wv[i];
  } else if (ob.hist) {  // a later piece of a long stream: the window in front of its first segment is output that exists
    const uint8_t* hp = d_out + out_off;
    for (uint32_t i = tid; i < ZES_WINDOW; i += 256) W[i] = (ZES_WINDOW - i <= ob.hist) ? hp[(int64_t)i - (int64_t)ZES_WINDOW] : (uint8_t)0;
  }
  __syncthreads();
  const uint64_t front = pre + ob.hist;  // bytes that exist in front of the segment
  const uint32_t first_ok = front >= ZES_WINDOW ? 0u : ZES_WINDOW - (uint32_t)front;  // window positions below this do not exist
  const uint16_t* sy = reinterpret_cast<const uint16_t*>(sym16_all + symoff[w]);  // (the buffer's part of symoff[])
  uint8_t* dst = d_out + out_off;
  const uint64_t end = min(pre + r.out_len, cap);
  uint32_t bad = 0;
  // 8 output bytes per thread and step, groups aligned in the output
  // (blockIdx.y: several workgroups share a long segment)
  for (uint64_t g = (pre & ~7ull) + ((uint64_t)blockIdx.y * 256 + tid) * 8; g < end; g += (uint64_t)gridDim.y * 256 * 8) {
    uint32_t o2[2] = {0, 0};
    const bool whole = g >= pre && g + 8 <= end;
    // a whole group's eight symbols: two aligned 16-byte reads and a byte shift that is the same for every group of the
    // workgroup (g is a multiple of 8, so (g - pre) * 2 mod 16 depends on pre alone) — eight 2-byte reads, each its own
    // instruction over 64 lanes 16 bytes apart, were most of this kernel's memory instructions
    uint32_t sd[4] = {0, 0, 0, 0};
    if (whole) {
      const uintptr_t sb = reinterpret_cast<uintptr_t>(sy) + 2u * (uintptr_t)(g - pre);
      const uint4* al = reinterpret_cast<const uint4*>(sb & ~(uintptr_t)15);
      const uint4 lo = al[0], hi = al[1];  // (behind the last symbol: the store's slack, or the next item's share)
      const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const uint32_t sh = (uint32_t)(sb & 15u), wa = sh >> 2, wb = sh & 3u;  // (uniform)
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        uint32_t x0 = 0, x1 = 0;
#pragma unroll
        for (uint32_t a = 0; a < 4; a++) {
          x0 = wa == a ? w[a + j] : x0;
          x1 = wa == a ? w[a + j + 1] : x1;
        }
        sd[j] = __builtin_amdgcn_alignbyte(x1, x0, wb);
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
      const uint64_t a = g + q;
      uint32_t b = 0;
      if (a >= pre && a < end) {
        const uint32_t v = whole ? ((sd[q >> 1] >> (16u * (q & 1u))) & 0xffffu) : (uint32_t)sy[a - pre];
        if (v >= 256u) {
          bad |= (v - 256u) < first_ok;
          b = W[(v - 256u) & (ZES_WINDOW - 1)];
        } else {
          b = v;
        }
        if (!whole) dst[a] = (uint8_t)b;
      }
      o2[q >> 2] |= b << ((q & 3u) * 8u);
    }
    if (whole) *reinterpret_cast<uint2*>(dst + g) = make_uint2(o2[0], o2[1]);
  }
  if (bad) atomicOr(fail, 1u);
}

// Chain segment k again, this time with its window and into its place.
__global__ __launch_bounds__(64) void k_inf_seg_decode(const uint8_t* __restrict__ d_in, uint64_t in_off, uint64_t c,
                                                       uint8_t* __restrict__ d_out, uint64_t out_off, uint64_t cap,
                                                       const uint32_t* __restrict__ cand, const ZesSegRes* __restrict__ sres,
                                                       const uint32_t* __restrict__ seg, const uint64_t* __restrict__ prefix,
                                                       const uint8_t* __restrict__ wins, uint32_t* __restrict__ fail,
                                                       uint32_t only_overflowed, uint32_t start0, uint32_t hist) {
  __shared__ __align__(16) InfSmem S;
  const uint32_t k = blockIdx.x, lane = threadIdx.x;
  const uint32_t w = seg[k];
  const ZesSegRes r = sres[w];
  if (only_overflowed && !(r.flags & 4u)) return;  // k_inf_seg_translate has this one
  const uint64_t pre = prefix[k];
  const uint64_t base = pre & ~15ull;
  WaveDec d;
  d.in32 = reinterpret_cast<const uint32_t*>(d_in + in_off);
  d.nbytes = c;
  d.out = d_out + out_off + base;
  d.cap = cap > base ? cap - base : 0;
  d.ostart = pre & 15u;
  d.o = d.ostart;
  d.flushed = 0;
  d.reach = (uint32_t)min(pre + hist, (uint64_t)ZES_WINDOW);
  d.unfl = (uint32_t)d.ostart;
  d.oi = 0;
  if (k == 0 && hist) {  // a later piece of a long stream: the bytes in front of its first segment are output that exists
    const uint8_t* hp = d_out + out_off;
    for (uint32_t i = lane; i < ZES_WINDOW; i += 64) {
      const uint32_t p = (uint32_t)d.ostart + ZES_WINDOW + i;  // == position mod 65536
      S.ring[p & (RING - 1)] = (ZES_WINDOW - i <= hist) ? hp[(int64_t)i - (int64_t)ZES_WINDOW] : (uint8_t)0;
    }
  }
  if (k > 0) {  // window bytes in front of the segment: positions ostart - 32768 .. ostart - 1
    const uint32_t* wv = reinterpret_cast<const uint32_t*>(wins + (size_t)(k - 1) * ZES_WINDOW);
    for (uint32_t i = lane; i < ZES_WINDOW / 4; i += 64) {
      const uint32_t v = wv[i];
      const uint32_t p = (uint32_t)d.ostart + ZES_WINDOW + 4u * i;  // == position mod 65536
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) S.ring[(p + q) & (RING - 1)] = (uint8_t)(v >> (8 * q));
    }
  }
  wd_seek(d, w ? (uint64_t)cand[w - 1] + 16 : (uint64_t)start0);
  uint32_t bfinal = 0;
  int rc;
  for (;;) {
    rc = wd_block<false>(S, d, &bfinal);
    if (rc != WD_OK || bfinal || wd_pos(d) >= r.end_bit) break;
  }
  if (rc == WD_OK && wd_pos(d) == r.end_bit && d.o - d.ostart == r.out_len) {
    wd_flush_range(S, d, d.flushed, d.o);
  } else if (lane == 0) {
    atomicOr(fail, 1u);
  }
}

// ------------------------------------------------------------------------------------------
// k_inf_chain: validates the parallel decode.  Success needs: candidate 0 at bit 16, every
// block ok, every non-final block exactly 131072 bytes, end bit of block k == start of k+1,
// and the k-th chain member being the k-th candidate (otherwise a remap pass is requested).
// res->status: 0 done, 2 remap needed (chain in map_out, length in res->aux), 1 give up (T2).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inf_chain(const ZesInfBuf* __restrict__ bufs, const uint32_t* __restrict__ cnt,
                                                   const uint32_t* __restrict__ cand_all, const ZesCandRes* __restrict__ cres_all,
                                                   const uint32_t* __restrict__ map_in_all, uint32_t* __restrict__ map_out_all,
                                                   ZesRes* __restrict__ res_all, const uint32_t* __restrict__ counters, uint32_t counter_words,
                                                   uint32_t* __restrict__ counters_host) {
  __shared__ uint32_t s_bad, s_final;
  __shared__ unsigned long long s_total;
  const uint32_t tid = threadIdx.x;
  // (one buffer: res_all and counters_host point into the host's page-locked read-back area; the counters — final since
  // the verify kernels — go there with this kernel, the host reads both after one synchronisation, no copy command)
  if (counters_host && blockIdx.x == 0)
    for (uint32_t i = tid; i < counter_words; i += blockDim.x) counters_host[i] = counters[i];
  const ZesInfBuf bf = bufs[blockIdx.x];
  const uint32_t ncand = min(cnt[blockIdx.x], bf.cand_cap);
  uint32_t nwork = bufs[blockIdx.x + 1].work_first - bf.work_first;
  const bool autow = bufs[gridDim.x].work_first == ZES_WORK_AUTO;  // one buffer: see k_inf_block_par
  if (autow) nwork = min(ncand, bufs[gridDim.x].cand_cap);
  const uint32_t* cand = cand_all + bf.cand_base;
  const ZesCandRes* cres = cres_all + bf.cand_base;
  const uint32_t* map_in = map_in_all ? map_in_all + bf.cand_base : nullptr;
  uint32_t* map_out = map_out_all + bf.cand_base;
  ZesRes* res = res_all + blockIdx.x;
  if (tid == 0) {
    s_bad = 0;
    s_final = 0xFFFFFFFFu;
    s_total = 0;
    res->status = 1;
    res->out_len = 0;
    res->aux = 0;
  }
  __syncthreads();
  if (ncand == 0 || nwork == 0 || cand[0] != bf.start_rel || cnt[blockIdx.x] > bf.cand_cap) return;  // (no work items: the buffer was left out, results would be stale)
  if (autow && ncand > nwork) return;  // more candidates than work items were launched: the host falls back
  // Fast check, all work items in parallel: item k is ok, non-final items give exactly one slot
  // and end where item k+1 starts, the first final item closes the chain.
  uint32_t first_final = 0xFFFFFFFFu;
  for (uint32_t k = tid; k < nwork; k += 256)
    if ((cres[k].flags & 3u) == 3u) {
      first_final = k;
      break;
    }
  atomicMin(&s_final, first_final);
  __syncthreads();
  const uint32_t K = s_final;  // index of the closing block
  if (K != 0xFFFFFFFFu) {
    unsigned long long part = 0;
    uint32_t bad = 0;
    for (uint32_t k = tid; k <= K; k += 256) {
      const ZesCandRes r = cres[k];
      if (!(r.flags & 1u)) bad = 1;
      part += r.out_len;
      if (k < K) {
        const uint32_t nxt = map_in ? map_in[k + 1] : k + 1;
        if (r.out_len != ZES_BLK || nxt >= ncand || (uint64_t)cand[nxt] + 16 != r.end_bit) bad = 1;
      }
      if (map_in == nullptr) map_out[k] = k;
    }
    if (bad) atomicOr(&s_bad, 1u);
    atomicAdd(&s_total, part);
    __syncthreads();
    if (!s_bad) {
      if (tid == 0) {
        res->status = 0;
        res->out_len = s_total;
        res->aux = 1;
      }
      return;
    }
  }
  if (map_in || tid != 0) return;  // a remapped pass that still fails goes to T2
  // Slow path (false candidates between the blocks): follow end bit -> next start serially.
  uint32_t j = 0, k = 0;
  uint64_t total = 0;
  for (;;) {
    const ZesCandRes r = cres[j];
    if (!(r.flags & 1u)) return;
    map_out[k] = j;
    k++;
    total += r.out_len;
    if (r.flags & 2u) break;
    if (r.out_len != ZES_BLK) return;
    const uint64_t want = r.end_bit - 16;
    uint32_t lo = j + 1, hi = ncand;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((uint64_t)cand[mid] < want) lo = mid + 1; else hi = mid;
    }
    if (lo >= ncand || (uint64_t)cand[lo] != want) return;
    j = lo;
  }
  res->out_len = total;
  res->aux = k;
  res->status = 2;  // the slots are shifted: decode the chain again in order
}

// ------------------------------------------------------------------------------------------
// k_inf_chain_range: the chain check of one PIECE of a reference-made stream (zes_inflate_range_dev): the blocks that
// start inside [start_rel, own_rel) — every one decoded fine, exactly 131072 bytes unless it is the stream's final
// block, each ending where the next candidate starts (the candidate behind the last own block included: it is the
// next piece's first block).  No repair of any kind: anything irregular (a false candidate among them) fails the piece
// and the caller decodes the stream in one go.
// res[0]: status 0, out_len = bytes of the own blocks, aux = their number | final << 31;  res[1]: out_len = bit
// position (relative to the piece) behind the last own block, aux|status = bit position of the first own block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inf_chain_range(const ZesInfBuf* __restrict__ bufs, const uint32_t* __restrict__ cnt,
                                                         const uint32_t* __restrict__ cand_all, const ZesCandRes* __restrict__ cres_all,
                                                         ZesRes* __restrict__ res, unsigned long long* __restrict__ acc) {
  __shared__ uint32_t s_bad, s_final;
  __shared__ unsigned long long s_total;
  const uint32_t tid = threadIdx.x;
  const ZesInfBuf bf = bufs[0];
  const uint32_t ncand = min(cnt[0], bf.cand_cap);
  const uint32_t* cand = cand_all + bf.cand_base;
  const ZesCandRes* cres = cres_all + bf.cand_base;
  if (tid == 0) {
    s_bad = 0;
    s_final = 0;
    s_total = 0;
    res[0].status = 1;
    res[0].out_len = 0;
    res[0].aux = 0;
    res[1].status = 0;
    res[1].out_len = 0;
    res[1].aux = 0;
  }
  __syncthreads();
  if (ncand == 0 || cnt[0] > bf.cand_cap) return;
  if (!(bf.range_flags & ZES_START_ANY) && cand[0] != bf.start_rel) return;
  // own blocks: candidates below own_rel (the list is sorted)
  uint32_t lo = 0, hi = ncand;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cand[mid] < bf.own_rel) lo = mid + 1; else hi = mid;
  }
  const uint32_t nown = lo;
  if (nown == 0) {  // no block starts in this piece (possible for a piece in the middle of one block)
    if (tid == 0) res[0].status = 0;
    return;
  }
  unsigned long long part = 0;
  uint32_t bad = 0, fin = 0;
  for (uint32_t k = tid; k < nown; k += 256) {
    const ZesCandRes r = cres[k];
    if (!(r.flags & 1u)) bad = 1;
    part += r.out_len;
    if (r.flags & 2u) {  // the stream's final block: must be the last own one
      if (k + 1 != nown) bad = 1;
      fin = 1;
    } else if (r.out_len != ZES_BLK || k + 1 >= ncand || (uint64_t)cand[k + 1] + 16 != r.end_bit) {
      bad = 1;  // (k + 1 == nown: the candidate behind the last own block is the next piece's first block)
    }
  }
  if (bad) atomicOr(&s_bad, 1u);
  if (fin) atomicOr(&s_final, 1u);
  atomicAdd(&s_total, part);
  __syncthreads();
  if (s_bad || tid != 0) return;
  res[0].status = 0;
  res[0].out_len = s_total;
  res[0].aux = nown | (s_final << 31);
  res[1].out_len = cres[nown - 1].end_bit;
  res[1].aux = cand[0] + 16u;
  if (acc) acc[0] += nown;  // (the next piece's table is set from it: k_inf_set_table_range)
}

// ------------------------------------------------------------------------------------------
// k_inf_exact (T3): single lane, state-for-state with the reference's BitReadStream and block
// decoders.  Starts at a block boundary (bit position, output offset) handed over by T2, or at
// bit 16 / offset 0.  Output goes straight to global memory (the same lane reads it back).
// ------------------------------------------------------------------------------------------
struct XReader {  // src/utils/BitReadStream.ts:1-50
  const uint8_t* buf;
  uint64_t len;
  int64_t idx;
  uint32_t now_bits;
  int now_len;
  int is_end;
  int err;
};
__device__ static inline uint32_t xr_byte(const XReader& r, int64_t i) { return ((uint64_t)i < r.len) ? r.buf[i] : 0u; }
__device__ static int xr_read(XReader& r) {  // :14-32
  if (r.is_end) {
    r.err = ZES_E_LACK;
    return 0;
  }
  const int bit = (int)(r.now_bits & 1u);
  if (r.now_len > 1) {
    r.now_len--;
    r.now_bits >>= 1;
  } else {
    r.idx++;
    if ((uint64_t)r.idx < r.len) {
      r.now_bits = r.buf[r.idx];
      r.now_len = 8;
    } else {
      r.now_len = 0;
      r.is_end = 1;
    }
  }
  return bit;
}
__device__ static uint32_t xr_range(XReader& r, int length) {  // :33-42 (eager refill, zeros past the end)
  while (r.now_len <= length) {
    r.idx++;
    r.now_bits |= xr_byte(r, r.idx) << r.now_len;
    r.now_len += 8;
  }
  const uint32_t bits = r.now_bits & ((1u << length) - 1u);
  r.now_bits >>= length;
  r.now_len -= length;
  return bits;
}
__device__ static uint32_t xr_coded(XReader& r, int length) {  // :43-49; length < 0 = empty table
  uint32_t bits = 0;
  if (length < 0) {
    while (!r.err) xr_read(r);
    return 0;
  }
  for (int i = 0; i < length && !r.err; i++) bits = (bits << 1) | (uint32_t)xr_read(r);
  return bits;
}

struct XTab {  // src/huffman.ts:8-39
  int lmin, lmax;
  uint32_t first[16];
  uint16_t count[16], offs[16];
  uint16_t syms[352];
};
__device__ static void xt_build(XTab& t, const uint8_t* lens, int nsym) {
  t.lmin = 99;
  t.lmax = 0;
  for (int l = 0; l < 16; l++) t.count[l] = 0;
  for (int s = 0; s < nsym; s++)
    if (lens[s]) {
      t.count[lens[s]]++;
      if (lens[s] < t.lmin) t.lmin = lens[s];
      if (lens[s] > t.lmax) t.lmax = lens[s];
    }
  if (t.lmax == 0) {
    t.lmin = -1;
    return;
  }
  uint32_t code = 0;
  uint16_t off = 0;
  uint16_t fill[16];
  for (int l = t.lmin; l <= t.lmax; l++) {
    t.first[l] = code;
    t.offs[l] = off;
    fill[l] = off;
    code += t.count[l];
    off = (uint16_t)(off + t.count[l]);
    code <<= 1;
  }
  for (int s = 0; s < nsym; s++)
    if (lens[s]) t.syms[fill[lens[s]]++] = (uint16_t)s;
}
__device__ static int xt_decode(const XTab& t, XReader& r, int* err) {  // src/inflate.ts:238-252 and siblings
  int cl = t.lmin;
  uint32_t code = xr_coded(r, t.lmin);
  if (r.err) {
    *err = r.err;
    return -1;
  }
  for (;;) {
    const uint32_t rel = code - t.first[cl];
    if (code >= t.first[cl] && rel < t.count[cl]) return t.syms[t.offs[cl] + rel];
    if (t.lmax <= cl) {
      *err = ZES_E_CORRUPT;
      return -1;
    }
    cl++;
    code = (code << 1) | (uint32_t)xr_read(r);
    if (r.err) {
      *err = r.err;
      return -1;
    }
  }
}

struct XOut {
  uint8_t* buf;
  uint64_t cap, idx;
  uint64_t limit;  // more output than any stream of this length can hold (1032:1): see runaway
  int overflow, runaway;
};
__device__ static inline void xo_write(XOut& o, uint8_t v) {
  // The reference refills with zero bits past the end of the data without noticing (readRange,
  // src/utils/BitReadStream.ts:33-42), so a truncated stream whose zero bits decode as tokens with extra bits is
  // decoded for ever, until its process runs out of memory.  This decoder stops with 'Lack of data length'.
  if (o.idx >= o.limit) {
    o.runaway = 1;
    return;
  }
  if (o.idx < o.cap) o.buf[o.idx] = v; else o.overflow = 1;
  o.idx++;
}

__device__ static int x_symbols(XReader& r, XOut& o, const XTab& lt, const XTab* dt) {  // src/inflate.ts:78-117, 237-291
  int err = 0;
  while (!r.is_end) {
    // all-zero bits for 1 KiB past the end: the token loop has settled into a cycle that never ends (see xo_write)
    if (o.runaway || r.idx > (int64_t)r.len + 1024) return ZES_E_LACK;
    const int v = xt_decode(lt, r, &err);
    if (v < 0) return err;
    if (v < 256) {
      xo_write(o, (uint8_t)v);
      continue;
    }
    if (v == 256) break;
    const int lc = v - 257;
    const bool have_len = lc < 29;  // codes 286/287: base undefined, nothing is copied
    uint32_t rl = have_len ? kLenBase[lc] : 0u;
    if (have_len && kLenXbits[lc]) rl += xr_range(r, kLenXbits[lc]);
    int dc;
    if (!dt) {
      dc = (int)xr_coded(r, 5);
      if (r.err) return r.err;
    } else {
      dc = xt_decode(*dt, r, &err);
      if (dc < 0) return err;
    }
    const bool have_dist = dc < 30;  // codes >= 30: NaN source index, zeros are written
    uint32_t rd = have_dist ? kDistBase[dc] : 0u;
    if (have_dist && kDistXbits[dc]) rd += xr_range(r, kDistXbits[dc]);
    if (!have_len) continue;
    for (uint32_t i = 0; i < rl; i++) {
      const int64_t src = have_dist ? (int64_t)o.idx - (int64_t)rd : -1;
      uint8_t b = 0;  // bytes beyond the capacity are lost, the length stays exact
      if (have_dist && src >= 0 && (uint64_t)src < o.cap) b = o.buf[src];
      xo_write(o, b);
    }
  }
  return 0;
}

__global__ void k_inf_exact(const uint8_t* __restrict__ d_in, uint64_t in_off, uint64_t c, uint8_t* __restrict__ d_out,
                            uint64_t out_off, uint64_t cap, const uint64_t* __restrict__ resume, ZesRes* __restrict__ res) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  __shared__ XTab lt, dt, ct;
  __shared__ uint8_t llens[296], dlens[64], clens[19];
  XReader r;
  r.buf = d_in + in_off;
  r.len = c;
  r.err = 0;
  r.is_end = 0;
  const uint64_t bit = resume ? resume[0] : 16;
  // state of the reference reader at a block boundary: a function of the bit position alone
  r.idx = (int64_t)(bit >> 3);
  r.now_len = 8 - (int)(bit & 7);
  r.now_bits = xr_byte(r, r.idx) >> (bit & 7);
  XOut o;
  o.buf = d_out + out_off;
  o.cap = cap;
  o.idx = resume ? resume[1] : 0;
  o.overflow = 0;
  o.runaway = 0;
  o.limit = 1032ull * (c > 2 ? c - 2 : 0) + 65536;
  int bfinal = 0, rc = 0;
  while (bfinal != 1) {  // src/inflate.ts:22-37
    bfinal = (int)xr_range(r, 1);
    const int btype = (int)xr_range(r, 2);
    if (btype == 0) {  // :42-55
      if (r.now_len < 8) xr_range(r, r.now_len);
      uint32_t LEN = xr_range(r, 8);
      LEN |= xr_range(r, 8) << 8;
      uint32_t NLEN = xr_range(r, 8);
      NLEN |= xr_range(r, 8) << 8;
      if (LEN + NLEN != 65535u) rc = ZES_E_CORRUPT;
      else
        for (uint32_t i = 0; i < LEN; i++) xo_write(o, (uint8_t)xr_range(r, 8));
    } else if (btype == 1) {  // :57-118
      for (int i = 0; i <= 287; i++) llens[i] = (uint8_t)(i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8);
      xt_build(lt, llens, 288);
      rc = x_symbols(r, o, lt, nullptr);
    } else if (btype == 2) {  // :120-292
      const int HLIT = (int)xr_range(r, 5) + 257;
      const int HDIST = (int)xr_range(r, 5) + 1;
      const int HCLEN = (int)xr_range(r, 4) + 4;
      for (int i = 0; i < 19; i++) clens[i] = 0;
      for (int i = 0; i < HCLEN; i++) clens[kClOrder[i]] = (uint8_t)xr_range(r, 3);
      xt_build(ct, clens, 19);
      for (int i = 0; i < 296; i++) llens[i] = 0;
      for (int i = 0; i < 64; i++) dlens[i] = 0;
      int repeat = 0, codelen = 0, err = 0;
      const int total = HLIT + HDIST;
      for (int i = 0; i < total && !rc;) {
        const int sy = xt_decode(ct, r, &err);
        if (sy < 0) {
          rc = err;
          break;
        }
        if (sy == 16) {
          repeat = 3 + (int)xr_range(r, 2);
        } else if (sy == 17) {
          repeat = 3 + (int)xr_range(r, 3);
          codelen = 0;
        } else if (sy == 18) {
          repeat = 11 + (int)xr_range(r, 7);
          codelen = 0;
        } else {
          repeat = 1;
          codelen = sy;
        }
        if (codelen <= 0) {
          i += repeat;
        } else {
          while (repeat) {  // may run past `total`: extra symbols land in the distance table
            if (i < HLIT) llens[i] = (uint8_t)codelen;
            else dlens[i - HLIT] = (uint8_t)codelen;
            i++;
            repeat--;
          }
        }
      }
      if (!rc) {
        xt_build(lt, llens, 288);
        xt_build(dt, dlens, 64);
        rc = x_symbols(r, o, lt, &dt);
      }
    } else {
      rc = ZES_E_BTYPE3;
    }
    if (!rc && o.runaway) rc = ZES_E_LACK;
    if (rc) break;
    if (bfinal == 0 && r.is_end) {  // :34-36
      rc = ZES_E_INSUFFICIENT;
      break;
    }
  }
  res->aux = 3;
  if (rc) {
    res->status = rc;
    res->out_len = 0;
  } else if (o.overflow) {
    res->status = -16;  // ZES_E_NOSPACE; out_len is the exact size needed
    res->out_len = o.idx;
  } else {
    res->status = 0;
    res->out_len = o.idx;
  }
}
