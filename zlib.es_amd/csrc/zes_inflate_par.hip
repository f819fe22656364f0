// zes_inflate_par.hip — T1 block decoder: one 1024-thread workgroup per 131072-byte block.
//
// DEFLATE symbol decoding is bit-serial, so a block is cut into 1024 bit segments and every
// lane starts decoding at its segment's first bit.  A lane that starts inside a token decodes
// garbage.  Self-synchronisation cannot be relied on (random data gets almost fixed 8-bit
// codes, which never re-align), so instead every lane computes its segment's *transfer table*:
// a token that starts before the segment can end at most 47 bits into it, so there are 48
// possible entry offsets; the lane decodes from each offset not already covered by an earlier
// trajectory (visited mask over the first 128 bits) and records where that trajectory leaves
// the segment.  The true chain is then table composition from the header end.
// Phases:
//
//   P0  wave 0 parses the dynamic header (reference src/inflate.ts:120-204) and builds root
//       tables; the other waves clear the match bitmap
//   P1  per-segment transfer tables (entry offset -> exit offset | EOB | fail), kept in the
//       not-yet-used LDS image; composition per wave, across waves, then per lane
//   P2  count pass from the true entries; workgroup scan of bytes produced -> output offsets
//   P3  emit decode: literals go straight into the LDS image of the block; a match leaves a
//       3-byte (distance, length) record at its own destination and a bit in the bitmap
//   P4  wave 0 walks the bitmap in order and resolves matches 64 at a time; a lane may copy as
//       soon as its source lies below the first unresolved match (src/inflate.ts:287-290)
//   P5  the 128 KiB image leaves LDS as coalesced 16-byte stores
//
// Anything unusual (non-dynamic block, over-subscribed or incomplete code that gets hit,
// out-of-table symbol, distance behind the block start, more than 131072 bytes, running off
// the data) marks the block not-ok; the chain check then sends the stream to T2/T3.
#include "zes_common.h"
#include "zes_kernels.h"

#define PL_ROOT 10u
#define PD_ROOT 9u
#define F_EOB 1u
#define F_FAIL 2u
#define F_VOID 4u
#define F_HIST 8u

struct ParSmem {
  uint8_t out[ZES_BLK];
  uint32_t bitmap[ZES_BLK / 32];
  uint32_t lut_l[1u << PL_ROOT];
  uint32_t lut_d[1u << PD_ROOT];
  uint16_t syms_l[288];
  uint16_t syms_d[32];
  uint32_t first_l[16], first_d[16];
  uint16_t cnt_l[16], cnt_d[16], offs_l[16], offs_d[16];
  uint8_t lens[352];
  uint8_t cl_lut[128];
  uint32_t wave_exit[PAR_WAVES];
  uint32_t wave_flags[PAR_WAVES];
  uint32_t wave_sum[PAR_WAVES];
  uint16_t mlist[704];
  uint8_t wtab[PAR_WAVES][48];   // composed transfer table of each wave
  uint8_t wentry[PAR_WAVES];      // entry code of each wave's first segment
  uint8_t lentry[PAR_THREADS];    // entry code of every segment
  uint32_t hdr_end, status, tail_entry, bfinal, tail_bytes, tail_end;
};

// LUT entry: [3:0] code length (0 = not in the root table), [7:4] extra bits, [9:8] kind
// (0 literal, 1 end of block, 2 length or distance base, 3 symbol outside the tables), [31:16] value
__device__ __forceinline__ static uint32_t entry_l(uint32_t sym, uint32_t len) {
  if (sym < 256u) return len | (sym << 16);
  if (sym == 256u) return len | (1u << 8);
  if (sym < 286u) return len | ((uint32_t)kLenXbits[sym - 257u] << 4) | (2u << 8) | ((uint32_t)kLenBase[sym - 257u] << 16);
  return len | (3u << 8);
}
__device__ __forceinline__ static uint32_t entry_d(uint32_t sym, uint32_t len) {
  if (sym < 30u) return len | ((uint32_t)kDistXbits[sym] << 4) | (2u << 8) | ((uint32_t)kDistBase[sym] << 16);
  return len | (3u << 8);
}

struct LaneBits {
  uint64_t bb;
  uint32_t nb;
  uint32_t pos;  // bit offset of bb's bit 0 inside the buffer; (pos + nb) % 32 == 0
};
__device__ __forceinline__ static uint32_t lb_ldw(const uint32_t* in32, uint32_t idx, uint32_t lastdw) {
  return in32[idx < lastdw ? idx : lastdw];
}
__device__ __forceinline__ static void lb_seek(LaneBits& b, const uint32_t* in32, uint32_t lastdw, uint32_t bit) {
  const uint32_t i = bit >> 5, sh = bit & 31u;
  const uint64_t w = (uint64_t)lb_ldw(in32, i, lastdw) | ((uint64_t)lb_ldw(in32, i + 1, lastdw) << 32);
  b.bb = w >> sh;
  b.nb = 64u - sh;
  b.pos = bit;
}
__device__ __forceinline__ static void lb_refill(LaneBits& b, const uint32_t* in32, uint32_t lastdw) {
  if (b.nb <= 32u) {
    b.bb |= (uint64_t)lb_ldw(in32, (b.pos + b.nb) >> 5, lastdw) << b.nb;
    b.nb += 32u;
  }
}
__device__ __forceinline__ static uint32_t lb_take(LaneBits& b, uint32_t k) {
  const uint32_t v = (uint32_t)b.bb & ((1u << k) - 1u);
  b.bb >>= k;
  b.nb -= k;
  b.pos += k;
  return v;
}

// canonical walk for codes longer than the root (same bit-at-a-time extension as the reference)
__device__ __forceinline__ static int slow_sym(const LaneBits& b, uint32_t root, const uint16_t* syms, const uint32_t* first,
                                               const uint16_t* cnt, const uint16_t* offs, uint32_t* len_out) {
  uint32_t code = __brev((uint32_t)b.bb & ((1u << root) - 1u)) >> (32u - root);
  for (uint32_t len = root + 1; len <= 15u; len++) {
    code = (code << 1) | (uint32_t)((b.bb >> (len - 1)) & 1u);
    const uint32_t f = first[len];
    const uint32_t rel = code - f;
    if (code >= f && rel < cnt[len]) {
      *len_out = len;
      return (int)syms[offs[len] + rel];
    }
  }
  return -1;
}

#define T_LIT 0u
#define T_EOB 1u
#define T_MATCH 2u
#define T_FAIL 3u
#define C_EOB 62u   // transfer-table codes besides exit offsets 0..47
#define C_FAIL 63u
#define C_NONE 255u

// One token at b.pos.  Returns its kind; literal value / (len, dist) through the references.
__device__ __forceinline__ static uint32_t tok_step(ParSmem& S, LaneBits& b, const uint32_t* in32, uint32_t lastdw, uint32_t& val,
                                                    uint32_t& len, uint32_t& dist) {
  lb_refill(b, in32, lastdw);
  uint32_t e = S.lut_l[(uint32_t)b.bb & ((1u << PL_ROOT) - 1u)];
  if ((e & 15u) == 0u) {
    uint32_t l2 = 0;
    const int sy = slow_sym(b, PL_ROOT, S.syms_l, S.first_l, S.cnt_l, S.offs_l, &l2);
    if (sy < 0) return T_FAIL;
    e = entry_l((uint32_t)sy, l2);
  }
  lb_take(b, e & 15u);
  const uint32_t kind = (e >> 8) & 3u;
  if (kind == 0u) {
    val = e >> 16;
    return T_LIT;
  }
  if (kind == 1u) return T_EOB;
  if (kind == 3u) return T_FAIL;
  lb_refill(b, in32, lastdw);
  len = (e >> 16) + lb_take(b, (e >> 4) & 15u);
  lb_refill(b, in32, lastdw);
  uint32_t ed = S.lut_d[(uint32_t)b.bb & ((1u << PD_ROOT) - 1u)];
  if ((ed & 15u) == 0u) {
    uint32_t l2 = 0;
    const int sy = slow_sym(b, PD_ROOT, S.syms_d, S.first_d, S.cnt_d, S.offs_d, &l2);
    if (sy < 0) return T_FAIL;
    ed = entry_d((uint32_t)sy, l2);
  }
  lb_take(b, ed & 15u);
  if (((ed >> 8) & 3u) != 2u) return T_FAIL;
  lb_refill(b, in32, lastdw);
  dist = (ed >> 16) + lb_take(b, (ed >> 4) & 15u);
  return T_MATCH;
}

// Transfer table of the segment [base, stop): tab[o] for o < 128 gets the exit code of the
// trajectory passing through token boundary base+o (only o < 48 can be entries; the rest of the
// window only detects merges early).
__device__ __forceinline__ static void seg_table(ParSmem& S, const uint32_t* in32, uint32_t lastdw, uint32_t limit, uint32_t base,
                                                 uint32_t stop, uint8_t* tab) {
  uint64_t V0 = 0, V1 = 0;  // visited boundaries, offsets 0..63 and 64..127
  for (uint32_t k = 0; k < 48u; k++) {
    if ((V0 >> k) & 1ull) continue;
    uint64_t M0 = 0, M1 = 0;
    uint32_t code;
    if (base + k >= limit) {
      code = C_FAIL;
      M0 = 1ull << k;
    } else {
      LaneBits b;
      lb_seek(b, in32, lastdw, base + k);
      for (;;) {
        if (b.pos >= stop) {
          code = b.pos - stop;  // 0..47: a token is at most 48 bits
          break;
        }
        const uint32_t off = b.pos - base;
        if (off < 64u) {
          if ((V0 >> off) & 1ull) {
            code = tab[off];
            break;
          }
          M0 |= 1ull << off;
        } else if (off < 128u) {
          if ((V1 >> (off - 64u)) & 1ull) {
            code = tab[off];
            break;
          }
          M1 |= 1ull << (off - 64u);
        }
        uint32_t v, l, d;
        const uint32_t kind = tok_step(S, b, in32, lastdw, v, l, d);
        if (kind == T_EOB) {
          code = C_EOB;
          break;
        }
        if (kind == T_FAIL || b.pos > limit) {
          code = C_FAIL;
          break;
        }
      }
    }
    V0 |= M0;
    V1 |= M1;
    while (M0) {
      const uint32_t o = (uint32_t)__builtin_ctzll(M0);
      M0 &= M0 - 1ull;
      tab[o] = (uint8_t)code;
    }
    while (M1) {
      const uint32_t o = (uint32_t)__builtin_ctzll(M1);
      M1 &= M1 - 1ull;
      tab[64u + o] = (uint8_t)code;
    }
  }
}

// Decodes tokens from bit `entry` while the token start is below `stop`.
template <bool EMIT>
__device__ __forceinline__ static void seg_decode(ParSmem& S, const uint32_t* in32, uint32_t lastdw, uint32_t limit, uint32_t entry,
                                                  uint32_t stop, uint32_t out_off, uint32_t& exit_pos, uint32_t& outbytes,
                                                  uint32_t& flags) {
  LaneBits b;
  lb_seek(b, in32, lastdw, entry);
  uint32_t ob = 0, fl = 0;
  while (b.pos < stop) {
    uint32_t v = 0, len = 0, dist = 0;
    const uint32_t kind = tok_step(S, b, in32, lastdw, v, len, dist);
    if (kind == T_EOB) {
      fl |= F_EOB;
      break;
    }
    if (kind == T_FAIL || b.pos > limit) {
      fl |= F_FAIL;
      break;
    }
    if (kind == T_LIT) {
      if (EMIT) S.out[out_off + ob] = (uint8_t)v;
      ob++;
    } else {
      if (EMIT) {
        const uint32_t p = out_off + ob;
        if (dist > p) {
          fl |= F_HIST;  // looks behind the block start: not a reference-made block
        } else {
          S.out[p] = (uint8_t)(dist - 1u);
          S.out[p + 1] = (uint8_t)((dist - 1u) >> 8);
          S.out[p + 2] = (uint8_t)(len - 3u);
          atomicOr(&S.bitmap[p >> 5], 1u << (p & 31u));
        }
      }
      ob += len;
    }
    if (ob > ZES_BLK) {  // more than a slot: not a reference-made block
      fl |= F_FAIL;
      break;
    }
  }
  exit_pos = b.pos;
  outbytes = ob;
  flags = fl;
}

// root table + canonical arrays of one alphabet (wave 0 only; lens at S.lens[base..base+nsym))
__device__ __forceinline__ static bool par_build(ParSmem& S, uint32_t base, uint32_t nsym, uint32_t root, bool is_dist, uint32_t* lut,
                                                 uint16_t* syms, uint32_t* first, uint16_t* cnt, uint16_t* offs) {
  const uint32_t lane = zes_lane();
  const uint8_t* lens = S.lens + base;
  for (uint32_t i = lane; i < (1u << root); i += 64) lut[i] = 0;
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
  uint32_t fst[16], ofs[16];
#pragma unroll
  for (int l = 1; l < 16; l++) {
    fst[l] = code;
    ofs[l] = off;
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
  for (uint32_t s0 = 0; s0 < nsym; s0 += 64) {
    const uint32_t s = s0 + lane;
    const uint32_t l = s < nsym ? lens[s] : 0u;
    if (l) {
      uint32_t rank = 0;
      for (uint32_t j = 0; j < s; j++) rank += (lens[j] == l);
      uint32_t f = 0, o2 = 0;
#pragma unroll
      for (int k = 1; k < 16; k++)
        if ((int)l == k) {
          f = fst[k];
          o2 = ofs[k];
        }
      syms[o2 + rank] = (uint16_t)s;
      if (l <= root) {
        const uint32_t rev = __brev(f + rank) >> (32u - l);
        const uint32_t ent = is_dist ? entry_d(s, l) : entry_l(s, l);
        for (uint32_t e = rev; e < (1u << root); e += 1u << l) lut[e] = ent;
      }
    }
  }
  return true;
}

// dynamic header by wave 0 (uniform): returns false on anything T2/T3 should look at
__device__ __forceinline__ static bool par_header(ParSmem& S, const uint32_t* in32, uint32_t lastdw, uint32_t limit, uint32_t start) {
  const uint32_t lane = zes_lane();
  LaneBits b;
  lb_seek(b, in32, lastdw, start);
  lb_refill(b, in32, lastdw);
  const uint32_t bfinal = lb_take(b, 1);
  if (lb_take(b, 2) != 2u) return false;
  const uint32_t HLIT = lb_take(b, 5) + 257u;
  const uint32_t HDIST = lb_take(b, 5) + 1u;
  const uint32_t HCLEN = lb_take(b, 4) + 4u;
  uint32_t mycl = 0;
  for (uint32_t k = 0; k < HCLEN; k++) {
    lb_refill(b, in32, lastdw);
    const uint32_t v = lb_take(b, 3);
    if (lane == kClOrder[k]) mycl = v;
  }
  for (uint32_t i = lane; i < 128; i += 64) S.cl_lut[i] = 0;
  uint32_t kraft = 0;
  {
    uint32_t code = 0;
    for (uint32_t l = 1; l <= 7; l++) {
      const bool mine = lane < 19 && mycl == l;
      const uint64_t m = __ballot(mine);
      if (mine) {
        const uint32_t rank = (uint32_t)__popcll(m & zes_lanemask_lt());
        const uint32_t rev = __brev(code + rank) >> (32u - l);
        for (uint32_t e = rev; e < 128; e += 1u << l) S.cl_lut[e] = (uint8_t)(lane | (l << 5));
      }
      const uint32_t n = (uint32_t)__popcll(m);
      kraft += n << (7 - l);
      code = (code + n) << 1;
    }
  }
  if (kraft > 128u) return false;
  for (uint32_t i = lane; i < 352; i += 64) S.lens[i] = 0;
  const uint32_t total = HLIT + HDIST;
  uint32_t prev = 0;
  for (uint32_t k = 0; k < total;) {
    lb_refill(b, in32, lastdw);
    const uint32_t e = S.cl_lut[(uint32_t)b.bb & 127u];
    const uint32_t l = e >> 5, sy = e & 31u;
    if (!l) return false;
    lb_take(b, l);
    uint32_t rep = 1, val = sy;
    if (sy == 16) {
      if (k == 0) return false;
      rep = 3 + lb_take(b, 2);
      val = prev;
    } else if (sy == 17) {
      rep = 3 + lb_take(b, 3);
      val = 0;
    } else if (sy == 18) {
      rep = 11 + lb_take(b, 7);
      val = 0;
    }
    if (k + rep > total) return false;
    if (val && lane < rep) {
      const uint32_t idx = k + lane;
      S.lens[idx < HLIT ? idx : 288 + (idx - HLIT)] = (uint8_t)val;
    }
    prev = val;
    k += rep;
  }
  if (b.pos > limit) return false;
  if (!par_build(S, 0, 288, PL_ROOT, false, S.lut_l, S.syms_l, S.first_l, S.cnt_l, S.offs_l)) return false;
  if (!par_build(S, 288, 32, PD_ROOT, true, S.lut_d, S.syms_d, S.first_d, S.cnt_d, S.offs_d)) return false;
  if (lane == 0) {
    S.hdr_end = b.pos;
    S.bfinal = bfinal;
  }
  return true;
}

__global__ __launch_bounds__(PAR_THREADS) void k_inf_block_par(const uint8_t* __restrict__ d_in, uint64_t in_off, uint64_t c,
                                                               uint8_t* __restrict__ d_out, uint64_t out_off, uint64_t cap,
                                                               const uint32_t* __restrict__ cand, const uint32_t* __restrict__ map,
                                                               uint32_t nwork, uint32_t ncand, ZesCandRes* __restrict__ cres,
                                                               unsigned long long* __restrict__ dbg) {
  __shared__ __align__(16) ParSmem S;
#define STAMP(i)                                                     \
  do {                                                               \
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); \
  } while (0)
  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (w >= nwork) return;
  const uint32_t* in32 = reinterpret_cast<const uint32_t*>(d_in + in_off);
  const uint32_t lastdw = (uint32_t)((c - 1) >> 2);
  const uint32_t limit = (uint32_t)(c * 8);
  const uint32_t ci = map ? map[w] : w;
  const uint32_t start = cand[ci] + 16u;
  // estimate of the block's end: the next candidate on the list (exact on a clean chain)
  uint32_t de_est = limit;
  if (map) {
    if (w + 1 < nwork) de_est = cand[map[w + 1]] + 16u;
  } else if (ci + 1 < ncand) {
    de_est = cand[ci + 1] + 16u;
  }

  STAMP(0);
  // ---- P0: header + tables (wave 0), bitmap clear (everyone else) ----
  if (tid == 0) S.status = 0;
  if (wave != 0) {
    for (uint32_t i = tid - 64; i < ZES_BLK / 32; i += PAR_THREADS - 64) S.bitmap[i] = 0;
  }
  __syncthreads();
  if (wave == 0) {
    const bool ok = par_header(S, in32, lastdw, limit, start);
    if (!ok && lane == 0) S.status = 1;
  }
  __syncthreads();
  if (S.status) {
    if (tid == 0) {
      ZesCandRes r;
      r.end_bit = start;
      r.out_len = 0;
      r.flags = 0;
      cres[w] = r;
    }
    return;
  }
  STAMP(1);
  const uint32_t ds = S.hdr_end;
  const uint32_t span = de_est > ds ? de_est - ds : 1u;
  const uint32_t seglen = max(64u, (span + PAR_THREADS - 1) / PAR_THREADS);

  // ---- P1: transfer tables (in the LDS image, 128 B per lane), then composition ----
  const uint64_t b_me64 = (uint64_t)ds + (uint64_t)tid * seglen;
  const uint32_t base = (uint32_t)(b_me64 < 0xFFFFFF00ull ? b_me64 : 0xFFFFFF00ull);
  const uint32_t stop = (uint32_t)((b_me64 + seglen) < 0xFFFFFF00ull ? (b_me64 + seglen) : 0xFFFFFF00ull);
  uint8_t* tab = S.out + (size_t)tid * 128u;
  seg_table(S, in32, lastdw, limit, base, stop, tab);
  __syncthreads();
  STAMP(2);
  if (lane < 48u) {  // composition over the 64 segments of this wave, one input offset per lane
    uint32_t cur = lane;
    for (uint32_t sgm = 0; sgm < 64u; sgm++) {
      if (cur < 48u) cur = S.out[(size_t)(wave * 64u + sgm) * 128u + cur];
    }
    S.wtab[wave][lane] = (uint8_t)cur;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t e = 0;  // segment 0 starts exactly at the header end
    for (uint32_t k = 0; k < PAR_WAVES; k++) {
      S.wentry[k] = (uint8_t)e;
      if (e < 48u) e = S.wtab[k][e];
    }
    S.tail_entry = e;  // state after the last segment: offset past its end, or EOB / fail
  }
  __syncthreads();
  if (lane == 0) {  // entries of the wave's 64 segments
    uint32_t e = S.wentry[wave];
    for (uint32_t sgm = 0; sgm < 64u; sgm++) {
      S.lentry[wave * 64u + sgm] = (uint8_t)e;
      if (e < 48u) e = S.out[(size_t)(wave * 64u + sgm) * 128u + e];
    }
  }
  __syncthreads();
  const uint32_t ecode = S.lentry[tid];
  const uint32_t tail_code = S.tail_entry;
  __syncthreads();  // the tables are dead from here on: the image may be written
  STAMP(3);

  // ---- P2: count pass from the true entries, totals, end bit, output offsets ----
  uint32_t entry = base + ecode, exit_pos = 0, outbytes = 0, flags = F_VOID;
  if (ecode < 48u) seg_decode<false>(S, in32, lastdw, limit, entry, stop, 0, exit_pos, outbytes, flags);
  if (tid == 0) {
    S.tail_bytes = 0;
    S.tail_end = 0;
  }
  __syncthreads();
  // a chain that is still alive after the last segment (the end estimate was short: a false
  // candidate sits inside this block) is finished serially by one lane
  uint32_t tail_start = 0;
  if (tail_code < 48u) {
    const uint64_t last_stop = (uint64_t)ds + (uint64_t)PAR_THREADS * seglen;
    tail_start = (uint32_t)(last_stop + tail_code);
    if (tid == 0) {
      uint32_t ex, ob, fl;
      seg_decode<false>(S, in32, lastdw, limit, tail_start, 0xFFFFFF00u, 0, ex, ob, fl);
      S.tail_bytes = ob;
      S.tail_end = ex;
      if (!(fl & F_EOB) || (fl & F_FAIL)) atomicOr(&S.status, 2u);
      else atomicOr(&S.status, 4u);
    }
  } else if (tail_code == C_FAIL) {
    if (tid == 0) atomicOr(&S.status, 2u);
  }
  {
    uint32_t incl = outbytes;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint32_t t = __shfl_up(incl, dlt);
      if ((int)lane >= dlt) incl += t;
    }
    if (lane == 63) S.wave_sum[wave] = incl;
    const uint64_t eobm = __ballot((flags & F_EOB) && !(flags & F_VOID));
    const uint64_t failm = __ballot((flags & F_FAIL) && !(flags & F_VOID));
    if (failm && lane == 0) atomicOr(&S.status, 2u);
    if (eobm && lane == (uint32_t)__builtin_ctzll(eobm)) {
      atomicOr(&S.status, 4u);
      S.hdr_end = exit_pos;  // reuse: end bit of the block
    }
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < PAR_WAVES; k++) {
      const uint32_t sm = S.wave_sum[k];
      if (k < wave) wbase += sm;
      total += sm;
    }
    const uint32_t seg_total = total;
    total += S.tail_bytes;
    const uint32_t st = S.status;
    const uint32_t end_bit = (tail_code < 48u) ? S.tail_end : S.hdr_end;
    const bool good = (st & 4u) && !(st & 2u) && total <= ZES_BLK;
    if (!good) {
      if (tid == 0) {
        ZesCandRes r;
        r.end_bit = end_bit;
        r.out_len = total;
        r.flags = 0;
        cres[w] = r;
      }
      return;
    }
    const uint32_t my_off = wbase + incl - outbytes;
    STAMP(4);

    // ---- P3: emit ----
    uint32_t f2 = 0;
    if (!(flags & F_VOID) && entry < stop) {
      uint32_t ex2, ob2;
      seg_decode<true>(S, in32, lastdw, limit, entry, stop, my_off, ex2, ob2, f2);
    }
    if (tid == 0 && tail_code < 48u) {
      uint32_t ex2, ob2, f3 = 0;
      seg_decode<true>(S, in32, lastdw, limit, tail_start, 0xFFFFFF00u, seg_total, ex2, ob2, f3);
      f2 |= f3;
    }
    if (f2 & F_HIST) atomicOr(&S.status, 8u);
    __syncthreads();
    if (S.status & 8u) {
      if (tid == 0) {
        ZesCandRes r;
        r.end_bit = end_bit;
        r.out_len = total;
        r.flags = 0;
        cres[w] = r;
      }
      return;
    }

    STAMP(5);
    // ---- P4: match resolution, in order, by wave 0 ----
    if (wave == 0) {
      for (uint32_t wb = 0; wb < total; wb += 2048) {
        const uint32_t wi = (wb >> 5) + lane;
        uint32_t bits = (wi < ZES_BLK / 32) ? S.bitmap[wi] : 0u;
        const uint32_t cntb = (uint32_t)__popc(bits);
        uint32_t inc2 = cntb;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
          const uint32_t t = __shfl_up(inc2, dlt);
          if ((int)lane >= dlt) inc2 += t;
        }
        const uint32_t nwin = __shfl(inc2, 63);
        uint32_t slot = inc2 - cntb;
        while (bits) {
          const uint32_t t = (uint32_t)__builtin_ctz(bits);
          bits &= bits - 1u;
          S.mlist[slot++] = (uint16_t)((lane << 5) + t);
        }
        for (uint32_t k = 0; k < nwin; k += 64) {
          const uint32_t i = k + lane;
          const bool have = i < nwin;
          uint32_t p = 0, L = 0, D = 1;
          if (have) {
            p = wb + S.mlist[i];
            D = ((uint32_t)S.out[p] | ((uint32_t)S.out[p + 1] << 8)) + 1u;
            L = (uint32_t)S.out[p + 2] + 3u;
          }
          const uint32_t srcend = min(p - D + L, p);
          uint64_t U = __ballot(have);
          while (U) {
            const uint32_t f = (uint32_t)__builtin_ctzll(U);
            const uint32_t pf = __shfl(p, (int)f);
            const bool ready = have && ((U >> lane) & 1ull) && srcend <= pf;
            if (ready) {
              const uint32_t src = p - D;
              for (uint32_t j = 0; j < L; j++) S.out[p + j] = S.out[src + j];
            }
            U &= ~__ballot(ready);
          }
        }
      }
    }
    __syncthreads();

    STAMP(6);
    // ---- P5: flush ----
    const uint64_t slot_off = (uint64_t)w * ZES_BLK;
    uint8_t* dst = d_out + out_off + slot_off;
    const uint64_t room = cap > slot_off ? cap - slot_off : 0;
    const uint32_t nstore = (uint32_t)min((uint64_t)total, room);
    const uint32_t full = nstore >> 4;
    for (uint32_t i = tid; i < full; i += PAR_THREADS)
      reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(S.out)[i];
    for (uint32_t i = (full << 4) + tid; i < nstore; i += PAR_THREADS) dst[i] = S.out[i];
    if (tid == 0) {
      ZesCandRes r;
      r.end_bit = end_bit;
      r.out_len = total;
      r.flags = 1u | (S.bfinal ? 2u : 0u);
      cres[w] = r;
    }
    STAMP(7);
  }
}
